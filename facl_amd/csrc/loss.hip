// Global / circle InfoNCE-style losses on a similarity matrix (utils_my.py:53-116 =
// cn3d_train_motion_GL.py:265-316), forward value and d(loss)/d(sim) in one kernel.
//
// sim is (R, J) = anchors @ keys^T with R = nA*B anchor rows (row r = i*B + n: anchor slot i of clip n;
// nA = 1 for the global loss, G-1 for the circle loss) and J = G*Bk key columns (column j belongs to
// clip j % Bk).  Reference semantics:
//   * same-clip columns are MULTIPLIED BY 0 (utils_my.py:72,106), i.e. they stay in the softmax as exp(0);
//   * all nA anchor slots of a clip share ONE negative set (the `repeat` at :74 / :108), so
//     lse[n] = log sum_{i,j} exp(sim'[i*B+n, j]);
//   * logits[i] = [pos[i,n] | negatives], label 0, CE = mean over the B clips, summed over i.
// pos[i,n] = sim[i*B+n, poscol[i*B+n]] (read BEFORE masking: the positive key is a same-clip column).
// Per clip: loss_n = sum_i (logaddexp(pos, lse) - pos) / B;   dsim follows by the chain rule.
// HBM-bound (2 reads + 1 write of R*J floats); one workgroup per clip.
#include "common.h"

int facl_reduce_rows(const double* part, int rows, int V, double* out, hipStream_t st);

namespace {

__device__ __forceinline__ float block_reduce_max(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if (lane_id() == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sm[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, sm[w]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ double block_reduce_sum(double v, double* sm) {
    v = wave_sum_f64(v);
    if (lane_id() == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sm[w];
    __syncthreads();
    return r;
}

// nA = similarity rows per clip that feed the shared negative set (row i*B+n); nS = positive slots per clip;
// slot s reads its positive from row (slot_rows ? s : 0)*B+n, column poscol[s*B+n].
//   circle: nA = nS = G-1, slot_rows = 1.      global: nA = 1, nS = G, slot_rows = 0.
__global__ __launch_bounds__(256) void k_contrast(const float* __restrict__ sim, int J, int B, int Bk, int nA, int nS,
                                                  int slot_rows, const int* __restrict__ poscol, int clip_offset,
                                                  float* __restrict__ dsim, double* __restrict__ part) {
    __shared__ float smf[4];
    __shared__ double smd[4];
    const int n = blockIdx.x;
    const int myclip = n + clip_offset;
    // pass 1: lse over all anchor slots of this clip (masked columns count as exp(0))
    float mx = 0.f;                                        // masked entries are 0, there is at least one
    for (int i = 0; i < nA; ++i) {
        const float* row = sim + (size_t)(i * B + n) * J;
        for (int j = threadIdx.x; j < J; j += 256) {
            const float v = (j % Bk == myclip) ? 0.f : row[j];
            mx = fmaxf(mx, v);
        }
    }
    mx = block_reduce_max(mx, smf);
    double se = 0;
    for (int i = 0; i < nA; ++i) {
        const float* row = sim + (size_t)(i * B + n) * J;
        for (int j = threadIdx.x; j < J; j += 256) {
            const float v = (j % Bk == myclip) ? 0.f : row[j];
            se += (double)__expf(v - mx);
        }
    }
    se = block_reduce_sum(se, smd);
    const float lse = mx + (float)log(se);
    // per-slot terms: t = logaddexp(pos, lse); loss += t - pos; dpos = (sigma - 1)/B; dlse += (1 - sigma)/B
    float dlse = 0.f;
    double loss = 0;
    for (int s = 0; s < nS; ++s) {
        const int r = (slot_rows ? s : 0) * B + n;
        const float pos = sim[(size_t)r * J + poscol[s * B + n]];
        const float m2 = fmaxf(pos, lse);
        const float t = m2 + log1pf(__expf(-fabsf(pos - lse)));
        loss += (double)(t - pos);
        dlse += (1.f - __expf(pos - t));
    }
    const float invB = 1.f / (float)B;
    dlse *= invB;
    // pass 2: negatives' gradient (same-clip columns are constants: 0)
    for (int i = 0; i < nA; ++i) {
        const float* row = sim + (size_t)(i * B + n) * J;
        float* drow = dsim + (size_t)(i * B + n) * J;
        for (int j = threadIdx.x; j < J; j += 256)
            drow[j] = (j % Bk == myclip) ? 0.f : dlse * __expf(row[j] - lse);
    }
    __syncthreads();
    // positives: distinct (row, column) per slot, all inside this clip's masked columns
    for (int s = threadIdx.x; s < nS; s += 256) {
        const int r = (slot_rows ? s : 0) * B + n;
        const int pc = poscol[s * B + n];
        const float pos = sim[(size_t)r * J + pc];
        const float m2 = fmaxf(pos, lse);
        const float t = m2 + log1pf(__expf(-fabsf(pos - lse)));
        dsim[(size_t)r * J + pc] = (__expf(pos - t) - 1.f) * invB;
    }
    if (threadIdx.x == 0) part[n] = loss * (double)invB;
}

// The same computation with 1024 threads per clip and the clip's nA x J similarities held in registers between the
// passes (one read of sim, no per-element integer modulo in the inner passes): the grid is only B workgroups, so the
// kernel is latency-bound and three streaming passes with 256 threads took 65 us for the circle loss at H.
constexpr int CK = 24;                      // cached values per thread: nA*J <= 24*1024

__global__ __launch_bounds__(1024) void k_contrast_reg(const float* __restrict__ sim, int J, int B, int Bk, int nA, int nS,
                                                      int slot_rows, const int* __restrict__ poscol, int clip_offset,
                                                      float* __restrict__ dsim, double* __restrict__ part) {
    __shared__ float smf[16];
    __shared__ double smd[16];
    const int n = blockIdx.x;
    const int myclip = n + clip_offset;
    const int total = nA * J;
    float v[CK];
    float mx = 0.f;                                        // masked entries are 0, there is at least one
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        const int e = threadIdx.x + k * 1024;
        float x = 0.f;                                     // beyond the end: behaves like a masked column
        if (e < total) {
            const int i = e / J, j = e - i * J;
            x = (j % Bk == myclip) ? 0.f : sim[(size_t)(i * B + n) * J + j];
        }
        v[k] = x;
        mx = fmaxf(mx, x);
    }
    mx = block_reduce_max(mx, smf);
    double se = 0;
#pragma unroll
    for (int k = 0; k < CK; ++k)
        if (threadIdx.x + k * 1024 < total) se += (double)__expf(v[k] - mx);
    se = block_reduce_sum(se, smd);
    const float lse = mx + (float)log(se);
    // per-slot terms, one slot per thread (the serial loop of k_contrast is nS dependent load pairs per thread)
    double loss_t = 0, dlse_t = 0;
    float dpos = 0.f;
    size_t ppos = 0;
    const float invB = 1.f / (float)B;
    for (int s = threadIdx.x; s < nS; s += 1024) {       // nS <= 1024 in every use: at most one trip per thread
        const int r = (slot_rows ? s : 0) * B + n;
        ppos = (size_t)r * J + poscol[s * B + n];
        const float pos = sim[ppos];
        const float m2 = fmaxf(pos, lse);
        const float t = m2 + log1pf(__expf(-fabsf(pos - lse)));
        loss_t += (double)(t - pos);
        dlse_t += (double)(1.f - __expf(pos - t));
        dpos = (__expf(pos - t) - 1.f) * invB;
    }
    const double loss = block_reduce_sum(loss_t, smd);
    const float dlse = (float)block_reduce_sum(dlse_t, smd) * invB;
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        const int e = threadIdx.x + k * 1024;
        if (e < total) {
            const int i = e / J, j = e - i * J;
            dsim[(size_t)(i * B + n) * J + j] = (j % Bk == myclip) ? 0.f : dlse * __expf(v[k] - lse);
        }
    }
    __syncthreads();                                       // the positives overwrite zeros written just above
    if (threadIdx.x < nS) dsim[ppos] = dpos;
    if (threadIdx.x == 0) part[n] = loss * (double)invB;
}

}  // namespace

extern "C" int facl_contrast(const float* sim, int R, int J, int B, int Bk, int nA, int nS, int slot_rows,
                             const int32_t* poscol, int clip_offset, float* dsim, double* loss, void* ws,
                             void* stream) {
    if (!sim || !poscol || !dsim || !loss || !ws) return FACL_E_NULL;
    if (B < 1 || nA < 1 || nS < 1 || R != nA * B || J < 1 || Bk < 1 || J % Bk) return FACL_E_SHAPE;
    if (slot_rows && nS != nA) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if ((long long)nA * J <= (long long)CK * 1024 && nS <= 1024)
        hipLaunchKernelGGL(k_contrast_reg, dim3(B), dim3(1024), 0, st, sim, J, B, Bk, nA, nS, slot_rows, poscol, clip_offset,
                           dsim, (double*)ws);
    else
        hipLaunchKernelGGL(k_contrast, dim3(B), dim3(256), 0, st, sim, J, B, Bk, nA, nS, slot_rows, poscol, clip_offset,
                           dsim, (double*)ws);
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, B, 1, loss, st);
}
