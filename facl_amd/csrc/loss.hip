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


// ---------------------------------------------------------------------------------------------------------------------
// Both losses of a step in ONE launch on ONE similarity matrix.  sim is ((G+1)*B, J) = [x ; x_global] @ keys^T: row
// block g < G holds view g of the local clips (row g*B + n), block G the clip-level embeddings x_global.  Workgroups
// 0..B-1 evaluate the global loss of clip n (anchor block G, one positive per view g in column g*Bk + clip), workgroups
// B..2B-1 the circle loss (anchor blocks order[0..G-2], positive of slot i in block order[i], column order[i+1]*Bk + clip;
// block order[G-1] is no anchor: its dsim row is zeroed here so that every row of dsim is written exactly once).
// The anchors gather, the positive-column index tensors and the second similarity GEMM of the two-call form disappear.
struct PairSpec {
    int nA, nS, slot_rows, circle, G, B, Bk, J, myclip, n;
    const long long* order;
    __device__ __forceinline__ int ord(int i) const {            // clamped: a corrupt entry cannot address outside sim / dsim
        const long long o = order[i];
        return o < 0 ? 0 : (o >= G ? G - 1 : (int)o);
    }
    __device__ __forceinline__ int row_block(int i) const { return circle ? ord(i) : G; }
    __device__ __forceinline__ size_t pos_index(int s) const {
        const int rb = row_block(slot_rows ? s : 0);
        const int col = (circle ? ord(s + 1) : s) * Bk + myclip;
        return (size_t)(rb * B + n) * J + col;
    }
};

__global__ __launch_bounds__(1024) void k_contrast_pair_reg(const float* __restrict__ sim, int G, int B, int Bk, int J,
                                                           const long long* __restrict__ order, int clip_offset,
                                                           float* __restrict__ dsim, double* __restrict__ part) {
    __shared__ float smf[16];
    __shared__ double smd[16];
    __shared__ int rb_s[1024];
    PairSpec sp;
    sp.circle = blockIdx.x >= (unsigned)B;
    sp.n = sp.circle ? blockIdx.x - B : blockIdx.x;
    sp.nA = sp.circle ? G - 1 : 1; sp.nS = sp.circle ? G - 1 : G; sp.slot_rows = sp.circle;
    sp.G = G; sp.B = B; sp.Bk = Bk; sp.J = J; sp.myclip = sp.n + clip_offset; sp.order = order;
    const int n = sp.n, myclip = sp.myclip, nA = sp.nA, nS = sp.nS;
    if ((int)threadIdx.x < nA) rb_s[threadIdx.x] = sp.row_block(threadIdx.x);
    __syncthreads();
    const int total = nA * J;
    // Element e = tid + 1024 k of the (nA x J) block: row i = e / J, column j = e % J, and j % Bk for the same-clip mask.  The
    // divisions are taken ONCE per thread and advanced by increments (one wrap at most per step: 1024 % J < J): with a division
    // pair per element and pass -- 4 x 18 runtime divisions of ~35 instructions each -- this kernel was VALU-bound on its index
    // arithmetic (23.5 us on 64 workgroups); the offsets are kept for the write pass (bit 31 = masked column).
    const int di = 1024 / J, dj = 1024 - di * J, djm = dj % Bk, Jm = J % Bk;
    int wi = (int)threadIdx.x / J, wj = (int)threadIdx.x - wi * J, wjm = wj % Bk;
    float v[CK];
    unsigned off[CK];
    float mx = 0.f;                                        // masked entries are 0, there is at least one
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        const int e = threadIdx.x + k * 1024;
        float x = 0.f;                                     // beyond the end: behaves like a masked column
        off[k] = 0x80000000u;
        if (e < total) {
            const unsigned o = (unsigned)((rb_s[wi] * B + n) * J + wj);
            const bool masked = wjm == myclip;
            off[k] = masked ? (o | 0x80000000u) : o;
            if (!masked) x = sim[o];
        }
        v[k] = x;
        mx = fmaxf(mx, x);
        wi += di; wj += dj; wjm += djm;
        if (wj >= J) { wj -= J; ++wi; wjm += Bk - Jm; }
        if (wjm >= Bk) wjm -= Bk;
        if (wjm >= Bk) wjm -= Bk;
    }
    // the positives' similarities do not depend on the log-sum-exp: requested with the block, not behind two reductions
    float pos = 0.f;
    size_t ppos = 0;
    const bool has_pos = (int)threadIdx.x < nS;            // nS <= 1024: at most one per thread
    if (has_pos) { ppos = sp.pos_index(threadIdx.x); pos = sim[ppos]; }
    mx = block_reduce_max(mx, smf);
    double se = 0;
#pragma unroll
    for (int k = 0; k < CK; ++k)
        if ((int)threadIdx.x + k * 1024 < total) se += (double)__expf(v[k] - mx);
    se = block_reduce_sum(se, smd);
    const float lse = mx + (float)log(se);
    double loss_t = 0, dlse_t = 0;
    float dpos = 0.f;
    const float invB = 1.f / (float)B;
    if (has_pos) {
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
        if (e < total) dsim[off[k] & 0x7fffffffu] = (off[k] >> 31) ? 0.f : dlse * __expf(v[k] - lse);
    }
    if (sp.circle) {                                       // the view that is no anchor: zero gradient row
        float* z = dsim + (size_t)(sp.ord(G - 1) * B + n) * J;
        for (int j = threadIdx.x; j < J; j += 1024) z[j] = 0.f;
    }
    __syncthreads();                                       // the positives overwrite zeros written just above
    if ((int)threadIdx.x < nS) dsim[ppos] = dpos;
    if (threadIdx.x == 0) part[2 * n + sp.circle] = loss * (double)invB;
}

// streaming form for shapes the register-cached kernel cannot hold ((G-1)*J > CK*1024 or G > 1024)
__global__ __launch_bounds__(256) void k_contrast_pair(const float* __restrict__ sim, int G, int B, int Bk, int J,
                                                       const long long* __restrict__ order, int clip_offset,
                                                       float* __restrict__ dsim, double* __restrict__ part) {
    __shared__ float smf[4];
    __shared__ double smd[4];
    PairSpec sp;
    sp.circle = blockIdx.x >= (unsigned)B;
    sp.n = sp.circle ? blockIdx.x - B : blockIdx.x;
    sp.nA = sp.circle ? G - 1 : 1; sp.nS = sp.circle ? G - 1 : G; sp.slot_rows = sp.circle;
    sp.G = G; sp.B = B; sp.Bk = Bk; sp.J = J; sp.myclip = sp.n + clip_offset; sp.order = order;
    const int n = sp.n, myclip = sp.myclip, nA = sp.nA, nS = sp.nS;
    float mx = 0.f;
    for (int i = 0; i < nA; ++i) {
        const float* row = sim + (size_t)(sp.row_block(i) * B + n) * J;
        for (int j = threadIdx.x; j < J; j += 256) mx = fmaxf(mx, (j % Bk == myclip) ? 0.f : row[j]);
    }
    mx = block_reduce_max(mx, smf);
    double se = 0;
    for (int i = 0; i < nA; ++i) {
        const float* row = sim + (size_t)(sp.row_block(i) * B + n) * J;
        for (int j = threadIdx.x; j < J; j += 256) se += (double)__expf(((j % Bk == myclip) ? 0.f : row[j]) - mx);
    }
    se = block_reduce_sum(se, smd);
    const float lse = mx + (float)log(se);
    float dlse = 0.f;
    double loss = 0;
    for (int s = 0; s < nS; ++s) {
        const float pos = sim[sp.pos_index(s)];
        const float t = fmaxf(pos, lse) + log1pf(__expf(-fabsf(pos - lse)));
        loss += (double)(t - pos);
        dlse += (1.f - __expf(pos - t));
    }
    const float invB = 1.f / (float)B;
    dlse *= invB;
    for (int i = 0; i < nA; ++i) {
        const size_t ro = (size_t)(sp.row_block(i) * B + n) * J;
        for (int j = threadIdx.x; j < J; j += 256) dsim[ro + j] = (j % Bk == myclip) ? 0.f : dlse * __expf(sim[ro + j] - lse);
    }
    if (sp.circle) {
        float* z = dsim + (size_t)(sp.ord(G - 1) * B + n) * J;
        for (int j = threadIdx.x; j < J; j += 256) z[j] = 0.f;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nS; s += 256) {
        const size_t pp = sp.pos_index(s);
        const float pos = sim[pp];
        const float t = fmaxf(pos, lse) + log1pf(__expf(-fabsf(pos - lse)));
        dsim[pp] = (__expf(pos - t) - 1.f) * invB;
    }
    if (threadIdx.x == 0) part[2 * n + sp.circle] = loss * (double)invB;
}

// dst[r][:] = src[r][:] * (r < R1 ? *g1 : *g2): the chain rule of the two loss values onto the shared d/dsim matrix
template <typename V>
__global__ __launch_bounds__(256) void k_scale_rows2(const V* __restrict__ src, V* __restrict__ dst, long long n,
                                                     long long split, const float* __restrict__ g1,
                                                     const float* __restrict__ g2) {
    const float a = *g1, b = *g2;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float g = i < split ? a : b;
        if constexpr (sizeof(V) == 16) {
            float4 v = src[i];
            v.x *= g; v.y *= g; v.z *= g; v.w *= g;
            dst[i] = v;
        } else {
            dst[i] = src[i] * g;
        }
    }
}

}  // namespace

extern "C" int facl_contrast(const float* sim, int R, int J, int B, int Bk, int nA, int nS, int slot_rows,
                             const int32_t* poscol, int clip_offset, float* dsim, double* loss, void* ws,
                             void* stream) {
    if (!sim || !poscol || !dsim || !loss || !ws) return FACL_E_NULL;
    if (B < 1 || nA < 1 || nS < 1 || R != nA * B || J < 1 || Bk < 1 || J % Bk) return FACL_E_SHAPE;
    if (slot_rows && nS != nA) return FACL_E_SHAPE;
    if (clip_offset < 0 || clip_offset + B > Bk) return FACL_E_SHAPE;        // the local clips must be columns of the keys
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

extern "C" int facl_contrast_pair(const float* sim, int G, int B, int Bk, int J, const int64_t* order, int clip_offset,
                                  float* dsim, double* losses /* [loss_c, loss_circle] */, void* ws, void* stream) {
    if (!sim || !order || !dsim || !losses || !ws) return FACL_E_NULL;
    if (G < 2 || B < 1 || Bk < 1 || J != G * Bk) return FACL_E_SHAPE;
    if (clip_offset < 0 || clip_offset + B > Bk) return FACL_E_SHAPE;        // the local clips must be columns of the keys
    hipStream_t st = (hipStream_t)stream;
    if ((long long)(G - 1) * J <= (long long)CK * 1024 && G <= 1024 && (long long)(G + 1) * B * J < 0x7fffffffLL)
        hipLaunchKernelGGL(k_contrast_pair_reg, dim3(2 * B), dim3(1024), 0, st, sim, G, B, Bk, J, (const long long*)order,
                           clip_offset, dsim, (double*)ws);
    else
        hipLaunchKernelGGL(k_contrast_pair, dim3(2 * B), dim3(256), 0, st, sim, G, B, Bk, J, (const long long*)order,
                           clip_offset, dsim, (double*)ws);
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, B, 2, losses, st);
}

// facl_contrast_pair whose final reduction also writes the fp32 values the training loop works with: losses32 =
// [loss_c, loss_circle, loss_circle + loss_c] (the sum in fp32, exactly the reference's `loss = loss_circle + loss_c` on the two
// fp32 losses, cn3d_train_motion_GL.py:329) -- one single-workgroup launch instead of a reduction, a dtype cast and an add.
namespace {
__global__ __launch_bounds__(64) void k_loss_finish(const double* __restrict__ part, int B, double* __restrict__ losses,
                                                    float* __restrict__ losses32) {
    double c = 0, o = 0;
    for (int n = threadIdx.x; n < B; n += 64) { c += part[2 * n]; o += part[2 * n + 1]; }
    c = wave_sum_f64(c);
    o = wave_sum_f64(o);
    if (threadIdx.x == 0) {
        losses[0] = c; losses[1] = o;
        const float fc = (float)c, fo = (float)o;
        losses32[0] = fc; losses32[1] = fo; losses32[2] = fo + fc;
    }
}
}  // namespace

extern "C" int facl_contrast_pair_sum(const float* sim, int G, int B, int Bk, int J, const int64_t* order, int clip_offset,
                                      float* dsim, double* losses, float* losses32, void* ws, void* stream) {
    if (!sim || !order || !dsim || !losses || !losses32 || !ws) return FACL_E_NULL;
    if (G < 2 || B < 1 || Bk < 1 || J != G * Bk) return FACL_E_SHAPE;
    if (clip_offset < 0 || clip_offset + B > Bk) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if ((long long)(G - 1) * J <= (long long)CK * 1024 && G <= 1024 && (long long)(G + 1) * B * J < 0x7fffffffLL)
        hipLaunchKernelGGL(k_contrast_pair_reg, dim3(2 * B), dim3(1024), 0, st, sim, G, B, Bk, J, (const long long*)order,
                           clip_offset, dsim, (double*)ws);
    else
        hipLaunchKernelGGL(k_contrast_pair, dim3(2 * B), dim3(256), 0, st, sim, G, B, Bk, J, (const long long*)order,
                           clip_offset, dsim, (double*)ws);
    int rc = facl_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(64), 0, st, (const double*)ws, B, losses, losses32);
    return facl_launch_status();
}

extern "C" int facl_scale_rows2(const float* src, float* dst, int64_t R1, int64_t R, int J, const float* g1, const float* g2,
                                void* stream) {
    if (!src || !dst || !g1 || !g2) return FACL_E_NULL;
    if (R1 < 0 || R1 > R || R < 1 || J < 1) return FACL_E_SHAPE;
    const long long n = R * (long long)J, split = R1 * (long long)J;
    hipStream_t st = (hipStream_t)stream;
    if (!(n & 3) && !(split & 3) && !((((uintptr_t)src) | ((uintptr_t)dst)) & 15)) {
        const long long n4 = n / 4;
        const int grid = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
        hipLaunchKernelGGL((k_scale_rows2<float4>), dim3(grid), dim3(256), 0, st, (const float4*)src, (float4*)dst, n4, split / 4, g1, g2);
    } else {
        const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
        hipLaunchKernelGGL((k_scale_rows2<float>), dim3(grid), dim3(256), 0, st, src, dst, n, split, g1, g2);
    }
    return facl_launch_status();
}
