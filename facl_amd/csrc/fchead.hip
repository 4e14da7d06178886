// netR_FC's BatchNorm1d + ReLU over the STACKED rows [x_pre ; gobaol_max_pool(x_pre)] (cn3d_model_conbag.py:201-207, :228-229).
// The reference calls netR_FC twice -- on the G*B view rows and on the B clip rows -- so its BatchNorm1d takes two batch
// statistics and updates its running buffers twice, view rows first.  facl_amd runs each Linear layer once on the stacked
// (G*B + B) rows (facl_amd/tail.py: _FCHead); the kernels here are that BatchNorm over the two ROW SEGMENTS [0, M) and
// [M, R) of one tensor, forward and backward, in three launches each instead of the nine / eleven single-segment ones
// (statistics, two-level reductions, finalisation, apply -- twice -- plus the torch adds of the two gamma / beta gradients):
//   forward : k_fc_stats (fp64 row-slice statistics; the GEMM epilogue's fp32 sums lose the 4..32-row clip segment's variance)
//             -> k_fc_finalize (both segments, the running statistics updated twice in the reference's order)
//             -> k_fc_apply    (relu(bn(y)), the segment's constants picked per row)
//   backward: k_fc_bwd_stats -> k_fc_bwd_consts (k1, k2 per segment; dgamma / dbeta = the sum of both segments') -> k_fc_bwd_apply
// Row-slice partials are small (R / 32 rows), so the kernel that needs a column total adds the slices itself: no reduction
// launch.  Under data parallelism the slices are first summed into a (2, C, 2) tensor for the SyncBN all-reduce
// (facl_fc_bn_stats / facl_fc_bn_bwd_stats with `sums2`), and the finalisation reads that tensor as one slice per segment.
// All HBM-trivial (800 x 1024 floats): these kernels are launch-latency-bound, which is why there are few of them.
#include "common.h"

namespace {

constexpr int FC_SL = 32;                  // rows per statistics slice

// slice s covers rows [r0, r1) of segment seg: na slices tile [0, M), the rest tile [M, R)
__device__ __forceinline__ void fc_slice(int s, int M, int R, int na, int& r0, int& r1, int& seg) {
    if (s < na) { seg = 0; r0 = s * FC_SL; r1 = r0 + FC_SL < M ? r0 + FC_SL : M; }
    else { seg = 1; r0 = M + (s - na) * FC_SL; r1 = r0 + FC_SL < R ? r0 + FC_SL : R; }
    if (r0 > r1) r0 = r1;
}

// column (sum, sumsq) of each slice: block = 64 channel quads x 4 row phases, combined through LDS in phase order
__global__ __launch_bounds__(256) void k_fc_stats(const float* __restrict__ y, int M, int R, int na, int C4,
                                                  double* __restrict__ part) {
    __shared__ double red[3][64][8];
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + lane;
    int r0, r1, seg;
    fc_slice(blockIdx.y, M, R, na, r0, r1, seg);
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c4 < C4)
        for (int r = r0 + ph; r < r1; r += 4) {
            const float4 v = reinterpret_cast<const float4*>(y)[(size_t)r * C4 + c4];
            acc[0] += (double)v.x; acc[1] += (double)v.x * (double)v.x;
            acc[2] += (double)v.y; acc[3] += (double)v.y * (double)v.y;
            acc[4] += (double)v.z; acc[5] += (double)v.z * (double)v.z;
            acc[6] += (double)v.w; acc[7] += (double)v.w * (double)v.w;
        }
    if (ph > 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[ph - 1][lane][e] = acc[e];
    }
    __syncthreads();
    if (ph == 0 && c4 < C4) {
        double* pr = part + (size_t)blockIdx.y * 8 * C4 + 8 * c4;
#pragma unroll
        for (int e = 0; e < 8; ++e) pr[e] = ((acc[e] + red[0][lane][e]) + red[1][lane][e]) + red[2][lane][e];
    }
}

// slices -> sums2 (2, C, 2): the tensor the SyncBN all-reduce takes (slices added in order)
__global__ void k_fc_reduce(const double* __restrict__ part, int na, int nb, int V, double* __restrict__ sums2) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    double a = 0, b = 0;
    for (int s = 0; s < na; ++s) a += part[(size_t)s * V + v];
    for (int s = na; s < na + nb; ++s) b += part[(size_t)s * V + v];
    sums2[v] = a;
    sums2[V + v] = b;
}

// Both segments' train-mode BatchNorm constants bnc2 (2, 5, C) = (mean, invstd, scale, shift, sign) from the slice sums,
// and the module's running statistics updated TWICE, segment a (the view rows, :228) first, segment b (the clip rows,
// :229) second: the same arithmetic as two facl_bn_finalize calls (the buffers round to fp32 in between).
__global__ __launch_bounds__(256) void k_fc_finalize(const double* __restrict__ part, int na, int nb, int C, double count_a,
                                                     double count_b, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, float momentum, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, float* __restrict__ bnc2) {
    // block = 64 channels x 4 slice phases (phase p adds slices p, p + 4, ...; the four partial sums meet in LDS in phase
    // order): a serial walk over the ~26 slices by one thread per channel was 11 us of pure load latency
    __shared__ double red[3][64][4];
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int V = 2 * C;
    double acc[4] = {0, 0, 0, 0};                                  // (sum, sumsq) of segment a, of segment b
    if (c < C) {
        for (int t = ph; t < na; t += 4) { acc[0] += part[(size_t)t * V + 2 * c]; acc[1] += part[(size_t)t * V + 2 * c + 1]; }
        for (int t = na + ph; t < na + nb; t += 4) { acc[2] += part[(size_t)t * V + 2 * c]; acc[3] += part[(size_t)t * V + 2 * c + 1]; }
    }
    if (ph > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[ph - 1][lane][e] = acc[e];
    }
    __syncthreads();
    if (ph != 0 || c >= C) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = ((acc[e] + red[0][lane][e]) + red[1][lane][e]) + red[2][lane][e];
    const double g = gamma[c], b = beta[c];
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        const double s = acc[2 * seg], q = acc[2 * seg + 1];
        const double count = seg ? count_b : count_a;
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0) var = 0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        float* o = bnc2 + (size_t)seg * 5 * C;
        o[c] = (float)mean;
        o[C + c] = (float)invstd;
        o[2 * C + c] = (float)(g * invstd);
        o[3 * C + c] = (float)(b - mean * g * invstd);
        o[4 * C + c] = g < 0 ? -1.0f : 1.0f;
        const double unb = count > 1 ? var * count / (count - 1) : var;
        rm = (float)((1.0 - momentum) * rm + momentum * mean);
        rv = (float)((1.0 - momentum) * rv + momentum * unb);
    }
    if (running_mean) { running_mean[c] = rm; running_var[c] = rv; }
}

// out = relu(scale_seg * y + shift_seg), seg = (row >= M); NaN-propagating like torch.relu
__global__ __launch_bounds__(256) void k_fc_apply(const float* __restrict__ y, int M, int R, int C4,
                                                  const float* __restrict__ bnc2, float* __restrict__ out) {
    const int c4 = blockIdx.x * 256 + threadIdx.x;
    if (c4 >= C4) return;
    const int C = 4 * C4;
    const float4 sa = reinterpret_cast<const float4*>(bnc2 + 2 * C)[c4], ta = reinterpret_cast<const float4*>(bnc2 + 3 * C)[c4];
    const float4 sb = reinterpret_cast<const float4*>(bnc2 + 7 * C)[c4], tb = reinterpret_cast<const float4*>(bnc2 + 8 * C)[c4];
    for (int r = blockIdx.y; r < R; r += gridDim.y) {
        const float4 sc = r < M ? sa : sb, sh = r < M ? ta : tb;
        const float4 v = reinterpret_cast<const float4*>(y)[(size_t)r * C4 + c4];
        float4 o;
        o.x = relu_nan(fmaf(sc.x, v.x, sh.x)); o.y = relu_nan(fmaf(sc.y, v.y, sh.y));
        o.z = relu_nan(fmaf(sc.z, v.z, sh.z)); o.w = relu_nan(fmaf(sc.w, v.w, sh.w));
        reinterpret_cast<float4*>(out)[(size_t)r * C4 + c4] = o;
    }
}

// backward of relu(bn(y)): per slice the sums of dz = dout * [z > 0] and dz * yhat (rows.hip: k_rows_bwd_stats4)
__global__ __launch_bounds__(256) void k_fc_bwd_stats(const float* __restrict__ dout, const float* __restrict__ y, int M, int R,
                                                      int na, int C4, const float* __restrict__ bnc2, double* __restrict__ part) {
    __shared__ double red[3][64][8];
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + lane;
    const int C = 4 * C4;
    int r0, r1, seg;
    fc_slice(blockIdx.y, M, R, na, r0, r1, seg);
    const float* bnc = bnc2 + (size_t)seg * 5 * C;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c4 < C4) {
        const float4 mean = reinterpret_cast<const float4*>(bnc)[c4], inv = reinterpret_cast<const float4*>(bnc + C)[c4];
        const float4 scale = reinterpret_cast<const float4*>(bnc + 2 * C)[c4], shift = reinterpret_cast<const float4*>(bnc + 3 * C)[c4];
        for (int r = r0 + ph; r < r1; r += 4) {
            const size_t o = (size_t)r * C4 + c4;
            const float4 v = reinterpret_cast<const float4*>(y)[o], g = reinterpret_cast<const float4*>(dout)[o];
            const float d0 = fmaf(scale.x, v.x, shift.x) > 0.f ? g.x : 0.f, d1 = fmaf(scale.y, v.y, shift.y) > 0.f ? g.y : 0.f;
            const float d2 = fmaf(scale.z, v.z, shift.z) > 0.f ? g.z : 0.f, d3 = fmaf(scale.w, v.w, shift.w) > 0.f ? g.w : 0.f;
            acc[0] += (double)d0; acc[1] += (double)d0 * (double)((v.x - mean.x) * inv.x);
            acc[2] += (double)d1; acc[3] += (double)d1 * (double)((v.y - mean.y) * inv.y);
            acc[4] += (double)d2; acc[5] += (double)d2 * (double)((v.z - mean.z) * inv.z);
            acc[6] += (double)d3; acc[7] += (double)d3 * (double)((v.w - mean.w) * inv.w);
        }
    }
    if (ph > 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[ph - 1][lane][e] = acc[e];
    }
    __syncthreads();
    if (ph == 0 && c4 < C4) {
        double* pr = part + (size_t)blockIdx.y * 2 * C + 8 * c4;
#pragma unroll
        for (int e = 0; e < 8; ++e) pr[e] = ((acc[e] + red[0][lane][e]) + red[1][lane][e]) + red[2][lane][e];
    }
}

// slice sums -> kk2 (2, 2, C) = (dbeta / P, dgamma / P) per segment of the SyncBN-reduced sums (`sums_g` (2, C, 2), or the
// local ones when null) and the parameter gradients of THIS rank: dbeta = dbeta_a + dbeta_b, dgamma likewise (fp32 adds of
// the fp32-rounded segment sums: what torch's accumulation of the two BatchNorm calls does)
__global__ __launch_bounds__(256) void k_fc_bwd_consts(const double* __restrict__ part, int na, int nb,
                                                       const double* __restrict__ sums_g, int C, double count_a, double count_b,
                                                       float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                       float* __restrict__ kk2) {
    __shared__ double red[3][64][4];                               // 64 channels x 4 slice phases, as k_fc_finalize
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int V = 2 * C;
    double acc[4] = {0, 0, 0, 0};
    if (c < C) {
        for (int t = ph; t < na; t += 4) { acc[0] += part[(size_t)t * V + 2 * c]; acc[1] += part[(size_t)t * V + 2 * c + 1]; }
        for (int t = na + ph; t < na + nb; t += 4) { acc[2] += part[(size_t)t * V + 2 * c]; acc[3] += part[(size_t)t * V + 2 * c + 1]; }
    }
    if (ph > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[ph - 1][lane][e] = acc[e];
    }
    __syncthreads();
    if (ph != 0 || c >= C) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = ((acc[e] + red[0][lane][e]) + red[1][lane][e]) + red[2][lane][e];
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        const double g0 = sums_g ? sums_g[(size_t)seg * V + 2 * c] : acc[2 * seg];
        const double g1 = sums_g ? sums_g[(size_t)seg * V + 2 * c + 1] : acc[2 * seg + 1];
        const float inv = (float)(1.0 / (seg ? count_b : count_a));         // fp32 sum times fp32 reciprocal count (k_bn_bwd_consts)
        kk2[(size_t)seg * V + c] = (float)g0 * inv;
        kk2[(size_t)seg * V + C + c] = (float)g1 * inv;
    }
    dbeta[c] = (float)acc[0] + (float)acc[2];
    dgamma[c] = (float)acc[1] + (float)acc[3];
}

// dy = scale * (dz - k1 - yhat * k2) with the row's segment constants (rows.hip: k_rows_bwd_apply4)
__global__ __launch_bounds__(256) void k_fc_bwd_apply(const float* __restrict__ dout, const float* __restrict__ y, int M, int R,
                                                      int C4, const float* __restrict__ bnc2, const float* __restrict__ kk2,
                                                      float* __restrict__ dy) {
    const int c4 = blockIdx.x * 256 + threadIdx.x;
    if (c4 >= C4) return;
    const int C = 4 * C4;
    float4 mean[2], inv[2], scale[2], shift[2], k1[2], k2[2];
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        const float* bnc = bnc2 + (size_t)seg * 5 * C;
        mean[seg] = reinterpret_cast<const float4*>(bnc)[c4]; inv[seg] = reinterpret_cast<const float4*>(bnc + C)[c4];
        scale[seg] = reinterpret_cast<const float4*>(bnc + 2 * C)[c4]; shift[seg] = reinterpret_cast<const float4*>(bnc + 3 * C)[c4];
        k1[seg] = reinterpret_cast<const float4*>(kk2 + (size_t)seg * 2 * C)[c4];
        k2[seg] = reinterpret_cast<const float4*>(kk2 + (size_t)seg * 2 * C + C)[c4];
    }
    for (int r = blockIdx.y; r < R; r += gridDim.y) {
        const int s = r < M ? 0 : 1;
        const size_t o = (size_t)r * C4 + c4;
        const float4 v = reinterpret_cast<const float4*>(y)[o], g = reinterpret_cast<const float4*>(dout)[o];
        float4 out;
        out.x = scale[s].x * ((fmaf(scale[s].x, v.x, shift[s].x) > 0.f ? g.x : 0.f) - k1[s].x - (v.x - mean[s].x) * inv[s].x * k2[s].x);
        out.y = scale[s].y * ((fmaf(scale[s].y, v.y, shift[s].y) > 0.f ? g.y : 0.f) - k1[s].y - (v.y - mean[s].y) * inv[s].y * k2[s].y);
        out.z = scale[s].z * ((fmaf(scale[s].z, v.z, shift[s].z) > 0.f ? g.z : 0.f) - k1[s].z - (v.z - mean[s].z) * inv[s].z * k2[s].z);
        out.w = scale[s].w * ((fmaf(scale[s].w, v.w, shift[s].w) > 0.f ? g.w : 0.f) - k1[s].w - (v.w - mean[s].w) * inv[s].w * k2[s].w);
        reinterpret_cast<float4*>(dy)[o] = out;
    }
}

// out[c] = sum_r x[r][c] (fp64 accumulation): the bias gradient of a Linear layer over a few hundred rows
__global__ __launch_bounds__(1024) void k_col_sums(const float* __restrict__ x, int R, int C4, float* __restrict__ out) {
    // block = 16 channel quads x 64 row phases (a few blocks of few threads walking hundreds of rows each was 52 us of
    // serial load latency); the phases meet in LDS: 8 partial sums per quad in phase order, then one
    __shared__ double red[64][16][4];
    const int qd = threadIdx.x & 15, ph = threadIdx.x >> 4;
    const int c4 = blockIdx.x * 16 + qd;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (c4 < C4)
        for (int r = ph; r < R; r += 64) {
            const float4 v = reinterpret_cast<const float4*>(x)[(size_t)r * C4 + c4];
            a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
        }
    red[ph][qd][0] = a0; red[ph][qd][1] = a1; red[ph][qd][2] = a2; red[ph][qd][3] = a3;
    __syncthreads();
    if (ph < 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double t = red[ph][qd][e];
#pragma unroll
            for (int k = 1; k < 8; ++k) t += red[ph + 8 * k][qd][e];
            red[ph][qd][e] = t;
        }
    }
    __syncthreads();
    if (ph == 0 && c4 < C4) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double t = red[0][qd][e];
#pragma unroll
            for (int k = 1; k < 8; ++k) t += red[k][qd][e];
            o[e] = (float)t;
        }
        reinterpret_cast<float4*>(out)[c4] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// gobaol_max_pool into the stacked input of netR_FC: h (G*B + B, C) = [x ; max over the G views of x], arg (B, C) = the
// first view that attains the maximum (rows.hip: k_viewmax_fwd, plus the copy of the rows it reads anyway)
// Workgroup = 64 (clip, channel quad) items x 4 view phases: phase p walks a contiguous quarter of the views, the four candidates
// meet in LDS and are merged in view order with the same rule (strictly greater, or NaN, replaces), so the first maximum / the last
// NaN wins exactly as in the one-thread walk over 24 views this replaces (an 11 us chain of loads on 32 workgroups).
__global__ __launch_bounds__(256) void k_viewmax_stack(const float* __restrict__ x, int G, int B, int C4, float* __restrict__ h,
                                                       int* __restrict__ arg) {
    __shared__ float4 sb[3][64];
    __shared__ int4 si[3][64];
    const int it = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + it;                    // (b, c4)
    const bool live = i < B * C4;
    const int b = live ? i / C4 : 0, c4 = live ? i - b * C4 : 0;
    const int per = (G + 3) >> 2;
    const int g0 = ph * per, g1 = g0 + per < G ? g0 + per : G;
    float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 bi = make_int4(-1, -1, -1, -1);                   // -1: this phase saw no view
    if (live && g0 < g1) {
        const size_t o0 = ((size_t)g0 * B + b) * C4 + c4;
        best = reinterpret_cast<const float4*>(x)[o0];
        reinterpret_cast<float4*>(h)[o0] = best;
        bi = make_int4(g0, g0, g0, g0);
        for (int g = g0 + 1; g < g1; ++g) {
            const size_t o = ((size_t)g * B + b) * C4 + c4;
            const float4 v = reinterpret_cast<const float4*>(x)[o];
            reinterpret_cast<float4*>(h)[o] = v;
            if (v.x > best.x || v.x != v.x) { best.x = v.x; bi.x = g; }
            if (v.y > best.y || v.y != v.y) { best.y = v.y; bi.y = g; }
            if (v.z > best.z || v.z != v.z) { best.z = v.z; bi.z = g; }
            if (v.w > best.w || v.w != v.w) { best.w = v.w; bi.w = g; }
        }
    }
    if (ph) { sb[ph - 1][it] = best; si[ph - 1][it] = bi; }
    __syncthreads();
    if (ph == 0 && live) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float4 v = sb[t][it];
            const int4 vi = si[t][it];
            if (vi.x < 0) continue;                        // empty phase (fewer than four views per phase quarter)
            if (v.x > best.x || v.x != v.x) { best.x = v.x; bi.x = vi.x; }
            if (v.y > best.y || v.y != v.y) { best.y = v.y; bi.y = vi.y; }
            if (v.z > best.z || v.z != v.z) { best.z = v.z; bi.z = vi.z; }
            if (v.w > best.w || v.w != v.w) { best.w = v.w; bi.w = vi.w; }
        }
        reinterpret_cast<float4*>(h)[((size_t)G * B + b) * C4 + c4] = best;
        reinterpret_cast<int4*>(arg)[i] = bi;
    }
}

inline bool fc_shape_ok(int64_t M, int64_t R, int C) {
    return M >= FC_SL && !(M % FC_SL) && R > M && R <= 0x7fffffff && C >= 4 && !(C & 3);
}
inline int fc_na(int64_t M) { return (int)(M / FC_SL); }
inline int fc_nb(int64_t M, int64_t R) { return (int)((R - M + FC_SL - 1) / FC_SL); }

}  // namespace

extern "C" int64_t facl_ws_bytes(void);

// Slice statistics of y (R, C) over the segments [0, M) | [M, R) into the workspace (32-row slices; M % 32 == 0); with
// `sums2` (2, C, 2) also the two segment totals (the tensor a SyncBN all-reduce takes).  Returns FACL_E_CONFIG when the
// shape does not fit (the caller keeps the single-segment kernels).
extern "C" int facl_fc_bn_stats(const float* y, int64_t M, int64_t R, int C, double* sums2, void* ws, void* stream) {
    if (!y || !ws) return FACL_E_NULL;
    if (!fc_shape_ok(M, R, C)) return FACL_E_CONFIG;
    if (((uintptr_t)y) & 15) return FACL_E_ALIGN;
    const int na = fc_na(M), nb = fc_nb(M, R);
    if ((size_t)(na + nb) * 2 * C * sizeof(double) > (size_t)facl_ws_bytes()) return FACL_E_CONFIG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fc_stats, dim3((C / 4 + 63) / 64, na + nb), dim3(256), 0, st, y, (int)M, (int)R, na, C / 4, (double*)ws);
    if (sums2) hipLaunchKernelGGL(k_fc_reduce, dim3((2 * C + 255) / 256), dim3(256), 0, st, (const double*)ws, na, nb, 2 * C, sums2);
    return facl_launch_status();
}

// Finalisation of both segments + relu(bn(y)) -> a.  The statistics come from `sums2` (2, C, 2) when given (the all-reduced
// totals; counts = global row counts), else from `part`: `nslices_a` + `nslices_b` slice rows of (C, 2) doubles (the
// workspace after facl_fc_bn_stats).  bnc2 (2, 5, C) is kept for the backward.
extern "C" int facl_fc_bn_apply(const float* y, int64_t M, int64_t R, int C, const double* sums2, const double* part,
                                int nslices_a, int nslices_b, double count_a, double count_b, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                float* bnc2, float* a, void* stream) {
    if (!y || !gamma || !beta || !bnc2 || !a || (!sums2 && !part)) return FACL_E_NULL;
    if (M < 1 || R <= M || R > 0x7fffffff || C < 4 || (C & 3) || count_a < 1 || count_b < 1) return FACL_E_SHAPE;
    if (!sums2 && (nslices_a < 1 || nslices_b < 1)) return FACL_E_SHAPE;
    if ((((uintptr_t)y) | ((uintptr_t)a) | ((uintptr_t)bnc2)) & 15) return FACL_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fc_finalize, dim3((C + 63) / 64), dim3(256), 0, st, sums2 ? sums2 : part, sums2 ? 1 : nslices_a,
                       sums2 ? 1 : nslices_b, C, count_a, count_b, gamma, beta, eps, momentum, running_mean, running_var, bnc2);
    const int gx = (C / 4 + 255) / 256;
    int gy = 1024 / gx;
    if (gy > R) gy = (int)R;
    hipLaunchKernelGGL(k_fc_apply, dim3(gx, gy), dim3(256), 0, st, y, (int)M, (int)R, C / 4, bnc2, a);
    return facl_launch_status();
}

// Backward statistics (slice sums of dz, dz * yhat) into the workspace; with `sums2` also the (2, C, 2) segment totals.
extern "C" int facl_fc_bn_bwd_stats(const float* dact, const float* y, int64_t M, int64_t R, int C, const float* bnc2,
                                    double* sums2, void* ws, void* stream) {
    if (!dact || !y || !bnc2 || !ws) return FACL_E_NULL;
    if (!fc_shape_ok(M, R, C)) return FACL_E_CONFIG;
    if ((((uintptr_t)y) | ((uintptr_t)dact) | ((uintptr_t)bnc2)) & 15) return FACL_E_ALIGN;
    const int na = fc_na(M), nb = fc_nb(M, R);
    if ((size_t)(na + nb) * 2 * C * sizeof(double) > (size_t)facl_ws_bytes()) return FACL_E_CONFIG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fc_bwd_stats, dim3((C / 4 + 63) / 64, na + nb), dim3(256), 0, st, dact, y, (int)M, (int)R, na, C / 4, bnc2,
                       (double*)ws);
    if (sums2) hipLaunchKernelGGL(k_fc_reduce, dim3((2 * C + 255) / 256), dim3(256), 0, st, (const double*)ws, na, nb, 2 * C, sums2);
    return facl_launch_status();
}

// dy, dgamma, dbeta from the slice sums facl_fc_bn_bwd_stats left in the workspace (the LOCAL sums: parameter gradients stay
// local) and, under data parallelism, the all-reduced totals `sums2_g` (2, C, 2) for the dense part.
extern "C" int facl_fc_bn_bwd_apply(const float* dact, const float* y, int64_t M, int64_t R, int C, const float* bnc2,
                                    const double* sums2_g, const void* ws, double count_a, double count_b, float* dgamma,
                                    float* dbeta, float* kk2, float* dy, void* stream) {
    if (!dact || !y || !bnc2 || !ws || !dgamma || !dbeta || !kk2 || !dy) return FACL_E_NULL;
    if (!fc_shape_ok(M, R, C)) return FACL_E_CONFIG;
    if (count_a < 1 || count_b < 1) return FACL_E_SHAPE;
    if ((((uintptr_t)y) | ((uintptr_t)dact) | ((uintptr_t)bnc2) | ((uintptr_t)kk2) | ((uintptr_t)dy)) & 15) return FACL_E_ALIGN;
    const int na = fc_na(M), nb = fc_nb(M, R);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fc_bwd_consts, dim3((C + 63) / 64), dim3(256), 0, st, (const double*)ws, na, nb, sums2_g, C, count_a,
                       count_b, dbeta, dgamma, kk2);
    const int gx = (C / 4 + 255) / 256;
    int gy = 1024 / gx;
    if (gy > R) gy = (int)R;
    hipLaunchKernelGGL(k_fc_bwd_apply, dim3(gx, gy), dim3(256), 0, st, dact, y, (int)M, (int)R, C / 4, bnc2, kk2, dy);
    return facl_launch_status();
}

extern "C" int facl_col_sums(const float* x, int64_t R, int C, float* out, void* stream) {
    if (!x || !out) return FACL_E_NULL;
    if (R < 1 || R > 0x7fffffff || C < 4 || (C & 3)) return FACL_E_SHAPE;
    if ((((uintptr_t)x) | ((uintptr_t)out)) & 15) return FACL_E_ALIGN;
    hipLaunchKernelGGL(k_col_sums, dim3((C / 4 + 15) / 16), dim3(1024), 0, (hipStream_t)stream, x, (int)R, C / 4, out);
    return facl_launch_status();
}

extern "C" int facl_viewmax_stack(const float* x, int G, int B, int C, float* h, int32_t* arg, void* stream) {
    if (!x || !h || !arg) return FACL_E_NULL;
    if (G < 1 || B < 1 || C < 4 || (C & 3)) return FACL_E_SHAPE;
    if ((((uintptr_t)x) | ((uintptr_t)h) | ((uintptr_t)arg)) & 15) return FACL_E_ALIGN;
    const int n = B * (C / 4);
    hipLaunchKernelGGL(k_viewmax_stack, dim3((n + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, G, B, C / 4, h, arg);
    return facl_launch_status();
}
