// Row-major (R,C) activation kernels of the encoder tail (net3DV_3 / netR_FC, cn3d_model_conbag.py:61-88):
// train-mode BatchNorm statistics, BN+ReLU apply, BN+ReLU+max-over-S (my_max_pool :80/:199), and their
// backward passes.  All HBM-bound streaming kernels (float4 per lane, channels fastest).
// Algorithmic bytes: stats R*C*4 read; apply 2*R*C*4; segmax R*C*4 read; bwd_stats 2*R*C*4 read;
// bwd_apply 3*R*C*4; bwd_sparse 2*R*C*4.
#include "common.h"
#include <stdlib.h>

int facl_reduce_rows(const double* part, int rows, int V, double* out, hipStream_t st);

namespace {

constexpr int ROWS_BLOCKS = 1024;     // row-slices of the grid (partial rows in ws)

// consts layout "bnc": (5,C) = mean, invstd, scale, shift, sgn   (facl_bn_finalize)

// ---- column sums / sums of squares -------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_stats(const float* __restrict__ y, int R, int C,
                                                    double* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0, q = 0;
    for (int r = blockIdx.y; r < R; r += gridDim.y) {
        const float v = y[(size_t)r * C + c];
        s += (double)v;
        q += (double)v * (double)v;
    }
    part[(size_t)blockIdx.y * 2 * C + 2 * c] = s;
    part[(size_t)blockIdx.y * 2 * C + 2 * c + 1] = q;
}

// ---- out = relu(scale*y + shift) ------------------------------------------------------------------
__global__ void k_rows_bn_relu(const float* __restrict__ y, long long n4, int C4, const float* __restrict__ scale,
                               const float* __restrict__ shift, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    // the channel quad advances by stride % C4 per trip (one 64-bit remainder per thread, not one per element: k_sa_pool)
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int c4 = (int)(i0 % C4);
    const int dc = (int)(stride % C4);
    for (long long i = i0; i < n4; i += stride, c4 = c4 + dc >= C4 ? c4 + dc - C4 : c4 + dc) {
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        const float4 sc = reinterpret_cast<const float4*>(scale)[c4], sh = reinterpret_cast<const float4*>(shift)[c4];
        float4 o;
        o.x = relu_nan(fmaf(sc.x, v.x, sh.x)); o.y = relu_nan(fmaf(sc.y, v.y, sh.y));
        o.z = relu_nan(fmaf(sc.z, v.z, sh.z)); o.w = relu_nan(fmaf(sc.w, v.w, sh.w));
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

// ---- x_pre[m,c] = max_s relu(bn(y[m,s,c])) = relu(|scale| * max_s(sgn*y) + shift), first max wins ----
__global__ __launch_bounds__(256) void k_rows_segmax(const float* __restrict__ y, int S, int C,
                                                     const float* __restrict__ bnc, float* __restrict__ out,
                                                     int* __restrict__ arg) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    if (c >= C) return;
    const float scale = bnc[2 * C + c], shift = bnc[3 * C + c], sgn = bnc[4 * C + c];
    const float* base = y + (size_t)m * S * C + c;
    float best = sgn * base[0];
    int bi = 0;
    for (int s = 1; s < S; ++s) {
        const float v = sgn * base[(size_t)s * C];
        if (v > best || v != v) { best = v; bi = s; }                          // a NaN wins and stays (MaxPool2d propagates it)
    }
    out[(size_t)m * C + c] = relu_nan(fmaf(fabsf(scale), best, shift));
    arg[(size_t)m * C + c] = bi;
}

// ---- backward of relu(bn(y)): sums of dz = dout*[z>0] and dz*yhat ---------------------------------
__global__ __launch_bounds__(256) void k_rows_bwd_stats(const float* __restrict__ dout, const float* __restrict__ y,
                                                        int R, int C, const float* __restrict__ bnc,
                                                        double* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float mean = bnc[c], inv = bnc[C + c], scale = bnc[2 * C + c], shift = bnc[3 * C + c];
    double s = 0, g = 0;
    for (int r = blockIdx.y; r < R; r += gridDim.y) {
        const float v = y[(size_t)r * C + c];
        const float d = fmaf(scale, v, shift) > 0.f ? dout[(size_t)r * C + c] : 0.f;
        s += (double)d;
        g += (double)d * (double)((v - mean) * inv);
    }
    part[(size_t)blockIdx.y * 2 * C + 2 * c] = s;
    part[(size_t)blockIdx.y * 2 * C + 2 * c + 1] = g;
}

// the same, 4 channels per lane (C % 4 == 0, 16-byte aligned tensors): block = 64 channel quads x 4 row phases,
// the 4 phases are combined through LDS so that a block still writes ONE partial row
__global__ __launch_bounds__(256) void k_rows_bwd_stats4(const float* __restrict__ dout, const float* __restrict__ y,
                                                         int R, int C4, const float* __restrict__ bnc,
                                                         double* __restrict__ part) {
    __shared__ double red[3][64][8];
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + lane;
    const int C = 4 * C4;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c4 < C4) {
        const float4 mean = reinterpret_cast<const float4*>(bnc)[c4], inv = reinterpret_cast<const float4*>(bnc + C)[c4];
        const float4 scale = reinterpret_cast<const float4*>(bnc + 2 * C)[c4], shift = reinterpret_cast<const float4*>(bnc + 3 * C)[c4];
        for (int r = blockIdx.y * 4 + ph; r < R; r += gridDim.y * 4) {
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
        double* pr = part + (size_t)blockIdx.y * 2 * C + 8 * c4;       // columns 4c4..4c4+3, (sum, sum*yhat) pairs
#pragma unroll
        for (int e = 0; e < 8; ++e) pr[e] = ((acc[e] + red[0][lane][e]) + red[1][lane][e]) + red[2][lane][e];
    }
}

// max|.| of the tensor a kernel writes, for the consumers that scale it by a power of two (fp16x3 GEMMs, common.h): the bit
// pattern of a non-negative float orders like the unsigned integer, NaN above everything (so a NaN gradient stays visible).
// The maximum lives in FACL_AMAX_SLOTS slots, one 128-byte line each (a workgroup uses slot = its linear id mod the slot
// count: thousands of atomics on ONE address serialise in the L2 -- measured, the pass doubled in time); the consumer takes
// the maximum over the slots.  A wave reads its slot when it STARTS and skips the atomic when it cannot raise that value
// (a stale read only costs a redundant atomic).  Lanes that left early (channel tail) are absent from the exchange.
__device__ __forceinline__ float abs_max4(float m, const float4& v) {
    const unsigned a = __float_as_uint(m);
    unsigned b = __float_as_uint(v.x) & 0x7fffffffu, c = __float_as_uint(v.y) & 0x7fffffffu;
    unsigned d = __float_as_uint(v.z) & 0x7fffffffu, e = __float_as_uint(v.w) & 0x7fffffffu;
    b = b > c ? b : c; d = d > e ? d : e; b = b > d ? b : d;
    return __uint_as_float(a > b ? a : b);
}
__device__ __forceinline__ unsigned* abs_max_slot(unsigned* amax) {
    return amax + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) & (FACL_AMAX_SLOTS - 1)) * FACL_AMAX_STRIDE;
}
__device__ __forceinline__ void publish_abs_max(unsigned* slot, unsigned seen, float m) {
    const unsigned b = __float_as_uint(m);
    if (b > seen) atomicMax(slot, b);                                   // the compiler folds a wave's lanes into one atomic
}

// dy = scale * (dz - k1 - yhat*k2),  kk = (2,C): k1 = dbeta/P, k2 = dgamma/P
__global__ void k_rows_bwd_apply(const float* __restrict__ dout, const float* __restrict__ y, long long n, int C,
                                 const float* __restrict__ bnc, const float* __restrict__ kk,
                                 float* __restrict__ dy) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int c = (int)(i0 % C);
    const int dc = (int)(stride % C);
    for (long long i = i0; i < n; i += stride, c = c + dc >= C ? c + dc - C : c + dc) {
        const float mean = bnc[c], inv = bnc[C + c], scale = bnc[2 * C + c], shift = bnc[3 * C + c];
        const float v = y[i];
        const float d = fmaf(scale, v, shift) > 0.f ? dout[i] : 0.f;
        dy[i] = scale * (d - kk[c] - (v - mean) * inv * kk[C + c]);
    }
}

// the same, 4 channels per lane (C % 4 == 0, 16-byte aligned tensors): one (row, channel-quad) per iteration
__global__ __launch_bounds__(256) void k_rows_bwd_apply4(const float* __restrict__ dout, const float* __restrict__ y,
                                                         int R, int C4, const float* __restrict__ bnc,
                                                         const float* __restrict__ kk, float* __restrict__ dy,
                                                         unsigned* __restrict__ amax) {
    const int c4 = blockIdx.x * 256 + threadIdx.x;
    if (c4 >= C4) return;
    float mx = 0.f;
    unsigned* const slot = amax ? abs_max_slot(amax) : nullptr;
    const unsigned seen = amax ? *slot : 0u;
    const int C = 4 * C4;
    const float4 mean = reinterpret_cast<const float4*>(bnc)[c4], inv = reinterpret_cast<const float4*>(bnc + C)[c4];
    const float4 scale = reinterpret_cast<const float4*>(bnc + 2 * C)[c4], shift = reinterpret_cast<const float4*>(bnc + 3 * C)[c4];
    const float4 k1 = reinterpret_cast<const float4*>(kk)[c4], k2 = reinterpret_cast<const float4*>(kk + C)[c4];
    for (int r = blockIdx.y; r < R; r += gridDim.y) {
        const size_t o = (size_t)r * C4 + c4;
        const float4 v = reinterpret_cast<const float4*>(y)[o], g = reinterpret_cast<const float4*>(dout)[o];
        float4 out;
        out.x = scale.x * ((fmaf(scale.x, v.x, shift.x) > 0.f ? g.x : 0.f) - k1.x - (v.x - mean.x) * inv.x * k2.x);
        out.y = scale.y * ((fmaf(scale.y, v.y, shift.y) > 0.f ? g.y : 0.f) - k1.y - (v.y - mean.y) * inv.y * k2.y);
        out.z = scale.z * ((fmaf(scale.z, v.z, shift.z) > 0.f ? g.z : 0.f) - k1.z - (v.z - mean.z) * inv.z * k2.z);
        out.w = scale.w * ((fmaf(scale.w, v.w, shift.w) > 0.f ? g.w : 0.f) - k1.w - (v.w - mean.w) * inv.w * k2.w);
        reinterpret_cast<float4*>(dy)[o] = out;
        mx = abs_max4(mx, out);
    }
    if (amax) publish_abs_max(slot, seen, mx);
}

// ---- backward through the max over S (+ BN + ReLU): sparse sums, then the dense dy -----------------
// dz[m,c] = dxpre[m,c] * [xpre > 0] lives at row arg[m,c]; yhat there = (y_at_arg - mean)*inv.
__global__ __launch_bounds__(256) void k_segmax_bwd_stats(const float* __restrict__ dxpre, const float* __restrict__ xpre,
                                                          const float* __restrict__ y, const int* __restrict__ arg,
                                                          int Mrows, int S, int C, const float* __restrict__ bnc,
                                                          double* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float mean = bnc[c], inv = bnc[C + c];
    double s = 0, g = 0;
    for (int m = blockIdx.y; m < Mrows; m += gridDim.y) {
        const size_t o = (size_t)m * C + c;
        const float d = xpre[o] > 0.f ? dxpre[o] : 0.f;
        const float v = y[((size_t)m * S + arg[o]) * C + c];
        s += (double)d;
        g += (double)d * (double)((v - mean) * inv);
    }
    part[(size_t)blockIdx.y * 2 * C + 2 * c] = s;
    part[(size_t)blockIdx.y * 2 * C + 2 * c + 1] = g;
}

// The same sums from ymax (M, C) = max over the S rows of sign(gamma) * y, which the forward's max-pool epilogue returns anyway: the
// value at the argmax is sign(gamma) * ymax EXACTLY, so the 4-byte gather y[(m*S + arg)*C + c] -- one 128-byte line per element,
// 100 MB of fills for 3 MB of values, 15 us -- is not needed.  bnc row 4 = sign(gamma).
__global__ __launch_bounds__(256) void k_segmax_bwd_stats_ymax(const float* __restrict__ dxpre, const float* __restrict__ xpre,
                                                               const float* __restrict__ ymax, int Mrows, int C,
                                                               const float* __restrict__ bnc, double* __restrict__ part,
                                                               unsigned* __restrict__ zamax, int zwords) {
    // `zamax` (or null): amax words this launch ZEROES for the kernels behind it, which raise them with atomics (spares a fill launch)
    if (zamax && blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < zwords; i += 256) zamax[i] = 0u;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float mean = bnc[c], inv = bnc[C + c], sg = bnc[4 * C + c];
    double s = 0, g = 0;
    for (int m = blockIdx.y; m < Mrows; m += gridDim.y) {
        const size_t o = (size_t)m * C + c;
        const float d = xpre[o] > 0.f ? dxpre[o] : 0.f;
        const float v = sg * ymax[o];
        s += (double)d;
        g += (double)d * (double)((v - mean) * inv);
    }
    part[(size_t)blockIdx.y * 2 * C + 2 * c] = s;
    part[(size_t)blockIdx.y * 2 * C + 2 * c + 1] = g;
}

__global__ __launch_bounds__(256) void k_segmax_bwd_apply(const float* __restrict__ dxpre, const float* __restrict__ xpre,
                                                          const float* __restrict__ y, const int* __restrict__ arg,
                                                          int S, int C, const float* __restrict__ bnc,
                                                          const float* __restrict__ kk, float* __restrict__ dy) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    if (c >= C) return;
    const float mean = bnc[c], inv = bnc[C + c], scale = bnc[2 * C + c];
    const float k1 = kk[c], k2 = kk[C + c];
    const size_t o = (size_t)m * C + c;
    const float d = xpre[o] > 0.f ? dxpre[o] : 0.f;
    const int a = arg[o];
    const float* yb = y + (size_t)m * S * C + c;
    float* db = dy + (size_t)m * S * C + c;
    for (int s = 0; s < S; ++s) {
        const float v = yb[(size_t)s * C];
        db[(size_t)s * C] = scale * ((s == a ? d : 0.f) - k1 - (v - mean) * inv * k2);
    }
}

// the same, 4 channels per lane (C % 4 == 0, 16-byte aligned tensors)
__global__ __launch_bounds__(256) void k_segmax_bwd_apply4(const float* __restrict__ dxpre, const float* __restrict__ xpre,
                                                           const float* __restrict__ y, const int* __restrict__ arg,
                                                           int S, int C4, const float* __restrict__ bnc,
                                                           const float* __restrict__ kk, float* __restrict__ dy,
                                                           unsigned* __restrict__ amax) {
    const int c4 = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    if (c4 >= C4) return;
    float mx = 0.f;
    unsigned* const slot = amax ? abs_max_slot(amax) : nullptr;
    const unsigned seen = amax ? *slot : 0u;
    const int C = 4 * C4;
    const float4 mean = reinterpret_cast<const float4*>(bnc)[c4], inv = reinterpret_cast<const float4*>(bnc + C)[c4];
    const float4 scale = reinterpret_cast<const float4*>(bnc + 2 * C)[c4];
    const float4 k1 = reinterpret_cast<const float4*>(kk)[c4], k2 = reinterpret_cast<const float4*>(kk + C)[c4];
    const size_t o = (size_t)m * C4 + c4;
    const float4 xp = reinterpret_cast<const float4*>(xpre)[o], dx = reinterpret_cast<const float4*>(dxpre)[o];
    const int4 a = reinterpret_cast<const int4*>(arg)[o];
    const float4 d = make_float4(xp.x > 0.f ? dx.x : 0.f, xp.y > 0.f ? dx.y : 0.f, xp.z > 0.f ? dx.z : 0.f, xp.w > 0.f ? dx.w : 0.f);
    const float4* yb = reinterpret_cast<const float4*>(y) + (size_t)m * S * C4 + c4;
    float4* db = reinterpret_cast<float4*>(dy) + (size_t)m * S * C4 + c4;
    for (int s = 0; s < S; ++s) {
        const float4 v = yb[(size_t)s * C4];
        float4 out;
        out.x = scale.x * ((s == a.x ? d.x : 0.f) - k1.x - (v.x - mean.x) * inv.x * k2.x);
        out.y = scale.y * ((s == a.y ? d.y : 0.f) - k1.y - (v.y - mean.y) * inv.y * k2.y);
        out.z = scale.z * ((s == a.z ? d.z : 0.f) - k1.z - (v.z - mean.z) * inv.z * k2.z);
        out.w = scale.w * ((s == a.w ? d.w : 0.f) - k1.w - (v.w - mean.w) * inv.w * k2.w);
        db[(size_t)s * C4] = out;
        mx = abs_max4(mx, out);
    }
    if (amax) publish_abs_max(slot, seen, mx);
}

// dWc[c][j] = sum_r dy[r][c] * centers[r][j]: the 3 centroid-xyz columns of the first per-centroid layer's weight
// gradient (torch.cat((yt, xt), 1), cn3d_model_conbag.py:219).  A (C x 3) output is 97 % padding for a 128x128-tile
// GEMM; this is one streaming pass over dy: block = 64 channel quads x 4 row phases, fp64 partial sums.
__global__ __launch_bounds__(256) void k_rows_center_wgrad(const float* __restrict__ dy, const float* __restrict__ ctr,
                                                           int R, int C4, double* __restrict__ part) {
    __shared__ double red[3][64][12];
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + lane;
    double acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (c4 < C4) {
        for (int r = blockIdx.y * 4 + ph; r < R; r += gridDim.y * 4) {
            const float4 g = reinterpret_cast<const float4*>(dy)[(size_t)r * C4 + c4];
            const float x0 = ctr[(size_t)r * 3], x1 = ctr[(size_t)r * 3 + 1], x2 = ctr[(size_t)r * 3 + 2];
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[3 * e] += (double)gv[e] * (double)x0;
                acc[3 * e + 1] += (double)gv[e] * (double)x1;
                acc[3 * e + 2] += (double)gv[e] * (double)x2;
            }
        }
    }
    if (ph > 0) {
#pragma unroll
        for (int e = 0; e < 12; ++e) red[ph - 1][lane][e] = acc[e];
    }
    __syncthreads();
    if (ph == 0 && c4 < C4) {
        double* pr = part + (size_t)blockIdx.y * 12 * C4 + 12 * c4;      // (channel, xyz) row-major: 3*(4c4+e) + j
#pragma unroll
        for (int e = 0; e < 12; ++e) pr[e] = ((acc[e] + red[0][lane][e]) + red[1][lane][e]) + red[2][lane][e];
    }
}

// ---- gobaol_max_pool (cn3d_model_conbag.py:225-226): max over the G views of a clip's per-view maxima -----------
// x (G*B, C) view-major rows g*B+b -> out (B,C) = max_g, arg (B,C) = first g that attains it; 4 channels per lane.
__global__ __launch_bounds__(256) void k_viewmax_fwd(const float* __restrict__ x, int G, int B, int C4,
                                                     float* __restrict__ out, int* __restrict__ arg) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // (b, c4)
    if (i >= B * C4) return;
    const int b = i / C4, c4 = i - b * C4;
    float4 best = reinterpret_cast<const float4*>(x)[(size_t)b * C4 + c4];
    int4 bi = make_int4(0, 0, 0, 0);
    for (int g = 1; g < G; ++g) {
        const float4 v = reinterpret_cast<const float4*>(x)[((size_t)g * B + b) * C4 + c4];
        if (v.x > best.x || v.x != v.x) { best.x = v.x; bi.x = g; }
        if (v.y > best.y || v.y != v.y) { best.y = v.y; bi.y = g; }
        if (v.z > best.z || v.z != v.z) { best.z = v.z; bi.z = g; }
        if (v.w > best.w || v.w != v.w) { best.w = v.w; bi.w = g; }
    }
    reinterpret_cast<float4*>(out)[i] = best;
    reinterpret_cast<int4*>(arg)[i] = bi;
}

// dx (G*B, C): dout routed to the winning view's row, zero elsewhere (every element written: no memset needed)
__global__ __launch_bounds__(256) void k_viewmax_bwd(const float* __restrict__ dout, const int* __restrict__ arg, int G,
                                                     int B, int C4, float* __restrict__ dx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C4) return;
    const int b = i / C4, c4 = i - b * C4;
    const float4 d = reinterpret_cast<const float4*>(dout)[i];
    const int4 a = reinterpret_cast<const int4*>(arg)[i];
    for (int g = 0; g < G; ++g)
        reinterpret_cast<float4*>(dx)[((size_t)g * B + b) * C4 + c4] =
            make_float4(a.x == g ? d.x : 0.f, a.y == g ? d.y : 0.f, a.z == g ? d.z : 0.f, a.w == g ? d.w : 0.f);
}

// F.normalize(p=2, dim=1, eps=1e-12) + mapping (Linear C -> K, no bias): one workgroup of 256 threads per row.
// The row is normalised into LDS; thread t then owns output k = t >> 2 (+64 per pass) and a quarter of the C columns.
__global__ __launch_bounds__(256) void k_normalize_map(const float* __restrict__ x, int C, const float* __restrict__ Wm,
                                                       int K, float* __restrict__ xn, float* __restrict__ code) {
    __shared__ __attribute__((aligned(16))) float row[4096];
    __shared__ float red[4];
    const size_t m = blockIdx.x;
    const float4* xr = reinterpret_cast<const float4*>(x + m * C);
    const int C4 = C >> 2;
    float ss = 0.f;
    for (int i = threadIdx.x; i < C4; i += 256) {
        const float4 v = xr[i];
        reinterpret_cast<float4*>(row)[i] = v;
        ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
    }
    ss = wave_sum_f32(ss);
    if (lane_id() == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    const float inv = 1.f / fmaxf(nrm, 1e-12f);
    float4* xo = reinterpret_cast<float4*>(xn + m * C);
    for (int i = threadIdx.x; i < C4; i += 256) {
        float4 v = reinterpret_cast<float4*>(row)[i];
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        reinterpret_cast<float4*>(row)[i] = v;
        xo[i] = v;
    }
    __syncthreads();
    const int qtr = threadIdx.x & 3, q4 = C4 >> 2;          // quarter of the row: float4s [qtr*q4, (qtr+1)*q4)
    for (int k = threadIdx.x >> 2; k < K; k += 64) {
        const float4* w = reinterpret_cast<const float4*>(Wm + (size_t)k * C) + qtr * q4;
        const float4* r = reinterpret_cast<const float4*>(row) + qtr * q4;
        float acc = 0.f;
        for (int i = 0; i < q4; ++i) {
            const float4 a = r[i], b = w[i];
            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (qtr == 0) code[m * K + k] = acc;
    }
}

// The same for FOUR rows per workgroup (C <= 1024, K = 4 * KPW): wave w normalises row 4b + w into LDS, then owns the outputs
// k = w, w + 4, ..: each float4 of a mapping row is loaded once (coalesced, all KPW rows' loads in flight) and multiplied into the
// four rows' accumulators; the wave's 4 * KPW = 64 partial sums are then reduced TOGETHER by a halving butterfly (63 exchanges,
// independent within a step; lane L ends up with the total of value L) instead of one 6-step chain per output.  The one-row form
// walked its mapping rows with 16-byte pieces of 64 different cache lines per load instruction.
template <int KPW, int NI>
__global__ __launch_bounds__(256) void k_normalize_map4(const float* __restrict__ x, int M, const float* __restrict__ Wm,
                                                        float* __restrict__ xn, float* __restrict__ code) {
    static_assert(KPW * 4 == 64, "the butterfly leaves one total per lane");
    constexpr int K = 4 * KPW, C4 = 64 * NI, C = 4 * C4;
    __shared__ __attribute__((aligned(16))) float rows[4][C];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = blockIdx.x * 4;
    // this wave's slices of the mapping rows do not depend on x: requested first, they fly under the normalisation (inside the
    // replayed step the matrix is cold -- nothing has touched it since the previous step)
    const float4* w4 = reinterpret_cast<const float4*>(Wm) + (size_t)wave * C4 + lane;
    float4 b[NI][KPW];
#pragma unroll
    for (int t = 0; t < NI; ++t)
#pragma unroll
        for (int kk = 0; kk < KPW; ++kk) b[t][kk] = w4[(size_t)(4 * kk) * C4 + 64 * t];
    {
        const int m = m0 + wave;                           // wave-uniform
        float4 v[NI];
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int i = lane + 64 * t;
            v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < M && i < C4) v[t] = reinterpret_cast<const float4*>(x + (size_t)m * C)[i];
            ss = fmaf(v[t].x, v[t].x, ss); ss = fmaf(v[t].y, v[t].y, ss); ss = fmaf(v[t].z, v[t].z, ss); ss = fmaf(v[t].w, v[t].w, ss);
        }
        ss = wave_sum_f32(ss);
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int i = lane + 64 * t;
            if (i < C4) {
                const float4 o = make_float4(v[t].x * inv, v[t].y * inv, v[t].z * inv, v[t].w * inv);
                reinterpret_cast<float4*>(rows[wave])[i] = o;
                if (m < M) reinterpret_cast<float4*>(xn + (size_t)m * C)[i] = o;
            }
        }
    }
    __syncthreads();
    float a[64];                                           // [kk * 4 + row]
#pragma unroll
    for (int j = 0; j < 64; ++j) a[j] = 0.f;
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        float4 r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = reinterpret_cast<const float4*>(rows[q])[lane + 64 * t];
#pragma unroll
        for (int kk = 0; kk < KPW; ++kk)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float u = a[4 * kk + q];
                u = fmaf(r[q].x, b[t][kk].x, u); u = fmaf(r[q].y, b[t][kk].y, u); u = fmaf(r[q].z, b[t][kk].z, u); u = fmaf(r[q].w, b[t][kk].w, u);
                a[4 * kk + q] = u;
            }
    }
    // halving butterfly: after the step with offset o, value index bit log2(o) equals the lane's bit log2(o)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const bool hi = (lane & o) != 0;
#pragma unroll
        for (int j = 0; j < o; ++j) {
            const float keep = hi ? a[j + o] : a[j], send = hi ? a[j] : a[j + o];
            a[j] = keep + __shfl_xor(send, o, 64);
        }
    }
    const int kk = lane >> 2, q = lane & 3;
    if (m0 + q < M) code[(size_t)(m0 + q) * K + wave + 4 * kk] = a[0];
}

// dx (G*B, C) += dout routed to the winning view's row (the rows of dx already hold the other gradient path of x_pre)
__global__ __launch_bounds__(256) void k_viewmax_bwd_add(const float* __restrict__ dout, const int* __restrict__ arg,
                                                         int B, int C4, float* __restrict__ dx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C4) return;
    const int b = i / C4, c4 = i - b * C4;
    const float4 d = reinterpret_cast<const float4*>(dout)[i];
    const int4 a = reinterpret_cast<const int4*>(arg)[i];
    const size_t C = (size_t)C4 * 4, col = (size_t)c4 * 4;
    dx[((size_t)a.x * B + b) * C + col + 0] += d.x;
    dx[((size_t)a.y * B + b) * C + col + 1] += d.y;
    dx[((size_t)a.z * B + b) * C + col + 2] += d.z;
    dx[((size_t)a.w * B + b) * C + col + 3] += d.w;
}

int rows_grid_y(int R, int C) {
    int gx = (C + 255) / 256;
    int gy = ROWS_BLOCKS / gx;
    if (gy > R) gy = R;
    if (gy < 1) gy = 1;
    return gy;
}

}  // namespace

extern "C" int facl_rows_stats(const float* y, int64_t R, int C, double* sums, void* ws, void* stream) {
    if (!y || !sums || !ws) return FACL_E_NULL;
    if (R < 1 || R > 0x7fffffff || C < 1 || 2 * C > 4608) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int gy = rows_grid_y((int)R, C);
    hipLaunchKernelGGL(k_rows_stats, dim3((C + 255) / 256, gy), dim3(256), 0, st, y, (int)R, C, (double*)ws);
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, gy, 2 * C, sums, st);
}

extern "C" int facl_rows_bn_relu(const float* y, int64_t R, int C, const float* scale, const float* shift, float* out,
                                 void* stream) {
    if (!y || !scale || !shift || !out) return FACL_E_NULL;
    if (R < 1 || C < 4 || (C & 3)) return FACL_E_SHAPE;
    const long long n4 = R * (long long)(C / 4);
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_rows_bn_relu, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, n4, C / 4, scale, shift, out);
    return facl_launch_status();
}

extern "C" int facl_rows_segmax(const float* y, int64_t M, int S, int C, const float* bnc, float* out, int32_t* arg,
                                void* stream) {
    if (!y || !bnc || !out || !arg) return FACL_E_NULL;
    if (M < 1 || M > 65535 || S < 1 || C < 1) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_rows_segmax, dim3((C + 255) / 256, (int)M), dim3(256), 0, (hipStream_t)stream, y, S, C, bnc,
                       out, arg);
    return facl_launch_status();
}

extern "C" int facl_rows_bwd_stats(const float* dout, const float* y, int64_t R, int C, const float* bnc, double* sums,
                                   void* ws, void* stream) {
    if (!dout || !y || !bnc || !sums || !ws) return FACL_E_NULL;
    if (R < 1 || R > 0x7fffffff || C < 1 || 2 * C > 4608) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    int gy = rows_grid_y((int)R, C);
    if (!(C & 3) && !((((uintptr_t)dout) | ((uintptr_t)y) | ((uintptr_t)bnc)) & 15)) {
        const int gx = (C / 4 + 63) / 64;
        gy = ROWS_BLOCKS / gx;
        if (gy > (R + 3) / 4) gy = (int)((R + 3) / 4);
        if (gy < 1) gy = 1;
        hipLaunchKernelGGL(k_rows_bwd_stats4, dim3(gx, gy), dim3(256), 0, st, dout, y, (int)R, C / 4, bnc, (double*)ws);
    } else {
        hipLaunchKernelGGL(k_rows_bwd_stats, dim3((C + 255) / 256, gy), dim3(256), 0, st, dout, y, (int)R, C, bnc,
                           (double*)ws);
    }
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, gy, 2 * C, sums, st);
}

extern "C" int facl_rows_bwd_apply(const float* dout, const float* y, int64_t R, int C, const float* bnc,
                                   const float* kk, float* dy, void* stream) {
    return facl_rows_bwd_apply_amax(dout, y, R, C, bnc, kk, dy, nullptr, stream);
}

// facl_rows_bwd_apply that also raises *amax (device scalar, zeroed by the caller) to the bits of max|dy|: the operand scale
// of the fp16x3 GEMMs that consume dy (facl_gemm_rs_dgrad, facl_gemm_rs_wgrad).  Needs the 4-channel form (C % 4 == 0,
// 16-byte aligned tensors): FACL_E_ALIGN otherwise when amax is given.
extern "C" int facl_rows_bwd_apply_amax(const float* dout, const float* y, int64_t R, int C, const float* bnc,
                                        const float* kk, float* dy, uint32_t* amax, void* stream) {
    if (!dout || !y || !bnc || !kk || !dy) return FACL_E_NULL;
    if (R < 1 || C < 1) return FACL_E_SHAPE;
    if (!(C & 3) && R <= 0x7fffffff && !((((uintptr_t)dout) | ((uintptr_t)y) | ((uintptr_t)dy) | ((uintptr_t)bnc) | ((uintptr_t)kk)) & 15)) {
        const int gx = (C / 4 + 255) / 256;
        int gy = 4096 / gx;
        if (gy > R) gy = (int)R;
        hipLaunchKernelGGL(k_rows_bwd_apply4, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, dout, y, (int)R, C / 4, bnc, kk, dy, amax);
        return facl_launch_status();
    }
    if (amax) return FACL_E_ALIGN;
    const long long n = R * (long long)C;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_rows_bwd_apply, dim3(grid), dim3(256), 0, (hipStream_t)stream, dout, y, n, C, bnc, kk, dy);
    return facl_launch_status();
}

extern "C" int facl_segmax_bwd_stats(const float* dxpre, const float* xpre, const float* y, const int32_t* arg,
                                     int64_t M, int S, int C, const float* bnc, double* sums, void* ws, void* stream) {
    if (!dxpre || !xpre || !y || !arg || !bnc || !sums || !ws) return FACL_E_NULL;
    if (M < 1 || M > 0x7fffffff || S < 1 || C < 1 || 2 * C > 4608) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int gy = rows_grid_y((int)M, C);
    hipLaunchKernelGGL(k_segmax_bwd_stats, dim3((C + 255) / 256, gy), dim3(256), 0, st, dxpre, xpre, y, arg, (int)M, S,
                       C, bnc, (double*)ws);
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, gy, 2 * C, sums, st);
}

extern "C" int facl_segmax_bwd_stats_ymax(const float* dxpre, const float* xpre, const float* ymax, int64_t M, int C,
                                          const float* bnc, double* sums, void* ws, uint32_t* zamax, int zwords, void* stream) {
    if (!dxpre || !xpre || !ymax || !bnc || !sums || !ws) return FACL_E_NULL;
    if (M < 1 || M > 0x7fffffff || C < 1 || 2 * C > 4608 || zwords < 0) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int gy = rows_grid_y((int)M, C);
    hipLaunchKernelGGL(k_segmax_bwd_stats_ymax, dim3((C + 255) / 256, gy), dim3(256), 0, st, dxpre, xpre, ymax, (int)M, C, bnc,
                       (double*)ws, zamax, zwords);
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, gy, 2 * C, sums, st);
}

extern "C" int facl_segmax_bwd_apply(const float* dxpre, const float* xpre, const float* y, const int32_t* arg,
                                     int64_t M, int S, int C, const float* bnc, const float* kk, float* dy,
                                     void* stream) {
    return facl_segmax_bwd_apply_amax(dxpre, xpre, y, arg, M, S, C, bnc, kk, dy, nullptr, stream);
}

// facl_segmax_bwd_apply that also raises *amax to the bits of max|dy| (see facl_rows_bwd_apply_amax)
extern "C" int facl_segmax_bwd_apply_amax(const float* dxpre, const float* xpre, const float* y, const int32_t* arg,
                                          int64_t M, int S, int C, const float* bnc, const float* kk, float* dy,
                                          uint32_t* amax, void* stream) {
    if (!dxpre || !xpre || !y || !arg || !bnc || !kk || !dy) return FACL_E_NULL;
    if (M < 1 || M > 65535 || S < 1 || C < 1) return FACL_E_SHAPE;
    if (!(C & 3) && !((((uintptr_t)dxpre) | ((uintptr_t)xpre) | ((uintptr_t)y) | ((uintptr_t)arg) | ((uintptr_t)dy) |
                       ((uintptr_t)bnc) | ((uintptr_t)kk)) & 15)) {
        hipLaunchKernelGGL(k_segmax_bwd_apply4, dim3((C / 4 + 255) / 256, (int)M), dim3(256), 0, (hipStream_t)stream, dxpre,
                           xpre, y, arg, S, C / 4, bnc, kk, dy, amax);
        return facl_launch_status();
    }
    if (amax) return FACL_E_ALIGN;
    hipLaunchKernelGGL(k_segmax_bwd_apply, dim3((C + 255) / 256, (int)M), dim3(256), 0, (hipStream_t)stream, dxpre,
                       xpre, y, arg, S, C, bnc, kk, dy);
    return facl_launch_status();
}

extern "C" int facl_rows_center_wgrad(const float* dy, const float* centers, int64_t R, int C, double* dWc, void* ws,
                                      void* stream) {
    if (!dy || !centers || !dWc || !ws) return FACL_E_NULL;
    if (R < 1 || R > 0x7fffffff || C < 4 || (C & 3) || 3 * C > 4608) return FACL_E_SHAPE;
    if (((uintptr_t)dy) & 15) return FACL_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int gx = (C / 4 + 63) / 64;
    int gy = ROWS_BLOCKS / gx;
    if (gy > (R + 3) / 4) gy = (int)((R + 3) / 4);
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(k_rows_center_wgrad, dim3(gx, gy), dim3(256), 0, st, dy, centers, (int)R, C / 4, (double*)ws);
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, gy, 3 * C, dWc, st);
}

extern "C" int facl_viewmax_fwd(const float* x, int G, int B, int C, float* out, int32_t* arg, void* stream) {
    if (!x || !out || !arg) return FACL_E_NULL;
    if (G < 1 || B < 1 || C < 4 || (C & 3)) return FACL_E_SHAPE;
    if ((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)arg)) & 15) return FACL_E_ALIGN;
    const int n = B * (C / 4);
    hipLaunchKernelGGL(k_viewmax_fwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, G, B, C / 4, out, arg);
    return facl_launch_status();
}

extern "C" int facl_viewmax_bwd(const float* dout, const int32_t* arg, int G, int B, int C, float* dx, void* stream) {
    if (!dout || !arg || !dx) return FACL_E_NULL;
    if (G < 1 || B < 1 || C < 4 || (C & 3)) return FACL_E_SHAPE;
    if ((((uintptr_t)dout) | ((uintptr_t)dx) | ((uintptr_t)arg)) & 15) return FACL_E_ALIGN;
    const int n = B * (C / 4);
    hipLaunchKernelGGL(k_viewmax_bwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dout, arg, G, B, C / 4, dx);
    return facl_launch_status();
}

extern "C" int facl_normalize_map(const float* x, int64_t M, int C, const float* Wm, int K, float* x_nor, float* code,
                                  void* stream) {
    if (!x || !Wm || !x_nor || !code) return FACL_E_NULL;
    if (M < 0 || M > 0x7fffffff || C < 16 || (C & 15) || C > 4096 || K < 1 || K > 256) return FACL_E_SHAPE;
    if ((((uintptr_t)x) | ((uintptr_t)Wm) | ((uintptr_t)x_nor)) & 15) return FACL_E_ALIGN;
    if (M == 0) return 0;
    static const int four = getenv("FACL_NORMMAP4") ? atoi(getenv("FACL_NORMMAP4")) : 1;     // A/B knob
    if (four && C == 512 && K == 64)
        hipLaunchKernelGGL((k_normalize_map4<16, 2>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (int)M, Wm,
                           x_nor, code);
    else if (four && C == 256 && K == 64)
        hipLaunchKernelGGL((k_normalize_map4<16, 1>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (int)M, Wm,
                           x_nor, code);
    else
        hipLaunchKernelGGL(k_normalize_map, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, x, C, Wm, K, x_nor, code);
    return facl_launch_status();
}

extern "C" int facl_viewmax_bwd_add(const float* dout, const int32_t* arg, int G, int B, int C, float* dx, void* stream) {
    if (!dout || !arg || !dx) return FACL_E_NULL;
    if (G < 1 || B < 1 || C < 4 || (C & 3)) return FACL_E_SHAPE;
    if ((((uintptr_t)dout) | ((uintptr_t)arg)) & 15) return FACL_E_ALIGN;
    const int n = B * (C / 4);
    hipLaunchKernelGGL(k_viewmax_bwd_add, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dout, arg, B, C / 4, dx);
    return facl_launch_status();
}
