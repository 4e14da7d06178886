// Small fp64 "finalize" kernels between the passes of the train-mode BatchNorm pipeline:
// partial-sum reduction, BN statistics -> (mean, invstd, scale, shift, sign), running-stat update.
// BN semantics: cn3d_model_conbag.py:46,50,54,64,68,72,84 (nn.BatchNorm2d/1d defaults: eps 1e-5,
// momentum 0.1, biased variance to normalise, unbiased for the running buffer).
#include "common.h"
#include <stdlib.h>

extern "C" int64_t facl_ws_bytes(void);

namespace {

// part[rows][V] -> out[V]  (deterministic: fixed order, no atomics).  Block = 64 columns x 16 row groups; blockIdx.y
// selects a contiguous slice of `rows_per_block` rows and writes row blockIdx.y of `out` (two-level reduction: the
// partial-sum matrices are up to 12 MB and V/64 blocks alone left the kernel latency-bound at ~10 us a call).
template <typename T>
__global__ __launch_bounds__(1024) void k_reduce_rows_t1(const T* __restrict__ part, int rows, int V,
                                                         int rows_per_block, double* __restrict__ out) {
    __shared__ double red[16][64];
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int v = blockIdx.x * 64 + col;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    double s0 = 0, s1 = 0;
    if (v < V) {
        int r = r0 + rg;
        for (; r + 16 < r1; r += 32) {
            s0 += (double)part[(size_t)r * V + v];
            s1 += (double)part[(size_t)(r + 16) * V + v];
        }
        if (r < r1) s0 += (double)part[(size_t)r * V + v];
    }
    red[rg][col] = s0 + s1;
    __syncthreads();
    if (rg == 0 && v < V) {
        double t = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][col];
        out[(size_t)blockIdx.y * V + v] = t;
    }
}

__global__ __launch_bounds__(1024) void k_reduce_rows(const double* __restrict__ part, int rows, int V,
                                                      int rows_per_block, double* __restrict__ out) {
    __shared__ double red[16][64];
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int v = blockIdx.x * 64 + col;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    double s0 = 0, s1 = 0;
    if (v < V) {
        int r = r0 + rg;
        for (; r + 16 < r1; r += 32) {
            s0 += part[(size_t)r * V + v];
            s1 += part[(size_t)(r + 16) * V + v];
        }
        if (r < r1) s0 += part[(size_t)r * V + v];
    }
    red[rg][col] = s0 + s1;
    __syncthreads();
    if (rg == 0 && v < V) {
        double t = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][col];
        out[(size_t)blockIdx.y * V + v] = t;
    }
}

// One workgroup: channels strided over the threads.  `aamax` (or null): FACL_AMAX_WORDS uint32 that receive -- in every one of
// the FACL_AMAX_SLOTS slots -- the bits of a BOUND of max|gamma (y - mean) invstd + beta| over the batch these statistics were
// taken on: Samuelson's inequality, |y_i - mean| <= sigma sqrt(n - 1) for ANY n numbers, gives
// |.| <= |gamma| sqrt(n - 1) sigma invstd + |beta| (sigma invstd = sqrt(var / (var + eps)) <= 1).  It is the fp16x3 scale of the
// layer's activation (common.h) -- rigorous, no pass over the data, and within a few octaves of the true maximum.
__global__ __launch_bounds__(1024) void k_bn_finalize(const double* __restrict__ sums, int C, double count,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                      float momentum, float* __restrict__ running_mean,
                                                      float* __restrict__ running_var, float* __restrict__ bnc,
                                                      unsigned* __restrict__ aamax, unsigned* __restrict__ zamax) {
    __shared__ float red[16];
    float bound = 0.f;
    if (zamax)
        for (int i = threadIdx.x; i < FACL_AMAX_WORDS; i += blockDim.x) zamax[i] = 0u;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double mean = sums[2 * c] / count;
        double var = sums[2 * c + 1] / count - mean * mean;
        if (var < 0) var = 0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const double g = gamma[c], b = beta[c];
        bnc[0 * C + c] = (float)mean;
        bnc[1 * C + c] = (float)invstd;
        bnc[2 * C + c] = (float)(g * invstd);
        bnc[3 * C + c] = (float)(b - mean * g * invstd);
        bnc[4 * C + c] = g < 0 ? -1.0f : 1.0f;
        if (running_mean) {
            const double unb = count > 1 ? var * count / (count - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
        const double bd = (fabs(g) * sqrt((count > 1 ? count - 1 : 1) * var) * invstd + fabs(b)) * 1.001;   // 1.001: fp32 rounding of the constants
        bound = fmaxf(bound, (float)bd);
    }
    if (!aamax) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bound;
    __syncthreads();
    if (threadIdx.x < FACL_AMAX_SLOTS) {
        float m = red[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
        aamax[threadIdx.x * FACL_AMAX_STRIDE] = __float_as_uint(m);
    }
}

// max|x| of a contiguous fp32 array, RAISED into the FACL_AMAX_SLOTS slots of `amax` (zeroed by the caller, or holding the
// maximum of another tensor that shares the scale): one atomic per wave, non-negative floats order like unsigned integers
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, long long n, unsigned* __restrict__ amax) {
    float m = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) {
        const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
        atomicMax(amax + (wave & (FACL_AMAX_SLOTS - 1)) * FACL_AMAX_STRIDE, __float_as_uint(m));
    }
}

// max of relu(scale[c] y[r][c] + shift[c]) over a (R, C) row-major array: the measured activation maximum of an eval-mode
// layer (running statistics give no bound on the data), raised into `amax` like k_absmax.  C % 4 == 0.
__global__ __launch_bounds__(256) void k_rows_act_amax(const float* __restrict__ y, long long n4, int C4,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       unsigned* __restrict__ amax) {
    float m = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    int c4 = (int)(i0 % C4);
    const int dc = (int)(stride % C4);
    for (long long i = i0; i < n4; i += stride, c4 = c4 + dc >= C4 ? c4 + dc - C4 : c4 + dc) {
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        const float4 sc = reinterpret_cast<const float4*>(scale)[c4], sh = reinterpret_cast<const float4*>(shift)[c4];
        m = fmaxf(fmaxf(m, fmaxf(fmaf(sc.x, v.x, sh.x), fmaf(sc.y, v.y, sh.y))), fmaxf(fmaf(sc.z, v.z, sh.z), fmaf(sc.w, v.w, sh.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) {
        const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
        atomicMax(amax + (wave & (FACL_AMAX_SLOTS - 1)) * FACL_AMAX_STRIDE, __float_as_uint(m));
    }
}

__global__ void k_bn_eval_consts(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                 float* __restrict__ bnc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double invstd = 1.0 / sqrt((double)rv[c] + (double)eps);
    const double g = gamma[c];
    bnc[0 * C + c] = rm[c];
    bnc[1 * C + c] = (float)invstd;
    bnc[2 * C + c] = (float)(g * invstd);
    bnc[3 * C + c] = (float)((double)beta[c] - (double)rm[c] * g * invstd);
    bnc[4 * C + c] = g < 0 ? -1.0f : 1.0f;
}

// BN1 statistics analytically from the input moments: y1 = W1 x + b1  =>
//   sum_p y1_c = W1_c . sx + P b1_c ;  sum_p y1_c^2 = W1_c X2 W1_c^T + 2 b1_c W1_c . sx + P b1_c^2
__global__ void k_bn1_sums_from_moments(const double* __restrict__ mom, double count, int D,
                                        const float* __restrict__ W1, const float* __restrict__ b1, int C,
                                        double* __restrict__ sums) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double* sx = mom;
    const double* X2 = mom + D;
    double ws = 0, q = 0;
    for (int i = 0; i < D; ++i) {
        const double wi = W1[c * D + i];
        ws += wi * sx[i];
        double t = 0;
        for (int j = 0; j < D; ++j) t += X2[i * D + j] * (double)W1[c * D + j];
        q += wi * t;
    }
    const double b = b1[c];
    sums[2 * c] = ws + count * b;
    sums[2 * c + 1] = q + 2.0 * b * ws + count * b * b;
}

// The three steps between the input moments and the layer-1 table in ONE single-wave launch (training, C = 64): the sums of
// k_bn1_sums_from_moments, the constants / running statistics / activation bound of k_bn_finalize and the folded table of k_l1tab --
// the same expressions in the same order, so every output is bit-identical to the three-launch chain (2 launches fewer per step).
__global__ __launch_bounds__(64) void k_sa_bn1_chain(const double* __restrict__ mom, double count, int D,
                                                     const float* __restrict__ W1, const float* __restrict__ b1,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                     float momentum, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, double* __restrict__ sums,
                                                     float* __restrict__ bnc, unsigned* __restrict__ aamax, float* __restrict__ tab) {
    constexpr int C = 64;
    const int c = threadIdx.x;
    const double* sx = mom;
    const double* X2 = mom + D;
    double ws = 0, q = 0;
    for (int i = 0; i < D; ++i) {
        const double wi = W1[c * D + i];
        ws += wi * sx[i];
        double t = 0;
        for (int j = 0; j < D; ++j) t += X2[i * D + j] * (double)W1[c * D + j];
        q += wi * t;
    }
    const double bb = b1[c];
    const double su = ws + count * bb, sq = q + 2.0 * bb * ws + count * bb * bb;
    sums[2 * c] = su;
    sums[2 * c + 1] = sq;
    const double mean = su / count;
    double var = sq / count - mean * mean;
    if (var < 0) var = 0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double g = gamma[c], b = beta[c];
    const float scale = (float)(g * invstd), shift = (float)(b - mean * g * invstd);
    bnc[0 * C + c] = (float)mean;
    bnc[1 * C + c] = (float)invstd;
    bnc[2 * C + c] = scale;
    bnc[3 * C + c] = shift;
    bnc[4 * C + c] = g < 0 ? -1.0f : 1.0f;
    if (running_mean) {
        const double unb = count > 1 ? var * count / (count - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
    if (aamax) {
        const double bd = (fabs(g) * sqrt((count > 1 ? count - 1 : 1) * var) * invstd + fabs(b)) * 1.001;
        float bound = (float)bd;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o, 64));
        if (threadIdx.x < FACL_AMAX_SLOTS) aamax[threadIdx.x * FACL_AMAX_STRIDE] = __float_as_uint(bound);
    }
    for (int i = 0; i < 4; ++i) tab[c * 8 + i] = i < D ? scale * W1[c * D + i] : 0.0f;
    tab[c * 8 + 4] = scale * b1[c] + shift;
    tab[c * 8 + 5] = tab[c * 8 + 6] = tab[c * 8 + 7] = 0.0f;
}

// Fold BN1 into the first 1x1 conv: a1 = relu((scale*W1) x + (scale*b1 + shift)); rows of 8 floats.
// `xamax` -> `a1amax` (both or neither): with X = max|x| over all coordinates (facl_absmax), |a1_c| <= X sum_i |w'_ci| + |b'_c|:
// the bound of the layer-1 activation for eval-mode constants (train mode takes BatchNorm's own bound, facl_bn_finalize).
__global__ void k_l1tab(const float* __restrict__ W1, const float* __restrict__ b1, int D,
                        const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ tab,
                        const unsigned* __restrict__ xamax, unsigned* __restrict__ a1amax) {
    const int c = threadIdx.x;                                          // 64 threads = one wave
    const float s = scale ? scale[c] : 1.0f, t = shift ? shift[c] : 0.0f;
    float l1 = 0.f;
    for (int i = 0; i < 4; ++i) {
        const float w = i < D ? s * W1[c * D + i] : 0.0f;
        tab[c * 8 + i] = w;
        l1 += fabsf(w);
    }
    const float b = s * b1[c] + t;
    tab[c * 8 + 4] = b;
    tab[c * 8 + 5] = tab[c * 8 + 6] = tab[c * 8 + 7] = 0.0f;
    if (a1amax) {
        const float X = __uint_as_float(amax_bits(xamax));
        float bd = fmaf(l1, X, fabsf(b)) * 1.001f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bd = fmaxf(bd, __shfl_xor(bd, o, 64));
        a1amax[c * FACL_AMAX_STRIDE] = __float_as_uint(bd);
    }
}

}  // namespace

// part[rows][V] -> out[V] in ONE launch (round 4 experiment, opt-in: FACL_REDUCE_1L=1; see facl_reduce_rows for the measurement).  Grid =
// (column blocks, row slices): every workgroup sums its slice of rows into scratch[slice][V] exactly as k_reduce_rows does, then
// takes a ticket of its column block; the workgroup that draws the LAST ticket adds the slices in slice order (fixed order, no
// floating-point atomics: the result is bit-identical to the two-level form) and hands the ticket counter back at zero, so the
// launch can be replayed from a captured graph.  Tickets: the last FACL_WS_TICKET_BYTES of the workspace, zero before first use
// (the caller's duty, once) and zero again after every launch.
__global__ __launch_bounds__(1024) void k_reduce_rows_t(const double* __restrict__ part, int rows, int V, int rows_per_block,
                                                        double* __restrict__ scratch, double* __restrict__ out,
                                                        unsigned* __restrict__ tickets) {
    __shared__ double red[16][64];
    __shared__ int is_last;
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int v = blockIdx.x * 64 + col;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    double s0 = 0, s1 = 0;
    if (v < V) {
        int r = r0 + rg;
        for (; r + 16 < r1; r += 32) {
            s0 += part[(size_t)r * V + v];
            s1 += part[(size_t)(r + 16) * V + v];
        }
        if (r < r1) s0 += part[(size_t)r * V + v];
    }
    red[rg][col] = s0 + s1;
    __syncthreads();
    if (rg == 0 && v < V) {
        double t = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][col];
        scratch[(size_t)blockIdx.y * V + v] = t;
        __threadfence();                                  // the slice row is visible device-wide before the ticket is drawn
    }
    __syncthreads();
    if (threadIdx.x == 0) is_last = (atomicAdd(&tickets[blockIdx.x], 1u) == gridDim.y - 1) ? 1 : 0;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    const int nb = gridDim.y;
    double t = 0;
    if (v < V)
        for (int b = rg; b < nb; b += 16) t += scratch[(size_t)b * V + v];
    red[rg][col] = t;
    __syncthreads();
    if (rg == 0 && v < V) {
        double a = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += red[i][col];
        out[v] = a;
    }
    if (threadIdx.x == 0) tickets[blockIdx.x] = 0u;
}

// The partials occupy part[0 : rows*V] of the caller's workspace (facl_ws_bytes); the slice rows go right behind them when
// they fit, else the reduction stays single-level.  `part` IS the workspace base in every caller (its tail holds the tickets).
int facl_reduce_rows(const double* part, int rows, int V, double* out, hipStream_t st) {
    const int cb = (V + 63) / 64;
    int nb = 1;
    // two levels only when one workgroup per 64 columns would have to walk many rows (FACL_REDUCE_MIN_ROWS: A/B knob)
    static const int min_rows = getenv("FACL_REDUCE_MIN_ROWS") ? atoi(getenv("FACL_REDUCE_MIN_ROWS")) : 128;
    if (rows >= min_rows && cb < 128) {
        nb = 256 / cb;                                   // ~256 workgroups
        if (nb > rows / 32) nb = rows / 32;              // at least 32 rows (2 per row group) per block
        if (nb < 1) nb = 1;
    }
    const size_t cap = ((size_t)facl_ws_bytes() - FACL_WS_TICKET_BYTES) / sizeof(double);
    if (nb > 1 && (size_t)rows * V + (size_t)nb * V <= cap && cb <= FACL_WS_TICKET_BYTES / 4) {
        const int rpb = (rows + nb - 1) / nb;
        nb = (rows + rpb - 1) / rpb;
        double* scratch = const_cast<double*>(part) + (size_t)rows * V;
        unsigned* tickets = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(const_cast<double*>(part)) + facl_ws_bytes() - FACL_WS_TICKET_BYTES);
        // Default: TWO launches.  The single-launch form (FACL_REDUCE_1L=1) is bit-identical and was measured SLOWER on MI355X:
        // 3.22 vs 3.10 ms per step, same box, alternating runs (gpurun_out/r4i) -- 17 reductions per step, ~7 us each.  Handing data
        // from many workgroups to one inside a kernel needs a device-scope release / acquire (__threadfence), and with one L2
        // per XCD (8 of them, not coherent with each other) that is an L2 write-back + invalidate -- far dearer than the ~1.5 us a
        // kernel boundary costs in graph replay.  VERDICT r3 #5's "last-workgroup-done" folding is therefore not taken.
        static const int one_level = getenv("FACL_REDUCE_1L") ? atoi(getenv("FACL_REDUCE_1L")) : 0;
        if (!one_level) {
            hipLaunchKernelGGL(k_reduce_rows, dim3(cb, nb), dim3(1024), 0, st, part, rows, V, rpb, scratch);
            hipLaunchKernelGGL(k_reduce_rows, dim3(cb, 1), dim3(1024), 0, st, scratch, nb, V, nb, out);
        } else {
            hipLaunchKernelGGL(k_reduce_rows_t, dim3(cb, nb), dim3(1024), 0, st, part, rows, V, rpb, scratch, out, tickets);
        }
    } else {
        hipLaunchKernelGGL(k_reduce_rows, dim3(cb, 1), dim3(1024), 0, st, part, rows, V, rows, out);
    }
    return facl_launch_status();
}

// fp32 partial rows (a kernel whose per-workgroup sums are fp32 anyway writes and re-reads half the bytes): part[rows][V]
// floats -> out[V] doubles, rows added in the same fixed order as facl_reduce_rows' single level
int facl_reduce_rows_f32(const float* part, int rows, int V, double* out, hipStream_t st) {
    const int cb = (V + 63) / 64;
    hipLaunchKernelGGL((k_reduce_rows_t1<float>), dim3(cb, 1), dim3(1024), 0, st, part, rows, V, rows, out);
    return facl_launch_status();
}

extern "C" int64_t facl_ws_bytes(void) { return (int64_t)FACL_WS_ROWS * 4608 * sizeof(double) + FACL_WS_TICKET_BYTES; }

extern "C" int facl_bn_finalize(const double* sums, int C, double count, const float* gamma, const float* beta,
                                float eps, float momentum, float* running_mean, float* running_var, float* bnc,
                                uint32_t* aamax, uint32_t* zamax, void* stream) {
    if (!sums || !gamma || !beta || !bnc) return FACL_E_NULL;
    if (C < 1 || count < 1) return FACL_E_SHAPE;
    const int threads = C >= 1024 ? 1024 : (C + 63) / 64 * 64;
    hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(threads), 0, (hipStream_t)stream, sums, C, count, gamma,
                       beta, eps, momentum, running_mean, running_var, bnc, aamax, zamax);
    return facl_launch_status();
}

extern "C" int facl_absmax(const float* x, int64_t n, uint32_t* amax, void* stream) {
    if (!x || !amax) return FACL_E_NULL;
    if (n < 1) return FACL_E_SHAPE;
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_absmax, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (long long)n, amax);
    return facl_launch_status();
}

extern "C" int facl_rows_act_amax(const float* y, int64_t R, int C, const float* scale, const float* shift, uint32_t* amax,
                                  void* stream) {
    if (!y || !scale || !shift || !amax) return FACL_E_NULL;
    if (R < 1 || C < 4 || (C & 3)) return FACL_E_SHAPE;
    if ((((uintptr_t)y) | ((uintptr_t)scale) | ((uintptr_t)shift)) & 15) return FACL_E_ALIGN;
    const long long n4 = (long long)R * (C / 4);
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_rows_act_amax, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, n4, C / 4, scale, shift, amax);
    return facl_launch_status();
}

extern "C" int facl_bn_eval_consts(int C, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* bnc, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !bnc) return FACL_E_NULL;
    if (C < 1) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_bn_eval_consts, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, C, gamma, beta,
                       running_mean, running_var, eps, bnc);
    return facl_launch_status();
}

extern "C" int facl_bn1_sums_from_moments(const double* mom, double count, int D, const float* W1, const float* b1,
                                          double* sums, void* stream) {
    if (!mom || !W1 || !b1 || !sums) return FACL_E_NULL;
    if (D != 3 && D != 4) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_bn1_sums_from_moments, dim3(1), dim3(64), 0, (hipStream_t)stream, mom, count, D, W1, b1, 64,
                       sums);
    return facl_launch_status();
}

extern "C" int facl_sa_bn1_chain(const double* mom, double count, int D, const float* W1, const float* b1, const float* gamma,
                                 const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                 double* sums, float* bnc, uint32_t* aamax, float* l1tab, void* stream) {
    if (!mom || !W1 || !b1 || !gamma || !beta || !sums || !bnc || !l1tab) return FACL_E_NULL;
    if ((D != 3 && D != 4) || count < 1) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_sa_bn1_chain, dim3(1), dim3(64), 0, (hipStream_t)stream, mom, count, D, W1, b1, gamma, beta, eps, momentum,
                       running_mean, running_var, sums, bnc, aamax, l1tab);
    return facl_launch_status();
}

extern "C" int facl_sa_l1tab(const float* W1, const float* b1, int D, const float* scale, const float* shift,
                             float* l1tab, const uint32_t* xamax, uint32_t* a1amax, void* stream) {
    if (!W1 || !b1 || !l1tab || ((xamax == nullptr) != (a1amax == nullptr))) return FACL_E_NULL;
    if (D != 3 && D != 4) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_l1tab, dim3(1), dim3(64), 0, (hipStream_t)stream, W1, b1, D, scale, shift, l1tab, xamax, a1amax);
    return facl_launch_status();
}

// ================================================================================================
// Closed-form assembly between the backward passes of the SA point-MLP (csrc/sa_bwd.hip), fp64.
namespace {

// G3 = W3^T diag(g) W3,  h3' = W3^T (-(1/P) s3 dbeta3 + g (b3 - mean3)),  g = -(1/P) s3 dgamma3 invstd3
// Workgroup = 64 outputs x 4 channel quarters: thread (output, quarter) adds its 64 channels in order, the quarters meet in LDS and
// are added in quarter order (one fixed order: deterministic).  65 workgroups instead of 17 whole-W3 ones (the 256-term fp64 dots
// of one thread per output were a 14 us serial chain between k_sa_bwd0 and k_sa_bwd1).
__global__ __launch_bounds__(256) void k_sa_bwd_consts3(const double* __restrict__ sums0, const float* __restrict__ bnc3,
                                                        const float* __restrict__ W3, const float* __restrict__ b3,
                                                        double P, float* __restrict__ G3, float* __restrict__ h3) {
    __shared__ double g[256], hc[256];
    __shared__ double red[3][64];
    const int c = threadIdx.x;
    {
        const double mean = bnc3[c], inv = bnc3[256 + c], sc = bnc3[512 + c];
        const double gg = -(sc * sums0[2 * c + 1] * inv) / P;
        g[c] = gg;
        hc[c] = -(sc * sums0[2 * c]) / P + gg * ((double)b3[c] - mean);
    }
    __syncthreads();
    const int j = threadIdx.x & 63, qt = threadIdx.x >> 6;          // output column, channel quarter
    const int k = blockIdx.x;                                        // 0..63: row k of G3; 64: h3
    double s = 0;
    if (k < 64) {
#pragma unroll 8
        for (int cc = 64 * qt; cc < 64 * qt + 64; ++cc) s += g[cc] * (double)W3[cc * 64 + k] * (double)W3[cc * 64 + j];
    } else {
#pragma unroll 8
        for (int cc = 64 * qt; cc < 64 * qt + 64; ++cc) s += hc[cc] * (double)W3[cc * 64 + j];
    }
    if (qt) red[qt - 1][j] = s;
    __syncthreads();
    if (qt == 0) {
        s += red[0][j]; s += red[1][j]; s += red[2][j];
        if (k < 64) G3[k * 64 + j] = (float)s;
        else h3[j] = (float)s;
    }
}

// bw2 (4,64): dy2 = scale2*dz2 + A + B*(y2 - mean2)
__global__ void k_sa_bwd_consts2(const double* __restrict__ sums1, const float* __restrict__ bnc2, double P,
                                 float* __restrict__ bw2) {
    const int c = threadIdx.x;
    if (c >= 64) return;
    const double inv = bnc2[64 + c], sc = bnc2[128 + c];
    bw2[c] = (float)sc;
    bw2[64 + c] = (float)(-sc * sums1[2 * c] / P);
    bw2[128 + c] = (float)(-sc * inv * sums1[2 * c + 1] / P);
    bw2[192 + c] = bnc2[c];
}

// parameter gradients of layers 3, 2 and 1 from the reduced partial sums
__global__ __launch_bounds__(256) void k_sa_bwd_final(
    const double* __restrict__ out3, const double* __restrict__ sums0_g, const double* __restrict__ sums0_l,
    const float* __restrict__ bnc3, const float* __restrict__ W3, const float* __restrict__ b3,
    const double* __restrict__ out2, const double* __restrict__ sums1_l, const double* __restrict__ R1_g,
    const double* __restrict__ mom_l, const float* __restrict__ bnc1, const float* __restrict__ W1,
    const float* __restrict__ b1, int D, double P, float* __restrict__ dW3, float* __restrict__ dg3,
    float* __restrict__ dbe3, float* __restrict__ dW2, float* __restrict__ dg2, float* __restrict__ dbe2,
    float* __restrict__ dW1, float* __restrict__ dg1, float* __restrict__ dbe1) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    const double* sparse3 = out3;
    const double* gram2 = out3 + 256 * 64;
    const double* s2 = out3 + 256 * 64 + 64 * 64;
    if (o < 16384) {                                            // dW3[c][k]
        const int c = o >> 6, k = o & 63;
        const double mean = bnc3[c], inv = bnc3[256 + c], sc = bnc3[512 + c];
        double wg = 0;
        for (int j = 0; j < 64; ++j) wg += (double)W3[c * 64 + j] * gram2[j * 64 + k];
        const double yhat_a2 = inv * (wg + ((double)b3[c] - mean) * s2[k]);       // sum_p yhat3[p,c] a2[p,k]
        dW3[o] = (float)(sparse3[o] - (sc / P) * (sums0_g[2 * c] * s2[k] + sums0_g[2 * c + 1] * yhat_a2));
        return;
    }
    int r = o - 16384;
    if (r < 256) { dbe3[r] = (float)sums0_l[2 * r]; dg3[r] = (float)sums0_l[2 * r + 1]; return; }
    r -= 256;
    if (r < 4096) { dW2[r] = (float)out2[r]; return; }
    r -= 4096;
    if (r < 64) { dbe2[r] = (float)sums1_l[2 * r]; dg2[r] = (float)sums1_l[2 * r + 1]; return; }
    r -= 64;
    if (r < 64) {                                               // layer 1, channel c = r
        const int c = r;
        const double* R1_l = out2 + 4096;                       // (8,64): rows x_d (d<D), then sum dz1
        const double mean = bnc1[c], inv = bnc1[64 + c], sc = bnc1[128 + c];
        const double bmm = (double)b1[c] - mean;
        double wr_l = 0, wr_g = 0;
        for (int d = 0; d < D; ++d) {
            wr_l += (double)W1[c * D + d] * R1_l[d * 64 + c];
            wr_g += (double)W1[c * D + d] * R1_g[d * 64 + c];
        }
        const double dbe_l = R1_l[D * 64 + c], dbe_g = R1_g[D * 64 + c];
        const double dg_l = inv * (wr_l + bmm * dbe_l), dg_g = inv * (wr_g + bmm * dbe_g);
        dbe1[c] = (float)dbe_l;
        dg1[c] = (float)dg_l;
        const double* sx = mom_l;
        const double* X2 = mom_l + D;
        for (int d = 0; d < D; ++d) {
            double wx = 0;
            for (int j = 0; j < D; ++j) wx += (double)W1[c * D + j] * X2[j * D + d];
            const double yhat_x = inv * (wx + bmm * sx[d]);                           // sum_p yhat1[p,c] x[p,d]
            dW1[c * D + d] = (float)(sc * (R1_l[d * 64 + c] - (dbe_g / P) * sx[d] - (dg_g / P) * yhat_x));
        }
    }
}

}  // namespace

extern "C" int facl_sa_bwd_consts3(const double* sums0, const float* bnc3, const float* W3, const float* b3, double P,
                                   float* G3, float* h3, void* stream) {
    if (!sums0 || !bnc3 || !W3 || !b3 || !G3 || !h3) return FACL_E_NULL;
    if (P < 1) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_sa_bwd_consts3, dim3(65), dim3(256), 0, (hipStream_t)stream, sums0, bnc3, W3, b3, P, G3, h3);
    return facl_launch_status();
}

// BN backward of a row layer: (dbeta, dgamma) sums -> the two fp32 parameter gradients (THIS rank's sums) and the (2,C)
// constants kk = (dbeta/P, dgamma/P) of the SyncBN-reduced sums that facl_rows_bwd_apply / facl_segmax_bwd_apply take.
__global__ void k_bn_bwd_consts(const double* __restrict__ sums_l, const double* __restrict__ sums_g, int C, double P,
                                float* __restrict__ dbeta, float* __restrict__ dgamma, float* __restrict__ kk) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = (float)sums_l[2 * c];
    dgamma[c] = (float)sums_l[2 * c + 1];
    // same arithmetic as the torch glue it replaces: fp32 sum times fp32 reciprocal count
    const float inv = (float)(1.0 / P);
    kk[c] = (float)sums_g[2 * c] * inv;
    kk[C + c] = (float)sums_g[2 * c + 1] * inv;
}

extern "C" int facl_bn_bwd_consts(const double* sums_local, const double* sums_global, int C, double P, float* dbeta,
                                  float* dgamma, float* kk, void* stream) {
    if (!sums_local || !sums_global || !dbeta || !dgamma || !kk) return FACL_E_NULL;
    if (C < 1 || P < 1) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_bn_bwd_consts, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums_local, sums_global,
                       C, P, dbeta, dgamma, kk);
    return facl_launch_status();
}

extern "C" int facl_sa_bwd_consts2(const double* sums1, const float* bnc2, double P, float* bw2, void* stream) {
    if (!sums1 || !bnc2 || !bw2) return FACL_E_NULL;
    if (P < 1) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_sa_bwd_consts2, dim3(1), dim3(64), 0, (hipStream_t)stream, sums1, bnc2, P, bw2);
    return facl_launch_status();
}

extern "C" int facl_sa_bwd_final(const double* out3, const double* sums0_g, const double* sums0_l, const float* bnc3,
                                 const float* W3, const float* b3, const double* out2, const double* sums1_l,
                                 const double* R1_g, const double* mom_l, const float* bnc1, const float* W1,
                                 const float* b1, int D, double P, float* dW3, float* dg3, float* dbe3, float* dW2,
                                 float* dg2, float* dbe2, float* dW1, float* dg1, float* dbe1, void* stream) {
    if (!out3 || !sums0_g || !sums0_l || !bnc3 || !W3 || !b3 || !out2 || !sums1_l || !R1_g || !mom_l || !bnc1 || !W1 ||
        !b1 || !dW3 || !dg3 || !dbe3 || !dW2 || !dg2 || !dbe2 || !dW1 || !dg1 || !dbe1)
        return FACL_E_NULL;
    if ((D != 3 && D != 4) || P < 1) return FACL_E_SHAPE;
    const int total = 16384 + 256 + 4096 + 64 + 64;
    hipLaunchKernelGGL(k_sa_bwd_final, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, out3, sums0_g,
                       sums0_l, bnc3, W3, b3, out2, sums1_l, R1_g, mom_l, bnc1, W1, b1, D, P, dW3, dg3, dbe3, dW2, dg2,
                       dbe2, dW1, dg1, dbe1);
    return facl_launch_status();
}
