// Small fp64 "finalize" kernels between the passes of the train-mode BatchNorm pipeline:
// partial-sum reduction, BN statistics -> (mean, invstd, scale, shift, sign), running-stat update.
// BN semantics: cn3d_model_conbag.py:46,50,54,64,68,72,84 (nn.BatchNorm2d/1d defaults: eps 1e-5,
// momentum 0.1, biased variance to normalise, unbiased for the running buffer).
#include "common.h"

namespace {

// part[rows][V] -> out[V]  (deterministic: fixed order, no atomics).  Block = 64 columns x 16 row groups.
__global__ __launch_bounds__(1024) void k_reduce_rows(const double* __restrict__ part, int rows, int V,
                                                      double* __restrict__ out) {
    __shared__ double red[16][64];
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int v = blockIdx.x * 64 + col;
    double s0 = 0, s1 = 0;
    if (v < V) {
        int r = rg;
        for (; r + 16 < rows; r += 32) {
            s0 += part[(size_t)r * V + v];
            s1 += part[(size_t)(r + 16) * V + v];
        }
        if (r < rows) s0 += part[(size_t)r * V + v];
    }
    red[rg][col] = s0 + s1;
    __syncthreads();
    if (rg == 0 && v < V) {
        double t = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][col];
        out[v] = t;
    }
}

__global__ void k_bn_finalize(const double* __restrict__ sums, int C, double count, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float eps, float momentum,
                              float* __restrict__ running_mean, float* __restrict__ running_var,
                              float* __restrict__ bnc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
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

// Fold BN1 into the first 1x1 conv: a1 = relu((scale*W1) x + (scale*b1 + shift)); rows of 8 floats.
__global__ void k_l1tab(const float* __restrict__ W1, const float* __restrict__ b1, int D,
                        const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ tab) {
    const int c = threadIdx.x;
    if (c >= 64) return;
    const float s = scale ? scale[c] : 1.0f, t = shift ? shift[c] : 0.0f;
    for (int i = 0; i < 4; ++i) tab[c * 8 + i] = i < D ? s * W1[c * D + i] : 0.0f;
    tab[c * 8 + 4] = s * b1[c] + t;
    tab[c * 8 + 5] = tab[c * 8 + 6] = tab[c * 8 + 7] = 0.0f;
}

}  // namespace

int facl_reduce_rows(const double* part, int rows, int V, double* out, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_rows, dim3((V + 63) / 64), dim3(1024), 0, st, part, rows, V, out);
    return facl_launch_status();
}

extern "C" int64_t facl_ws_bytes(void) { return (int64_t)FACL_WS_ROWS * 4608 * sizeof(double); }

extern "C" int facl_bn_finalize(const double* sums, int C, double count, const float* gamma, const float* beta,
                                float eps, float momentum, float* running_mean, float* running_var, float* bnc,
                                void* stream) {
    if (!sums || !gamma || !beta || !bnc) return FACL_E_NULL;
    if (C < 1 || count < 1) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, C, count, gamma,
                       beta, eps, momentum, running_mean, running_var, bnc);
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

extern "C" int facl_sa_l1tab(const float* W1, const float* b1, int D, const float* scale, const float* shift,
                             float* l1tab, void* stream) {
    if (!W1 || !b1 || !l1tab) return FACL_E_NULL;
    if (D != 3 && D != 4) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_l1tab, dim3(1), dim3(64), 0, (hipStream_t)stream, W1, b1, D, scale, shift, l1tab);
    return facl_launch_status();
}
