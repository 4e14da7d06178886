// The two optional loss terms of the training loop (both switched off in the shipped loop):
//   facl_sinkhorn  = distributed_sinkhorn + shoot_infs (cn3d_model_conbag.py:391-425) on a (K prototypes x n samples)
//                    score matrix: ~25 tiny torch launches and a host round trip (torch.nonzero) become one launch;
//   facl_kmeans    = KMeans (cn3d_train_motion_GL.py:54-70): Lloyd iterations from the first K rows, argmin with the
//                    first minimum on ties, empty clusters keep count 1 (centroid 0).
// Both problems are a few thousand elements: ONE workgroup, phases separated by workgroup barriers, every sum in a fixed
// order (deterministic).  Latency-bound by construction; what they remove is launches and the sync.
#include "common.h"
#include <math.h>

namespace {

constexpr int SK_T = 1024;

__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum_f32(v);
    __syncthreads();
    if (lane_id() == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < SK_T / 64; ++w) r += sm[w];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if (lane_id() == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sm[0];
    for (int w = 1; w < SK_T / 64; ++w) r = fmaxf(r, sm[w]);
    return r;
}
__device__ __forceinline__ int block_or(int v, int* sm) {
    v = __any(v);
    __syncthreads();
    if (lane_id() == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = 0;
    for (int w = 0; w < SK_T / 64; ++w) r |= sm[w];
    return r;
}

// shoot_infs on `n` elements of q: +-inf -> max of the tensor with those entries zeroed (:409-425)
__device__ void shoot_infs_dev(float* q, int n, float* smf, int* smi) {
    int any = 0;
    for (int i = threadIdx.x; i < n; i += SK_T) any |= isinf(q[i]) ? 1 : 0;
    if (!block_or(any, smi)) return;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += SK_T) mx = fmaxf(mx, isinf(q[i]) ? 0.f : q[i]);
    mx = block_max(mx, smf);
    for (int i = threadIdx.x; i < n; i += SK_T)
        if (isinf(q[i])) q[i] = mx;
    __syncthreads();
}

// Q (R, C) row-major in `q` (scratch copy of the input), out (C, R)
__global__ __launch_bounds__(SK_T) void k_sinkhorn(const float* __restrict__ Qin, int R, int C, int iters,
                                                   float* __restrict__ q, float* __restrict__ u,
                                                   float* __restrict__ out) {
    __shared__ float smf[SK_T / 64];
    __shared__ int smi[SK_T / 64];
    const int n = R * C;
    for (int i = threadIdx.x; i < n; i += SK_T) q[i] = Qin[i];
    __syncthreads();
    shoot_infs_dev(q, n, smf, smi);
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += SK_T) s += q[i];
    s = block_sum(s, smf);
    for (int i = threadIdx.x; i < n; i += SK_T) q[i] /= s;
    __syncthreads();
    const float r = 1.f / (float)R, c = 1.f / (float)C;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    for (int it = 0; it < iters; ++it) {
        for (int row = wave; row < R; row += SK_T / 64) {                 // u = r / rowsum
            float t = 0.f;
            for (int j = lane; j < C; j += 64) t += q[row * C + j];
            t = wave_sum_f32(t);
            if (lane == 0) u[row] = r / t;
        }
        __syncthreads();
        shoot_infs_dev(u, R, smf, smi);
        for (int i = threadIdx.x; i < n; i += SK_T) q[i] *= u[i / C];
        __syncthreads();
        for (int j = threadIdx.x; j < C; j += SK_T) {                     // column scaling
            float t = 0.f;
            for (int row = 0; row < R; ++row) t += q[row * C + j];
            const float f = c / t;
            for (int row = 0; row < R; ++row) q[row * C + j] *= f;
        }
        __syncthreads();
    }
    for (int j = threadIdx.x; j < C; j += SK_T) {                         // (Q / colsum)^T
        float t = 0.f;
        for (int row = 0; row < R; ++row) t += q[row * C + j];
        for (int row = 0; row < R; ++row) out[(size_t)j * R + row] = q[row * C + j] / t;
    }
}

// x (N, D), K <= N clusters.  labels (N) int32, cent (K, D); scratch: dist-free (each (point) thread block-loops).
__global__ __launch_bounds__(SK_T) void k_kmeans(const float* __restrict__ x, int N, int D, int K, int iters,
                                                 int32_t* __restrict__ labels, float* __restrict__ cent,
                                                 int32_t* __restrict__ counts) {
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int K0 = K < N ? K : N;                                         // c = x[:K] has only N rows when N < K (:56)
    for (int i = threadIdx.x; i < K0 * D; i += SK_T) cent[i] = x[i];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const int Kc = it == 0 ? K0 : K;                                  // later iterations: all K (the extra ones are 0)
        // assignment: one wave per point, lanes over the channels; first minimum wins (argmin)
        for (int p = wave; p < N; p += SK_T / 64) {
            float best = INFINITY;
            int bk = 0;
            for (int k = 0; k < Kc; ++k) {
                float t = 0.f;
                for (int d = lane; d < D; d += 64) {
                    const float df = x[(size_t)p * D + d] - cent[(size_t)k * D + d];
                    t = fmaf(df, df, t);
                }
                t = wave_sum_f32(t);
                if (t < best) { best = t; bk = k; }
            }
            if (lane == 0) labels[p] = bk;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += SK_T) {
            int n = 0;
            for (int p = 0; p < N; ++p) n += labels[p] == k;
            counts[k] = n > 0 ? n : 1;                                    // empty cluster: count 1, centroid 0 (:64-66)
        }
        __syncthreads();
        for (int i = threadIdx.x; i < K * D; i += SK_T) {                 // c = scatter_add(x by label) / count, in index order
            const int k = i / D, d = i - k * D;
            float t = 0.f;
            for (int p = 0; p < N; ++p)
                if (labels[p] == k) t += x[(size_t)p * D + d];
            cent[i] = t / (float)counts[k];
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int facl_sinkhorn(const float* Q, int R, int C, int iters, float* scratch /* R*C + R */, float* out /* (C,R) */,
                             void* stream) {
    if (!Q || !scratch || !out) return FACL_E_NULL;
    if (R < 1 || C < 1 || iters < 0 || (long long)R * C > (1 << 24)) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_sinkhorn, dim3(1), dim3(SK_T), 0, (hipStream_t)stream, Q, R, C, iters, scratch, scratch + (size_t)R * C, out);
    return facl_launch_status();
}

extern "C" int facl_kmeans(const float* x, int N, int D, int K, int iters, int32_t* labels, float* cent, int32_t* counts,
                           void* stream) {
    if (!x || !labels || !cent || !counts) return FACL_E_NULL;
    if (N < 1 || D < 1 || K < 1 || iters < 1) return FACL_E_SHAPE;
    hipLaunchKernelGGL(k_kmeans, dim3(1), dim3(SK_T), 0, (hipStream_t)stream, x, N, D, K, iters, labels, cent, counts);
    return facl_launch_status();
}
