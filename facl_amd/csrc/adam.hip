// Adam step of the training loop (cn3d_train_motion_GL.py:180,:332: torch.optim.Adam(lr 3e-4, betas (0.5, 0.999), eps 1e-6),
// no weight decay, no amsgrad) for ALL parameters in one launch.  torch's fused multi-tensor kernel walks 64 K-element
// chunks: 2.36 M parameters are 36 workgroups (47 us on 256 CUs); here a workgroup takes 2048 elements (~1150 workgroups,
// HBM-bound: 7 x 9.4 MB per step).  Tensor pointers travel BY VALUE in the kernel arguments (no device-side table to keep
// coherent; a HIP graph bakes them at capture time, when the gradient buffers of the captured step are fixed).
//   facl_adam_prep: step += 1; consts = (lr / (1 - b1^t), 1 / sqrt(1 - b2^t))      -- one thread, device-resident lr / step
//   facl_adam_apply: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= step_size * m / (sqrt(v) * inv_sqrt_bc2 + eps)
#include "common.h"
#include <math.h>

#define FACL_ADAM_MAX_TENSORS 64
#define FACL_ADAM_CHUNK 2048

struct FaclAdamTable {
    float* p[FACL_ADAM_MAX_TENSORS];
    const float* g[FACL_ADAM_MAX_TENSORS];
    float* m[FACL_ADAM_MAX_TENSORS];
    float* v[FACL_ADAM_MAX_TENSORS];
    int n[FACL_ADAM_MAX_TENSORS];
    int chunk0[FACL_ADAM_MAX_TENSORS + 1];        // first chunk of tensor i (prefix sums of ceil(n / CHUNK))
    int nt;
};

namespace {

__global__ void k_adam_prep(const float* __restrict__ lr, float* __restrict__ step, float b1, float b2,
                            float* __restrict__ consts) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float t = step[0] + 1.f;
    step[0] = t;
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    consts[0] = (float)((double)lr[0] / bc1);
    consts[1] = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(256) void k_adam_apply(FaclAdamTable tb, const float* __restrict__ consts, float b1, float b2,
                                                    float eps) {
    // tensor of this chunk: binary search in the prefix table (wave-uniform)
    int lo = 0, hi = tb.nt;
    const int c = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tb.chunk0[mid] <= c) lo = mid; else hi = mid;
    }
    const int t = lo, base = (c - tb.chunk0[t]) * FACL_ADAM_CHUNK;
    const int n = tb.n[t];
    float* __restrict__ p = tb.p[t];
    const float* __restrict__ g = tb.g[t];
    float* __restrict__ m = tb.m[t];
    float* __restrict__ v = tb.v[t];
    const float step_size = consts[0], isb2 = consts[1];
#pragma unroll
    for (int k = 0; k < FACL_ADAM_CHUNK / 256; ++k) {
        const int i = base + k * 256 + threadIdx.x;
        if (i < n) {
            const float gi = g[i];
            const float mi = b1 * m[i] + (1.f - b1) * gi;
            const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
            m[i] = mi;
            v[i] = vi;
            p[i] -= step_size * (mi / (sqrtf(vi) * isb2 + eps));
        }
    }
}

}  // namespace

extern "C" int facl_adam_prep(const float* lr, float* step, float b1, float b2, float* consts, void* stream) {
    if (!lr || !step || !consts) return FACL_E_NULL;
    hipLaunchKernelGGL(k_adam_prep, dim3(1), dim3(64), 0, (hipStream_t)stream, lr, step, b1, b2, consts);
    return facl_launch_status();
}

// p / g / m / v: HOST arrays of nt device pointers; n: HOST array of element counts
extern "C" int facl_adam_apply(int nt, float* const* p, const float* const* g, float* const* m, float* const* v, const int* n,
                               const float* consts, float b1, float b2, float eps, void* stream) {
    if (!p || !g || !m || !v || !n || !consts) return FACL_E_NULL;
    if (nt < 1 || nt > FACL_ADAM_MAX_TENSORS) return FACL_E_SHAPE;
    FaclAdamTable tb;
    int chunks = 0;
    for (int i = 0; i < nt; ++i) {
        if (!p[i] || !g[i] || !m[i] || !v[i]) return FACL_E_NULL;
        if (n[i] < 1) return FACL_E_SHAPE;
        tb.p[i] = p[i]; tb.g[i] = g[i]; tb.m[i] = m[i]; tb.v[i] = v[i]; tb.n[i] = n[i];
        tb.chunk0[i] = chunks;
        chunks += (n[i] + FACL_ADAM_CHUNK - 1) / FACL_ADAM_CHUNK;
    }
    tb.chunk0[nt] = chunks;
    tb.nt = nt;
    hipLaunchKernelGGL(k_adam_apply, dim3(chunks), dim3(256), 0, (hipStream_t)stream, tb, consts, b1, b2, eps);
    return facl_launch_status();
}
