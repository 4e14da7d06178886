// Adam step of the training loop (cn3d_train_motion_GL.py:180,:332: torch.optim.Adam(lr 3e-4, betas (0.5, 0.999), eps 1e-6),
// no weight decay, no amsgrad) for ALL parameters in one launch.  torch's fused multi-tensor kernel walks 64 K-element
// chunks: 2.36 M parameters are 36 workgroups (47 us on 256 CUs); here a workgroup takes 2048 elements (~1150 workgroups,
// HBM-bound: 7 x 9.4 MB per step).  Tensor pointers travel BY VALUE in the kernel arguments (no device-side table to keep
// coherent; a HIP graph bakes them at capture time, when the gradient buffers of the captured step are fixed).
//   facl_adam_prep: step += 1; consts = (lr / (1 - b1^t), 1 / sqrt(1 - b2^t))      -- one thread, device-resident lr / step
//   facl_adam_apply: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= step_size * m / (sqrt(v) * inv_sqrt_bc2 + eps)
#include "common.h"
#include <math.h>
#include <stdint.h>

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
    // 16 bytes per lane where the tensor allows it (every tensor of the model does): a quarter of the memory instructions of the
    // scalar walk below, all of a thread's loads in flight before its first store
    if (!(n & 3) && !((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15)) {
        float4 g4[2], m4[2], v4[2], p4[2];
        int i4[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            i4[k] = (base >> 2) + k * 256 + threadIdx.x;
            if (4 * i4[k] < n) {
                g4[k] = reinterpret_cast<const float4*>(g)[i4[k]]; m4[k] = reinterpret_cast<const float4*>(m)[i4[k]];
                v4[k] = reinterpret_cast<const float4*>(v)[i4[k]]; p4[k] = reinterpret_cast<const float4*>(p)[i4[k]];
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (4 * i4[k] >= n) continue;
            float gg[4] = {g4[k].x, g4[k].y, g4[k].z, g4[k].w}, mm[4] = {m4[k].x, m4[k].y, m4[k].z, m4[k].w};
            float vv[4] = {v4[k].x, v4[k].y, v4[k].z, v4[k].w}, pp[4] = {p4[k].x, p4[k].y, p4[k].z, p4[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float mi = b1 * mm[e] + (1.f - b1) * gg[e];
                const float vi = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
                mm[e] = mi; vv[e] = vi;
                pp[e] -= step_size * (mi / (sqrtf(vi) * isb2 + eps));
            }
            reinterpret_cast<float4*>(m)[i4[k]] = make_float4(mm[0], mm[1], mm[2], mm[3]);
            reinterpret_cast<float4*>(v)[i4[k]] = make_float4(vv[0], vv[1], vv[2], vv[3]);
            reinterpret_cast<float4*>(p)[i4[k]] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        }
        return;
    }
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
