// build + run on a GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/l2bw tools/probe_l2bw.hip && /tmp/l2bw   (output of round 4: profiles/r04_probe_l2bw.txt)
// per-CU fill bandwidth probe: each workgroup re-reads a region of `region` bytes `iters` times with 16-byte loads, DEPTH loads in flight per thread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ src, size_t region4, int iters, int shared, float* out) {
    const float4* base = src + (shared ? (size_t)(blockIdx.x & 7) * region4 : (size_t)blockIdx.x * region4);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        for (size_t i = threadIdx.x; i + (DEPTH - 1) * 256 < region4; i += DEPTH * 256) {
            float4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { typedef float f4 __attribute__((ext_vector_type(4))); const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(base + i + d * 256)); v[d] = make_float4(t.x, t.y, t.z, t.w); }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
}
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read_c(const float4* __restrict__ src, size_t region4, int iters, int shared, float* out) {
    const float4* base = src + (shared ? (size_t)(blockIdx.x & 7) * region4 : (size_t)blockIdx.x * region4);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        for (size_t i = threadIdx.x; i + (DEPTH - 1) * 256 < region4; i += DEPTH * 256) {
            float4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = base[i + d * 256];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
}
int main() {
    const int nwg_list[2] = {256, 512};
    float4* buf; float* out;
    const size_t total = (size_t)512 << 20;
    hipMalloc(&buf, total); hipMalloc(&out, 4); hipMemset(buf, 0, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t regions[4] = {64 << 10, 256 << 10, 1 << 20, 1 << 20};
    for (int ri = 0; ri < 3; ++ri)
      for (int shared = 0; shared < 2; ++shared)
        for (int wi = 0; wi < 2; ++wi)
          for (int cached = 0; cached < 2; ++cached) {
            const int nwg = nwg_list[wi];
            const size_t region = regions[ri], region4 = region / 16;
            const int iters = (int)((size_t)(64 << 20) / region);          // 64 MB per workgroup
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (cached) hipLaunchKernelGGL(k_read_c<8>, dim3(nwg), dim3(256), 0, 0, buf, region4, iters, shared, out);
                else hipLaunchKernelGGL(k_read<8>, dim3(nwg), dim3(256), 0, 0, buf, region4, iters, shared, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)nwg * region * iters;
            printf("region %4zu KB %s nwg %d %s: %.3f ms  %.2f TB/s  %.1f KB/us per CU\n", region >> 10, shared ? "shared-by-8" : "private", nwg,
                   cached ? "plain" : "nontemporal", ms, bytes / ms / 1e9, bytes / 256 / ms / 1e3 / 1024);
          }
    return 0;
}
