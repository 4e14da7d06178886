// placement probe: which workgroups share a CU, in what order (HW_ID / XCC_ID / start time)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void k(unsigned long long* out, int spin) {
    extern __shared__ char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // busy for a while so that the grid needs several rounds
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    lds[threadIdx.x] = (char)x;
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        out[4 * b] = hw; out[4 * b + 1] = xcc; out[4 * b + 2] = t0; out[4 * b + 3] = t1 + (lds[0] == 77 ? 1 : 0);
    }
}
int main() {
    const int gx = 4, gy = 384, n = gx * gy;
    unsigned long long* d;
    hipMalloc(&d, n * 4 * sizeof(unsigned long long));
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024);
    hipLaunchKernelGGL(k, dim3(gx, gy), dim3(256), 68 * 1024, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(n * 4);
    hipMemcpy(h.data(), d, n * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < n; ++b) tmin = h[4 * b + 2] < tmin ? h[4 * b + 2] : tmin;
    std::map<unsigned long long, std::vector<int>> cu;
    for (int b = 0; b < n; ++b) {
        const unsigned hw = (unsigned)h[4 * b], xcc = (unsigned)h[4 * b + 1];
        const unsigned long long key = ((unsigned long long)(xcc & 0xf) << 32) | (hw & 0x0000ff00u);   // se / sh / cu
        cu[key].push_back(b);
    }
    printf("distinct (xcc, hw_id & ~0xff) keys: %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) {
        if (shown++ >= 6) break;
        printf("key xcc=%llu hw=%08llx :", kv.first >> 32, kv.first & 0xffffffffull);
        for (int b : kv.second) printf(" b%d(t0=%llu,t1=%llu,hw=%08x)", b, h[4 * b + 2] - tmin, h[4 * b + 3] - tmin, (unsigned)h[4 * b]);
        printf("\n");
    }
    // first 24 blocks: raw
    for (int b = 0; b < 24; ++b) printf("b%d hw=%08x xcc=%llx t0=%llu\n", b, (unsigned)h[4 * b], h[4 * b + 1], h[4 * b + 2] - tmin);
    return 0;
}
