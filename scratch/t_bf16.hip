#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mi, unsigned& lo) {
    f32x2 v = {x0, x1};
    bf16x2 h = __builtin_convertvector(v, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    f32x2 rv = {r0, r1};
    bf16x2 m = __builtin_convertvector(rv, bf16x2);
    mi = __builtin_bit_cast(unsigned, m);
    float s0 = r0 - __builtin_bit_cast(float, mi << 16), s1 = r1 - __builtin_bit_cast(float, mi & 0xffff0000u);
    f32x2 sv = {s0, s1};
    bf16x2 l = __builtin_convertvector(sv, bf16x2);
    lo = __builtin_bit_cast(unsigned, l);
}
__global__ void k(const float* x, unsigned* o, float* c) {
    int t = threadIdx.x;
    unsigned h, m, l;
    split2(x[2 * t], x[2 * t + 1], h, m, l);
    o[3 * t] = h; o[3 * t + 1] = m; o[3 * t + 2] = l;
    bf16x8 a = *(const bf16x8*)(o + 4 * t), b = *(const bf16x8*)(o + 1024 + 4 * t);
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) c[16 * t + i] = acc[i];
}
