#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + threadIdx.x * 4));
    out[threadIdx.x * 4 + 0] = v[0]; out[threadIdx.x * 4 + 1] = v[1]; out[threadIdx.x * 4 + 2] = v[2]; out[threadIdx.x * 4 + 3] = v[3];
}
