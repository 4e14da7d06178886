// Shared device helpers for libfacl_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/facl_hip.h"

#define FACL_WAVE 64

static inline int facl_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// lanes strictly below this lane, as a 64-bit mask
__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- MFMA f32 32x32x2 helpers (gfx950) ---------------------------------------------------
// D[i][j] += A[i][0..1] * B[0..1][j]; lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D: lane l holds column j = l&31, rows rowmap(r, l>>5), r = 0..15.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ constexpr int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// "Fragment layout" of a (64 positions x 64 channels) fp32 activation tile in HBM (one "unit"):
// element (p, c) lives at  ((ct*2 + rt)*4 + r4)*256 + lane*4 + e   with
//   ct = p>>5, q = p&31, rt = c>>5, r4 = (c&31)>>3, h = (c>>2)&1, e = c&3, lane = 32*h + q.
// It is exactly the register image of a transposed-orientation MFMA result D^T[c][p] (lane =
// position, register = channel) and, equally, the A-operand image of the next layer, so tiles
// move HBM <-> registers as 16 perfectly coalesced 1-KiB float4 accesses, with no LDS transpose.
#define FACL_UNIT 64
#define FACL_UNIT_ELEMS 4096

// partial-sum workspace: every wave (or block) of a reducing kernel writes one row of doubles
#define FACL_WS_ROWS 4096
