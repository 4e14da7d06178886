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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: `done` = one flag per device ordinal (a
// process may drive several devices through this C ABI; a forward on the main thread can race a backward on the autograd
// thread, in which case the attribute is merely set twice).  Returns a HIP error code or 0.
static inline int facl_set_dynamic_lds(bool (&done)[64], const void* const* fns, int nfns, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (done[dev]) return 0;
    for (int i = 0; i < nfns; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
    }
    done[dev] = true;
    return 0;
}

// Phase offset between the workgroups that share a CU.  Co-resident workgroups of one launch start together and do identical work,
// so they stay in lockstep: every wave of a SIMD waits for memory at the same time and competes for the pipes at the same time.
// The workgroup in the CU's second thread-group slot (HW_ID.TG_ID bit 0) waits `ticks` of the 100-MHz real-time counter ONCE, at
// its start; in a persistent (grid-stride) kernel that offset then persists.  ticks <= 0: no-op.
__device__ __forceinline__ void facl_phase_wait(int ticks) {
    if (ticks <= 0) return;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((hw >> 16) & 1u) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(32);
    }
}

// "sign of gamma" arguments (facl_sa_fwd3, facl_gemm_fwd_segmax): callers may pass the BatchNorm weight itself -- the
// kernels only use its sign, with sign(0) = +1 (a +-1 array, the round-1 convention, maps onto itself)
__device__ __forceinline__ float sgn_of(float g) { return g < 0.f ? -1.f : 1.f; }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ReLU that keeps NaN (torch.relu / nn.ReLU do; fmaxf(NaN, 0) = 0 would turn a poisoned activation into a clean zero):
// max(x, 0) + 0 * x -- the product is (+-)0 for every finite x and NaN for NaN (and for +-inf: relu(inf) comes out NaN
// instead of inf, relu(-inf) NaN instead of 0: still loud).  Deliberately NOT `x < 0 ? 0 : x`: in k_sa_fwd2_sb that form,
// behind the SLP-packed v_pk_fma_f32 chain of layer 1, produced wrong and run-to-run different y2 tiles on gfx950 (round 4,
// gpurun_out/r4e: 6.7e-2 instead of 1.2e-7 with `fmaxf`; v_cmp_ngt_f32 + v_cndmask_b32 on VCC; cause not isolated).
__device__ __forceinline__ float relu_nan(float x) { return fmaf(x, 0.f, fmaxf(x, 0.f)); }

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
// the last bytes of the workspace: ticket counters of the single-launch partial-row reduction (finalize.hip), zero between launches
#define FACL_WS_TICKET_BYTES 4096

// ---- split-bf16 ("bf16x6") helpers ---------------------------------------------------------------------------
// gfx950 runs the fp32-input MFMA at 1/16 of the bf16 rate.  An fp32 value is split EXACTLY into three bf16 pieces
// (x = hi + mid + lo, round-to-nearest at each level) and a product is accumulated from the six piece products that
// reach 2^-24 of |a||b|:  ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh); the three dropped ones are <= 2^-25.
// bf16 x bf16 products are exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32: fp32-GEMM accuracy
// (measured slightly better than the fp32 MFMA chain: fewer roundings) at 6/16 of the fp32-MFMA time.
// Operand lane map of the 32x32x16 MFMA: lane l (r = l&31, h = l>>5) holds A[r][8h+j] / B[8h+j][r], j = 0..7;
// the result layout equals the 32x32x2 one (rowmap above).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// smallest terms first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
#define FACL_SB_PA {2, 0, 1, 1, 0, 0}
#define FACL_SB_PB {0, 2, 1, 0, 1, 0}
// "bf16x3" (opt-in, precision "x3"): the two leading pieces only, products (hi,mid) (mid,hi) (hi,hi); the dropped terms
// are <= 3 * 2^-16 |a||b| per product (measured ~1e-5 relative on the GEMM results), half the MFMA work and 2/3 of the LDS
// traffic of bf16x6.  Never the default: the headline path stays fp32-grade.
#define FACL_SB3_PA {0, 1, 0}
#define FACL_SB3_PB {1, 0, 0}

__device__ __forceinline__ unsigned pk_bf16(float x0, float x1) {
    const f32x2v v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));       // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float sub_f32(float a, float b) {
    float d;
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// two values -> packed (hi, mid, lo) bf16 pairs (element 0 in the low half)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& mi, unsigned& lo) {
    hi = pk_bf16(x0, x1);
    // scalar subtractions on purpose: the SLP vectoriser would fuse each pair into v_pk_add_f32, which costs far more
    // than two v_sub_f32 beside MFMAs (MI355X guide, "packed f32 VALU ... an anti-lever beside MFMAs")
    const float r0 = sub_f32(x0, __builtin_bit_cast(float, hi << 16)), r1 = sub_f32(x1, __builtin_bit_cast(float, hi & 0xffff0000u));
    mi = pk_bf16(r0, r1);
    const float s0 = sub_f32(r0, __builtin_bit_cast(float, mi << 16)), s1 = sub_f32(r1, __builtin_bit_cast(float, mi & 0xffff0000u));
    lo = pk_bf16(s0, s1);
}
__device__ __forceinline__ bf16x8 as_bf16x8(unsigned a, unsigned b, unsigned c, unsigned d) {
    const uint4 u = make_uint4(a, b, c, d);
    return __builtin_bit_cast(bf16x8, u);
}

// ---- "fp16x3": fp32 results from THREE fp16 products ---------------------------------------------------------------------
// fp16 carries 11 significand bits: x * 2^S = h1 + h2 with h1 = fp16(x 2^S), h2 = fp16(x 2^S - h1) reproduces 22 bits
// (|x 2^S - h1 - h2| <= 2^-22 |x 2^S|; an h2 that falls into fp16's subnormal range keeps an ABSOLUTE error <= 2^-25, and
// gfx950's fp16 MFMA takes subnormal inputs at full precision -- checked on the hardware with inputs down to 2^-24), and
// a*b = h1a*h1b + (h1a*h2b + h2a*h1b) + [h2a*h2b <= 2^-22 |a||b|, dropped].  fp16 x fp16 products are exact in fp32 and
// v_mfma_f32_32x32x16_f16 accumulates in fp32: a GEMM built this way is 7e-8 from the exact product before accumulation
// rounding (bf16x6: 6e-9; the fp32 accumulation both share: 2.4e-7), i.e. fp32-GEMM accuracy at HALF the MFMA work, 2/3
// of the LDS plane traffic and ~2/3 of the split VALU of bf16x6.
// What fp16 lacks is range, so EVERY operand is pre-scaled by an exact power of two chosen from (a bound of) the operand's
// own maximum -- the power of two that puts the maximum in [2^13, 2^14) (fp16 overflows at 2^16) -- and the accumulator is
// scaled back by the exact inverse in the epilogue.  Round 4: no operand has a FIXED scale any more (rounds 1-3: activations
// 2^4, weights 2^8, i.e. |a| < 4094, |w| < 255 or NaN, and 15-18 bits for uniformly tiny tensors):
//   * weights: max|w| of the tensor (set-abstraction kernels: found by each workgroup while it splits the fragments) or of the
//     32-column tile (row-streamed GEMMs: k_rs_planes, one scale per column tile stored behind the planes);
//   * gradients: max|dy| maintained by the kernel that writes dy (64 hashed slots, rows.hip);
//   * activations: a device-side BOUND of max|a| in the same 64-slot format -- train-mode BatchNorm outputs obey
//     |gamma (y - mean) invstd + beta| <= |gamma| sqrt(n - 1) + |beta| (Samuelson's inequality; facl_bn_finalize writes it),
//     pooled features carry their exact maximum (k_sa_pool), raw inputs / eval-mode layers a measured one (facl_absmax,
//     facl_rows_act_amax).  A bound that is 2^t too large costs t of the 16 octaves below the maximum in which an element
//     keeps its 22 bits; elements below that keep an absolute error <= 2^-38 of the bound.
// Scales are clamped to [2^-40, 2^40] so that a scale, its inverse and the product of two inverses stay normal numbers: tensors
// with maxima in [2^-27, 2^53] are inside the design range, beyond it precision degrades (tiny) or the result is inf (huge).
typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2h __attribute__((ext_vector_type(2)));
#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
// smallest terms first: (lo,hi) (hi,lo) (hi,hi)
#define FACL_H3_PA {1, 0, 0}
#define FACL_H3_PB {0, 1, 0}
// max|dy| of a gradient tensor is kept as float bits in FACL_AMAX_SLOTS slots FACL_AMAX_STRIDE dwords apart (rows.hip writes,
// gemm_rs.hip reads); the buffer the C ABI calls `amax` is FACL_AMAX_SLOTS * FACL_AMAX_STRIDE uint32, zeroed by the caller
#define FACL_AMAX_SLOTS 64
#define FACL_AMAX_STRIDE 32
static_assert(FACL_AMAX_SLOTS * FACL_AMAX_STRIDE == FACL_AMAX_WORDS, "amax buffer layout");

// two ALREADY SCALED values -> packed (h1, h2) fp16 pairs (element 0 in the low half)
__device__ __forceinline__ void split_pair_h(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2v v = {x0, x1};
    const f16x2h h = __builtin_convertvector(v, f16x2h);                               // round to nearest even
    hi = __builtin_bit_cast(unsigned, h);
    // residuals x - float(h): ONE v_fma_mix_f32 each (h's half converted inside the instruction, times -1, plus x: exact, and
    // identical to v_cvt_f32_f16 + v_sub_f32, which the compiler emitted for the plain expression) -- 4 VALU per pair, not 6
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
    const f32x2v r = {r0, r1};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2h));
}
__device__ __forceinline__ f16x8h as_f16x8(unsigned a, unsigned b, unsigned c, unsigned d) {
    const uint4 u = make_uint4(a, b, c, d);
    return __builtin_bit_cast(f16x8h, u);
}

// biased exponent of the fp16x3 scale for a tensor whose maximum has the float bits `b` (>= 0): 2^(13 - floor(log2 max)),
// clamped to [2^-40, 2^40] (max = 0 / denormal -> 2^40; inf / NaN -> 2^-40, and the NaN propagates through the products).
// h3_se_wide: the clamp for GRADIENT-like operands (dy, G3), [2^-80, 2^80] -- losses and their gradients span far more
// octaves than weights and normalised activations do.  One operand of a product may be wide: 381 - wide - narrow stays in
// [7, 247], a normal number.
__device__ __forceinline__ int h3_se(unsigned b) {
    int se = 267 - (int)((b >> 23) & 0xff);
    se = se > 167 ? 167 : se;
    return se < 87 ? 87 : se;
}
__device__ __forceinline__ int h3_se_wide(unsigned b) {
    int se = 267 - (int)((b >> 23) & 0xff);
    se = se > 207 ? 207 : se;
    return se < 47 ? 47 : se;
}
__device__ __forceinline__ float pow2_biased(int e) { return __uint_as_float((unsigned)e << 23); }
// maximum of a non-negative value over the wave on the DPP crossbar (row_shr 1,2,4,8, row_bcast 15 / 31; no LDS round trip),
// uniform result
template <int CTRL, int RMASK>
__device__ __forceinline__ float facl_dpp_max_step(float v) {
    const int o = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, false);
    return fmaxf(v, __builtin_bit_cast(float, o));
}
__device__ __forceinline__ float facl_wave_max_nonneg(float v) {
    v = facl_dpp_max_step<0x111, 0xf>(v); v = facl_dpp_max_step<0x112, 0xf>(v); v = facl_dpp_max_step<0x114, 0xf>(v);
    v = facl_dpp_max_step<0x118, 0xf>(v); v = facl_dpp_max_step<0x142, 0xa>(v); v = facl_dpp_max_step<0x143, 0xc>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// 1 / (2^(seA-127) 2^(seB-127)): biased exponent 381 - seA - seB (two narrow scales: [47, 207]; one wide: [7, 247])
__device__ __forceinline__ float h3_unscale(int seA, int seB) { return pow2_biased(381 - seA - seB); }
// wave-uniform maximum over the FACL_AMAX_SLOTS slots of an amax buffer (rows.hip: abs_max_slot writes them)
__device__ __forceinline__ unsigned amax_bits(const unsigned* amax) {
    unsigned b = amax[(threadIdx.x & (FACL_AMAX_SLOTS - 1)) * FACL_AMAX_STRIDE];    // one slot per lane
#pragma unroll
    for (int o = 32; o; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)b, o, 64); b = b > t ? b : t; }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)b);
}
__device__ __forceinline__ int h3_se_of(const unsigned* amax) { return h3_se(amax_bits(amax)); }
__device__ __forceinline__ int h3_se_wide_of(const unsigned* amax) { return h3_se_wide(amax_bits(amax)); }

// max|w| over `n` floats, taken by the whole workgroup (all threads must call; two barriers) -> biased exponent of the scale.
// `red`: >= 16 floats of LDS scratch that nobody else touches during the call.
__device__ __forceinline__ int wg_h3_se(const float* __restrict__ w, int n, float* red) {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[0];
    for (int w2 = 1; w2 < (int)(blockDim.x >> 6); ++w2) m = fmaxf(m, red[w2]);
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(h3_se(__float_as_uint(m)));
}
