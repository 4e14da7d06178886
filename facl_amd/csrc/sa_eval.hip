// Set-abstraction point-MLP, EVAL mode, as ONE kernel (the feature-extraction path,
// /root/reference/training_code/extract_motion_feature.py:143-221 runs the encoder under eval(): every BatchNorm is a constant
// per-channel affine, so nothing has to be known about the whole batch before a position can be finished):
//     x (12-16 B per position) -> a1 = relu(bn1(W1 x + b1))          VALU, folded table (facl_sa_l1tab)
//                              -> y2 = a1 W2^T + b2                   48 MFMAs per unit (fp16x3)
//                              -> a2 = relu(bn2(y2))                  registers
//                              -> y3 = a2 W3^T + b3                  192 MFMAs per unit (fp16x3)
//                              -> pooled = relu(bn3(max_k y3))        256 floats per unit
// One wave owns one unit (64 positions = one group) at a time; nothing but the pooled (groups, 256) features is stored: the
// training passes it replaces in the extraction entries (facl_sa_fwd2 + facl_sa_fwd3 + facl_sa_pool with folded constants)
// wrote the 16 KiB y2 tile of every unit and read it back, and kept statistics / argmax nobody asked for.
// fp16x3 operand scales (common.h): W2 / W3 by the power of two of their own maxima (taken by the workgroup), a1 by the one
// of its bound from max|x| (facl_sa_l1tab), a2 by the power of two of the UNIT's own maximum (one wave-wide reduction in
// registers: eval-mode constants give no bound, and a unit's output depends on nothing but that unit).
// A NaN / inf coordinate poisons its group's 256 outputs (MaxPool2d propagates NaN; the max below would skip it).
// Roofline: MFMA fp16 (2.5 PFLOP/s dense); 3 * 2 * 64 * (64 + 256) * 64 executed FLOP per unit; 64*D*4 B in, 1 KiB out.
#include "common.h"
#include <stdlib.h>

namespace {

template <int CTRL, int RMASK>
__device__ __forceinline__ float dpp_max_step_e(float v) {
    const int o = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, false);
    return fmaxf(v, __builtin_bit_cast(float, o));
}
// maximum of a non-negative value over the wave (row_shr 1/2/4/8, row_bcast 15/31), broadcast from lane 63
__device__ __forceinline__ float wave_max_nonneg_e(float v) {
    v = dpp_max_step_e<0x111, 0xf>(v); v = dpp_max_step_e<0x112, 0xf>(v); v = dpp_max_step_e<0x114, 0xf>(v); v = dpp_max_step_e<0x118, 0xf>(v);
    v = dpp_max_step_e<0x142, 0xa>(v); v = dpp_max_step_e<0x143, 0xc>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

constexpr int EV_LDS_BYTES = (4096 + 1024) * 16 + (128 + 16 + 16 + 16) * 16 + 3 * 256 * 4 + 64;

template <int D, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_sa_eval(const float* __restrict__ x, int nunits, const float* __restrict__ l1tab_g,
                                                 const float* __restrict__ W2, const float* __restrict__ b2,
                                                 const float* __restrict__ sc2, const float* __restrict__ sh2,
                                                 const float* __restrict__ W3, const float* __restrict__ b3,
                                                 const float* __restrict__ sc3, const float* __restrict__ sh3,
                                                 float* __restrict__ pooled, const unsigned* __restrict__ a1amax) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    uint4* w3p = reinterpret_cast<uint4*>(lds4);          // B fragments of sgn3*W3: [(ct3*4 + kk)*2 + plane][lane]   64 KiB
    uint4* w2p = w3p + 4096;                              // A fragments of W2:      [(rt*4 + kk)*2 + plane][lane]    16 KiB
    float4* l1tab = lds4 + 5120;                          // folded layer 1 (x a1's scale): [c][w0 w1 w2 w3 | b 0 0 0]
    float4* b2s = l1tab + 128;                            // b2 x (a1 scale)(W2 scale)
    float4* sc2s = b2s + 16;
    float4* sh2s = sc2s + 16;
    float* b3s = reinterpret_cast<float*>(sh2s + 16);     // sgn3 * b3
    float* sc3s = b3s + 256;                              // |scale3|
    float* sh3s = sc3s + 256;
    float* red = sh3s + 256;

    // ---- operand scales of the weights: one pass over W3 with the values kept in registers, W2 by the helper
    constexpr int NIT = (2048 + 64 * WAVES - 1) / (64 * WAVES);   // W3 fragments per thread
    float4 wv[NIT][2];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + 64 * WAVES * it;
        const int ln = i & 63, kk = (i >> 6) & 3, ct3 = (i >> 8) & 7;
        const float* wrow = W3 + (32 * ct3 + (ln & 31)) * 64 + 16 * kk + 4 * (ln >> 5);
        wv[it][0] = *reinterpret_cast<const float4*>(wrow);
        wv[it][1] = *reinterpret_cast<const float4*>(wrow + 8);
    }
    float m3 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < 2; ++e)
            m3 = fmaxf(fmaxf(m3, fmaxf(fabsf(wv[it][e].x), fabsf(wv[it][e].y))), fmaxf(fabsf(wv[it][e].z), fabsf(wv[it][e].w)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m3 = fmaxf(m3, __shfl_xor(m3, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m3;
    __syncthreads();
    m3 = red[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) m3 = fmaxf(m3, red[w]);
    __syncthreads();
    const int seW3 = __builtin_amdgcn_readfirstlane(h3_se(__float_as_uint(m3)));
    const int seW2 = wg_h3_se(W2, 64 * 64, red);
    const int seA1 = h3_se_of(a1amax);
    const float sW3 = pow2_biased(seW3), sW2 = pow2_biased(seW2), sA1 = pow2_biased(seA1);
    const float sAW = sA1 * sW2, unsAW = h3_unscale(seA1, seW2);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + 64 * WAVES * it;
        if (i >= 2048) break;
        const int ln = i & 63, kk = (i >> 6) & 3, ct3 = i >> 8;
        const float sw = sgn_of(sc3[32 * ct3 + (ln & 31)]) * sW3;       // sign(gamma3) = sign(scale3): max(sgn*y3) serves BN3 + ReLU
        const float4 w0 = wv[it][0], w1 = wv[it][1];
        unsigned hi[4], lo[4];
        split_pair_h(w0.x * sw, w0.y * sw, hi[0], lo[0]);
        split_pair_h(w0.z * sw, w0.w * sw, hi[1], lo[1]);
        split_pair_h(w1.x * sw, w1.y * sw, hi[2], lo[2]);
        split_pair_h(w1.z * sw, w1.w * sw, hi[3], lo[3]);
        uint4* d = w3p + ((ct3 * 4 + kk) * 2) * 64 + ln;
        d[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        d[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
    if (threadIdx.x < 512) {
        const int i = threadIdx.x;                                       // 512 W2 fragments
        const int ln = i & 63, kk = (i >> 6) & 3, rt = i >> 8;
        const float* wrow = W2 + (32 * rt + (ln & 31)) * 64 + 16 * kk + 8 * (ln >> 5);
        const float4 w0 = *reinterpret_cast<const float4*>(wrow), w1 = *reinterpret_cast<const float4*>(wrow + 4);
        unsigned hi[4], lo[4];
        split_pair_h(w0.x * sW2, w0.y * sW2, hi[0], lo[0]);
        split_pair_h(w0.z * sW2, w0.w * sW2, hi[1], lo[1]);
        split_pair_h(w1.x * sW2, w1.y * sW2, hi[2], lo[2]);
        split_pair_h(w1.z * sW2, w1.w * sW2, hi[3], lo[3]);
        uint4* d = w2p + ((rt * 4 + kk) * 2) * 64 + ln;
        d[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        d[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
    if (threadIdx.x < 128) {
        float4 t = reinterpret_cast<const float4*>(l1tab_g)[threadIdx.x];
        t.x *= sA1; t.y *= sA1; t.z *= sA1; t.w *= sA1;                  // relu(s w.x + s b) = s relu(w.x + b): exact for a power of two
        l1tab[threadIdx.x] = t;
    }
    if (threadIdx.x < 16) {
        float4 bb = reinterpret_cast<const float4*>(b2)[threadIdx.x];
        bb.x *= sAW; bb.y *= sAW; bb.z *= sAW; bb.w *= sAW;              // the layer-2 accumulators ride at (a1 scale)(W2 scale)
        b2s[threadIdx.x] = bb;
        sc2s[threadIdx.x] = reinterpret_cast<const float4*>(sc2)[threadIdx.x];
        sh2s[threadIdx.x] = reinterpret_cast<const float4*>(sh2)[threadIdx.x];
    }
    if (threadIdx.x < 256) {
        const float s3 = sc3[threadIdx.x];
        b3s[threadIdx.x] = b3[threadIdx.x] * sgn_of(s3);
        sc3s[threadIdx.x] = fabsf(s3);
        sh3s[threadIdx.x] = sh3[threadIdx.x];
    }
    __syncthreads();

    const int lane = lane_id(), h = lane >> 5, q = lane & 31;
    const int wave_g = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6)), nwaves = gridDim.x * WAVES;
    auto load_x = [&](int u, float (&xv)[2][4]) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const size_t p = (size_t)u * 64 + 32 * ct + q;
            if (D == 4) {
                const float4 t = *reinterpret_cast<const float4*>(x + p * 4);
                xv[ct][0] = t.x; xv[ct][1] = t.y; xv[ct][2] = t.z; xv[ct][3] = t.w;
            } else {
                xv[ct][0] = x[p * 3]; xv[ct][1] = x[p * 3 + 1]; xv[ct][2] = x[p * 3 + 2]; xv[ct][3] = 0.f;
            }
        }
    };
    auto upper = [&](unsigned v) { return __builtin_amdgcn_permlane32_swap(v, v, false, false)[1]; };
    constexpr int HA[3] = FACL_H3_PA, HB[3] = FACL_H3_PB;
    float xn[2][4];
    if (wave_g < nunits) load_x(wave_g, xn);
    for (int u = wave_g; u < nunits; u += nwaves) {
        asm volatile("" ::: "memory");       // LDS tables are re-read per unit instead of living in registers
        float xv[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[ct][i] = xn[ct][i];
        load_x(u + nwaves < nunits ? u + nwaves : u, xn);   // unconditional prefetch (see k_sa_fwd3_sb)
        // a NaN / inf coordinate anywhere in the group -> every output of the group is NaN (x - x is +0 for finite x)
        float chk = (xv[0][0] - xv[0][0]) + (xv[0][1] - xv[0][1]) + (xv[0][2] - xv[0][2]) + (xv[0][3] - xv[0][3]);
        chk += (xv[1][0] - xv[1][0]) + (xv[1][1] - xv[1][1]) + (xv[1][2] - xv[1][2]) + (xv[1][3] - xv[1][3]);
        const float poison = __builtin_amdgcn_ballot_w64(chk != chk) ? __uint_as_float(0x7fc00000u) : 0.f;

        // ---- layer 1 (VALU) -> fp16 planes of a1 * sA1 for this lane's two positions, k-slots in the order layer 2's B operand wants
        f16x8h ap[2][4][2];                  // [position tile][k16 block][plane]
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float a1[2][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 16 * kk + 8 * h + j;
                const float4 w = l1tab[c * 2];
                const float b = l1tab[c * 2 + 1].x;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    float v = fmaf(w.x, xv[ct][0], b);
                    v = fmaf(w.y, xv[ct][1], v);
                    v = fmaf(w.z, xv[ct][2], v);
                    if (D == 4) v = fmaf(w.w, xv[ct][3], v);
                    a1[ct][j] = fmaxf(v, 0.f);
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                unsigned hi[4], lo[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) split_pair_h(a1[ct][2 * t], a1[ct][2 * t + 1], hi[t], lo[t]);
                ap[ct][kk][0] = as_f16x8(hi[0], hi[1], hi[2], hi[3]);
                ap[ct][kk][1] = as_f16x8(lo[0], lo[1], lo[2], lo[3]);
            }
        }
        // ---- layer 2 on the MFMA, transposed orientation (lane = position, register = channel): y2 = a1 W2^T + b2
        f32x16 y2[2][2];                     // [rt][ct]
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 bb = b2s[8 * rt + 2 * r4 + h];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    y2[rt][ct][4 * r4 + 0] = bb.x; y2[rt][ct][4 * r4 + 1] = bb.y;
                    y2[rt][ct][4 * r4 + 2] = bb.z; y2[rt][ct][4 * r4 + 3] = bb.w;
                }
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f16x8h wf[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) wf[rt][pl] = __builtin_bit_cast(f16x8h, w2p[((rt * 4 + kk) * 2 + pl) * 64 + lane]);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) y2[rt][ct] = MFMA_F16(wf[rt][HA[t]], ap[ct][kk][HB[t]], y2[rt][ct]);
        }
        // ---- a2 = relu(bn2(y2)) in place, the unit's maximum, its power-of-two scale
        float mx = 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 sc = sc2s[8 * rt + 2 * r4 + h], sh = sh2s[8 * rt + 2 * r4 + h];
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = fmaxf(fmaf(scv[e], y2[rt][ct][4 * r4 + e] * unsAW, shv[e]), 0.f);
                        y2[rt][ct][4 * r4 + e] = a;
                        mx = fmaxf(mx, a);
                    }
            }
        mx = wave_max_nonneg_e(mx);
        const int seA2 = h3_se(__float_as_uint(mx));
        const float sA2 = pow2_biased(seA2), UNS = h3_unscale(seA2, seW3);
        // fp16 planes of a2 * sA2: the fragment registers 8m .. 8m+7 of (rt, ct) are the 8 k-slots of k-block kk = 2 rt + m
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int rt = kk >> 1, m = kk & 1;
                unsigned hi[4], lo[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    split_pair_h(y2[rt][ct][8 * m + 2 * t] * sA2, y2[rt][ct][8 * m + 2 * t + 1] * sA2, hi[t], lo[t]);
                ap[ct][kk][0] = as_f16x8(hi[0], hi[1], hi[2], hi[3]);
                ap[ct][kk][1] = as_f16x8(lo[0], lo[1], lo[2], lo[3]);
            }
        // ---- layer 3: eight column tiles of 32 channels; per tile 24 MFMAs, then the maximum over the unit's 64 positions
        float* orow = pooled + (size_t)u * 256;
#pragma unroll 1
        for (int ct3 = 0; ct3 < 8; ++ct3) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f16x8h bfr[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) bfr[p] = __builtin_bit_cast(f16x8h, w3p[((ct3 * 4 + kk) * 2 + p) * 64 + lane]);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc0 = MFMA_F16(ap[0][kk][HA[t]], bfr[HB[t]], acc0);
                    acc1 = MFMA_F16(ap[1][kk][HA[t]], bfr[HB[t]], acc1);
                }
            }
            float best = acc0[0];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                best = fmaxf(fmaxf(best, acc0[r]), acc0[r + 1]);
                best = fmaxf(fmaxf(best, acc1[r]), acc1[r + 1]);
            }
            best = fmaxf(best, __uint_as_float(upper(__float_as_uint(best))));     // the other lane half holds the other 32 positions
            if (h == 0) {
                const int c = 32 * ct3 + q;
                const float ym = fmaf(best, UNS, b3s[c]);                            // max_k sgn3 * y3
                orow[c] = relu_nan(fmaf(sc3s[c], ym, sh3s[c])) + poison;
            }
        }
    }
}

}  // namespace

// pooled (nunits, 256) = net3DV_1 in eval mode (cn3d_model_conbag.py:43-58 under model.eval()).  x (nunits*64, D) grouped rows;
// l1tab (64,8) from facl_sa_l1tab with the eval-mode constants of BN1; scale2 / shift2 (64) and scale3 / shift3 (256): BN2 / BN3
// folded (facl_bn_eval_consts rows 2 and 3); a1amax: the bound of max|a1| facl_sa_l1tab derived from max|x| (facl_absmax).
extern "C" int facl_sa_eval(const float* x, int64_t nunits, int D, const float* l1tab, const float* W2, const float* b2,
                            const float* scale2, const float* shift2, const float* W3, const float* b3, const float* scale3,
                            const float* shift3, float* pooled, const uint32_t* a1amax, void* stream) {
    if (!x || !l1tab || !W2 || !b2 || !scale2 || !shift2 || !W3 || !b3 || !scale3 || !shift3 || !pooled || !a1amax) return FACL_E_NULL;
    if ((D != 3 && D != 4) || nunits < 1 || nunits > 0x7fffffff) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    // 12 waves per workgroup = three per SIMD (the kernel needs 145 registers; one workgroup per CU: 86 KiB of LDS):
    // 0.304 ms vs 0.3135 ms with 8 waves at the headline shape, same box, alternating runs (FACL_EVAL_WAVES=8 for the A/B)
    static const int waves = getenv("FACL_EVAL_WAVES") ? atoi(getenv("FACL_EVAL_WAVES")) : 12;
    const int W = waves == 8 ? 8 : 12;
    const int grid = (int)(nunits < 256 * W ? (nunits + W - 1) / W : 256);
    static bool attr_done_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done_dev[dev]) {
        const void* fns[4] = {(const void*)k_sa_eval<3, 8>, (const void*)k_sa_eval<4, 8>, (const void*)k_sa_eval<3, 12>, (const void*)k_sa_eval<4, 12>};
        for (int i = 0; i < 4; ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, EV_LDS_BYTES);
            if (e != hipSuccess) return (int)e;
        }
        attr_done_dev[dev] = true;
    }
#define FACL_EVAL_LAUNCH(DD, WW) hipLaunchKernelGGL((k_sa_eval<DD, WW>), dim3(grid), dim3(64 * WW), EV_LDS_BYTES, st, x, (int)nunits, l1tab, W2, \
                                                    b2, scale2, shift2, W3, b3, scale3, shift3, pooled, a1amax)
    if (D == 4 && W == 12) FACL_EVAL_LAUNCH(4, 12);
    else if (D == 4) FACL_EVAL_LAUNCH(4, 8);
    else if (W == 12) FACL_EVAL_LAUNCH(3, 12);
    else FACL_EVAL_LAUNCH(3, 8);
#undef FACL_EVAL_LAUNCH
    return facl_launch_status();
}
