// Set-abstraction point-MLP backward, pass 2 (layer 2 + layer 1 of net3DV_1, autograd of cn3d_model_conbag.py:43-52)
// on the bf16 MFMA ("bf16x6", common.h): per position   da1 = dy2 W2,  dz1 = da1 * [z1 > 0],  dW2 += dy2^T a1,
// R1 += [x | 1]^T dz1  -- the two 64x64 contractions run as exact 3-way bf16 splits on v_mfma_f32_32x32x16_bf16
// (192 MFMAs of 32 cycles per 64 positions instead of 256 + 64 fp32-input ones of 64 / 32 cycles in k_sa_bwd2,
// sa_bwd.hip, which stays selectable with FACL_BWD2_F32=1).
//
// Dataflow per half unit (one wave, 32 positions), chosen so that only ONE tile ever needs a transpose, and that one
// happens on bf16 planes through the hardware transposing LDS read (ds_read_b64_tr_b16):
//   1. dy2 (fragment layout: lane = position, 4 consecutive channels per float4) is split into 3 bf16 planes in
//      registers; two float4s are the 8 k-slots of a 32x32x16 A operand [row p][k = c2] (as in k_sa_fwd3_sb).
//      The same packed pairs go to a per-wave LDS image [plane][c2 tile][p][32 ch] (64-B rows, 16-B chunks
//      XOR-swizzled by (p >> 1) & 3: 2-way on the b64 stores, conflict-free on the transposed reads).
//   2. da1 = dy2 W2: A = those registers, B = pre-split W2 fragments [k = c2][col c1] (24 KiB LDS, once per
//      workgroup).  The result tiles have lane = c1 (column), registers = positions (rows).
//   3. In THAT layout a1 = relu(w1'[c1] . x[p] + b1'[c1]) needs the lane's own layer-1 row (loop-invariant registers)
//      and x[p] as an LDS broadcast read; dz1 = da1 * [z1 > 0]; R1 += [x | 1]^T dz1 is 4 VALU FMAs per element into
//      per-lane accumulators (exact fp32, no MFMA, no transpose of dz1).
//   4. dW2 += dy2^T a1 sums over positions = the ROW index of the a1 tiles: registers 8s..8s+7 of a tile, split
//      into bf16 planes, ARE the B operand of k-step s (k order 16s + 8(j>>2) + 4h + (j&3): cdna guide, "An
//      accumulator tile as the next MFMA's operand"); the A operand dy2^T [row c2][k = p] comes from the LDS image
//      through two ds_read_b64_tr_b16 per plane, which deliver exactly that k order.
// Built with -fno-slp-vectorize (facl_amd/build.py): packed f32 VALU beside MFMAs is an anti-lever on gfx950.
// Output row per wave: [ dW2 (64 x 64, [c2][c1]) | R1 (8 x 64, rows x_0..x_{D-1}, 1, 0..) ] doubles.
// Roofline: MFMA bf16 (2.5 PFLOP/s dense; 6 executed FLOPs per algorithmic one); 2 workgroups of 4 waves per CU.
#include "common.h"

namespace {

constexpr int B2_V = 64 * 64 + 8 * 64;
constexpr int B2S_PLANE = 2 * 32 * 64;          // bytes of one plane of a wave's half-unit image: [c2 tile][32 p][64 B]
constexpr int B2S_IMG = 3 * B2S_PLANE;          // 12 KiB per wave
constexpr int B2S_WAVES = 4;

#define WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// maximum of a non-negative value over the wave on the DPP crossbar (row_shr 1,2,4,8, row_bcast 15 / 31; no LDS round trip)
template <int CTRL, int RMASK>
__device__ __forceinline__ float dpp_max_step(float v) {
    const int o = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, false);
    return fmaxf(v, __builtin_bit_cast(float, o));
}
__device__ __forceinline__ float wave_max_nonneg(float v) {
    v = dpp_max_step<0x111, 0xf>(v); v = dpp_max_step<0x112, 0xf>(v); v = dpp_max_step<0x114, 0xf>(v); v = dpp_max_step<0x118, 0xf>(v);
    v = dpp_max_step<0x142, 0xa>(v); v = dpp_max_step<0x143, 0xc>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ bf16x8 tr_read_pair(const char* p_lo, const char* p_hi) {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p_lo));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p_hi));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// H3: fp16x3 arithmetic (common.h).  W2 by the power of two of its own maximum, a1 by the one of its bound; dy2 -- a gradient, and computed here -- by the power of
// two that puts the maximum of the wave's half unit in [2^13, 2^14) (one DPP reduction per half unit, no LDS, no host).  The
// da1 tiles are scaled back when they are consumed; dW2's contribution of a half unit is accumulated in a temporary tile
// and added to the running sums with the inverse scale, so half units of different magnitude mix exactly.
// PREF (round 4 experiment, FACL_BWD2_PREF=1): the NEXT half unit's 16 loads are issued as soon as this half unit's dz2 / y2
// registers are consumed (step 1), so they fly under the MFMA / layer-1 / dW2 work instead of being waited for at the top.
template <int D, bool H3, bool PREF = false>
__global__ __launch_bounds__(64 * B2S_WAVES, 2) void k_sa_bwd2_sb(
    const float* __restrict__ dz2f, const float* __restrict__ y2f, const float* __restrict__ x, int nunits,
    const float* __restrict__ bw2 /* (4,64): scale2, A, B, mean2 */, const float* __restrict__ W2,
    const float* __restrict__ l1tab_g, double* __restrict__ part, int rev, const unsigned* __restrict__ a1amax) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    uint4* w2p = reinterpret_cast<uint4*>(lds4);            // [(ct1*4 + kk)*3 + plane][lane]: 1536 uint4 = 24 KiB
    float4* tab = lds4 + 1536;                              // 4 x 16 float4
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // fp16x3 operand scales (common.h): W2 by the power of two of its own maximum (taken here; `tab` is filled further down),
    // a1 by the one of the bound its forward used
    int seW = 127, seA1 = 127;
    if (H3) { seW = wg_h3_se(W2, 64 * 64, reinterpret_cast<float*>(tab)); seA1 = h3_se_of(a1amax); }
    const float sW2 = pow2_biased(seW), sA1 = pow2_biased(seA1);
    float4* xs4 = tab + 64 + wave * 64;                     // per wave: x of the unit's 64 positions
    char* img = reinterpret_cast<char*>(tab + 64 + B2S_WAVES * 64) + wave * B2S_IMG;
    for (int i = threadIdx.x; i < 512; i += 64 * B2S_WAVES) {
        const int ln = i & 63, kk = (i >> 6) & 3, ct1 = i >> 8;
        const int c1 = 32 * ct1 + (ln & 31), k0 = 16 * kk + 4 * (ln >> 5);
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = W2[(k0 + j) * 64 + c1]; v[4 + j] = W2[(k0 + 8 + j) * 64 + c1]; }
        unsigned hi[4], mi[4], lo[4];
        if (H3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { split_pair_h(v[2 * j] * sW2, v[2 * j + 1] * sW2, hi[j], mi[j]); lo[j] = 0u; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], hi[j], mi[j], lo[j]);
        }
        uint4* d = w2p + ((ct1 * 4 + kk) * 3) * 64 + ln;
        d[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        d[64] = make_uint4(mi[0], mi[1], mi[2], mi[3]);
        d[128] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
    if (threadIdx.x < 64) tab[threadIdx.x] = reinterpret_cast<const float4*>(bw2)[threadIdx.x];
    __syncthreads();
    const float4* sc2 = tab; const float4* cA = tab + 16; const float4* cB = tab + 32; const float4* mean2 = tab + 48;

    const int lane = lane_id(), h = lane >> 5, q = lane & 31;
    const int wave_g = blockIdx.x * B2S_WAVES + wave, nwaves = gridDim.x * B2S_WAVES;
    // this lane's two layer-1 rows (c1 = q and 32 + q): loop-invariant
    float4 w1r[2];
    float b1r[2];
#pragma unroll
    for (int ct1 = 0; ct1 < 2; ++ct1) {
        w1r[ct1] = reinterpret_cast<const float4*>(l1tab_g)[(32 * ct1 + q) * 2];
        b1r[ct1] = l1tab_g[(32 * ct1 + q) * 8 + 4];
    }
    // transposed-read addresses inside a (plane, c2 tile) sub-image: lane (16-lane group g = lane >> 4, i = lane & 15)
    // supplies row 4*(g>>1) + (i>>2), 16-B chunk 2*(g&1) + ((i&3)>>1), 8-B half (i&1); the rows of a fragment's second
    // half are 8 further, a k-step 16 further (neither changes the swizzle key (row >> 1) & 3 above bit 1)
    const char* tr_ptr[2];
    {
        const int tr_row = 4 * (lane >> 5) + ((lane & 15) >> 2);
        const int tr_chunk = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = tr_row + 8 * t;
            tr_ptr[t] = img + row * 64 + ((tr_chunk ^ ((row >> 1) & 3)) << 4) + 8 * (lane & 1);
        }
    }
    // image store address of this lane's position row q (chunk r4 ^ key, 8-B half h)
    char* st_row = img + q * 64 + 8 * h;
    const int st_key = (q >> 1) & 3;

    f32x16 dw2[2][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) dw2[0][0][r] = dw2[0][1][r] = dw2[1][0][r] = dw2[1][1][r] = 0.f;
    float r1[2][5];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int d = 0; d < 5; ++d) r1[a][d] = 0.f;
    constexpr int PA[6] = FACL_SB_PA, PB[6] = FACL_SB_PB;

    // rev: walk the units from the LAST one down.  The pass before this one (k_sa_bwd1) wrote dz2 and read y2 front to back, so
    // the ends of both arrays are what the memory-side cache (256 MB) still holds when this kernel starts.
    float4 zv[8], yv[8];                                    // [rt*4 + r4] of the current half unit
    auto load_half = [&](int u, int ct) {
        const float* zt = dz2f + (size_t)u * FACL_UNIT_ELEMS + ct * 2048;
        const float* yt = y2f + (size_t)u * FACL_UNIT_ELEMS + ct * 2048;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            zv[i] = *reinterpret_cast<const float4*>(zt + (i * 64 + lane) * 4);
            yv[i] = *reinterpret_cast<const float4*>(yt + (i * 64 + lane) * 4);
        }
    };
    if (PREF && wave_g < nunits) load_half(rev ? nunits - 1 - wave_g : wave_g, 0);
    for (int uu = wave_g; uu < nunits; uu += nwaves) {
        const int u = rev ? nunits - 1 - uu : uu;
        {   // x of the unit -> LDS (one float4 per position)
            const size_t p = (size_t)u * 64 + lane;
            float4 xv;
            if (D == 4) xv = *reinterpret_cast<const float4*>(x + p * 4);
            else xv = make_float4(x[p * 3], x[p * 3 + 1], x[p * 3 + 2], 0.f);
            xs4[lane] = xv;
        }
#pragma unroll 1
        for (int ct = 0; ct < 2; ++ct) {
            if (!PREF) load_half(u, ct);                    // all 16 loads in flight before the first use
            // ---- 1. dy2 -> 16-bit planes: registers (A operand of da1) + LDS image (transposed A operand of dW2)
            bf16x8 ap[4][3];                                // [k16 block][plane]
            float uns_da = 1.f, uns_dw = 1.f;               // H3: inverse scales of this half unit's da1 / dW2 tiles
            if (H3) {
                float dyv[32];
                float mx = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int rt = kk >> 1, m = kk & 1;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int r4 = 2 * m + t;
                        const float4 z = zv[rt * 4 + r4], y = yv[rt * 4 + r4];
                        const int ti = 8 * rt + 2 * r4 + h;
                        const float4 s = sc2[ti], a = cA[ti], b = cB[ti], mm = mean2[ti];
                        float* d = &dyv[8 * kk + 4 * t];
                        d[0] = fmaf(s.x, z.x, fmaf(b.x, y.x - mm.x, a.x));
                        d[1] = fmaf(s.y, z.y, fmaf(b.y, y.y - mm.y, a.y));
                        d[2] = fmaf(s.z, z.z, fmaf(b.z, y.z - mm.z, a.z));
                        d[3] = fmaf(s.w, z.w, fmaf(b.w, y.w - mm.w, a.w));
                        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
                    }
                }
                mx = wave_max_nonneg(mx);
                const int se = h3_se_wide(__float_as_uint(mx));                 // 2^(13 - floor(log2 max)); NaN / inf pass through as such
                const float sD = pow2_biased(se);
                uns_da = h3_unscale(se, seW);                                   // 1 / (sD sW2)
                uns_dw = h3_unscale(se, seA1);                                  // 1 / (sD sA1)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int rt = kk >> 1, m = kk & 1;
                    unsigned hi[4], lo[4];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int r4 = 2 * m + t;
                        const float* d = &dyv[8 * kk + 4 * t];
                        split_pair_h(d[0] * sD, d[1] * sD, hi[2 * t], lo[2 * t]);
                        split_pair_h(d[2] * sD, d[3] * sD, hi[2 * t + 1], lo[2 * t + 1]);
                        char* dst = st_row + rt * (32 * 64) + ((r4 ^ st_key) << 4);
                        *reinterpret_cast<uint2*>(dst) = make_uint2(hi[2 * t], hi[2 * t + 1]);
                        *reinterpret_cast<uint2*>(dst + B2S_PLANE) = make_uint2(lo[2 * t], lo[2 * t + 1]);
                    }
                    ap[kk][0] = as_bf16x8(hi[0], hi[1], hi[2], hi[3]);
                    ap[kk][1] = as_bf16x8(lo[0], lo[1], lo[2], lo[3]);
                }
            } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int rt = kk >> 1, m = kk & 1;
                unsigned hi[4], mi[4], lo[4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int r4 = 2 * m + t;
                    const float4 z = zv[rt * 4 + r4], y = yv[rt * 4 + r4];
                    const int ti = 8 * rt + 2 * r4 + h;
                    const float4 s = sc2[ti], a = cA[ti], b = cB[ti], mm = mean2[ti];
                    const float d0 = fmaf(s.x, z.x, fmaf(b.x, y.x - mm.x, a.x));
                    const float d1 = fmaf(s.y, z.y, fmaf(b.y, y.y - mm.y, a.y));
                    const float d2 = fmaf(s.z, z.z, fmaf(b.z, y.z - mm.z, a.z));
                    const float d3 = fmaf(s.w, z.w, fmaf(b.w, y.w - mm.w, a.w));
                    split_pair(d0, d1, hi[2 * t], mi[2 * t], lo[2 * t]);
                    split_pair(d2, d3, hi[2 * t + 1], mi[2 * t + 1], lo[2 * t + 1]);
                    char* dst = st_row + rt * (32 * 64) + ((r4 ^ st_key) << 4);
                    *reinterpret_cast<uint2*>(dst) = make_uint2(hi[2 * t], hi[2 * t + 1]);
                    *reinterpret_cast<uint2*>(dst + B2S_PLANE) = make_uint2(mi[2 * t], mi[2 * t + 1]);
                    *reinterpret_cast<uint2*>(dst + 2 * B2S_PLANE) = make_uint2(lo[2 * t], lo[2 * t + 1]);
                }
                ap[kk][0] = as_bf16x8(hi[0], hi[1], hi[2], hi[3]);
                ap[kk][1] = as_bf16x8(mi[0], mi[1], mi[2], mi[3]);
                ap[kk][2] = as_bf16x8(lo[0], lo[1], lo[2], lo[3]);
            }
            }
            if (PREF) {                                     // zv / yv are consumed: the next half unit's loads take their place
                const int un = uu + nwaves < nunits ? (rev ? nunits - 1 - (uu + nwaves) : uu + nwaves) : u;   // last round: re-reads its own (L2 hit)
                if (ct == 0) load_half(u, 1); else load_half(un, 0);
            }
            // ---- 2. da1[p][c1] = sum_c2 dy2[p][c2] W2[c2][c1]   (tiles: rows = positions, lane = c1)
            f32x16 da1[2];                                  // [c1 tile]
#pragma unroll
            for (int r = 0; r < 16; ++r) da1[0][r] = da1[1][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ct1 = 0; ct1 < 2; ++ct1) {
                    bf16x8 bfr[3];
#pragma unroll
                    for (int p = 0; p < (H3 ? 2 : 3); ++p) bfr[p] = __builtin_bit_cast(bf16x8, w2p[((ct1 * 4 + kk) * 3 + p) * 64 + lane]);
                    if (H3) {
                        constexpr int HA[3] = FACL_H3_PA, HB[3] = FACL_H3_PB;
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            da1[ct1] = MFMA_F16(__builtin_bit_cast(f16x8h, ap[kk][HA[t]]), __builtin_bit_cast(f16x8h, bfr[HB[t]]), da1[ct1]);
                    } else {
#pragma unroll
                        for (int t = 0; t < 6; ++t) da1[ct1] = MFMA_BF16(ap[kk][PA[t]], bfr[PB[t]], da1[ct1]);
                    }
                }
            WAVE_LDS_FENCE();                               // xs4 / image written by this wave are visible to its reads
            // ---- 3. a1, dz1, R1 in the (rows = positions, lane = c1) layout
            float a1v[2][16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 xv = xs4[32 * ct + rowmap(r, h)];              // 2 addresses per instruction: broadcast
#pragma unroll
                for (int ct1 = 0; ct1 < 2; ++ct1) {
                    float v = fmaf(w1r[ct1].x, xv.x, b1r[ct1]);
                    v = fmaf(w1r[ct1].y, xv.y, v);
                    v = fmaf(w1r[ct1].z, xv.z, v);
                    if (D == 4) v = fmaf(w1r[ct1].w, xv.w, v);
                    a1v[ct1][r] = fmaxf(v, 0.f);
                    const float dz = v > 0.f ? (H3 ? da1[ct1][r] * uns_da : da1[ct1][r]) : 0.f;
                    r1[ct1][0] = fmaf(xv.x, dz, r1[ct1][0]);
                    r1[ct1][1] = fmaf(xv.y, dz, r1[ct1][1]);
                    r1[ct1][2] = fmaf(xv.z, dz, r1[ct1][2]);
                    if (D == 4) r1[ct1][3] = fmaf(xv.w, dz, r1[ct1][3]);
                    r1[ct1][4] += dz;
                }
            }
            // ---- 4. dW2 += dy2^T a1
            if (H3) {
                f16x8h bx[2][2][2];                         // [k-step s][c1 tile][plane]: a1 rows 16s + 8(j>>2) + 4h + (j&3), scaled 2^4
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int ct1 = 0; ct1 < 2; ++ct1) {
                        unsigned hi[4], lo[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            split_pair_h(a1v[ct1][8 * s + 2 * j] * sA1, a1v[ct1][8 * s + 2 * j + 1] * sA1, hi[j], lo[j]);
                        bx[s][ct1][0] = as_f16x8(hi[0], hi[1], hi[2], hi[3]);
                        bx[s][ct1][1] = as_f16x8(lo[0], lo[1], lo[2], lo[3]);
                    }
                constexpr int HA[3] = FACL_H3_PA, HB[3] = FACL_H3_PB;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    f16x8h at[2][2];                        // [k-step s][plane]: dy2^T [row c2 = 32 rt + q][k = position]
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int off = (rt * 32 + 16 * s) * 64;
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            at[s][p] = __builtin_bit_cast(f16x8h, tr_read_pair(tr_ptr[0] + p * B2S_PLANE + off, tr_ptr[1] + p * B2S_PLANE + off));
                    }
                    f32x16 tw[2];                           // this half unit's contribution at scale sD 2^4: two tiles interleaved
#pragma unroll
                    for (int r = 0; r < 16; ++r) tw[0][r] = tw[1][r] = 0.f;
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            tw[0] = MFMA_F16(at[s][HA[t]], bx[s][0][HB[t]], tw[0]);
                            tw[1] = MFMA_F16(at[s][HA[t]], bx[s][1][HB[t]], tw[1]);
                        }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        dw2[rt][0][r] = fmaf(tw[0][r], uns_dw, dw2[rt][0][r]);
                        dw2[rt][1][r] = fmaf(tw[1][r], uns_dw, dw2[rt][1][r]);
                    }
                }
            } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 bx[2][3];                            // B operand of k-step s: a1 rows 16s + 8(j>>2) + 4h + (j&3)
#pragma unroll
                for (int ct1 = 0; ct1 < 2; ++ct1) {
                    unsigned hi[4], mi[4], lo[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) split_pair(a1v[ct1][8 * s + 2 * j], a1v[ct1][8 * s + 2 * j + 1], hi[j], mi[j], lo[j]);
                    bx[ct1][0] = as_bf16x8(hi[0], hi[1], hi[2], hi[3]);
                    bx[ct1][1] = as_bf16x8(mi[0], mi[1], mi[2], mi[3]);
                    bx[ct1][2] = as_bf16x8(lo[0], lo[1], lo[2], lo[3]);
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    bf16x8 at[3];                           // A operand: dy2^T [row c2 = 32 rt + q][k = position]
                    const int off = (rt * 32 + 16 * s) * 64;
#pragma unroll
                    for (int p = 0; p < 3; ++p) at[p] = tr_read_pair(tr_ptr[0] + p * B2S_PLANE + off, tr_ptr[1] + p * B2S_PLANE + off);
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        dw2[rt][0] = MFMA_BF16(at[PA[t]], bx[0][PB[t]], dw2[rt][0]);
                        dw2[rt][1] = MFMA_BF16(at[PA[t]], bx[1][PB[t]], dw2[rt][1]);
                    }
                }
            }
            }
            WAVE_LDS_FENCE();                               // the next half unit overwrites the image (and xs4 after ct = 1)
        }
    }
    // ONE partial row per WORKGROUP: the four waves add their tiles in wave order into an fp64 area over the (dead) images and the
    // workgroup writes it out coalesced -- a quarter of the 75 MB of per-wave rows the reduction used to read back (its first
    // level: 13.6 -> ~4 us)
    __syncthreads();
    double* comb = reinterpret_cast<double*>(tab + 64 + B2S_WAVES * 64);
    float r1v[2][5];
#pragma unroll
    for (int ct1 = 0; ct1 < 2; ++ct1)
#pragma unroll
        for (int d = 0; d < 5; ++d) r1v[ct1][d] = r1[ct1][d] + __shfl_xor(r1[ct1][d], 32, 64);   // the two lane halves hold different positions of the same c1
    for (int w = 0; w < B2S_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        double* d = comb + (32 * a + rowmap(r, h)) * 64 + 32 * b + q;
                        *d = (w == 0 ? 0.0 : *d) + (double)dw2[a][b][r];
                    }
            if (h == 0) {                                      // R1 rows: x_0..x_{D-1}, 1, zeros
#pragma unroll
                for (int ct1 = 0; ct1 < 2; ++ct1)
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        float o = 0.f;
                        if (d < D) o = r1v[ct1][d];
                        else if (d == D) o = r1v[ct1][4];
                        double* dd = comb + 64 * 64 + d * 64 + 32 * ct1 + q;
                        *dd = (w == 0 ? 0.0 : *dd) + (double)o;
                    }
            }
        }
        __syncthreads();
    }
    double* row = part + (size_t)blockIdx.x * B2_V;
    for (int i = threadIdx.x; i < B2_V; i += 64 * B2S_WAVES) row[i] = comb[i];
}

}  // namespace

// launcher for facl_sa_bwd2 (sa_bwd.hip): `grid` workgroups of B2S_WAVES waves, one partial row per WORKGROUP in `ws`
int facl_sa_bwd2_sb_launch(const float* dz2f, const float* y2f, const float* x, int nunits, int D, const float* bw2,
                           const float* W2, const float* l1tab, double* ws, int grid, const uint32_t* a1amax, hipStream_t st) {
    const size_t lds = (1536 + 64 + B2S_WAVES * 64) * sizeof(float4) + B2S_WAVES * (size_t)B2S_IMG;
    static bool attr_done[64] = {};
    const void* fns[6] = {(const void*)k_sa_bwd2_sb<4, true>, (const void*)k_sa_bwd2_sb<3, true>, (const void*)k_sa_bwd2_sb<4, false>,
                          (const void*)k_sa_bwd2_sb<3, false>, (const void*)k_sa_bwd2_sb<4, true, true>, (const void*)k_sa_bwd2_sb<3, true, true>};
    if (int rc = facl_set_dynamic_lds(attr_done, fns, 6, (int)lds)) return rc;
    static const int rev = getenv("FACL_BWD2_REV") ? atoi(getenv("FACL_BWD2_REV")) : 1;
    static const int h3 = getenv("FACL_BWD_H3") ? atoi(getenv("FACL_BWD_H3")) : 1;     // 0: bf16x6 (A/B)
    const dim3 g(grid), b(64 * B2S_WAVES);
    static const int pref = getenv("FACL_BWD2_PREF") ? atoi(getenv("FACL_BWD2_PREF")) : 0;
    if (h3 && pref) {
        if (D == 4) hipLaunchKernelGGL((k_sa_bwd2_sb<4, true, true>), g, b, lds, st, dz2f, y2f, x, nunits, bw2, W2, l1tab, ws, rev, a1amax);
        else hipLaunchKernelGGL((k_sa_bwd2_sb<3, true, true>), g, b, lds, st, dz2f, y2f, x, nunits, bw2, W2, l1tab, ws, rev, a1amax);
    } else if (h3) {
        if (D == 4) hipLaunchKernelGGL((k_sa_bwd2_sb<4, true>), g, b, lds, st, dz2f, y2f, x, nunits, bw2, W2, l1tab, ws, rev, a1amax);
        else hipLaunchKernelGGL((k_sa_bwd2_sb<3, true>), g, b, lds, st, dz2f, y2f, x, nunits, bw2, W2, l1tab, ws, rev, a1amax);
    } else {
        if (D == 4) hipLaunchKernelGGL((k_sa_bwd2_sb<4, false>), g, b, lds, st, dz2f, y2f, x, nunits, bw2, W2, l1tab, ws, rev, a1amax);
        else hipLaunchKernelGGL((k_sa_bwd2_sb<3, false>), g, b, lds, st, dz2f, y2f, x, nunits, bw2, W2, l1tab, ws, rev, a1amax);
    }
    return facl_launch_status();
}

int facl_sa_bwd2_sb_waves() { return B2S_WAVES; }
