// Set-abstraction point-MLP, backward (autograd of net3DV_1, cn3d_model_conbag.py:43-58).
//
// Key identity ("BN-affine folding").  The max-pool sends the gradient to ONE position per
// (group, channel), but train-mode BN3 makes dL/dy3 dense again:
//     dy3[p,c] = s3_c*dz3[p,c]  -  (s3_c/P) * (dbeta3_c + yhat3[p,c]*dgamma3_c)
// The dense part is affine in y3 = a2 W3^T + b3, hence quadratic-form algebra on the 64-wide a2:
//     da2[p,:] = a2[p,:] G3 + h3 + (sparse rows)        G3 = W3^T diag(g) W3   (64x64)
//     dW3      = sparse gather  +  closed form in  sum_p a2^T a2  and  sum_p a2
// so the 256-channel activations are never re-materialised (the stock autograd path stores and
// re-reads 4.8 GB of them at the headline shape) and layer 3's backward costs 64x64 MFMA work
// per position instead of 2 x 64x256.
//
// Passes (all activations in the fragment layout of common.h, lane = position, register = channel):
//   facl_sa_bwd0   dpooled -> coef = s3*dz3 (sparse values), sums (dbeta3, dgamma3)
//   facl_sa_bwd1   y2 -> a2 ; da2 = a2 G3 + h3' + scatter(coef, arg, W3) ; dz2 = da2*[z2>0] (stored);
//                  sums (dbeta2, dgamma2)
//   facl_sa_bwd_w3 y2 -> a2 ; Gram sum_p a2^T a2, sum_p a2, sparse part of dW3
//   facl_sa_bwd2   dz2,y2,x -> dy2 ; da1 = dy2 W2 ; dz1 = da1*[z1>0] ; dW2 += dy2^T a1 ;
//                  R1 += [x|1]^T dz1   (everything layer 1 needs: dW1, dgamma1, dbeta1 follow in closed form)
// Roofline: MFMA fp32 for bwd1/bwd2/bwd_w3 (128..320 MFMA 32x32x2 per 64 positions), HBM for bwd0.
#include "common.h"
#include "sa_bwd1_scatter.inc"
#include <stdlib.h>

int facl_reduce_rows(const double* part, int rows, int V, double* out, hipStream_t st);
int facl_reduce_rows_f32(const float* part, int rows, int V, double* out, hipStream_t st);

namespace {

constexpr int SA_GRID = 256;
constexpr int TP = 68;    // padded row (floats) of a [64 pos][64 ch] LDS tile read with b128 (272 B: 16-B aligned rows)
constexpr int TQ = 65;    // padded row of a tile accessed with b32 only (conflict-free columns)

// Lanes of ONE wave exchange data through LDS (write in one layout, read in another).  The hardware
// executes a wave's DS operations in order; this only has to stop the COMPILER from moving LDS
// accesses across the hand-off (and drain the counter so the data has landed).
#define WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// ---------------------------------------------------------------------------------------------
// bwd0: one thread per channel, rows strided over the grid.
__global__ __launch_bounds__(256) void k_sa_bwd0(const float* __restrict__ dpooled, const float* __restrict__ ymax,
                                                 int rows, const float* __restrict__ bnc3,
                                                 float* __restrict__ coef, double* __restrict__ part) {
    // block = 64 channel quads (16 B per lane) x 4 row phases; the phases are combined through LDS so that a block
    // writes ONE partial row of 512 doubles
    __shared__ double red[3][64][8];
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const float4 mean = reinterpret_cast<const float4*>(bnc3)[lane], invstd = reinterpret_cast<const float4*>(bnc3 + 256)[lane];
    const float4 scale = reinterpret_cast<const float4*>(bnc3 + 512)[lane], shift = reinterpret_cast<const float4*>(bnc3 + 768)[lane];
    const float4 sgn = reinterpret_cast<const float4*>(bnc3 + 1024)[lane];
    const float mn[4] = {mean.x, mean.y, mean.z, mean.w}, iv[4] = {invstd.x, invstd.y, invstd.z, invstd.w};
    const float sc[4] = {scale.x, scale.y, scale.z, scale.w}, sh[4] = {shift.x, shift.y, shift.z, shift.w};
    const float sg[4] = {sgn.x, sgn.y, sgn.z, sgn.w};
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = blockIdx.x * 4 + ph; r < rows; r += gridDim.x * 4) {
        const float4 ym4 = reinterpret_cast<const float4*>(ymax)[(size_t)r * 64 + lane];
        const float4 dp4 = reinterpret_cast<const float4*>(dpooled)[(size_t)r * 64 + lane];
        const float ym[4] = {ym4.x, ym4.y, ym4.z, ym4.w}, dp[4] = {dp4.x, dp4.y, dp4.z, dp4.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = fmaf(fabsf(sc[e]), ym[e], sh[e]);
            const float dz = z > 0.f ? dp[e] : 0.f;
            const float yhat = (sg[e] * ym[e] - mn[e]) * iv[e];
            o[e] = sc[e] * dz;
            acc[2 * e] += (double)dz;
            acc[2 * e + 1] += (double)dz * (double)yhat;
        }
        reinterpret_cast<float4*>(coef)[(size_t)r * 64 + lane] = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (ph > 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[ph - 1][lane][e] = acc[e];
    }
    __syncthreads();
    if (ph == 0) {
        double* pr = part + (size_t)blockIdx.x * 512 + 8 * lane;          // channels 4*lane..4*lane+3, (dbeta, dgamma) pairs
#pragma unroll
        for (int e = 0; e < 8; ++e) pr[e] = ((acc[e] + red[0][lane][e]) + red[1][lane][e]) + red[2][lane][e];
    }
}

// ---------------------------------------------------------------------------------------------
// bwd1.  LDS: g3f 16 KiB | w3n 64 KiB | tables | 4 x T (64 x 68 floats).
// Producer/consumer workgroup of 8 waves = 4 pairs, one unit per pair and round.  The two halves of the work use
// different hardware and used to run back to back in one wave (40k cycles per unit at 1 wave/SIMD):
//   * "scatter" wave (waves 4-7): the sparse rows  T[arg[c]][j] += coef[c] * W3[c][j]  -- VALU/SALU/LDS-read bound,
//     64 row accumulators in registers through GPR indexing (sa_bwd1_scatter.inc);
//   * "dense" wave (waves 0-3): a2 -> a2 G3 + h3' on the MFMA, then combine with T, mask, store dz2, dbeta2/dgamma2.
// Wave w and w+4 land on the same SIMD (4 SIMDs, cyclic placement), so every SIMD runs one wave of each kind and the
// MFMA pipe works in the shadow of the scatter's instruction stream.  Hand-off through the pair's single T tile with two
// workgroup barriers per round:  A = "dense has finished reading the previous T" (inside the scatter's asm block, just
// before its 64 row stores),  B = "T is written".  Between B and the next A the dense wave does ALL of its unit (MFMAs
// and epilogue, one output-channel half at a time to stay inside 256 registers) while the scatter wave accumulates the
// next unit in registers: the two roles are skewed by one unit.
__global__ __launch_bounds__(512) void k_sa_bwd1(const float* __restrict__ y2f, int nunits,
                                                 const float* __restrict__ bnc2, const float* __restrict__ G3,
                                                 const float* __restrict__ h3, const float* __restrict__ W3,
                                                 const float* __restrict__ coef, const unsigned char* __restrict__ arg,
                                                 float* __restrict__ dz2f, double* __restrict__ part,
                                                 const unsigned* __restrict__ a2amax) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    float g3scale, g3uns, hs;                             // hs = (G3 scale)(a2 scale): what the accumulators ride at
    float* w3n = reinterpret_cast<float*>(lds4 + 1024);   // W3 natural (256,64)
    float4* tab = lds4 + 1024 + 4096;                     // mean2, invstd2, scale2, shift2, h3: 5 x 16 float4
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave & 3;
    float* T = reinterpret_cast<float*>(tab + 80) + pair * (64 * TP);
    // G3 = W3^T diag(g) W3 carries the gradient's magnitude: its fp16x3 planes (common.h) use the power of two that puts
    // max|G3| in [2^13, 2^14) -- found by the workgroup itself, no host round trip.  Fragment order of the A operand of
    // v_mfma_f32_32x32x16_f16: entry ((ro*4 + kb)*2 + plane)*64 + lane holds, for lane (row j = 32 ro + q, half h), the 8
    // k-slots t of k-block kb = 2 rt + m: channel 32 rt + 16 m + (t & 3) + 8 (t >> 2) + 4 h -- exactly the channels the lane's
    // y2 fragment registers 8m .. 8m+7 hold (rowmap), so the activations need no shuffle.
    uint4* g3p = reinterpret_cast<uint4*>(lds4);          // 2 x 4 x 2 x 64 uint4 = 16 KiB (in place of the fp32 fragments)
    float* red = reinterpret_cast<float*>(lds4 + 1024 + 4096 + 80) ;   // the T tiles are idle until the first round: 8 floats
    {
        float m = 0.f;
        for (int i = threadIdx.x; i < 1024; i += 512) {
            const float4 v = reinterpret_cast<const float4*>(G3)[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        m = red[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
        __syncthreads();
        // scale 2^(13 - floor(log2 m)) (m = 0, denormal or huge: clamped so that scale and its inverse stay normal)
        const int se = __builtin_amdgcn_readfirstlane(h3_se_wide(__float_as_uint(m)));   // wave-uniform: the factors stay in scalar registers
        const int seA2 = h3_se_of(a2amax);                             // fp16x3 scale of a2 = relu(bn2(y2)): the bound the forward used
        g3scale = pow2_biased(se);
        g3uns = h3_unscale(se, seA2);                                  // 1 / (G3 scale * a2 scale)
        hs = g3scale * pow2_biased(seA2);
        if (threadIdx.x >= 32 && threadIdx.x < 64) {                   // (scale2, shift2) ride at a2's scale: relu(s sc y + s sh) = s relu(sc y + sh)
            const float sA2 = pow2_biased(seA2);                       // exactly for a power of two s; the sign test of the ReLU mask is unchanged
            float4 t = reinterpret_cast<const float4*>(bnc2)[threadIdx.x];
            t.x *= sA2; t.y *= sA2; t.z *= sA2; t.w *= sA2;
            tab[threadIdx.x] = t;
        }
    }
    for (int i = threadIdx.x; i < 512; i += 512) {
        const int ln = i & 63, kb = (i >> 6) & 3, ro = i >> 8;
        const int rt = kb >> 1, mm = kb & 1;
        const float* row = G3 + (32 * ro + (ln & 31)) * 64 + 32 * rt + 16 * mm + 4 * (ln >> 5);
        const float4 v0 = *reinterpret_cast<const float4*>(row), v1 = *reinterpret_cast<const float4*>(row + 8);
        unsigned hi[4], lo[4];
        split_pair_h(v0.x * g3scale, v0.y * g3scale, hi[0], lo[0]);
        split_pair_h(v0.z * g3scale, v0.w * g3scale, hi[1], lo[1]);
        split_pair_h(v1.x * g3scale, v1.y * g3scale, hi[2], lo[2]);
        split_pair_h(v1.z * g3scale, v1.w * g3scale, hi[3], lo[3]);
        g3p[((ro * 4 + kb) * 2) * 64 + ln] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        g3p[((ro * 4 + kb) * 2 + 1) * 64 + ln] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
    for (int i = threadIdx.x; i < 4096; i += 512)
        reinterpret_cast<float4*>(w3n)[i] = reinterpret_cast<const float4*>(W3)[i];
    if (threadIdx.x < 32) tab[threadIdx.x] = reinterpret_cast<const float4*>(bnc2)[threadIdx.x];   // mean2, invstd2 (scale2, shift2: above)
    if (threadIdx.x < 16) tab[64 + threadIdx.x] = reinterpret_cast<const float4*>(h3)[threadIdx.x];
    __syncthreads();
    const float4* mean2 = tab; const float4* inv2 = tab + 16; const float4* sc2 = tab + 32; const float4* sh2 = tab + 48;
    const float4* h3s = tab + 64;

    const int lane = lane_id(), h = lane >> 5, q = lane & 31;
    const int nw = gridDim.x * 4;                         // units in flight over the whole grid
    const int u0 = blockIdx.x * 4 + pair;
    // every wave of the workgroup runs the same number of rounds (the barriers are workgroup-wide): pair 0's count
    const int rounds = (nunits - (int)blockIdx.x * 4 + nw - 1) / nw;

    if (wave >= 4) {
        // =================================================================== scatter role
        // The asm block reads the unit's 256 coefficients / argmax bytes through scalar loads; the vector loads below
        // only warm the NEXT unit's rows into L2 one round ahead (their values are handed to the block as unused
        // operands so that they are not optimised away).
        float4 cfn = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned arn = 0;
        const int ulast = nunits - 1;
        for (int r = 0; r < rounds; ++r) {
            const int u = u0 + r * nw;
            // an inactive pair (u >= nunits in the last round) re-runs the last unit: its T is never read
            const int uc = u < nunits ? u : ulast, un = u + nw < nunits ? u + nw : ulast;
            const float* cu = coef + (size_t)uc * 256;
            const unsigned char* au = arg + (size_t)uc * 256;
            const float4 cf4 = cfn;
            const unsigned ar4 = arn;
            cfn = *reinterpret_cast<const float4*>(coef + (size_t)un * 256 + 4 * lane);
            arn = *reinterpret_cast<const unsigned*>(arg + (size_t)un * 256 + 4 * lane);
            {
                // 64 row accumulators in v192..v255, addressed by GPR indexing (SRC2+DST) inside ONE hand-written block
                // (tools/gen_scatter_asm.py): per element s_bfe + s_set_gpr_idx_idx + s_nop + v_fma + one ds_read_b32
                // (hipcc's own indexed RMW costs ~110 cycles per element because it toggles the index mode around
                // separate read / add / write moves).  The block ends with  s_barrier (A)  +  the 64 row stores into T.
                const unsigned waddr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float*)(w3n + lane));   // LDS byte address
                const unsigned taddr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float*)(T + lane));
                asm volatile(FACL_SCATTER_ASM_TEXT
                             :
                             : [cp] "s"(cu), [ap] "s"(au), [wa] "v"(waddr), [ta] "v"(taddr), "v"(cf4.x), "v"(ar4)
                             : FACL_SCATTER_ASM_CLOBBERS);
            }
            WAVE_LDS_FENCE();                 // the row stores have landed
            __builtin_amdgcn_s_barrier();     // B
        }
        return;
    }

    // ======================================================================= dense role
    float pb[2][16], pg[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { pb[a][r] = 0.f; pg[a][r] = 0.f; }

    // No register double buffer here (256 registers: y 64 + dbeta/dgamma partials 64 + accumulators 32 + operands): the
    // unit's y2 tile is requested BEFORE the two barriers and consumed after them, so the time this wave waits for
    // its scatter partner covers the load latency.
    for (int r = 0; r < rounds; ++r) {
        const int u = u0 + r * nw;
        const bool active = u < nunits;       // wave-uniform
        asm volatile("" ::: "memory");
        float4 yn[16];
        if (active) {
            const float* tl = y2f + (size_t)u * FACL_UNIT_ELEMS;
#pragma unroll
            for (int i = 0; i < 16; ++i) yn[i] = *reinterpret_cast<const float4*>(tl + (i * 64 + lane) * 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();         // A: this wave's reads of the previous T are complete
        __builtin_amdgcn_s_barrier();         // B: the scatter wave has written this unit's T
        asm volatile("" ::: "memory");
        if (!active) continue;
        float yv[2][2][16];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 y = yn[(ct * 2 + rt) * 4 + r4];
                    yv[ct][rt][4 * r4] = y.x; yv[ct][rt][4 * r4 + 1] = y.y; yv[ct][rt][4 * r4 + 2] = y.z; yv[ct][rt][4 * r4 + 3] = y.w;
                }
        // One output-channel half (ro) at a time: 64 MFMAs into 32 accumulators, then that half's epilogue.  The whole
        // unit runs while the scatter wave accumulates the NEXT unit.  Table / fragment / T reads are software-pipelined
        // one step ahead by hand and fenced with sched_barrier: left alone, the scheduler hoists all of a section's LDS
        // reads to its top (100+ extra live registers -> scratch spills at the 256-register budget).
        float* otile = dz2f + (size_t)u * FACL_UNIT_ELEMS;
#pragma unroll
        for (int ro = 0; ro < 2; ++ro) {
            // ---- dense part on the MFMA: D^T[j][p] = sum_k G3[j][k] a2[p][k] + h3'[j], fp16x3 (common.h): G3 planes from LDS
            //      (scaled by g3scale), a2 = relu(bn2(y2)) * 2^4 split in registers; 4 k-blocks x 3 products x 2 position tiles
            f32x16 acc[2];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 hh = h3s[8 * ro + 2 * r4 + h];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc[ct][4 * r4] = hh.x * hs; acc[ct][4 * r4 + 1] = hh.y * hs; acc[ct][4 * r4 + 2] = hh.z * hs; acc[ct][4 * r4 + 3] = hh.w * hs;
                }
            }
            constexpr int HA[3] = FACL_H3_PA, HB[3] = FACL_H3_PB;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const int rt = kb >> 1, mm = kb & 1;
                // (no register prefetch of the next block's fragments / tables here: the dense role has no registers to spare,
                // and its SIMD partner, the scatter wave, fills the LDS latency)
                const uint4 g0 = g3p[((ro * 4 + kb) * 2) * 64 + lane], g1 = g3p[((ro * 4 + kb) * 2 + 1) * 64 + lane];
                const float4 sc0 = sc2[8 * rt + 4 * mm + h], sc1 = sc2[8 * rt + 4 * mm + 2 + h];
                const float4 sh0 = sh2[8 * rt + 4 * mm + h], sh1 = sh2[8 * rt + 4 * mm + 2 + h];
                __builtin_amdgcn_sched_barrier(0);
                const f16x8h gf[2] = {__builtin_bit_cast(f16x8h, g0), __builtin_bit_cast(f16x8h, g1)};
                const float scv[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
                const float shv[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    float a2[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) a2[t] = fmaxf(fmaf(scv[t], yv[ct][rt][8 * mm + t], shv[t]), 0.f);   // a2 * sA2 (tables)
                    unsigned hi[4], lo[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) split_pair_h(a2[2 * t], a2[2 * t + 1], hi[t], lo[t]);
                    const f16x8h af[2] = {as_f16x8(hi[0], hi[1], hi[2], hi[3]), as_f16x8(lo[0], lo[1], lo[2], lo[3])};
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[ct] = MFMA_F16(gf[HA[t]], af[HB[t]], acc[ct]);   // (lo,hi) (hi,lo) (hi,hi)
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] *= g3uns;              // exact: a power of two
            // ---- combine, mask with z2 > 0, store dz2, accumulate dbeta2 / dgamma2
            float4 tn0 = *reinterpret_cast<const float4*>(&T[q * TP + 32 * ro + 4 * h]);
            float4 tn1 = *reinterpret_cast<const float4*>(&T[(32 + q) * TP + 32 * ro + 4 * h]);
            float4 escn = sc2[8 * ro + h], eshn = sh2[8 * ro + h], emun = mean2[8 * ro + h], eivn = inv2[8 * ro + h];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 t0 = tn0, t1 = tn1, sc = escn, sh = eshn, mu = emun, iv = eivn;
                if (r4 < 3) {
                    tn0 = *reinterpret_cast<const float4*>(&T[q * TP + 32 * ro + 8 * (r4 + 1) + 4 * h]);
                    tn1 = *reinterpret_cast<const float4*>(&T[(32 + q) * TP + 32 * ro + 8 * (r4 + 1) + 4 * h]);
                    escn = sc2[8 * ro + 2 * (r4 + 1) + h]; eshn = sh2[8 * ro + 2 * (r4 + 1) + h];
                    emun = mean2[8 * ro + 2 * (r4 + 1) + h]; eivn = inv2[8 * ro + 2 * (r4 + 1) + h];
                }
                __builtin_amdgcn_sched_barrier(0);
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
                const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, ivv[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const float4 t = ct ? t1 : t0;
                    const float tv[4] = {t.x, t.y, t.z, t.w};
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = yv[ct][ro][4 * r4 + e];
                        const bool on = fmaf(scv[e], y, shv[e]) > 0.f;
                        const float d = on ? acc[ct][4 * r4 + e] + tv[e] : 0.f;
                        o[e] = d;
                        pb[ro][4 * r4 + e] += d;
                        pg[ro][4 * r4 + e] = fmaf(d, (y - muv[e]) * ivv[e], pg[ro][4 * r4 + e]);
                    }
                    *reinterpret_cast<float4*>(otile + (((ct * 2 + ro) * 4 + r4) * 64 + lane) * 4) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
    const int wave_g = blockIdx.x * 4 + pair;
#pragma unroll
    for (int ro = 0; ro < 2; ++ro)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double sdb = pb[ro][r], g = pg[ro][r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { sdb += __shfl_xor(sdb, o, 64); g += __shfl_xor(g, o, 64); }
            if (q == 0) {
                const int c = 32 * ro + rowmap(r, h);
                part[(size_t)wave_g * 128 + 2 * c] = sdb;
                part[(size_t)wave_g * 128 + 2 * c + 1] = g;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// bwd_w3: one wave per unit (no workgroup barriers in the loop).  Per unit: a2 -> padded LDS tile;
//   Gram sum_p a2^T a2 on MFMA (3 of the 4 symmetric 32x32 tiles), sum_p a2, and the sparse part of dW3:
//   lane owns channels c = lane + 64e and adds coef[c] * a2[arg[c]][:] (one LDS row, 16 b128 reads) into 64
//   register accumulators per channel.  Output row per WORKGROUP (the 4 waves are combined through LDS):
//   [ sparse dW3 (256 x 64) | Gram (64 x 64) | sum a2 (64) ]  = 20544 doubles
constexpr int W3_V = 256 * 64 + 64 * 64 + 64;

__global__ __launch_bounds__(256) void k_sa_bwd_w3(const float* __restrict__ y2f, int nunits,
                                                   const float* __restrict__ bnc2, const float* __restrict__ coef,
                                                   const unsigned char* __restrict__ arg, double* __restrict__ part,
                                                   const unsigned* __restrict__ a2amax) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    const int seA2 = h3_se_of(a2amax);                    // fp16x3 scale of a2 (both operands of the Gram)
    const float sA2 = pow2_biased(seA2), GU = h3_unscale(seA2, seA2), iA2 = pow2_biased(254 - seA2);
    float4* tab = lds4;                                         // scale2, shift2: 2 x 16 float4
    float* comb = reinterpret_cast<float*>(lds4 + 32);          // [256][65] sparse dW3 combine (padded rows)
    float* gcomb = comb + 256 * 65;                             // [64][64] Gram combine
    float* scomb = gcomb + 64 * 64;                             // [64] sum a2
    float* T = scomb + 64 + (threadIdx.x >> 6) * (64 * TP);     // per-wave a2 tile; (256*65+4096+64)*4 B is 16-B aligned
    if (threadIdx.x < 32) {
        float4 t = reinterpret_cast<const float4*>(bnc2 + 128)[threadIdx.x];
        t.x *= sA2; t.y *= sA2; t.z *= sA2; t.w *= sA2;      // the tile holds a2 * sA2 (exact: a power of two); sums are scaled back below
        tab[threadIdx.x] = t;
    }
    __syncthreads();
    const float4* sc2 = tab; const float4* sh2 = tab + 16;
    const int lane = lane_id(), h = lane >> 5, q = lane & 31, wave = threadIdx.x >> 6;
    const int wave_g = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;

    float accs[4][64];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 64; ++k) accs[e][k] = 0.f;
    f32x16 g00, g01, g11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { g00[r] = 0.f; g01[r] = 0.f; g11[r] = 0.f; }
    float s2a = 0.f, s2b = 0.f;

    // register double buffer: the NEXT unit's tile (+ its coef/arg) is requested before this unit's compute, so
    // the HBM latency is covered by the MFMA / LDS work instead of being exposed once per unit (1 wave per SIMD)
    float4 yn[16];
    float cfn[4];
    int psn[4];
    auto issue_loads = [&](int u) {
        const float* tile = y2f + (size_t)u * FACL_UNIT_ELEMS;
#pragma unroll
        for (int i = 0; i < 16; ++i) yn[i] = *reinterpret_cast<const float4*>(tile + (i * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cfn[e] = coef[(size_t)u * 256 + 64 * e + lane];
            psn[e] = arg[(size_t)u * 256 + 64 * e + lane];
        }
    };
    if (wave_g < nunits) issue_loads(wave_g);
    for (int u = wave_g; u < nunits; u += nwaves) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 y = yn[(ct * 2 + rt) * 4 + r4];
                    const float4 sc = sc2[8 * rt + 2 * r4 + h], sh = sh2[8 * rt + 2 * r4 + h];
                    float4 a;
                    a.x = fmaxf(fmaf(sc.x, y.x, sh.x), 0.f); a.y = fmaxf(fmaf(sc.y, y.y, sh.y), 0.f);
                    a.z = fmaxf(fmaf(sc.z, y.z, sh.z), 0.f); a.w = fmaxf(fmaf(sc.w, y.w, sh.w), 0.f);
                    *reinterpret_cast<float4*>(&T[(32 * ct + q) * TP + 32 * rt + 8 * r4 + 4 * h]) = a;
                }
        float cfv[4];
        int psv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { cfv[e] = cfn[e]; psv[e] = psn[e]; }
        if (u + nwaves < nunits) issue_loads(u + nwaves);   // (unconditional, as in k_sa_fwd3_sb: measured 5 % slower here)
        WAVE_LDS_FENCE();
        // Gram: G[i][j] += sum_p a2[p][i] a2[p][j] on fp16x3 (common.h; both operands are activations at the scale of a2's bound,
        // the accumulators are scaled back when the waves are combined).  Operand fragments (lane = channel q of
        // the tile, k-slots = positions 16 ks + 8 h + t) come out of the padded tile with 8 ds_read_b32 each -- the same 128
        // reads per unit the fp32 form issued -- and serve as A and as B operand alike; 36 MFMAs of 8 passes per unit instead
        // of 96 of 16 (ablation: the fp32 Gram was 0.115 of the pass's 0.37 ms).
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float v0[8], v1[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v0[t] = T[(16 * ks + 8 * h + t) * TP + q];
                v1[t] = T[(16 * ks + 8 * h + t) * TP + 32 + q];
            }
            unsigned h0[4], l0[4], h1[4], l1[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s2a += v0[2 * t] + v0[2 * t + 1];
                s2b += v1[2 * t] + v1[2 * t + 1];
                split_pair_h(v0[2 * t], v0[2 * t + 1], h0[t], l0[t]);
                split_pair_h(v1[2 * t], v1[2 * t + 1], h1[t], l1[t]);
            }
            const f16x8h H0 = as_f16x8(h0[0], h0[1], h0[2], h0[3]), L0 = as_f16x8(l0[0], l0[1], l0[2], l0[3]);
            const f16x8h H1 = as_f16x8(h1[0], h1[1], h1[2], h1[3]), L1 = as_f16x8(l1[0], l1[1], l1[2], l1[3]);
            g00 = MFMA_F16(L0, H0, g00); g00 = MFMA_F16(H0, L0, g00); g00 = MFMA_F16(H0, H0, g00);   // smallest terms first
            g01 = MFMA_F16(L0, H1, g01); g01 = MFMA_F16(H0, L1, g01); g01 = MFMA_F16(H0, H1, g01);
            g11 = MFMA_F16(L1, H1, g11); g11 = MFMA_F16(H1, L1, g11); g11 = MFMA_F16(H1, H1, g11);
        }
        // sparse part of dW3 (same basic block as the Gram MFMAs: the scheduler sinks these FMAs into the MFMA shadow)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float4* row = reinterpret_cast<const float4*>(&T[psv[e] * TP]);
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) {
                const float4 v = row[k4];
                accs[e][4 * k4 + 0] = fmaf(cfv[e], v.x, accs[e][4 * k4 + 0]);
                accs[e][4 * k4 + 1] = fmaf(cfv[e], v.y, accs[e][4 * k4 + 1]);
                accs[e][4 * k4 + 2] = fmaf(cfv[e], v.z, accs[e][4 * k4 + 2]);
                accs[e][4 * k4 + 3] = fmaf(cfv[e], v.w, accs[e][4 * k4 + 3]);
            }
        }
        WAVE_LDS_FENCE();      // next unit overwrites T
    }
    // combine the 4 waves through LDS (wave w adds in phase w), then one fp64 row per workgroup
    const float s2ta = s2a + __shfl_xor(s2a, 32, 64), s2tb = s2b + __shfl_xor(s2b, 32, 64);
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < 64; ++k) {
                    float* d = &comb[(64 * e + lane) * 65 + k];
                    *d = (w == 0 ? 0.f : *d) + accs[e][k] * iA2;              // the tile rode at a2 * sA2: exact inverse
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = rowmap(r, h);
                float* d00 = &gcomb[i * 64 + q];
                float* d01 = &gcomb[i * 64 + 32 + q];
                float* d11 = &gcomb[(32 + i) * 64 + 32 + q];
                *d00 = (w == 0 ? 0.f : *d00) + g00[r] * GU;
                *d01 = (w == 0 ? 0.f : *d01) + g01[r] * GU;
                *d11 = (w == 0 ? 0.f : *d11) + g11[r] * GU;
            }
            if (h == 0) {
                scomb[q] = (w == 0 ? 0.f : scomb[q]) + s2ta * iA2;
                scomb[32 + q] = (w == 0 ? 0.f : scomb[32 + q]) + s2tb * iA2;
            }
        }
        __syncthreads();
    }
    double* row = part + (size_t)blockIdx.x * W3_V;
    for (int i = threadIdx.x; i < 256 * 64; i += 256) row[i] = (double)comb[(i >> 6) * 65 + (i & 63)];
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int a = i >> 6, b = i & 63;
        row[256 * 64 + i] = (double)((a >= 32 && b < 32) ? gcomb[b * 64 + a] : gcomb[i]);     // lower-left = upper-right^T
    }
    if (threadIdx.x < 64) row[256 * 64 + 64 * 64 + threadIdx.x] = (double)scomb[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// bwd_w3, round 4 (late): TWO waves per unit, free-running.  k_sa_bwd_w3 keeps 4 x 64 sparse accumulators per lane (474 registers:
// one wave per SIMD, every LDS / HBM wait exposed -- PMC: 51 % of the wave cycles parked).  Here waves (2p, 2p + 1) of a
// workgroup walk the SAME units; each builds the unit's whole a2 tile in its OWN LDS tile (the partner's second read of the
// 16 KiB comes out of the L2; no barrier, no hand-off) and takes HALF of the work on it: wave `role` the sparse rows of
// channels e in {2 role, 2 role + 1} (128 accumulators), the diagonal Gram tile G[role][role] and half the k-steps of
// G[0][1] (18 of the 36 MFMAs).  <= 256 registers -> two waves per SIMD, and the two waves of a SIMD (w, w + 4) belong to
// different pairs, i.e. sit in different phases of different units.  A first form with ONE shared tile per pair, each wave
// building half of it behind a workgroup barrier per unit, measured 0.33 ms against the one-wave kernel's 0.267
// (gpurun_out/r5e_ab.log): eight waves in lockstep wait for memory together.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_sa_bwd_w3p(const float* __restrict__ y2f, int nunits, const float* __restrict__ bnc2, const float* __restrict__ coef,
                  const unsigned char* __restrict__ arg, double* __restrict__ part, const unsigned* __restrict__ a2amax) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    const int seA2 = h3_se_of(a2amax);
    const float sA2 = pow2_biased(seA2), GU = h3_unscale(seA2, seA2), iA2 = pow2_biased(254 - seA2);
    float4* tab = lds4;                                         // scale2, shift2: 2 x 16 float4
    float* base = reinterpret_cast<float*>(lds4 + 32);
    // loop: 8 tiles of [64][TP], one per wave; afterwards the same bytes hold the combine areas
    float* comb = base;                                         // [256][65]
    float* gcomb = comb + 256 * 65;                             // [64][64]
    float* scomb = gcomb + 64 * 64;                             // [64]
    const int lane = lane_id(), h = lane >> 5, q = lane & 31, wave = threadIdx.x >> 6;
    const int pair = wave >> 1, role = wave & 1;
    float* T = base + wave * (64 * TP);
    if (threadIdx.x < 32) {
        float4 t = reinterpret_cast<const float4*>(bnc2 + 128)[threadIdx.x];
        t.x *= sA2; t.y *= sA2; t.z *= sA2; t.w *= sA2;
        tab[threadIdx.x] = t;
    }
    __syncthreads();
    const float4* sc2 = tab; const float4* sh2 = tab + 16;
    const int pair_g = blockIdx.x * 4 + pair, npairs = gridDim.x * 4;

    float accs[2][64];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int k = 0; k < 64; ++k) accs[e][k] = 0.f;
    // Gram tiles of this wave: gm = G[role][role] over all 64 positions, gx = its half (k-steps 2 role, 2 role + 1) of G[0][1]
    f32x16 gm, gx;
#pragma unroll
    for (int r = 0; r < 16; ++r) { gm[r] = 0.f; gx[r] = 0.f; }
    float s2m = 0.f;                                            // sum over positions of a2[:, 32 role + q] (this lane: its h-half of them)

    for (int u = pair_g; u < nunits; u += npairs) {
        {
            const float* tile = y2f + (size_t)u * FACL_UNIT_ELEMS;
            float4 yn[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) yn[i] = *reinterpret_cast<const float4*>(tile + (i * 64 + lane) * 4);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float4 y = yn[(ct * 2 + rt) * 4 + r4];
                        const float4 sc = sc2[8 * rt + 2 * r4 + h], sh = sh2[8 * rt + 2 * r4 + h];
                        float4 a;
                        a.x = fmaxf(fmaf(sc.x, y.x, sh.x), 0.f); a.y = fmaxf(fmaf(sc.y, y.y, sh.y), 0.f);
                        a.z = fmaxf(fmaf(sc.z, y.z, sh.z), 0.f); a.w = fmaxf(fmaf(sc.w, y.w, sh.w), 0.f);
                        *reinterpret_cast<float4*>(&T[(32 * ct + q) * TP + 32 * rt + 8 * r4 + 4 * h]) = a;
                    }
        }
        float cfv[2];
        int psv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            cfv[e] = coef[(size_t)u * 256 + 64 * (2 * role + e) + lane];
            psv[e] = arg[(size_t)u * 256 + 64 * (2 * role + e) + lane];
        }
        WAVE_LDS_FENCE();
        // Gram on fp16x3 (as k_sa_bwd_w3): operand fragments = columns of the tile (lane = channel, k-slots = positions
        // 16 ks + 8 h + t).  This wave: its diagonal tile over all four k-steps, the off-diagonal tile over two of them.
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float vm[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) vm[t] = T[(16 * ks + 8 * h + t) * TP + 32 * role + q];
            unsigned hm[4], lm[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s2m += vm[2 * t] + vm[2 * t + 1];
                split_pair_h(vm[2 * t], vm[2 * t + 1], hm[t], lm[t]);
            }
            const f16x8h HM = as_f16x8(hm[0], hm[1], hm[2], hm[3]), LM = as_f16x8(lm[0], lm[1], lm[2], lm[3]);
            gm = MFMA_F16(LM, HM, gm); gm = MFMA_F16(HM, LM, gm); gm = MFMA_F16(HM, HM, gm);         // smallest terms first
            if ((ks >> 1) == role) {                            // wave-uniform
                float vo[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) vo[t] = T[(16 * ks + 8 * h + t) * TP + 32 * (role ^ 1) + q];   // (role ^ 1, not 1 - role: a subtraction keeps the constant part out of the ds_read offset field)
                unsigned ho[4], lo[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) split_pair_h(vo[2 * t], vo[2 * t + 1], ho[t], lo[t]);
                const f16x8h HO = as_f16x8(ho[0], ho[1], ho[2], ho[3]), LO = as_f16x8(lo[0], lo[1], lo[2], lo[3]);
                if (role == 0) {                                // G[0][1] = sum a2[:, 0:32]^T a2[:, 32:64]: A = block 0, B = block 1
                    gx = MFMA_F16(LM, HO, gx); gx = MFMA_F16(HM, LO, gx); gx = MFMA_F16(HM, HO, gx);
                } else {
                    gx = MFMA_F16(LO, HM, gx); gx = MFMA_F16(HO, LM, gx); gx = MFMA_F16(HO, HM, gx);
                }
            }
        }
        // sparse part of dW3: lane owns channels 64 (2 role + e) + lane and adds coef * a2[arg][:] (one LDS row each)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float4* row = reinterpret_cast<const float4*>(&T[psv[e] * TP]);
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) {
                const float4 v = row[k4];
                accs[e][4 * k4 + 0] = fmaf(cfv[e], v.x, accs[e][4 * k4 + 0]);
                accs[e][4 * k4 + 1] = fmaf(cfv[e], v.y, accs[e][4 * k4 + 1]);
                accs[e][4 * k4 + 2] = fmaf(cfv[e], v.z, accs[e][4 * k4 + 2]);
                accs[e][4 * k4 + 3] = fmaf(cfv[e], v.w, accs[e][4 * k4 + 3]);
            }
        }
        WAVE_LDS_FENCE();      // next unit overwrites T
    }
    __syncthreads();                                            // the tiles are dead: their bytes become the combine areas
    const float s2t = s2m + __shfl_xor(s2m, 32, 64);
    // sparse rows, diagonal Gram tiles and sum a2: the two roles own disjoint halves, the four pairs add in pair order;
    // the off-diagonal tile G[0][1] has a partial sum in all eight waves, added in wave order
    for (int w = 0; w < 4; ++w) {
        if (pair == w) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int k = 0; k < 64; ++k) {
                    float* d = &comb[(64 * (2 * role + e) + lane) * 65 + k];
                    *d = (w == 0 ? 0.f : *d) + accs[e][k] * iA2;
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* d = &gcomb[(32 * role + rowmap(r, h)) * 64 + 32 * role + q];
                *d = (w == 0 ? 0.f : *d) + gm[r] * GU;
            }
            if (h == 0) scomb[32 * role + q] = (w == 0 ? 0.f : scomb[32 * role + q]) + s2t * iA2;
        }
        __syncthreads();
    }
    for (int w = 0; w < 8; ++w) {
        if (wave == w) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* d = &gcomb[rowmap(r, h) * 64 + 32 + q];
                *d = (w == 0 ? 0.f : *d) + gx[r] * GU;
            }
        }
        __syncthreads();
    }
    // the workgroup's row as fp32 (its sums ARE fp32): half the bytes of the 42 MB of fp64 rows, out and back in (facl_reduce_rows_f32)
    float* row = reinterpret_cast<float*>(part) + (size_t)blockIdx.x * W3_V;
    for (int i = threadIdx.x; i < 256 * 64; i += 512) row[i] = comb[(i >> 6) * 65 + (i & 63)];
    for (int i = threadIdx.x; i < 64 * 64; i += 512) {
        const int a = i >> 6, b = i & 63;
        row[256 * 64 + i] = (a >= 32 && b < 32) ? gcomb[b * 64 + a] : gcomb[i];               // lower-left = upper-right^T
    }
    if (threadIdx.x < 64) row[256 * 64 + 64 * 64 + threadIdx.x] = scomb[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// bwd2.  Output row per wave: [ dW2 (64 x 64, [c2][c1]) | R1 (8 x 64, rows: x_0..x_{D-1}, 1, 0..) ] = 4608 doubles
constexpr int B2_V = 64 * 64 + 8 * 64;

template <int D>
__global__ __launch_bounds__(256) void k_sa_bwd2(const float* __restrict__ dz2f, const float* __restrict__ y2f,
                                                 const float* __restrict__ x, int nunits,
                                                 const float* __restrict__ bw2 /* (4,64): scale2, A, B, mean2 */,
                                                 const float* __restrict__ W2, const float* __restrict__ l1tab_g,
                                                 double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    float4* w2t = lds4;                                    // A frags of W2^T: [rt'][rt][r4][lane] (1024 float4)
    float4* l1tab = lds4 + 1024;                           // 128 float4
    float4* tab = l1tab + 128;                             // 4 x 16 float4
    float* Tbase = reinterpret_cast<float*>(tab + 64);
    float* T = Tbase + (threadIdx.x >> 6) * (2 * 64 * TQ + 64 * 8);   // tile A: dy2, later dz1
    float* T2 = T + 64 * TQ;                               // tile B: a1
    float* xs = T2 + 64 * TQ;                              // [64 pos][8]: x_0..x_{D-1}, 1, 0...
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int ln = i & 63, r4 = (i >> 6) & 3, rt = (i >> 8) & 1, rto = i >> 9;
        const int c1 = 32 * rto + (ln & 31), c2 = 32 * rt + 8 * r4 + 4 * (ln >> 5);
        w2t[i] = make_float4(W2[(c2 + 0) * 64 + c1], W2[(c2 + 1) * 64 + c1], W2[(c2 + 2) * 64 + c1], W2[(c2 + 3) * 64 + c1]);
    }
    if (threadIdx.x < 128) l1tab[threadIdx.x] = reinterpret_cast<const float4*>(l1tab_g)[threadIdx.x];
    if (threadIdx.x < 64) tab[threadIdx.x] = reinterpret_cast<const float4*>(bw2)[threadIdx.x];
    __syncthreads();
    const float4* sc2 = tab; const float4* cA = tab + 16; const float4* cB = tab + 32; const float4* mean2 = tab + 48;

    const int lane = lane_id(), h = lane >> 5, q = lane & 31;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x16 dw2[2][2];
    f32x4v r1[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) dw2[0][0][r] = dw2[0][1][r] = dw2[1][0][r] = dw2[1][1][r] = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) r1[jt][r] = 0.f;

    for (int u = wave_g; u < nunits; u += nwaves) {
        asm volatile("" ::: "memory");
        const float* zt = dz2f + (size_t)u * FACL_UNIT_ELEMS;
        const float* yt = y2f + (size_t)u * FACL_UNIT_ELEMS;
        // x tile of the unit -> LDS (position-major, padded to 8) and this lane's two positions in registers
        float xv[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const size_t p = (size_t)u * 64 + 32 * ct + q;
            if (D == 4) {
                const float4 t = *reinterpret_cast<const float4*>(x + p * 4);
                xv[ct][0] = t.x; xv[ct][1] = t.y; xv[ct][2] = t.z; xv[ct][3] = t.w;
            } else {
                xv[ct][0] = x[p * 3]; xv[ct][1] = x[p * 3 + 1]; xv[ct][2] = x[p * 3 + 2]; xv[ct][3] = 0.f;
            }
            if (h == 0) {
                float4 lo = make_float4(xv[ct][0], xv[ct][1], xv[ct][2], D == 4 ? xv[ct][3] : 1.f);
                float4 hi = make_float4(D == 4 ? 1.f : 0.f, 0.f, 0.f, 0.f);
                reinterpret_cast<float4*>(xs)[(32 * ct + q) * 2] = lo;
                reinterpret_cast<float4*>(xs)[(32 * ct + q) * 2 + 1] = hi;
            }
        }
        // dy2 (lane = position, register = c2) and its copy in T for the transposed read-back
        float dy2[2][2][16];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const size_t o = (((ct * 2 + rt) * 4 + r4) * 64 + lane) * 4;
                    const float4 z = *reinterpret_cast<const float4*>(zt + o);
                    const float4 y = *reinterpret_cast<const float4*>(yt + o);
                    const int ti = 8 * rt + 2 * r4 + h;
                    const float4 s = sc2[ti], a = cA[ti], b = cB[ti], m = mean2[ti];
                    const float d0 = fmaf(s.x, z.x, fmaf(b.x, y.x - m.x, a.x));
                    const float d1 = fmaf(s.y, z.y, fmaf(b.y, y.y - m.y, a.y));
                    const float d2 = fmaf(s.z, z.z, fmaf(b.z, y.z - m.z, a.z));
                    const float d3 = fmaf(s.w, z.w, fmaf(b.w, y.w - m.w, a.w));
                    dy2[ct][rt][4 * r4] = d0; dy2[ct][rt][4 * r4 + 1] = d1; dy2[ct][rt][4 * r4 + 2] = d2; dy2[ct][rt][4 * r4 + 3] = d3;
                    float* dst = &T[(32 * ct + q) * TQ + 32 * rt + 8 * r4 + 4 * h];
                    dst[0] = d0; dst[1] = d1; dst[2] = d2; dst[3] = d3;
                }
        WAVE_LDS_FENCE();
        // da1^T[c1][p] = sum_c2 W2[c2][c1] dy2[p][c2]
        f32x16 da1[2][2];
#pragma unroll
        for (int r = 0; r < 16; ++r) da1[0][0][r] = da1[0][1][r] = da1[1][0][r] = da1[1][1][r] = 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 f0 = w2t[((0 * 2 + rt) * 4 + r4) * 64 + lane], f1 = w2t[((1 * 2 + rt) * 4 + r4) * 64 + lane];
                const float fa0[4] = {f0.x, f0.y, f0.z, f0.w}, fa1[4] = {f1.x, f1.y, f1.z, f1.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    da1[0][0] = MFMA32(fa0[e], dy2[0][rt][4 * r4 + e], da1[0][0]);
                    da1[0][1] = MFMA32(fa0[e], dy2[1][rt][4 * r4 + e], da1[0][1]);
                    da1[1][0] = MFMA32(fa1[e], dy2[0][rt][4 * r4 + e], da1[1][0]);
                    da1[1][1] = MFMA32(fa1[e], dy2[1][rt][4 * r4 + e], da1[1][1]);
                }
            }
        // a1 (same lane = position layout as da1), dz1 = da1 * [a1 > 0]; a1 -> T
        float dz1[2][2][16];
#pragma unroll
        for (int ro = 0; ro < 2; ++ro)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c1 = 32 * ro + rowmap(r, h);
                const float4 w = l1tab[c1 * 2];
                const float b = l1tab[c1 * 2 + 1].x;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    float v = fmaf(w.x, xv[ct][0], b);
                    v = fmaf(w.y, xv[ct][1], v);
                    v = fmaf(w.z, xv[ct][2], v);
                    if (D == 4) v = fmaf(w.w, xv[ct][3], v);
                    const float a1 = fmaxf(v, 0.f);
                    dz1[ro][ct][r] = v > 0.f ? da1[ro][ct][r] : 0.f;
                    T2[(32 * ct + q) * TQ + c1] = a1;
                }
            }
        WAVE_LDS_FENCE();
        // dW2[c2][c1] += sum_p dy2[p][c2] a1[p][c1]; operands read back in (lane = channel, k = position 32h+s)
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float dy0 = T[(32 * h + s) * TQ + q], dy1 = T[(32 * h + s) * TQ + 32 + q];
            const float aa0 = T2[(32 * h + s) * TQ + q], aa1 = T2[(32 * h + s) * TQ + 32 + q];
            dw2[0][0] = MFMA32(dy0, aa0, dw2[0][0]);
            dw2[0][1] = MFMA32(dy0, aa1, dw2[0][1]);
            dw2[1][0] = MFMA32(dy1, aa0, dw2[1][0]);
            dw2[1][1] = MFMA32(dy1, aa1, dw2[1][1]);
        }
        WAVE_LDS_FENCE();
        // dz1 -> T -> (lane = c1, register = position) ; R1[d][c1] += sum_p xe[p][d] dz1[p][c1]
#pragma unroll
        for (int ro = 0; ro < 2; ++ro)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) T[(32 * ct + q) * TQ + 32 * ro + rowmap(r, h)] = dz1[ro][ct][r];
        WAVE_LDS_FENCE();
        // R1 has only 8 useful rows: 16x16x4 tiles (rows = x_0..x_{D-1}, 1, zeros; 4 column tiles of 16 channels; 4 positions
        // per instruction) do it in 64 x 32 cycles instead of 64 x 64 with 32-row tiles.
        // lane l: A[row l&15][k = l>>4], B[k = l>>4][col l&15]; result col = l&15, row = 4*(l>>4) + reg
        {
            const int l15 = lane & 15, kq = lane >> 4;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int p = 4 * s + kq;
                const float xa = l15 < 8 ? xs[p * 8 + l15] : 0.f;
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
                    r1[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, T[p * TQ + 16 * jt + l15], r1[jt], 0, 0, 0);
            }
        }
        WAVE_LDS_FENCE();      // next unit overwrites T / xs
    }
    double* row = part + (size_t)wave_g * B2_V;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) row[(32 * a + rowmap(r, h)) * 64 + 32 * b + q] = (double)dw2[a][b][r];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 4 * (lane >> 4) + r;
            if (d < 8) row[64 * 64 + d * 64 + 16 * jt + (lane & 15)] = (double)r1[jt][r];
        }
}


}  // namespace

extern "C" int facl_sa_bwd0(const float* dpooled, const float* ymax, int64_t rows, const float* bnc3, float* coef,
                            double* sums, void* ws, void* stream) {
    if (!dpooled || !ymax || !bnc3 || !coef || !sums || !ws) return FACL_E_NULL;
    if (rows < 1 || rows > 0x7fffffff) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)(rows < 1024 ? rows : 1024);
    hipLaunchKernelGGL(k_sa_bwd0, dim3(grid), dim3(256), 0, st, dpooled, ymax, (int)rows, bnc3, coef, (double*)ws);
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, grid, 512, sums, st);
}

extern "C" int facl_sa_bwd1(const float* y2f, int64_t nunits, const float* bnc2, const float* G3, const float* h3,
                            const float* W3, const float* coef, const uint8_t* arg, float* dz2f, double* sums,
                            void* ws, const uint32_t* a2amax, void* stream) {
    if (!y2f || !bnc2 || !G3 || !h3 || !W3 || !coef || !arg || !dz2f || !sums || !ws || !a2amax) return FACL_E_NULL;
    if (nunits < 1 || nunits > 0x7fffffff) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)(nunits < SA_GRID * 4 ? (nunits + 3) / 4 : SA_GRID);
    const size_t lds = (1024 + 4096 + 80) * sizeof(float4) + 4 * 64 * TP * sizeof(float);
    static bool attr_done[64] = {};
    const void* fns[1] = {(const void*)k_sa_bwd1};
    if (int rc = facl_set_dynamic_lds(attr_done, fns, 1, (int)lds)) return rc;
    hipLaunchKernelGGL(k_sa_bwd1, dim3(grid), dim3(512), lds, st, y2f, (int)nunits, bnc2, G3, h3, W3, coef, arg, dz2f,
                       (double*)ws, a2amax);
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, grid * 4, 128, sums, st);
}

extern "C" int facl_sa_bwd_w3(const float* y2f, int64_t nunits, const float* bnc2, const float* coef,
                              const uint8_t* arg, double* out /* 20544 */, void* ws, const uint32_t* a2amax, void* stream) {
    if (!y2f || !bnc2 || !coef || !arg || !out || !ws || !a2amax) return FACL_E_NULL;
    if (nunits < 1 || nunits > 0x7fffffff) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)(nunits < SA_GRID * 4 ? (nunits + 3) / 4 : SA_GRID);
    // round 4 (late): two waves per unit (k_sa_bwd_w3p) unless FACL_BWD_W3_PAIR=0 selects the one-wave-per-unit kernel (A/B)
    static const int pairk = getenv("FACL_BWD_W3_PAIR") ? atoi(getenv("FACL_BWD_W3_PAIR")) : 1;
    if (pairk) {
        const size_t tiles = (size_t)8 * 64 * TP, combs = (size_t)256 * 65 + 64 * 64 + 64;
        const size_t lds = 32 * sizeof(float4) + (tiles > combs ? tiles : combs) * sizeof(float);
        static bool attr_done[64] = {};
        const void* fns[1] = {(const void*)k_sa_bwd_w3p};
        if (int rc = facl_set_dynamic_lds(attr_done, fns, 1, (int)lds)) return rc;
        hipLaunchKernelGGL(k_sa_bwd_w3p, dim3(grid), dim3(512), lds, st, y2f, (int)nunits, bnc2, coef, arg, (double*)ws, a2amax);
        int rc = facl_launch_status();
        if (rc) return rc;
        return facl_reduce_rows_f32((const float*)ws, grid, W3_V, out, st);
    } else {
        const size_t lds = 32 * sizeof(float4) + (256 * 65 + 64 * 64 + 64 + 4 * 64 * TP) * sizeof(float);
        static bool attr_done[64] = {};
        const void* fns[1] = {(const void*)k_sa_bwd_w3};
        if (int rc = facl_set_dynamic_lds(attr_done, fns, 1, (int)lds)) return rc;
        hipLaunchKernelGGL(k_sa_bwd_w3, dim3(grid), dim3(256), lds, st, y2f, (int)nunits, bnc2, coef, arg, (double*)ws, a2amax);
    }
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, grid, W3_V, out, st);
}

int facl_sa_bwd2_sb_launch(const float* dz2f, const float* y2f, const float* x, int nunits, int D, const float* bw2,
                           const float* W2, const float* l1tab, double* ws, int grid, const uint32_t* a1amax, hipStream_t st);   // sa_bwd2.hip

extern "C" int facl_sa_bwd2(const float* dz2f, const float* y2f, const float* x, int64_t nunits, int D,
                            const float* bw2, const float* W2, const float* l1tab, double* out /* 4608 */, void* ws,
                            const uint32_t* a1amax, void* stream) {
    if (!dz2f || !y2f || !x || !bw2 || !W2 || !l1tab || !out || !ws || !a1amax) return FACL_E_NULL;
    if ((D != 3 && D != 4) || nunits < 1 || nunits > 0x7fffffff) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    // FACL_BWD2_F32=1 selects the exact-fp32 MFMA kernel (v_mfma_f32_32x32x2_f32) instead of the split-bf16 one
    static const int use_f32 = getenv("FACL_BWD2_F32") ? atoi(getenv("FACL_BWD2_F32")) : 0;
    if (!use_f32) {
        const int grid = (int)(nunits < SA_GRID * 8 ? (nunits + 3) / 4 : 2 * SA_GRID);       // 2 workgroups of 4 waves per CU
        int rc = facl_sa_bwd2_sb_launch(dz2f, y2f, x, (int)nunits, D, bw2, W2, l1tab, (double*)ws, grid, a1amax, st);
        if (rc) return rc;
        return facl_reduce_rows((const double*)ws, grid, B2_V, out, st);       // one row per workgroup (combined in LDS)
    }
    const int grid = (int)(nunits < SA_GRID * 4 ? (nunits + 3) / 4 : SA_GRID);
    const size_t lds = (1024 + 128 + 64) * sizeof(float4) + 4 * (2 * 64 * TQ + 64 * 8) * sizeof(float);
    static bool attr_done[64] = {};
    const void* fns[2] = {(const void*)k_sa_bwd2<4>, (const void*)k_sa_bwd2<3>};
    if (int rc = facl_set_dynamic_lds(attr_done, fns, 2, (int)lds)) return rc;
    if (D == 4) hipLaunchKernelGGL((k_sa_bwd2<4>), dim3(grid), dim3(256), lds, st, dz2f, y2f, x, (int)nunits, bw2, W2, l1tab, (double*)ws);
    else hipLaunchKernelGGL((k_sa_bwd2<3>), dim3(grid), dim3(256), lds, st, dz2f, y2f, x, (int)nunits, bw2, W2, l1tab, (double*)ws);
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, grid * 4, B2_V, out, st);
}
