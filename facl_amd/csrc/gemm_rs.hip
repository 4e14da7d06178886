// Row-streamed split-bf16 GEMM for the per-centroid layers (net3DV_3, cn3d_model_conbag.py:61-77: 49,152 rows x
// {256,512,1024} channels) -- forward  y = f(a) W^T + b [+ centres Wc^T]  and dgrad  da = dy W.
//
// Why a second GEMM shape.  k_gemm_sb (gemm.hip) stages BOTH operands through LDS as three bf16 planes per 32-deep stage:
// per MFMA it pays ~0.5 ds_read_b128, ~250 B of LDS fill and ~3.7 VALU instructions (split of both operands), and its
// PMC profile says so (MFMA busy 0.39-0.46, issue-stalled 0.44-0.53).  Here the two operands are treated by what they
// are:
//   * the WEIGHTS are constant during a step: they are split ONCE (k_rs_planes) into fragment-ordered bf16 planes
//     [k-step][32-column tile][plane][lane] x 16 B, and a workgroup streams the planes of its 256 output columns through a
//     3-slot LDS ring with LDS-DMA (global_load_lds_dwordx4: no registers, no VALU, no ds_write);
//   * the ACTIVATIONS are private to the wave that owns their 32 rows: each wave DMAs its own 32 x 32 fp32 stage into a
//     private LDS slot (coalesced 128-B rows, XOR swizzle on the SOURCE address), reads its A fragments (8 consecutive
//     k of its row per lane) with two ds_read_b128 per k-step, applies the previous layer's BatchNorm + ReLU on the fly
//     (the PRO slot: k_rows_bn_relu and its activation round trip disappear) and splits them into bf16 planes IN REGISTERS.
// One wave = one 32-row tile x 256 columns (8 accumulator tiles, 128 VGPRs), 8 waves per workgroup, one workgroup per
// CU.  Per MFMA: 0.5 ds_read_b128, 64 B of LDS fill, ~1.3 VALU.  Same arithmetic as k_gemm_sb (six bf16 products per
// multiply-add, smallest first, 16-deep k-steps in the same order): results are bit-identical to it.
// Epilogue (lane = column, registers = rows): bias, per-column (sum, sumsq) for the BatchNorm that follows (8 waves
// combined in LDS in fixed order, one fp64 partial row per workgroup), optional my_max_pool over each cloud's 64 rows
// (two waves per cloud, merged through LDS, first row wins ties), 128-B row segments stored straight from registers.
// Roofline: bf16 MFMA (2.5 PFLOP/s dense; 6 executed FLOPs per algorithmic one).
#include "common.h"
#include <stdlib.h>

int facl_reduce_rows(const double* part, int rows, int V, double* out, hipStream_t st);
extern "C" int64_t facl_ws_bytes(void);

namespace {

constexpr int RS_CT = 8;                          // 32-column tiles per workgroup (256 output columns)
constexpr int RS_WAVES = 4;                       // one 32-row tile per wave (128 rows per workgroup), two workgroups per CU
constexpr int RS_WPP = RS_CT * 3 / RS_WAVES;      // 1-KiB weight pieces a wave issues per k-step
constexpr int RS_WSLOT = RS_CT * 3 * 1024;        // bytes of one k-step (16 k) of planes for 256 columns: 24 KiB
constexpr int RS_ASLOT = 32 * 128;                // bytes of one stage (32 k) of a wave's 32 rows, fp32: 4 KiB
constexpr int RS_LDS_W = 2 * RS_WSLOT;            // 48 KiB: k-step j in slot j & 1
constexpr int RS_LDS_A = RS_WAVES * RS_ASLOT;     // 16 KiB (one stage slot per wave)
constexpr int RS_KMAX = 1024;                     // longest contraction
#ifndef RS_PREP_IN_ODD
#define RS_PREP_IN_ODD 1                           // next stage's fragment read + first planes inside the odd k-step's MFMA stream (0: behind the barrier)
#endif
#ifndef RS_H3_AHEAD2
#define RS_H3_AHEAD2 0                             // 1: fp16x3 kernels request their B fragments TWO MFMA groups ahead (round 4: measured equal / slower)
#endif
constexpr int RS_KPRO = 512;                      // longest contraction WITH a prologue: tables scale[K] | shift[K], 4 KiB
constexpr int RS_LDS = RS_LDS_W + RS_LDS_A + 2 * RS_KPRO * 4;     // 68 KiB

struct RsArgs {
    const float* A; int lda; int M; int K;        // K % 32 == 0
    const uint4* Wp; int NT;                      // planes [k-step][NT column tiles][3][64]; NT = N / 32
    int N;                                        // output columns, N % 256 == 0
    const float* bias;                            // (N) or null
    const float* pscale; const float* pshift;     // (K) prologue relu(scale*a + shift), or null
    const float* centers;                         // (M,3) or null: k-step K/16 of Wp holds the centre columns
    float* C; int ldc;
    double* part;                                 // statistics partial rows [gridDim.y][2N], or null
    const float* sgn; float* smax; int* sarg;     // my_max_pool over blocks of 64 rows: (M/64, N), or null
    // dgrad only: the output IS dL/da of a layer a = relu(bn(y)); with `by` (M,N) = that layer's raw output and `bbnc`
    // (>= 4 x N: mean | invstd | scale | shift) the partial rows hold the BatchNorm-backward sums of the column instead:
    // (sum_r d, sum_r d * yhat), d = C[r] where scale*y + shift > 0 else 0, yhat = (y - mean) * invstd
    const float* by; const float* bbnc;
    int h3;                                       // 1: fp16x3 planes / arithmetic, 0: bf16x6
    const unsigned* amax;                         // h3 only: bits of (a bound of) max|A| in FACL_AMAX_SLOTS slots -> the
                                                  // power-of-two scale of A (activations: their bound, gradients: max|dy|)
    const int* wse;                               // h3 only: biased exponent of the weight scale per 32-column tile (k_rs_planes)
    int phase_ticks;                              // experiment (FACL_RS_PHASE): first-round workgroups in the CU's second slot start this many 100-MHz ticks late
    int first_round;                              // number of workgroups resident at launch (2 per CU)
};



// ---- weights -> fragment-ordered bf16 planes -------------------------------------------------------------------------
// value(o, c) = W[o*so + c*sc]: o = output channel (lane of the B operand), c = contraction index.  forward: so = ldw,
// sc = 1 (y = a W^T); dgrad: so = 1, sc = ldw (da = dy W).  Entry ((ks*NT + o/32)*3 + plane)*64 + 32h + o%32 holds the 8
// k-slots c = 16ks + 8h + j of column o.  `xc` (forward only): one more k-step whose slots 0..2 (h = 0) are
// Wc[o][0..2] -- the centroid-xyz columns of torch.cat((yt, xt), 1), cn3d_model_conbag.py:219.
// `half`: fp16x3 planes (common.h): TWO planes per fragment (h1, h2 of w * scale) instead of three bf16 planes; same indexing
// with 2 in place of 3.  The scale is a power of two PER 32-COLUMN TILE (output columns are independent, so the epilogue
// can undo it per tile): 2^(13 - floor(log2 max|w|)) over the tile's 32 columns x the whole contraction (centre columns
// included), stored as its biased exponent in `hdr[tile]` behind the planes.  One workgroup per (matrix, column tile): a
// first pass over the tile takes the maximum, a second one splits (the tile is 4-128 KiB: L2 hits).
constexpr int RS_PL_ITEMS = 16;                   // items (8 weights each) a thread holds: at most 64 k-steps (contraction <= 1024)
__device__ __forceinline__ void rs_planes_tile(const float* __restrict__ W, long long so, long long sc, int NO, int NC,
                                               const float* __restrict__ xc, int ldxc, uint4* __restrict__ out, int half,
                                               int* __restrict__ hdr, int ot) {
    __shared__ float red[16];
    const int NT = NO >> 5, nks = NC >> 4, nitems = (nks + (xc ? 1 : 0)) * 64;
    // ONE pass over the tile: every thread keeps its items in registers (all loads in flight at once; the first version read
    // the tile twice, one dependent item at a time: 21 us per step instead of 7), takes the maximum, then splits from registers
    float v[RS_PL_ITEMS][8];
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < RS_PL_ITEMS; ++t) {
        const int item = threadIdx.x + 256 * t;
        const int ln = item & 63, ks = item >> 6;
        const int o = 32 * ot + (ln & 31), hh = ln >> 5;
        if (item < nitems) {
            if (ks < nks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[t][j] = W[(long long)o * so + (long long)(16 * ks + 8 * hh + j) * sc];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[t][j] = (hh == 0 && j < 3) ? xc[(long long)o * ldxc + j] : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[t][j] = 0.f;
        }
    }
    float sw = 1.f;
    if (half) {
#pragma unroll
        for (int t = 0; t < RS_PL_ITEMS; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[t][j]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const int se = h3_se(__float_as_uint(m));
        if (threadIdx.x == 0) hdr[ot] = se;
        sw = pow2_biased(se);
    }
#pragma unroll
    for (int t = 0; t < RS_PL_ITEMS; ++t) {
        const int item = threadIdx.x + 256 * t;
        if (item >= nitems) break;
        const int ln = item & 63, ks = item >> 6;
        unsigned hi[4], mi[4], lo[4];
        if (half) {
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pair_h(v[t][2 * j] * sw, v[t][2 * j + 1] * sw, hi[j], lo[j]);
            uint4* d = out + ((long long)(ks * NT + ot) * 2) * 64 + ln;
            d[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            d[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair(v[t][2 * j], v[t][2 * j + 1], hi[j], mi[j], lo[j]);
        uint4* d = out + ((long long)(ks * NT + ot) * 3) * 64 + ln;
        d[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        d[64] = make_uint4(mi[0], mi[1], mi[2], mi[3]);
        d[128] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// max|x| of a small array RAISED into the slots of an amax buffer by the extra workgroups of the planes launch (the centroid
// coordinates share the scale of the first layer's row operand: no launch of their own)
__device__ __forceinline__ void rs_absmax_block(const float* __restrict__ x, long long n, unsigned* __restrict__ amax, int b, int nb) {
    float m = 0.f;
    for (long long i = (long long)b * 256 + threadIdx.x; i < n; i += (long long)nb * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0)
        atomicMax(amax + ((b * 4 + (threadIdx.x >> 6)) & (FACL_AMAX_SLOTS - 1)) * FACL_AMAX_STRIDE, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void k_rs_planes(const float* __restrict__ W, long long so, long long sc, int NO, int NC,
                                                   const float* __restrict__ xc, int ldxc, uint4* __restrict__ out, int half,
                                                   int* __restrict__ hdr) {
    rs_planes_tile(W, so, sc, NO, NC, xc, ldxc, out, half, hdr, blockIdx.x);
}

// several weight matrices in one launch (the forward and dgrad planes of the three layers of net3DV_3: one ~8 us launch
// per step instead of six); block b serves column tile b - first[j] of matrix j
constexpr int RS_MAXJOBS = 8;
struct RsPlaneJobs {
    int n;
    const float* W[RS_MAXJOBS]; long long so[RS_MAXJOBS]; long long sc[RS_MAXJOBS]; int NO[RS_MAXJOBS]; int NC[RS_MAXJOBS];
    const float* xc[RS_MAXJOBS]; int ldxc[RS_MAXJOBS]; uint4* out[RS_MAXJOBS]; int* hdr[RS_MAXJOBS]; int first[RS_MAXJOBS + 1];
    int half[RS_MAXJOBS];
    const float* ax; long long an; unsigned* aamax; int anb;      // optional absmax job: the last `anb` workgroups
};
__global__ __launch_bounds__(256) void k_rs_planes_multi(RsPlaneJobs jb) {
    if ((int)blockIdx.x >= jb.first[jb.n]) {
        rs_absmax_block(jb.ax, jb.an, jb.aamax, (int)blockIdx.x - jb.first[jb.n], jb.anb);
        return;
    }
    int j = 0;
#pragma unroll
    for (int t = 1; t < RS_MAXJOBS; ++t) j += (t < jb.n && (int)blockIdx.x >= jb.first[t]) ? 1 : 0;
    rs_planes_tile(jb.W[j], jb.so[j], jb.sc[j], jb.NO[j], jb.NC[j], jb.xc[j], jb.ldxc[j], jb.out[j], jb.half[j], jb.hdr[j],
                   (int)blockIdx.x - jb.first[j]);
}

struct RsTile { int x, y; };
// XCD-aware order (as gemm.hip: xcd_tile): each XCD walks a contiguous range of (row group, column block) tiles,
// column blocks fastest, so the column blocks that re-read one 256-row panel of A hit the same L2.
__device__ __forceinline__ RsTile rs_tile() {
    const int nbx = gridDim.x, total = nbx * gridDim.y;
    const int b = blockIdx.x + nbx * blockIdx.y;
    const int per = total >> 3, rem = total & 7, xcd = b & 7, slot = b >> 3;
    const int L = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    RsTile t;
    t.x = L % nbx; t.y = L / nbx;
    return t;
}

// One LDS-DMA piece: 64 lanes x 16 B from each lane's global address `gsrc` to LDS bytes [lds_dst, lds_dst + 1024).
// Issued from inline asm on purpose: hipcc counts a __builtin_amdgcn_global_load_lds as a pending LDS write and puts
// `s_waitcnt vmcnt(0)` in front of later ds_reads that may alias it -- inside this pipeline that drains the pieces issued a
// moment ago (seen in the ISA of the builtin form).  Here every wait is counted by hand (see the pipeline comment).
// M0 (the DMA's LDS base) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void rs_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

// WV = waves per workgroup: 4 (128 rows, two workgroups per CU: the default) or 8 (256 rows, one workgroup per CU: every k-step
// of weight planes serves twice the rows; opt-in experiment, measured equal: rs_waves below)
template <bool PRO, bool SEG, bool BST, bool H3 = false, int WV = RS_WAVES>
__global__ __launch_bounds__(64 * WV, WV == 8 ? 1 : 2) void k_gemm_rs(RsArgs g) {
    constexpr int NPL = H3 ? 2 : 3;                                     // planes per fragment (fp16x3 / bf16x6, common.h)
    constexpr int WPP = RS_CT * NPL / WV;                         // 1-KiB weight pieces a wave issues per k-step
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // The two workgroups of a CU start together and take the same time, so they stay in phase for the whole launch -- both in
    // the main loop (MFMA pipe shared) or both draining their output (pipe idle).  On multi-round launches the first-round
    // workgroup in the CU's second slot waits about half a workgroup's life once (common.h: facl_phase_wait); the offset
    // then persists round after round.  (The persistent set-abstraction kernels do not need it: k_sa_bwd2_sb with the same
    // offset measured unchanged, gpurun_out/r5v_ab.log.)
    if ((int)(blockIdx.x + gridDim.x * blockIdx.y) < g.first_round) facl_phase_wait(g.phase_ticks);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, q = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const wring = lds;
    char* const abuf = lds + RS_LDS_W + wave * RS_ASLOT;
    float* const tab = reinterpret_cast<float*>(lds + RS_LDS_W + WV * RS_ASLOT);
    const unsigned wring_a = __builtin_amdgcn_readfirstlane(lds_addr(wring));
    const unsigned abuf_a = __builtin_amdgcn_readfirstlane(lds_addr(abuf));
    const RsTile tile = rs_tile();
    const int cb = tile.x;
    const int row0 = (tile.y * WV + wave) * 32;
    const int nks = g.K >> 4, nst = g.K >> 5;
    const int nks_all = nks + (g.centers ? 1 : 0);
    int seA = 127;                                                      // fp16x3: biased exponent of the A operand's scale
    if (H3) seA = h3_se_wide_of(g.amax);                                // the wide clamp: A is a gradient in the dgrad form
    const float sA = pow2_biased(seA);

    // (PRO: the prologue tables are filled BEHIND the first DMAs -- one cold memory round trip instead of two)

    // ---- DMA issue helpers (wave-uniform control flow; VM-counter bookkeeping in the pipeline comment below)
    // this wave's share of the k-step's column tiles: WPP consecutive 1-KiB pieces
    const uint4* wsrc = g.Wp + ((size_t)(RS_CT * cb) * NPL + wave * WPP) * 64 + lane;
    const size_t wstep = (size_t)g.NT * NPL * 64;                                      // uint4 per k-step
    auto issueW = [&](int j) {
        if (j < nks_all) {
            const unsigned dst = wring_a + (j & 1) * RS_WSLOT + wave * WPP * 1024;
            const uint4* src = wsrc + (size_t)j * wstep;
#pragma unroll
            for (int p = 0; p < WPP; ++p) rs_dma16(src + p * 64, dst + p * 1024);
        }
    };
    // activations: piece i = rows 8i + (lane>>3); the 16-B chunk that lands in physical chunk c' of row r is logical
    // chunk c' ^ ((r>>1)&7) (conflict-free ds_read_b128 of one logical chunk over the 32 rows, cf. k_gemm_dma)
    const float* asrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3);
        int gr = row0 + r;
        gr = gr < g.M ? gr : g.M - 1;                                  // rows past M re-read the last row (never stored / counted)
        asrc[i] = g.A + (size_t)gr * g.lda + 4 * ((lane & 7) ^ ((r >> 1) & 7));
    }
    auto issueA = [&](int s) {                                         // 4 pieces of 1 KiB
        if (s < nst) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rs_dma16(asrc[i] + 32 * s, abuf_a + i * 1024);
        }
    };
    // fragment reads of a whole stage: lane (row q, half h) takes the 8 k = 16 t + 8h + 0..7 of its row for both k-steps
    // t = 0, 1 of the stage: logical chunks 4t + 2h, 4t + 2h + 1
    const int akey = (q >> 1) & 7;
    auto read_stage = [&](float4 (&r)[4]) {
        const char* base = abuf + q * 128;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int c = 4 * t + 2 * h;
            r[2 * t] = *reinterpret_cast<const float4*>(base + ((c ^ akey) << 4));
            r[2 * t + 1] = *reinterpret_cast<const float4*>(base + (((c + 1) ^ akey) << 4));
        }
    };
    // planes of k-step j as 12 packed registers pk[4*plane + e] (e = pair of k-slots); `half` = which float4 of the two
    // (k-slots 0..3 / 4..7): the split runs in two instalments, each inside the shadow of one group of six MFMAs
    auto planes_half = [&](int j, const float4& r, unsigned (&pk)[12], int half) {
        float v[4] = {r.x, r.y, r.z, r.w};
        if (PRO) {
            const float4 s4 = *reinterpret_cast<const float4*>(tab + 16 * j + 8 * h + 4 * half);
            const float4 t4 = *reinterpret_cast<const float4*>(tab + RS_KPRO + 16 * j + 8 * h + 4 * half);
            // NaN-propagating ReLU (torch.relu keeps NaN; fmaxf(NaN, 0) = 0 would turn a poisoned input into a clean zero)
            v[0] = relu_nan(fmaf(s4.x, v[0], t4.x)); v[1] = relu_nan(fmaf(s4.y, v[1], t4.y));
            v[2] = relu_nan(fmaf(s4.z, v[2], t4.z)); v[3] = relu_nan(fmaf(s4.w, v[3], t4.w));
        }
        if (H3) {
            split_pair_h(v[0] * sA, v[1] * sA, pk[2 * half], pk[4 + 2 * half]);
            split_pair_h(v[2] * sA, v[3] * sA, pk[2 * half + 1], pk[4 + 2 * half + 1]);
        } else {
            split_pair(v[0], v[1], pk[2 * half], pk[4 + 2 * half], pk[8 + 2 * half]);
            split_pair(v[2], v[3], pk[2 * half + 1], pk[4 + 2 * half + 1], pk[8 + 2 * half + 1]);
        }
    };
    auto frag = [&](const unsigned (&pk)[12], int p) { return as_bf16x8(pk[4 * p], pk[4 * p + 1], pk[4 * p + 2], pk[4 * p + 3]); };

    f32x16 acc[RS_CT];
#pragma unroll
    for (int ct = 0; ct < RS_CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    constexpr int PA[6] = FACL_SB_PA, PB[6] = FACL_SB_PB;
    // MFMAs of one k-step.  The B fragments of a column tile are requested one whole group (six MFMAs, 192 cycles) before
    // their first use; sched_barriers pin that order (left alone, the scheduler sinks each read to just above its consumer
    // and every group then waits out the LDS latency) and keep it from hoisting all 24 reads of the k-step (96 VGPRs) at
    // once.  `mid_a()` / `mid_b()` run inside the groups of column tiles 2 and 4 (the next k-step's plane split).
    constexpr int HA[3] = FACL_H3_PA, HB[3] = FACL_H3_PB;
    auto mfma_group = [&](f32x16& c, const bf16x8 (&P)[3], const bf16x8 (&b)[3]) {
        if (H3) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
                c = MFMA_F16(__builtin_bit_cast(f16x8h, P[HA[t]]), __builtin_bit_cast(f16x8h, b[HB[t]]), c);   // smallest terms first
        } else {
#pragma unroll
            for (int t = 0; t < 6; ++t) c = MFMA_BF16(P[PA[t]], b[PB[t]], c);                                    // smallest terms first
        }
    };
    auto mfma_step = [&](int j, const unsigned (&pk)[12], auto&& mid_a, auto&& mid_b) {
        const uint4* ws = reinterpret_cast<const uint4*>(wring + (j & 1) * RS_WSLOT) + lane;
        const bf16x8 P[3] = {frag(pk, 0), frag(pk, 1), frag(pk, H3 ? 1 : 2)};
        if (H3 && RS_H3_AHEAD2) {
            // (experiment, off: an fp16x3 group is THREE MFMAs (96 cycles), shorter than an LDS round trip under load; here the
            // fragments of a column tile are requested TWO groups ahead (three rotating register sets), as far ahead in cycles as
            // the six-MFMA groups of bf16x6 were with one.  Same box, bit-identical: forward 49152x512x1024 0.197 vs 0.198 ms,
            // dgrad 1024->512 0.201 vs 0.185: the B-fragment latency is not what these kernels wait for.  gpurun_out/r5m_ab.log)
            bf16x8 bq[3][3];
#pragma unroll
            for (int p = 0; p < NPL; ++p) { bq[0][p] = __builtin_bit_cast(bf16x8, ws[p * 64]); bq[1][p] = __builtin_bit_cast(bf16x8, ws[(NPL + p) * 64]); }
#pragma unroll
            for (int ct = 0; ct < RS_CT; ++ct) {
                if (ct + 2 < RS_CT) {
#pragma unroll
                    for (int p = 0; p < NPL; ++p) bq[(ct + 2) % 3][p] = __builtin_bit_cast(bf16x8, ws[((ct + 2) * NPL + p) * 64]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ct == 2) mid_a();
                if (ct == 4) mid_b();
                mfma_group(acc[ct], P, bq[ct % 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        bf16x8 b0[3], b1[3];
#pragma unroll
        for (int p = 0; p < NPL; ++p) b0[p] = __builtin_bit_cast(bf16x8, ws[p * 64]);
#pragma unroll
        for (int ct = 0; ct < RS_CT; ct += 2) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) b1[p] = __builtin_bit_cast(bf16x8, ws[((ct + 1) * NPL + p) * 64]);
            __builtin_amdgcn_sched_barrier(0);
            if (ct == 2) mid_a();
            if (ct == 4) mid_b();
            mfma_group(acc[ct], P, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (ct + 2 < RS_CT) {
#pragma unroll
                for (int p = 0; p < NPL; ++p) b0[p] = __builtin_bit_cast(bf16x8, ws[((ct + 2) * NPL + p) * 64]);
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma_group(acc[ct + 1], P, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto nop = []() {};

    // ---- pipeline.  Weight ring: k-step j in slot j & 1.  VM operations of a wave, in issue order: prologue A(0), W(0);
    // top of k-step j (behind the barrier): W(j+1) [RS_WPP pieces]; at even j = 2s, once the stage's fragments are in
    // registers: A(s+1) [4 pieces] into the wave's single activation slot.  At the top of k-step j the wave needs W(j),
    // issued at the top of j-1: at odd j only A((j+1)/2) came after it -> vmcnt(4); at even j nothing did -> vmcnt(0), which
    // also lands A(j/2), needed now.  The workgroup barrier behind the wait makes every wave's pieces of W(j) visible and
    // proves that all waves have left k-step j-1, whose slot the next issue overwrites.
    issueA(0); issueW(0);
    if (PRO) {
        for (int i = tid; i < g.K; i += 64 * WV) { tab[i] = g.pscale[i]; tab[RS_KPRO + i] = g.pshift[i]; }
    }
    unsigned P0[12], P1[12];
    float4 raw[4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                   // DMA pieces landed, tables written (LDS stores drained)
    read_stage(raw);
    planes_half(0, raw[0], P0, 0);
    planes_half(0, raw[1], P0, 1);
    issueW(1);
    for (int j = 0; j + 2 < nks; j += 2) {
        // -- even k-step j (W(j) and this stage's activations landed at the previous wait): MFMAs on P0, P1 from the
        //    stage's second half; the activation slot is free once `raw` is in registers -> next stage's DMA
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issueA((j >> 1) + 1);
        mfma_step(j, P0, [&]() { planes_half(j + 1, raw[2], P1, 0); }, [&]() { planes_half(j + 1, raw[3], P1, 1); });
        // -- odd k-step j+1
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issueW(j + 2);
        if (RS_PREP_IN_ODD) {
            // The next stage's activation fragments and first planes are prepared INSIDE this k-step's MFMA stream (its hooks were
            // empty): in flight here are A(j/2+1) [4 pieces, issued during the even k-step] and then W(j+2) [WPP pieces], so
            // vmcnt(WPP) means the activations have landed; the slot is this wave's own (no barrier), `raw` and P0 are dead since
            // the even k-step.  In-kernel stamps (s_memtime, scratch/exp_rs) had put this chain -- LDS round trip + 40 VALU behind
            // the barrier -- at 15 % of the loop.
            mfma_step(j + 1, P1,
                      [&]() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WPP) : "memory"); read_stage(raw); },
                      [&]() { planes_half(j + 2, raw[0], P0, 0); planes_half(j + 2, raw[1], P0, 1); });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issueW(j + 3);
            continue;
        }
        mfma_step(j + 1, P1, nop, nop);
        // -- top of the next stage: W(j+2) and A(j/2+1) have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issueW(j + 3);
        read_stage(raw);
        planes_half(j + 2, raw[0], P0, 0);
        planes_half(j + 2, raw[1], P0, 1);
    }
    {   // last stage (k-steps nks-2, nks-1): nothing more to stream but the centre k-step
        const int j = nks - 2;
        mfma_step(j, P0, [&]() { planes_half(j + 1, raw[2], P1, 0); }, [&]() { planes_half(j + 1, raw[3], P1, 1); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issueW(j + 2);
        mfma_step(j + 1, P1, nop, nop);
    }
    if (g.centers) {                                                   // the centroid-xyz columns: one more k-step
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (h == 0) {
            int gr = row0 + q;
            gr = gr < g.M ? gr : g.M - 1;
            c0 = g.centers[(size_t)gr * 3]; c1 = g.centers[(size_t)gr * 3 + 1]; c2 = g.centers[(size_t)gr * 3 + 2];
        }
        unsigned hi[2], mi[2] = {0u, 0u}, lo[2];
        if (H3) {
            split_pair_h(c0 * sA, c1 * sA, hi[0], mi[0]);
            split_pair_h(c2 * sA, 0.f, hi[1], mi[1]);
            lo[0] = lo[1] = 0u;
        } else {
            split_pair(c0, c1, hi[0], mi[0], lo[0]);
            split_pair(c2, 0.f, hi[1], mi[1], lo[1]);
        }
        const unsigned Pc[12] = {hi[0], hi[1], 0u, 0u, mi[0], mi[1], 0u, 0u, lo[0], lo[1], 0u, 0u};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mfma_step(nks, Pc, nop, nop);
    }

    // ---- epilogue: lane = column 256 cb + 32 ct + q, register r = row row0 + rowmap(r, h).
    // The tile leaves through a per-wave 32 x 32 staging image (the wave's activation slot, free now): 16 ds_write_b32 in the
    // accumulator layout, 4 ds_read_b128 in row-major order, 4 stores of 16 B per lane = whole 128-B lines.  Storing the
    // accumulator layout directly (16 dword stores per tile, 128 per wave) was store-ISSUE bound: 12 of the 22 us of a
    // one-wave-of-workgroups launch (ablation).  16-B chunks are XOR-swizzled by (row >> 1) & 7 (the key of the
    // activation slots): conflict-free for the b32 writes and the b128 reads.  The BatchNorm-backward variant brings
    // its y tile in the same way, reversed (4 wide loads -> image -> 16 ds_read_b32).
    // Statistics / max-pool hand-off areas live in the weight-ring slot that the final k-step does NOT read.
    float* const stg = reinterpret_cast<float*>(abuf);
    // (8-wave workgroups: 8 x 6 KiB do not fit a ring slot; they have an area of their own behind the prologue tables)
    char* const freeslot = WV == 8 ? lds + RS_LDS_W + WV * RS_ASLOT + 2 * RS_KPRO * 4 : wring + (((nks_all - 1) & 1) ^ 1) * RS_WSLOT;
    float* const wstat = reinterpret_cast<float*>(freeslot + wave * 6144);       // (sum, sumsq) per column: 2 KiB
    float* const wbest = wstat + 512;                                            // SEG, odd waves: (max, arg): 2 KiB
    float* const ystg = wstat + 512;                                             // BST: y tile image, 4 KiB
    const int srow = lane >> 3, schunk = lane & 7;                                // row-major side: 8 lanes per 128-B row
    // EVERY global load of the epilogue is issued before the first store (the per-column constants of all eight tiles here, the
    // BatchNorm-backward variant's y tile one tile ahead): a load behind a store waits for it -- vmcnt counts stores, in order --
    // and the per-tile `bias[n]` / `wse[tile]` loads used to put a full HBM store round trip between consecutive tiles: in-kernel
    // stamps (s_memtime, scratch/exp_rs) showed 17k cycles for the eight tiles of a workgroup, 20 % of its life.
    float biasv[RS_CT], sgv[RS_CT], unsv[RS_CT];
#pragma unroll
    for (int ct = 0; ct < RS_CT; ++ct) {
        const int n = 256 * cb + 32 * ct + q;
        biasv[ct] = g.bias ? g.bias[n] : 0.f;
        sgv[ct] = SEG ? sgn_of(g.sgn[n]) : 1.f;
        unsv[ct] = H3 ? h3_unscale(seA, g.wse[RS_CT * cb + ct]) : 1.f;             // this column tile's 1 / (sA sW): exact
    }
    // BST: the layer's raw output under a tile + its BatchNorm constants, fetched one tile ahead into NAMED registers (arrays
    // written through a helper ended up in scratch)
    struct YTile { float4 y0, y1, y2, y3; float mean, inv, sc, sh; };
    auto fetch_y = [&](int ct) -> YTile {
        YTile t;
        const int n_ = 256 * cb + 32 * ct + q;
        t.mean = g.bbnc[n_]; t.inv = g.bbnc[g.N + n_]; t.sc = g.bbnc[2 * g.N + n_]; t.sh = g.bbnc[3 * g.N + n_];
        const float* base = g.by + 256 * cb + 32 * ct + 4 * schunk;
        int r0_ = row0 + srow, r1_ = row0 + 8 + srow, r2_ = row0 + 16 + srow, r3_ = row0 + 24 + srow;
        r0_ = r0_ < g.M ? r0_ : g.M - 1; r1_ = r1_ < g.M ? r1_ : g.M - 1; r2_ = r2_ < g.M ? r2_ : g.M - 1; r3_ = r3_ < g.M ? r3_ : g.M - 1;
        t.y0 = *reinterpret_cast<const float4*>(base + (size_t)r0_ * g.N);
        t.y1 = *reinterpret_cast<const float4*>(base + (size_t)r1_ * g.N);
        t.y2 = *reinterpret_cast<const float4*>(base + (size_t)r2_ * g.N);
        t.y3 = *reinterpret_cast<const float4*>(base + (size_t)r3_ * g.N);
        return t;
    };
    YTile ycur = {}, ynxt = {};
    if (BST) ycur = fetch_y(0);
#pragma unroll
    for (int ct = 0; ct < RS_CT; ++ct) {
        const float bias = biasv[ct], sg = sgv[ct], uns = unsv[ct];
        float s = 0.f, sq = 0.f, best = 0.f;
        int bp = 0;
        float bmean = 0.f, binv = 0.f, bsc = 0.f, bsh = 0.f;
        if (BST) {
            bmean = ycur.mean; binv = ycur.inv; bsc = ycur.sc; bsh = ycur.sh;
            *reinterpret_cast<float4*>(ystg + (srow) * 32 + ((schunk ^ ((srow >> 1) & 7)) << 2)) = ycur.y0;
            *reinterpret_cast<float4*>(ystg + (8 + srow) * 32 + ((schunk ^ (((8 + srow) >> 1) & 7)) << 2)) = ycur.y1;
            *reinterpret_cast<float4*>(ystg + (16 + srow) * 32 + ((schunk ^ (((16 + srow) >> 1) & 7)) << 2)) = ycur.y2;
            *reinterpret_cast<float4*>(ystg + (24 + srow) * 32 + ((schunk ^ (((24 + srow) >> 1) & 7)) << 2)) = ycur.y3;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ct + 1 < RS_CT) ynxt = fetch_y(ct + 1);                // ahead of this tile's stores
        }
        float q4[4] = {0.f, 0.f, 0.f, 0.f}, g4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = rowmap(r, h);
            const float v = H3 ? fmaf(acc[ct][r], uns, bias) : acc[ct][r] + bias;      // exact power-of-two rescale
            stg[p * 32 + (((q >> 2) ^ ((p >> 1) & 7)) << 2) + (q & 3)] = v;
            if (row0 + p < g.M) {
                if (BST) {                                             // four short chains, then pairwise: ~6 eps on 32 rows
                    const float yy = ystg[p * 32 + (((q >> 2) ^ ((p >> 1) & 7)) << 2) + (q & 3)];
                    const float d = fmaf(bsc, yy, bsh) > 0.f ? v : 0.f;
                    q4[r >> 2] += d;
                    g4[r >> 2] = fmaf(d, (yy - bmean) * binv, g4[r >> 2]);
                } else {
                    s += v; sq = fmaf(v, v, sq);
                }
            }
            if (SEG) {
                const float sv = sg * v;
                if (r == 0 || sv > best) { best = sv; bp = rowmap(r, 0); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // same-wave LDS hand-off (lanes swap roles)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 8 * t + srow;
            const float4 v4 = *reinterpret_cast<const float4*>(stg + row * 32 + ((schunk ^ ((row >> 1) & 7)) << 2));
            if (row0 + row < g.M)
                *reinterpret_cast<float4*>(g.C + (size_t)(row0 + row) * g.ldc + 256 * cb + 32 * ct + 4 * schunk) = v4;
        }
        asm volatile("" ::: "memory");                                 // the next tile rewrites the image (DS ops stay in order)
        if (BST) { s = (q4[0] + q4[1]) + (q4[2] + q4[3]); sq = (g4[0] + g4[1]) + (g4[2] + g4[3]); }
        if (g.part) {
            const float st = s + __shfl_xor(s, 32, 64), sqt = sq + __shfl_xor(sq, 32, 64);
            if (h == 0) { wstat[2 * (32 * ct + q)] = st; wstat[2 * (32 * ct + q) + 1] = sqt; }
        }
        if (SEG) {
            bp += 4 * h;
            const float ob = __shfl_xor(best, 32, 64);
            const int op = __shfl_xor(bp, 32, 64);
            if (ob > best || (ob == best && op < bp)) { best = ob; bp = op; }          // first max wins (MaxPool2d)
            // MaxPool2d propagates NaN; the compares above skip it.  The column sum over the wave's 32 rows is NaN / inf exactly
            // when one of them is: s - s is then NaN (else +0) and poisons the maximum at the cost of three instructions per tile
            const float stp = s + __shfl_xor(s, 32, 64);
            best += stp - stp;
            if (wave & 1) {
                if (h == 0) { wbest[2 * (32 * ct + q)] = best; wbest[2 * (32 * ct + q) + 1] = __int_as_float(bp); }
            } else {
                acc[ct][0] = best; acc[ct][1] = __int_as_float(bp);                    // parked until the partner has written
            }
        }
        if (BST) ycur = ynxt;
    }
    if (!g.part && !SEG) return;
    __syncthreads();
    if (SEG && !(wave & 1) && h == 0) {                                // even wave: rows 0..31 of the cloud; partner: rows 32..63
        const float* pb = reinterpret_cast<const float*>(freeslot + (wave + 1) * 6144) + 512;
        const int cloud = row0 >> 6;
        if (row0 < g.M) {
#pragma unroll
            for (int ct = 0; ct < RS_CT; ++ct) {
                float best = acc[ct][0];
                int bp = __float_as_int(acc[ct][1]);
                const float ob = pb[2 * (32 * ct + q)];
                if (ob > best) { best = ob; bp = 32 + __float_as_int(pb[2 * (32 * ct + q) + 1]); }
                best += ob - ob;                                                       // the partner's poison (NaN / inf), else +0
                const size_t o = (size_t)cloud * g.N + 256 * cb + 32 * ct + q;
                g.smax[o] = best;
                g.sarg[o] = bp;
            }
        }
    }
    if (g.part && tid < 256) {                                         // the waves in wave order, fp64: one partial row per workgroup
        double s = 0.0, sq = 0.0;
#pragma unroll
        for (int w = 0; w < WV; ++w) {
            const float* ps = reinterpret_cast<const float*>(freeslot + w * 6144);
            s += (double)ps[2 * tid]; sq += (double)ps[2 * tid + 1];
        }
        double* pr = g.part + ((size_t)tile.y * g.N + 256 * cb + tid) * 2;
        pr[0] = s; pr[1] = sq;
    }
}


// ---- weight gradient of the widest layer: dW (N,K) = dy^T f(y), contraction over the rows ------------------------------
// k_gemm_sb's weight gradient stages BOTH operands through LDS with ~4.3 VALU instructions per MFMA (two waves per SIMD:
// VALU-issue bound, 0.39 of the bf16 peak at 49152 x 1024 x 512).  Here the contraction index is the ROW index, so an
// operand fragment (lane = channel, 8 consecutive rows) is 8 dword loads of 128-B row segments -- coalesced as they
// stand.  The dy^T fragments are private to the wave that owns their 64 output rows n: loaded, split and used in
// registers.  The f(y) fragments (f = the previous layer's BatchNorm + ReLU, per channel = per lane) are shared by the
// workgroup's 8 waves: each thread loads, activates and splits ONE fragment of a 32-row stage and stores its three
// planes in fragment order (3 ds_write_b128); consumers read them back with ds_read_b128.  No transposing reads, no DMA.
// Workgroup = 8 waves = 512 (n) x 128 (k) outputs, wave = 64 x 128 (8 accumulator tiles); ~2.5 VALU and 0.25 ds_read_b128
// per MFMA.  grid.z slices the rows; slices are summed in order by k_sum_slices (deterministic).  Same six products per
// multiply-add in the same order, 16 rows per k-step in row order: bit-identical to k_gemm_sb's weight gradient when the
// slice boundaries agree (they need not; the tests hold both to fp64).
constexpr int WG_WAVES = 8;
struct WgArgs {
    const float* dy; const float* y; int M, N, K;     // dy (M,N), y (M,K) row-major
    const float* pscale; const float* pshift;         // (K) or null
    float* slices; int rows_per_slice;                // slices[z][N][K]; rows_per_slice % 32 == 0
    const unsigned* amax;                             // fp16x3 form: bits of max|dy| (FACL_AMAX_SLOTS slots)
    const unsigned* amax_b;                           // fp16x3 form: bound of max|f(y)| (same format)
};

// H3: fp16x3 arithmetic (common.h): dy scaled by the power of two that puts max|dy| in [2^13, 2^14), f(y) by 2^4, two fp16
// planes each, three products; the accumulators are rescaled (exactly) before the slices are written.
// WIDE (round 4, late): the dy^T fragments (lane = channel, 8 consecutive rows) used to be 32 global_load_dword per wave and stage,
// 256 B each -- the CU's address path, not the MFMA pipe, paced the kernel (ablation: -33 % without them; 8 waves x 40 vector-memory
// instructions of 64 addresses per stage against 2 x 1536 MFMA cycles per SIMD).  Now a wave fetches its 32 rows x 64 channels as
// 8 global_load_dwordx4 (1 KiB each: four whole 256-B row pieces), parks them in a per-wave LDS image [row][68] (272-B rows:
// 16-B aligned for the b128 writes, and rows 8 apart sit 32 banks apart, so the two lane halves of a fragment read never
// collide) and reads the fragments back with ds_read_b32: 16 vector-memory instructions per wave and stage instead of 40.
constexpr int WG_TP = 68;
template <bool PRO, bool H3, bool WIDE = false>
__global__ __launch_bounds__(64 * WG_WAVES) void k_wgrad_rs(WgArgs g) {
    constexpr int NPL = H3 ? 2 : 3;
    constexpr int BRING_U4 = 2 * (2 * 4 * NPL * 64);                                // 2 stages x (2 k-steps x 4 k-tiles x planes x 64 lanes): 48 / 32 KiB
    constexpr int WSTG_F = WIDE ? 32 * WG_TP : 32 * 32;                             // per-wave staging: dy image (WIDE) / epilogue tile
    extern __shared__ __attribute__((aligned(16))) char wg_lds[];
    uint4 (*bring)[2 * 4 * NPL * 64] = reinterpret_cast<uint4 (*)[2 * 4 * NPL * 64]>(wg_lds);
    float* const stg_base = reinterpret_cast<float*>(wg_lds + BRING_U4 * 16);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, q = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: the K/128 workgroups that read the same dy rows (same n block, same row slice) are consecutive
    // logical tiles, and each XCD (one L2) walks a contiguous range of them (PMC: 1.0 GB fetched per launch without it)
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const int bl = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int per = total >> 3, rem = total & 7, xcd = bl & 7, slot = bl >> 3;
    const int L = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    const int bx = L % gx, by = (L / gx) % gy, bz = L / (gx * gy);
    const int k0 = bx * 128, n0 = by * 512 + wave * 64;
    const int p0 = bz * g.rows_per_slice;
    const int p1 = p0 + g.rows_per_slice < g.M ? p0 + g.rows_per_slice : g.M;
    const int nst = (p1 - p0 + 31) >> 5;                                           // stages of 32 rows (the last may be ragged)
    float sD = 1.f, sB = 1.f, uns = 1.f;
    if (H3) {
        const int seD = h3_se_wide_of(g.amax), seB = h3_se_of(g.amax_b);
        sD = pow2_biased(seD); sB = pow2_biased(seB); uns = h3_unscale(seD, seB);
    }

    // producer role: this thread's fragment of a stage = (k-step ks_b, k tile kt_b, lane lb): channel kb, rows 16 ks_b + 8 hb + e
    const int ks_b = tid >> 8, kt_b = (tid >> 6) & 3, lb = tid & 63;
    const int kb = k0 + 32 * kt_b + (lb & 31), hb = lb >> 5;
    const float ps = PRO ? g.pscale[kb] : 1.f, pt = PRO ? g.pshift[kb] : 0.f;
    const float* ysrc = g.y + kb;
    const float* dsrc = g.dy + n0 + q;
    auto load_b = [&](int s, float (&r)[8]) {                                      // rows past p1 contribute zeros (via dy = 0)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int p = p0 + 32 * s + 16 * ks_b + 8 * hb + e;
            p = p < p1 ? p : p1 - 1;
            r[e] = ysrc[(size_t)p * g.K];
        }
    };
    auto store_b = [&](int s, const float (&r)[8]) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = PRO ? fmaxf(fmaf(ps, r[e], pt), 0.f) : r[e];   // (backward = training: a NaN is loud through the layer's statistics)
        unsigned hi[4], mi[4], lo[4];
        uint4* d = &bring[s & 1][((ks_b * 4 + kt_b) * NPL) * 64 + lb];
        if (H3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) split_pair_h(v[2 * e] * sB, v[2 * e + 1] * sB, hi[e], mi[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) split_pair(v[2 * e], v[2 * e + 1], hi[e], mi[e], lo[e]);
            d[128] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
        d[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        d[64] = make_uint4(mi[0], mi[1], mi[2], mi[3]);
    };
    // consumer role: dy^T fragments of the wave's two n tiles for both k-steps of a stage: [ks][nt][8]
    auto load_a = [&](int s, float (&r)[2][2][8]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int p = p0 + 32 * s + 16 * ks + 8 * h + e;
                const bool in = p < p1;
                const size_t o = (size_t)(in ? p : p1 - 1) * g.N;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float v = dsrc[o + 32 * nt];
                    r[ks][nt][e] = in ? v : 0.f;
                }
            }
    };

    // WIDE: the same rows as 8 x 16 B per lane (instruction i: rows 4i + (lane >> 4), channels 4 (lane & 15) .. + 3), and the
    // trip through the wave's LDS image
    float* const dimg = stg_base + wave * WSTG_F;
    const float* dsrc4 = g.dy + n0 + 4 * (lane & 15);
    auto load_raw = [&](int s, float4 (&r)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = p0 + 32 * s + 4 * i + (lane >> 4);
            const bool in = p < p1;
            const float4 v = *reinterpret_cast<const float4*>(dsrc4 + (size_t)(in ? p : p1 - 1) * g.N);
            r[i] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto raw_to_frags = [&](const float4 (&r)[8], float (&f)[2][2][8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<float4*>(dimg + (4 * i + (lane >> 4)) * WG_TP + 4 * (lane & 15)) = r[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // same-wave hand-off (lanes swap roles)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) f[ks][nt][e] = dimg[(16 * ks + 8 * h + e) * WG_TP + 32 * nt + q];
        // (no wait here: the reads land under the MFMAs that follow; the image is rewritten a whole stage later, and a wave's
        // DS operations execute in order)
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    constexpr int PA[6] = FACL_SB_PA, PB[6] = FACL_SB_PB;
    constexpr int HA[3] = FACL_H3_PA, HB[3] = FACL_H3_PB;

    float rb[8], ra[2][2][8];
    float4 raw[8];
    load_b(0, rb);
    if (WIDE) { load_raw(0, raw); raw_to_frags(raw, ra); }
    else load_a(0, ra);
    store_b(0, rb);
    if (nst > 1) load_b(1, rb);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        float ran[2][2][8];
        if (s + 1 < nst) {                                                         // next stage's dy^T rows / fragments in flight
            if (WIDE) load_raw(s + 1, raw);
            else load_a(s + 1, ran);
        }
        const uint4* bs = &bring[s & 1][lane];
        // k-step ks: split the wave's two dy^T fragments into planes, then 4 k tiles x (3 | 6) products x 2 n tiles
        auto split_a = [&](int ks, bf16x8 (&af)[2][NPL]) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                unsigned hi[4], mi[4], lo[4];
                if (H3) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) split_pair_h(ra[ks][nt][2 * e] * sD, ra[ks][nt][2 * e + 1] * sD, hi[e], mi[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) split_pair(ra[ks][nt][2 * e], ra[ks][nt][2 * e + 1], hi[e], mi[e], lo[e]);
                    af[nt][NPL - 1] = as_bf16x8(lo[0], lo[1], lo[2], lo[3]);
                }
                af[nt][0] = as_bf16x8(hi[0], hi[1], hi[2], hi[3]);
                af[nt][1] = as_bf16x8(mi[0], mi[1], mi[2], mi[3]);
            }
        };
        auto mfma_ks = [&](int ks, const bf16x8 (&af)[2][NPL]) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                bf16x8 bf[NPL];
#pragma unroll
                for (int p = 0; p < NPL; ++p) bf[p] = __builtin_bit_cast(bf16x8, bs[((ks * 4 + kt) * NPL + p) * 64]);
                if (H3) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        acc[0][kt] = MFMA_F16(__builtin_bit_cast(f16x8h, af[0][HA[t]]), __builtin_bit_cast(f16x8h, bf[HB[t]]), acc[0][kt]);
                        acc[1][kt] = MFMA_F16(__builtin_bit_cast(f16x8h, af[1][HA[t]]), __builtin_bit_cast(f16x8h, bf[HB[t]]), acc[1][kt]);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        acc[0][kt] = MFMA_BF16(af[0][PA[t]], bf[PB[t]], acc[0][kt]);
                        acc[1][kt] = MFMA_BF16(af[1][PA[t]], bf[PB[t]], acc[1][kt]);
                    }
                }
            }
        };
        {
            bf16x8 af0[2][NPL], af1[2][NPL];
            split_a(0, af0);
            mfma_ks(0, af0);
            split_a(1, af1);
            // WIDE: every fragment of this stage is in planes now -> the next stage's rows go through the LDS image HERE, so
            // that the round trip runs under the second k-step's 24 MFMAs instead of in front of the barrier
            if (WIDE && s + 1 < nst) raw_to_frags(raw, ra);
            mfma_ks(1, af1);
        }
        if (s + 1 < nst) {
            store_b(s + 1, rb);                                                    // into the buffer stage s-1 used: all waves left it at the last barrier
            if (s + 2 < nst) load_b(s + 2, rb);
            if (!WIDE) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int e = 0; e < 8; ++e) ra[ks][nt][e] = ran[ks][nt][e];
            }
        }
        __syncthreads();                                                           // stage s+1 is written, stage s is read by everyone
    }

    // ---- epilogue: lane = column k0 + 32 kt + q, register r = row n0 + 32 nt + rowmap(r, h); through the per-wave LDS image
    float* const stg = stg_base + wave * WSTG_F;
    float* const out = g.slices + (size_t)bz * g.N * g.K;
    const int srow = lane >> 3, schunk = lane & 7;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = rowmap(r, h);
                stg[p * 32 + (((q >> 2) ^ ((p >> 1) & 7)) << 2) + (q & 3)] = H3 ? acc[nt][kt][r] * uns : acc[nt][kt][r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = 8 * t + srow;
                const float4 v4 = *reinterpret_cast<const float4*>(stg + row * 32 + ((schunk ^ ((row >> 1) & 7)) << 2));
                *reinterpret_cast<float4*>(out + (size_t)(n0 + 32 * nt + row) * g.K + k0 + 32 * kt + 4 * schunk) = v4;
            }
            asm volatile("" ::: "memory");
        }
}

// sum over the row slices, in order (deterministic); four slice loads in flight per thread
__global__ void k_wg_sum_slices(const float* __restrict__ part, int nz, long long n4, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float4* p4 = reinterpret_cast<const float4*>(part);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 s = p4[i];
        int z = 1;
        for (; z + 3 < nz; z += 4) {
            const float4 v0 = p4[(size_t)z * n4 + i], v1 = p4[(size_t)(z + 1) * n4 + i];
            const float4 v2 = p4[(size_t)(z + 2) * n4 + i], v3 = p4[(size_t)(z + 3) * n4 + i];
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; z < nz; ++z) {
            const float4 v = p4[(size_t)z * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(out)[i] = s;
    }
}

constexpr int RS_LDS8 = RS_LDS_W + 8 * RS_ASLOT + 2 * RS_KPRO * 4 + 8 * 6144;     // 132 KiB: one 8-wave workgroup per CU

// waves per workgroup of the fp16x3 kernels.  FACL_RS_W8=1: 8-wave workgroups (256 rows per k-step of weight planes: half the
// plane re-reads).  Measured equal or slower (round 4, same box, bit-identical results: forward 49152x512x1024 0.182 vs
// 0.184 ms, dgrad 1024->512 0.187 vs 0.176, step 3.023 vs 3.006 ms; gpurun_out/r5g_ab.log) -- the weight-plane traffic is
// not what bounds these kernels.  Default: 4 waves, two workgroups per CU.
int rs_waves(int h3) {
    static const int w8 = getenv("FACL_RS_W8") ? atoi(getenv("FACL_RS_W8")) : 0;
    return (h3 && w8) ? 8 : RS_WAVES;
}

int rs_launch(const RsArgs& g0, hipStream_t st) {
    RsArgs g = g0;
    {   // Phase offset of the CU's second workgroup slot (see the kernel): ticks x k-steps + 500 of the 100-MHz counter, i.e. about
        // half a workgroup's life; only where the launch runs MORE than one round of workgroups (else the wait is pure loss:
        // measured +5..9 % on the 384-workgroup launches).  FACL_RS_PHASE=<ticks per k-step> overrides, 0 = off.
        static const int per_ks = getenv("FACL_RS_PHASE") ? atoi(getenv("FACL_RS_PHASE")) : 30;
        const long long wgs = (long long)(g.N / 256) * ((g.M + 32 * rs_waves(g.h3) - 1) / (32 * rs_waves(g.h3)));
        g.first_round = 512;
        g.phase_ticks = (per_ks > 0 && wgs > g.first_round && rs_waves(g.h3) == RS_WAVES) ? per_ks * (g.K >> 4) + 500 : 0;
    }
    // the dynamic-LDS attribute is per device: one flag per device ordinal (a process may drive several devices, and a
    // forward on the main thread can race a backward on the autograd thread: the worst case sets the attribute twice)
    static bool attr_done_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    bool& attr_done = attr_done_dev[dev];
    if (!attr_done) {
        const void* fns[10] = {(const void*)k_gemm_rs<false, false, true, true>,(const void*)k_gemm_rs<false, false, false>, (const void*)k_gemm_rs<true, false, false>,
                              (const void*)k_gemm_rs<false, true, false>, (const void*)k_gemm_rs<true, true, false>,
                              (const void*)k_gemm_rs<false, false, true>,
                              (const void*)k_gemm_rs<false, false, false, true>, (const void*)k_gemm_rs<true, false, false, true>,
                              (const void*)k_gemm_rs<false, true, false, true>, (const void*)k_gemm_rs<true, true, false, true>};
        for (int i = 0; i < 10; ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
            if (e != hipSuccess) return (int)e;
        }
        const void* fns8[5] = {(const void*)k_gemm_rs<false, false, true, true, 8>, (const void*)k_gemm_rs<false, false, false, true, 8>,
                               (const void*)k_gemm_rs<true, false, false, true, 8>, (const void*)k_gemm_rs<false, true, false, true, 8>,
                               (const void*)k_gemm_rs<true, true, false, true, 8>};
        for (int i = 0; i < 5; ++i) {
            hipError_t e = hipFuncSetAttribute(fns8[i], hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS8);
            if (e != hipSuccess) return (int)e;
        }
        attr_done = true;
    }
    if (rs_waves(g.h3) == 8) {
        dim3 grid(g.N / 256, (g.M + 255) / 256);
        const dim3 blk(512);
        if (g.by) hipLaunchKernelGGL((k_gemm_rs<false, false, true, true, 8>), grid, blk, RS_LDS8, st, g);
        else if (g.pscale && g.smax) hipLaunchKernelGGL((k_gemm_rs<true, true, false, true, 8>), grid, blk, RS_LDS8, st, g);
        else if (g.pscale) hipLaunchKernelGGL((k_gemm_rs<true, false, false, true, 8>), grid, blk, RS_LDS8, st, g);
        else if (g.smax) hipLaunchKernelGGL((k_gemm_rs<false, true, false, true, 8>), grid, blk, RS_LDS8, st, g);
        else hipLaunchKernelGGL((k_gemm_rs<false, false, false, true, 8>), grid, blk, RS_LDS8, st, g);
        return facl_launch_status();
    }
    dim3 grid(g.N / 256, (g.M + 32 * RS_WAVES - 1) / (32 * RS_WAVES));
    const dim3 blk(64 * RS_WAVES);
    if (g.by && g.h3) hipLaunchKernelGGL((k_gemm_rs<false, false, true, true>), grid, blk, RS_LDS, st, g);
    else if (g.by) hipLaunchKernelGGL((k_gemm_rs<false, false, true>), grid, blk, RS_LDS, st, g);
    else if (g.h3 && g.pscale && g.smax) hipLaunchKernelGGL((k_gemm_rs<true, true, false, true>), grid, blk, RS_LDS, st, g);
    else if (g.h3 && g.pscale) hipLaunchKernelGGL((k_gemm_rs<true, false, false, true>), grid, blk, RS_LDS, st, g);
    else if (g.h3 && g.smax) hipLaunchKernelGGL((k_gemm_rs<false, true, false, true>), grid, blk, RS_LDS, st, g);
    else if (g.h3) hipLaunchKernelGGL((k_gemm_rs<false, false, false, true>), grid, blk, RS_LDS, st, g);
    else if (g.pscale && g.smax) hipLaunchKernelGGL((k_gemm_rs<true, true, false>), grid, blk, RS_LDS, st, g);
    else if (g.pscale) hipLaunchKernelGGL((k_gemm_rs<true, false, false>), grid, blk, RS_LDS, st, g);
    else if (g.smax) hipLaunchKernelGGL((k_gemm_rs<false, true, false>), grid, blk, RS_LDS, st, g);
    else hipLaunchKernelGGL((k_gemm_rs<false, false, false>), grid, blk, RS_LDS, st, g);
    return facl_launch_status();
}

}  // namespace

// bytes of the plane buffer of a matrix with N OUTPUT columns and contraction K: the planes (sized for the three-plane form)
// followed by the per-column-tile scale exponents (N/32 int32, padded to 256 B)
static int64_t rs_planes_only_bytes(int N, int K, int with_centers) {
    return (int64_t)(K / 16 + (with_centers ? 1 : 0)) * (N / 32) * 3 * 64 * 16;
}
extern "C" int64_t facl_gemm_rs_planes_bytes(int N, int K, int with_centers) {
    if (N < 32 || K < 16) return 0;
    return rs_planes_only_bytes(N, K, with_centers) + (((int64_t)(N / 32) * 4 + 255) / 256) * 256;
}

// planes for y = a W^T (transposed = 0: W (N,K) row-major, leading dimension ldw; output columns N, contraction K;
// Wc (N,3), leading dimension ldwc, adds the centre k-step) or for da = dy W (transposed = 1: output columns K,
// contraction N).  Output columns must be a multiple of 32, the contraction a multiple of 16.
extern "C" int facl_gemm_rs_planes(const float* W, int ldw, int N, int K, int transposed, const float* Wc, int ldwc,
                                   int half, void* planes, void* stream) {
    if (!W || !planes) return FACL_E_NULL;
    if (N < 1 || K < 1 || ldw < K) return FACL_E_SHAPE;
    const int NO = transposed ? K : N, NC = transposed ? N : K;
    if ((NO & 31) || (NC & 15) || (transposed && Wc) || NC / 16 + (Wc ? 1 : 0) > RS_PL_ITEMS * 4) return FACL_E_SHAPE;
    const long long so = transposed ? 1 : ldw, sc = transposed ? ldw : 1;
    int* hdr = (int*)((char*)planes + rs_planes_only_bytes(NO, NC, Wc ? 1 : 0));
    hipLaunchKernelGGL(k_rs_planes, dim3(NO / 32), dim3(256), 0, (hipStream_t)stream, W, so, sc, NO, NC, Wc, ldwc, (uint4*)planes,
                       half ? 1 : 0, hdr);
    return facl_launch_status();
}

// n <= 8 matrices in ONE launch; arrays of the per-matrix arguments of facl_gemm_rs_planes.  absmax_x / absmax_n / absmax_amax
// (all or none): the same launch also raises the slots of `absmax_amax` (FACL_AMAX_WORDS uint32) to max|absmax_x[0 .. n)| --
// the centroid coordinates, whose k-step shares the fp16x3 scale of the first layer's row operand.
extern "C" int facl_gemm_rs_planes_multi(int n, const float* const* W, const int* ldw, const int* N, const int* K,
                                         const int* transposed, const float* const* Wc, const int* ldwc, const int* half,
                                         void* const* planes, const float* absmax_x, int64_t absmax_n, uint32_t* absmax_amax,
                                         void* stream) {
    if (!W || !ldw || !N || !K || !transposed || !Wc || !ldwc || !half || !planes) return FACL_E_NULL;
    if ((absmax_x == nullptr) != (absmax_amax == nullptr)) return FACL_E_NULL;
    if (n < 1 || n > RS_MAXJOBS || (absmax_x && absmax_n < 1)) return FACL_E_SHAPE;
    RsPlaneJobs jb;
    jb.n = n;
    jb.first[0] = 0;
    for (int j = 0; j < RS_MAXJOBS; ++j) {
        if (j >= n) { jb.W[j] = nullptr; jb.xc[j] = nullptr; jb.out[j] = nullptr; jb.hdr[j] = nullptr; jb.so[j] = jb.sc[j] = 0; jb.NO[j] = jb.NC[j] = 32; jb.ldxc[j] = 0; jb.half[j] = 0; jb.first[j + 1] = jb.first[j]; continue; }
        if (!W[j] || !planes[j]) return FACL_E_NULL;
        if (N[j] < 1 || K[j] < 1 || ldw[j] < K[j]) return FACL_E_SHAPE;
        const int NO = transposed[j] ? K[j] : N[j], NC = transposed[j] ? N[j] : K[j];
        if ((NO & 31) || (NC & 15) || (transposed[j] && Wc[j]) || NC / 16 + (Wc[j] ? 1 : 0) > RS_PL_ITEMS * 4) return FACL_E_SHAPE;
        jb.W[j] = W[j]; jb.so[j] = transposed[j] ? 1 : ldw[j]; jb.sc[j] = transposed[j] ? ldw[j] : 1;
        jb.NO[j] = NO; jb.NC[j] = NC; jb.xc[j] = Wc[j]; jb.ldxc[j] = ldwc[j]; jb.out[j] = (uint4*)planes[j];
        jb.hdr[j] = (int*)((char*)planes[j] + rs_planes_only_bytes(NO, NC, Wc[j] ? 1 : 0));
        jb.half[j] = half[j] ? 1 : 0;
        jb.first[j + 1] = jb.first[j] + NO / 32;
    }
    jb.ax = absmax_x; jb.an = absmax_n; jb.aamax = absmax_amax;
    jb.anb = absmax_x ? (int)((absmax_n + 4095) / 4096 < 16 ? (absmax_n + 4095) / 4096 : 16) : 0;
    hipLaunchKernelGGL(k_rs_planes_multi, dim3(jb.first[n] + jb.anb), dim3(256), 0, (hipStream_t)stream, jb);
    return facl_launch_status();
}

// 1 when facl_gemm_rs_fwd / _dgrad take the shape (rows M, contraction K, output columns N), else 0
extern "C" int facl_gemm_rs_supported(int64_t M, int K, int N) {
    return (M >= 2048 && M <= 0x7fffffff && K >= 64 && K <= RS_KMAX && !(K & 31) && N >= 256 && !(N & 255)) ? 1 : 0;
}

// y (M,N) = f(a) W^T + bias [+ centers Wc^T]   with `planes` = facl_gemm_rs_planes(W, ..., transposed 0, Wc).
// f = relu(pscale*a + pshift) per input channel when pscale is given (the previous layer's BatchNorm + ReLU).
// sums (N,2): per-column (sum, sumsq) of y (or null).  sgn / ymax / arg (all or none): fused my_max_pool over blocks of 64
// rows as in facl_gemm_fwd_segmax (M % 64 == 0).
// half = 1 (fp16x3): `amax_a` = FACL_AMAX_WORDS uint32 holding the bits of (a bound of) max|f(a)| and, with centres, of
// max|centre coordinate| (the centre k-step shares A's scale); facl_bn_finalize / facl_sa_pool / facl_absmax /
// facl_rows_act_amax produce it.
extern "C" int facl_gemm_rs_fwd(const float* a, int64_t M, int K, const void* planes, int half, const uint32_t* amax_a, int N,
                                const float* bias, const float* pscale, const float* pshift, const float* centers, float* y,
                                double* sums, const float* sgn, float* ymax, int32_t* arg, void* ws, void* stream) {
    if (!a || !planes || !y || (sums && !ws) || (half && !amax_a)) return FACL_E_NULL;
    if (!facl_gemm_rs_supported(M, K, N)) return FACL_E_SHAPE;
    if ((pscale == nullptr) != (pshift == nullptr)) return FACL_E_NULL;
    if (pscale && K > RS_KPRO) return FACL_E_SHAPE;
    if ((sgn == nullptr) != (ymax == nullptr) || (sgn == nullptr) != (arg == nullptr)) return FACL_E_NULL;
    if (sgn && (M & 63)) return FACL_E_SHAPE;
    if (((uintptr_t)a | (uintptr_t)planes) & 15) return FACL_E_ALIGN;
    const int rpw = 32 * rs_waves(half ? 1 : 0);
    const int prow = (int)((M + rpw - 1) / rpw);
    if (sums && (size_t)prow * N * 2 * sizeof(double) > ((size_t)facl_ws_bytes() - FACL_WS_TICKET_BYTES)) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int* wse = (const int*)((const char*)planes + rs_planes_only_bytes(N, K, centers ? 1 : 0));
    RsArgs g{a, K, (int)M, K, (const uint4*)planes, N / 32, N, bias, pscale, pshift, centers, y, N,
             sums ? (double*)ws : nullptr, sgn, ymax, arg, nullptr, nullptr, half ? 1 : 0, amax_a, wse};
    int rc = rs_launch(g, st);
    if (rc || !sums) return rc;
    return facl_reduce_rows((const double*)ws, prow, 2 * N, sums, st);
}

// da (M,K) = dy (M,N) W   with `planes` = facl_gemm_rs_planes(W, ..., transposed 1, half).  half = 1 (fp16x3): `amax` = device
// scalar holding the bits of max|dy| (facl_rows_bwd_apply_amax / facl_segmax_bwd_apply_amax produce it), from which the
// kernel takes dy's power-of-two scale.
extern "C" int facl_gemm_rs_dgrad(const float* dy, int64_t M, int N, const void* planes, int half, const uint32_t* amax,
                                  int K, float* da, void* stream) {
    if (!dy || !planes || !da || (half && !amax)) return FACL_E_NULL;
    if (!facl_gemm_rs_supported(M, N, K)) return FACL_E_SHAPE;
    if (((uintptr_t)dy | (uintptr_t)planes) & 15) return FACL_E_ALIGN;
    const int* wse = (const int*)((const char*)planes + rs_planes_only_bytes(K, N, 0));
    RsArgs g{dy, N, (int)M, N, (const uint4*)planes, K / 32, K, nullptr, nullptr, nullptr, nullptr, da, K,
             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, half ? 1 : 0, amax, wse};
    return rs_launch(g, (hipStream_t)stream);
}

// facl_gemm_rs_dgrad + the statistics pass of the BatchNorm backward that consumes da: with y (M,K) = the raw output of
// the layer whose activation relu(bn(y)) fed this GEMM's forward, and bnc (5,K) its constants, sums (K,2) = per column
// (sum_r d, sum_r d*yhat), d = da[r] where the ReLU was open, yhat = (y - mean)*invstd -- what facl_rows_bwd_stats(da, y, ...)
// computes, taken from the accumulator tile before it is stored (da and y are not re-read by a statistics pass).
extern "C" int facl_gemm_rs_dgrad_bnstats(const float* dy, int64_t M, int N, const void* planes, int half,
                                          const uint32_t* amax, int K, float* da, const float* y, const float* bnc,
                                          double* sums, void* ws, void* stream) {
    if (!dy || !planes || !da || !y || !bnc || !sums || !ws || (half && !amax)) return FACL_E_NULL;
    if (!facl_gemm_rs_supported(M, N, K)) return FACL_E_SHAPE;
    if (((uintptr_t)dy | (uintptr_t)planes) & 15) return FACL_E_ALIGN;
    const int rpw = 32 * rs_waves(half ? 1 : 0);
    const int prow = (int)((M + rpw - 1) / rpw);
    if ((size_t)prow * K * 2 * sizeof(double) > ((size_t)facl_ws_bytes() - FACL_WS_TICKET_BYTES)) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int* wse = (const int*)((const char*)planes + rs_planes_only_bytes(K, N, 0));
    RsArgs g{dy, N, (int)M, N, (const uint4*)planes, K / 32, K, nullptr, nullptr, nullptr, nullptr, da, K,
             (double*)ws, nullptr, nullptr, nullptr, y, bnc, half ? 1 : 0, amax, wse};
    int rc = rs_launch(g, st);
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, prow, 2 * K, sums, st);
}

// dW (N,K) = dy^T (N,M) f(y) (M,K), f = relu(pscale*y + pshift) per column of y when pscale is given (else identity), on
// the register-streamed weight-gradient kernel.  N % 512 == 0, K % 128 == 0, M >= 4096; `slices` = scratch for
// facl_gemm_rs_wgrad_slices(M, N, K) * N * K floats.  FACL_E_CONFIG when the shape is better served by facl_gemm_wgrad[_pro]
// (fewer than 8 output blocks: the slice count, and with it the partial-slab traffic, would explode).
extern "C" int facl_gemm_rs_wgrad_slices(int64_t M, int N, int K) {
    if (M < 4096 || N < 512 || (N & 511) || K < 128 || (K & 127)) return 0;
    const int tiles = (N / 512) * (K / 128);
    if (tiles < 8) return 0;
    int nz = 256 / tiles;
    if (nz < 1) nz = 1;
    int rps = (int)((M + nz - 1) / nz);
    rps = (rps + 31) / 32 * 32;
    return (int)((M + rps - 1) / rps);
}
// amax (or null): fp16x3 arithmetic with dy's scale taken from the bits of max|dy| it points to; `amax_b` (required with
// amax): the bound of max|f(y)| in the same format (the one the forward GEMM that consumed f(y) was given).
extern "C" int facl_gemm_rs_wgrad(const float* dy, const float* y, int64_t M, int N, int K, const float* pscale,
                                  const float* pshift, const uint32_t* amax, const uint32_t* amax_b, float* dW, float* slices,
                                  void* stream) {
    if (!dy || !y || !dW || !slices || (amax && !amax_b)) return FACL_E_NULL;
    if ((pscale == nullptr) != (pshift == nullptr)) return FACL_E_NULL;
    if (M > 0x7fffffff) return FACL_E_SHAPE;
    const int nz = facl_gemm_rs_wgrad_slices(M, N, K);
    if (nz < 1) return FACL_E_CONFIG;
    if (((uintptr_t)slices | (uintptr_t)dW) & 15) return FACL_E_ALIGN;
    const int tiles = (N / 512) * (K / 128);
    int nz0 = 256 / tiles;
    if (nz0 < 1) nz0 = 1;
    int rps = (int)((M + nz0 - 1) / nz0);
    rps = (rps + 31) / 32 * 32;
    hipStream_t st = (hipStream_t)stream;
    WgArgs g{dy, y, (int)M, N, K, pscale, pshift, slices, rps, amax, amax_b};
    dim3 grid(K / 128, N / 512, nz);
    const dim3 blk(64 * WG_WAVES);
    // dynamic LDS: plane ring (48 / 32 KiB) + per-wave staging (WIDE: the dy image, 8.5 KiB per wave; else the 4-KiB epilogue tile)
    static const int wide_env = getenv("FACL_WGRAD_WIDE") ? atoi(getenv("FACL_WGRAD_WIDE")) : 1;  // 0: dword fragment loads (A/B)
    const int wide = wide_env && !(((uintptr_t)dy) & 15);                                          // the 16-byte row loads need an aligned dy
    const int lds3 = 2 * (2 * 4 * 3 * 64) * 16, lds2 = 2 * (2 * 4 * 2 * 64) * 16;
    const int stg_n = WG_WAVES * 32 * 32 * 4, stg_w = WG_WAVES * 32 * WG_TP * 4;
    static bool attr_done[64] = {};
    const void* fns[6] = {(const void*)k_wgrad_rs<true, true>, (const void*)k_wgrad_rs<true, false>, (const void*)k_wgrad_rs<false, true>,
                          (const void*)k_wgrad_rs<false, false>, (const void*)k_wgrad_rs<true, true, true>,
                          (const void*)k_wgrad_rs<false, true, true>};
    if (int rc0 = facl_set_dynamic_lds(attr_done, fns, 6, lds3 + stg_w)) return rc0;
    if (pscale && amax && wide) hipLaunchKernelGGL((k_wgrad_rs<true, true, true>), grid, blk, lds2 + stg_w, st, g);
    else if (amax && wide) hipLaunchKernelGGL((k_wgrad_rs<false, true, true>), grid, blk, lds2 + stg_w, st, g);
    else if (pscale && amax) hipLaunchKernelGGL((k_wgrad_rs<true, true>), grid, blk, lds2 + stg_n, st, g);
    else if (pscale) hipLaunchKernelGGL((k_wgrad_rs<true, false>), grid, blk, lds3 + stg_n, st, g);
    else if (amax) hipLaunchKernelGGL((k_wgrad_rs<false, true>), grid, blk, lds2 + stg_n, st, g);
    else hipLaunchKernelGGL((k_wgrad_rs<false, false>), grid, blk, lds3 + stg_n, st, g);
    int rc = facl_launch_status();
    if (rc) return rc;
    const long long n4 = (long long)N * K / 4;
    const int rgrid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_wg_sum_slices, dim3(rgrid), dim3(256), 0, st, slices, nz, n4, dW);
    return facl_launch_status();
}
