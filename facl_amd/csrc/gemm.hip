// fp32 MFMA GEMM for the dense 1x1 "point-MLP" channel contractions of the encoder tail
// (net3DV_3 / netR_FC: cn3d_model_conbag.py:61-88), forward, dgrad and wgrad, with the BatchNorm plumbing
// fused in:  * prologue  A' = relu(pscale[k]*A + pshift[k])  applied while the A tile is staged (forward),
//            * epilogue  + bias[n], + rank-3 centre term (the xyz columns of torch.cat((yt, xt),1), :219),
//                        per-column (sum, sumsq) partials for the train-mode BN that follows,
//            * split-K partial tiles for the weight gradient (contraction over the 49,152 centroid rows).
//
//   C[i][j] = sum_k  opA(i,k) * opB(j,k)        i < MI, j < NJ, k < KK
// Operand layouts (element (idx,k)):  KC ("k-contiguous")  ptr[idx*ld + k]   rows of activations / weights
//                                     IC ("idx-contiguous") ptr[k*ld + idx]   the transposed views of dgrad/wgrad
//   forward  y  = a W^T      : A = a  (KC), B = W (KC)
//   dgrad    da = dy W       : A = dy (KC), B = W (IC)          (k = output channel)
//   wgrad    dW = dy^T a     : A = dy (IC), B = a (IC)          (k = row, split over blockIdx.z)
// Tile 128x128x32, 256 threads = 2x2 waves of 64x64 (2x2 v_mfma_f32_32x32x2_f32 tiles, 64 accumulators),
// operands staged global -> registers -> LDS as [k][idx] (+4 pad) so that the MFMA fragments are
// conflict-free ds_read_b32; register prefetch of the next k-tile overlaps the 64 MFMAs of the current one.
// Roofline: MFMA fp32 (157.3 TFLOP/s); 2*MI*NJ*KK FLOP per call.
#include "common.h"
#include <stdlib.h>

int facl_reduce_rows(const double* part, int rows, int V, double* out, hipStream_t st);

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;      // KC tiles: LDS image [idx][k], row = 36 floats: b128 writes AND b128 fragment reads
                                 // are conflict-free (36*q mod 64 hits 16 distinct 4-dword slots for 16 lanes)
// IC tiles: LDS image [k][idx], row = NIDX+4 floats (16-B aligned rows).  NIDX = 64*T (T = 32x32 tiles per wave side)
enum { KC = 0, IC = 1 };

struct GemmArgs {
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    int MI, NJ, KK;
    const float* bias;                 // (NJ) or null
    const float* pscale; const float* pshift;   // (KK) prologue on A, or null
    const float* xa; const float* xb; int ldxb;  // rank-3 extra term: C += xa[i][0..2] . xb[j][0..2]  (or null)
    double* part;                      // column statistics partials [(MI/64)][NJ][2], or null
    int kchunk;                        // split-K: k range per blockIdx.z (wgrad); C then is [z][MI][NJ]
    // fused max over each block of 64 consecutive rows (my_max_pool over the S = 64 centroids of a cloud): per
    // (row block, column) max of sgn[j]*C and the FIRST row that attains it; 128x128 tiles only.  Null when unused.
    const float* sgn; float* smax; int* sarg;
    int prec;                          // 0: fp32 result (bf16x6 or fp32 MFMA), 1: fp16 inputs, one MFMA product, fp32 accumulate
    const unsigned* amax;              // NP = 4 (fp16x3 weight gradient): bits of max|A| in hashed slots (common.h), A = dy
    const unsigned* amax_b;            // NP = 4: bound of max|B| (the activation operand), same format
    int accum;                         // 1: C += result (gemm_epilogue, vectorised stores only: facl_gemm_wgrad_acc)
};

template <int LAY, int T>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int idx0, int nidx, int k0, int kend,
                                          float4 (&r)[2 * T], int tid) {
    // (64T) idx x 32 k = 512T float4; thread t takes 2T of them
    if (LAY == IC) {      // float4 along idx: 16T float4 per k-row
#pragma unroll
        for (int i = 0; i < 2 * T; ++i) {
            const int k = k0 + tid / (16 * T) + (16 / T) * i, idx = idx0 + 4 * (tid % (16 * T));
            if (k < kend && idx + 3 < nidx) r[i] = *reinterpret_cast<const float4*>(P + (size_t)k * ld + idx);
            else {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (k < kend && idx + e < nidx) ? P[(size_t)k * ld + idx + e] : 0.f;
                r[i] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
    } else {              // float4 along k: 8 lanes cover one 128-B row segment (full cache lines per wave-instruction;
                          // one-row-per-lane "fragment-shaped" loads touch 64 lines per instruction instead of 8)
#pragma unroll
        for (int i = 0; i < 2 * T; ++i) {
            const int idx = idx0 + (tid >> 3) + 32 * i, k = k0 + 4 * (tid & 7);
            if (idx < nidx && k + 3 < kend) r[i] = *reinterpret_cast<const float4*>(P + (size_t)idx * ld + k);
            else {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (idx < nidx && k + e < kend) ? P[(size_t)idx * ld + k + e] : 0.f;
                r[i] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
    }
}

template <int LAY, bool PRO, int TT>
__device__ __forceinline__ void store_tile(float* __restrict__ T, const float4 (&r)[2 * TT], int tid, int k0,
                                           const float* __restrict__ ps, const float* __restrict__ pt) {
    constexpr int LDT = 64 * TT + 4;
    if (LAY == IC) {
#pragma unroll
        for (int i = 0; i < 2 * TT; ++i) {
            const int k = tid / (16 * TT) + (16 / TT) * i;
            *reinterpret_cast<float4*>(&T[k * LDT + 4 * (tid % (16 * TT))]) = r[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2 * TT; ++i) {
            const int k = 4 * (tid & 7), il = (tid >> 3) + 32 * i;
            float v[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
            if (PRO) {
                const float4 s = *reinterpret_cast<const float4*>(ps + k0 + k);
                const float4 t = *reinterpret_cast<const float4*>(pt + k0 + k);
                v[0] = relu_nan(fmaf(s.x, v[0], t.x)); v[1] = relu_nan(fmaf(s.y, v[1], t.y));
                v[2] = relu_nan(fmaf(s.z, v[2], t.z)); v[3] = relu_nan(fmaf(s.w, v[3], t.w));
            }
            *reinterpret_cast<float4*>(&T[il * LDK + k]) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// XCD-aware tile order: workgroup b of the linearised grid is dispatched to XCD b % 8 (8 XCDs, one L2 each).  Remap so
// that each XCD works through a CONTIGUOUS range of logical tiles: the column tiles that share an A row-panel (and
// all tiles of one split-K slice) then hit the same L2 instead of pulling the panel through the fabric 8 times.
struct TileId { int x, y, z; };
__device__ __forceinline__ TileId xcd_tile() {
    const int nbx = gridDim.x, nby = gridDim.y;
    const int total = nbx * nby * gridDim.z;
    const int b = blockIdx.x + nbx * (blockIdx.y + nby * blockIdx.z);
    const int per = total >> 3, rem = total & 7, xcd = b & 7, slot = b >> 3;
    const int L = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    TileId t;
    t.x = L % nbx; t.y = (L / nbx) % nby; t.z = L / (nbx * nby);
    return t;
}

template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[TM][TN], float* smem, int i0, int j0,
                                              int wave, int lane, int by, int bz) {
    const int h = lane >> 5, q = lane & 31, wr = wave >> 1, wc = wave & 1;
    // ---- epilogue.  The accumulators hold (lane = column, register = row): column statistics are in-lane sums;
    // the tile itself is transposed through LDS (the staging buffers are free now) so that it leaves as
    // 16-byte-per-lane row-major stores (4-byte stores are store-ISSUE bound).
    constexpr int WR = 32 * TM, WC = 32 * TN, SP = WC + 4;
    float* Cz = g.C + (size_t)bz * g.MI * g.ldc;
    float* stg = smem + wave * (WR * SP);
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int j = j0 + WC * wc + 32 * b + q;
        const bool jin = j < g.NJ;
        const float bias = (g.bias && jin) ? g.bias[j] : 0.f;
        float xb0 = 0.f, xb1 = 0.f, xb2 = 0.f;
        if (g.xa && jin) { xb0 = g.xb[(size_t)j * g.ldxb]; xb1 = g.xb[(size_t)j * g.ldxb + 1]; xb2 = g.xb[(size_t)j * g.ldxb + 2]; }
        float s = 0.f, sq = 0.f;
        const float sg = (g.smax && jin) ? sgn_of(g.sgn[j]) : 1.f;
        float best = 0.f;
        int bp = 0;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = 32 * a + rowmap(r, h);
                const int i = i0 + WR * wr + il;
                float v = acc[a][b][r] + bias;
                if (g.xa && i < g.MI)
                    v = fmaf(g.xa[(size_t)i * 3], xb0, fmaf(g.xa[(size_t)i * 3 + 1], xb1, fmaf(g.xa[(size_t)i * 3 + 2], xb2, v)));
                stg[il * SP + 32 * b + q] = v;
                if (i < g.MI && jin) { s += v; sq = fmaf(v, v, sq); }
                if (TM == 2 && g.smax) {
                    const float sv = sg * v;
                    if ((a == 0 && r == 0) || sv > best || sv != sv) { best = sv; bp = 32 * a + rowmap(r, 0); }   // a NaN wins and stays (MaxPool2d)
                }
            }
        if (TM == 2 && g.smax) {                                           // the wave tile's 64 rows = one cloud
            bp += 4 * h;
            const float ob = __shfl_xor(best, 32, 64);
            const int op = __shfl_xor(bp, 32, 64);
            if ((ob > best || (ob == best && op < bp) || ob != ob) && best == best) { best = ob; bp = op; }   // first max wins; NaN stays (MaxPool2d)
            if (h == 0 && jin) {
                const size_t o = (size_t)((i0 + WR * wr) >> 6) * g.NJ + j;
                g.smax[o] = best;
                g.sarg[o] = bp;
            }
        }
        if (g.part) {
            const float st = s + __shfl_xor(s, 32, 64), sqt = sq + __shfl_xor(sq, 32, 64);
            if (h == 0 && jin) {
                double* pr = g.part + ((size_t)(by * 2 + wr) * g.NJ + j) * 2;
                pr[0] = (double)st; pr[1] = (double)sqt;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // same-wave LDS hand-off (lanes swap roles)
    const int jw = j0 + WC * wc;
    const bool vec_ok = ((g.ldc & 3) == 0) && (jw + WC <= g.NJ);
    constexpr int LPR = WC / 4;                                            // lanes per row (float4 each)
#pragma unroll
    for (int t = 0; t < WR * LPR / 64; ++t) {
        const int il = (64 / LPR) * t + lane / LPR, c4 = (lane % LPR) * 4;
        const int i = i0 + WR * wr + il;
        if (i >= g.MI) continue;
        float4 v = *reinterpret_cast<const float4*>(&stg[il * SP + c4]);
        if (vec_ok) {
            float4* dst = reinterpret_cast<float4*>(&Cz[(size_t)i * g.ldc + jw + c4]);
            if (g.accum) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *dst = v;
        } else {
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (jw + c4 + k < g.NJ) Cz[(size_t)i * g.ldc + jw + c4 + k] = e[k] + (g.accum ? Cz[(size_t)i * g.ldc + jw + c4 + k] : 0.f);
        }
    }
}

// The same epilogue staged 32 rows (one MFMA row tile) at a time: half the LDS of gemm_epilogue for 128x128 blocks (35 KiB
// instead of 69 KiB).  Used by the fp16-input instantiation, whose operand images are small (one plane): three workgroups
// per CU instead of two for GEMMs that are latency-bound (4-8 stages per tile at K = 128 / 256).  Per-column state
// (statistics, running max) lives in registers across the row-tile passes; sums accumulate in the same order.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_h(const GemmArgs& g, f32x16 (&acc)[TM][TN], float* smem, int i0, int j0,
                                                int wave, int lane, int by, int bz) {
    const int h = lane >> 5, q = lane & 31, wr = wave >> 1, wc = wave & 1;
    constexpr int WR = 32 * TM, WC = 32 * TN, SP = WC + 4;
    float* Cz = g.C + (size_t)bz * g.MI * g.ldc;
    float* stg = smem + wave * (32 * SP);
    float bias[TN], xb0[TN], xb1[TN], xb2[TN], s[TN], sq[TN], sg[TN], best[TN];
    int bp[TN];
    bool jin[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int j = j0 + WC * wc + 32 * b + q;
        jin[b] = j < g.NJ;
        bias[b] = (g.bias && jin[b]) ? g.bias[j] : 0.f;
        xb0[b] = xb1[b] = xb2[b] = 0.f;
        if (g.xa && jin[b]) { xb0[b] = g.xb[(size_t)j * g.ldxb]; xb1[b] = g.xb[(size_t)j * g.ldxb + 1]; xb2[b] = g.xb[(size_t)j * g.ldxb + 2]; }
        s[b] = sq[b] = 0.f;
        sg[b] = (g.smax && jin[b]) ? sgn_of(g.sgn[j]) : 1.f;
        best[b] = 0.f;
        bp[b] = 0;
    }
    const int jw = j0 + WC * wc;
    const bool vec_ok = ((g.ldc & 3) == 0) && (jw + WC <= g.NJ);
    constexpr int LPR = WC / 4;                                            // lanes per row (float4 each)
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = rowmap(r, h);
                const int i = i0 + WR * wr + 32 * a + il;
                float v = acc[a][b][r] + bias[b];
                if (g.xa && i < g.MI)
                    v = fmaf(g.xa[(size_t)i * 3], xb0[b], fmaf(g.xa[(size_t)i * 3 + 1], xb1[b], fmaf(g.xa[(size_t)i * 3 + 2], xb2[b], v)));
                stg[il * SP + 32 * b + q] = v;
                if (i < g.MI && jin[b]) { s[b] += v; sq[b] = fmaf(v, v, sq[b]); }
                if (TM == 2 && g.smax) {
                    const float sv = sg[b] * v;
                    if ((a == 0 && r == 0) || sv > best[b] || sv != sv) { best[b] = sv; bp[b] = 32 * a + rowmap(r, 0); }   // a NaN wins and stays
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // same-wave LDS hand-off (lanes swap roles)
#pragma unroll
        for (int t = 0; t < 32 * LPR / 64; ++t) {
            const int il = (64 / LPR) * t + lane / LPR, c4 = (lane % LPR) * 4;
            const int i = i0 + WR * wr + 32 * a + il;
            if (i >= g.MI) continue;
            const float4 v = *reinterpret_cast<const float4*>(&stg[il * SP + c4]);
            if (vec_ok) *reinterpret_cast<float4*>(&Cz[(size_t)i * g.ldc + jw + c4]) = v;
            else {
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (jw + c4 + k < g.NJ) Cz[(size_t)i * g.ldc + jw + c4 + k] = e[k];
            }
        }
        asm volatile("" ::: "memory");                                     // the next row tile reuses the staging rows
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int j = j0 + WC * wc + 32 * b + q;
        if (TM == 2 && g.smax) {                                           // the wave tile's 64 rows = one cloud
            int p = bp[b] + 4 * h;
            float bv = best[b];
            const float ob = __shfl_xor(bv, 32, 64);
            const int op = __shfl_xor(p, 32, 64);
            if ((ob > bv || (ob == bv && op < p) || ob != ob) && bv == bv) { bv = ob; p = op; }      // first max wins; NaN stays (MaxPool2d)
            if (h == 0 && jin[b]) {
                const size_t o = (size_t)((i0 + WR * wr) >> 6) * g.NJ + j;
                g.smax[o] = bv;
                g.sarg[o] = p;
            }
        }
        if (g.part) {
            const float st = s[b] + __shfl_xor(s[b], 32, 64), sqt = sq[b] + __shfl_xor(sq[b], 32, 64);
            if (h == 0 && jin[b]) {
                double* pr = g.part + ((size_t)(by * 2 + wr) * g.NJ + j) * 2;
                pr[0] = (double)st; pr[1] = (double)sqt;
            }
        }
    }
}

template <int LA, int LB, bool PRO, int TM, int TN>
__global__ __launch_bounds__(256, 2) void k_gemm(GemmArgs g) {
    constexpr int BM = 64 * TM, BN = 64 * TN, LDA = BM + 4, LDB = BN + 4;
    constexpr int AF = (LA == KC) ? BM * LDK : BK * LDA, BF = (LB == KC) ? BN * LDK : BK * LDB;
    constexpr int STG = 4 * (32 * TM) * (32 * TN + 4);                 // epilogue staging: 4 waves x rows x padded cols
    constexpr int SMEM = (2 * (AF + BF) > STG) ? 2 * (AF + BF) : STG;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* const sA0 = smem;                     // stage s of A at sA0 + s*AF, of B at sB0 + s*BF
    float* const sB0 = smem + 2 * AF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, q = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const TileId tile = xcd_tile();
    const int i0 = tile.y * BM, j0 = tile.x * BN;
    const int kbeg = tile.z * g.kchunk;
    const int kend = (kbeg + g.kchunk < g.KK) ? kbeg + g.kchunk : g.KK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[2 * TM], rb[2 * TN];
    load_tile<LA, TM>(g.A, g.lda, i0, g.MI, kbeg, kend, ra, tid);
    load_tile<LB, TN>(g.B, g.ldb, j0, g.NJ, kbeg, kend, rb, tid);
    store_tile<LA, PRO, TM>(sA0, ra, tid, kbeg, g.pscale, g.pshift);
    store_tile<LB, false, TN>(sB0, rb, tid, kbeg, nullptr, nullptr);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {
            load_tile<LA, TM>(g.A, g.lda, i0, g.MI, k0 + BK, kend, ra, tid);
            load_tile<LB, TN>(g.B, g.ldb, j0, g.NJ, k0 + BK, kend, rb, tid);
        }
        // MFMA k-slot of half h in step s is k = 16h + s for BOTH operands (any bijection works)
        const float* pa = sA0 + cur * AF;
        const float* pb = sB0 + cur * BF;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float av[TM][4], bv[TN][4];
            // KC operands: one b128 read fetches this lane's 4 consecutive k of the step group;
            // IC operands are read one k at a time right before their MFMAs (finer LDS / MFMA interleave)
            if (LA == KC) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const float4 u = *reinterpret_cast<const float4*>(&pa[(32 * TM * wr + 32 * a + q) * LDK + 16 * h + 4 * s4]);
                    av[a][0] = u.x; av[a][1] = u.y; av[a][2] = u.z; av[a][3] = u.w;
                }
            }
            if (LB == KC) {
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const float4 u = *reinterpret_cast<const float4*>(&pb[(32 * TN * wc + 32 * b + q) * LDK + 16 * h + 4 * s4]);
                    bv[b][0] = u.x; bv[b][1] = u.y; bv[b][2] = u.z; bv[b][3] = u.w;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (LA == IC) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) av[a][e] = pa[(16 * h + 4 * s4 + e) * LDA + 32 * TM * wr + 32 * a + q];
                }
                if (LB == IC) {
#pragma unroll
                    for (int b = 0; b < TN; ++b) bv[b][e] = pb[(16 * h + 4 * s4 + e) * LDB + 32 * TN * wc + 32 * b + q];
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) acc[a][b] = MFMA32(av[a][e], bv[b][e], acc[a][b]);
            }
        }
        if (more) {
            store_tile<LA, PRO, TM>(sA0 + (cur ^ 1) * AF, ra, tid, k0 + BK, g.pscale, g.pshift);
            store_tile<LB, false, TN>(sB0 + (cur ^ 1) * BF, rb, tid, k0 + BK, nullptr, nullptr);
        }
        __syncthreads();
        cur ^= 1;
    }

    gemm_epilogue<TM, TN>(g, acc, smem, i0, j0, wave, lane, tile.y, tile.z);
}

// ---------------------------------------------------------------------------------------------------------
// LDS-DMA variant: tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write
// instructions, no vmcnt wait in front of an LDS store).  The DMA destination is lane-linear (wave-uniform base +
// lane*16 B), so the LDS images are UNPADDED and the bank-conflict swizzle lives on the SOURCE address:
//   KC tile [idx][32 floats]: 8 lanes per 128-B row; the 16-B piece that lands in physical quad c' of row r is
//        logical quad c' ^ ((r>>1)&7); the b128 fragment read of logical quad c uses physical c ^ ((r>>1)&7)
//        (conflict-free for the ds_read_b128 lane groups of gfx950: checked by enumeration in DESIGN.md);
//   IC tile [k][64T floats]: rows are already read along idx with b32 -> no swizzle.
// Out-of-range rows / k read a 16-byte zero word instead (the source address is per lane).
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

template <int LAY, int T>
__device__ __forceinline__ void dma_tile(const float* __restrict__ P, int ld, int idx0, int nidx, int k0, int kend,
                                         float* lds_tile, int tid) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (LAY == KC) {                 // pass i: rows 32i + 8w + (lane>>3), physical quad lane&7
#pragma unroll
        for (int i = 0; i < 2 * T; ++i) {
            const int r = 32 * i + 8 * wave + (lane >> 3);
            const int cq = (lane & 7) ^ ((r >> 1) & 7);
            const int idx = idx0 + r, k = k0 + 4 * cq;
            const float* src = (idx < nidx && k < kend) ? P + (size_t)idx * ld + k : g_zero16;
            float* dst = lds_tile + (32 * i + 8 * wave) * 32;          // wave-uniform; lane*16 B is implicit
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    } else {                         // [k][64T]: one wave-instruction = 256 floats = 4/T k-rows... (64T floats per row)
        constexpr int RPI = 256 / (64 * T);          // k-rows per wave-instruction (2 for T=2, 4 for T=1)
#pragma unroll
        for (int i = 0; i < 2 * T; ++i) {
            const int krow = RPI * (4 * i + wave) + lane / (16 * T);
            const int idx = idx0 + 4 * (lane % (16 * T)), k = k0 + krow;
            const float* src = (idx < nidx && k < kend) ? P + (size_t)k * ld + idx : g_zero16;
            float* dst = lds_tile + RPI * (4 * i + wave) * (64 * T);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    }
}

template <int LA, int LB, int TM, int TN>
__global__ __launch_bounds__(256, 2) void k_gemm_dma(GemmArgs g) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int AF = BM * BK, BF = BN * BK;                          // unpadded images
    constexpr int STG = 4 * (32 * TM) * (32 * TN + 4);
    constexpr int SMEM = (2 * (AF + BF) > STG) ? 2 * (AF + BF) : STG;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* const sA0 = smem;
    float* const sB0 = smem + 2 * AF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, q = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const TileId tile = xcd_tile();
    const int i0 = tile.y * BM, j0 = tile.x * BN;
    const int kbeg = tile.z * g.kchunk;
    const int kend = (kbeg + g.kchunk < g.KK) ? kbeg + g.kchunk : g.KK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    dma_tile<LA, TM>(g.A, g.lda, i0, g.MI, kbeg, kend, sA0, tid);
    dma_tile<LB, TN>(g.B, g.ldb, j0, g.NJ, kbeg, kend, sB0, tid);
    __syncthreads();                                                   // drains the DMAs (vmcnt(0)) + barrier
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        if (k0 + BK < kend) {
            dma_tile<LA, TM>(g.A, g.lda, i0, g.MI, k0 + BK, kend, sA0 + (cur ^ 1) * AF, tid);
            dma_tile<LB, TN>(g.B, g.ldb, j0, g.NJ, k0 + BK, kend, sB0 + (cur ^ 1) * BF, tid);
        }
        const float* pa = sA0 + cur * AF;
        const float* pb = sB0 + cur * BF;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float av[TM][4], bv[TN][4];
            if (LA == KC) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int r = 32 * TM * wr + 32 * a + q;
                    const float4 u = *reinterpret_cast<const float4*>(&pa[r * 32 + 4 * ((4 * h + s4) ^ ((r >> 1) & 7))]);
                    av[a][0] = u.x; av[a][1] = u.y; av[a][2] = u.z; av[a][3] = u.w;
                }
            }
            if (LB == KC) {
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int r = 32 * TN * wc + 32 * b + q;
                    const float4 u = *reinterpret_cast<const float4*>(&pb[r * 32 + 4 * ((4 * h + s4) ^ ((r >> 1) & 7))]);
                    bv[b][0] = u.x; bv[b][1] = u.y; bv[b][2] = u.z; bv[b][3] = u.w;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (LA == IC) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) av[a][e] = pa[(16 * h + 4 * s4 + e) * BM + 32 * TM * wr + 32 * a + q];
                }
                if (LB == IC) {
#pragma unroll
                    for (int b = 0; b < TN; ++b) bv[b][e] = pb[(16 * h + 4 * s4 + e) * BN + 32 * TN * wc + 32 * b + q];
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) acc[a][b] = MFMA32(av[a][e], bv[b][e], acc[a][b]);
            }
        }
        __syncthreads();                                               // next tile landed, this one free
        cur ^= 1;
    }
    gemm_epilogue<TM, TN>(g, acc, smem, i0, j0, wave, lane, tile.y, tile.z);
}

// ---------------------------------------------------------------------------------------------------------
// Split-bf16 variant ("bf16x6"): the fp32 MFMA of gfx950 runs at 1/16 of the bf16 MFMA rate, so each fp32 operand
// is split EXACTLY into three bf16 pieces while its tile is staged (x = hi + mid + lo, round-to-nearest at each
// level: 3 x 8 mantissa bits + the sign trick cover the 24-bit significand) and the product is accumulated from the
// six piece products whose magnitude reaches 2^-24 of |a||b|:
//      a*b = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh)   [+ am*bl + al*bm + al*bl  <= 2^-25 |a||b|, dropped]
// Every bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the result has
// fp32-GEMM accuracy (tests/test_gpu_gemm.py compares both paths with an fp64 product) at 6/16 of the MFMA time.
// LDS image per operand: [piece][idx][32 k + 8 pad] bf16 (80-B rows: the ds_read_b128 fragment reads and the
// staging writes are conflict-free); both global layouts are transposed into it by the register staging pass.
constexpr int SBROW = BK + 8;                  // bf16 elements per LDS row

// Branch-free tile fetches for the split path (a tile whose loads sit in 2T if/else blocks defeats the scheduler):
// addresses are CLAMPED into the operand (rows beyond nidx re-read the last row: they only feed output rows / columns
// that the epilogue never stores or counts), and k beyond kend -- possible only in the last, partial stage of a
// contraction that is not a multiple of 32 -- is zeroed by select (FULL = false instantiation).
template <int T, bool FULL>
__device__ __forceinline__ void load_tile_kc4(const float* __restrict__ P, int ld, int idx0, int nidx, int k0, int kend,
                                              float4 (&r)[2 * T], int tid) {
    const int k = k0 + 4 * (tid & 7);
    const int kc = FULL ? k : (k + 3 < kend ? k : kend - 4);        // KK % 4 == 0 and KK >= 4 (checked by the callers)
#pragma unroll
    for (int i = 0; i < 2 * T; ++i) {
        int idx = idx0 + (tid >> 3) + 32 * i;
        idx = idx < nidx ? idx : nidx - 1;
        float4 v = *reinterpret_cast<const float4*>(P + (size_t)idx * ld + kc);
        if (!FULL && k >= kend) v = make_float4(0.f, 0.f, 0.f, 0.f);
        r[i] = v;
    }
}

// IC operands for the split path: thread t owns idx = t % (64T) and T chunks of 8 consecutive k (dword loads,
// coalesced along idx), so that the 8 k of a chunk leave as ONE 16-byte LDS row piece per bf16 plane.
template <int T, bool FULL>
__device__ __forceinline__ void load_tile_ic8(const float* __restrict__ P, int ld, int idx0, int nidx, int k0, int kend,
                                              float (&r)[8 * T], int tid) {
    int idx = idx0 + tid % (64 * T);
    idx = idx < nidx ? idx : nidx - 1;
    const int kc = tid / (64 * T);
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + 8 * (kc + (4 / T) * i) + j;
            const int kk = FULL ? k : (k < kend ? k : kend - 1);
            const float v = P[(size_t)kk * ld + idx];
            r[8 * i + j] = (FULL || k < kend) ? v : 0.f;
        }
}

// Staging is two-phase so that the conversion overlaps the MFMAs: split_tile_* turns the raw fp32 registers of the
// NEXT stage into packed bf16 planes (VALU only, scheduled between the MFMAs of the current stage), write_tile_*
// is the bare LDS store between the two workgroup barriers.  PK = 12T packed registers per operand.
// NP = 3: exact 3-way bf16 split; NP = 2: the two leading bf16 pieces ("bf16x3", common.h); NP = 1: one fp16 plane
// (round to nearest), for the fp16-input MFMA
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pk_f16(float x0, float x1) {
    const f32x2v v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v));
}
// PROI: relu(ps * x + pt) with the constants of the thread's OWN idx (an idx-contiguous operand whose prologue runs over
// idx, not over k: the `a` operand of a weight gradient, a = relu(bn(y)) recomputed from the raw layer output)
// NP = 4: fp16x3 (common.h): two fp16 planes of x * sc (sc = the operand's power-of-two scale)
template <int T, int NP, bool PROI = false>
__device__ __forceinline__ void split_tile_ic8(const float (&r)[8 * T], unsigned (&pk)[12 * T], float ps = 1.f, float pt = 0.f,
                                               float sc = 1.f) {
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x0 = r[8 * i + 2 * j], x1 = r[8 * i + 2 * j + 1];
            if (PROI) { x0 = relu_nan(fmaf(ps, x0, pt)); x1 = relu_nan(fmaf(ps, x1, pt)); }
            if (NP == 4) split_pair_h(x0 * sc, x1 * sc, pk[12 * i + j], pk[12 * i + 4 + j]);
            else if (NP >= 2) split_pair(x0, x1, pk[12 * i + j], pk[12 * i + 4 + j], pk[12 * i + 8 + j]);
            else pk[12 * i + j] = pk_f16(x0, x1);
        }
}
template <int T, int NP>
__device__ __forceinline__ void write_tile_ic8(unsigned short* __restrict__ S, const unsigned (&pk)[12 * T], int tid) {
    constexpr int PLANE = 64 * T * SBROW;
    const int il = tid % (64 * T), kc = tid / (64 * T);
#pragma unroll
    for (int i = 0; i < T; ++i) {
        unsigned short* d = S + il * SBROW + 8 * (kc + (4 / T) * i);
#pragma unroll
        for (int p = 0; p < (NP == 4 ? 2 : NP); ++p)
            *reinterpret_cast<uint4*>(d + p * PLANE) =
                make_uint4(pk[12 * i + 4 * p], pk[12 * i + 4 * p + 1], pk[12 * i + 4 * p + 2], pk[12 * i + 4 * p + 3]);
    }
}

template <bool PRO, int T, int NP>
__device__ __forceinline__ void split_tile_kc4(const float4 (&r)[2 * T], unsigned (&pk)[12 * T], int tid, int k0,
                                               const float* __restrict__ ps, const float* __restrict__ pt, float sc = 1.f) {
#pragma unroll
    for (int i = 0; i < 2 * T; ++i) {
        float v[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
        if (PRO) {
            const int k = 4 * (tid & 7);
            const float4 s = *reinterpret_cast<const float4*>(ps + k0 + k);
            const float4 t = *reinterpret_cast<const float4*>(pt + k0 + k);
            v[0] = relu_nan(fmaf(s.x, v[0], t.x)); v[1] = relu_nan(fmaf(s.y, v[1], t.y));
            v[2] = relu_nan(fmaf(s.z, v[2], t.z)); v[3] = relu_nan(fmaf(s.w, v[3], t.w));
        }
        if (NP == 4) {                                                  // fp16x3: two fp16 planes of x * sc
            split_pair_h(v[0] * sc, v[1] * sc, pk[6 * i], pk[6 * i + 2]);
            split_pair_h(v[2] * sc, v[3] * sc, pk[6 * i + 1], pk[6 * i + 3]);
        } else if (NP >= 2) {
            split_pair(v[0], v[1], pk[6 * i], pk[6 * i + 2], pk[6 * i + 4]);
            split_pair(v[2], v[3], pk[6 * i + 1], pk[6 * i + 3], pk[6 * i + 5]);
        } else {
            pk[6 * i] = pk_f16(v[0], v[1]);
            pk[6 * i + 1] = pk_f16(v[2], v[3]);
        }
    }
}
template <int T, int NP>
__device__ __forceinline__ void write_tile_kc4(unsigned short* __restrict__ S, const unsigned (&pk)[12 * T], int tid) {
    constexpr int PLANE = 64 * T * SBROW;
#pragma unroll
    for (int i = 0; i < 2 * T; ++i) {
        unsigned short* d = S + ((tid >> 3) + 32 * i) * SBROW + 4 * (tid & 7);
#pragma unroll
        for (int p = 0; p < (NP == 4 ? 2 : NP); ++p) *reinterpret_cast<uint2*>(d + p * PLANE) = make_uint2(pk[6 * i + 2 * p], pk[6 * i + 2 * p + 1]);
    }
}


template <int LA, int LB, bool PRO, int TM, int TN, int NP = 3>
__global__ __launch_bounds__(256, NP == 1 ? 3 : 2) void k_gemm_sb(GemmArgs g) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int APL = BM * SBROW, BPL = BN * SBROW;                   // one bf16 plane of each operand (elements)
    // epilogue staging (floats): the fp16-input instantiation stages one 32-row tile at a time (gemm_epilogue_h) and keeps
    // only its single operand plane -> 35 KiB of LDS, three workgroups per CU
    constexpr int STG = NP == 1 ? 4 * 32 * (32 * TN + 4) : 4 * (32 * TM) * (32 * TN + 4);
    constexpr int NPL = NP == 4 ? 2 : NP;                               // planes per operand in LDS (fp16x3: two fp16 planes)
    static_assert(NP != 4 || (LA == IC && LB == IC), "fp16x3 is wired for the weight gradient (both operands idx-contiguous)");
    constexpr int TILE_F = ((NP == 1 ? 1 : NP == 4 ? 2 : 3) * (APL + BPL) * 2 + 3) / 4;   // operand images in floats
    constexpr int SMEM = TILE_F > STG ? TILE_F : STG;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    unsigned short* const sA = reinterpret_cast<unsigned short*>(smem);
    unsigned short* const sB = sA + (NP == 1 ? 1 : NP == 4 ? 2 : 3) * APL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, q = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const TileId tile = xcd_tile();
    float sca = 1.f, scb = 1.f, uns = 1.f;                              // fp16x3: A = dy and B = activations by their power-of-two scales
    if (NP == 4) {
        const int seA = h3_se_wide_of(g.amax), seB = h3_se_of(g.amax_b);
        sca = pow2_biased(seA); scb = pow2_biased(seB); uns = h3_unscale(seA, seB);
    }
    const int i0 = tile.y * BM, j0 = tile.x * BN;
    const int kbeg = tile.z * g.kchunk;
    const int kend = (kbeg + g.kchunk < g.KK) ? kbeg + g.kchunk : g.KK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra4[2 * TM], rb4[2 * TN];
    float ra8[8 * TM], rb8[8 * TN];
    auto fetch = [&](int k0) {
        if (k0 + BK <= kend) {                                         // wave-uniform
            if (LA == KC) load_tile_kc4<TM, true>(g.A, g.lda, i0, g.MI, k0, kend, ra4, tid);
            else load_tile_ic8<TM, true>(g.A, g.lda, i0, g.MI, k0, kend, ra8, tid);
            if (LB == KC) load_tile_kc4<TN, true>(g.B, g.ldb, j0, g.NJ, k0, kend, rb4, tid);
            else load_tile_ic8<TN, true>(g.B, g.ldb, j0, g.NJ, k0, kend, rb8, tid);
        } else {
            if (LA == KC) load_tile_kc4<TM, false>(g.A, g.lda, i0, g.MI, k0, kend, ra4, tid);
            else load_tile_ic8<TM, false>(g.A, g.lda, i0, g.MI, k0, kend, ra8, tid);
            if (LB == KC) load_tile_kc4<TN, false>(g.B, g.ldb, j0, g.NJ, k0, kend, rb4, tid);
            else load_tile_ic8<TN, false>(g.B, g.ldb, j0, g.NJ, k0, kend, rb8, tid);
        }
    };
    unsigned pka[12 * TM], pkb[12 * TN];
    // PRO: k-contiguous A (forward): prologue over k; idx-contiguous A and B (weight gradient): prologue on B over ITS idx
    constexpr bool PROB = PRO && LA == IC && LB == IC;
    float psb = 1.f, ptb = 0.f;
    if (PROB) {
        int jb = j0 + tid % (64 * TN);
        jb = jb < g.NJ ? jb : g.NJ - 1;
        psb = g.pscale[jb]; ptb = g.pshift[jb];
    }
    auto split = [&](int k0) {
        if (LA == KC) split_tile_kc4<PRO, TM, NP>(ra4, pka, tid, k0, g.pscale, g.pshift);
        else split_tile_ic8<TM, NP>(ra8, pka, 1.f, 0.f, sca);
        if (LB == KC) split_tile_kc4<false, TN, NP>(rb4, pkb, tid, k0, nullptr, nullptr);
        else split_tile_ic8<TN, NP, PROB>(rb8, pkb, psb, ptb, scb);
    };
    auto write = [&]() {
        if (LA == KC) write_tile_kc4<TM, NP>(sA, pka, tid);
        else write_tile_ic8<TM, NP>(sA, pka, tid);
        if (LB == KC) write_tile_kc4<TN, NP>(sB, pkb, tid);
        else write_tile_ic8<TN, NP>(sB, pkb, tid);
    };
    auto mfma_block = [&](int kk) {
        bf16x8 af[TM][NPL], bf[TN][NPL];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                af[a][p] = *reinterpret_cast<const bf16x8*>(sA + p * APL + (32 * TM * wr + 32 * a + q) * SBROW + 16 * kk + 8 * h);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                bf[b][p] = *reinterpret_cast<const bf16x8*>(sB + p * BPL + (32 * TN * wc + 32 * b + q) * SBROW + 16 * kk + 8 * h);
        if constexpr (NP == 4) {                                        // fp16x3: (lo,hi) (hi,lo) (hi,hi)
            constexpr int HA[3] = FACL_H3_PA, HB[3] = FACL_H3_PB;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = MFMA_F16(__builtin_bit_cast(f16x8h, af[a][HA[t]]), __builtin_bit_cast(f16x8h, bf[b][HB[t]]), acc[a][b]);
            return;
        }
        if constexpr (NP == 1) {                                        // fp16 inputs: ONE product per multiply-add
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[a][0]),
                                                                       __builtin_bit_cast(f16x8, bf[b][0]), acc[a][b], 0, 0, 0);
            return;
        }
        // smallest terms first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
        constexpr int PA[6] = FACL_SB_PA, PB[6] = FACL_SB_PB, PA3[3] = FACL_SB3_PA, PB3[3] = FACL_SB3_PB;
#pragma unroll
        for (int t = 0; t < (NP == 3 ? 6 : 3); ++t)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = MFMA_BF16(af[a][NP == 3 ? PA[t] : PA3[t]], bf[b][NP == 3 ? PB[t] : PB3[t]], acc[a][b]);
    };
    fetch(kbeg);
    split(kbeg);
    write();
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        // the next stage (the last step re-reads its own tile: harmless, keeps the loop body one straight block)
        const int kn = (k0 + BK < kend) ? k0 + BK : k0;
        fetch(kn);
        __builtin_amdgcn_sched_barrier(0);                             // all loads in flight before the MFMAs
        mfma_block(0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(1);
        split(kn);                                                     // VALU work for the scheduler to sink into the MFMA shadow
        __syncthreads();                                               // every wave has read this stage
        write();
        __syncthreads();
    }
    if constexpr (NP == 4) {                                            // exact rescale (a power of two)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] *= uns;
    }
    if constexpr (NP == 1) gemm_epilogue_h<TM, TN>(g, acc, smem, i0, j0, wave, lane, tile.y, tile.z);
    else gemm_epilogue<TM, TN>(g, acc, smem, i0, j0, wave, lane, tile.y, tile.z);
}

// ---- small problems: split-K INSIDE the workgroup ---------------------------------------------------------------------------
// A 64x64-tile GEMM whose grid is one wave of workgroups (the FC head: 800 or 80 rows) is a serial chain of K/32
// stages per workgroup at ~0.75 us each (split -> LDS -> 12 dependent MFMAs, measured: 4 us + 0.75 us per stage,
// independent of M and N), i.e. pure latency.  Here KG groups of 4 waves take every KG-th stage of the SAME output tile,
// each with its own LDS image, and the partial tiles meet in LDS in fixed group order (deterministic); group 0 runs the
// ordinary epilogue, so bias / statistics / centre term keep working and no slice buffer or second kernel is needed.
template <int LA, int LB, bool PRO, int NP, int KG>
__global__ __launch_bounds__(256 * KG, 1) void k_gemm_sbk(GemmArgs g) {
    // NP = 4 ("self-scaled fp16x3", round 4, PRO = false only): two fp16 planes per operand and three products per stage, each
    // STAGE (a 64 x 32 tile of A and of B, one group's work) scaled by the power of two of its OWN maximum -- no operand scale
    // comes in from outside -- and its 32-deep partial product added to the accumulator with the exact inverse.  These kernels
    // are bound by their split + MFMA instructions: 6 instead of 11 vector instructions per operand pair, 6
    // instead of 12 MFMAs per stage.  A tile-local scale is at least as fine as a tensor-wide one (tile maximum <= tensor maximum).
    constexpr int NPL = NP == 4 ? 2 : NP;                               // planes per operand in LDS
    constexpr int PL = 64 * SBROW;                                      // one plane of one operand (elements)
    constexpr int TILE_F = NPL * 2 * PL * 2 / 4;                        // both operand images of a group, in floats
    constexpr int STG = 4 * 32 * 36;                                    // epilogue staging of group 0 (floats)
    constexpr int MXOFF = KG * TILE_F > STG + (KG - 1) * 4096 ? KG * TILE_F : STG + (KG - 1) * 4096;   // [KG][4][2] stage maxima behind everything
    static_assert(NP != 4 || !PRO, "the self-scaled form takes the maximum of the raw registers: no prologue");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, h = lane >> 5, q = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    unsigned short* const sA = reinterpret_cast<unsigned short*>(smem + grp * TILE_F);
    unsigned short* const sB = sA + NPL * PL;
    const TileId tile = xcd_tile();
    const int i0 = tile.y * 64, j0 = tile.x * 64;
    const int kbeg = tile.z * g.kchunk;
    const int kend = (kbeg + g.kchunk < g.KK) ? kbeg + g.kchunk : g.KK;
    const int nst = (kend - kbeg + BK - 1) / BK, nit = (nst + KG - 1) / KG;

    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

    float4 ra4[2], rb4[2];
    float ra8[8], rb8[8];
    auto fetch = [&](int k0) {
        if (k0 + BK <= kend) {                                         // wave-uniform
            if (LA == KC) load_tile_kc4<1, true>(g.A, g.lda, i0, g.MI, k0, kend, ra4, tid);
            else load_tile_ic8<1, true>(g.A, g.lda, i0, g.MI, k0, kend, ra8, tid);
            if (LB == KC) load_tile_kc4<1, true>(g.B, g.ldb, j0, g.NJ, k0, kend, rb4, tid);
            else load_tile_ic8<1, true>(g.B, g.ldb, j0, g.NJ, k0, kend, rb8, tid);
        } else {
            if (LA == KC) load_tile_kc4<1, false>(g.A, g.lda, i0, g.MI, k0, kend, ra4, tid);
            else load_tile_ic8<1, false>(g.A, g.lda, i0, g.MI, k0, kend, ra8, tid);
            if (LB == KC) load_tile_kc4<1, false>(g.B, g.ldb, j0, g.NJ, k0, kend, rb4, tid);
            else load_tile_ic8<1, false>(g.B, g.ldb, j0, g.NJ, k0, kend, rb8, tid);
        }
    };
    unsigned pka[12], pkb[12];
    float* const mxs = smem + MXOFF;
    float uns_nxt = 1.f;                                                // NP = 4: inverse scale of the stage `split` has just prepared
    auto split = [&](int k0) {
        float sca = 1.f, scb = 1.f;
        if constexpr (NP == 4) {
            // maxima of the group's raw stage tiles: registers -> wave (DPP) -> the group's four waves (LDS, behind a workgroup
            // barrier that ALSO is the one after which the current LDS tile may be overwritten: every wave's MFMA reads are done)
            float ma = 0.f, mb = 0.f;
            if (LA == KC) {
#pragma unroll
                for (int i = 0; i < 2; ++i) ma = fmaxf(fmaxf(ma, fmaxf(fabsf(ra4[i].x), fabsf(ra4[i].y))), fmaxf(fabsf(ra4[i].z), fabsf(ra4[i].w)));
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) ma = fmaxf(ma, fabsf(ra8[i]));
            }
            if (LB == KC) {
#pragma unroll
                for (int i = 0; i < 2; ++i) mb = fmaxf(fmaxf(mb, fmaxf(fabsf(rb4[i].x), fabsf(rb4[i].y))), fmaxf(fabsf(rb4[i].z), fabsf(rb4[i].w)));
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) mb = fmaxf(mb, fabsf(rb8[i]));
            }
            ma = facl_wave_max_nonneg(ma); mb = facl_wave_max_nonneg(mb);
            if (lane == 0) { mxs[(grp * 4 + wave) * 2] = ma; mxs[(grp * 4 + wave) * 2 + 1] = mb; }
            __syncthreads();
            const float4 m01 = *reinterpret_cast<const float4*>(mxs + grp * 8), m23 = *reinterpret_cast<const float4*>(mxs + grp * 8 + 4);
            ma = fmaxf(fmaxf(m01.x, m01.z), fmaxf(m23.x, m23.z));
            mb = fmaxf(fmaxf(m01.y, m01.w), fmaxf(m23.y, m23.w));
            const int seA = h3_se(__float_as_uint(ma)), seB = h3_se(__float_as_uint(mb));      // NaN / inf pass through as such
            sca = pow2_biased(seA); scb = pow2_biased(seB);
            uns_nxt = h3_unscale(seA, seB);
        }
        if (LA == KC) split_tile_kc4<PRO, 1, NP>(ra4, pka, tid, k0, g.pscale, g.pshift, sca);
        else split_tile_ic8<1, NP>(ra8, pka, 1.f, 0.f, sca);
        if (LB == KC) split_tile_kc4<false, 1, NP>(rb4, pkb, tid, k0, nullptr, nullptr, scb);
        else split_tile_ic8<1, NP>(rb8, pkb, 1.f, 0.f, scb);
    };
    auto write = [&]() {
        if (LA == KC) write_tile_kc4<1, NP>(sA, pka, tid);
        else write_tile_ic8<1, NP>(sA, pka, tid);
        if (LB == KC) write_tile_kc4<1, NP>(sB, pkb, tid);
        else write_tile_ic8<1, NP>(sB, pkb, tid);
    };
    f32x16 stg_acc;                                                     // NP = 4: the current stage's partial product (scaled)
    auto mfma_block = [&](int kk) {
        bf16x8 af[NPL], bf[NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            af[p] = *reinterpret_cast<const bf16x8*>(sA + p * PL + (32 * wr + q) * SBROW + 16 * kk + 8 * h);
            bf[p] = *reinterpret_cast<const bf16x8*>(sB + p * PL + (32 * wc + q) * SBROW + 16 * kk + 8 * h);
        }
        if constexpr (NP == 4) {                                        // smallest terms first: (lo,hi) (hi,lo) (hi,hi)
            const f16x8h a0 = __builtin_bit_cast(f16x8h, af[0]), a1 = __builtin_bit_cast(f16x8h, af[1]);
            const f16x8h b0 = __builtin_bit_cast(f16x8h, bf[0]), b1 = __builtin_bit_cast(f16x8h, bf[1]);
            stg_acc = MFMA_F16(a1, b0, stg_acc); stg_acc = MFMA_F16(a0, b1, stg_acc); stg_acc = MFMA_F16(a0, b0, stg_acc);
        } else if constexpr (NP == 1) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, bf[0]),
                                                               acc[0][0], 0, 0, 0);
        } else {
            constexpr int PA[6] = FACL_SB_PA, PB[6] = FACL_SB_PB, PA3[3] = FACL_SB3_PA, PB3[3] = FACL_SB3_PB;
#pragma unroll
            for (int t = 0; t < (NP == 3 ? 6 : 3); ++t)
                acc[0][0] = MFMA_BF16(af[NP == 3 ? PA[t] : PA3[t]], bf[NP == 3 ? PB[t] : PB3[t]], acc[0][0]);
        }
    };
    // stage s of the chunk belongs to group s % KG; a group that runs out of stages re-stages its last one (harmless)
    // and skips the MFMAs, so that every wave reaches every barrier
    auto kof = [&](int it) {
        int s = grp + KG * it;
        s = s < nst ? s : nst - 1;
        return kbeg + s * BK;
    };
    { const int k0 = kof(0); fetch(k0); split(k0); write(); }
    __syncthreads();
    float uns_cur = uns_nxt;                                            // NP = 4: inverse scale of the stage now in LDS
    for (int it = 0; it < nit; ++it) {
        const int kn = kof(it + 1 < nit ? it + 1 : it);
        fetch(kn);
        if (grp + KG * it < nst) {                                     // wave-uniform
            if constexpr (NP == 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) stg_acc[r] = 0.f;
            }
            mfma_block(0);
            mfma_block(1);
            if constexpr (NP == 4) {                                    // exact rescale (a power of two), then the ordinary fp32 add
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] = fmaf(stg_acc[r], uns_cur, acc[0][0][r]);
            }
        }
        split(kn);                                                      // (NP = 4: holds the barrier behind the MFMA reads)
        if constexpr (NP != 4) __syncthreads();
        write();
        __syncthreads();
        uns_cur = uns_nxt;
    }
    // partial tiles -> LDS (behind group 0's staging area), summed by group 0 in group order
    float* red = smem + STG;
    if (grp > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(grp - 1) * 4096 + wave * 1024 + r * 64 + lane] = acc[0][0][r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int gg = 0; gg < KG - 1; ++gg)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += red[gg * 4096 + wave * 1024 + r * 64 + lane];
    gemm_epilogue<1, 1>(g, acc, smem, i0, j0, wave, lane, tile.y, tile.z);
}

constexpr int SBK_KG = 4;
template <int LA, int LB, bool PRO, int NP>
int launch_sbk(const GemmArgs& g, int nz, hipStream_t st) {
    constexpr int TILE_F = (NP == 4 ? 2 : NP) * 2 * 64 * SBROW * 2 / 4, STG = 4 * 32 * 36;
    constexpr int F = SBK_KG * TILE_F > STG + (SBK_KG - 1) * 4096 ? SBK_KG * TILE_F : STG + (SBK_KG - 1) * 4096;
    constexpr int lds = (F + SBK_KG * 8) * 4;                           // + the stage maxima of the self-scaled form
    static bool attr_done[64] = {};                   // per template instantiation and device ordinal
    const void* fns[1] = {(const void*)k_gemm_sbk<LA, LB, PRO, NP, SBK_KG>};
    if (int rc = facl_set_dynamic_lds(attr_done, fns, 1, lds)) return rc;
    dim3 grid((g.NJ + 63) / 64, (g.MI + 63) / 64, nz);
    hipLaunchKernelGGL((k_gemm_sbk<LA, LB, PRO, NP, SBK_KG>), grid, dim3(256 * SBK_KG), lds, st, g);
    return facl_launch_status();
}
// one wave of 64x64 workgroups and enough stages to share out
static inline bool sbk_fits(const GemmArgs& g, int nz) {
    static const int off = getenv("FACL_GEMM_NOSBK") ? atoi(getenv("FACL_GEMM_NOSBK")) : 0;
    const long long t64 = (long long)((g.NJ + 63) / 64) * ((g.MI + 63) / 64) * nz;
    return !off && t64 <= 320 && g.kchunk >= 2 * SBK_KG * BK;
}

// sum over split-K slices: out[e] = sum_z part[z][e], slices added in order (deterministic); four slice loads in flight
// per thread (a serial chain of nz dependent 16-byte loads per thread ran at a third of the HBM rate)
__global__ void k_sum_slices(const float* __restrict__ part, int nz, long long n4, float* __restrict__ out) {
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


// The same with the slices spread over four thread groups of the workgroup (round 4): thread (column, phase) adds a contiguous
// quarter of the slices in order, the four partial sums meet in LDS and are added in phase order -- still one fixed order, so
// deterministic, with four times the loads in flight per output (the serial form ran at 2.6-4 TB/s on 33-67 MB of slices).
// FACL_SUM_SLICES_PAR=0 restores the serial kernel (A/B).
__global__ __launch_bounds__(256) void k_sum_slices_par(const float* __restrict__ part, int nz, long long n4, float* __restrict__ out) {
    __shared__ float4 red[3][64];
    const int col = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + col;
    const float4* p4 = reinterpret_cast<const float4*>(part);
    const int per = (nz + 3) >> 2;
    const int z0 = ph * per, z1 = z0 + per < nz ? z0 + per : nz;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        int z = z0;
        for (; z + 3 < z1; z += 4) {
            const float4 v0 = p4[(size_t)z * n4 + i], v1 = p4[(size_t)(z + 1) * n4 + i];
            const float4 v2 = p4[(size_t)(z + 2) * n4 + i], v3 = p4[(size_t)(z + 3) * n4 + i];
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; z < z1; ++z) {
            const float4 v = p4[(size_t)z * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (ph) red[ph - 1][col] = s;
    __syncthreads();
    if (ph == 0 && i < n4) {
#pragma unroll
        for (int t = 0; t < 3; ++t) { const float4 v = red[t][col]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        reinterpret_cast<float4*>(out)[i] = s;
    }
}

static int launch_k_sum_slices(const float* slices, int nz, long long n4, float* dW, hipStream_t st) {
    static const int par = getenv("FACL_SUM_SLICES_PAR") ? atoi(getenv("FACL_SUM_SLICES_PAR")) : 1;
    if (par && nz >= 8) {
        hipLaunchKernelGGL(k_sum_slices_par, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, slices, nz, n4, dW);
    } else {
        const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_sum_slices, dim3(grid), dim3(256), 0, st, slices, nz, n4, dW);
    }
    return facl_launch_status();
}

template <int LA, int LB, bool PRO, int NP>
int launch_sb_np(const GemmArgs& g, int nz, long long big, hipStream_t st, int* rows_per_part) {
    if (sbk_fits(g, nz)) {
        if (rows_per_part) *rows_per_part = 32;
        return launch_sbk<LA, LB, PRO, NP>(g, nz, st);
    }
    if (big >= 256) {
        dim3 grid((g.NJ + 127) / 128, (g.MI + 127) / 128, nz);
        hipLaunchKernelGGL((k_gemm_sb<LA, LB, PRO, 2, 2, NP>), grid, dim3(256), 0, st, g);
        if (rows_per_part) *rows_per_part = 64;
    } else {
        dim3 grid((g.NJ + 63) / 64, (g.MI + 63) / 64, nz);
        hipLaunchKernelGGL((k_gemm_sb<LA, LB, PRO, 1, 1, NP>), grid, dim3(256), 0, st, g);
        if (rows_per_part) *rows_per_part = 32;
    }
    return facl_launch_status();
}

// tile choice: 128x128 blocks when they already fill the chip, else 64x64 blocks (4x the workgroups) -- the FC head
// (M = 768 or 32 rows) would otherwise run on 48 or 8 of the 256 CUs
template <int LA, int LB, bool PRO>
int launch(const GemmArgs& g, int nz, hipStream_t st, int* rows_per_part) {
    const long long big = (long long)((g.NJ + 127) / 128) * ((g.MI + 127) / 128) * nz;
    static const int use_dma = getenv("FACL_GEMM_DMA") ? atoi(getenv("FACL_GEMM_DMA")) : 1;
    // FACL_GEMM_F32=1 selects the exact-fp32 MFMA kernels (v_mfma_f32_32x32x2_f32) instead of the split-bf16 ones
    static const int use_f32 = getenv("FACL_GEMM_F32") ? atoi(getenv("FACL_GEMM_F32")) : 0;
    if (g.prec == 1) return launch_sb_np<LA, LB, PRO, 1>(g, nz, big, st, rows_per_part);   // fp16-input MFMA, fp32 accumulation
    if (g.prec == 2) return launch_sb_np<LA, LB, PRO, 2>(g, nz, big, st, rows_per_part);   // bf16x3 (opt-in)
    if (!use_f32) {
        if (sbk_fits(g, nz)) {
            if (rows_per_part) *rows_per_part = 32;
            // OPT-IN FACL_SBK_H3=1: self-scaled fp16x3 (k_gemm_sbk, NP = 4; not with a prologue).  Measured inside the step: 2.938 vs
            // 2.946 ms (gpurun_out/r7b_ab.log), stand-alone -7..-9 % at K = 1024 -- the split is ~20 % cheaper, not 2.7x: the maximum
            // exchange, the per-stage rescale and the accumulator reset take back most of what the shorter split gives.  Default: bf16x6.
            static const int sbk_h3 = getenv("FACL_SBK_H3") ? atoi(getenv("FACL_SBK_H3")) : 0;
            if constexpr (!PRO) {
                if (sbk_h3) return launch_sbk<LA, LB, false, 4>(g, nz, st);
            }
            return launch_sbk<LA, LB, PRO, 3>(g, nz, st);
        }
        if (big >= 256) {
            dim3 grid((g.NJ + 127) / 128, (g.MI + 127) / 128, nz);
            hipLaunchKernelGGL((k_gemm_sb<LA, LB, PRO, 2, 2>), grid, dim3(256), 0, st, g);
            if (rows_per_part) *rows_per_part = 64;
        } else {
            dim3 grid((g.NJ + 63) / 64, (g.MI + 63) / 64, nz);
            hipLaunchKernelGGL((k_gemm_sb<LA, LB, PRO, 1, 1>), grid, dim3(256), 0, st, g);
            if (rows_per_part) *rows_per_part = 32;
        }
        return facl_launch_status();
    }
    // LDS-DMA needs 16-byte aligned 4-element pieces: leading dimensions and extents multiples of 4
    // Measured A/B in one process (49152-row layers): the DMA path wins when BOTH operands are idx-contiguous
    // (wgrad: 0.431 vs 0.538 ms at 1024x512) and loses a few % when a k-contiguous operand needs the
    // source-side swizzle (forward 0.564 vs 0.552, dgrad 0.475 vs 0.460) -> used for wgrad only.
    const bool dma_ok = use_dma && !PRO && LA == IC && LB == IC && !(g.lda & 3) && !(g.ldb & 3) && !(g.MI & 3) &&
                        !(g.NJ & 3) && !(((uintptr_t)g.A | (uintptr_t)g.B) & 15);
    if (big >= 256) {
        dim3 grid((g.NJ + 127) / 128, (g.MI + 127) / 128, nz);
        if (dma_ok) hipLaunchKernelGGL((k_gemm_dma<LA, LB, 2, 2>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((k_gemm<LA, LB, PRO, 2, 2>), grid, dim3(256), 0, st, g);
        if (rows_per_part) *rows_per_part = 64;
    } else {
        dim3 grid((g.NJ + 63) / 64, (g.MI + 63) / 64, nz);
        if (dma_ok) hipLaunchKernelGGL((k_gemm_dma<LA, LB, 1, 1>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((k_gemm<LA, LB, PRO, 1, 1>), grid, dim3(256), 0, st, g);
        if (rows_per_part) *rows_per_part = 32;
    }
    return facl_launch_status();
}

}  // namespace

// y (M,N) = opA(a) W^T + bias (+ centres term), optional BN+ReLU prologue on a, optional column statistics
static int gemm_fwd_p(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                      const float* pscale, const float* pshift, const float* centers, const float* Wc,
                      int ldwc, float* y, double* sums, void* ws, void* stream, int prec) {
    if (!a || !W || !y || (sums && !ws)) return FACL_E_NULL;
    if (M < 1 || M > 0x7fffffff || K < 4 || (K & 3) || N < 1 || (ldw & 3)) return FACL_E_SHAPE;
    if ((pscale == nullptr) != (pshift == nullptr) || (centers == nullptr) != (Wc == nullptr)) return FACL_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g{a, K, W, ldw, y, N, (int)M, N, K, bias, pscale, pshift, centers, Wc, ldwc,
               sums ? (double*)ws : nullptr, K, nullptr, nullptr, nullptr, prec};
    int rpp = 64;
    if (sums) {                                                        // the statistics partials must fit BEFORE anything is written
        const long long big = (long long)((N + 127) / 128) * ((M + 127) / 128);
        const int rpp0 = (big >= 256 && !sbk_fits(g, 1)) ? 64 : 32;
        const long long prow0 = ((M + 2 * rpp0 - 1) / (2 * rpp0)) * 2;
        if ((size_t)prow0 * N * 2 * sizeof(double) > ((size_t)facl_ws_bytes() - FACL_WS_TICKET_BYTES)) return FACL_E_SHAPE;
    }
    int rc = pscale ? launch<KC, KC, true>(g, 1, st, &rpp) : launch<KC, KC, false>(g, 1, st, &rpp);
    if (rc || !sums) return rc;
    const int prow = (int)((M + 2 * rpp - 1) / (2 * rpp)) * 2;
    if ((size_t)prow * N * 2 * sizeof(double) > ((size_t)facl_ws_bytes() - FACL_WS_TICKET_BYTES)) return FACL_E_SHAPE;
    // rows of the last (partial) tile that no wave wrote hold stale data only if M % 64 != 0 for the last
    // wave-row; such partial rows contribute nothing because those waves stored s = sq = 0.
    return facl_reduce_rows((const double*)ws, prow, 2 * N, sums, st);
}

// facl_gemm_fwd + my_max_pool over blocks of S = 64 consecutive rows fused into the epilogue: ymax (M/64,N) =
// max_s sgn[j]*y, arg = first s that attains it (BN + ReLU are monotone per channel, so the pooled activation follows
// from ymax once the statistics are final: facl_sa_pool).  Saves the 201 MB re-read of y by facl_rows_segmax.
static int gemm_fwd_segmax_p(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                             const float* sgn, float* y, double* sums, float* ymax, int32_t* arg, void* ws,
                             void* stream, int prec) {
    if (!a || !W || !y || !sgn || !ymax || !arg || (sums && !ws)) return FACL_E_NULL;
    if (M < 64 || M > 0x7fffffff || (M & 63) || K < 4 || (K & 3) || N < 1 || (ldw & 3)) return FACL_E_SHAPE;
    if ((long long)((N + 127) / 128) * ((M + 127) / 128) < 256) return FACL_E_CONFIG;    // needs the 128x128 tiles
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g{a, K, W, ldw, y, N, (int)M, N, K, bias, nullptr, nullptr, nullptr, nullptr, 0,
               sums ? (double*)ws : nullptr, K, sgn, ymax, arg, prec};
    int rpp = 64;
    if (sums && (size_t)(((M + 127) / 128) * 2) * N * 2 * sizeof(double) > ((size_t)facl_ws_bytes() - FACL_WS_TICKET_BYTES)) return FACL_E_SHAPE;
    int rc = launch<KC, KC, false>(g, 1, st, &rpp);
    if (rc || !sums) return rc;
    const int prow = (int)((M + 2 * rpp - 1) / (2 * rpp)) * 2;
    if ((size_t)prow * N * 2 * sizeof(double) > ((size_t)facl_ws_bytes() - FACL_WS_TICKET_BYTES)) return FACL_E_SHAPE;
    return facl_reduce_rows((const double*)ws, prow, 2 * N, sums, st);
}

// da (M,K) = dy (M,N) W (N,K)        (W row-major with leading dimension ldw; pass W + offset to skip columns)
static int gemm_dgrad_p(const float* dy, int64_t M, int N, const float* W, int ldw, int K, float* da,
                        void* stream, int prec) {
    if (!dy || !W || !da) return FACL_E_NULL;
    if (M < 1 || M > 0x7fffffff || N < 4 || (N & 3) || K < 1) return FACL_E_SHAPE;
    GemmArgs g{dy, N, W, ldw, da, K, (int)M, K, N, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, N, nullptr, nullptr, nullptr, prec};
    return launch<KC, IC, false>(g, 1, (hipStream_t)stream, nullptr);
}

// dW (N,K) = dy^T (N,M) a (M,K), split over nz row chunks; `slices` is scratch for nz*N*K floats
static int gemm_wgrad_p(const float* dy, const float* a, int64_t M, int N, int K, int lda, float* dW,
                        float* slices, int nz, void* stream, int prec) {
    if (!dy || !a || !dW || !slices) return FACL_E_NULL;
    if (M < 1 || M > 0x7fffffff || N < 4 || (N & 3) || K < 4 || (K & 3) || nz < 1 || nz > 1024) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    {   // few rows (the FC head): the workgroup splits the contraction itself and writes dW directly, no slices
        GemmArgs g1{dy, N, a, lda, dW, K, N, K, (int)M, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, (int)M, nullptr, nullptr, nullptr, prec};
        if (M <= 8192 && sbk_fits(g1, 1)) return launch<IC, IC, false>(g1, 1, st, nullptr);
    }
    int kchunk = (int)((M + nz - 1) / nz);
    kchunk = (kchunk + BK - 1) / BK * BK;
    nz = (int)((M + kchunk - 1) / kchunk);
    GemmArgs g{dy, N, a, lda, slices, K, N, K, (int)M, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, kchunk, nullptr, nullptr, nullptr, prec};
    int rc = launch<IC, IC, false>(g, nz, st, nullptr);
    if (rc) return rc;
    const long long n4 = (long long)N * K / 4;
    return launch_k_sum_slices(slices, nz, n4, dW, st);
}

// dW (N,K) = dy^T (N,M) relu(pscale * y + pshift) (M,K): the weight gradient of a layer whose input activation is the
// BatchNorm + ReLU of the previous layer's raw output y -- recomputed while the tile is staged, never materialised
// (the forward does the same in its prologue: csrc/gemm_rs.hip).  Large row counts only (the 128x128-tile kernel):
// FACL_E_CONFIG otherwise, callers then materialise the activation and use facl_gemm_wgrad.
static int gemm_wgrad_pro_p(const float* dy, const float* y, int64_t M, int N, int K, int ldy, const float* pscale,
                            const float* pshift, float* dW, float* slices, int nz, void* stream, int prec) {
    if (!dy || !y || !pscale || !pshift || !dW || !slices) return FACL_E_NULL;
    if (M < 1 || M > 0x7fffffff || N < 4 || (N & 3) || K < 4 || (K & 3) || nz < 1 || nz > 1024) return FACL_E_SHAPE;
    static const int use_f32 = getenv("FACL_GEMM_F32") ? atoi(getenv("FACL_GEMM_F32")) : 0;
    if (use_f32 || prec == 1) return FACL_E_CONFIG;
    hipStream_t st = (hipStream_t)stream;
    int kchunk = (int)((M + nz - 1) / nz);
    kchunk = (kchunk + BK - 1) / BK * BK;
    nz = (int)((M + kchunk - 1) / kchunk);
    GemmArgs g{dy, N, y, ldy, slices, K, N, K, (int)M, nullptr, pscale, pshift, nullptr, nullptr, 0, nullptr, kchunk, nullptr, nullptr, nullptr, prec};
    const long long big = (long long)((K + 127) / 128) * ((N + 127) / 128) * nz;
    if (big < 256 || sbk_fits(g, nz)) return FACL_E_CONFIG;
    dim3 grid((K + 127) / 128, (N + 127) / 128, nz);
    if (prec == 2) hipLaunchKernelGGL((k_gemm_sb<IC, IC, true, 2, 2, 2>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((k_gemm_sb<IC, IC, true, 2, 2, 3>), grid, dim3(256), 0, st, g);
    int rc = facl_launch_status();
    if (rc) return rc;
    const long long n4 = (long long)N * K / 4;
    return launch_k_sum_slices(slices, nz, n4, dW, st);
}

// fp16x3 weight gradient on the 128x128-tile kernel: dW (N,K) = dy^T f(y), f = relu(pscale*y + pshift) per column when pscale
// is given, else identity; dy by the power-of-two scale read from `amax` (facl_rows_bwd_apply_amax), f(y) by the one read
// from `amax_b` (the bound the forward GEMM that consumed f(y) was given).  FACL_E_CONFIG when the shape is not served
// (callers use the bf16x6 entries).
extern "C" int facl_gemm_wgrad_h3(const float* dy, const float* y, int64_t M, int N, int K, int ldy, const float* pscale,
                                  const float* pshift, const uint32_t* amax, const uint32_t* amax_b, float* dW, float* slices,
                                  int nz, void* stream) {
    if (!dy || !y || !amax || !amax_b || !dW || !slices || (pscale == nullptr) != (pshift == nullptr)) return FACL_E_NULL;
    if (M < 1 || M > 0x7fffffff || N < 4 || (N & 3) || K < 4 || (K & 3) || nz < 1 || nz > 1024) return FACL_E_SHAPE;
    static const int use_f32 = getenv("FACL_GEMM_F32") ? atoi(getenv("FACL_GEMM_F32")) : 0;
    if (use_f32) return FACL_E_CONFIG;
    hipStream_t st = (hipStream_t)stream;
    int kchunk = (int)((M + nz - 1) / nz);
    kchunk = (kchunk + BK - 1) / BK * BK;
    nz = (int)((M + kchunk - 1) / kchunk);
    GemmArgs g{dy, N, y, ldy, slices, K, N, K, (int)M, nullptr, pscale, pshift, nullptr, nullptr, 0, nullptr, kchunk, nullptr, nullptr, nullptr, 0, amax, amax_b};
    const long long big = (long long)((K + 127) / 128) * ((N + 127) / 128) * nz;
    if (big < 256 || sbk_fits(g, nz)) return FACL_E_CONFIG;
    dim3 grid((K + 127) / 128, (N + 127) / 128, nz);
    if (pscale) hipLaunchKernelGGL((k_gemm_sb<IC, IC, true, 2, 2, 4>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((k_gemm_sb<IC, IC, false, 2, 2, 4>), grid, dim3(256), 0, st, g);
    int rc = facl_launch_status();
    if (rc) return rc;
    const long long n4 = (long long)N * K / 4;
    return launch_k_sum_slices(slices, nz, n4, dW, st);
}

extern "C" int facl_gemm_wgrad_pro(const float* dy, const float* y, int64_t M, int N, int K, int ldy, const float* pscale,
                                   const float* pshift, float* dW, float* slices, int nz, void* stream) {
    return gemm_wgrad_pro_p(dy, y, M, N, K, ldy, pscale, pshift, dW, slices, nz, stream, 0);
}
extern "C" int facl_gemm_wgrad_pro_x3(const float* dy, const float* y, int64_t M, int N, int K, int ldy, const float* pscale,
                                      const float* pshift, float* dW, float* slices, int nz, void* stream) {
    return gemm_wgrad_pro_p(dy, y, M, N, K, ldy, pscale, pshift, dW, slices, nz, stream, 2);
}

// ---- C ABI: fp32-result entries and their fp16-input twins (same arguments; inputs rounded to fp16 while staged, one
// v_mfma_f32_32x32x16_f16 product per multiply-add, fp32 accumulation and storage) ----------------------------------------
extern "C" int facl_gemm_fwd(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                             const float* pscale, const float* pshift, const float* centers, const float* Wc,
                             int ldwc, float* y, double* sums, void* ws, void* stream) {
    return gemm_fwd_p(a, M, K, W, ldw, N, bias, pscale, pshift, centers, Wc, ldwc, y, sums, ws, stream, 0);
}
extern "C" int facl_gemm_fwd_f16(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                                 const float* pscale, const float* pshift, const float* centers, const float* Wc,
                                 int ldwc, float* y, double* sums, void* ws, void* stream) {
    return gemm_fwd_p(a, M, K, W, ldw, N, bias, pscale, pshift, centers, Wc, ldwc, y, sums, ws, stream, 1);
}
extern "C" int facl_gemm_fwd_segmax(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                                    const float* sgn, float* y, double* sums, float* ymax, int32_t* arg, void* ws,
                                    void* stream) {
    return gemm_fwd_segmax_p(a, M, K, W, ldw, N, bias, sgn, y, sums, ymax, arg, ws, stream, 0);
}
extern "C" int facl_gemm_fwd_segmax_f16(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                                        const float* sgn, float* y, double* sums, float* ymax, int32_t* arg, void* ws,
                                        void* stream) {
    return gemm_fwd_segmax_p(a, M, K, W, ldw, N, bias, sgn, y, sums, ymax, arg, ws, stream, 1);
}
extern "C" int facl_gemm_dgrad(const float* dy, int64_t M, int N, const float* W, int ldw, int K, float* da, void* stream) {
    return gemm_dgrad_p(dy, M, N, W, ldw, K, da, stream, 0);
}
extern "C" int facl_gemm_dgrad_f16(const float* dy, int64_t M, int N, const float* W, int ldw, int K, float* da, void* stream) {
    return gemm_dgrad_p(dy, M, N, W, ldw, K, da, stream, 1);
}
extern "C" int facl_gemm_wgrad(const float* dy, const float* a, int64_t M, int N, int K, int lda, float* dW,
                               float* slices, int nz, void* stream) {
    return gemm_wgrad_p(dy, a, M, N, K, lda, dW, slices, nz, stream, 0);
}
extern "C" int facl_gemm_wgrad_f16(const float* dy, const float* a, int64_t M, int N, int K, int lda, float* dW,
                                   float* slices, int nz, void* stream) {
    return gemm_wgrad_p(dy, a, M, N, K, lda, dW, slices, nz, stream, 1);
}

// ---- "bf16x3" twins (opt-in precision "x3": two bf16 pieces per operand, three products; common.h) -------------------------
extern "C" int facl_gemm_fwd_x3(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                                const float* pscale, const float* pshift, const float* centers, const float* Wc,
                                int ldwc, float* y, double* sums, void* ws, void* stream) {
    return gemm_fwd_p(a, M, K, W, ldw, N, bias, pscale, pshift, centers, Wc, ldwc, y, sums, ws, stream, 2);
}
extern "C" int facl_gemm_fwd_segmax_x3(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                                       const float* sgn, float* y, double* sums, float* ymax, int32_t* arg, void* ws,
                                       void* stream) {
    return gemm_fwd_segmax_p(a, M, K, W, ldw, N, bias, sgn, y, sums, ymax, arg, ws, stream, 2);
}
extern "C" int facl_gemm_dgrad_x3(const float* dy, int64_t M, int N, const float* W, int ldw, int K, float* da, void* stream) {
    return gemm_dgrad_p(dy, M, N, W, ldw, K, da, stream, 2);
}
extern "C" int facl_gemm_wgrad_x3(const float* dy, const float* a, int64_t M, int N, int K, int lda, float* dW,
                                  float* slices, int nz, void* stream) {
    return gemm_wgrad_p(dy, a, M, N, K, lda, dW, slices, nz, stream, 2);
}

// ---- round 4, late (prec: 0 = fp32-grade, 1 = fp16 inputs, 2 = bf16x3) -----------------------------------------------------------------
// dW (N,K) += dy^T a: the weight-gradient GEMM accumulating INTO its output -- the second gradient path of the loss's keys
// (utils_my._ContrastivePair.backward: d x += dsim^T @ stacked) without a separate add launch.  Few rows only (the
// workgroup-level split-K kernel writes its tile once): FACL_E_CONFIG otherwise, the caller then adds itself.
extern "C" int facl_gemm_wgrad_acc(const float* dy, const float* a, int64_t M, int N, int K, int lda, float* dW, int prec,
                                   void* stream) {
    if (!dy || !a || !dW) return FACL_E_NULL;
    if (M < 1 || M > 8192 || N < 4 || (N & 3) || K < 4 || (K & 3) || prec < 0 || prec > 2) return FACL_E_SHAPE;
    GemmArgs g1{dy, N, a, lda, dW, K, N, K, (int)M, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, (int)M, nullptr, nullptr, nullptr, prec};
    g1.accum = 1;
    if (!sbk_fits(g1, 1)) return FACL_E_CONFIG;
    return launch<IC, IC, false>(g1, 1, (hipStream_t)stream, nullptr);
}
