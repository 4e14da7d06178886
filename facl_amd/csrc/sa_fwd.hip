// Set-abstraction point-MLP, forward (net3DV_1: cn3d_model_conbag.py:43-58 / :162-177):
//   3 x [1x1 conv -> BatchNorm2d -> ReLU]  D -> 64 -> 64 -> 256, then max over the K neighbours.
//
// Train-mode BN needs the statistics of every layer over ALL positions before that layer's ReLU,
// so the forward is a short pipeline of passes; nothing wider than 64 channels ever reaches HBM:
//   facl_sa_x_moments : sum x, sum x x^T  (BN1 statistics follow analytically: y1 is affine in x)
//   facl_sa_fwd2      : x -> a1 (VALU) -> y2 = a1 W2^T + b2 (MFMA) ; stores y2 ("fragment layout"),
//                       accumulates sum / sumsq of y2
//   facl_sa_fwd3      : y2 -> a2 -> y3 = a2 W3^T + b3 (MFMA) ; accumulates sum / sumsq of y3 and
//                       keeps only max_k (sgn*y3) + argmax per (group, channel): BN+ReLU are monotone
//                       per channel, so pooled = relu(|scale| * max(sgn*y3) + shift), sgn = sign(gamma)
//   facl_sa_pool      : that last elementwise step.
// Eval mode runs fwd2 + fwd3 + pool with constants folded from the running statistics.
//
// A "unit" is 64 consecutive positions (= one group when K = 64).  One wave owns one unit at a time.
// Roofline: MFMA fp32 (157.3 TFLOP/s).  FLOPs per unit: fwd2 2*64*64*64, fwd3 2*64*64*256.
#include "common.h"
#include <stdlib.h>

int facl_reduce_rows(const double* part, int rows, int V, double* out, hipStream_t st);

namespace {

constexpr int SA_GRID = 256;   // one workgroup per CU; waves grid-stride over the units

// ------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_x_moments(const float* __restrict__ x, long long P,
                                                   double* __restrict__ part) {
    constexpr int V = D + D * D;
    double acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    auto load = [&](long long p, float (&v)[D]) {
        if (D == 4) {
            const float4 t = *reinterpret_cast<const float4*>(x + p * 4);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[D - 1] = t.w;
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) v[i] = x[p * D + i];
        }
    };
    auto add = [&](const float (&v)[D]) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            acc[i] += (double)v[i];
#pragma unroll
            for (int j = 0; j < D; ++j) acc[D + i * D + j] += (double)v[i] * (double)v[j];
        }
    };
    // eight points' loads in flight per thread (one dependent round trip per point made this a 14 us kernel on 38 MB); the points
    // are added in the same order as before, so the sums are bit-identical
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; p + 7 * stride < P; p += 8 * stride) {
        float v[8][D];
#pragma unroll
        for (int u = 0; u < 8; ++u) load(p + u * stride, v[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) add(v[u]);
    }
    for (; p < P; p += stride) {
        float v[D];
        load(p, v);
        add(v);
    }
    const int wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const double s = wave_sum_f64(acc[i]);
        if (lane_id() == 0) part[(size_t)wave_g * V + i] = s;
    }
}

// ------------------------------------------------------------------------------------------
// fwd2: transposed orientation  D2^T[c2][p] = sum_k W2[c2][k] a1[p][k]  so that the result
// registers are (lane = position, register = channel) = the fragment layout.
template <int D>
__global__ __launch_bounds__(256, 2) void k_sa_fwd2(const float* __restrict__ x, int nunits,
                                                 const float* __restrict__ l1tab_g, const float* __restrict__ W2,
                                                 const float* __restrict__ b2, float* __restrict__ y2f,
                                                 double* __restrict__ part) {
    __shared__ float4 w2f[2 * 8 * 64];      // A fragments of W2: [rt][s4][lane] -> W2[32rt+q][32h+4s4 .. +3]
    __shared__ float4 l1tab[64 * 2];        // folded layer 1: [c][w0 w1 w2 w3 | b 0 0 0]
    __shared__ float4 b2s[16];
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int ln = i & 63, s4 = (i >> 6) & 7, rt = i >> 9;
        w2f[i] = *reinterpret_cast<const float4*>(W2 + (32 * rt + (ln & 31)) * 64 + 32 * (ln >> 5) + 4 * s4);
    }
    if (threadIdx.x < 128) l1tab[threadIdx.x] = reinterpret_cast<const float4*>(l1tab_g)[threadIdx.x];
    if (threadIdx.x < 16) b2s[threadIdx.x] = reinterpret_cast<const float4*>(b2)[threadIdx.x];
    __syncthreads();

    const int lane = lane_id(), h = lane >> 5, q = lane & 31;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;

    float ps[2][16], pq[2][16];            // per-lane partial sum / sumsq of y2, channel = 32rt+rowmap(r,h)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ps[rt][r] = 0.f; pq[rt][r] = 0.f; }

    // software prefetch: with one wave per SIMD the HBM latency of the next unit's input is otherwise exposed
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
    float xn[2][4];
    if (wave_g < nunits) load_x(wave_g, xn);
    for (int u = wave_g; u < nunits; u += nwaves) {
        asm volatile("" ::: "memory");       // LDS tables are re-read per unit instead of living in registers
        float xv[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[ct][i] = xn[ct][i];
        if (u + nwaves < nunits) load_x(u + nwaves, xn);
        // layer 1 on the VALU: this lane's two positions x the 32 channels of its half
        float a1[2][32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float4 w = l1tab[(32 * h + s) * 2];
            const float b = l1tab[(32 * h + s) * 2 + 1].x;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float v = fmaf(w.x, xv[ct][0], b);
                v = fmaf(w.y, xv[ct][1], v);
                v = fmaf(w.z, xv[ct][2], v);
                if (D == 4) v = fmaf(w.w, xv[ct][3], v);
                a1[ct][s] = fmaxf(v, 0.f);
            }
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 bb = b2s[8 * rt + 2 * r4 + h];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc[rt][ct][4 * r4 + 0] = bb.x; acc[rt][ct][4 * r4 + 1] = bb.y;
                    acc[rt][ct][4 * r4 + 2] = bb.z; acc[rt][ct][4 * r4 + 3] = bb.w;
                }
            }
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const float4 f0 = w2f[(0 * 8 + s4) * 64 + lane];
            const float4 f1 = w2f[(1 * 8 + s4) * 64 + lane];
            const float fa0[4] = {f0.x, f0.y, f0.z, f0.w};
            const float fa1[4] = {f1.x, f1.y, f1.z, f1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = MFMA32(fa0[e], a1[0][4 * s4 + e], acc[0][0]);
                acc[0][1] = MFMA32(fa0[e], a1[1][4 * s4 + e], acc[0][1]);
                acc[1][0] = MFMA32(fa1[e], a1[0][4 * s4 + e], acc[1][0]);
                acc[1][1] = MFMA32(fa1[e], a1[1][4 * s4 + e], acc[1][1]);
            }
        }
        float* tile = y2f + (size_t)u * FACL_UNIT_ELEMS;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 v = make_float4(acc[rt][ct][4 * r4], acc[rt][ct][4 * r4 + 1],
                                                 acc[rt][ct][4 * r4 + 2], acc[rt][ct][4 * r4 + 3]);
                    *reinterpret_cast<float4*>(tile + (((ct * 2 + rt) * 4 + r4) * 64 + lane) * 4) = v;
                }
        if (part) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v0 = acc[rt][0][r], v1 = acc[rt][1][r];
                    ps[rt][r] += v0 + v1;
                    pq[rt][r] = fmaf(v0, v0, fmaf(v1, v1, pq[rt][r]));
                }
        }
    }
    if (part) {
        // reduce over the 32 positions-lanes of each half in fp64; lane q == 0 of each half writes
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double s = ps[rt][r], sq = pq[rt][r];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); sq += __shfl_xor(sq, o, 64); }
                if (q == 0) {
                    const int c = 32 * rt + rowmap(r, h);
                    part[(size_t)wave_g * 128 + 2 * c] = s;
                    part[(size_t)wave_g * 128 + 2 * c + 1] = sq;
                }
            }
    }
}

// Split-bf16 (bf16x6, common.h) version of fwd2: same orientation, outputs and statistics; W2's A fragments are
// pre-split planes in LDS (24 KiB), the lane computes layer 1 for its two positions on exactly the 32 channels its
// k-slots need (block kk, half h, slot j <-> channel 16kk + 8h + j) and splits them in registers.  96 MFMAs per unit
// instead of 128 four-times-slower ones: the kernel becomes bound by its 16 KiB/unit store stream.
template <int D, bool H3>
__global__ __launch_bounds__(256, 2) void k_sa_fwd2_sb(const float* __restrict__ x, int nunits,
                                                       const float* __restrict__ l1tab_g, const float* __restrict__ W2,
                                                       const float* __restrict__ b2, float* __restrict__ y2f,
                                                       double* __restrict__ part, const unsigned* __restrict__ a1amax) {
    __shared__ uint4 w2p[2 * 4 * 3 * 64];   // A fragments of W2, [(rt*4 + kk)*3 + plane][lane]: W2[32rt+r][16kk+8h .. +7]
    __shared__ float4 l1tab[64 * 2];        // folded layer 1: [c][w0 w1 w2 w3 | b 0 0 0]
    __shared__ float4 b2s[16];
    __shared__ float red[16];
    // fp16x3 operand scales (common.h): W2 by the power of two of its own maximum (taken here), a1 by the one of its bound
    int seW = 127, seA = 127;
    if (H3) { seW = wg_h3_se(W2, 64 * 64, red); seA = h3_se_of(a1amax); }
    const float sW = pow2_biased(seW), sA = pow2_biased(seA), sAW = sA * sW, unsAW = h3_unscale(seA, seW);
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int ln = i & 63, kk = (i >> 6) & 3, rt = i >> 8;
        const float* wrow = W2 + (32 * rt + (ln & 31)) * 64 + 16 * kk + 8 * (ln >> 5);
        const float4 w0 = *reinterpret_cast<const float4*>(wrow), w1 = *reinterpret_cast<const float4*>(wrow + 4);
        unsigned hi[4], mi[4], lo[4];
        if (H3) {                                            // fp16x3 (common.h): two fp16 planes of w * 2^8 in slots 0 and 1
            split_pair_h(w0.x * sW, w0.y * sW, hi[0], mi[0]);
            split_pair_h(w0.z * sW, w0.w * sW, hi[1], mi[1]);
            split_pair_h(w1.x * sW, w1.y * sW, hi[2], mi[2]);
            split_pair_h(w1.z * sW, w1.w * sW, hi[3], mi[3]);
            lo[0] = lo[1] = lo[2] = lo[3] = 0u;
        } else {
            split_pair(w0.x, w0.y, hi[0], mi[0], lo[0]);
            split_pair(w0.z, w0.w, hi[1], mi[1], lo[1]);
            split_pair(w1.x, w1.y, hi[2], mi[2], lo[2]);
            split_pair(w1.z, w1.w, hi[3], mi[3], lo[3]);
        }
        uint4* d = w2p + ((rt * 4 + kk) * 3) * 64 + ln;
        d[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        d[64] = make_uint4(mi[0], mi[1], mi[2], mi[3]);
        d[128] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
    if (threadIdx.x < 128) {
        float4 t = reinterpret_cast<const float4*>(l1tab_g)[threadIdx.x];
        if (H3) { t.x *= sA; t.y *= sA; t.z *= sA; t.w *= sA; }   // relu(s w.x + s b) = s relu(w.x + b) for a power of two s: exact
        l1tab[threadIdx.x] = t;
    }
    if (threadIdx.x < 16) b2s[threadIdx.x] = reinterpret_cast<const float4*>(b2)[threadIdx.x];
    __syncthreads();

    const int lane = lane_id(), h = lane >> 5, q = lane & 31;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;

    float ps[2][16], pq[2][16];            // per-lane partial sum / sumsq of y2, channel = 32rt+rowmap(r,h)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ps[rt][r] = 0.f; pq[rt][r] = 0.f; }

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
    float xn[2][4];
    if (wave_g < nunits) load_x(wave_g, xn);
    for (int u = wave_g; u < nunits; u += nwaves) {
        asm volatile("" ::: "memory");       // LDS tables are re-read per unit instead of living in registers
        float xv[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[ct][i] = xn[ct][i];
        load_x(u + nwaves < nunits ? u + nwaves : u, xn);   // unconditional: a conditional prefetch is waited for on the spot (see k_sa_fwd3_sb)
        // layer 1 on the VALU for this lane's two positions, straight into the bf16 planes of the B operand
        bf16x8 ap[2][4][3];                  // [position tile][k16 block][plane]
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
                unsigned hi[4], mi[4], lo[4];
                if (H3) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) { split_pair_h(a1[ct][2 * t], a1[ct][2 * t + 1], hi[t], mi[t]); lo[t] = 0u; }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) split_pair(a1[ct][2 * t], a1[ct][2 * t + 1], hi[t], mi[t], lo[t]);
                }
                ap[ct][kk][0] = as_bf16x8(hi[0], hi[1], hi[2], hi[3]);
                ap[ct][kk][1] = as_bf16x8(mi[0], mi[1], mi[2], mi[3]);
                ap[ct][kk][2] = as_bf16x8(lo[0], lo[1], lo[2], lo[3]);
            }
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float4 bb = b2s[8 * rt + 2 * r4 + h];
                if (H3) { bb.x *= sAW; bb.y *= sAW; bb.z *= sAW; bb.w *= sAW; }   // the accumulator runs at (a sA)(w sW)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc[rt][ct][4 * r4 + 0] = bb.x; acc[rt][ct][4 * r4 + 1] = bb.y;
                    acc[rt][ct][4 * r4 + 2] = bb.z; acc[rt][ct][4 * r4 + 3] = bb.w;
                }
            }
        constexpr int PA[6] = FACL_SB_PA, PB[6] = FACL_SB_PB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 wf[2][3];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int pl = 0; pl < (H3 ? 2 : 3); ++pl) wf[rt][pl] = __builtin_bit_cast(bf16x8, w2p[((rt * 4 + kk) * 3 + pl) * 64 + lane]);
            if (H3) {
                constexpr int HA[3] = FACL_H3_PA, HB[3] = FACL_H3_PB;
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            acc[rt][ct] = MFMA_F16(__builtin_bit_cast(f16x8h, wf[rt][HA[t]]), __builtin_bit_cast(f16x8h, ap[ct][kk][HB[t]]), acc[rt][ct]);
            } else {
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = MFMA_BF16(wf[rt][PA[t]], ap[ct][kk][PB[t]], acc[rt][ct]);
            }
        }
        if (H3) {                                            // exact rescale (2^-12)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[rt][ct][r] *= unsAW;
        }
        float* tile = y2f + (size_t)u * FACL_UNIT_ELEMS;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 v = make_float4(acc[rt][ct][4 * r4], acc[rt][ct][4 * r4 + 1],
                                                 acc[rt][ct][4 * r4 + 2], acc[rt][ct][4 * r4 + 3]);
                    *reinterpret_cast<float4*>(tile + (((ct * 2 + rt) * 4 + r4) * 64 + lane) * 4) = v;
                }
        if (part) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v0 = acc[rt][0][r], v1 = acc[rt][1][r];
                    ps[rt][r] += v0 + v1;
                    pq[rt][r] = fmaf(v0, v0, fmaf(v1, v1, pq[rt][r]));
                }
        }
    }
    if (part) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double s = ps[rt][r], sq = pq[rt][r];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); sq += __shfl_xor(sq, o, 64); }
                if (q == 0) {
                    const int c = 32 * rt + rowmap(r, h);
                    part[(size_t)wave_g * 128 + 2 * c] = s;
                    part[(size_t)wave_g * 128 + 2 * c + 1] = sq;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// fwd3: normal orientation  D3[p][c3] = sum_k a2[p][k] W3'[c3][k]  (lane = channel, registers =
// positions), so BN statistics and the max over the group's positions are in-lane reductions.
__global__ __launch_bounds__(512) void k_sa_fwd3(const float* __restrict__ y2f, int nunits,
                                                 const float* __restrict__ sc2, const float* __restrict__ sh2,
                                                 const float* __restrict__ W3, const float* __restrict__ b3,
                                                 const float* __restrict__ sgn3, float* __restrict__ ymax,
                                                 unsigned char* __restrict__ arg, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    float4* w3f = lds4;                      // B fragments of sgn*W3: [ct3][rt][r4][lane] (64 KiB)
    float4* sc2s = lds4 + 4096;              // 16
    float4* sh2s = sc2s + 16;                // 16
    float* b3s = reinterpret_cast<float*>(sh2s + 16);   // 256
    // per-wave, per-lane fp64 (sum, sumsq) of y3 for each column tile: kept in LDS so that the
    // column-tile loop can stay a real loop (registers: a2 64 + acc 32 instead of 8 unrolled tiles)
    double2* stat = reinterpret_cast<double2*>(b3s + 256) + (threadIdx.x >> 6) * 512;   // [ct3][lane]
    for (int i = threadIdx.x; i < 4096; i += 512) {
        const int ln = i & 63, r4 = (i >> 6) & 3, rt = (i >> 8) & 1, ct3 = i >> 9;
        const int c3 = 32 * ct3 + (ln & 31);
        float4 w = *reinterpret_cast<const float4*>(W3 + c3 * 64 + 32 * rt + 8 * r4 + 4 * (ln >> 5));
        const float s = sgn_of(sgn3[c3]);
        w.x *= s; w.y *= s; w.z *= s; w.w *= s;
        w3f[i] = w;
    }
    if (threadIdx.x < 16) {
        sc2s[threadIdx.x] = reinterpret_cast<const float4*>(sc2)[threadIdx.x];
        sh2s[threadIdx.x] = reinterpret_cast<const float4*>(sh2)[threadIdx.x];
    }
    if (threadIdx.x < 256) b3s[threadIdx.x] = b3[threadIdx.x] * sgn_of(sgn3[threadIdx.x]);
    __syncthreads();

    const int lane = lane_id(), h = lane >> 5, q = lane & 31;
    const int wave_g = blockIdx.x * 8 + (threadIdx.x >> 6), nwaves = gridDim.x * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) stat[i * 64 + lane] = make_double2(0.0, 0.0);

    // register double buffer: the next unit's y2 tile is requested before this unit's 512 MFMAs
    float4 yn[16];
    auto issue_loads = [&](int u) {
        const float* tile = y2f + (size_t)u * FACL_UNIT_ELEMS;
#pragma unroll
        for (int i = 0; i < 16; ++i) yn[i] = *reinterpret_cast<const float4*>(tile + (i * 64 + lane) * 4);
    };
    if (wave_g < nunits) issue_loads(wave_g);
    for (int u = wave_g; u < nunits; u += nwaves) {
        float a2[2][2][16];                   // [pt][rt][r] = a2[p = 32pt+q][k = 32rt+rowmap(r,h)]
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 y = yn[(ct * 2 + rt) * 4 + r4];
                    const float4 sc = sc2s[8 * rt + 2 * r4 + h], sh = sh2s[8 * rt + 2 * r4 + h];
                    a2[ct][rt][4 * r4 + 0] = fmaxf(fmaf(sc.x, y.x, sh.x), 0.f);
                    a2[ct][rt][4 * r4 + 1] = fmaxf(fmaf(sc.y, y.y, sh.y), 0.f);
                    a2[ct][rt][4 * r4 + 2] = fmaxf(fmaf(sc.z, y.z, sh.z), 0.f);
                    a2[ct][rt][4 * r4 + 3] = fmaxf(fmaf(sc.w, y.w, sh.w), 0.f);
                }
        if (u + nwaves < nunits) issue_loads(u + nwaves);
#pragma unroll 1
        for (int ct3 = 0; ct3 < 8; ++ct3) {
            const float bias = b3s[32 * ct3 + q];
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = bias; acc1[r] = bias; }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 f = w3f[((ct3 * 2 + rt) * 4 + r4) * 64 + lane];
                    const float fb[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = MFMA32(a2[0][rt][4 * r4 + e], fb[e], acc0);
                        acc1 = MFMA32(a2[1][rt][4 * r4 + e], fb[e], acc1);
                    }
                }
            float s = 0.f, sq = 0.f, best = acc0[0];
            int bp = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc0[r];
                s += v; sq = fmaf(v, v, sq);
                if (v > best) { best = v; bp = rowmap(r, 0); }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc1[r];
                s += v; sq = fmaf(v, v, sq);
                if (v > best) { best = v; bp = 32 + rowmap(r, 0); }
            }
            bp += 4 * h;
            const float ob = __shfl_xor(best, 32, 64);
            const int op = __shfl_xor(bp, 32, 64);
            if (ob > best || (ob == best && op < bp)) { best = ob; bp = op; }   // first max wins (MaxPool2d)
            if (part) {
                double2 st = stat[ct3 * 64 + lane];
                st.x += (double)s; st.y += (double)sq;
                stat[ct3 * 64 + lane] = st;
            }
            if (h == 0) {
                ymax[(size_t)u * 256 + 32 * ct3 + q] = best;
                arg[(size_t)u * 256 + 32 * ct3 + q] = (unsigned char)bp;
            }
        }
    }
    if (part) {
#pragma unroll
        for (int ct3 = 0; ct3 < 8; ++ct3) {
            const double2 st = stat[ct3 * 64 + lane];
            const double s = st.x + __shfl_xor(st.x, 32, 64);
            const double sq = st.y + __shfl_xor(st.y, 32, 64);
            if (h == 0) {
                part[(size_t)wave_g * 512 + 2 * (32 * ct3 + q)] = s * (double)sgn_of(sgn3[32 * ct3 + q]);   // statistics of y3, not of sgn3*y3
                part[(size_t)wave_g * 512 + 2 * (32 * ct3 + q) + 1] = sq;
            }
        }
    }
}

// Split-bf16 (bf16x6, common.h) version of fwd3: same tiling, same epilogue, same outputs; the 64x64 a2 tile of a
// unit never leaves the wave's registers -- the fragment-layout float4s of y2 already hold 4 consecutive channels of
// one position, so two of them give the 8 k-slots of a 32x32x16 A operand (slot (h,j) of block kk <-> channel
// 16kk + 4h + j for j < 4, 16kk + 8 + 4h + (j-4) else; W3's B fragments use the same map).  Per unit and wave:
// 64 values/lane split into 3 bf16 planes (96 VGPRs), 96 ds_read_b128 of pre-split sgn*W3 fragments, 384 MFMAs.
// Roofline: MFMA bf16 (2.5 PFLOP/s dense); 6 * 2*64*64*256 executed FLOP per unit.
// NP = 3: exact 3-way bf16 split (6 products per multiply-add); NP = 2: "bf16x3" (two pieces, three products: opt-in
// precision "x3", common.h); NP = 1: fp16-input variant of the dense configuration --
// a2 and W3 rounded to fp16, ONE v_mfma_f32_32x32x16_f16 product, fp32 accumulation (64 MFMAs per unit: HBM-bound).
typedef _Float16 f16x2q __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8q __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pk_f16q(float x0, float x1) {
    const f32x2v v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2q));
}
// WAVES / PREF: 8 waves per workgroup (two per SIMD) with the next unit's y2 tile prefetched into a second register set (206
// registers), or -- the fp16x3 default since round 4 -- 12 waves (three per SIMD) without the prefetch (160 registers).
template <int NP, int WAVES = 8, bool PREF = true>
__global__ __launch_bounds__(64 * WAVES) void k_sa_fwd3_sb(const float* __restrict__ y2f, int nunits,
                                                    const float* __restrict__ sc2, const float* __restrict__ sh2,
                                                    const float* __restrict__ W3, const float* __restrict__ b3,
                                                    const float* __restrict__ sgn3, float* __restrict__ ymax,
                                                    unsigned char* __restrict__ arg, double* __restrict__ part,
                                                    const unsigned* __restrict__ a2amax) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    uint4* w3p = reinterpret_cast<uint4*>(lds4);         // [(ct3*4 + kk)*3 + plane][lane]: 96 KiB
    float4* sc2s = lds4 + 6144;              // 16
    float4* sh2s = sc2s + 16;                // 16
    float* b3s = reinterpret_cast<float*>(sh2s + 16);   // 256
    // per-wave fp64 (sum, sumsq) of y3 per channel: [ct3][q] (the two lane halves are merged before the update)
    double2* stat = reinterpret_cast<double2*>(b3s + 256) + (threadIdx.x >> 6) * 256;
    // fp16x3 operand scales (common.h): W3 by the power of two of its own maximum -- every thread first loads the 32 weights
    // it is going to split (one pass over W3, all loads in flight), the workgroup takes the maximum, then the fragments are
    // split from registers --, a2 by the one of its bound
    constexpr int NIT = (2048 + 64 * WAVES - 1) / (64 * WAVES);
    float4 wv[NIT][2];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + 64 * WAVES * it;
        const int ln = i & 63, kk = (i >> 6) & 3, ct3 = (i >> 8) & 7;
        const float* wrow = W3 + (32 * ct3 + (ln & 31)) * 64 + 16 * kk + 4 * (ln >> 5);
        wv[it][0] = *reinterpret_cast<const float4*>(wrow);
        wv[it][1] = *reinterpret_cast<const float4*>(wrow + 8);
    }
    int seW = 127, seA = 127;
    if (NP == 4) {
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                m = fmaxf(fmaxf(m, fmaxf(fabsf(wv[it][e].x), fabsf(wv[it][e].y))), fmaxf(fabsf(wv[it][e].z), fabsf(wv[it][e].w)));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) b3s[threadIdx.x >> 6] = m;          // LDS scratch: b3s is filled further down
        __syncthreads();
        m = b3s[0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) m = fmaxf(m, b3s[w]);
        __syncthreads();
        seW = __builtin_amdgcn_readfirstlane(h3_se(__float_as_uint(m)));
        seA = h3_se_of(a2amax);
    }
    const float sW3 = pow2_biased(seW), sA2 = pow2_biased(seA), UNS = h3_unscale(seA, seW);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + 64 * WAVES * it;
        if (i >= 2048) break;
        const int ln = i & 63, kk = (i >> 6) & 3, ct3 = i >> 8;
        const int c3 = 32 * ct3 + (ln & 31);
        float4 w0 = wv[it][0], w1 = wv[it][1];
        const float s = sgn_of(sgn3[c3]);
        unsigned hi[4], mi[4], lo[4];
        if (NP == 4) {                                      // fp16x3: two fp16 planes of w * 2^8 (plane slots 0 and 1)
            const float sw = s * sW3;
            split_pair_h(w0.x * sw, w0.y * sw, hi[0], mi[0]);
            split_pair_h(w0.z * sw, w0.w * sw, hi[1], mi[1]);
            split_pair_h(w1.x * sw, w1.y * sw, hi[2], mi[2]);
            split_pair_h(w1.z * sw, w1.w * sw, hi[3], mi[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) lo[j] = 0u;
        } else if (NP >= 2) {
            split_pair(w0.x * s, w0.y * s, hi[0], mi[0], lo[0]);
            split_pair(w0.z * s, w0.w * s, hi[1], mi[1], lo[1]);
            split_pair(w1.x * s, w1.y * s, hi[2], mi[2], lo[2]);
            split_pair(w1.z * s, w1.w * s, hi[3], mi[3], lo[3]);
        } else {
            hi[0] = pk_f16q(w0.x * s, w0.y * s); hi[1] = pk_f16q(w0.z * s, w0.w * s);
            hi[2] = pk_f16q(w1.x * s, w1.y * s); hi[3] = pk_f16q(w1.z * s, w1.w * s);
#pragma unroll
            for (int j = 0; j < 4; ++j) mi[j] = lo[j] = 0u;
        }
        uint4* d = w3p + ((ct3 * 4 + kk) * 3) * 64 + ln;
        d[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        d[64] = make_uint4(mi[0], mi[1], mi[2], mi[3]);
        d[128] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
    if (threadIdx.x < 16) {
        float4 a = reinterpret_cast<const float4*>(sc2)[threadIdx.x], b = reinterpret_cast<const float4*>(sh2)[threadIdx.x];
        if (NP == 4) {                                       // relu(16 s y + 16 t) = 16 relu(s y + t) exactly: the activation scale is free
            a.x *= sA2; a.y *= sA2; a.z *= sA2; a.w *= sA2;
            b.x *= sA2; b.y *= sA2; b.z *= sA2; b.w *= sA2;
        }
        sc2s[threadIdx.x] = a;
        sh2s[threadIdx.x] = b;
    }
    // fp16x3: the accumulators start at 0 (an inline constant: no 32 register moves per tile) and hold u = y - bias scaled by
    // 2^12; the bias joins the maximum in the epilogue and the statistics once, in fp64, when the partial row is written
    if (threadIdx.x < 256) b3s[threadIdx.x] = b3[threadIdx.x] * sgn_of(sgn3[threadIdx.x]);
    __syncthreads();

    const int lane = lane_id(), h = lane >> 5, q = lane & 31;
    const int wave_g = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6)), nwaves = gridDim.x * WAVES;   // uniform: scalar addresses
#pragma unroll
    for (int i = 0; i < 4; ++i) stat[i * 64 + lane] = make_double2(0.0, 0.0);

    float4 yn[16];
    auto issue_loads = [&](int u) {
        const float* tile = y2f + (size_t)u * FACL_UNIT_ELEMS;
#pragma unroll
        for (int i = 0; i < 16; ++i) yn[i] = *reinterpret_cast<const float4*>(tile + (i * 64 + lane) * 4);
    };
    if (PREF && wave_g < nunits) issue_loads(wave_g);
    int nun = 0;
    for (int u = wave_g; u < nunits; u += nwaves, ++nun) {
        if (!PREF) issue_loads(u);           // three waves per SIMD cover the latency instead of a second register set
        bf16x8 ap[2][4][3];                  // [position tile][k16 block][plane]
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int rt = kk >> 1, m = kk & 1;
                unsigned hi[4], mi[4], lo[4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 y = yn[(ct * 2 + rt) * 4 + 2 * m + t];
                    const float4 sc = sc2s[8 * rt + 2 * (2 * m + t) + h], sh = sh2s[8 * rt + 2 * (2 * m + t) + h];
                    if (NP == 4) {
                        // (fmaxf drops a NaN here; in training the layer's statistics -- sums of y2 taken by facl_sa_fwd2 -- carry it)
                        split_pair_h(fmaxf(fmaf(sc.x, y.x, sh.x), 0.f), fmaxf(fmaf(sc.y, y.y, sh.y), 0.f), hi[2 * t], mi[2 * t]);
                        split_pair_h(fmaxf(fmaf(sc.z, y.z, sh.z), 0.f), fmaxf(fmaf(sc.w, y.w, sh.w), 0.f), hi[2 * t + 1], mi[2 * t + 1]);
                        lo[2 * t] = lo[2 * t + 1] = 0u;
                    } else if (NP >= 2) {
                        split_pair(fmaxf(fmaf(sc.x, y.x, sh.x), 0.f), fmaxf(fmaf(sc.y, y.y, sh.y), 0.f), hi[2 * t], mi[2 * t], lo[2 * t]);
                        split_pair(fmaxf(fmaf(sc.z, y.z, sh.z), 0.f), fmaxf(fmaf(sc.w, y.w, sh.w), 0.f), hi[2 * t + 1], mi[2 * t + 1], lo[2 * t + 1]);
                    } else {
                        hi[2 * t] = pk_f16q(fmaxf(fmaf(sc.x, y.x, sh.x), 0.f), fmaxf(fmaf(sc.y, y.y, sh.y), 0.f));
                        hi[2 * t + 1] = pk_f16q(fmaxf(fmaf(sc.z, y.z, sh.z), 0.f), fmaxf(fmaf(sc.w, y.w, sh.w), 0.f));
                        mi[2 * t] = mi[2 * t + 1] = lo[2 * t] = lo[2 * t + 1] = 0u;
                    }
                }
                ap[ct][kk][0] = as_bf16x8(hi[0], hi[1], hi[2], hi[3]);
                ap[ct][kk][1] = as_bf16x8(mi[0], mi[1], mi[2], mi[3]);
                ap[ct][kk][2] = as_bf16x8(lo[0], lo[1], lo[2], lo[3]);
            }
        // unconditional (the last round re-reads its own unit, an L2 hit): under `if (u + nwaves < nunits)` the register
        // allocator merged the loaded / not-loaded values with copies OUT of the freshly loaded registers, behind an
        // s_waitcnt vmcnt(8) -- every wave stalled for a full memory round trip per unit (~100 us of the kernel's 360)
        if (PREF) issue_loads(u + nwaves < nunits ? u + nwaves : u);
        // One column tile (32 channels) = 8 / 12 / 24 MFMAs into two accumulators (the unit's two position tiles), then an
        // epilogue of ~130 VALU instructions on them.
        auto mma_tile = [&](int ct3, f32x16& acc0, f32x16& acc1) {
            const float bias = NP == 4 ? 0.f : b3s[32 * ct3 + q];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = bias; acc1[r] = bias; }
            constexpr int PA[6] = FACL_SB_PA, PB[6] = FACL_SB_PB;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 bfr[3];
#pragma unroll
                for (int p = 0; p < (NP == 4 ? 2 : 3); ++p) bfr[p] = __builtin_bit_cast(bf16x8, w3p[((ct3 * 4 + kk) * 3 + p) * 64 + lane]);
                if (NP == 4) {                                              // fp16x3: (lo,hi) (hi,lo) (hi,hi)
                    constexpr int HA[3] = FACL_H3_PA, HB[3] = FACL_H3_PB;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        acc0 = MFMA_F16(__builtin_bit_cast(f16x8h, ap[0][kk][HA[t]]), __builtin_bit_cast(f16x8h, bfr[HB[t]]), acc0);
                        acc1 = MFMA_F16(__builtin_bit_cast(f16x8h, ap[1][kk][HA[t]]), __builtin_bit_cast(f16x8h, bfr[HB[t]]), acc1);
                    }
                } else if (NP == 3) {
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        acc0 = MFMA_BF16(ap[0][kk][PA[t]], bfr[PB[t]], acc0);
                        acc1 = MFMA_BF16(ap[1][kk][PA[t]], bfr[PB[t]], acc1);
                    }
                } else if (NP == 2) {                                       // bf16x3 (opt-in): (hi,mid) (mid,hi) (hi,hi)
                    constexpr int PA3[3] = FACL_SB3_PA, PB3[3] = FACL_SB3_PB;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        acc0 = MFMA_BF16(ap[0][kk][PA3[t]], bfr[PB3[t]], acc0);
                        acc1 = MFMA_BF16(ap[1][kk][PA3[t]], bfr[PB3[t]], acc1);
                    }
                } else {
                    const f16x8q wb = __builtin_bit_cast(f16x8q, bfr[0]);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8q, ap[0][kk][0]), wb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8q, ap[1][kk][0]), wb, acc1, 0, 0, 0);
                }
            }
        };
        struct Epi { float s, sq, best; int bp; };
        auto epi_compute = [&](int ct3, const f32x16& acc0, const f32x16& acc1) -> Epi {
            // two interleaved sum chains; the maximum with v_max3_f32, then its FIRST position without
            // compares (a v_cmp -> v_cndmask chain costs a hazard nop per value): d = v - best is <= 0 and exactly +0 where v is
            // the maximum, so bits(d) | position is the position there and >= 2^31 elsewhere; the unsigned minimum is the answer
            float2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
            float best = acc0[0];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {                   // scalar on purpose: packed f32 VALU is slow beside MFMAs (MI355X guide)
                s2.x += acc0[r]; s2.y += acc0[r + 1]; q2.x = fmaf(acc0[r], acc0[r], q2.x); q2.y = fmaf(acc0[r + 1], acc0[r + 1], q2.y);
                s2.x += acc1[r]; s2.y += acc1[r + 1]; q2.x = fmaf(acc1[r], acc1[r], q2.x); q2.y = fmaf(acc1[r + 1], acc1[r + 1], q2.y);
                best = fmaxf(fmaxf(best, acc0[r]), acc0[r + 1]);
                best = fmaxf(fmaxf(best, acc1[r]), acc1[r + 1]);
            }
            unsigned key = 0xffffffffu;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const unsigned k0 = __float_as_uint(acc0[r] - best) | (unsigned)rowmap(r, 0), k1 = __float_as_uint(acc0[r + 1] - best) | (unsigned)rowmap(r + 1, 0);
                const unsigned k2 = __float_as_uint(acc1[r] - best) | (unsigned)(32 + rowmap(r, 0)), k3 = __float_as_uint(acc1[r + 1] - best) | (unsigned)(32 + rowmap(r + 1, 0));
                key = min(min(key, k0), k1);
                key = min(min(key, k2), k3);
            }
            int bp = (int)(key & 63u);
            float s = s2.x + s2.y, sq = q2.x + q2.y;
            if (NP == 4) {                                      // exact rescale (powers of two); statistics stay those of u = y - bias
                s *= UNS; sq = (sq * UNS) * UNS;
                best = fmaf(best, UNS, b3s[32 * ct3 + q]);
            }
            bp += 4 * h;
            return Epi{s, sq, best, bp};
        };
        // The two lane halves hold the two position halves of a channel: lanes 0..31 fetch their partner's (lane + 32) results
        // with v_permlane32_swap (a VALU instruction: ds_bpermute would park the wave on an LDS round trip per tile) and update
        // the wave's fp64 statistics with no-return LDS atomics (ds_add_f64: no read to wait for either).
        auto upper = [&](unsigned x) { return __builtin_amdgcn_permlane32_swap(x, x, false, false)[1]; };
        auto epi_store = [&](int ct3, const Epi& e) {
            float best = e.best;
            int bp = e.bp;
            const float ob = __uint_as_float(upper(__float_as_uint(best)));
            const int op = (int)upper((unsigned)bp);
            if (ob > best || (ob == best && op < bp)) { best = ob; bp = op; }   // first max wins (MaxPool2d)
            if (part) {
                const float st = e.s + __uint_as_float(upper(__float_as_uint(e.s))), sqt = e.sq + __uint_as_float(upper(__float_as_uint(e.sq)));
                if (h == 0) {
                    double* d = reinterpret_cast<double*>(stat + ct3 * 32 + q);
                    __hip_atomic_fetch_add(d, (double)st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(d + 1, (double)sqt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            if (h == 0) {
                ymax[(size_t)u * 256 + 32 * ct3 + q] = best;
                arg[(size_t)u * 256 + 32 * ct3 + q] = (unsigned char)bp;
            }
        };
        // (tried on this loop, measured equal within 1-2 %: the next tile's MFMAs interleaved by hand with this tile's epilogue
        // in the wave; waves 4..7 running one MFMA phase ahead of their SIMD partners; B fragments requested a k-block ahead)
        f32x16 a0, a1;
#pragma unroll 1
        for (int ct3 = 0; ct3 < 8; ++ct3) {
            mma_tile(ct3, a0, a1);
            epi_store(ct3, epi_compute(ct3, a0, a1));
        }
    }
    if (part && h == 0) {
#pragma unroll
        for (int ct3 = 0; ct3 < 8; ++ct3) {
            double2 st = stat[ct3 * 32 + q];
            if (NP == 4) {                       // sums of u = y - bias over n positions -> sums of y
                const double b = (double)b3s[32 * ct3 + q], n = 64.0 * (double)nun;
                st.y += 2.0 * b * st.x + n * b * b;
                st.x += n * b;
            }
            part[(size_t)wave_g * 512 + 2 * (32 * ct3 + q)] = st.x * (double)sgn_of(sgn3[32 * ct3 + q]);   // statistics of y3, not of sgn3*y3
            part[(size_t)wave_g * 512 + 2 * (32 * ct3 + q) + 1] = st.y;
        }
    }
}

__global__ void k_sa_pool(const float* __restrict__ ymax, long long n4, int C4, const float* __restrict__ scale,
                          const float* __restrict__ shift, float* __restrict__ pooled, unsigned* __restrict__ amax) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    float mx = 0.f;
    // the channel quad of element i advances by stride % C4 per trip: two 32-bit remainders per thread instead of a 64-bit one
    // (~150 instructions) per element -- the kernel was VALU-bound on `i % C4`, not on its 100 MB
    const unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x;              // < 2^31: the grid is capped well below
    int c4 = (int)(i0 % (unsigned)C4);
    const int dc = (int)((unsigned)stride % (unsigned)C4);
    for (long long i = i0; i < n4; i += stride, c4 = c4 + dc >= C4 ? c4 + dc - C4 : c4 + dc) {
        const float4 y = reinterpret_cast<const float4*>(ymax)[i];
        const float4 sc = reinterpret_cast<const float4*>(scale)[c4];
        const float4 sh = reinterpret_cast<const float4*>(shift)[c4];
        float4 o;
        o.x = relu_nan(fmaf(fabsf(sc.x), y.x, sh.x));
        o.y = relu_nan(fmaf(fabsf(sc.y), y.y, sh.y));
        o.z = relu_nan(fmaf(fabsf(sc.z), y.z, sh.z));
        o.w = relu_nan(fmaf(fabsf(sc.w), y.w, sh.w));
        reinterpret_cast<float4*>(pooled)[i] = o;
        mx = fmaxf(fmaxf(mx, fmaxf(o.x, o.y)), fmaxf(o.z, o.w));
    }
    // the exact maximum of the pooled features (>= 0) for the fp16x3 scale of the GEMM that consumes them: one atomic per
    // wave into one of the FACL_AMAX_SLOTS hashed slots (non-negative floats order like unsigned integers)
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0) {
            const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
            atomicMax(amax + (wave & (FACL_AMAX_SLOTS - 1)) * FACL_AMAX_STRIDE, __float_as_uint(mx));
        }
    }
}

}  // namespace

extern "C" int facl_sa_x_moments(const float* x, int64_t P, int D, double* mom, void* ws, void* stream) {
    if (!x || !mom || !ws) return FACL_E_NULL;
    if ((D != 3 && D != 4) || P < 1) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int grid = 512, V = D + D * D;       // 512 blocks x 4 waves = 2048 partial rows
    if (D == 4) hipLaunchKernelGGL((k_x_moments<4>), dim3(grid), dim3(256), 0, st, x, (long long)P, (double*)ws);
    else hipLaunchKernelGGL((k_x_moments<3>), dim3(grid), dim3(256), 0, st, x, (long long)P, (double*)ws);
    int rc = facl_launch_status();
    if (rc) return rc;
    return facl_reduce_rows((const double*)ws, grid * 4, V, mom, st);
}

// a1amax: FACL_AMAX_WORDS uint32 holding the bits of a bound of max|a1| (facl_sa_l1tab writes it): the fp16x3 scale of a1
extern "C" int facl_sa_fwd2(const float* x, int64_t nunits, int D, const float* l1tab, const float* W2,
                            const float* b2, float* y2f, double* sums2, void* ws, const uint32_t* a1amax, void* stream) {
    if (!x || !l1tab || !W2 || !b2 || !y2f || (sums2 && !ws) || !a1amax) return FACL_E_NULL;
    if ((D != 3 && D != 4) || nunits < 1 || nunits > 0x7fffffff) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)(nunits < 2 * SA_GRID * 4 ? (nunits + 3) / 4 : 2 * SA_GRID);   // 2 workgroups per CU
    double* part = sums2 ? (double*)ws : nullptr;
    static const int use_f32 = getenv("FACL_SA_F32") ? atoi(getenv("FACL_SA_F32")) : 0;     // exact-fp32 MFMA kernel instead
    if (use_f32) {
        if (D == 4) hipLaunchKernelGGL((k_sa_fwd2<4>), dim3(grid), dim3(256), 0, st, x, (int)nunits, l1tab, W2, b2, y2f, part);
        else hipLaunchKernelGGL((k_sa_fwd2<3>), dim3(grid), dim3(256), 0, st, x, (int)nunits, l1tab, W2, b2, y2f, part);
    } else {
        // fp16x3 (two fp16 planes, three products: csrc/common.h) unless FACL_FWD_H3=0 selects bf16x6 (A/B)
        static const int h3 = getenv("FACL_FWD_H3") ? atoi(getenv("FACL_FWD_H3")) : 1;
        if (h3) {
            if (D == 4) hipLaunchKernelGGL((k_sa_fwd2_sb<4, true>), dim3(grid), dim3(256), 0, st, x, (int)nunits, l1tab, W2, b2, y2f, part, a1amax);
            else hipLaunchKernelGGL((k_sa_fwd2_sb<3, true>), dim3(grid), dim3(256), 0, st, x, (int)nunits, l1tab, W2, b2, y2f, part, a1amax);
        } else {
            if (D == 4) hipLaunchKernelGGL((k_sa_fwd2_sb<4, false>), dim3(grid), dim3(256), 0, st, x, (int)nunits, l1tab, W2, b2, y2f, part, a1amax);
            else hipLaunchKernelGGL((k_sa_fwd2_sb<3, false>), dim3(grid), dim3(256), 0, st, x, (int)nunits, l1tab, W2, b2, y2f, part, a1amax);
        }
    }
    int rc = facl_launch_status();
    if (rc || !sums2) return rc;
    // waves beyond nunits never enter the loop and still write their (zero) rows
    return facl_reduce_rows(part, grid * 4, 128, sums2, st);
}

static int sa_fwd3_p(const float* y2f, int64_t nunits, const float* scale2, const float* shift2,
                     const float* W3, const float* b3, const float* sgn3, float* ymax, uint8_t* arg,
                     double* sums3, void* ws, void* stream, int prec, const uint32_t* a2amax = nullptr) {
    if (!y2f || !scale2 || !shift2 || !W3 || !b3 || !sgn3 || !ymax || !arg || (sums3 && !ws) || (prec == 3 && !a2amax)) return FACL_E_NULL;
    if (nunits < 1 || nunits > 0x7fffffff) return FACL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    // 12 waves per workgroup (three per SIMD, 160 registers, no register prefetch of the next y2 tile) since round 4: 0.350 vs
    // 0.359 ms at the headline shape, three alternating same-box runs (gpurun_out: FACL_FWD3_W12=0 restores 8 waves + prefetch)
    static const int w12 = getenv("FACL_FWD3_W12") ? atoi(getenv("FACL_FWD3_W12")) : 1;
    const int WV = (w12 && prec == 3) ? 12 : 8;
    const int grid = (int)(nunits < SA_GRID * WV ? (nunits + WV - 1) / WV : SA_GRID);
    // FACL_SA_F32=1 selects the exact-fp32 MFMA kernel (v_mfma_f32_32x32x2_f32) instead of the split-bf16 one
    static const int env_f32 = getenv("FACL_SA_F32") ? atoi(getenv("FACL_SA_F32")) : 0;
    const int use_f32 = env_f32 && prec == 0;
    const size_t lds = use_f32 ? (4096 + 32) * sizeof(float4) + 256 * sizeof(float) + 8 * 512 * sizeof(double2)
                               : (6144 + 32) * sizeof(float4) + 256 * sizeof(float) + WV * 256 * sizeof(double2);
    const void* fn = use_f32 ? (const void*)k_sa_fwd3 : prec == 1 ? (const void*)k_sa_fwd3_sb<1>
                   : prec == 2 ? (const void*)k_sa_fwd3_sb<2> : prec == 3 ? (const void*)k_sa_fwd3_sb<4> : (const void*)k_sa_fwd3_sb<3>;
    if (prec < 0 || prec > 3) return FACL_E_CONFIG;
    static bool attr_done[64][5] = {};                 // per device ordinal (the attribute is per device) and kernel variant
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    const int slot = WV == 12 ? 4 : prec;
    if (!attr_done[dev][slot]) {
        hipError_t e = hipFuncSetAttribute(WV == 12 ? (const void*)k_sa_fwd3_sb<4, 12, false> : fn,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done[dev][slot] = true;
    }
    double* part = sums3 ? (double*)ws : nullptr;
    if (use_f32)
        hipLaunchKernelGGL(k_sa_fwd3, dim3(grid), dim3(512), lds, st, y2f, (int)nunits, scale2, shift2, W3, b3, sgn3, ymax,
                           arg, part);
    else if (prec == 1)
        hipLaunchKernelGGL((k_sa_fwd3_sb<1>), dim3(grid), dim3(512), lds, st, y2f, (int)nunits, scale2, shift2, W3, b3, sgn3,
                           ymax, arg, part, a2amax);
    else if (prec == 2)
        hipLaunchKernelGGL((k_sa_fwd3_sb<2>), dim3(grid), dim3(512), lds, st, y2f, (int)nunits, scale2, shift2, W3, b3, sgn3,
                           ymax, arg, part, a2amax);
    else if (prec == 3 && WV == 12)
        hipLaunchKernelGGL((k_sa_fwd3_sb<4, 12, false>), dim3(grid), dim3(768), lds, st, y2f, (int)nunits, scale2, shift2, W3, b3, sgn3,
                           ymax, arg, part, a2amax);
    else if (prec == 3)
        hipLaunchKernelGGL((k_sa_fwd3_sb<4>), dim3(grid), dim3(512), lds, st, y2f, (int)nunits, scale2, shift2, W3, b3, sgn3,
                           ymax, arg, part, a2amax);
    else
        hipLaunchKernelGGL((k_sa_fwd3_sb<3>), dim3(grid), dim3(512), lds, st, y2f, (int)nunits, scale2, shift2, W3, b3, sgn3,
                           ymax, arg, part, a2amax);
    int rc = facl_launch_status();
    if (rc || !sums3) return rc;
    return facl_reduce_rows(part, grid * WV, 512, sums3, st);
}

extern "C" int facl_sa_fwd3(const float* y2f, int64_t nunits, const float* scale2, const float* shift2,
                            const float* W3, const float* b3, const float* sgn3, float* ymax, uint8_t* arg,
                            double* sums3, void* ws, void* stream) {
    return sa_fwd3_p(y2f, nunits, scale2, shift2, W3, b3, sgn3, ymax, arg, sums3, ws, stream, 0);
}
// fp16-input twin (dense configuration): a2 and W3 rounded to fp16, one MFMA product per multiply-add, fp32 accumulation
extern "C" int facl_sa_fwd3_f16(const float* y2f, int64_t nunits, const float* scale2, const float* shift2,
                                const float* W3, const float* b3, const float* sgn3, float* ymax, uint8_t* arg,
                                double* sums3, void* ws, void* stream) {
    return sa_fwd3_p(y2f, nunits, scale2, shift2, W3, b3, sgn3, ymax, arg, sums3, ws, stream, 1);
}

// fp16x3 twin (csrc/common.h): a2 * 2^4 and W3' * 2^8 as two fp16 planes each, three products per multiply-add, fp32
// accumulation -- fp32-GEMM accuracy at half the MFMA work of facl_sa_fwd3; the default forward arithmetic of the model
extern "C" int facl_sa_fwd3_h3(const float* y2f, int64_t nunits, const float* scale2, const float* shift2, const float* W3,
                               const float* b3, const float* sgn3, float* ymax, uint8_t* arg, double* sums3, void* ws,
                               const uint32_t* a2amax, void* stream) {
    return sa_fwd3_p(y2f, nunits, scale2, shift2, W3, b3, sgn3, ymax, arg, sums3, ws, stream, 3, a2amax);
}

// "bf16x3" twin (opt-in precision "x3")
extern "C" int facl_sa_fwd3_x3(const float* y2f, int64_t nunits, const float* scale2, const float* shift2,
                               const float* W3, const float* b3, const float* sgn3, float* ymax, uint8_t* arg,
                               double* sums3, void* ws, void* stream) {
    return sa_fwd3_p(y2f, nunits, scale2, shift2, W3, b3, sgn3, ymax, arg, sums3, ws, stream, 2);
}

// amax (or null): FACL_AMAX_WORDS uint32, zeroed or holding earlier maxima; raised to the bits of max(pooled)
extern "C" int facl_sa_pool(const float* ymax, int64_t rows, int C, const float* scale, const float* shift,
                            float* pooled, uint32_t* amax, void* stream) {
    if (!ymax || !scale || !shift || !pooled) return FACL_E_NULL;
    if (rows < 1 || C < 4 || (C & 3)) return FACL_E_SHAPE;
    const long long n4 = rows * (long long)(C / 4);
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_sa_pool, dim3(grid), dim3(256), 0, (hipStream_t)stream, ymax, n4, C / 4, scale, shift, pooled, amax);
    return facl_launch_status();
}
