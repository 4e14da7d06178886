// kNN-then-radius grouping, fused with the gather + centre-subtract.
// Replaces the torch op chain of utils_my.py:265-284 (expand/sub/mul/sum -> topk -> S masked
// assigns -> gather -> subtract -> transpose views).  The (M,S,N) distance matrix is never
// materialised: one wave owns one centroid row, keeps its N distances in registers and finds
// the K-th smallest by an MSB-first radix select on the float bit patterns.
//
// Roofline: HBM.  Algorithmic bytes per cloud = N*D*4 (read) + S*K*(4 + 4*D) + S*12 (write).
#include "common.h"

namespace {

constexpr int GROUP_THREADS = 256;           // 4 waves
constexpr int CENTROIDS_PER_WG = 16;         // 4 per wave

// dist^2 exactly as the reference's fp32 chain: (dx*dx + dy*dy) + dz*dz, no FMA contraction.
__device__ __forceinline__ float dist2_exact(float px, float py, float pz, float cx, float cy, float cz) {
    const float dx = __fsub_rn(px, cx), dy = __fsub_rn(py, cy), dz = __fsub_rn(pz, cz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

template <int D, int NPL>
__global__ __launch_bounds__(GROUP_THREADS) void k_group(const float* __restrict__ points, int N, int S,
                                                         int K, float r2, int32_t* __restrict__ idx_out,
                                                         float* __restrict__ xt_out, float* __restrict__ yt_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* ys = xs + N;
    float* zs = ys + N;
    float* cs = zs + N;                       // only when D == 4
    const int m = blockIdx.y;
    const float* cloud = points + (size_t)m * N * D;

    // stage the cloud as SoA (conflict-free ds_read_b32 with consecutive lanes on consecutive points)
    for (int i = threadIdx.x; i < N; i += GROUP_THREADS) {
        if (D == 4) {
            const float4 p = *reinterpret_cast<const float4*>(cloud + (size_t)i * 4);
            xs[i] = p.x; ys[i] = p.y; zs[i] = p.z; cs[i] = p.w;
        } else {
            xs[i] = cloud[(size_t)i * 3 + 0];
            ys[i] = cloud[(size_t)i * 3 + 1];
            zs[i] = cloud[(size_t)i * 3 + 2];
        }
    }
    __syncthreads();

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const unsigned long long lt = lanemask_lt();

    for (int ci = wave; ci < CENTROIDS_PER_WG; ci += GROUP_THREADS / 64) {
        const int c = blockIdx.x * CENTROIDS_PER_WG + ci;       // wave-uniform
        if (c >= S) break;
        const float cx = xs[c], cy = ys[c], cz = zs[c];

        uint32_t key[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int i = j * 64 + lane;
            key[j] = (i < N) ? __float_as_uint(dist2_exact(xs[i], ys[i], zs[i], cx, cy, cz)) : 0x7F800000u;
        }

        // radix select on the float bit patterns (non-negative floats order like their bits), MSB first.
        // Invariant: `cand` keys share the resolved prefix, `remaining` of them belong to the K smallest.
        // Early exit as soon as cand == remaining (all candidates are kept): on continuous data the candidate
        // set shrinks to `remaining` after ~log2(N) mantissa bits, i.e. roughly half of the 31 rounds.
        uint32_t prefix = 0, hi = 0;
        int remaining = K, cand = NPL * 64;                     // padding keys (+inf) are ordinary candidates
        for (int bit = 30; bit >= 0 && cand != remaining; --bit) {
            hi = ~((2u << bit) - 1u);                           // bits above `bit`
            const uint32_t sel = hi | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) cnt += ((key[j] & sel) == prefix) ? 1 : 0;
            const int total = wave_sum_i32(cnt);                // candidates whose `bit` is 0
            if (total < remaining) { prefix |= (1u << bit); remaining -= total; cand -= total; }
            else cand = total;
            hi = sel;                                           // `bit` is resolved now
        }
        // keys with (key & hi) < prefix are kept; of those equal to prefix under `hi`, the first `remaining`
        // in index order (all of them when the loop exited early; exact-tie rule otherwise)

        const size_t grp = (size_t)m * S + c;
        int base = 0, eq_taken = 0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int i = j * 64 + lane;
            const uint32_t kh = key[j] & hi;
            const bool is_eq = kh == prefix;
            const unsigned long long eqm = __ballot(is_eq);
            const int eq_rank = eq_taken + __popcll(eqm & lt);
            eq_taken += __popcll(eqm);
            const bool take = (kh < prefix) || (is_eq && eq_rank < remaining);
            const unsigned long long tm = __ballot(take);
            if (take && N < 0) {
                const int pos = base + __popcll(tm & lt);
                const int id = (__uint_as_float(key[j]) > r2) ? c : i;      // strict >, utils_my.py:272
                const size_t o = grp * K + pos;
                if (idx_out) idx_out[o] = id;
                if (xt_out) {
                    const float gx = __fsub_rn(xs[id], cx), gy = __fsub_rn(ys[id], cy), gz = __fsub_rn(zs[id], cz);
                    if (D == 4) {
                        *reinterpret_cast<float4*>(xt_out + o * 4) = make_float4(gx, gy, gz, cs[id]);
                    } else {
                        xt_out[o * 3 + 0] = gx; xt_out[o * 3 + 1] = gy; xt_out[o * 3 + 2] = gz;
                    }
                }
            }
            base += __popcll(tm);
        }
        if (yt_out && lane < 3) yt_out[grp * 3 + lane] = (lane == 0) ? cx : (lane == 1) ? cy : cz;
    }
}

template <int D, int NPL>
int launch_group(const float* points, int M, int N, int S, int K, float r2, int32_t* idx, float* xt, float* yt,
                 hipStream_t st) {
    dim3 grid((S + CENTROIDS_PER_WG - 1) / CENTROIDS_PER_WG, M);
    const size_t lds = (size_t)N * 4 * sizeof(float);
    hipLaunchKernelGGL((k_group<D, NPL>), grid, dim3(GROUP_THREADS), lds, st, points, N, S, K, r2, idx, xt, yt);
    return facl_launch_status();
}

template <int D>
int dispatch_group(const float* points, int M, int N, int S, int K, float r2, int32_t* idx, float* xt, float* yt,
                   hipStream_t st) {
    if (N <= 512) return launch_group<D, 8>(points, M, N, S, K, r2, idx, xt, yt, st);
    if (N <= 1024) return launch_group<D, 16>(points, M, N, S, K, r2, idx, xt, yt, st);
    if (N <= 2048) return launch_group<D, 32>(points, M, N, S, K, r2, idx, xt, yt, st);
    return launch_group<D, 64>(points, M, N, S, K, r2, idx, xt, yt, st);
}

}  // namespace

extern "C" int facl_group(const float* points, int M, int N, int D, int S, int K, float r2, int32_t* idx,
                          float* xt, float* yt, void* stream) {
    if (!points) return FACL_E_NULL;
    if (M < 0 || N < 1 || N > 4096 || S < 1 || S > N || K < 1 || K > N || (D != 3 && D != 4)) return FACL_E_SHAPE;
    if (M > 65535) return FACL_E_SHAPE;
    if (D == 4 && ((((uintptr_t)points) & 15) || (xt && (((uintptr_t)xt) & 15)))) return FACL_E_ALIGN;
    if (M == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    return D == 4 ? dispatch_group<4>(points, M, N, S, K, r2, idx, xt, yt, st)
                  : dispatch_group<3>(points, M, N, S, K, r2, idx, xt, yt, st);
}

extern "C" int facl_version(void) { return (1 << 16) | 0; }
