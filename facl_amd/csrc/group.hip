// kNN-then-radius grouping, fused with the gather + centre-subtract.
// Replaces the torch op chain of utils_my.py:265-284 (expand/sub/mul/sum -> topk -> S masked
// assigns -> gather -> subtract -> transpose views).  The (M,S,N) distance matrix is never
// materialised: one wave owns one centroid row, keeps its N distances in registers and finds
// the K-th smallest by an MSB-first radix select on the float bit patterns.
//
// Roofline: HBM.  Algorithmic bytes per cloud = N*D*4 (read) + S*K*(4 + 4*D) + S*12 (write).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int GROUP_THREADS = 256;           // 4 waves
constexpr int CENTROIDS_PER_WG = 16;         // 4 per wave (the default; the kernel takes the count as a template argument)
constexpr int CKEYS = 4;                     // keys per lane once the candidate set fits (radix select, second phase)
constexpr int CKE = 7;                       // K <= 256: entries per lane of the compacted list (keys under the candidate range's top)
constexpr int CAP = 64 * CKE;                // capacity of that list (per wave: CAP (key, index) pairs of 8 bytes in LDS)
constexpr int SLOT_BYTES = CAP * 8;          // per wave: the compacted list; its first words are re-used as the emission slot

// dist^2 exactly as the reference's fp32 chain: (dx*dx + dy*dy) + dz*dz, no FMA contraction.
__device__ __forceinline__ float dist2_exact(float px, float py, float pz, float cx, float cy, float cz) {
    const float dx = __fsub_rn(px, cx), dy = __fsub_rn(py, cy), dz = __fsub_rn(pz, cz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// base + (set bits of mask m below this lane): the hardware's masked bit count, two instructions with the base folded in
// (`base + __popcll(m & lanemask_lt)` compiles to two ands, two v_bcnt and an add)
__device__ __forceinline__ int rank_below(unsigned long long m, int base) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base));
}

template <int D, int NPL, int CPW = CENTROIDS_PER_WG, bool FULL = false>
__global__ __launch_bounds__(GROUP_THREADS) void k_group(const float* __restrict__ points, int N, int S,
                                                         int K, float r2, int32_t* __restrict__ idx_out,
                                                         float* __restrict__ xt_out, float* __restrict__ yt_out,
                                                         int clipB) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* ys = xs + N;
    float* zs = ys + N;
    float* cs = zs + N;                       // only when D == 4
    const int m = blockIdx.y;
    // clipB > 0: the input is the loader's (B, G, N, D) clip-major batch and cloud m = g*B + b is read in place (the
    // permute(1,0,2,3).reshape copy of cn3d_train_motion_GL.py:226 is folded into this address)
    const size_t src = clipB > 0 ? (size_t)(m % clipB) * (gridDim.y / clipB) + m / clipB : (size_t)m;
    const float* cloud = points + src * N * D;

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
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: the per-wave LDS slot's address stays scalar

    uint32_t eguess = 122;                    // wave-persistent exponent guess of the K-th distance^2 (2^-5 .. 2^-4 to start with)
    for (int ci = wave; ci < CPW; ci += GROUP_THREADS / 64) {
        const int c = blockIdx.x * CPW + ci;                     // wave-uniform
        if (c >= S) break;
        const float cx = xs[c], cy = ys[c], cz = zs[c];

        uint32_t key[NPL];
        // Two keys per packed instruction, written as 2-vectors in the order the paired LDS reads deliver them (left to the
        // vectoriser the same packing came with six register moves per pair: the pair order reversed).  Same arithmetic as
        // dist2_exact -- (dx*dx + dy*dy) + dz*dz, every operation rounded, nothing fused (the file is built with contraction off).
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < NPL; j += 2) {
            const int i0 = j * 64 + lane, i1 = i0 + 64;
            const f32x2 px = {xs[i0], xs[i1]}, py = {ys[i0], ys[i1]}, pz = {zs[i0], zs[i1]};
            const f32x2 dx = px - cx, dy = py - cy, dz = pz - cz;
            const f32x2 d = (dx * dx + dy * dy) + dz * dz;
            // FULL (N == 64 * NPL, the headline's 2048): no padding keys, no compare / select per key
            key[j] = (FULL || i0 < N) ? __float_as_uint(d.x) : 0x7F800000u;
            key[j + 1] = (FULL || i1 < N) ? __float_as_uint(d.y) : 0x7F800000u;
        }

        // radix select on the float bit patterns (non-negative floats order like their bits), MSB first.
        // Invariant: `cand` keys share the resolved prefix, `remaining` of them belong to the K smallest.
        // Early exit as soon as cand == remaining (all candidates are kept): on continuous data the candidate
        // set shrinks to `remaining` after ~log2(N) mantissa bits, i.e. roughly half of the 31 rounds.
        // One round = ONE compare per key: with `below` = number of keys under the resolved prefix, the keys under
        // pivot = prefix | bit number below + (candidates whose bit is 0), so the wave count of (key < pivot) -- a ballot
        // per register, popcounted on the scalar unit -- decides the bit; no mask / equality test per key.
        uint32_t prefix = 0, hi = 0;
        int below = 0, cand = NPL * 64;                         // padding keys (+inf) are ordinary candidates
        int bit = 30;
        auto count_below = [&](uint32_t pivot) {                // keys under pivot, over the whole wave
            int c = 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) c += __popcll(__ballot(key[j] < pivot));
            return c;
        };
        // Exponent bucket first.  MSB-first, the 8 rounds of bits 30..23 run at full width (NPL compares + ballots each)
        // and, on clouds of one scale, resolve the SAME exponent for every centroid.  Start from the exponent the wave's
        // previous centroid found: two counts decide whether the K-th smallest lies in [e*2^23, (e+1)*2^23) -- that IS the
        // loop invariant below < K <= below + cand at bit 22 -- and a miss moves the bucket one step (one more count).
        // After a few misses the plain MSB-first loop takes over; exactness never depends on the guess.
        {
            uint32_t e = eguess;
            int cl = e ? count_below(e << 23) : 0, ch = count_below((e + 1) << 23);
            int tries = 0;
            for (; tries < 4; ++tries) {
                if (cl >= K) { ch = cl; --e; cl = e ? count_below(e << 23) : 0; }                 // cl >= K >= 1 implies e > 0
                else if (ch < K) { cl = ch; ++e; ch = count_below((e + 1) << 23); }               // e + 1 <= 255: K <= N real keys are < +inf
                else break;
            }
            if (cl < K && K <= ch) {
                prefix = e << 23; hi = 0xFF800000u; below = cl; cand = ch - cl; bit = 22;
                eguess = e;
            }
        }
        if (K <= 256) {
            // ---- K <= 256 (round 4): ONE pass over the registers, then everything on a compacted list ------------------------
            // Full-width rounds only until the keys under the TOP of the candidate range (the `below` kept ones and the `cand`
            // candidates) fit the per-wave list: CAP entries, i.e. normally none after the exponent bucket.  The list holds
            // (key, index) in ascending index order, so the remaining select rounds AND the emission walk <= CKE entries per
            // lane instead of NPL registers per lane twice (candidate gather + emission pass: ~450 of the ~1,100 vector
            // instructions per centroid of the previous form).
            for (; bit >= 0 && cand != K - below && below + cand > CAP; --bit) {
                const uint32_t pivot = prefix | (1u << bit);
                const int c = count_below(pivot);
                if (c < K) { prefix = pivot; cand -= c - below; below = c; }
                else cand = c - below;
                hi = ~((1u << bit) - 1u);
            }
        }
        // (more than CAP entries can only be left when the rounds ran out of bits on a mass of exact ties: the full-width tie
        // ranking below handles that)
        if (K <= 256 && below + cand <= CAP) {
            char* sl = reinterpret_cast<char*>(lds + (size_t)D * N) + SLOT_BYTES * wave;
            uint2* slotp = reinterpret_cast<uint2*>(sl);        // (key, index): one 8-byte store / load per entry
            int E = 0;
            {
                const uint32_t top = hi ? prefix + (~hi + 1u) : 0x80000000u;      // exclusive top of the candidate range
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const bool inc = key[j] < top;
                    const unsigned long long m = __ballot(inc);
                    if (inc) {
                        // two dword stores (ds_write2_b32): an 8-byte store needs key and index in a register PAIR, i.e. a copy
                        // per entry into a pair the previous store may still be reading (an lgkmcnt wait every other entry)
                        // (the running entry count stays on the scalar side: folded into the store's base address, not into the rank)
                        uint32_t* e2 = reinterpret_cast<uint32_t*>(slotp + E) + 2 * rank_below(m, 0);
                        e2[0] = key[j];
                        e2[1] = (uint32_t)(j * 64 + lane);
                    }
                    E += __popcll(m);
                }
            }
            // E = below + cand <= CAP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off
            uint32_t ck[CKE];
            int ci[CKE];
#pragma unroll
            for (int t = 0; t < CKE; ++t) {
                const int pp = lane + 64 * t;
                const uint2 e = pp < E ? slotp[pp] : make_uint2(0xFFFFFFFFu, 0u);   // padding: never under a pivot, never kept
                ck[t] = e.x;
                ci[t] = (int)e.y;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the list is in registers: its LDS words become the emission slot
            for (; bit >= 0 && cand != K - below; --bit) {
                const uint32_t pivot = prefix | (1u << bit);
                int c = 0;                                      // entries under the pivot (the `below` ones are all of them)
#pragma unroll
                for (int t = 0; t < CKE; ++t)
                    if (64 * t < E) c += __popcll(__ballot(ck[t] < pivot));
                if (c < K) { prefix = pivot; cand -= c - below; below = c; }
                else cand = c - below;
                hi = ~((1u << bit) - 1u);
            }
            const int remaining = K - below;
            uint32_t* es = reinterpret_cast<uint32_t*>(sl);     // K <= 256 words
            int base = 0;
            if (cand == remaining) {
                const uint32_t upper = hi ? prefix + (~hi + 1u) : 0x80000000u;
#pragma unroll
                for (int t = 0; t < CKE; ++t) {
                    if (64 * t >= E) break;
                    const bool take = ck[t] < upper;
                    const unsigned long long tm = __ballot(take);
                    if (take) es[rank_below(tm, base)] = (uint32_t)ci[t];
                    base += __popcll(tm);
                }
            } else {                                            // exact ties at the K-th value: the first `remaining` in index order
                int eq_taken = 0;
#pragma unroll
                for (int t = 0; t < CKE; ++t) {
                    if (64 * t >= E) break;
                    const uint32_t kh = ck[t] & hi;
                    const bool is_eq = kh == prefix && lane + 64 * t < E;
                    const unsigned long long eqm = __ballot(is_eq);
                    const int eq_rank = rank_below(eqm, eq_taken);
                    eq_taken += __popcll(eqm);
                    const bool take = (kh < prefix) || (is_eq && eq_rank < remaining);
                    const unsigned long long tm = __ballot(take);
                    if (take) es[rank_below(tm, base)] = (uint32_t)ci[t];
                    base += __popcll(tm);
                }
            }
            const size_t grp = (size_t)m * S + c;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off
            for (int pos = lane; pos < K; pos += 64) {
                const int i = (int)es[pos];
                const float px = xs[i], py = ys[i], pz = zs[i];
                const bool outside = dist2_exact(px, py, pz, cx, cy, cz) > r2;              // strict >, utils_my.py:272
                const int id = outside ? c : i;
                const size_t o = grp * K + pos;
                if (idx_out) idx_out[o] = id;
                if (xt_out) {
                    const float gx = outside ? 0.f : __fsub_rn(px, cx), gy = outside ? 0.f : __fsub_rn(py, cy);
                    const float gz = outside ? 0.f : __fsub_rn(pz, cz);
                    if (D == 4) *reinterpret_cast<float4*>(xt_out + o * 4) = make_float4(gx, gy, gz, cs[id]);
                    else { xt_out[o * 3 + 0] = gx; xt_out[o * 3 + 1] = gy; xt_out[o * 3 + 2] = gz; }
                }
            }
            asm volatile("" ::: "memory");                      // the slot is rewritten by the next centroid
            if (yt_out && lane < 3) yt_out[grp * 3 + lane] = (lane == 0) ? cx : (lane == 1) ? cy : cz;
            continue;
        }
        for (; bit >= 0 && cand != K - below && cand > 64 * CKEYS; --bit) {
            const uint32_t pivot = prefix | (1u << bit);
            const int c = count_below(pivot);
            if (c < K) { prefix = pivot; cand -= c - below; below = c; }   // the K-th smallest is >= pivot: bit = 1
            else cand = c - below;
            hi = ~((1u << bit) - 1u);                           // `bit` and everything above it are resolved now
        }
        int remaining = K - below;
        if (bit >= 0 && cand != remaining) {
            // At most 64*CKEYS candidates are left (the range that holds the K-th smallest halves every round): gather
            // them into CKEYS keys per lane through a small LDS slot and finish the remaining rounds -- typically half
            // of them -- on those registers instead of walking all NPL.  Only prefix / hi / remaining come out of it;
            // the emission below still uses the original registers, so no index bookkeeping is needed.
            uint32_t* slot = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds + (size_t)D * N) + SLOT_BYTES * wave);
            int base = 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const bool isc = (key[j] & hi) == prefix;
                const unsigned long long m = __ballot(isc);
                if (isc) slot[rank_below(m, base)] = key[j];
                base += __popcll(m);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off
            uint32_t ck[CKEYS];
#pragma unroll
            for (int i = 0; i < CKEYS; ++i) ck[i] = lane + 64 * i < cand ? slot[lane + 64 * i] : 0xFFFFFFFFu;   // never < pivot
            int cbelow = 0;                                     // compacted keys under the prefix (none yet: all are >= it)
            for (; bit >= 0 && cand != remaining; --bit) {
                const uint32_t pivot = prefix | (1u << bit);
                int c = 0;
#pragma unroll
                for (int i = 0; i < CKEYS; ++i) c += __popcll(__ballot(ck[i] < pivot));
                const int zeros = c - cbelow;                   // candidates whose `bit` is 0
                if (zeros < remaining) { prefix = pivot; remaining -= zeros; cand -= zeros; cbelow = c; }
                else cand = zeros;
                hi = ~((1u << bit) - 1u);
            }
            asm volatile("" ::: "memory");                      // the slot is rewritten by the next centroid
        }
        // keys with (key & hi) < prefix are kept; of those equal to prefix under `hi`, the first `remaining`
        // in index order (all of them when the loop exited early; exact-tie rule otherwise)

        const size_t grp = (size_t)m * S + c;
        // Emission in two steps: the kept indices are compacted in ascending order into a per-wave LDS slot (ballot /
        // prefix-popcount positions; ~K/NPL lanes are active per pass), then ALL K neighbours are gathered, radius-tested
        // (the distance is recomputed from the coordinates the gather reads anyway: same exact arithmetic), centred and
        // stored by K lanes at once -- K/64 full-width store passes per output instead of NPL passes of a few lanes each.
        uint32_t* eslot = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds + (size_t)D * N) + SLOT_BYTES * wave);      // >= 256 words per wave
        // the direct form of one kept neighbour (index i, output slot pos): radius rule, gather, centre, store
        auto emit = [&](int i, int pos) {
            const float px = xs[i], py = ys[i], pz = zs[i];
            const bool outside = dist2_exact(px, py, pz, cx, cy, cz) > r2;              // strict >, utils_my.py:272
            const int id = outside ? c : i;
            const size_t o = grp * K + pos;
            if (idx_out) idx_out[o] = id;
            if (xt_out) {
                // the centroid's own centred coordinates are exactly +0 (x - x)
                const float gx = outside ? 0.f : __fsub_rn(px, cx), gy = outside ? 0.f : __fsub_rn(py, cy);
                const float gz = outside ? 0.f : __fsub_rn(pz, cz);
                if (D == 4) *reinterpret_cast<float4*>(xt_out + o * 4) = make_float4(gx, gy, gz, cs[id]);
                else { xt_out[o * 3 + 0] = gx; xt_out[o * 3 + 1] = gy; xt_out[o * 3 + 2] = gz; }
            }
        };
        int base = 0;
        if (cand == remaining) {
            // every candidate is kept (the usual exit on continuous data): kept <=> key < prefix + 2^(lowest resolved bit),
            // ONE compare per key, no tie ranking
            const uint32_t upper = hi ? prefix + (~hi + 1u) : 0x80000000u;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const bool take = key[j] < upper;
                const unsigned long long tm = __ballot(take);
                if (take) {
                    const int pos = rank_below(tm, base);
                    if (K <= 256) eslot[pos] = (uint32_t)(j * 64 + lane);
                    else emit(j * 64 + lane, pos);
                }
                base += __popcll(tm);
            }
        } else {
            int eq_taken = 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint32_t kh = key[j] & hi;
                const bool is_eq = kh == prefix;
                const unsigned long long eqm = __ballot(is_eq);
                const int eq_rank = rank_below(eqm, eq_taken);
                eq_taken += __popcll(eqm);
                const bool take = (kh < prefix) || (is_eq && eq_rank < remaining);
                const unsigned long long tm = __ballot(take);
                if (take) {
                    const int pos = rank_below(tm, base);
                    if (K <= 256) eslot[pos] = (uint32_t)(j * 64 + lane);
                    else emit(j * 64 + lane, pos);
                }
                base += __popcll(tm);
            }
        }
        if (K <= 256) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off
            for (int pos = lane; pos < K; pos += 64) emit((int)eslot[pos], pos);
            asm volatile("" ::: "memory");                      // the slot is rewritten by the next centroid
        }
        if (yt_out && lane < 3) yt_out[grp * 3 + lane] = (lane == 0) ? cx : (lane == 1) ? cy : cz;
    }
}

template <int D, int NPL>
int launch_group(const float* points, int M, int N, int S, int K, float r2, int32_t* idx, float* xt, float* yt,
                 int clipB, hipStream_t st) {
    const size_t lds = (size_t)N * D * sizeof(float) + 4 * SLOT_BYTES;   // cloud (SoA, D arrays) + one compacted-list / emission slot per wave
    // centroids per workgroup (A/B knob FACL_GROUP_CPW: 16 = four workgroups stage each cloud, 32 = two, 64 = one)
    static const int cpw = getenv("FACL_GROUP_CPW") ? atoi(getenv("FACL_GROUP_CPW")) : CENTROIDS_PER_WG;
    if (cpw == 32) {
        hipLaunchKernelGGL((k_group<D, NPL, 32>), dim3((S + 31) / 32, M), dim3(GROUP_THREADS), lds, st, points, N, S, K, r2, idx, xt, yt, clipB);
    } else if (cpw == 8) {
        hipLaunchKernelGGL((k_group<D, NPL, 8>), dim3((S + 7) / 8, M), dim3(GROUP_THREADS), lds, st, points, N, S, K, r2, idx, xt, yt, clipB);
    } else if (cpw == 4) {
        hipLaunchKernelGGL((k_group<D, NPL, 4>), dim3((S + 3) / 4, M), dim3(GROUP_THREADS), lds, st, points, N, S, K, r2, idx, xt, yt, clipB);
    } else if (cpw == 64) {
        hipLaunchKernelGGL((k_group<D, NPL, 64>), dim3((S + 63) / 64, M), dim3(GROUP_THREADS), lds, st, points, N, S, K, r2, idx, xt, yt, clipB);
    } else if (N == 64 * NPL) {
        hipLaunchKernelGGL((k_group<D, NPL, CENTROIDS_PER_WG, true>), dim3((S + CENTROIDS_PER_WG - 1) / CENTROIDS_PER_WG, M), dim3(GROUP_THREADS), lds, st, points, N, S, K, r2, idx, xt, yt, clipB);
    } else {
        hipLaunchKernelGGL((k_group<D, NPL>), dim3((S + CENTROIDS_PER_WG - 1) / CENTROIDS_PER_WG, M), dim3(GROUP_THREADS), lds, st, points, N, S, K, r2, idx, xt, yt, clipB);
    }
    return facl_launch_status();
}

template <int D>
int dispatch_group(const float* points, int M, int N, int S, int K, float r2, int32_t* idx, float* xt, float* yt,
                   int clipB, hipStream_t st) {
    if (N <= 512) return launch_group<D, 8>(points, M, N, S, K, r2, idx, xt, yt, clipB, st);
    if (N <= 1024) return launch_group<D, 16>(points, M, N, S, K, r2, idx, xt, yt, clipB, st);
    if (N <= 2048) return launch_group<D, 32>(points, M, N, S, K, r2, idx, xt, yt, clipB, st);
    return launch_group<D, 64>(points, M, N, S, K, r2, idx, xt, yt, clipB, st);
}

}  // namespace

static int group_entry(const float* points, int M, int N, int D, int S, int K, float r2, int32_t* idx, float* xt,
                       float* yt, int clipB, void* stream) {
    if (!points) return FACL_E_NULL;
    if (M < 0 || N < 1 || N > 4096 || S < 1 || S > N || K < 1 || K > N || (D != 3 && D != 4)) return FACL_E_SHAPE;
    if (M > 65535 || clipB < 0 || (clipB > 0 && M % clipB)) return FACL_E_SHAPE;
    if (D == 4 && ((((uintptr_t)points) & 15) || (xt && (((uintptr_t)xt) & 15)))) return FACL_E_ALIGN;
    if (M == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    return D == 4 ? dispatch_group<4>(points, M, N, S, K, r2, idx, xt, yt, clipB, st)
                  : dispatch_group<3>(points, M, N, S, K, r2, idx, xt, yt, clipB, st);
}

extern "C" int facl_group(const float* points, int M, int N, int D, int S, int K, float r2, int32_t* idx,
                          float* xt, float* yt, void* stream) {
    return group_entry(points, M, N, D, S, K, r2, idx, xt, yt, 0, stream);
}

extern "C" int facl_group_clips(const float* clips, int B, int G, int N, int D, int S, int K, float r2, int32_t* idx,
                                float* xt, float* yt, void* stream) {
    if (B < 1 || G < 1) return FACL_E_SHAPE;
    return group_entry(clips, B * G, N, D, S, K, r2, idx, xt, yt, B, stream);
}

extern "C" int facl_version(void) { return (1 << 16) | 4; }   // minor = the round whose signatures the header describes
