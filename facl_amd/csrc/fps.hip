// Farthest point sampling (one 4-wave workgroup per cloud, points and running min-dist^2 in registers) and
// the FPS reorder.  Replaces the NumPy loops of cn3d_data_load.py:301-320 / cn3D_data_set.py:665-694.
// Roofline: HBM (the cloud is read once: N*ld*sizeof(T) per cloud; the m passes stay on chip).
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ T sub_rn(T a, T b);
template <> __device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
template <> __device__ __forceinline__ double sub_rn(double a, double b) { return __dsub_rn(a, b); }
template <typename T> __device__ __forceinline__ T mul_rn(T a, T b);
template <> __device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
template <> __device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
template <typename T> __device__ __forceinline__ T add_rn(T a, T b);
template <> __device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
template <> __device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }

// ---- wave-wide argmax on the DPP crossbar --------------------------------------------------------------------------------
// __shfl_xor lowers to ds_bpermute (an LDS round trip, ~100 cycles) and the (value, index) butterfly needs 12 of them in
// a dependent chain per pick: that chain, not the distance arithmetic, was the cost of a pick.  DPP row shifts / row
// broadcasts move data between lanes inside the VALU: max over the wave in 6 dependent v_max, then the lowest index among
// the lanes that hold the maximum in 6 dependent v_min.  Result valid in lane 63, broadcast with readlane.
template <int CTRL, int RMASK, int BMASK>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RMASK, BMASK, false); }
__device__ __forceinline__ float wave_max_dpp(float v) {
#define FPS_STEP(C, R, B) v = fmaxf(v, __builtin_bit_cast(float, dpp_i<C, R, B>(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v))))
    FPS_STEP(0x111, 0xf, 0xf); FPS_STEP(0x112, 0xf, 0xf); FPS_STEP(0x114, 0xf, 0xf); FPS_STEP(0x118, 0xf, 0xf);   // row_shr 1,2,4,8
    FPS_STEP(0x142, 0xa, 0xf); FPS_STEP(0x143, 0xc, 0xf);                                                         // row_bcast 15, 31
#undef FPS_STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_max_dpp(double v) {
#define FPS_STEP(C, R, B) {                                                                                  \
        const long long b = __builtin_bit_cast(long long, v);                                                 \
        const int lo = dpp_i<C, R, B>((int)b, (int)b), hi = dpp_i<C, R, B>((int)(b >> 32), (int)(b >> 32));    \
        const double o = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);                    \
        v = o > v ? o : v; }
    FPS_STEP(0x111, 0xf, 0xf) FPS_STEP(0x112, 0xf, 0xf) FPS_STEP(0x114, 0xf, 0xf) FPS_STEP(0x118, 0xf, 0xf)
    FPS_STEP(0x142, 0xa, 0xf) FPS_STEP(0x143, 0xc, 0xf)
#undef FPS_STEP
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int wave_min_dpp(int v) {                      // non-negative values
#define FPS_STEP(C, R, B) { const int o = dpp_i<C, R, B>(v, v); v = o < v ? o : v; }
    FPS_STEP(0x111, 0xf, 0xf) FPS_STEP(0x112, 0xf, 0xf) FPS_STEP(0x114, 0xf, 0xf) FPS_STEP(0x118, 0xf, 0xf)
    FPS_STEP(0x142, 0xa, 0xf) FPS_STEP(0x143, 0xc, 0xf)
#undef FPS_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

// One workgroup of FPS_WAVES waves per cloud: thread t owns points t, t + 256, ... with their coordinates AND their
// running min-dist^2 in registers (round 2 kept one wave per cloud and re-read the coordinates from LDS on every pick:
// 768 waves on 1,024 SIMDs, 96 ds_read per lane and pick).  Per pick: NPL distance updates per lane, an in-wave argmax
// (np.argmax tie rule: lowest index among equal values), one LDS slot per wave + ONE workgroup barrier (the slots are
// double-buffered by pick parity), then every wave reads the four candidates and the winner's coordinates (LDS broadcast).
// Same arithmetic in the same order per point as before: identical picks.
constexpr int FPS_WAVES = 4;
template <typename T, int NPL>
__global__ __launch_bounds__(64 * FPS_WAVES) void k_fps(const T* __restrict__ xyz, int N, int ld, int m,
                                                        const int32_t* __restrict__ start, int32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* xs = reinterpret_cast<T*>(lds_raw);
    T* ys = xs + N;
    T* zs = ys + N;
    T* cand_v = zs + N;                                   // [2][FPS_WAVES]
    int* cand_i = reinterpret_cast<int*>(cand_v + 2 * FPS_WAVES);
    const int cloud = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* base = xyz + (size_t)cloud * N * ld;
    constexpr int TPB = 64 * FPS_WAVES;

    T px[NPL], py[NPL], pz[NPL], md[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int i = j * TPB + tid;
        const int ii = i < N ? i : 0;
        px[j] = base[(size_t)ii * ld + 0]; py[j] = base[(size_t)ii * ld + 1]; pz[j] = base[(size_t)ii * ld + 2];
        if (i < N) { xs[i] = px[j]; ys[i] = py[j]; zs[i] = pz[j]; }
    }
    __syncthreads();

    int cur = start[cloud];
    if (tid == 0) out[(size_t)cloud * m] = cur;
    {
        const T cx = xs[cur], cy = ys[cur], cz = zs[cur];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int i = j * TPB + tid;
            const T dx = sub_rn(px[j], cx), dy = sub_rn(py[j], cy), dz = sub_rn(pz[j], cz);
            const T d = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
            md[j] = (i < N) ? d : (T)-1;          // padding can never win the argmax (real d >= 0)
        }
    }
    for (int s = 1; s < m; ++s) {
        // argmax with np.argmax tie-break (lowest index): in-lane (ascending index), in-wave, then across the waves
        T bv = md[0];
        int bi = tid;
#pragma unroll
        for (int j = 1; j < NPL; ++j)
            if (md[j] > bv) { bv = md[j]; bi = j * TPB + tid; }
        {   // wave argmax: the maximum, then the lowest index among the lanes that hold it (padding is -1, real values >= 0)
            const T wm = wave_max_dpp(bv);
            bi = wave_min_dpp(bv == wm ? bi : 0x7fffffff);
            bv = wm;
        }
        const int par = (s & 1) * FPS_WAVES;
        if (lane == 0) { cand_v[par + wave] = bv; cand_i[par + wave] = bi; }
        __syncthreads();
        bv = cand_v[par]; bi = cand_i[par];
#pragma unroll
        for (int w = 1; w < FPS_WAVES; ++w) {
            const T ov = cand_v[par + w];
            const int oi = cand_i[par + w];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        cur = bi;
        if (tid == 0) out[(size_t)cloud * m + s] = cur;
        if (s < m - 1) {                          // cn3d_data_load.py:315
            const T cx = xs[cur], cy = ys[cur], cz = zs[cur];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int i = j * TPB + tid;
                const T dx = sub_rn(px[j], cx), dy = sub_rn(py[j], cy), dz = sub_rn(pz[j], cz);
                const T d = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
                md[j] = (i < N) ? (d < md[j] ? d : md[j]) : (T)-1;
            }
        }
    }
}

template <typename T>
int fps_dispatch(const T* xyz, int M, int N, int ld, int m, const int32_t* start, int32_t* out, void* stream) {
    if (!xyz || !start || !out) return FACL_E_NULL;
    if (M < 0 || N < 1 || N > 4096 || m < 1 || ld < 3) return FACL_E_SHAPE;
    if (M == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)N * 3 * sizeof(T) + 2 * FPS_WAVES * (sizeof(T) + sizeof(int));
    dim3 grid(M), block(64 * FPS_WAVES);
    if (N <= 512) hipLaunchKernelGGL((k_fps<T, 2>), grid, block, lds, st, xyz, N, ld, m, start, out);
    else if (N <= 1024) hipLaunchKernelGGL((k_fps<T, 4>), grid, block, lds, st, xyz, N, ld, m, start, out);
    else if (N <= 2048) hipLaunchKernelGGL((k_fps<T, 8>), grid, block, lds, st, xyz, N, ld, m, start, out);
    else hipLaunchKernelGGL((k_fps<T, 16>), grid, block, lds, st, xyz, N, ld, m, start, out);
    return facl_launch_status();
}

// picks first, then the unpicked rows in ascending order, truncated to N rows.
__global__ __launch_bounds__(256) void k_fps_reorder(const float* __restrict__ points, int N, int D,
                                                     const int32_t* __restrict__ picks, int m,
                                                     float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) int lds_i[];
    int* src = lds_i;            // N: source row of each output row
    int* flag = lds_i + N;       // N: 1 if picked
    __shared__ int wave_tot[4];
    __shared__ int carry;
    const int cloud = blockIdx.x;
    const float* in = points + (size_t)cloud * N * D;
    float* o = out + (size_t)cloud * N * D;
    for (int i = threadIdx.x; i < N; i += 256) flag[i] = 0;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += 256) {
        const int p = picks[(size_t)cloud * m + i];
        flag[p] = 1;
        if (i < N) src[i] = p;
    }
    __syncthreads();
    // stable compaction of the unpicked rows (block-wide scan, 256 rows per round)
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int b0 = 0; b0 < N; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const bool un = (i < N) && !flag[i];
        const unsigned long long bm = __ballot(un);
        if (lane == 0) wave_tot[wave] = __popcll(bm);
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        const int pos = m + off + __popcll(bm & lanemask_lt());
        if (un && pos < N) src[pos] = i;
        __syncthreads();
        if (threadIdx.x == 0) carry += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    for (int e = threadIdx.x; e < N * D; e += 256) {
        const int r = e / D, ch = e - r * D;
        o[e] = in[(size_t)src[r] * D + ch];
    }
}

}  // namespace

extern "C" int facl_fps_f32(const float* xyz, int M, int N, int ld, int m, const int32_t* start, int32_t* out,
                            void* stream) {
    return fps_dispatch<float>(xyz, M, N, ld, m, start, out, stream);
}
extern "C" int facl_fps_f64(const double* xyz, int M, int N, int ld, int m, const int32_t* start, int32_t* out,
                            void* stream) {
    return fps_dispatch<double>(xyz, M, N, ld, m, start, out, stream);
}
extern "C" int facl_fps_reorder(const float* points, int M, int N, int D, const int32_t* picks, int m, float* out,
                                void* stream) {
    if (!points || !picks || !out) return FACL_E_NULL;
    if (M < 0 || N < 1 || N > 16384 || D < 1 || m < 0 || m > N) return FACL_E_SHAPE;
    if (M == 0) return 0;
    hipLaunchKernelGGL(k_fps_reorder, dim3(M), dim3(256), (size_t)N * 2 * sizeof(int), (hipStream_t)stream, points,
                       N, D, picks, m, out);
    return facl_launch_status();
}
