// Farthest point sampling (one wave per cloud, points and running min-dist^2 in registers) and
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

template <typename T, int NPL>
__global__ __launch_bounds__(64) void k_fps(const T* __restrict__ xyz, int N, int ld, int m,
                                            const int32_t* __restrict__ start, int32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* xs = reinterpret_cast<T*>(lds_raw);
    T* ys = xs + N;
    T* zs = ys + N;
    const int cloud = blockIdx.x;
    const int lane = lane_id();
    const T* base = xyz + (size_t)cloud * N * ld;

    // coordinates live in LDS (SoA, conflict-free reads); only the running min-dist^2 is in registers
    T md[NPL];
    for (int i = lane; i < N; i += 64) {
        xs[i] = base[(size_t)i * ld + 0]; ys[i] = base[(size_t)i * ld + 1]; zs[i] = base[(size_t)i * ld + 2];
    }
    __syncthreads();

    int cur = start[cloud];
    if (lane == 0) out[(size_t)cloud * m] = cur;
    {
        const T cx = xs[cur], cy = ys[cur], cz = zs[cur];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int i = j * 64 + lane;
            const int ii = i < N ? i : 0;
            const T dx = sub_rn(xs[ii], cx), dy = sub_rn(ys[ii], cy), dz = sub_rn(zs[ii], cz);
            const T d = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
            md[j] = (i < N) ? d : (T)-1;          // padding can never win the argmax (real d >= 0)
        }
    }
    for (int s = 1; s < m; ++s) {
        // argmax with np.argmax tie-break (lowest index)
        T bv = md[0];
        int bi = lane;
#pragma unroll
        for (int j = 1; j < NPL; ++j)
            if (md[j] > bv) { bv = md[j]; bi = j * 64 + lane; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const T ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        cur = bi;
        if (lane == 0) out[(size_t)cloud * m + s] = cur;
        if (s < m - 1) {                          // cn3d_data_load.py:315
            const T cx = xs[cur], cy = ys[cur], cz = zs[cur];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int i = j * 64 + lane;
                const int ii = i < N ? i : 0;
                const T dx = sub_rn(xs[ii], cx), dy = sub_rn(ys[ii], cy), dz = sub_rn(zs[ii], cz);
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
    const size_t lds = (size_t)N * 3 * sizeof(T);
    dim3 grid(M), block(64);
    if (N <= 512) hipLaunchKernelGGL((k_fps<T, 8>), grid, block, lds, st, xyz, N, ld, m, start, out);
    else if (N <= 1024) hipLaunchKernelGGL((k_fps<T, 16>), grid, block, lds, st, xyz, N, ld, m, start, out);
    else if (N <= 2048) hipLaunchKernelGGL((k_fps<T, 32>), grid, block, lds, st, xyz, N, ld, m, start, out);
    else hipLaunchKernelGGL((k_fps<T, 64>), grid, block, lds, st, xyz, N, ld, m, start, out);
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
