// Second-level grouping on channel-first features (utils_my.py:332-381, group_points_2 / group_points_2_3DV): the kNN +
// radius rule runs on the level-1 centroid coordinates through facl_group (group.hip: same select, same exact-tie rule,
// idx and centred xyz out); this file moves the FEATURES: a row gather by those indices (forward) and its transpose,
// a deterministic scatter-add (backward).  Features live row-major, (M, S1, C) -- the layout the level-1 pooling
// writes and the level-2 point-MLP GEMM reads; the reference's channel-first (M, 3+C, S2, K) tensor is a permuted view
// of the gathered rows (facl_amd/dense.py).
// Roofline: HBM.  Forward bytes per output row = 4*C read + 4*C (+12) written; the dense configuration's level 2
// (M*S2*K = 2.1 M rows of 256 channels per 8 clips) moves 4.3 GB per step in this kernel pair.
#include "common.h"

namespace {

// out[r][col_off + c] = feat[m][idx[r]][c]; with xyz != nullptr also out[r][0..2] = xyz[r][0..2] (col_off = 3)
template <bool VEC>
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ feat, int ldf, int S1, int C,
                                                     const int32_t* __restrict__ idx, long long rows, int rows_per_cloud,
                                                     const float* __restrict__ xyz, float* __restrict__ out, int ldo,
                                                     int col_off) {
    const int lane = threadIdx.x & 63;
    const long long w0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    for (long long r = w0; r < rows; r += nw) {
        const long long m = r / rows_per_cloud;
        const int id = idx[r];                                           // wave-uniform
        const float* src = feat + ((size_t)m * S1 + id) * ldf;
        float* dst = out + (size_t)r * ldo + col_off;
        if (VEC) {
            for (int c4 = lane; c4 < (C >> 2); c4 += 64)
                reinterpret_cast<float4*>(dst)[c4] = reinterpret_cast<const float4*>(src)[c4];
        } else {
            for (int c = lane; c < C; c += 64) dst[c] = src[c];
        }
        if (xyz && lane < 3) out[(size_t)r * ldo + lane] = xyz[(size_t)r * 3 + lane];
    }
}

// d_feat[m][s][c] = sum over the rows r of cloud m with idx[r] == s of drows[r][col_off + c].  One single-wave workgroup
// per (cloud, 64-channel chunk): an LDS accumulator [S1][64] (lane = channel: no two lanes ever touch one word), the
// cloud's rows added in index order -> deterministic, no atomics.
__global__ __launch_bounds__(64) void k_scatter_rows(const float* __restrict__ drows, int ldd, int col_off, int C,
                                                     const int32_t* __restrict__ idx, int rows_per_cloud, int S1,
                                                     float* __restrict__ dfeat) {
    extern __shared__ float acc[];                                       // [S1][64]
    const int lane = threadIdx.x, m = blockIdx.y, c = blockIdx.x * 64 + lane;
    for (int i = lane; i < S1 * 64; i += 64) acc[i] = 0.f;
    const int32_t* ix = idx + (size_t)m * rows_per_cloud;
    const float* src = drows + (size_t)m * rows_per_cloud * ldd + col_off + c;
    const bool on = c < C;
    int r = 0;
    for (; r + 8 <= rows_per_cloud; r += 8) {                            // 8 independent loads in flight, adds in order
        float v[8];
        int id[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { id[j] = ix[r + j]; v[j] = on ? src[(size_t)(r + j) * ldd] : 0.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[id[j] * 64 + lane] += v[j];
    }
    for (; r < rows_per_cloud; ++r) acc[ix[r] * 64 + lane] += on ? src[(size_t)r * ldd] : 0.f;
    if (on)
        for (int s = 0; s < S1; ++s) dfeat[((size_t)m * S1 + s) * C + c] = acc[s * 64 + lane];
}

// The same sum with the destination range of a cloud cut into `parts` slices: workgroup (part, cloud) owns the destinations
// [lo, lo + SP), keeps a [SP][C] accumulator in LDS (32 KiB for the dense level 2: 3-4 workgroups per CU instead of ONE
// single-wave workgroup per CU with the 128 KiB [S1][64] accumulator above), builds the ascending list of ITS rows once
// (wave 0: ballot compaction of the cloud's idx, 4 bytes per entry in LDS) and then every wave adds its own 64-channel
// chunks of those rows, 16 row loads in flight, in list order -> deterministic, no atomics; each row is read by exactly one
// workgroup.  3.3 -> ~0.7 ms for the dense configuration's 2.1 M rows x 256 channels.
__global__ __launch_bounds__(256) void k_scatter_rows_p(const float* __restrict__ drows, int ldd, int col_off, int C,
                                                        const int32_t* __restrict__ idx, int rows_per_cloud, int S1, int SP,
                                                        float* __restrict__ dfeat) {
    extern __shared__ float acc[];                                       // [SP][C] then the row list
    unsigned* list = reinterpret_cast<unsigned*>(acc + (size_t)SP * C);
    __shared__ int nlist;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = blockIdx.y;
    const int lo = blockIdx.x * SP, hi = lo + SP < S1 ? lo + SP : S1;
    for (int i = threadIdx.x; i < SP * C; i += 256) acc[i] = 0.f;
    const int32_t* ix = idx + (size_t)m * rows_per_cloud;
    if (wave == 0) {
        int n = 0;
        for (int r0 = 0; r0 < rows_per_cloud; r0 += 64) {
            const int r = r0 + lane;
            const int id = r < rows_per_cloud ? ix[r] : -1;
            const bool mine = id >= lo && id < hi;
            const unsigned long long b = __ballot(mine);
            if (mine) list[n + __popcll(b & lanemask_lt())] = (unsigned)r | ((unsigned)(id - lo) << 20);
            n += __popcll(b);
        }
        if (lane == 0) nlist = n;
    }
    __syncthreads();
    const int n = nlist;
    const float* src = drows + (size_t)m * rows_per_cloud * ldd + col_off;
    for (int c0 = 64 * wave; c0 < C; c0 += 256) {                        // this wave's channel chunks
        const int c = c0 + lane;
        const bool on = c < C;
        int e = 0;
        for (; e + 16 <= n; e += 16) {                                   // 16 row loads in flight per wave
            float v[16];
            unsigned ent[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                ent[j] = list[e + j];
                v[j] = on ? src[(size_t)(ent[j] & 0xfffffu) * ldd + c] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (on) acc[(ent[j] >> 20) * C + c] += v[j];
        }
        for (; e < n; ++e) {
            const unsigned ent = list[e];
            if (on) acc[(ent >> 20) * C + c] += src[(size_t)(ent & 0xfffffu) * ldd + c];
        }
    }
    __syncthreads();
    float* dst = dfeat + ((size_t)m * S1 + lo) * C;                      // [hi - lo][C] is contiguous
    for (int i = threadIdx.x; i < (hi - lo) * C; i += 256) dst[i] = acc[i];
}

}  // namespace

extern "C" int facl_gather_rows(const float* feat, int ldf, int M, int S1, int C, const int32_t* idx, int rows_per_cloud,
                                const float* xyz, float* out, int ldo, int col_off, void* stream) {
    if (!feat || !idx || !out) return FACL_E_NULL;
    if (M < 0 || S1 < 1 || C < 1 || rows_per_cloud < 1 || ldf < C || col_off < 0 || ldo < col_off + C) return FACL_E_SHAPE;
    if (xyz && col_off < 3) return FACL_E_SHAPE;
    if (M == 0) return 0;
    const long long rows = (long long)M * rows_per_cloud;
    const bool vec = !(C & 3) && !(ldf & 3) && !(ldo & 3) && !(col_off & 3) && !((((uintptr_t)feat) | ((uintptr_t)out)) & 15);
    const int grid = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (vec) hipLaunchKernelGGL((k_gather_rows<true>), dim3(grid), dim3(256), 0, st, feat, ldf, S1, C, idx, rows, rows_per_cloud, xyz, out, ldo, col_off);
    else hipLaunchKernelGGL((k_gather_rows<false>), dim3(grid), dim3(256), 0, st, feat, ldf, S1, C, idx, rows, rows_per_cloud, xyz, out, ldo, col_off);
    return facl_launch_status();
}

extern "C" int facl_scatter_rows(const float* drows, int ldd, int col_off, int M, int S1, int C, const int32_t* idx,
                                 int rows_per_cloud, float* dfeat, void* stream) {
    if (!drows || !idx || !dfeat) return FACL_E_NULL;
    if (M < 0 || S1 < 1 || C < 1 || rows_per_cloud < 1 || col_off < 0 || ldd < col_off + C) return FACL_E_SHAPE;
    if (M == 0) return 0;
    if (M > 65535) return FACL_E_SHAPE;
    {   // destination slices of <= 32 KiB of accumulator each (several workgroups per CU) + the row list
        int SP = (int)(32768 / ((size_t)C * sizeof(float)));
        if (SP > S1) SP = S1;
        const size_t lds = (size_t)SP * C * sizeof(float) + (size_t)rows_per_cloud * sizeof(unsigned);
        if (SP >= 1 && SP <= 4096 && rows_per_cloud <= (1 << 20) && lds <= 96 * 1024) {
            static size_t attr_p = 0;
            if (lds > attr_p) {
                hipError_t e = hipFuncSetAttribute((const void*)k_scatter_rows_p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return (int)e;
                attr_p = lds;
            }
            hipLaunchKernelGGL(k_scatter_rows_p, dim3((S1 + SP - 1) / SP, M), dim3(256), lds, (hipStream_t)stream, drows, ldd,
                               col_off, C, idx, rows_per_cloud, S1, SP, dfeat);
            return facl_launch_status();
        }
    }
    if ((size_t)S1 * 64 * sizeof(float) > 160 * 1024) return FACL_E_SHAPE;
    const size_t lds = (size_t)S1 * 64 * sizeof(float);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute((const void*)k_scatter_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_lds = lds;
    }
    hipLaunchKernelGGL(k_scatter_rows, dim3((C + 63) / 64, M), dim3(64), lds, (hipStream_t)stream, drows, ldd, col_off, C, idx,
                       rows_per_cloud, S1, dfeat);
    return facl_launch_status();
}
