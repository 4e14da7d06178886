// View construction of a batch of clips (cn3D_data_set.py:285-350 get_data_train, :654-663, :708-713,
// :734-749, :767-778): gather-with-replacement, jitter, x-mirror, y-rotation, temporal-channel select, float64 ->
// float32 and the (B,G,N,D) -> view-major (G*B,N,D) re-layout of cn3d_train_motion_GL.py:225-228, in ONE launch.
// The reference draws every random number from NumPy's global generator; to stay reproducible against it the HOST
// draws them in the reference's order (facl_amd/views.py) and this kernel consumes them: row indices (already
// composed with the non-zero filter of the temporal views), the standard-normal jitter draws and cos/sin of the two
// rotation angles (evaluated by NumPy in float64).  Arithmetic follows the reference's dtype walk exactly:
// jitter in float64, written back into the source-typed array, mirror / rotation on a float32 copy, rotation as
// float32 xyz times a float64 matrix rounded to float32.  HBM-bound: 16 B written per point.
#include "common.h"

namespace {

constexpr int NV = 10, NP = 512, NJ = 7;   // views, points per view (NUM_POINT, :24), jitter draws per clip

__device__ __forceinline__ double jit(double n) {          // np.clip(0.01 * n, -0.05, 0.05)
    const double v = 0.01 * n;
    return v < -0.05 ? -0.05 : (v > 0.05 ? 0.05 : v);
}

// block = one (clip, view); thread = one point
template <typename S>
__global__ __launch_bounds__(NP) void k_build_views(const S* __restrict__ src, int C, const int* __restrict__ idx,
                                                    const double* __restrict__ noise, const double* __restrict__ cs,
                                                    int B, float* __restrict__ out) {
    const int b = blockIdx.x / NV, v = blockIdx.x % NV, n = threadIdx.x;
    const long long row = idx[((size_t)b * NV + v) * NP + n];
    const S* r = src + row * C;
    const int c3 = v == 6 ? 4 : (v == 7 ? 7 : 3);            // temporal views: xyz + channel 4 / 7 (:116-117)
    S x[3] = {r[0], r[1], r[2]};
    const S w = r[c3];
    float o[4];
    o[3] = (float)w;
    // jitter slots per clip: rev 0,1 | ke1 2 | ke2 3,4 | ro1 5 | ro2 6
    const double* nz = noise + ((size_t)b * NJ * NP + n) * 3;
    auto jitter_into_src = [&](int slot) {                   // arr[:, :, :3] = jitter(arr[:, :, :3]) on the S-typed array
#pragma unroll
        for (int d = 0; d < 3; ++d) x[d] = (S)((double)x[d] + jit(nz[(size_t)slot * NP * 3 + d]));
    };
    if (v == 1 || v == 3) {                                  // jitter, then reverse_transform (:708-713)
        const int s0 = v == 1 ? 0 : 3;
        jitter_into_src(s0);
        float f[3] = {(float)x[0], (float)x[1], (float)x[2]};
        f[0] = -f[0];
#pragma unroll
        for (int d = 0; d < 3; ++d) o[d] = (float)((double)f[d] + jit(nz[(size_t)(s0 + 1) * NP * 3 + d]));
        o[3] = (float)w;
    } else if (v == 2) {
        jitter_into_src(2);
        o[0] = (float)x[0]; o[1] = (float)x[1]; o[2] = (float)x[2];
    } else if (v == 4 || v == 5) {                           // jitter, then rotate_trans (:734-749)
        jitter_into_src(v == 4 ? 5 : 6);
        const double c = cs[((size_t)b * 2 + (v - 4)) * 2], s = cs[((size_t)b * 2 + (v - 4)) * 2 + 1];
        const double fx = (double)(float)x[0], fy = (double)(float)x[1], fz = (double)(float)x[2];
        // [x y z] @ [[c,0,s],[0,1,0],[-s,0,c]]
        o[0] = (float)(fx * c + fz * (-s));
        o[1] = (float)fy;
        o[2] = (float)(fx * s + fz * c);
    } else {                                                 // raw, temporal, low-resolution views: plain gather
        o[0] = (float)x[0]; o[1] = (float)x[1]; o[2] = (float)x[2];
    }
    // view-major row g*B + b
    *reinterpret_cast<float4*>(out + (((size_t)v * B + b) * NP + n) * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace

template <typename S>
static int launch_views(const S* src, int64_t rows, int C, const int* idx, const double* noise, const double* cs, int B,
                        float* out, void* stream) {
    if (!src || !idx || !noise || !cs || !out) return FACL_E_NULL;
    if (B < 1 || rows < 1 || C < 8 || B > (1 << 20)) return FACL_E_SHAPE;
    hipLaunchKernelGGL((k_build_views<S>), dim3(B * NV), dim3(NP), 0, (hipStream_t)stream, src, C, idx, noise, cs, B, out);
    return facl_launch_status();
}

extern "C" int facl_build_views_f32(const float* src, int64_t rows, int C, const int32_t* idx, const double* noise,
                                    const double* cossin, int B, float* out, void* stream) {
    return launch_views<float>(src, rows, C, idx, noise, cossin, B, out, stream);
}

extern "C" int facl_build_views_f64(const double* src, int64_t rows, int C, const int32_t* idx, const double* noise,
                                    const double* cossin, int B, float* out, void* stream) {
    return launch_views<double>(src, rows, C, idx, noise, cossin, B, out, stream);
}
