/*
 * facl_hip.h -- C ABI of libfacl_hip.so, the MI355X (gfx950) implementation of FACL's
 * contrastive-step hot path.
 *
 * The reference (tangent-T/FACL) is 100 % Python on stock PyTorch ops and has no FFI of its own;
 * each entry point below replaces a composition of torch ops at the cited reference file:line
 * (paths relative to the reference's training_code/).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add to call them.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory owned by the caller (PyTorch's allocator in
 *     our host code); the library never allocates, frees or copies persistent memory;
 *   - every launch goes to `stream` (a hipStream_t passed as void*), nothing synchronises;
 *   - return value: 0 on success, a positive hipError_t if a launch failed, a negative
 *     FACL_E_* code if the arguments are unsupported (nothing is launched in that case);
 *   - re-entrant, no global state, no host threads.
 *   - fp32 tensors unless said otherwise; "rows" of activations are positions (cloud-major,
 *     then centroid, then neighbour), channels are the fastest axis.
 */
#ifndef FACL_HIP_H
#define FACL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FACL_E_SHAPE  (-1)   /* unsupported shape (see each function)            */
#define FACL_E_NULL   (-2)   /* a required pointer is NULL                       */
#define FACL_E_ALIGN  (-3)   /* a pointer is not aligned as the kernel requires  */

/* Library / ABI version: (major << 16) | minor. */
int facl_version(void);

/* ---- farthest point sampling ------------------------------------------------------------
 * cn3d_data_load.py:301-320 (= cn3D_data_set.py:675-694): iterative FPS with the start index
 * given explicitly (the reference draws it with np.random.randint).  argmax takes the lowest
 * index among equal maxima (np.argmax).  dist^2 = (dx*dx+dy*dy)+dz*dz in the input dtype.
 *   xyz       (M, N, ld) rows; the first 3 columns of each row are x,y,z   [f32 | f64]
 *   start     (M) int32, each in [0,N)
 *   out_idx   (M, m) int32
 * Supported: 1 <= N <= 4096, 1 <= m, ld >= 3. */
int facl_fps_f32(const float* xyz, int M, int N, int ld, int m, const int32_t* start,
                 int32_t* out_idx, void* stream);
int facl_fps_f64(const double* xyz, int M, int N, int ld, int m, const int32_t* start,
                 int32_t* out_idx, void* stream);

/* cn3D_data_set.py:665-672: reorder each cloud so that rows picks[0..m) come first and the
 * remaining rows follow in ascending order (np.setdiff1d), truncated to N rows.
 *   points (M,N,D) f32 -> out (M,N,D) f32 (must not alias);  picks (M,m) int32. */
int facl_fps_reorder(const float* points, int M, int N, int D, const int32_t* picks, int m,
                     float* out, void* stream);

/* ---- kNN-then-radius grouping -----------------------------------------------------------
 * utils_my.py:255-291 (group_points_3DV) / :7-42 (group_points_3DV_2048): centroids are rows
 * 0..S-1; fp32 dist^2 = (dx*dx+dy*dy)+dz*dz; the K nearest are kept; a kept neighbour with
 * dist^2 > r2 (strict) is replaced by the centroid's own row; all D channels are gathered and
 * xyz is centred on the centroid.
 *   points (M,N,D) f32, D in {3,4}
 *   idx    (M,S,K) int32, ascending along K before the radius replacement   (may be NULL)
 *   xt     (M,S,K,D) f32 -- the memory behind the reference's (M,D,S,K) view (may be NULL)
 *   yt     (M,S,3)   f32 -- the memory behind the reference's (M,3,S,1) view (may be NULL)
 * Supported: S <= N <= 4096, 1 <= K <= N. */
int facl_group(const float* points, int M, int N, int D, int S, int K, float r2,
               int32_t* idx, float* xt, float* yt, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FACL_HIP_H */
