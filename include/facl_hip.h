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
#define FACL_E_CONFIG (-4)   /* the fused variant does not cover this size: use the unfused entry points */

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
/* The same on the loader's clip-major batch: clips (B,G,N,D) contiguous; outputs are view-major, cloud m = g*B + b
 * (the permute(1,0,2,3).reshape(-1,N,D) copy of cn3d_train_motion_GL.py:226 is folded into the kernel's addressing). */
int facl_group_clips(const float* clips, int B, int G, int N, int D, int S, int K, float r2, int32_t* idx,
                     float* xt, float* yt, void* stream);

/* ---- train-mode BatchNorm plumbing (fp64) -------------------------------------------------
 * BN semantics: cn3d_model_conbag.py:46,50,54,64,68,72,84 (nn.BatchNorm2d/1d defaults).
 * `sums` is (C,2) double = per-channel (sum, sum of squares) over `count` positions; under DDP the
 * host all-reduces `sums` (and adds the counts) between the producing pass and facl_bn_finalize.
 * `bnc` is (5,C) float: mean, invstd, scale = gamma*invstd, shift = beta - mean*scale, sign(gamma).
 * running_mean / running_var (may both be NULL) are updated in place with `momentum` and the
 * unbiased variance, like F.batch_norm(training=True). */
/* size of the scratch buffer `ws` the reducing calls need.  Its LAST 4096 bytes are ticket counters of the single-launch
 * partial-row reduction: the caller zeroes them once after allocating the buffer; every call leaves them zero again. */
int64_t facl_ws_bytes(void);
/* aamax (or NULL): FACL_AMAX_WORDS uint32 that receive (in every slot) the bits of a rigorous BOUND of the layer's activation
 * max|gamma (y - mean) invstd + beta| over the batch: |gamma| sqrt(count - 1) sigma invstd + |beta| (Samuelson's inequality).
 * It is the fp16x3 operand scale of that activation (see "fp16x3 operand scales" below). */
/* zamax (or NULL): another FACL_AMAX_WORDS buffer that the call ZEROES (the next kernel raises it with atomics: spares a fill launch) */
int facl_bn_finalize(const double* sums, int C, double count, const float* gamma, const float* beta,
                     float eps, float momentum, float* running_mean, float* running_var, float* bnc,
                     uint32_t* aamax, uint32_t* zamax, void* stream);
int facl_bn_eval_consts(int C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* bnc, void* stream);

/* ---- set-abstraction point-MLP forward (net3DV_1, cn3d_model_conbag.py:43-58 / :162-177) ----
 * x is the grouped input, (P,D) rows = the memory behind the reference's (M,D,S,K) view, P = M*S*K,
 * K = 64 ("unit" = 64 consecutive rows = one group), D in {3,4}.  Weights are the reference's
 * tensors as stored in the checkpoint: W1 (64,D), W2 (64,64), W3 (256,64), row-major (Cout,Cin).
 *   facl_sa_x_moments           x -> mom = [sum_p x (D) | sum_p x x^T (D*D)]  (double)
 *   facl_bn1_sums_from_moments  mom -> sums of y1 = W1 x + b1 (exact: y1 is affine in x)
 *   facl_sa_l1tab               fold BN1 into layer 1: l1tab (64,8) = [scale*W1 | scale*b1+shift | 0]
 *   facl_sa_fwd2                x -> y2 = relu(bn1(y1)) W2^T + b2, stored in "fragment layout"
 *                               (nunits*4096 floats, see csrc/common.h); sums2 (64,2) or NULL
 *   facl_sa_fwd3                y2 -> per (group,channel) max_k sgn3*y3 and its argmax k (uint8),
 *                               y3 = relu(bn2(y2)) W3^T + b3; sums3 (256,2) = (sum, sumsq) of y3 or NULL
 *   facl_sa_pool                pooled = relu(|scale3| * ymax + shift3)   (rows,C)
 */
int facl_sa_x_moments(const float* x, int64_t P, int D, double* mom, void* ws, void* stream);
int facl_bn1_sums_from_moments(const double* mom, double count, int D, const float* W1, const float* b1,
                               double* sums, void* stream);
/* training, the 64-channel first layer: facl_bn1_sums_from_moments + facl_bn_finalize + facl_sa_l1tab in ONE launch (sums (64,2),
 * bnc (5,64), running statistics, the activation bound in aamax, l1tab (64,8)); every output bit-identical to the three calls */
int facl_sa_bn1_chain(const double* mom, double count, int D, const float* W1, const float* b1, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                      double* sums, float* bnc, uint32_t* aamax, float* l1tab, void* stream);
/* xamax -> a1amax (both or neither; FACL_AMAX_WORDS uint32 each): the bound of max|a1| from max|x| (eval-mode constants give
 * no bound of their own; in training facl_bn_finalize's aamax of BN1 serves) */
int facl_sa_l1tab(const float* W1, const float* b1, int D, const float* scale, const float* shift,
                  float* l1tab, const uint32_t* xamax, uint32_t* a1amax, void* stream);
/* a1amax: bits of a bound of max|a1| (FACL_AMAX_WORDS uint32): the fp16x3 scale of the layer-1 activation */
int facl_sa_fwd2(const float* x, int64_t nunits, int D, const float* l1tab, const float* W2,
                 const float* b2, float* y2f, double* sums2, void* ws, const uint32_t* a1amax, void* stream);
int facl_sa_fwd3(const float* y2f, int64_t nunits, const float* scale2, const float* shift2,
                 const float* W3, const float* b3, const float* sgn3, float* ymax, uint8_t* arg,
                 double* sums3, void* ws, void* stream);
/* (sgn3 / sgn arguments of facl_sa_fwd3 and facl_gemm_fwd_segmax: only the SIGN of each entry is used, sign(0) = +1, so the
 * BatchNorm weight itself can be passed.) */
/* fp16-input twin of facl_sa_fwd3 (dense configuration): a2 and W3 rounded to fp16, one v_mfma_f32_32x32x16_f16 product
 * per multiply-add, fp32 accumulation; same outputs. */
int facl_sa_fwd3_f16(const float* y2f, int64_t nunits, const float* scale2, const float* shift2, const float* W3,
                     const float* b3, const float* sgn3, float* ymax, uint8_t* arg, double* sums3, void* ws,
                     void* stream);
/* fp16x3 twin of facl_sa_fwd3 (csrc/common.h: a2 and W3 as two fp16 planes each of the operand times a power of two
 * taken from its own maximum / bound, three products per multiply-add, fp32 accumulation): fp32-GEMM accuracy at half the
 * MFMA work; same outputs.  The model's default.  a2amax: bits of a bound of max|relu(bn2(y2))| (FACL_AMAX_WORDS uint32). */
int facl_sa_fwd3_h3(const float* y2f, int64_t nunits, const float* scale2, const float* shift2, const float* W3,
                    const float* b3, const float* sgn3, float* ymax, uint8_t* arg, double* sums3, void* ws,
                    const uint32_t* a2amax, void* stream);
/* amax (or NULL): FACL_AMAX_WORDS uint32 (zeroed, or holding the maximum of a tensor that shares the scale) raised to the
 * bits of max(pooled) */
int facl_sa_pool(const float* ymax, int64_t rows, int C, const float* scale, const float* shift,
                 float* pooled, uint32_t* amax, void* stream);
/* net3DV_1 in EVAL mode as ONE kernel (csrc/sa_eval.hip; SURVEY 8 f-1: extract_motion_feature.py:143-221 runs the encoder under
 * eval(), every BatchNorm a constant affine): x (nunits*64, D) grouped rows -> pooled (nunits, 256); nothing else is stored (no y2
 * tile, no statistics, no argmax).  l1tab from facl_sa_l1tab with BN1's eval constants (and xamax -> a1amax); scale2 / shift2 (64),
 * scale3 / shift3 (256): rows 2 and 3 of facl_bn_eval_consts. */
int facl_sa_eval(const float* x, int64_t nunits, int D, const float* l1tab, const float* W2, const float* b2,
                 const float* scale2, const float* shift2, const float* W3, const float* b3, const float* scale3,
                 const float* shift3, float* pooled, const uint32_t* a1amax, void* stream);
/* ---- fp16x3 operand scales: measured maxima for operands no BatchNorm bounds ----
 *   facl_absmax          raises the slots of `amax` (FACL_AMAX_WORDS uint32, zeroed by the caller or pre-seeded) to max|x|
 *   facl_rows_act_amax   the same for max relu(scale[c] y[r][c] + shift[c]) of a (R,C) array (eval-mode layers: running
 *                        statistics say nothing about the data) */
int facl_absmax(const float* x, int64_t n, uint32_t* amax, void* stream);
int facl_rows_act_amax(const float* y, int64_t R, int C, const float* scale, const float* shift, uint32_t* amax, void* stream);

/* ---- set-abstraction point-MLP backward (autograd of net3DV_1) ---------------------------
 * See csrc/sa_bwd.hip for the algebra ("BN-affine folding": the dense part of BN3's backward is
 * affine in y3, so layer 3's backward is 64x64 work per position and y3 is never re-materialised).
 *   facl_sa_bwd0    dpooled (rows,256), ymax, bnc3 (5,256) -> coef = scale3*dz3 (rows,256),
 *                   sums (256,2) = (dbeta3, dgamma3)
 *   facl_sa_bwd1    y2f, bnc2 (5,64), G3 (64,64), h3 (64), W3, coef, arg -> dz2f (fragment layout),
 *                   sums (64,2) = (dbeta2, dgamma2)
 *   facl_sa_bwd_w3  y2f, bnc2, coef, arg -> out (20544 doubles) =
 *                   [ sum_g coef*a2[arg] (256,64) | sum_p a2^T a2 (64,64) | sum_p a2 (64) ]
 *   facl_sa_bwd2    dz2f, y2f, x, bw2 (4,64) = [scale2 | A | B | mean2] with dy2 = scale2*dz2 + A + B*(y2-mean2),
 *                   W2, l1tab -> out (4608 doubles) = [ dW2 (64,64) | R1 (8,64) ], R1 rows = sum_p x_d*dz1
 *                   (d < D), then sum_p dz1, then zeros.
 */
int facl_sa_bwd0(const float* dpooled, const float* ymax, int64_t rows, const float* bnc3, float* coef,
                 double* sums, void* ws, void* stream);
int facl_sa_bwd1(const float* y2f, int64_t nunits, const float* bnc2, const float* G3, const float* h3,
                 const float* W3, const float* coef, const uint8_t* arg, float* dz2f, double* sums,
                 void* ws, const uint32_t* a2amax, void* stream);
int facl_sa_bwd_w3(const float* y2f, int64_t nunits, const float* bnc2, const float* coef,
                   const uint8_t* arg, double* out, void* ws, const uint32_t* a2amax, void* stream);
/* a1amax / a2amax: the activation bounds the forward passes were given (facl_sa_fwd2 / facl_sa_fwd3_h3) */
int facl_sa_bwd2(const float* dz2f, const float* y2f, const float* x, int64_t nunits, int D,
                 const float* bw2, const float* W2, const float* l1tab, double* out, void* ws,
                 const uint32_t* a1amax, void* stream);
/* fp64 closed-form assembly between those passes (csrc/finalize.hip).  "_g" = after the SyncBN all-reduce,
 * "_l" = this rank's sums (parameter gradients stay local; the data-parallel wrapper averages them).
 *   facl_sa_bwd_consts3  sums0 (dbeta3,dgamma3) -> G3 (64,64), h3 (64) for facl_sa_bwd1
 *   facl_sa_bwd_consts2  sums1 (dbeta2,dgamma2) -> bw2 (4,64) for facl_sa_bwd2
 *   facl_sa_bwd_final    all partial sums -> dW3,dgamma3,dbeta3, dW2,dgamma2,dbeta2, dW1,dgamma1,dbeta1
 *                        (R1_g (8,64) = all-reduced tail of facl_sa_bwd2's output, mom_l = local x moments) */
/* BN backward constants of a row layer (tail): sums (C,2) = (dbeta, dgamma) of THIS rank / after the SyncBN
 * all-reduce -> dbeta (C), dgamma (C) fp32 parameter gradients (local) and kk (2,C) = reduced sums / P */
int facl_bn_bwd_consts(const double* sums_local, const double* sums_global, int C, double P, float* dbeta,
                       float* dgamma, float* kk, void* stream);
int facl_sa_bwd_consts3(const double* sums0, const float* bnc3, const float* W3, const float* b3, double P,
                        float* G3, float* h3, void* stream);
int facl_sa_bwd_consts2(const double* sums1, const float* bnc2, double P, float* bw2, void* stream);
int facl_sa_bwd_final(const double* out3, const double* sums0_g, const double* sums0_l, const float* bnc3,
                      const float* W3, const float* b3, const double* out2, const double* sums1_l,
                      const double* R1_g, const double* mom_l, const float* bnc1, const float* W1,
                      const float* b1, int D, double P, float* dW3, float* dg3, float* dbe3, float* dW2,
                      float* dg2, float* dbe2, float* dW1, float* dg1, float* dbe1, void* stream);

/* ---- encoder tail: row-major (R,C) BatchNorm / ReLU / max-over-S kernels ----------------------
 * net3DV_3 + my_max_pool + netR_FC (cn3d_model_conbag.py:61-88, :199-207).  The dense contractions
 * between them are the facl_gemm_* entries below (hand-written MFMA GEMMs); these kernels are everything else.
 * bnc = (5,C) constants of facl_bn_finalize / facl_bn_eval_consts; kk = (2,C) = (dbeta/P, dgamma/P).
 *   facl_rows_stats        y (R,C) -> sums (C,2)
 *   facl_rows_bn_relu      out = relu(scale*y + shift)                         (may run in place)
 *   facl_rows_segmax       x_pre[m,c] = max_s relu(bn(y[m,s,c])), arg = first maximising s
 *   facl_rows_bwd_stats    sums (C,2) = (sum dz, sum dz*yhat), dz = dout*[bn(y) > 0]
 *   facl_rows_bwd_apply    dy = scale*(dz - k1 - yhat*k2)
 *   facl_segmax_bwd_stats / _apply : the same two steps for the gradient arriving through the max over S
 */
int facl_rows_stats(const float* y, int64_t R, int C, double* sums, void* ws, void* stream);
int facl_rows_bn_relu(const float* y, int64_t R, int C, const float* scale, const float* shift, float* out,
                      void* stream);
int facl_rows_segmax(const float* y, int64_t M, int S, int C, const float* bnc, float* out, int32_t* arg,
                     void* stream);
int facl_rows_bwd_stats(const float* dout, const float* y, int64_t R, int C, const float* bnc, double* sums,
                        void* ws, void* stream);
/* gobaol_max_pool (cn3d_model_conbag.py:225-226) on the per-view maxima: x (G*B,C) view-major (row g*B+b) ->
 * out (B,C) = max over the G views, arg (B,C) = first view that attains it; backward routes dout to that view's row
 * of dx (G*B,C) and writes zeros elsewhere. */
int facl_viewmax_fwd(const float* x, int G, int B, int C, float* out, int32_t* arg, void* stream);
int facl_viewmax_bwd(const float* dout, const int32_t* arg, int G, int B, int C, float* dx, void* stream);
/* the same routing ADDED into dx (G*B,C), whose rows already hold the other gradient path of x_pre (each (clip, channel)
 * touches exactly one element: no atomics) */
int facl_viewmax_bwd_add(const float* dout, const int32_t* arg, int G, int B, int C, float* dx, void* stream);
/* gobaol_max_pool straight into netR_FC's stacked input (cn3d_model_conbag.py:225-229): h (G*B + B, C) = [x ; max over the G views
 * of x], arg (B,C) = the first view that attains the maximum -- facl_viewmax_fwd plus the copy of the rows it reads anyway. */
int facl_viewmax_stack(const float* x, int G, int B, int C, float* h, int32_t* arg, void* stream);

/* ---- netR_FC's BatchNorm1d + ReLU over the STACKED rows (cn3d_model_conbag.py:203-204 called at :228 on the G*B view rows and
 * at :229 on the B clip rows): y (R,C), segment a = rows [0,M), segment b = rows [M,R); two batch statistics, the running
 * buffers updated twice (a first), one set of kernels (csrc/fchead.hip).  M % 32 == 0 (else FACL_E_CONFIG: use the
 * single-segment entries).  Statistics are kept as 32-row SLICE sums that the consumer adds itself (no reduction launch):
 *   facl_fc_bn_stats      slice (sum, sumsq) of y into `ws`; with sums2 (2,C,2) also the segment totals (SyncBN all-reduce).
 *   facl_fc_bn_apply      bnc2 (2,5,C) = (mean, invstd, scale, shift, sign) per segment from `sums2` (totals) or from `part`
 *                         (nslices_a + nslices_b slice rows of (C,2) doubles: `ws` after facl_fc_bn_stats);
 *                         running statistics (momentum, unbiased variance) for a then b; a_out = relu(bn(y)).
 *   facl_fc_bn_bwd_stats  slice sums of dz = dact*[z>0] and dz*yhat into `ws` (+ totals into sums2).
 *   facl_fc_bn_bwd_apply  kk2 (2,2,C), dy = scale (dz - k1 - yhat k2) with k = (all-reduced sums2_g, or the local slice sums
 *                         in `ws`) / count; dgamma / dbeta (C) = this rank's sums of both segments (fp32 add of the two). */
int facl_fc_bn_stats(const float* y, int64_t M, int64_t R, int C, double* sums2, void* ws, void* stream);
int facl_fc_bn_apply(const float* y, int64_t M, int64_t R, int C, const double* sums2, const double* part, int nslices_a,
                     int nslices_b, double count_a, double count_b, const float* gamma, const float* beta, float eps,
                     float momentum, float* running_mean, float* running_var, float* bnc2, float* a_out, void* stream);
int facl_fc_bn_bwd_stats(const float* dact, const float* y, int64_t M, int64_t R, int C, const float* bnc2, double* sums2,
                         void* ws, void* stream);
int facl_fc_bn_bwd_apply(const float* dact, const float* y, int64_t M, int64_t R, int C, const float* bnc2,
                         const double* sums2_g, const void* ws, double count_a, double count_b, float* dgamma, float* dbeta,
                         float* kk2, float* dy, void* stream);
/* out (C) = column sums of x (R,C), fp64 accumulation: the bias gradient of netR_FC's last Linear (dout.sum(0)). */
int facl_col_sums(const float* x, int64_t R, int C, float* out, void* stream);
/* dWc (C,3) fp64 = dy^T centers: the centroid-xyz columns of the first per-centroid layer's weight gradient
 * (the input of net3DV_3 is torch.cat((yt, xt), 1), cn3d_model_conbag.py:219) in one streaming pass over dy */
int facl_rows_center_wgrad(const float* dy, const float* centers, int64_t R, int C, double* dWc, void* ws,
                           void* stream);
int facl_rows_bwd_apply(const float* dout, const float* y, int64_t R, int C, const float* bnc,
                        const float* kk, float* dy, void* stream);
int facl_segmax_bwd_stats(const float* dxpre, const float* xpre, const float* y, const int32_t* arg,
                          int64_t M, int S, int C, const float* bnc, double* sums, void* ws, void* stream);
/* the same sums from ymax (M,C) = max over the S rows of sign(gamma)*y, as facl_gemm_rs_fwd / facl_gemm_fwd_segmax return it: the
 * value at the argmax is sign(gamma)*ymax exactly (bnc row 4 holds the sign), so y is not gathered (one cache line per element) */
int facl_segmax_bwd_stats_ymax(const float* dxpre, const float* xpre, const float* ymax, int64_t M, int C,
                               const float* bnc, double* sums, void* ws, uint32_t* zamax, int zwords, void* stream);
/* (zamax, zwords: amax words -- or NULL, 0 -- that the call zeroes for the *_amax kernels behind it: spares a fill launch) */
int facl_segmax_bwd_apply(const float* dxpre, const float* xpre, const float* y, const int32_t* arg,
                          int64_t M, int S, int C, const float* bnc, const float* kk, float* dy, void* stream);
/* the two dy producers of the BatchNorm backward, also maintaining max|dy| in `amax`: a buffer of FACL_AMAX_WORDS uint32 the
 * caller zeroed, in which the kernels raise one of 64 slots (one 128-byte line each: atomics on a single address serialise)
 * to the bit pattern of the largest |dy| they wrote (non-negative floats order like unsigned integers; a NaN lands above
 * everything); max|dy| = the maximum over the buffer.  The fp16x3 GEMMs that consume dy take its power-of-two operand scale
 * from there, on the device, with no host round trip.  amax = null: the plain functions.  With amax the 4-channel kernels
 * are required (C % 4 == 0, 16-byte aligned tensors), else FACL_E_ALIGN. */
#define FACL_AMAX_WORDS 2048
int facl_rows_bwd_apply_amax(const float* dout, const float* y, int64_t R, int C, const float* bnc, const float* kk,
                             float* dy, uint32_t* amax, void* stream);
int facl_segmax_bwd_apply_amax(const float* dxpre, const float* xpre, const float* y, const int32_t* arg, int64_t M, int S,
                               int C, const float* bnc, const float* kk, float* dy, uint32_t* amax, void* stream);

/* ---- fp32 MFMA GEMMs of the tail's dense 1x1 channel contractions (csrc/gemm.hip) ------------------
 * Weights are the checkpoint tensors, row-major (Cout,Cin) with leading dimension ldw (multiple of 4).
 *   facl_gemm_fwd    y (M,N) = a' W^T + bias [+ centers (M,3) Wc (N,ldwc)^T], a' = a or relu(pscale*a+pshift);
 *                    sums (N,2) = column (sum, sumsq) of y for the BN that follows (or NULL)
 *   facl_gemm_dgrad  da (M,K) = dy (M,N) W (N,K)
 *   facl_gemm_wgrad  dW (N,K) = dy^T a, contraction over the M rows split into nz slices
 *                    (`slices` = scratch of nz*N*K floats), summed in slice order (deterministic) */
int facl_gemm_fwd(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                  const float* pscale, const float* pshift, const float* centers, const float* Wc,
                  int ldwc, float* y, double* sums, void* ws, void* stream);
/* facl_gemm_fwd with my_max_pool over blocks of S = 64 consecutive rows fused into its epilogue
 * (cn3d_model_conbag.py:71-73 + :80/:199): ymax (M/64,N) = max over the block of sgn[j]*y, arg (M/64,N) = first row of
 * the block that attains it.  FACL_E_CONFIG when the problem is too small for the 128x128-tile kernel (callers then use
 * facl_gemm_fwd + facl_rows_segmax). */
int facl_gemm_fwd_segmax(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                         const float* sgn, float* y, double* sums, float* ymax, int32_t* arg, void* ws,
                         void* stream);
int facl_gemm_dgrad(const float* dy, int64_t M, int N, const float* W, int ldw, int K, float* da,
                    void* stream);
int facl_gemm_wgrad(const float* dy, const float* a, int64_t M, int N, int K, int lda, float* dW,
                    float* slices, int nz, void* stream);
/* facl_gemm_wgrad_acc: dW (N,K) += dy^T a -- the weight-gradient GEMM accumulating INTO its output (the second gradient path of
 *   the loss's keys, d x += dsim^T @ [x ; x_global], without an add launch).  prec: 0 = fp32-grade (bf16x6), 1 = fp16 inputs,
 *   2 = bf16x3 (the twins below).  Few rows only (M <= 8192, one wave of 64x64 workgroups): FACL_E_CONFIG otherwise. */
int facl_gemm_wgrad_acc(const float* dy, const float* a, int64_t M, int N, int K, int lda, float* dW, int prec, void* stream);
/* fp16-input twins (dense configuration, BASELINE configs[4] "fp16 MFMA point-MLP"): same arguments and fp32 storage;
 * the operands are rounded to fp16 while their tiles are staged and every multiply-add is ONE
 * v_mfma_f32_32x32x16_f16 product with fp32 accumulation (instead of the six bf16 products of the fp32-grade path). */
int facl_gemm_fwd_f16(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                      const float* pscale, const float* pshift, const float* centers, const float* Wc, int ldwc,
                      float* y, double* sums, void* ws, void* stream);
int facl_gemm_fwd_segmax_f16(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                             const float* sgn, float* y, double* sums, float* ymax, int32_t* arg, void* ws,
                             void* stream);
int facl_gemm_dgrad_f16(const float* dy, int64_t M, int N, const float* W, int ldw, int K, float* da,
                        void* stream);
int facl_gemm_wgrad_f16(const float* dy, const float* a, int64_t M, int N, int K, int lda, float* dW,
                        float* slices, int nz, void* stream);
/* ---- row-streamed forward / dgrad for the 49,152-row layers (csrc/gemm_rs.hip) -------------------------------------------
 * Replaces the same torch chain as facl_gemm_fwd / facl_gemm_dgrad (Conv2d 1x1 of net3DV_3, cn3d_model_conbag.py:61-77,
 * and its autograd) with the weight operand pre-split ONCE per step into fragment-ordered bf16 planes
 * (facl_gemm_rs_planes; caller-allocated buffer of facl_gemm_rs_planes_bytes) and the activation rows streamed through
 * per-wave LDS slots.  Same arithmetic as facl_gemm_fwd (six bf16 products per multiply-add): bit-identical results.
 *   facl_gemm_rs_supported   1 when (rows M, contraction K, output columns N) is served: M >= 2048, 64 <= K <= 1024,
 *                            K % 32 == 0, N % 256 == 0; else 0 (callers then use facl_gemm_fwd / facl_gemm_dgrad)
 *   facl_gemm_rs_planes      transposed = 0: planes for y = a W^T (W (N,K), leading dimension ldw; Wc (N,3) optional:
 *                            the centroid-xyz columns of torch.cat((yt, xt), 1), :219); transposed = 1: planes for da = dy W
 *   facl_gemm_rs_fwd         y = f(a) W^T + bias [+ centers Wc^T], f = relu(pscale*a + pshift) when pscale is given (the
 *                            previous layer's BatchNorm2d + ReLU, :62-63: the activation is never materialised); sums (N,2)
 *                            as facl_gemm_fwd; sgn / ymax / arg (all or none) as facl_gemm_fwd_segmax (M % 64 == 0)
 *   `half` (planes and forward): 0 = three bf16 planes, six products per multiply-add (bf16x6); 1 = two fp16 planes of the
 *                            operands times a power of two, THREE products (fp16x3, csrc/common.h): the same fp32-GEMM
 *                            accuracy (22-bit operands, fp32 accumulation) at half the MFMA work.  No operand has a fixed
 *                            scale: the weights carry one power of two per 32-column tile (max|w| of the tile, taken by
 *                            facl_gemm_rs_planes and stored behind the planes), the row operand takes its scale from
 *                            `amax_a` (forward: FACL_AMAX_WORDS uint32 with the bits of a bound of max|f(a)| -- and of
 *                            max|centers| when centres are given -- from facl_bn_finalize / facl_sa_pool / facl_absmax /
 *                            facl_rows_act_amax); the planes must have been built with the same `half`
 *   facl_gemm_rs_dgrad       da (M,K) = dy (M,N) W.  half = 1: fp16x3 with dy's scale chosen per launch from `amax` (the
 *                            FACL_AMAX_WORDS buffer of facl_rows_bwd_apply_amax / facl_segmax_bwd_apply_amax): the
 *                            power of two that puts the maximum in [2^13, 2^14); elements down to 2^-16 of the maximum keep
 *                            22 bits, smaller ones lose at most 2^-38 of the maximum
 *   facl_gemm_rs_wgrad       amax non-null: the same arithmetic (dy by its scale, f(y) by the scale of `amax_b`, the bound the
 *                            forward GEMM that consumed f(y) was given)
 *   facl_gemm_wgrad_pro      dW (N,K) = dy^T relu(pscale*y + pshift): facl_gemm_wgrad whose `a` operand is recomputed from
 *                            the previous layer's raw output y (M,K) while it is staged (the companion of the forward
 *                            prologue: the activation tensor never exists); `_x3`: the opt-in three-product arithmetic.
 *                            FACL_E_CONFIG when the shape is not served by the 128x128-tile kernel (callers materialise a). */
int facl_gemm_wgrad_pro(const float* dy, const float* y, int64_t M, int N, int K, int ldy, const float* pscale,
                        const float* pshift, float* dW, float* slices, int nz, void* stream);
int facl_gemm_wgrad_pro_x3(const float* dy, const float* y, int64_t M, int N, int K, int ldy, const float* pscale,
                           const float* pshift, float* dW, float* slices, int nz, void* stream);
/* the same weight gradient in fp16x3 (see `half` above): dy by the scale read from `amax` (the FACL_AMAX_WORDS buffer of
 * facl_rows_bwd_apply_amax), the activation operand by the one read from `amax_b` (its bound, same format); pscale / pshift
 * null: `y` IS the activation.  FACL_E_CONFIG when the shape is not served by the 128x128-tile kernel (callers use the
 * bf16x6 entries). */
int facl_gemm_wgrad_h3(const float* dy, const float* y, int64_t M, int N, int K, int ldy, const float* pscale,
                       const float* pshift, const uint32_t* amax, const uint32_t* amax_b, float* dW, float* slices, int nz,
                       void* stream);
/* Weight gradient of the widest layer on the register-streamed kernel (csrc/gemm_rs.hip: k_wgrad_rs): dW (N,K) =
 * dy^T f(y), f = relu(pscale*y + pshift) when pscale is given (else identity).  facl_gemm_rs_wgrad_slices returns the number
 * of row slices the call will use (scratch = that many x N x K floats), or 0 when the shape is not served (N % 512, K % 128,
 * M >= 4096 and at least 8 output blocks): facl_gemm_rs_wgrad then returns FACL_E_CONFIG and callers use facl_gemm_wgrad[_pro]. */
int facl_gemm_rs_wgrad_slices(int64_t M, int N, int K);
int facl_gemm_rs_wgrad(const float* dy, const float* y, int64_t M, int N, int K, const float* pscale, const float* pshift,
                       const uint32_t* amax, const uint32_t* amax_b, float* dW, float* slices, void* stream);
int64_t facl_gemm_rs_planes_bytes(int N, int K, int with_centers);
int facl_gemm_rs_planes(const float* W, int ldw, int N, int K, int transposed, const float* Wc, int ldwc, int half,
                        void* planes, void* stream);
/* n <= 8 matrices in one launch: arrays (length n) of the per-matrix arguments of facl_gemm_rs_planes */
/* absmax_x / absmax_n / absmax_amax (all or none): the launch also raises `absmax_amax` (FACL_AMAX_WORDS uint32) to
 * max|absmax_x[0 .. absmax_n)| -- the centroid coordinates share the fp16x3 scale of the first layer's row operand */
int facl_gemm_rs_planes_multi(int n, const float* const* W, const int* ldw, const int* N, const int* K,
                              const int* transposed, const float* const* Wc, const int* ldwc, const int* half,
                              void* const* planes, const float* absmax_x, int64_t absmax_n, uint32_t* absmax_amax,
                              void* stream);
int facl_gemm_rs_supported(int64_t M, int K, int N);
int facl_gemm_rs_fwd(const float* a, int64_t M, int K, const void* planes, int half, const uint32_t* amax_a, int N,
                     const float* bias, const float* pscale, const float* pshift, const float* centers, float* y, double* sums,
                     const float* sgn, float* ymax, int32_t* arg, void* ws, void* stream);
int facl_gemm_rs_dgrad(const float* dy, int64_t M, int N, const void* planes, int half, const uint32_t* amax, int K,
                       float* da, void* stream);
/* facl_gemm_rs_dgrad + facl_rows_bwd_stats(da, y, bnc) in one pass: sums (K,2) = the BatchNorm-backward column sums of the
 * layer whose activation relu(bn(y)) was this GEMM's forward input, taken from the accumulator tile before it is stored */
int facl_gemm_rs_dgrad_bnstats(const float* dy, int64_t M, int N, const void* planes, int half, const uint32_t* amax, int K,
                               float* da, const float* y, const float* bnc, double* sums, void* ws, void* stream);
/* "bf16x3" twins (opt-in precision "x3"; never the default): each operand keeps its two leading bf16 pieces and a
 * multiply-add is three products (hi*mid, mid*hi, hi*hi) instead of six -- relative error of a product <= 3 * 2^-16
 * (results ~1e-5 of an fp64 GEMM; the north_star's tolerance for features / loss is 1e-4), half the MFMA work. */
int facl_gemm_fwd_x3(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                     const float* pscale, const float* pshift, const float* centers, const float* Wc, int ldwc,
                     float* y, double* sums, void* ws, void* stream);
int facl_gemm_fwd_segmax_x3(const float* a, int64_t M, int K, const float* W, int ldw, int N, const float* bias,
                            const float* sgn, float* y, double* sums, float* ymax, int32_t* arg, void* ws,
                            void* stream);
int facl_gemm_dgrad_x3(const float* dy, int64_t M, int N, const float* W, int ldw, int K, float* da,
                       void* stream);
int facl_gemm_wgrad_x3(const float* dy, const float* a, int64_t M, int N, int K, int lda, float* dW,
                       float* slices, int nz, void* stream);
int facl_sa_fwd3_x3(const float* y2f, int64_t nunits, const float* scale2, const float* shift2, const float* W3,
                    const float* b3, const float* sgn3, float* ymax, uint8_t* arg, double* sums3, void* ws,
                    void* stream);

/* ---- second-level grouping on channel-first features (utils_my.py:332-381 group_points_2 / group_points_2_3DV) ----
 * The kNN + radius rule of a second level runs on the level-1 centroid coordinates through facl_group (idx + centred xyz).
 * facl_gather_rows moves the features: out[r][col_off + c] = feat[m][idx[r]][c] for the rows r = (m, s2, k) of cloud m
 * (feat (M,S1,C) rows of stride ldf; out rows of stride ldo); with xyz (rows,3) also out[r][0..2] = xyz[r] (col_off = 3:
 * the reference's concatenated (3+C)-channel rows).  facl_scatter_rows is its transpose (gradient of the gather):
 * dfeat[m][s][c] = sum_{r: idx[r] = s} drows[r][col_off + c], rows added in index order (deterministic, no atomics). */
int facl_gather_rows(const float* feat, int ldf, int M, int S1, int C, const int32_t* idx, int rows_per_cloud,
                     const float* xyz, float* out, int ldo, int col_off, void* stream);
int facl_scatter_rows(const float* drows, int ldd, int col_off, int M, int S1, int C, const int32_t* idx,
                      int rows_per_cloud, float* dfeat, void* stream);

/* ---- contrastive losses on a similarity matrix (utils_my.py:53-116) ----------------------------
 * sim (R,J) = anchors @ keys^T, R = nA*B rows (row i*B+n: clip n), J = G*Bk columns (column j belongs to
 * clip j % Bk).  Same-clip columns count as exp(0) (the reference multiplies them by 0); the nA rows of a
 * clip form ONE shared negative set; nS positive slots per clip, slot s reads its positive from row
 * (slot_rows ? s : 0)*B+n, column poscol[s*B+n]; CE = mean over B, summed over the slots.
 *   circle loss: nA = nS = G-1, slot_rows = 1;   global loss: nA = 1, nS = G, slot_rows = 0.
 * clip_offset = rank*B under data parallelism.  Outputs: loss (1 double), dsim (R,J) = d loss / d sim. */
int facl_contrast(const float* sim, int R, int J, int B, int Bk, int nA, int nS, int slot_rows,
                  const int32_t* poscol, int clip_offset, float* dsim, double* loss, void* ws, void* stream);
/* Both losses of a training step (cn3d_train_motion_GL.py:265-316) in one launch on ONE similarity matrix
 * sim ((G+1)*B, J) = [x ; x_global] @ keys^T (row block g < G: view g of the local clips, block G: x_global; J = G*Bk).
 * order (G) int64 = the circle loss's view permutation (utils_my.py:96-97).  losses = [loss_c, loss_circle];
 * dsim ((G+1)*B, J) = d(loss_c + loss_circle)/d sim, every row written (the block of view order[G-1], which is no
 * anchor, is zero). */
int facl_contrast_pair(const float* sim, int G, int B, int Bk, int J, const int64_t* order, int clip_offset,
                       float* dsim, double* losses, void* ws, void* stream);
/* facl_contrast_pair that also writes the fp32 values of the loop body: losses32 = [loss_c, loss_circle, loss_circle + loss_c]
 * (the fp32 sum of the two fp32 losses: `loss = loss_circle + loss_c`, cn3d_train_motion_GL.py:329), one finishing launch
 * instead of a reduction, a cast and an add. */
int facl_contrast_pair_sum(const float* sim, int G, int B, int Bk, int J, const int64_t* order, int clip_offset,
                           float* dsim, double* losses, float* losses32, void* ws, void* stream);
/* dst (R,J) = src scaled by *g1 in rows [0,R1) and by *g2 in rows [R1,R): the two upstream gradients (device scalars)
 * of loss_circle / loss_c applied to the shared d/dsim matrix. */
int facl_scale_rows2(const float* src, float* dst, int64_t R1, int64_t R, int J, const float* g1, const float* g2,
                     void* stream);

/* ---- F.normalize + mapping (cn3d_model_conbag.py:231-232) --------------------------------------
 * x (M,C) -> x_nor = x / max(||x||_2, 1e-12) row-wise, code (M,K) = x_nor @ Wm^T (Wm (K,C), no bias).
 * C % 4 == 0, C <= 4096, K <= 256. */
int facl_normalize_map(const float* x, int64_t M, int C, const float* Wm, int K, float* x_nor, float* code,
                       void* stream);

/* ---- OPT-IN one-shot all-reduce of a small fp64 buffer through peer-mapped mailboxes (csrc/mailbox.hip; FACL_ONESHOT_SYNCBN=1):
 * the 14 SyncBN reductions of the data-parallel step (0.1-16 KB each: the statistics the reference's single-process BatchNorm
 * layers take over the whole batch, cn3d_model_conbag.py:46-84) as ONE capturable kernel launch per rank and call instead of a
 * latency-bound RCCL collective.  Every rank allocates a mailbox (facl_mailbox_alloc: device memory + its 64-byte IPC handle),
 * opens every peer's (facl_mailbox_open), and calls facl_mailbox_allreduce in the same order: out = sum over the R ranks of
 * their `in`, added in rank order (bit-identical on every rank).  boxes: DEVICE array of R mailbox pointers, own at [rank]; seq: a
 * device uint64 that starts at 0 (the kernel advances it: a replayed graph posts fresh sequence numbers); err: device word
 * raised -- and the output poisoned with NaN -- when a peer does not post within 2 s (the wait never hangs the GPU).
 * Rehearsed with 2 / 4 processes on one GPU; cross-device visibility over xGMI is untested: never the default. */
int64_t facl_mailbox_bytes(int R, int n_max);
int facl_mailbox_alloc(int64_t bytes, void** ptr, void* handle64);
int facl_mailbox_open(const void* handle64, void** ptr);
int facl_mailbox_close(void* ptr);
int facl_mailbox_free(void* ptr);
int facl_mailbox_allreduce(const double* in, double* out, int n, int n_max, void* const* boxes, int rank, int R, uint64_t* seq,
                           uint32_t* err, void* stream);

/* ---- optimizer step (cn3d_train_motion_GL.py:180,:332: Adam, betas (0.5, 0.999), eps 1e-6, no weight decay) ------------
 * facl_adam_prep: step[0] += 1 (device float), consts = (lr[0] / (1 - b1^t), 1 / sqrt(1 - b2^t)) -- lr and the step
 * counter live on the device so that a captured HIP graph advances them by itself.
 * facl_adam_apply: ALL parameter tensors in one launch.  p / g / m / v are HOST arrays of nt <= 64 device pointers
 * (parameter, gradient, exp_avg, exp_avg_sq), n their element counts; the pointers travel by value in the kernel
 * arguments. */
int facl_adam_prep(const float* lr, float* step, float b1, float b2, float* consts, void* stream);
int facl_adam_apply(int nt, float* const* p, const float* const* g, float* const* m, float* const* v, const int* n,
                    const float* consts, float b1, float b2, float eps, void* stream);

/* ---- optional loss terms of the training loop (switched off in the shipped loop) ----------------------
 * facl_sinkhorn: distributed_sinkhorn + shoot_infs (cn3d_model_conbag.py:391-425).  Q (R,C) = exp(scores)^T, R prototypes
 * x C samples; `iters` row/column scalings; out (C,R).  scratch: R*C + R floats.
 * facl_kmeans: KMeans (cn3d_train_motion_GL.py:54-70): `iters` Lloyd iterations from the first K rows of x (N,D);
 * labels (N) = final assignment (first minimum on ties), cent (K,D) = final means, counts (K) (an empty cluster: 1). */
int facl_sinkhorn(const float* Q, int R, int C, int iters, float* scratch, float* out, void* stream);
int facl_kmeans(const float* x, int N, int D, int K, int iters, int32_t* labels, float* cent, int32_t* counts,
                void* stream);

/* ---- view construction of a batch of clips (replaces the NumPy pipeline of
 * training_code/cn3D_data_set.py:285-350 get_data_train + :654-663 + :708-713 + :734-749 + :767-778 and the
 * float64->float32 / permute head of cn3d_train_motion_GL.py:225-228) ----------------------------
 * src (rows,C>=8): the batch's source clouds packed row-wise (xyz, channels 3..7); idx (B,10,512): row of src for
 * every output point (views: raw, reversed, key, key-reversed, rotated x2, temporal ch4, temporal ch7, low-res x2);
 * noise (B,7,512,3): the standard-normal jitter draws in the reference's order; cossin (B,2,2): cos, sin of the two
 * rotation angles.  The host draws all of them (NumPy order) -- facl_amd/views.py.
 * out (10*B,512,4) float32, view-major (row g*B+b).  _f32/_f64 = dtype of the source arrays. */
int facl_build_views_f32(const float* src, int64_t rows, int C, const int32_t* idx, const double* noise,
                         const double* cossin, int B, float* out, void* stream);
int facl_build_views_f64(const double* src, int64_t rows, int C, const int32_t* idx, const double* noise,
                         const double* cossin, int B, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FACL_HIP_H */
