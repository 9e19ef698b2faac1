/*
 * e2emv.h - C ABI of libe2emv.so: the MI355X (gfx950) implementation of the
 * matcher -> Sinkhorn -> weighted-8-point hot path of barbararoessle/e2e_multi_view_matching.
 *
 * The reference has NO native/FFI seam for this path: the seam is two Python callables,
 *   MultiViewMatcher.forward(data) -> dict      (absent submodule; call sites
 *       helpers.py:246, eval_pairs.py:212, eval_multi_view.py:160)
 *   estimate_relative_pose_w8pt(...) -> (T, info)
 *       (pose_optimization/two_view/estimate_relative_pose.py:84-128)
 * The Python package e2e_multi_view_matching_amd re-creates those callables on top of the
 * entry points below through ctypes (INTEGRATION.md shows the binding).  Every function is
 * extern "C", takes plain pointers and sizes, never throws, and returns 0 or a negative
 * e2emv_status; the message for the last failure of a context is e2emv_last_error().
 *
 * Pointers named d_* are DEVICE pointers (hipMalloc / torch.cuda / e2emv_malloc memory on
 * the context's device).  Everything else is host memory.  `stream` is a hipStream_t passed
 * as void* (NULL = the default stream); all compute calls are asynchronous w.r.t. it.
 * The caller owns every input/output buffer; the library owns only its context (weights,
 * workspace arena grown lazily, never shrunk).  One context per (process, device); calls on
 * one context must be serialised by the caller; different contexts are independent.
 */
#ifndef E2EMV_H
#define E2EMV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the functions declared between this push and its pop are its whole
 * dynamic symbol table (tests/test_host_and_abi.py compares `nm -D` with this header). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define E2EMV_ABI_VERSION 1
#define E2EMV_MAX_TUPLE 8
#define E2EMV_MAX_LAYERS 64
#define E2EMV_MAX_KENC 8

typedef struct e2emv_ctx e2emv_ctx;

typedef enum {
    E2EMV_OK = 0,
    E2EMV_EINVAL = -1, /* bad argument / NULL pointer                                     */
    E2EMV_ENOMEM = -2, /* device or host allocation failed                                */
    E2EMV_EHIP = -3,   /* HIP runtime error (no device, launch failure, ...)              */
    E2EMV_ESHAPE = -4, /* unsupported or inconsistent shape                               */
    E2EMV_ESTATE = -5  /* weights missing / not committed                                 */
} e2emv_status;

/* ---- lifecycle -------------------------------------------------------------------- */
int e2emv_version(void);
/* Fails with E2EMV_EHIP when no gfx950 device is usable: there is NO CPU fallback. */
int e2emv_create(e2emv_ctx** out, int device);
void e2emv_destroy(e2emv_ctx* ctx);
const char* e2emv_last_error(const e2emv_ctx* ctx);

/* ---- memory helpers (so a harness without torch can drive the library) ------------ */
int e2emv_malloc(e2emv_ctx* ctx, void** d_ptr, size_t bytes);
int e2emv_free(e2emv_ctx* ctx, void* d_ptr);
int e2emv_h2d(e2emv_ctx* ctx, void* d_dst, const void* src, size_t bytes, void* stream);
int e2emv_d2h(e2emv_ctx* ctx, void* dst, const void* d_src, size_t bytes, void* stream);
int e2emv_sync(e2emv_ctx* ctx, void* stream);

/* ---- weights ----------------------------------------------------------------------
 * Replaces MultiViewMatcher.load_state_dict (helpers.py:47-52).  Keys are the upstream
 * SuperGlue parameter names the reference's checkpoints use:
 *   kenc.encoder.{0,3,..}.{weight,bias}, kenc.encoder.{1,4,..}.{weight,bias,running_mean,
 *   running_var}, gnn.layers.{i}.attn.proj.{0,1,2}.{weight,bias}, gnn.layers.{i}.attn.merge.*,
 *   gnn.layers.{i}.mlp.{0,3}.*, gnn.layers.{i}.mlp.1.* (BatchNorm), final_proj.*, bin_score,
 *   conf_mlp.{0,3}.*, conf_mlp.1.*   (a leading "module." is stripped).
 * Tensors are host fp32, dense row-major, Conv1d weights [out,in,1] or [out,in].
 * e2emv_commit_weights folds eval-mode BatchNorm into the preceding conv, re-orders the
 * attention channels from upstream's c = dd*H + h to head-major, and uploads.             */
typedef struct {
    int32_t desc_dim;                        /* D, 256                                    */
    int32_t num_heads;                       /* H, 4 (head dim D/H must be 64)            */
    int32_t n_kenc;                          /* hidden keypoint-encoder layers            */
    int32_t kenc[E2EMV_MAX_KENC];            /* [32,64,128,256]                           */
    int32_t n_layers;                        /* L = len(GNN_layers)                       */
    int32_t layer_types[E2EMV_MAX_LAYERS];   /* 0 = 'self', 1 = 'cross'                   */
    int32_t conf_mlp;                        /* 1: conf_mlp.* weights present             */
} e2emv_model_desc;

int e2emv_set_weight(e2emv_ctx* ctx, const char* key, const float* data, const int64_t* shape, int ndim);
int e2emv_commit_weights(e2emv_ctx* ctx, const e2emv_model_desc* model);

/* ---- matcher forward --------------------------------------------------------------
 * Replaces MultiViewMatcher.forward(data) (call sites above).  B tuples of T images with
 * N keypoints each; P = T(T-1)/2 pairs ordered (0,1),(0,2),(1,2),(0,3)... i.e. for j in
 * range(T): for i in range(j) - the order helpers.py:250-251 iterates.                    */
#define E2EMV_FLAG_FULL_OUTPUT 1 /* also produce matches / scores / confidences           */
#define E2EMV_FLAG_MULTI_FRAME 2 /* joint GNN over all T images (cross = all other images) */
#define E2EMV_DESC_F32 0
#define E2EMV_DESC_F16 1

typedef struct {
    int32_t batch;          /* B                                                          */
    int32_t tuple_size;     /* T <= E2EMV_MAX_TUPLE                                       */
    int32_t n_kpts;         /* N (same for every image of the call)                       */
    int32_t sinkhorn_iters; /* 100                                                        */
    float match_threshold;  /* 0.2                                                        */
    int32_t desc_dtype;     /* E2EMV_DESC_F32 | E2EMV_DESC_F16                            */
    int32_t flags;
    float img_w[E2EMV_MAX_TUPLE], img_h[E2EMV_MAX_TUPLE]; /* data['image{m}'].shape[-1/-2] */
    int32_t n_kpts_img[E2EMV_MAX_TUPLE]; /* per-image keypoint count N_m (0 = n_kpts): eval_pairs.py feeds images with
                                            different numbers of SuperPoint keypoints; n_kpts must be the maximum   */
} e2emv_forward_desc;

/* d_kpts[m]   [B,N_m,2] f32 pixel xy      (data['keypoints{m}'])
 * d_kscores[m][B,N_m]   f32               (data['scores{m}'])
 * d_desc[m]   [B,D,N_m] f32|f16, N_m contiguous (data['descriptors{m}'])
 * outputs, one pointer per pair p = (i, j) (any may be NULL = not wanted):
 * d_logZ[p]     [B,N_i+1,N_j+1] f32  result['scores_{i}_{j}']
 * d_matches0[p] [B,N_i] int64    result['matches{i}_{i}_{j}'] (-1 = unmatched; int64 because
 *                                the reference uses it as a fancy index, estimate_relative_pose.py:26-30)
 * d_matches1[p] [B,N_j] int64    result['matches{j}_{i}_{j}']
 * d_mscores0/1[p] [B,N_i]/[B,N_j] f32  matching scores
 * d_conf[p]     [B,N_i] f32      result['conf_scores_{i}_{j}'] (caller views it [B,N_i,1])  */
int e2emv_matcher_forward(e2emv_ctx* ctx, const e2emv_forward_desc* fd,
                          const float* const* d_kpts, const float* const* d_kscores, const void* const* d_desc,
                          float* const* d_logZ, int64_t* const* d_matches0, int64_t* const* d_matches1,
                          float* const* d_mscores0, float* const* d_mscores1, float* const* d_conf,
                          void* stream);

/* ---- Sinkhorn / matching as stand-alone operators ---------------------------------
 * log_optimal_transport(scores, bin_score, iters) of upstream superglue.py (used inside the
 * forward above): d_scores [B,M,N] f32 -> d_logZ [B,M+1,N+1] f32.                         */
int e2emv_sinkhorn(e2emv_ctx* ctx, int B, int M, int N, const float* d_scores, float bin_score, int iters,
                   float* d_logZ, void* stream);
/* mutual arg-max block of SuperGlue.forward on d_logZ [B,M+1,N+1]. */
int e2emv_extract_matches(e2emv_ctx* ctx, int B, int M, int N, const float* d_logZ, float match_threshold,
                          int64_t* d_matches0, int64_t* d_matches1, float* d_mscores0, float* d_mscores1,
                          void* stream);

/* ---- two-view pose ----------------------------------------------------------------
 * get_kpts (estimate_relative_pose.py:16-31): d_kpts1_g[b,n] = d_kpts1[b, matches[b,n]]
 * (index -1 wraps to the last keypoint), d_conf_out = (matches >= 0) * d_conf.             */
int e2emv_gather_matched(e2emv_ctx* ctx, int B, int N0, int N1, const float* d_kpts1, const int64_t* d_matches,
                         const float* d_conf, float* d_kpts1_g, float* d_conf_out, void* stream);

/* estimate_relative_pose_w8pt (estimate_relative_pose.py:84-128).
 * d_kpts0/1 [B,N,2]; d_intr0/1 [B,kdim,kdim] (kdim 3 or 4, or batch-broadcast when
 * intr_batch == 1); d_conf [B,N]; d_T_gt [B,4,4] needed iff choose_closest.
 * Outputs: d_T [B,4,4]; d_kpts0n/d_kpts1n [B,N,2]; d_conf_n [B,N] (normalised weights);
 * d_inliers [B,N] u8 (written iff determine_inliers); d_posdepth [B,N] u8; d_F [B,3,3]
 * (the estimated essential matrix, may be NULL); d_status [B] int32 (bit0: sum(conf)<=1e-6,
 * bit1: non-finite result, bit2: essential matrix of rank < 2 - no usable correspondence; the pose is then finite but
 * arbitrary, like a library SVD's null-space choice in the reference), may be NULL.  Returns E2EMV_ESHAPE when N < 8 (the Python shim
 * maps that to the reference's (None, None), estimate_relative_pose.py:85-86).             */
int e2emv_w8pt(e2emv_ctx* ctx, int B, int N, const float* d_kpts0, const float* d_kpts1, const float* d_intr0,
               const float* d_intr1, int kdim, int intr_batch, const float* d_conf, int choose_closest,
               const float* d_T_gt, int determine_inliers, float* d_T, float* d_kpts0n, float* d_kpts1n,
               float* d_conf_n, uint8_t* d_inliers, uint8_t* d_posdepth, float* d_F, int32_t* d_status,
               void* stream);

/* e2emv_w8pt on a RAGGED batch: sample b uses its first d_n_per[b] (8 <= n <= N) correspondences; every per-correspondence
 * buffer keeps the row stride N and the rows beyond n read 0 / false in the outputs.  The Hartley statistics of sample b
 * run over its own n rows only (a zero-weight padding row would change them - estimate_relative_pose.py:56-57), which is
 * what lets the pairs of a tuple with different numbers of matches share one launch (bundle_adjust_io.py:98-131). */
int e2emv_w8pt_ragged(e2emv_ctx* ctx, int B, int N, const int32_t* d_n_per, const float* d_kpts0, const float* d_kpts1,
                      const float* d_intr0, const float* d_intr1, int kdim, int intr_batch, const float* d_conf,
                      int choose_closest, const float* d_T_gt, int determine_inliers, float* d_T, float* d_kpts0n,
                      float* d_kpts1n, float* d_conf_n, uint8_t* d_inliers, uint8_t* d_posdepth, float* d_F, int32_t* d_status,
                      void* stream);

/* All T(T-1)/2 pairs of B tuples in ONE solve (the loop of helpers.py:250-258 / bundle_adjust_io.py:62-100 over the pairs
 * of a tuple, batched): pair q enumerates (i, j), i < j, j outer (q = 0: (0,1), 1: (0,2), 2: (1,2), ...).  get_kpts
 * (estimate_relative_pose.py:16-31) of every pair is fused in: d_kpts[t] [B,N,2] and d_intr[t] [intr_batch,kdim,kdim] per
 * image, d_matches[q] [B,N] int64 / d_conf[q] [B,N] per pair (matches of image i in image j, -1 wraps to the last keypoint
 * with weight 0), d_T_gt[q] [B,4,4] per pair when choose_closest.  All images carry the same N (the fixed-shape batches of
 * helpers.py:89-92).  Outputs are pair-major: element (q, b) at index q*B + b of d_T [P*B,4,4], d_kpts0n / d_kpts1n
 * [P*B,N,2], d_conf_n [P*B,N], d_inliers / d_posdepth [P*B,N], d_F [P*B,3,3], d_status [P*B]; semantics per element as
 * e2emv_w8pt.  3 kernel launches + 1 gather instead of 4 per pair.                                                    */
int e2emv_w8pt_tuple(e2emv_ctx* ctx, int B, int T, int N, const float* const* d_kpts, const float* const* d_intr, int kdim,
                     int intr_batch, const int64_t* const* d_matches, const float* const* d_conf, int choose_closest,
                     const float* const* d_T_gt, int determine_inliers, float* d_T, float* d_kpts0n, float* d_kpts1n,
                     float* d_conf_n, uint8_t* d_inliers, uint8_t* d_posdepth, float* d_F, int32_t* d_status, void* stream);

/* d_out[i] = d_mask[i] ? d_x[i] : 0 - the confidence masking between w8pt and the two-view BA
 * (eval_pairs.py:250-251, bundle_adjust_io.py:18-19: confidence[~pos_depth_mask] = 0). */
int e2emv_apply_mask(e2emv_ctx* ctx, int64_t n, const float* d_x, const uint8_t* d_mask, float* d_out, void* stream);

/* normalize (estimate_relative_pose.py:9-14): d_out[b,n] = ((x - cx) / fx, (y - cy) / fy), fp32. */
int e2emv_normalize_kpts(e2emv_ctx* ctx, int B, int N, const float* d_kpts, const float* d_intr, int kdim, int intr_batch,
                         float* d_out, void* stream);

/* T_a_to_b = inv(pose_b) @ pose_a per batch element (helpers.py:219, 254: the relative pose the GT-match builder and
 * the pose losses use), 4x4 row-major, solved in fp64. */
int e2emv_relative_pose(e2emv_ctx* ctx, int B, const float* d_pose_a, const float* d_pose_b, float* d_T_a2b, void* stream);

/* e2emv_pose_errors plus the reference's reductions (compute_pose_error.py:12, 22): d_transl_valid[b] = |t0||t1| > 1e-6,
 * d_means2 (optional) = {mean rotation error over B, mean translation error over the valid entries (NaN if none)}. */
int e2emv_pose_error_means(e2emv_ctx* ctx, int B, const float* d_T, const float* d_T_gt, float* d_rot_err, float* d_transl_err,
                           uint8_t* d_transl_valid, float* d_means2, void* stream);

/* compute_rotation_error / compute_translation_error_as_angle(reduce=False)
 * (compute_pose_error.py:3-22), radians; entries with |t0||t1| <= 1e-6 give 0.             */
int e2emv_pose_errors(e2emv_ctx* ctx, int B, const float* d_T, const float* d_T_gt, float* d_rot_err,
                      float* d_transl_err, void* stream);

/* Two-view Levenberg-Marquardt bundle adjustment: run_bundle_adjust_2_view / BundleAdjustGaussNewton2View.run
 * (estimate_relative_pose.py:138-144, bundle_adjust_gauss_newton_2_view.py:127-201) - the refinement the reference
 * applies to the w8pt pose in its default eval mode.  d_kpts0n/d_kpts1n [B,N,2] intrinsics-normalised keypoints
 * (info["kpts*_norm"]), d_conf [B,N] (entries <= 0 are ignored), d_T_init [B,4,4].  d_T_out [B,4,4] = best pose
 * (copy of d_T_init where d_valid[b] == 0, i.e. fewer than 7 usable matches).                                  */
int e2emv_ba_2view(e2emv_ctx* ctx, int B, int N, const float* d_kpts0n, const float* d_kpts1n, const float* d_conf,
                   const float* d_T_init, int n_iterations, float* d_T_out, uint8_t* d_valid, void* stream);

/* ---- validation targets and loss -----------------------------
 * compute_gt_matches_of_image_pair (helpers.py:121-203): ground-truth matches of a pair from depth maps and the
 * relative pose.  d_kpts0/1 [B,N,2] pixel xy (truncated to integers like the reference), d_K0/1 [B,4,4],
 * d_T0to1 [B,4,4], d_depth0/1 [B,H,W].  Outputs d_indices [B,2,N+1] int64 (slot N = dustbin, -1 = no match) and
 * d_weights [B,2,N+1] f32 - the tensors helpers.py:203 stacks ("gt_indices_i_j", "gt_weights_i_j").            */
int e2emv_gt_matches(e2emv_ctx* ctx, int B, int N, const float* d_kpts0, const float* d_kpts1, const float* d_K0,
                     const float* d_K1, const float* d_T0to1, const float* d_depth0, const float* d_depth1, int H, int W,
                     float max_matched_reproj_err, float min_unmatched_reproj_err, int64_t* d_indices, float* d_weights,
                     void* stream);
/* compute_match_loss (helpers.py:228-241) on d_logZ [B,N+1,N+1] -> d_loss[0] (already divided by B). */
int e2emv_match_loss(e2emv_ctx* ctx, int B, int N, const float* d_logZ, const int64_t* d_indices, const float* d_weights,
                     float* d_loss, void* stream);

/* ---- multi-view pose back-end --------------------------------
 * HOST-side global initialisation = the reference's `ba_initializer` executable (bundle_adjustment/ba_init/src/
 * ba_init.cpp:10-90): robust rotation averaging (Chatterjee & Govindu 2013; L1 steps then IRLS) followed by
 * least-unsquared-deviation positions (Ozyesil & Singer 2015), both fp64 on the host like the reference (tiny
 * problems: <= 64 views).  All pointers are HOST memory, no context needed.  Rotations are COLUMN-major 3x3
 * (the CSV order), world -> camera; pair e = (id0, id1) carries R_021 and the position of camera id1 in the frame of
 * camera id0 (ba_init.cpp:30-50).  View 0 is the gauge.  out_t = -R * position (ba_init.cpp:65).
 * *status: bit 1 = rotation averaging failed, bit 2 = position estimation failed (results then = best so far).      */
int e2emv_mv_init(int n_views, const double* init_R, int n_pairs, const int32_t* pair_ids, const double* pair_R,
                  const double* pair_pos, double* out_R, double* out_t, int32_t* status);
/* The two estimators on their own (angle-axis in/out), as the reference's gtests exercise them
 * (ba_init/test/test_ba_init.cpp:95-268).  rot_aa [n_views,3] is the initial guess on input.                        */
int e2emv_mv_estimate_rotations(int n_views, int n_pairs, const int32_t* pair_ids, const double* pair_rot_aa,
                                double* rot_aa);
int e2emv_mv_estimate_positions(int n_views, int n_pairs, const int32_t* pair_ids, const double* pair_pos,
                                const double* rot_aa, double* out_pos);
/* `ba_init_in.csv` -> `ba_init_out.csv` (ba_initializer.cpp:7-23; wire format ba_init.cpp:13-51, 58-75).           */
int e2emv_mv_init_files(const char* in_csv, const char* out_csv);

/* Weighted reprojection bundle adjustment of <= E2EMV_MAX_TUPLE cameras on the DEVICE = the reference's
 * `bundle_adjuster` executable (problem/include/ba_problem.h:60-151, problem/src/ba_problem.cpp:115-157: Ceres
 * DENSE_SCHUR, squared loss, default options, restated as a Levenberg-Marquardt trust-region loop with an analytic
 * Schur complement; one workgroup runs the whole optimisation).  All pointers are HOST memory (the wire format is
 * CSV / numpy): intr = {fx, fy, cx, cy}; observation o = (cam_idx, pt_idx, obs_xy[2], obs_w[2] = residual weights);
 * cams [n_cams,6] = angle-axis + translation, world -> camera, in/out; pts [n_pts,3] in/out.  The fixed camera is
 * predicted with the IDENTITY pose and its row is returned untouched (ba_problem.cpp:129-137).  summary[4] =
 * {initial cost, final cost, iterations, termination: 0 max-iterations 1 gradient 2 parameter 3 function tolerance,
 * 4 five invalid steps, 5 radius underflow}.  Synchronous.                                                           */
int e2emv_mv_bundle_adjust(e2emv_ctx* ctx, int n_cams, int fixed_cam, int n_pts, int n_obs, const double* intr,
                           const int32_t* cam_idx, const int32_t* pt_idx, const double* obs_xy, const double* obs_w,
                           double* cams, double* pts, int max_iterations, double* summary, void* stream);
/* `ba_in.csv` -> `ba_out.csv` (bundle_adjuster.cpp:7-23; parser ba_problem.cpp:8-88, writer :98-113), 50 iterations. */
int e2emv_mv_bundle_adjust_files(e2emv_ctx* ctx, const char* in_csv, const char* out_csv, void* stream);
/* cv2.triangulatePoints as used by write_bundle_adjust_problem (bundle_adjust_io.py:226-227): homogeneous DLT of
 * two views, one thread per point, fp64.  HOST pointers: P0, P1 [3,4] row-major, x0, x1 [n,2], xyz [n,3].           */
int e2emv_mv_triangulate(e2emv_ctx* ctx, int n, const double* P0, const double* P1, const double* x0, const double* x1,
                         double* xyz, void* stream);

/* ---- SuperPoint front-end --------------------------------------
 * Replaces models.models.superpoint.SuperPoint (absent submodule; call sites helpers.py:73-96, configs train.py:335-341,
 * eval_pairs.py:197-202, eval_multi_view.py:135-140) = upstream magicleap SuperPoint.  Weights go in through
 * e2emv_set_weight with the upstream parameter names prefixed by "superpoint." (superpoint.conv1a.weight [64,1,3,3] ...
 * superpoint.convDb.bias [256]; layers conv1a conv1b conv2a conv2b conv3a conv3b conv4a conv4b convPa convPb convDa
 * convDb), then e2emv_superpoint_commit repacks them for NHWC implicit-GEMM convolutions.                          */
typedef struct {
    int32_t batch;             /* images in this call (a merged tuple batch, helpers.py:73-81)                      */
    int32_t height, width;     /* multiples of 8 (the storage grid; see valid_height / valid_width)                   */
    int32_t nms_radius;        /* config "nms_radius"                                                                 */
    int32_t max_keypoints;     /* K: capacity of the outputs, 1..4096 (config "max_keypoints"; -1 is mapped by the
                                  Python layer to the capacity)                                                       */
    int32_t remove_borders;    /* config "remove_borders"                                                             */
    int32_t fill_random;       /* fork option "fill_with_random_keypoints": pad to K with pseudo-random pixels        */
    float keypoint_threshold;  /* config "keypoint_threshold"                                                         */
    uint32_t seed;             /* for fill_random                                                                     */
    int32_t valid_height, valid_width; /* 0 = height / width.  Otherwise the size of the images themselves, stored zero-padded
                                  on the (height, width) grid with height - valid_height < 8 (same for the width): the
                                  convolutions see the whole image and every max-pool floors, like upstream            */
} e2emv_superpoint_desc;
int e2emv_superpoint_commit(e2emv_ctx* ctx);
/* d_images [B,H,W] fp32 grey in [0,1].  Outputs: d_kpts [B,K,2] pixel (x, y), d_scores [B,K], d_desc [B,256,K]
 * (descriptor-major, what the matcher's descriptors{m} input wants), d_count [B] valid entries per image (the rest is
 * zero).  Order: row-major when an image has <= K candidates, else score-descending (ties: lower pixel index).
 * d_score_map (optional) [B,H8,W8] (H8 = valid height rounded down to a multiple of 8) receives the NMS-ed score map.                                                        */
int e2emv_superpoint_forward(e2emv_ctx* ctx, const e2emv_superpoint_desc* desc, const float* d_images, float* d_kpts,
                             float* d_scores, float* d_desc, int32_t* d_count, float* d_score_map, void* stream);

/* ---- building blocks exported for per-kernel parity tests and micro-benchmarks ----- */
/* C[z][m][n] = act(sum_k A[z][m][k] W[z][n][k] * scale + bias[n]) (+ R[z][m][n]); all f32;
 * A may be split in two K-segments (A: k < K1, A2: K1 <= k < K).  flags: bit0 relu.        */
int e2emv_gemm_nt(e2emv_ctx* ctx, int batch, int M, int Nout, int K, int K1, const float* d_A, int64_t lda,
                  int64_t strideA, const float* d_A2, int64_t lda2, int64_t strideA2, const float* d_W,
                  int64_t ldw, int64_t strideW, const float* d_bias, const float* d_R, int64_t ldr,
                  int64_t strideR, float* d_C, int64_t ldc, int64_t strideC, float scale, int flags,
                  void* stream);
/* Multi-head attention over head-major channels: d_qkv [n_img, n_rows, 3*D] (q|k|v), image
 * g = b*T + t attends to itself (cross == 0) or to every other image of its tuple;
 * n_valid keys/queries per image; output d_out [n_img, n_rows, D].                         */
int e2emv_attention(e2emv_ctx* ctx, int B, int T, int n_rows, int n_valid, int D, int H, const float* d_qkv,
                    int cross, float* d_out, void* stream);

/* ---- arithmetic of the dense GNN contractions -----------------------------------------
 * E2EMV_PRECISION_F32    : v_mfma_f32_32x32x2_f32, exact fp32 products (the audit mode).
 * E2EMV_PRECISION_BF16X3 : fp32 operands split into three bf16 planes, six bf16 MFMA products per
 *                          block accumulated in fp32 - fp32-class rounding (|err| ~ 2^-24 per
 *                          product) at 2.67x the fp32-MFMA ceiling.  Same API, same outputs within
 *                          the 1e-4 parity bar (tests/test_gpu_matcher.py runs every mode).
 * E2EMV_PRECISION_F16X2  : fp32 operands carried as two fp16 planes (hi + 2^-11 lo', 22 significant bits, the low plane
 *                          kept at the exponent range of the high one; static weights pre-scaled by a power of two per
 *                          matrix), three fp16 MFMA products per block accumulated in fp32: half the matrix-pipe work
 *                          of bf16x3.  Operand representation error <= 2^-22 for 6e-5 <= |x| <= 65504 (a K=512
 *                          contraction: 1e-7 rms, below fp32 accumulation noise); an activation beyond 65504 becomes
 *                          +-inf and surfaces as the sticky non-finite error of e2emv_sync.  DEFAULT.
 * Also selectable with the environment variable E2EMV_PRECISION=f32|bf16x3|f16x2 read at e2emv_create. */
#define E2EMV_PRECISION_F32 0
#define E2EMV_PRECISION_BF16X3 1
#define E2EMV_PRECISION_F16X2 2
int e2emv_set_precision(e2emv_ctx* ctx, int precision);
int e2emv_get_precision(e2emv_ctx* ctx, int* precision);
/* The split-operand modes use the fp32-MFMA kernels for calls with fewer than `min_rows` keypoint rows (images x
 * keypoints; default -1 = half a 128-row tile per CU, i.e. 16384 on an MI355X): their 64 x 64 tiles and key-split
 * attention are the latency-tuned forms for a pair or two.  0 = always the split kernels (how the parity tests reach
 * them at small sizes). */
int e2emv_set_split_min_rows(e2emv_ctx* ctx, int64_t min_rows);
/* building blocks of the bf16x3 path on fp32 buffers (split / merge done internally; for tests):
 * C = act(A W^T + bias), A [M,K], W [N,K], C [M,N]; flags bit0 relu, bit1 = first-generation kernel reading pre-split
 * activation planes (gemm3.hip) instead of the default gemm_x3.hip (fp32 activations split on the way into LDS),
 * bit2 = the f16x2 GEMM, gemm_h2.hip (host-synchronising: the weight planes are made on the host). */
int e2emv_gemm_bf16x3(e2emv_ctx* ctx, int M, int Nout, int K, const float* d_A, const float* d_W, const float* d_bias,
                      float* d_C, int flags, void* stream);
/* same contract as e2emv_attention; cross: bit0 = cross layer, bit1 = the forward pass's kernel (fp32 q|k|v in, the
 * planes made inside the kernel) instead of the pre-split building block, bit2 = its f16x2 form. */
int e2emv_attention_bf16x3(e2emv_ctx* ctx, int B, int T, int n_rows, int n_valid, int D, int H, const float* d_qkv,
                           int cross, float* d_out, void* stream);

/* ---- f16x2 on plane activations (the default implementation of E2EMV_PRECISION_F16X2) ---------------------------
 * Activations live in HBM as the two fp16 planes of the f16x2 arithmetic (4 bytes per element like fp32, 32-column
 * blocks of {hi, lo}); every producer splits its output once in its epilogue, consumers move the planes from global
 * memory straight into LDS.  generation 5 (default) = these kernels (gemm_p2.hip; attention_p2w.hip - one wave per SIMD,
 * matrix and softmax work interleaved inside the wave - above 256 keys, attention_p2.hip below; needs descriptor_dim 256 /
 * 4 heads, other widths use generation 2) with the row-local GEMMs between two attentions (MLP0, MLP1, the next layer's
 * q | k | v) chained per 256-row block in ONE launch where that takes no more tile rounds than three launches (gemm_p2c.hip;
 * bit-identical results; 105 = chained on every shape that allows it), 4 = a launch per GEMM, 3 = the same with the round-3 attention (attention_p2.hip everywhere), 2 = the
 * round-2 kernels that keep fp32 activations and split them inside the consuming GEMM / attention (kept as A/B arms and
 * for other widths).  Also E2EMV_F16X2_KERNELS=r2 | r3 | r4 at e2emv_create. */
int e2emv_set_f16x2_kernels(e2emv_ctx* ctx, int generation);
/* attention_p2w (f16x2, above 256 keys): when the (image, head, 256-query) items of a launch leave the last round of workgroups at most
 * half full - tuple_size 5 at 1024 keypoints: 640 items on 256 CUs, three rounds for 2.5 of work; a pair or two per call: fewer items
 * than CUs - the leftover items are split along the keys over the idle CUs and merged by a small second launch (on = 1, default; the
 * reference's shapes: eval_multi_view.py:154-162).  Results differ from the unsplit kernel by the order of the softmax sums only
 * (~1e-7 relative).  on = 0: never (A/B runs, tests). */
int e2emv_set_attention_key_split(e2emv_ctx* ctx, int on);
/* building blocks on fp32 buffers (conversion to / from planes done by helper kernels; for tests and micro-benchmarks):
 * C = act([A | A2] W^T + bias) (+ R); A [M,K1], A2 [M,K-K1] or NULL, W [N,K], R [M,N] or NULL.  flags: bit0 relu, bit1 the
 * kernel writes planes (converted back to fp32 afterwards) instead of fp32, bit2 = with the tile exponents of the range
 * side-band (M, K1, K - K1, N multiples of 64), bits 8.. = number of timed repetitions.  The
 * weight planes are made on the host as e2emv_commit_weights makes them (host-synchronising). */
int e2emv_gemm_p2(e2emv_ctx* ctx, int M, int Nout, int K, int K1, const float* d_A, const float* d_A2, const float* d_W,
                  const float* d_bias, const float* d_R, float* d_C, int flags, void* stream);
/* the q|k|v projection with its attention-operand epilogue (q | k planes + transposed V planes), read back as one fp32
 * matrix: d_X [n_img*n_rows, D], d_W [3D, D] head-major, d_qkv [n_img*n_rows, 3D]. */
int e2emv_qkv_p2(e2emv_ctx* ctx, int n_img, int n_rows, int D, int H, const float* d_X, const float* d_W, const float* d_bias,
                 float* d_qkv, void* stream);
/* same contract as e2emv_attention on the plane kernel; flags: bit0 cross, bit1 / bit2 force attention_p2 with 4 / 8 waves per workgroup, bit3 forces attention_p2w (one wave per SIMD),
 * bits 8.. = timed repetitions. */
int e2emv_attention_p2(e2emv_ctx* ctx, int B, int T, int n_rows, int n_valid, int D, int H, const float* d_qkv, int flags,
                       float* d_out, void* stream);

/* The matched descriptors of the LAST e2emv_matcher_forward call on this context (upstream's mdesc0 / mdesc1 = final_proj
 * output, models/superglue.py:269 upstream; with multi_frame_matching off and T > 2: of its last pair): d_out
 * [n_img = B*T][n_kpts][dim] fp32, keypoint-major.  d_out == NULL only reports the three sizes.  They live in the context's
 * workspace: any other call that uses the workspace (w8pt, BA, sinkhorn, superpoint, ...) ends their life, and a later
 * get_descriptors returns E2EMV_ESTATE instead of stale memory.  An audit output: the
 * quantity the arithmetic modes of the GNN differ in (tests/test_gpu_round4.py, tools/parity_margins.py). */
int e2emv_get_descriptors(e2emv_ctx* ctx, float* d_out, int64_t capacity_floats, int* n_img, int* n_kpts, int* dim, void* stream);

/* ---- range statistics ------------------------------------------------------------------------------------------------
 * The f16x2 mode keeps its fp16 planes inside fp16's range with one exponent per block of 64 x 64 activations (see
 * DESIGN.md 4d): there is no out-of-range fallback to take.  stats[0] = plane blocks that needed a non-zero exponent since
 * the last reset (0 for an ordinary network: every block sat inside the dead zone and took the plain paths), stats[1] =
 * Sinkhorn problems whose scores were non-finite (their outputs are NaN / inf and e2emv_sync reports them), stats[2] =
 * Sinkhorn problems the exponential-domain resident kernel could not finish (a scaling left fp32's range, or an
 * inter-workgroup wait gave up under contention) and the rescue pass behind it re-solved in the log domain inside the same
 * call: their outputs are correct, nothing is raised.  The host tells the two causes apart: the SECOND observation (here or
 * in e2emv_sync) of calls with range rescues moves the context to the log-domain launch chain - after 16 calls on it the
 * resident kernel gets another try -, rescues behind a timeout (stats[4] counts those) never do; stats[5] = Sinkhorn calls served by the 128-rows-per-workgroup
 * kernel (a batch of 513 .. 1024-column problems that it brings through in fewer rounds); stats[3] = (wave, stream, 64-key tile) softmaxes that
 * attention_p2w redid on its slow path (a performance counter: results are the same).  Host-synchronising. */
int e2emv_get_stats(e2emv_ctx* ctx, uint64_t* stats, int n, int reset);

/* ---- which Sinkhorn kernel serves a call (upstream log_optimal_transport: models/superglue.py:156-186; reference call sites
 * /root/reference/helpers.py:246, eval_pairs.py:212) --------------------------------------------------------------------------
 * E2EMV_SINKHORN_AUTO (default): by shape AND batch size - the kernels with the couplings in registers addressed by number take
 * their full rounds, the remainder goes to the compiler-allocated kernel.  The kernels sum in different orders: a
 * problem's log-assignment can differ by < 2e-5 between two batch compositions.  ROWS64 / ROWS128 pin one kernel kind for every
 * call of the context: a problem's result then does not depend on its batch neighbours, bit for bit.  STREAM = the log-domain
 * launch chain (no resident kernel; also taken for iters == 0).  The initial value comes from the environment variable
 * E2EMV_SINKHORN = rows64 | rows128 | stream, read ONCE at e2emv_create. */
#define E2EMV_SINKHORN_AUTO 0
#define E2EMV_SINKHORN_ROWS64 1
#define E2EMV_SINKHORN_ROWS128 2
#define E2EMV_SINKHORN_STREAM 3
int e2emv_set_sinkhorn_kernel(e2emv_ctx* ctx, int kernel);
/* The launcher's plan for a batch of B problems of M x N scores and `iters` iterations on this context (what e2emv_sinkhorn /
 * e2emv_matcher_forward would launch now; nothing is launched): plan[0] = number of resident launches (0 = the log-domain chain, 1
 * or 2), then per launch 4 ints: rows of a problem per workgroup, problems of the batch it takes, problems resident at a time,
 * rounds.  n = capacity of plan (>= 9 for everything).  bench.py reports its Sinkhorn bound from this instead of re-deriving it. */
int e2emv_sinkhorn_plan(e2emv_ctx* ctx, int B, int M, int N, int iters, int* plan, int n);

/* ---- training: the matcher with a tape, the backward of the match loss and (through the confidences) of the pose loss ------
 * Replaces, for stage-1 training (match loss only), what torch.autograd does under the reference's
 *   pred = run_matcher(...); train_loss.backward()        (/root/reference/helpers.py:243-260, train.py:406-425)
 * with the reference's own arithmetic (fp32).  Differentiated: keypoint encoder, every GNN layer (projections, attention,
 * merge, MLP), final_proj, the score matrix, bin_score and the unrolled log-domain Sinkhorn (SuperGlue's
 * log_optimal_transport, models/superglue.py:156-186 upstream).  BatchNorm layers use their running statistics (frozen) and
 * still hand back gradients for their affine parameters.  The pose loss (helpers.py:253-258) reaches the network through the
 * confidences: e2emv_w8pt_backward (below) -> e2emv_conf_forward_train / d_dconf here.  Forward-only: batch-statistics
 * BatchNorm, images with differing keypoint counts, the confidences of a model WITHOUT conf_mlp (the match score).
 *
 * e2emv_train_commit        folds the weights handed over with e2emv_set_weight into the training arena (call again after
 *                           every optimiser step, after re-sending the changed tensors).
 * e2emv_matcher_forward_train   same inputs as e2emv_matcher_forward (fd->n_kpts keypoints in every image); outputs the
 *                           log assignment matrices d_logZ[pair] = [batch][n_kpts + 1][n_kpts + 1] fp32, pairs in the
 *                           order (0,1), (0,2), (1,2), ... and keeps the tape of this call in the context.
 * e2emv_conf_forward_train  (models with conf_mlp; the pose loss) confidence head of one pair on the tape's matched descriptors:
 *                           d_conf [batch][n_kpts] = sigmoid(conf_mlp([mdesc_i[n] | mdesc_j[match n]])) for matched n, else 0;
 *                           d_matches0 = that pair's matches (e.g. from e2emv_extract_matches on d_logZ), kept alive by the
 *                           caller until the backward.
 * e2emv_matcher_backward    d_dlogZ[pair] = dLoss / dlogZ of the last forward_train (same layout; a null entry = zero),
 *                           d_dconf (may be null) [pair] = dLoss / dconf of e2emv_conf_forward_train (a null entry = zero):
 *                           leaves the gradient of every upstream parameter in the context.
 * e2emv_get_grad            copies the gradient of parameter `key` (the reference's state_dict name, an optional "module."
 *                           prefix is ignored) into d_dst (device, fp32, numel elements); stream-ordered.                 */
int e2emv_train_commit(e2emv_ctx* ctx, const e2emv_model_desc* model);
/* After an optimiser step (helpers.py / train.py: optimizer.step() between two forwards): the new values of the model's
 * floating-point state_dict tensors straight from DEVICE memory (d_params[i] = fp32, contiguous, numels[i] elements, name
 * keys[i]; tensors the differentiable path does not use are ignored) into the training arena that e2emv_train_commit built for
 * this model - device-to-device copies and the BatchNorm / head-order folds as kernels, stream-ordered, no host copy and no
 * synchronisation.  bin_score = the scalar parameter's value.  E2EMV_ESTATE when the context holds no arena of this model
 * (then: e2emv_set_weight for every tensor + e2emv_train_commit, once).  The host-side weight store of e2emv_set_weight is
 * NOT updated: hand the tensors over again before the next e2emv_commit_weights.                                          */
int e2emv_train_update(e2emv_ctx* ctx, const e2emv_model_desc* model, int n, const char* const* keys, const float* const* d_params,
                       const int64_t* numels, float bin_score, void* stream);
int e2emv_matcher_forward_train(e2emv_ctx* ctx, const e2emv_forward_desc* fd, const float* const* d_kpts, const float* const* d_kscores,
                                const void* const* d_desc, float* const* d_logZ, void* stream);
int e2emv_conf_forward_train(e2emv_ctx* ctx, int pair, const int64_t* d_matches0, float* d_conf, void* stream);
int e2emv_matcher_backward(e2emv_ctx* ctx, const float* const* d_dlogZ, const float* const* d_dconf, void* stream);
int e2emv_get_grad(e2emv_ctx* ctx, const char* key, float* d_dst, int64_t numel, void* stream);

/* Backward of e2emv_w8pt with respect to the confidences (training, the pose loss of helpers.py:253-258: rot / translation
 * error of run_weighted_8_point's pose).  Inputs: the normalised correspondences and the pose the forward returned (info
 * "kpts0_norm" / "kpts1_norm", T), the confidences it was given, d_gT = dLoss/dT [B][4][4]; output d_gconf [B][N].  The
 * derivative of the eigenvector is analytic, the 9 -> 12 Jacobian of the rank-2 / decomposition tail is taken by central
 * differences in fp64.  The candidate choice (choose_closest / cheirality) is piecewise constant: the candidate the forward
 * chose is differentiated. */
int e2emv_w8pt_backward(e2emv_ctx* ctx, int B, int N, const float* d_kpts0n, const float* d_kpts1n, const float* d_conf, const float* d_T,
                        const float* d_gT, float* d_gconf, void* stream);

/* Backward of e2emv_pose_errors with respect to d_T (compute_rotation_error / compute_translation_error_as_angle as the pose
 * loss, helpers.py:256-258): d_gT [B][4][4] = d_g_rot[b] d rot_b / dT + d_g_transl[b] d transl_b / dT. */
int e2emv_pose_errors_backward(e2emv_ctx* ctx, int B, const float* d_T, const float* d_T_gt, const float* d_g_rot, const float* d_g_transl,
                               float* d_gT, void* stream);

/* ---- timing hooks used by bench.py (HIP events on the caller's stream) -------------
 * After e2emv_profile(ctx, 1) every kernel family launched by the library is bracketed by
 * HIP events on its stream; e2emv_profile_read returns accumulated milliseconds and launch
 * counts per family since the last reset (host-synchronising).                             */
#define E2EMV_PROF_SLOTS 16
int e2emv_profile(e2emv_ctx* ctx, int enable);
int e2emv_profile_read(e2emv_ctx* ctx, float* ms, int64_t* launches, int n_slots, int reset);
const char* e2emv_profile_name(int slot);

/* ---- the one collective of the path: metric gather / reduction over the ranks of a node (RCCL over xGMI) --------------------------
 * Replaces, for a host without torch.distributed, the reference's init_process_group(backend="nccl") (/root/reference/train.py:
 * 270-277) and its only explicit collective, the all_reduce of the validation loss (train.py:102-106); the evaluation scripts'
 * per-pair pose errors are gathered with the same communicator for the exact sort-based AUC.  One process per GPU; tuples are
 * sharded, the data path has no collective.  librccl.so.1 is dlopen'ed by the first of these calls (E2EMV_ESTATE if absent).
 *   rank 0: e2emv_comm_unique_id -> hand the 128 bytes to every rank (file, environment, the launcher's store) ->
 *   every rank: e2emv_comm_init (collective: returns when all `world` ranks called it) -> e2emv_metric_* -> e2emv_comm_destroy.
 * e2emv_comm_init_file does the hand-over through a file all ranks see (rank 0 writes path atomically, the others poll up to
 * timeout_s seconds).  d_* buffers are device memory of the context's device; calls are ordered on `stream`. */
#define E2EMV_COMM_ID_BYTES 128
typedef struct e2emv_comm e2emv_comm;
int e2emv_comm_unique_id(e2emv_ctx* ctx, void* id_out /* [E2EMV_COMM_ID_BYTES] */);
int e2emv_comm_init(e2emv_ctx* ctx, const void* id, int rank, int world, e2emv_comm** out);
int e2emv_comm_init_file(e2emv_ctx* ctx, const char* path, int rank, int world, double timeout_s, e2emv_comm** out);
int e2emv_comm_destroy(e2emv_ctx* ctx, e2emv_comm* comm);
int e2emv_comm_rank(const e2emv_comm* comm, int* rank, int* world);
/* d_all [world][n] <- every rank's d_local [n], rank order, on every rank */
int e2emv_metric_allgather(e2emv_ctx* ctx, e2emv_comm* comm, const float* d_local, int n, float* d_all, void* stream);
/* d_buf [n] <- op over the ranks, in place */
#define E2EMV_REDUCE_SUM 0
#define E2EMV_REDUCE_MAX 1
#define E2EMV_REDUCE_MIN 2
int e2emv_metric_allreduce(e2emv_ctx* ctx, e2emv_comm* comm, float* d_buf, int n, int op, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* E2EMV_H */
