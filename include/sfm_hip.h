/*
 * sfm_hip.h — C-ABI of libsfmhip.so, the MI355X (gfx950) back-end for the
 * incremental-SfM hot path of FlagArihant2000/sfm-mvs.
 *
 * The reference has no FFI of its own: its operator boundary is the set of
 * cv2.* calls made by sfm.py.  Each entry point below names the reference call
 * site (file:line under /root/reference) whose arithmetic it replaces; the
 * ctypes stubs a maintainer would add to sfm.py are shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Every `*_dev` pointer is a
 *     DEVICE pointer (HBM) owned by the caller; `stream` is a hipStream_t
 *     passed as void* (NULL = the null stream).
 *   - All functions are stream-ordered and asynchronous; none synchronises.
 *   - The library allocates nothing persistent: scratch is caller-provided
 *     workspace, sized by the `_ws_bytes` twin of each call.
 *   - Return value: 0 = OK, negative = error code below; a human-readable
 *     message for the calling thread's last error is at sfm_last_error().
 *   - There is NO CPU fallback.  If no HIP device is usable the call fails
 *     with SFM_ERR_DEVICE.
 */
#ifndef SFM_HIP_H
#define SFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_ABI_VERSION 3   /* 3 (round 6): + sfm_host_poll_count, sfm_debug_pnp_sweep_server; sfm_build_id() names every source file */

#define SFM_OK             0
#define SFM_ERR_ARG       -1   /* null pointer, negative size, unsupported dim, misaligned pointer/stride */
#define SFM_ERR_WORKSPACE -2   /* ws_bytes smaller than the _ws_bytes twin reports */
#define SFM_ERR_DEVICE    -3   /* HIP runtime error (launch failure, no device) */

int         sfm_abi_version(void);
const char* sfm_last_error(void);
/* What the loaded binary was built from: "knn.hip:<sha256 of csrc/knn.hip's code> assoc.hip:<first 16 digits> ... sfm_hip.h:<16>",
 * one entry per source file (comments and whitespace do not count; scripts/knn_code_hash.py --build-id prints the same for a tree),
 * " dev-build" appended when the library honours the SFM_KNN_* tuning variables (release builds read none).  The parity sweeps and
 * PMC stamps under profiles/ name these hashes, each family its own sources. */
const char* sfm_build_id(void);

/* ------------------------------------------------------------------------
 * A2  cv2.BFMatcher().knnMatch(des0, des1, k=2)           sfm.py:259-260
 *     (also isfm.py:47,71; test.py:41-42,225,352)
 *
 * Brute-force L2 2-nearest-neighbour of every query row among the train rows.
 * Semantics of OpenCV's batchDistance(NORM_L2, K=2): dist = sqrtf(sum (q-t)^2)
 * in float32, best-2 kept sorted ascending, a candidate replaces only on
 * strict `<`, so on ties the LOWER train index stays ahead.
 *
 *   q_dev   [nq x dim] float32, row stride ldq (elements)   — des0
 *   t_dev   [nt x dim] float32, row stride ldt (elements)   — des1
 *   idx_dev [nq x 2]   int32   trainIdx of 1st / 2nd neighbour (-1 if nt < k)
 *   dist_dev[nq x 2]   float32 DMatch.distance            (+inf if nt < k)
 *   stats_dev (optional, may be NULL) int32[4]:
 *       [0] queries for which at least one filter stream had to be rescanned exactly
 *       [1] filter workgroups launched   [2] candidate streams reserved per query
 *       [3] filter arithmetic that ran: 0 fp16 single product, exact inputs; 1 fp16 single product;
 *           2 bf16 hi/mid split; 3 fp32 MFMA; 4 exact-integer i8 MFMA (u8-integer descriptors: real SIFT output);
 *           5 the i8 MFMA body on float descriptors QUANTISED to 8 bits (certified by their measured residual norms)
 *
 * dim must be 128 (SIFT); q_dev/t_dev 16-byte aligned; ldq, ldt multiples of 4; nt <= 4 000 000.
 * The result is bit-identical to the direct-form float32 evaluation for any finite input whose squared row
 * norms are finite in float32 (|x| up to ~1e18; see docs/knn.md, "Domain of the parity claim").  Beyond that the
 * filters' scores ||t||^2 + ||q||^2 - 2 q.t are +inf / NaN while some direct-form distances are still finite: measured
 * wrong at 3e18, right again from 1e19 on, where every distance is +inf and the index order decides (scripts/dev/q8_huge.py of the round-5 tree).
 *
 * `filter` — the candidate filter that runs before the exact refine.  A per-call argument (ABI 1 had
 * a process-global switch); results are bit-identical whichever runs, the _ws_bytes twin takes the same value:
 *   SFM_KNN_FILTER_AUTO       MFMA filter, fragments streamed L2 -> registers (knn_filter_q4_kernel); its
 *                             arithmetic is chosen ON THE DEVICE from the data: the exact-integer body
 *                             (v_mfma_i32_32x32x32_i8, i32 scores, integer certificate) when every value of the batch is an
 *                             integer 0 .. 255 — what cv2 SIFT emits (sfm.py:246-252) —; the same body on the pairs'
 *                             values QUANTISED to 8 bits (x ~ lo + s k, one grid per pair from a sample of its rows, the
 *                             residual norms measured, | ||q - t|| - s sqrt(D) | <= ||q - q^|| + ||t - t^|| as the
 *                             certificate, survivors re-evaluated in float32) when the sampled values have compact
 *                             support (range <= 5 standard deviations: e.g. uniform data) and the grid turns out to
 *                             fit; else one fp16 product when the values fit fp16's range, else the three-product
 *                             bf16 hi/mid split
 *   SFM_KNN_FILTER_NOQUANT    as AUTO, but float data never run quantised (the exact-integer body still serves u8 data)
 *   SFM_KNN_FILTER_HALF       as AUTO without the exact-integer body (16-bit arithmetic whatever the data)
 *   SFM_KNN_FILTER_F32        fp32 MFMA filter (single pair only)
 *   SFM_KNN_FILTER_SPLIT      as AUTO but pinned to the bf16 split
 *   SFM_KNN_FILTER_LDS, SFM_KNN_FILTER_LDS_SPLIT   round 2's LDS-ring kernel (knn_filter_split2_kernel), auto / pinned
 * ---------------------------------------------------------------------- */
#define SFM_KNN_FILTER_AUTO      0
#define SFM_KNN_FILTER_F32       1
#define SFM_KNN_FILTER_SPLIT     2
#define SFM_KNN_FILTER_LDS       3
#define SFM_KNN_FILTER_LDS_SPLIT 4
#define SFM_KNN_FILTER_HALF      5
#define SFM_KNN_FILTER_NOQUANT   6
size_t sfm_knn2_l2_f32_ws_bytes(int64_t nq, int64_t nt, int dim, int filter);
int    sfm_knn2_l2_f32(const float* q_dev, int64_t nq, int64_t ldq,
                       const float* t_dev, int64_t nt, int64_t ldt, int dim, int filter,
                       int32_t* idx_dev, float* dist_dev, int32_t* stats_dev,
                       void* ws_dev, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * A3  Lowe ratio loop + gather                           sfm.py:262-268
 *     `if m.distance < 0.70 * n.distance: good.append(m)`
 *
 * The reference compares float32 distances promoted to Python double against
 * a double constant; the kernel does exactly that in fp64.  Survivors are
 * written in ascending queryIdx order (the order of the Python loop).
 *
 *   out_q_dev, out_t_dev [nq] int32   queryIdx / trainIdx of survivors
 *   out_count_dev        [1]  int32   number of survivors M
 *   mask_dev (optional)  [nq] uint8   1 where the query passed
 *   idx_dev / dist_dev must be 8-byte aligned (they are the outputs of sfm_knn2_l2_f32)
 * ---------------------------------------------------------------------- */
size_t sfm_ratio_compact_ws_bytes(int64_t nq);
int sfm_ratio_compact(const int32_t* idx_dev, const float* dist_dev, int64_t nq,
                      double ratio, int32_t* out_q_dev, int32_t* out_t_dev,
                      int32_t* out_count_dev, uint8_t* mask_dev,
                      void* ws_dev, size_t ws_bytes, void* stream);

/* A2 + A3 in one call: the matcher part of find_features (sfm.py:259-266).  Same outputs as
 * sfm_knn2_l2_f32 followed by sfm_ratio_compact, bit for bit; the Lowe test is folded into the
 * last KNN kernel, which saves one launch per image pair.  Workspace: sfm_match_l2_f32_ws_bytes. */
size_t sfm_match_l2_f32_ws_bytes(int64_t nq, int64_t nt, int dim, int filter);
int sfm_match_l2_f32(const float* q_dev, int64_t nq, int64_t ldq,
                     const float* t_dev, int64_t nt, int64_t ldt, int dim, int filter, double ratio,
                     int32_t* idx_dev, float* dist_dev,
                     int32_t* out_q_dev, int32_t* out_t_dev, int32_t* out_count_dev,
                     uint8_t* mask_dev /*optional*/, int32_t* stats_dev /*optional*/,
                     void* ws_dev, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * The same for `batch` (1..8) image pairs of ONE shape in one set of launches — what a caller
 * with many independent pairs (sfm.py:347 matches every consecutive pair; isfm.py:56-71 all
 * pairs) should use: the filter's unit space becomes batch x query row blocks x train tiles
 * under a single work partition, so a workgroup's prologue, the launch ramp and the kernel
 * boundaries are paid once per batch instead of once per pair (10k x 10k: 31 -> ~23 us of
 * filter time per pair at batch 4).  Results are those of `batch` separate calls, bit for bit.
 * q, t, idx, dist, out_q, out_t, out_count, mask, stats: HOST arrays of `batch` device
 * pointers (mask / stats: NULL array or NULL entries allowed); all pairs share nq, nt, ldq, ldt.
 * Not available with SFM_KNN_FILTER_F32 (the fp32-MFMA variant is single-pair).
 * ---------------------------------------------------------------------- */
size_t sfm_match_batch_l2_f32_ws_bytes(int64_t nq, int64_t nt, int dim, int batch, int filter);
int sfm_match_batch_l2_f32(int batch, const float* const* q_dev, int64_t nq, int64_t ldq,
                           const float* const* t_dev, int64_t nt, int64_t ldt, int dim, int filter, double ratio,
                           int32_t* const* idx_dev, float* const* dist_dev, int32_t* const* out_q_dev,
                           int32_t* const* out_t_dev, int32_t* const* out_count_dev, uint8_t* const* mask_dev,
                           int32_t* const* stats_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* Self-test of the hardware property the KNN certificate's "MFMA chain" term rests on: how far one MFMA is from the exact
 * c + sum a_k b_k, in units of 2^-24 (|c| + sum |a_k b_k|), maximum over 4096 waves x trials_per_wave MFMAs x 1024 outputs, for
 * seven operand regimes (equal exponents ... 2^16 spread with cancellation; the last one is the filter's own).
 * kind: 0 v_mfma_f32_32x32x16_f16, 1 v_mfma_f32_32x32x16_bf16, 2 v_mfma_f32_32x32x8_bf16 (the accumulator-init MFMA).
 * regime_max_host: 7 doubles; ws: 32 KiB + 512 B of device memory.  Synchronises `stream`. */
int sfm_selftest_mfma_accumulation(int kind, int trials_per_wave, double* regime_max_host, void* ws_dev, size_t ws_bytes,
                                   void* stream);

/* What the library measured on the CURRENT device the first time a 16-bit KNN filter ran there (it runs the self-test above
 * once per device and process: a few ms and one synchronisation of a private stream, inside that first call): the largest E
 * over the three MFMA kinds and seven regimes, and the factor applied to the certificate's chain term — 1 while E <= 8 (the
 * certificate assumes 16), E / 8 above: an unknown matrix pipe costs speed (more rescans), never exactness. */
int sfm_knn_mfma_selftest_result(double* worst_units_host, float* chain_scale_host);

/* Gather keypoint coordinates of the survivors: pts0 = kp0[out_q], pts1 = kp1[out_t]
 * (sfm.py:267-268).  kp*_dev are [n x 2] float32 (KeyPoint.pt); count_dev is the
 * device scalar written by sfm_ratio_compact; capacity = rows available in pts*_dev. */
int sfm_gather_matches(const float* kp0_dev, const float* kp1_dev,
                       const int32_t* out_q_dev, const int32_t* out_t_dev,
                       const int32_t* count_dev, int64_t capacity,
                       float* pts0_dev, float* pts1_dev, void* stream);

/* ------------------------------------------------------------------------
 * A2, train set split over `shards` devices (SURVEY 8e's fallback for one pair
 * whose train descriptors do not fit one device): merge of the shards' partial
 * knnMatch(k=2) results into the result of a single scan (sfm.py:259-260).
 *   cand_dev [shards][2][nq][2] int32: per shard the (GLOBAL) trainIdx plane and
 *   the distance-bits plane of its partial result (idx < 0: no neighbour) — the
 *   layout one all-gather of the shards' result blocks produces;
 *   out_idx_dev [nq][2] int32 (-1: none), out_dist_dev [nq][2] float32 (0: none).
 * Order: (distance, trainIdx) — the lower index wins ties, as BFMatcher's scan.
 * ---------------------------------------------------------------------- */
int sfm_knn_merge_top2(const int32_t* cand_dev, int shards, int64_t nq, int32_t* out_idx_dev, float* out_dist_dev, void* stream);

/* ------------------------------------------------------------------------
 * A9  common_points(pts1, pts2, pts3)                       sfm.py:215-239
 *
 * For every row i of pts1 ([n1 x 2] float32) the FIRST row of pts2 ([n2 x 2])
 * whose x OR y is bit-equal (the reference's broadcast `pts2 == pts1[i,:]`,
 * SURVEY 3.6-2).  Matches are written in ascending i:
 *   idx1_dev, idx2_dev [n1] int32   (indx1, indx2), first *count_dev entries valid
 *   keep2_dev [n2] uint8            0 for rows of pts2 (and pts3) named in indx2 —
 *                                   the complement the reference mask-compresses
 *   first_ws_dev [n1] int32         scratch
 * ---------------------------------------------------------------------- */
int sfm_common_points(const float* pts1_dev, int64_t n1, const float* pts2_dev, int64_t n2,
                      int32_t* first_ws_dev, int32_t* idx1_dev, int32_t* idx2_dev,
                      int32_t* count_dev, uint8_t* keep2_dev, void* stream);

/* ------------------------------------------------------------------------
 * Mask -> row indices: `pts0[mask.ravel() == 1]` (sfm.py:309: OpenCV's {0,1} essential-matrix mask),
 * `pts0[mask.ravel() > 0]` (sfm.py:313: the {0,255} cheirality mask), and the complement rows of
 * common_points (sfm.py:233-238).  idx_out_dev [n] int32 receives the rows whose mask byte passes —
 * mode 0: byte == 1, mode 1: byte != 0 — in ascending order, *count_dev their number; the row gather
 * itself is then an indexed copy.  Deterministic (two passes, no atomics).
 * ---------------------------------------------------------------------- */
size_t sfm_mask_indices_ws_bytes(int64_t n);
int sfm_mask_indices(const uint8_t* mask_dev, int64_t n, int mode, int32_t* idx_out_dev, int32_t* count_dev,
                     void* ws_dev, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * A4  cv2.triangulatePoints(P1, P2, points1, points2)    sfm.py:53
 *     + `cloud = cloud / cloud[3]`                        sfm.py:54
 *
 * Per correspondence: homogeneous DLT in fp64, X = right singular vector of the
 * smallest singular value (one-sided Jacobi as in OpenCV's SVD), cast to
 * float32.  rows = 4 → modern 4x4 system (x*P3-P1, y*P3-P2 per view);
 * rows = 6 → legacy cvTriangulatePoints 6x4 system (adds x*P2-y*P1).
 * normalise_w = 1 additionally performs the reference's float32 division by w.
 * normalise_w = 2 (rows = 4 only) is the same normalised result by a ~20x cheaper route:
 *   the smallest eigenvector of A^T A by inverse iteration (LDL^T, fp64), unit-normalised,
 *   cast and divided exactly like the faithful path — bit-identical to it on > 99.9 % of
 *   points, within 1 float32 ulp otherwise; lanes that do not converge (degenerate geometry)
 *   run the Jacobi sweeps.  The un-normalised vector (normalise_w = 0) has OpenCV's sign and
 *   is only offered by the faithful path.
 * normalise_w = 3 (rows = 4 only) is the GUARDED fast path: the result of 2 is kept only where every component of the unit
 *   vector keeps a margin of max(2^-40, 16 eps lambda1/lambda3) from the nearest float32 rounding boundary (the distance
 *   two backward-stable solutions of this point can have; lambda1/lambda3 is read off the iteration), the other points —
 *   a fraction of a percent on well-conditioned geometry, all of them when the baseline vanishes — are redone by the
 *   Jacobi sweeps in a second, compacted pass: bit-identical to normalise_w = 1 on every point.
 *   Scratch: this entry point takes no workspace.  For n >= 2^18 the guarded path asks the device's default memory pool
 *   for n / 8 + 2 ints ON THE CALLER'S STREAM (hipMallocAsync / hipFreeAsync: stream-ordered, no synchronisation) — the
 *   first pass's compact list of rejected points; if the request is refused, or the list overflows, the second pass finds
 *   the marked points by scanning X4 as it does for smaller calls.  Same results either way.
 *
 *   P1, P2        HOST pointers, 12 doubles each, row-major 3x4
 *   x1_dev,x2_dev float32; point i has x at [i*stride_pt] and y at
 *                 [i*stride_pt + stride_xy] — covers (2,N) (stride_pt=1,
 *                 stride_xy=N), (N,2) (2,1) and the reference's transposed views
 *   X4_dev        float32 [4 x n] row-major (the cv2 output layout)
 * ---------------------------------------------------------------------- */
int sfm_triangulate_dlt(const double* P1_host, const double* P2_host,
                        const float* x1_dev, const float* x2_dev, int64_t n,
                        int64_t stride_pt, int64_t stride_xy, int rows,
                        int normalise_w, float* X4_dev, void* stream);

/* ------------------------------------------------------------------------
 * A3 + A4 for `batch` (1..8) image pairs in one set of launches: the Lowe loop and keypoint gather of
 * find_features (sfm.py:262-268) followed by Triangulation (sfm.py:53-54: cv2.triangulatePoints and
 * `cloud / cloud[3]`) — what the pair-sharded path runs on a rank's own pairs before the all-gather of
 * the 3-D points.  Input are the KNN blocks sfm_knn2_l2_f32 / sfm_match_batch_l2_f32 wrote (here or on
 * another rank); nothing goes through the host and no survivor list or gathered coordinate array is
 * materialised:  survivors = queries with a second neighbour and (double)d0 < ratio * (double)d1, in
 * ascending queryIdx order; column e of the output is the triangulation of (kp0[queryIdx_e],
 * kp1[trainIdx_e]) under (P[b][0], P[b][1]), computed as sfm_triangulate_dlt(rows = 4, normalise_w = 3)
 * does (bit-identical to normalise_w = 1); columns >= the survivor count are zeroed.
 *   knn_idx_dev, knn_dist_dev, kp0_dev, kp1_dev, X4_dev, count_dev: HOST arrays of `batch` device pointers
 *     knn_idx_dev[b] [nq_b x 2] int32, knn_dist_dev[b] [nq_b x 2] float32 (8-byte aligned)
 *     kp0_dev[b], kp1_dev[b]   [n x 2] float32 KeyPoint.pt of the query / train image (8-byte aligned)
 *     X4_dev[b]                float32 [4 x cap] row-major;  count_dev[b] int32[1] (array or entries may be NULL)
 *   nq_host  `batch` query counts (<= cap);  P_host [batch][2][12] doubles (row-major 3x4 of both views)
 * ---------------------------------------------------------------------- */
size_t sfm_triangulate_matches_batch_ws_bytes(int batch, int64_t cap);
int sfm_triangulate_matches_batch(int batch, const int32_t* const* knn_idx_dev, const float* const* knn_dist_dev,
                                  const int64_t* nq_host, double ratio,
                                  const float* const* kp0_dev, const float* const* kp1_dev,
                                  const double* P_host, int64_t cap, float* const* X4_dev,
                                  int32_t* const* count_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * A5  ReprojectionError: cv2.Rodrigues + cv2.projectPoints + cv2.norm
 *                                                        sfm.py:79-100
 * A6  solvePnPRansac inlier scoring                      sfm.py:67
 * A8  OptimReprojectionError / BundleAdjustment sweep    sfm.py:104-157
 *
 * One sweep over `nobs` observations.  Observation o sees point pt_idx[o]
 * through camera cam_idx[o]; either index array may be NULL (NULL cam_idx =
 * camera 0; NULL pt_idx = point o).  Projection is OpenCV's distortion-free
 * projectPoints in fp64: X' = R(rvec) X + t, u = fx X'/Z' + cx, v = fy Y'/Z' + cy
 * (skew ignored).  Optional outputs (pass NULL to skip):
 *   proj_dev    [nobs x 2] float32  projected pixel (the `p` of sfm.py:88-90)
 *   sumsq_dev   [1] float64  sum over o of ||f32(proj) - obs||^2, i.e. the
 *               square of cv2.norm(p, pts, NORM_L2) (sfm.py:93,95); caller
 *               zeroes it; accumulation is a fixed-order two-level sum
 *   inlier_dev  [nobs] uint8  1 iff float32 squared error <= thr2 (PnP-RANSAC)
 *   JtJ_cam_dev [ncam x 36], Jtr_cam_dev [ncam x 6] float64: Gauss-Newton normal
 *               equations per camera in (rvec, tvec), residual = proj - obs
 *   JtJ_pt_dev  [npt x 9],  Jtr_pt_dev [npt x 3] float64: same per 3D point
 *   res2_dev    [1] float64  sum of ||proj - obs||^2 in fp64 (the squared error norm
 *               OpenCV's Levenberg-Marquardt compares); accumulated (+=), caller zeroes
 *   cams_dev    [ncam x 6] float64 (rvec3, tvec3);  K_host 9 doubles row-major
 *   X_dev       [npt x 3] float32 with row stride ldx
 * ---------------------------------------------------------------------- */
size_t sfm_project_residual_ws_bytes(int64_t nobs, int64_t ncam, int64_t npt);
int sfm_project_residual(const double* cams_dev, int64_t ncam, const double* K_host,
                         const float* X_dev, int64_t npt, int64_t ldx,
                         const float* obs_dev, const int32_t* cam_idx_dev,
                         const int32_t* pt_idx_dev, int64_t nobs,
                         float* proj_dev, double* sumsq_dev,
                         uint8_t* inlier_dev, float thr2,
                         double* JtJ_cam_dev, double* Jtr_cam_dev,
                         double* JtJ_pt_dev, double* Jtr_pt_dev, double* res2_dev,
                         void* ws_dev, size_t ws_bytes, void* stream);

/* Dense visibility variant (BASELINE config 4: every camera sees every point).
 * obs_dev is [ncam x npt x 2] float32.  Same arithmetic as above; per-camera
 * blocks are reduced in a fixed order and per-point blocks are owned by one lane,
 * so the result is deterministic and needs no atomics.  All outputs (sumsq
 * included) are OVERWRITTEN, not accumulated; JtJ_* / Jtr_* may be NULL. */
size_t sfm_ba_dense_sweep_ws_bytes(int64_t ncam, int64_t npt);
int sfm_ba_dense_sweep(const double* cams_dev, int64_t ncam, const double* K_host,
                       const float* X_dev, int64_t npt, int64_t ldx,
                       const float* obs_dev, double* sumsq_dev,
                       double* JtJ_cam_dev, double* Jtr_cam_dev,
                       double* JtJ_pt_dev, double* Jtr_pt_dev,
                       void* ws_dev, size_t ws_bytes, void* stream);

/* Matrix-free products with the camera-point coupling W of the dense normal equations
 *     [ B   W ] [dc]   [g_c]       W_ij = Jc_ij^T Jp_ij  (6x3),  B, C, g_c, g_p from sfm_ba_dense_sweep
 *     [ W^T C ] [dp] = [g_p]
 * for a Schur-complement solver (replaces scipy least_squares + finite differences,
 * sfm.py:138-157): S = B - W C^-1 W^T is applied as  S x = B x - W (C^-1 (W^T x)).
 *   sfm_ba_schur_wt:  u_pt_dev  [npt x 3]  = W^T x,   x_cam_dev [ncam x 6]
 *   sfm_ba_schur_w :  w_cam_dev [ncam x 6] = W v,     v_pt_dev  [npt x 3]
 * The W blocks are never stored: both Jacobians are re-derived from (cams, X) in registers, so a
 * product reads no observation data.  Fixed-order reductions: deterministic.  fp64 throughout. */
size_t sfm_ba_schur_ws_bytes(int64_t ncam, int64_t npt);
int sfm_ba_schur_wt(const double* cams_dev, int64_t ncam, const double* K_host,
                    const float* X_dev, int64_t npt, int64_t ldx,
                    const double* x_cam_dev, double* u_pt_dev,
                    void* ws_dev, size_t ws_bytes, void* stream);
int sfm_ba_schur_w(const double* cams_dev, int64_t ncam, const double* K_host,
                   const float* X_dev, int64_t npt, int64_t ldx,
                   const double* v_pt_dev, double* w_cam_dev,
                   void* ws_dev, size_t ws_bytes, void* stream);
/* Sparse visibility (observation o = camera cam_idx[o] sees point pt_idx[o]): mode 0 → out = W^T in,
 * mode 1 → out = W in.  One lane per observation, fp64 atomics (order-dependent in the last bits). */
size_t sfm_ba_schur_indexed_ws_bytes(int64_t ncam);
int sfm_ba_schur_indexed(const double* cams_dev, int64_t ncam, const double* K_host,
                         const float* X_dev, int64_t npt, int64_t ldx,
                         const int32_t* cam_idx_dev, const int32_t* pt_idx_dev, int64_t nobs, int mode,
                         const double* in_dev, double* out_dev,
                         void* ws_dev, size_t ws_bytes, void* stream);

/* One damped Gauss-Newton (Levenberg-Marquardt) step of the DENSE problem, solved on the device: the reduced camera system
 *     S dc = g_c - W Cd^-1 g_p,   S = Bd - W Cd^-1 W^T,   dp = Cd^-1 (g_p - W^T dc),   Bd / Cd = blocks with diagonals x (1 + lam)
 * by block-Jacobi-preconditioned conjugate gradients; S is applied matrix-free (the two products above), the CG vectors and
 * scalars stay on the device, the host reads the residual norm every fifth iteration.  Inputs: the blocks
 * sfm_ba_dense_sweep returned at (cams, X).  Outputs: dc_dev [ncam x 6], dp_dev [npt x 3] (update = parameters - step);
 * iters_host = CG iterations run; status_host bit 0 = a singular camera block was met (its preconditioner block is the
 * identity), bit 1 = a singular point block.  Synchronises `stream`.  Replaces the reference's
 * scipy.optimize.least_squares call (sfm.py:146), which differentiates the dense problem by finite differences. */
size_t sfm_ba_schur_solve_ws_bytes(int64_t ncam, int64_t npt);
int sfm_ba_schur_solve(const double* cams_dev, int64_t ncam, const double* K_host, const float* X_dev, int64_t npt, int64_t ldx,
                       const double* JtJ_cam_dev, const double* Jtr_cam_dev, const double* JtJ_pt_dev, const double* Jtr_pt_dev,
                       double lam, int fix_first_camera, double cg_tol, int cg_iters, double* dc_dev, double* dp_dev,
                       int32_t* iters_host, int32_t* status_host, void* ws_dev, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Small batched block kernels of the Schur-complement solver (row f-3; the reference hands the
 * problem to SciPy's dense least_squares, sfm.py:146) and cv2.norm of the metric (sfm.py:93,95).
 *   sfm_block_inverse  A_dev [n x k x k] f64 (k = 3 point blocks, 6 camera blocks) -> Ainv_dev
 *   sfm_block_matvec   y_i = A_i x_i;  x_dev, y_dev [n x k]
 *   sfm_norm_l2        cv2.norm(a, b, NORM_L2) over n elements (float32, or float64 with is_f64):
 *                      differences in the inputs' type, squares accumulated in double, fixed
 *                      order; b_dev may be NULL (norm of a).  out_dev: one double (device).
 * ---------------------------------------------------------------------- */
int sfm_block_inverse(const double* A_dev, int64_t n, int k, double* Ainv_dev, void* stream);
/* ... with a device status word: *bad_count_dev = number of blocks whose Gauss-Jordan met a zero or non-finite pivot (their
 * inverse is non-finite); the caller decides (sfm_mvs_amd.ops.block_inverse raises unless told otherwise). */
int sfm_block_inverse_checked(const double* A_dev, int64_t n, int k, double* Ainv_dev, int32_t* bad_count_dev, void* stream);
int sfm_block_matvec(const double* A_dev, const double* x_dev, int64_t n, int k, double* y_dev, void* stream);
size_t sfm_norm_l2_ws_bytes(void);
int sfm_norm_l2(const void* a_dev, const void* b_dev, int64_t n, int is_f64, double* out_dev,
                void* ws_dev, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * A7  cv2.findEssentialMat RANSAC scoring                sfm.py:307
 *
 * For h candidate essential matrices (host-generated by the 5-point solver)
 * score n K-normalised correspondences with the Sampson distance in fp64,
 * cast to float32 and compared `<= thr2` as OpenCV's RANSAC does.
 *   E_dev [h x 9] f64;  x1n_dev, x2n_dev [n x 2] f64
 *   counts_dev [h] int32;  mask_dev (optional) [h x n] uint8
 * ---------------------------------------------------------------------- */
int sfm_score_essential(const double* E_dev, int h, const double* x1n_dev,
                        const double* x2n_dev, int64_t n, float thr2,
                        int32_t* counts_dev, uint8_t* mask_dev, void* stream);

/* cv2.recoverPose cheirality vote (sfm.py:311): for up to 4 candidate poses [R|t]
 * (P_host: h x 12 doubles, row-major 3x4) triangulate every K-normalised pair
 * against [I|0] in fp64 and count points with positive depth < dist_thresh in
 * both views.  mask (optional) [h x n] uint8 is 255/0 like OpenCV's. */
int sfm_recover_pose_score(const double* P_host, int h, const double* x1n_dev,
                           const double* x2n_dev, int64_t n, double dist_thresh, int rows,
                           int32_t* counts_dev, uint8_t* mask_dev, void* stream);

/* PnP-RANSAC scoring for h hypotheses (rvec,tvec) over n 3D-2D pairs:
 * err = ||proj - obs||^2 as float32, inlier iff err <= thr2 (sfm.py:67 defaults:
 * reprojectionError 8.0 → thr2 = 64). */
int sfm_score_pnp(const double* poses_dev, int h, const double* K_host,
                  const float* X_dev, const float* obs_dev, int64_t n, float thr2,
                  int32_t* counts_dev, uint8_t* mask_dev, void* stream);

/* ------------------------------------------------------------------------
 * A7  cv2.findEssentialMat(points1, points2, K, method=RANSAC, prob, threshold)      sfm.py:307
 *     (also isfm.py:80, test.py:240)
 * The whole call: K-normalisation, five-point hypotheses (host) drawn with OpenCV's RNG,
 * Sampson scoring of a chunk of hypotheses per launch (device), OpenCV's sequential
 * best-model / niters bookkeeping.  Returns host scalars, so it synchronises `stream`.
 *   pts0_dev, pts1_dev [n x 2] float32 (pixels);  K_host 9 doubles (row-major 3x3)
 *   E_host      9 doubles (row-major);  90 when n == 5 (OpenCV then returns all the solver's models stacked)
 *   info_host   int32[4]: [0] models written to E_host (0 = no model), [1] inlier count,
 *               [2] RANSAC iterations run, [3] models scored
 *   mask_dev    [n] uint8, 1 = inlier of the returned model (OpenCV's {0,1} mask)
 * ---------------------------------------------------------------------- */
size_t sfm_find_essential_mat_ws_bytes(int64_t n);
int sfm_find_essential_mat(const float* pts0_dev, const float* pts1_dev, int64_t n, const double* K_host,
                           double prob, double threshold, int max_iters, double* E_host, int32_t* info_host,
                           uint8_t* mask_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * A7' cv2.recoverPose(E, points1, points2, K)                                        sfm.py:311
 * decomposeEssentialMat (host) + the cheirality vote of the four (R, t) candidates over all
 * correspondences (device, fp64 DLT per point, distanceThresh = 50 by default) + OpenCV's
 * cascade of >= tests.  rows = 4 (current cv::triangulatePoints system) or 6 (legacy).
 *   R_host 9 doubles, t_host 3 doubles (unit norm), good_host int32[1] = return value of recoverPose
 *   mask_dev (optional) [n] uint8, 255 = point passed the vote (OpenCV's {0,255} mask)
 * ---------------------------------------------------------------------- */
size_t sfm_recover_pose_ws_bytes(int64_t n);
int sfm_recover_pose(const double* E_host, const float* pts0_dev, const float* pts1_dev, int64_t n,
                     const double* K_host, double distance_thresh, int rows, double* R_host, double* t_host,
                     int32_t* good_host, uint8_t* mask_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * A6  cv2.solvePnPRansac(objectPoints, imagePoints, K, distCoeffs = 0)               sfm.py:67
 * (the reference's 5th positional argument lands in OpenCV's `rvec` slot — SURVEY 3.6-1 — so
 * every tunable is the default: iterations 100, reprojectionError 8.0, confidence 0.99,
 * flags ITERATIVE).  EPnP hypotheses on 5-point samples (host), reprojection scoring of a
 * chunk of hypotheses per launch (device), then solvePnP(ITERATIVE) on the inliers: DLT
 * initialisation (host) and Levenberg-Marquardt whose residual / J^T J sweeps run on the
 * device.  n >= 5 (OpenCV's P3P branch for n == 4 is not on this path: SFM_ERR_ARG).
 *   X_dev [n x 3] float32, uv_dev [n x 2] float32;  rvec_host, tvec_host 3 doubles each
 *   info_host   int32[4]: [0] 1 = pose found, [1] inlier count, [2] initialisation of the
 *               refinement (0 DLT, 1 planar object / 2 fewer than 6 inliers: the RANSAC model
 *               is refined instead — OpenCV uses a homography / raises there), [3] LM iterations
 *   inliers_dev [n] int32: indices of the best RANSAC model's inliers, ascending (first info[1])
 * ---------------------------------------------------------------------- */
size_t sfm_solve_pnp_ransac_ws_bytes(int64_t n);
int sfm_solve_pnp_ransac(const float* X_dev, const float* uv_dev, int64_t n, const double* K_host, int iterations,
                         float reproj_error, double confidence, double* rvec_host, double* tvec_host,
                         int32_t* info_host, int32_t* inliers_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* cv2.projectPoints(X, rvec, tvec, K, None) on float64 object points (sfm.py:119-121: the residual of the
 * reference's BundleAdjustment is fp64 end to end).  rvec, tvec, K on the host; X_dev [n x 3], proj_dev [n x 2] f64. */
int sfm_project_points_f64(const double* rvec_host, const double* tvec_host, const double* K_host,
                           const double* X_dev, int64_t n, double* proj_dev, void* stream);

/* ------------------------------------------------------------------------
 * The host-side hypothesis generators behind the three calls above, reachable on their own
 * (pure host code, HOST pointers, no GPU needed): unit-tested against the oracle.
 *   sfm_host_epnp                 EPnP on 4..64 correspondences (solvePnPRansac's minimal solver):
 *                                 K 9 doubles, Xw [n x 3], uv [n x 2] pixels -> R 9, t 3 doubles
 *   sfm_host_p3p                  solvePnP(P3P) on exactly four correspondences (solvePnPRansac's npoints == 4
 *                                 branch): Xw [4 x 3], uv [4 x 2] pixels -> R 9, t 3 doubles, ok_host 1 / 0
 *   sfm_host_five_point           five K-normalised correspondences [5 x 2] each -> up to 10
 *                                 essential matrices (E_host 90 doubles), count_host int32
 *   sfm_host_decompose_essential  E -> R1, R2 (9 doubles each), t (3)
 *   sfm_host_pnp_dlt_init         ITERATIVE's non-planar initialisation: X [n x 3], uv [n x 2]
 *                                 doubles -> rvec, tvec; status 0 ok / 1 planar / 2 fewer than 6 points
 *   sfm_host_rodrigues            cv2.Rodrigues (sfm.py:69,84,119): src 3 (vector) or 9 (matrix)
 *                                 doubles; jac_host (optional, vector input) dR/dr 3 x 9
 * ---------------------------------------------------------------------- */
int sfm_host_epnp(const double* K_host, const double* Xw_host, const double* uv_host, int n,
                  double* R_host, double* t_host);
int sfm_host_p3p(const double* K_host, const double* Xw_host, const double* uv_host, double* R_host, double* t_host,
                 int32_t* ok_host);
int sfm_host_five_point(const double* x1n_host, const double* x2n_host, double* E_host, int32_t* count_host);
int sfm_host_decompose_essential(const double* E_host, double* R1_host, double* R2_host, double* t_host);
int sfm_host_pnp_dlt_init(const double* K_host, const double* X_host, const double* uv_host, int64_t n,
                          double* rvec_host, double* tvec_host, int32_t* status_host);
int sfm_host_rodrigues(const double* src_host, int src_is_matrix, double* dst_host, double* jac_host);

/* ------------------------------------------------------------------------
 * Next row f-1: image preprocessing + SIFT
 *   cv2.pyrDown(img)                                  sfm.py:40
 *   cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)             sfm.py:243-244
 *   cv2.xfeatures2d.SIFT_create().detectAndCompute(gray, None)   sfm.py:246-252
 *
 * sfm_bgr2gray_u8 / sfm_pyrdown_u8: OpenCV's uint8 fixed-point programs
 * ((1868 B + 9617 G + 4899 R + 8192) >> 14; 5x5 binomial, (sum + 128) >> 8,
 * BORDER_REFLECT_101, output ((w+1)/2, (h+1)/2)).  All pointers are device memory.
 *
 * sfm_sift_detect_and_compute: the whole of detectAndCompute (nfeatures = 0), stream
 * ordered, no host round trip: 2x bilinear base image, Gaussian / DoG pyramid,
 * 26-neighbour extrema with sub-pixel refinement, contrast and edge tests,
 * orientation histograms (one keypoint per peak >= 0.8 max), OpenCV's keypoint
 * ordering + duplicate removal, 4x4x8 descriptors (clip 0.2, x512, u8-valued float32).
 *   gray_dev        [h x stride] uint8
 *   max_keypoints   capacity of the outputs and of the internal lists, 64 <= max_keypoints < 2^24
 *   keypoints_dev   [max_keypoints x 8] float32: x, y, size, angle (degrees), response,
 *                   octave (int32 bits, OpenCV packing), class_id (int32 bits, -1), 0
 *   descriptors_dev [max_keypoints x 128] float32 (NULL: detect only)
 *   count_dev       int32[4]: [0] keypoints written (ordered), [1] keypoints before
 *                   duplicate removal, [2] refined extrema, [3] raw extrema (before the
 *                   contrast / edge tests); [1] or [2] > max_keypoints, or [3] >
 *                   8 * max_keypoints, means a capacity was exceeded and the output is truncated
 * Blur kernels longer than 55 taps (a per-layer sigma above 6.8) are rejected with
 * SFM_ERR_ARG; the defaults (3, 0.04, 10, 1.6) need 27.
 * ---------------------------------------------------------------------- */
int sfm_bgr2gray_u8(const uint8_t* bgr_dev, int64_t w, int64_t h, int64_t stride_bytes, uint8_t* gray_dev, void* stream);
int sfm_pyrdown_u8(const uint8_t* src_dev, int64_t w, int64_t h, int channels, uint8_t* dst_dev, void* stream);
size_t sfm_sift_ws_bytes(int64_t w, int64_t h, int n_octave_layers, int64_t max_keypoints);
int sfm_sift_detect_and_compute(const uint8_t* gray_dev, int64_t w, int64_t h, int64_t stride_bytes,
                                int n_octave_layers, double contrast_threshold, double edge_threshold,
                                double sigma, int64_t max_keypoints, float* keypoints_dev,
                                float* descriptors_dev, int32_t* count_dev, void* ws, size_t ws_bytes,
                                void* stream);

/* ------------------------------------------------------------------------
 * Measurement hook (no reference counterpart): when enabled, the library brackets
 * its dominant kernels with hipEvents recorded on the launch stream.
 * sfm_profile_read synchronises those events, returns the summed device time
 * and launch count of one slot, and resets the slot (toggling the switch does not).
 *   slot 0 knn filter (MFMA)   1 knn refine (+ rescans)   2 triangulate
 *        3 dense BA sweep       4 indexed residual sweep    5 Schur products (W^T x, W v)
 *        6 SIFT scale space (all blur + decimate launches of one image = 1 "launch")   7 SIFT descriptors
 * sfm_profile_enable(n) with n > 1: as 1, and the knn filter kernel is launched n times back-to-back
 * inside one event pair (idempotent), so the few microseconds an event pair adds to a single short
 * launch are amortised; sfm_profile_read then reports n launches per bracket.
 * ---------------------------------------------------------------------- */
int sfm_profile_enable(int on);
/* Dev diagnostics: when non-NULL, every knn filter workgroup b writes int64[4] =
 * {start tick, end tick (100 MHz), HW_ID, XCC_ID} at dev_buf[4*b..] and every refine workgroup w
 * int64[16] phase ticks at dev_buf[16384 + 16*w..]; NULL disables. */
int sfm_debug_set_trace(void* dev_buf);
/* Test hook: workgroup `workgroup` of every following knn_split_images_kernel launch starts `microseconds` (<= 100 000) late — the
 * situation of a grid that is not co-resident (tests/test_gpu_knn_q8.py: the repair of pairs quantised in vain must not depend on
 * when a workgroup is dispatched).  workgroup = -1 switches it off. */
int sfm_debug_knn_split_delay(int workgroup, int microseconds);
int sfm_profile_read(int slot, double* total_ms_host, int64_t* launches_host);
/* How often library calls of this process have WAITED for the device so far (cumulative; the RANSAC entry points read the
 * hypothesis scores back chunk by chunk, the Schur solver its convergence scalars).  Diagnostics: bench.py reports the
 * difference over a 57-camera run per registered camera. */
int64_t sfm_host_sync_count(void);
/* Where sfm_solve_pnp_ransac's host time went since the last reset (microseconds, accumulated over calls): out10[0] calls,
 * [1..7] copy-in, EPnP hypotheses (host), scoring + wait, mask / inlier bookkeeping, DLT initialisation, LM sweeps (device +
 * wait), LM host algebra; [8] hypothesis chunks, [9] LM sweeps.  reset != 0 clears the accumulators. */
int sfm_pnp_profile_read(double* out10_host, int reset);
/* sfm_solve_pnp_ransac's Levenberg-Marquardt sweeps are answered by ONE resident workgroup per call (the "sweep server",
 * csrc/ransac.hip) that the host drives through its pinned mailbox instead of a launch + stream synchronisation per sweep.
 * sfm_host_poll_count: spins of the host's waiting loop since load (a measure of time spent waiting on the mailbox, not of API
 * calls; those waits do NOT count in sfm_host_sync_count).  sfm_debug_pnp_sweep_server(0) selects the launch-per-sweep path
 * (A/B measurements, parity tests of one path against the other), (1) the server (default); returns the previous setting. */
int64_t sfm_host_poll_count(void);
int sfm_debug_pnp_sweep_server(int on);

#ifdef __cplusplus
}
#endif
#endif /* SFM_HIP_H */
