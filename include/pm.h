/* pm.h — C ABI of the MI355X-native two-view point matcher (libpm_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of wenxiaoshuai/Points-Matching:
 *   descriptor match  ->  strong-match filter  ->  point gather  ->  robust F  ->  residual report
 * i.e. `Points Matching/main.cpp:42-46, 49-69, 71-79, 89-91, 95-98, 103-123` (cited per entry
 * point below as main.cpp:N).  The reference has no FFI of its own: its boundary is the two
 * OpenCV call sites (main.cpp:46, main.cpp:95-98) plus the glue around them, so every symbol
 * here replaces one of those call sites or glue blocks.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; every function returns a pm_status
 *     (0 = ok, <0 = error) and never throws across the boundary (the reference's OpenCV calls
 *     raise cv::Exception instead).
 *   - caller owns every buffer; the library owns only the context's scratch arena.
 *   - a context is bound to one HIP device + one HIP stream and is NOT thread-safe; distinct
 *     contexts are independent.
 *   - functions without a `_dev` suffix take HOST pointers and block until the result is in
 *     the caller's buffers.  `_dev` variants take DEVICE pointers (hipMalloc'd / torch
 *     tensor .data_ptr()), enqueue on the context's stream and return without synchronising;
 *     call pm_ctx_synchronize() (or synchronise the stream you attached) before reading.
 *   - exact arithmetic (op order, tie rules, RNG) is frozen in docs/SPEC.md.
 */
#ifndef PM_H_
#define PM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_VERSION_MAJOR 0
#define PM_VERSION_MINOR 1

/* ---- status codes ------------------------------------------------------------------------- */
typedef enum pm_status {
    PM_OK            =  0,
    PM_E_INVALID     = -1,  /* bad argument (null pointer, negative size, k out of range ...)   */
    PM_E_TOO_FEW     = -2,  /* fewer than 8 correspondences (cv::findFundamentalMat: count<7 fails) */
    PM_E_NO_MODEL    = -3,  /* every hypothesis in the range was degenerate; F = 0, mask = 0     */
    PM_E_HIP         = -4,  /* a HIP runtime call failed; see pm_last_error()                    */
    PM_E_NOMEM       = -5,
    PM_E_UNSUPPORTED = -6
} pm_status;

/* One match record.  Mirrors cv::DMatch (OpenCV 2.4: {int queryIdx; int trainIdx; int imgIdx;
 * float distance;}) as used at main.cpp:45, :54-55, :65, :76-78, :110, :113.  16 bytes. */
typedef struct pm_match {
    int32_t queryIdx;
    int32_t trainIdx;   /* -1 when fewer than k train rows exist */
    int32_t imgIdx;     /* always 0 (single train image, as in main.cpp:46) */
    float   distance;   /* L2: sqrt of the squared distance; Hamming: bit count as float */
} pm_match;

typedef struct pm_ctx pm_ctx;   /* opaque */

/* ---- context -------------------------------------------------------------------------------
 * Replaces the implicit OpenCV global state behind main.cpp:44-46 / :95-98. */
int  pm_ctx_create(int device, pm_ctx** out);
int  pm_ctx_destroy(pm_ctx* ctx);
/* Attach an externally owned hipStream_t (passed as void*); NULL restores the context's own
 * stream.  Lets a caller time the kernels with events on its own stream. */
int  pm_ctx_set_stream(pm_ctx* ctx, void* hip_stream);
int  pm_ctx_synchronize(pm_ctx* ctx);
/* Per-kernel timing with hipEvents on the context's stream.  enable!=0 starts collecting.
 * pm_ctx_timing_get: mean milliseconds and launch count of the named kernel since the last
 * pm_ctx_timing_reset (synchronises the stream).  Names: "knn_l2_prep", "knn_l2_mfma_f16",
 * "knn_l2_mfma", "knn_l2_mfma_u8", "knn_l2_mfma_f16s", "knn_l2_refine", "knn_l2_exact", "knn_hamming_expand", "knn_hamming_mfma_i8",
 * "knn_hamming_refine", "knn_hamming", "knn_hamming_merge", "filter_gather", "concat_points",
 * "ransac_fused", "ransac_finish", "ransac_solve", "ransac_score", "ransac_select", "ransac_final", "lmeds_solve", "lmeds_median",
 * "lmeds_final", "fm_count", "flann_search". */
int  pm_ctx_timing_enable(pm_ctx* ctx, int enable);
int  pm_ctx_timing_reset(pm_ctx* ctx);
int  pm_ctx_timing_get(pm_ctx* ctx, const char* kernel, double* mean_ms, int* launches);
/* Diagnostics of the MFMA route of pm_bf_knn_l2_f32[_dev]: while enabled, each call records how
 * many queries took the exact re-scan branch of the refinement and whether a non-finite input
 * was seen; pm_ctx_knn_stats returns the values of the last such call (synchronises). */
int  pm_ctx_knn_diag_enable(pm_ctx* ctx, int enable);
int  pm_ctx_knn_stats(pm_ctx* ctx, int* rescans, int* nonfinite);
/* ... and which coarse pass the refinement of the last such call read: 0 = f16 matrix pass on exact integer copies,
 * 1 = f16 matrix pass on rounded copies of general floats, 2 = f32-input matrix pass, 3 = i8 matrix pass on centred
 * u8-valued copies (synchronises). */
int  pm_ctx_knn_route(pm_ctx* ctx, int* route);
/* pm_bf_knn_l2_ratio_dev in its fused form (no record buffer / PM_OPT_FILTER_FUSION = 2) compacts with bounded look-back
 * polls.  *gave_up != 0: a poll of the LAST such call on this context ran out — that call's survivors and count are not to
 * be used (never observed on hardware; the two-launch form has an always-correct fallback instead).  Synchronises. */
int  pm_ctx_filter_fusion_status(pm_ctx* ctx, int* gave_up);
/* Explicit per-context switches for tests and A/B timing (the library reads no environment variables).
 * Every option defaults to 0 = automatic; a value outside an option's range is PM_E_INVALID. */
enum {
    PM_OPT_RANSAC_PATH    = 1,  /* 1: hypothesis-per-lane kernels (solve + score launches), 2: one-launch kernel  */
    PM_OPT_SCORE_OPERANDS = 2,  /* hypothesis-per-lane scorer: 1 LDS-staged points, 2 scalar-operand pair records */
    PM_OPT_HAMMING_ROUTE  = 3,  /* 1: integer-VALU scan, 2: matrix-core route with 64-bit refinement keys          */
    PM_OPT_KNN_F16_WAVES  = 4,  /* f16/i8 coarse kernel: 1 = 8 waves x 32 queries, 2 = 4 waves x 64 queries,
                                   3 (f16 only) = 8 waves x 64 queries in two row groups                         */
    PM_OPT_KNN_STAGING    = 6,  /* f16/i8 coarse kernel, train tiles: 1 = through registers, 2 = LDS-DMA (default)  */
    PM_OPT_KNN_WG_PER_CU  = 7,  /* f16 coarse kernel: train splits sized for 1 (default) or 2 workgroups per CU     */
    PM_OPT_FILTER_FUSION  = 5,  /* pm_bf_knn_l2_ratio_dev: 1 = filter as its own launch, 2 = inside the refinement  */
    PM_OPT_KNN_XCD_TILE   = 8,  /* coarse kernels, workgroup order: 1 = launch order, 2 = one 2-D grid tile per XCD     */
    PM_OPT_KNN_GENERAL_F16 = 9, /* automatic L2 route on general floats: 1 = f32-input matrix pass (enqueued next to the
                                   f16 one), 2 = f16-rounded scaled copies, wider refinement window (default)   */
    PM_OPT_KNN_SEEDED     = 10, /* f16 hint route, row term -||t||^2/2: 1 = a k-chunk of its own (9 chunks, default: faster), 2 =
                                   starts the accumulators from a per-tile LDS array (8 chunks).  1 also sends PM_KNN_HINT_U8
                                   to the f16 pass (the u8 route exists in the seeded form only)                    */
    PM_OPT_KNN_U8_GROUP   = 11, /* u8 route: rows per coarse candidate group, 1 = 4, 2 = 8 (default), 3 = 16                 */
    PM_OPT_KNN_RING       = 12, /* u8 coarse kernel, train tiles: 1 = two LDS buffers (default), 2 = ring of eight with counted
                                   waits and a workgroup barrier per tile, 3 = the ring with split-phase LDS counters
                                   (2, 3: 8-row groups only; measured, not faster: DESIGN.md 2.1); register-operand forms
                                   (8-row groups only): a wave keeps 128 queries as B operands and takes 32-row blocks of
                                   train rows by itself, no tile is shared between waves: 4 = blocks straight from global
                                   memory, 5 = through a private LDS buffer per wave (LDS-DMA), 6 = 5 with one train split
                                   per WAVE (no merge, no barrier; long sweeps only, else form 1 runs)               */
    PM_OPT_KNN_U8_REFINE  = 13, /* u8 route refinement: 1 = canonical f32 kernel (4-row groups only), 2 = integer
                                   re-evaluation on the byte copies, one lane per row (default)                     */
    PM_OPT_KNN_RING_PROLOGUE = 14, /* u8 ring kernel: train tiles requested before the sweep starts, 2 .. 8 (default 2)    */
    PM_OPT_KNN_WIDE       = 15, /* L2 matcher beyond dim % 4 == 0 && dim <= 128 && 16-byte aligned rows: 1 = exact VALU kernel (round 2),
                                   2 = f16 matrix passes on padded copies, up to 256 dimensions (default)             */
    PM_OPT_KNN_PREP_ROWS  = 16, /* u8 route, prep kernel: 1 = 64 rows per workgroup, 2 = 16 rows per workgroup (default)         */
    PM_OPT_RANSAC_FORM    = 17, /* one-launch RANSAC kernel: 1 = correspondences in registers, two teams of four waves (round 2),
                                   2 = correspondences in LDS, one wave per hypothesis, 12 waves (default)           */
    PM_OPT_RANSAC_WG_IDS  = 18, /* one-launch RANSAC kernel: hypothesis ids per workgroup.  0 = automatic (ids spread over ALL
                                   CUs: lowest latency for one run; a pm_batch lane defaults to 32 instead), 1 .. 128 = that
                                   many: fewer, fuller workgroups, so the runs of several streams share the GPU — the fp64
                                   solve costs a wave the same ~21k cycles whether 8 or 64 of its lanes are in use       */
    PM_OPT_HAMMING_REFINE = 19, /* Hamming matrix-core route, refinement: 1 = one wave per query (round 1), 2 = four queries per
                                   wave, one 16-lane row each (default where a query has <= 64 candidate entries); in both
                                   the rare whole-sub-list scans are done by the whole workgroup                       */
    PM_OPT_COUNT_         = 20
};
int  pm_ctx_set_option(pm_ctx* ctx, int option, int value);
int  pm_ctx_get_option(pm_ctx* ctx, int option, int* value);
const char* pm_last_error(void);       /* thread-local text of the last PM_E_HIP / PM_E_* */
const char* pm_status_string(int status);
int  pm_version(void);                 /* major*100 + minor */

/* ---- descriptor matching (replaces main.cpp:46 `matcher.match(imageDesc1, imageDesc2, ...)`,
 *      matcher = BruteForceMatcher<L2<float>> of the commented main.cpp:43, generalised to k-NN)
 *
 * q: nq x dim row-major float (imageDesc1), t: nt x dim row-major float (imageDesc2).
 * out: nq*k records, row i at out[i*k .. i*k+k-1], ascending by (distance bits, trainIdx):
 *      smaller distance first, equal distances -> lower trainIdx first (SPEC S3).
 *      If nt < k the tail of each row has trainIdx = -1, distance = +inf.
 * 1 <= k <= PM_MAX_K.  nq == 0 is ok (no output).  nt == 0 gives all -1 rows.
 * `flags`: 0 = automatic.  For dim%4==0 && dim<=128 && k<=2: MFMA coarse pass + canonical
 *          refinement; the coarse pass is the exact f16-MFMA route when the data are integer-valued
 *          (decided on the device, no host round trip: both coarse kernels are enqueued and the one
 *          that does not apply exits at once) and the f32-MFMA route otherwise.  Anything else:
 *          exact VALU kernel.  All routes produce bit-identical output (tests assert it). */
/* Device pointers of the _dev form: rows are read as 16-byte vectors when dim % 4 == 0 and both base pointers are
 * 16-byte aligned (hipMalloc / torch allocations are); otherwise the call takes the exact scalar-load kernel. */
#define PM_MAX_K 16
#define PM_KNN_FORCE_EXACT  1   /* exact VALU kernel only                                           */
#define PM_KNN_FORCE_F32    2   /* f32-MFMA coarse route only (any finite floats)                   */
#define PM_KNN_HINT_INTEGER 4   /* caller states the descriptors are integer-valued with |x| <= 361 */
                                /* (OpenCV SIFT: 0..255): only the exact f16-MFMA coarse route is   */
                                /* launched.  The claim is verified on the device; a wrong hint     */
                                /* costs time (exact re-scan), never correctness.                   */
#define PM_KNN_HINT_U8      8   /* caller states the descriptors are integers in [0, 255] (OpenCV SIFT):  */
                                /* ranked on the i8 matrix cores (x - 128, 4 k-chunks at D = 128, twice   */
                                /* the f16 rate).  Verified on the device like PM_KNN_HINT_INTEGER.        */
#define PM_KNN_HINT_UNIT_NORM 16 /* caller states that every TRAIN row has ||t||^2 <= 1 + 2^-10 (SURF, L2-normalised      */
                                /* descriptors: what main.cpp:37-40 itself produces): general floats, ranked on the      */
                                /* f16 matrix pass with ONE prep launch instead of two (the train scale needs no norm    */
                                /* maximum).  Verified on the device; a wrong hint costs time (exact re-scan).           */
int pm_bf_knn_l2_f32(pm_ctx* ctx, const float* q, int nq, const float* t, int nt,
                     int dim, int k, int flags, pm_match* out);
int pm_bf_knn_l2_f32_dev(pm_ctx* ctx, const float* d_q, int nq, const float* d_t, int nt,
                         int dim, int k, int flags, pm_match* d_out);

/* The same matcher on u8 DESCRIPTOR ROWS (nq x dim / nt x dim bytes, 4-byte aligned device pointers in the _dev forms):
 * what a SIFT extractor holds before it widens to float, a quarter of the bytes on the host link (BASELINE config 5).
 * Output = pm_bf_knn_l2_f32 on the same values converted to float, bit for bit (distances are the canonical f32 ones).
 * dim % 4 == 0, dim <= 128, k <= 2: the u8 route (i8 matrix cores on x - 128, integer refinement); anything else is
 * widened on the device and takes the f32 matcher.  pm_bf_knn_l2_u8_ratio_dev: + ratio test + compaction + gather, as
 * pm_bf_knn_l2_ratio_dev (the record buffer d_knn is required). */
int pm_bf_knn_l2_u8(pm_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int dim, int k, pm_match* out);
int pm_bf_knn_l2_u8_dev(pm_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int dim, int k, pm_match* d_out);
int pm_bf_knn_l2_u8_ratio_dev(pm_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int dim, float ratio,
                              const float* d_kp1_xy, const float* d_kp2_xy, pm_match* d_knn, pm_match* d_good,
                              float* d_xy1, float* d_xy2, int32_t* d_n_good);

/* main.cpp:46 + :49-69 (ratio form) + :77-78 + :89-91 in ONE call for batches that stay in HBM: 2-NN, ratio test
 * (d1 < ratio * d2, float multiply, strict), stable compaction in query order and keypoint gather — the outputs of
 * pm_bf_knn_l2_f32_dev(k = 2) followed by pm_filter_ratio_gather_dev, bit for bit.  d_knn (nq x 2 records) may be
 * NULL on the MFMA routes: the filter then rides the refinement launch and no record is written (shapes that take
 * the exact kernel — dim % 4 != 0, dim > 128, unaligned rows — need the buffer).  With d_knn the filter is its own
 * launch, which measured faster, except with PM_KNN_HINT_U8 up to 768 queries, where the fused launch
 * is 1.6-2.9 us shorter and is taken; PM_OPT_FILTER_FUSION pins either form.
 * Graph capture: the matcher and the compaction calls (pm_bf_knn_l2_*, pm_filter_*_gather_dev) pass a per-call epoch
 * as a kernel argument and therefore return PM_E_UNSUPPORTED on a stream that is capturing (a replay would reuse the
 * epoch); enqueue them directly — a replay measured slower than direct launches anyway. */
int pm_bf_knn_l2_ratio_dev(pm_ctx* ctx, const float* d_q, int nq, const float* d_t, int nt, int dim, int flags,
                           float ratio, const float* d_kp1_xy, const float* d_kp2_xy, pm_match* d_knn,
                           pm_match* d_good, float* d_xy1, float* d_xy2, int32_t* d_n_good);

/* ---- approximate matcher compatible with `FlannBasedMatcher matcher;` (main.cpp:44, the reference's ACTIVE matcher,
 * called at main.cpp:46) — SURVEY.md 8f-4, docs/SPEC.md S17.  cv::flann defaults: 4 randomised kd-trees
 * (KDTreeIndexParams(4)), 32 checks, eps 0, sorted results (SearchParams(32, 0, true)).  The forest is built on the
 * host from the train descriptors (every random draw comes from a counter-based stream keyed by `seed`, not from C
 * rand(): reproducible everywhere), uploaded once, and searched on the GPU (one lane per query, best-bin-first over
 * all trees with one heap).  Output records like pm_bf_knn_l2_f32 (distance = sqrt of the canonical squared L2), rows
 * ascending; with fewer than k examined points the tail has trainIdx = -1.  1 <= k <= 4.
 * Approximate by design: tests report recall against the exact matcher; the exact matcher is also faster at the
 * reference's sizes, so this path exists for behavioural compatibility (pm_cli --matcher flann). */
typedef struct pm_flann_params {
    int32_t  trees;     /* <= 0: 4 */
    int32_t  checks;    /* <= 0: 32 */
    uint64_t seed;
} pm_flann_params;
typedef struct pm_flann_index pm_flann_index;   /* opaque; bound to the context's device */
int pm_flann_build(pm_ctx* ctx, const float* train, int nt, int dim, const pm_flann_params* prm, pm_flann_index** out);
int pm_flann_destroy(pm_flann_index* ix);
int pm_flann_knn_l2_f32(pm_ctx* ctx, pm_flann_index* ix, const float* q, int nq, int k, pm_match* out);
int pm_flann_knn_l2_f32_dev(pm_ctx* ctx, pm_flann_index* ix, const float* d_q, int nq, int k, pm_match* d_out);
/* The forest as built (tests, inspection): *n_nodes records of 16 bytes {int32 child1, child2 (-1: leaf), int32 cut
 * dimension | point id of a leaf, float cut value}; roots[t] = root record of tree t.  nodes may be NULL (size query). */
int pm_flann_export(const pm_flann_index* ix, int32_t* n_nodes, int32_t* roots, void* nodes, int32_t cap_nodes);

/* Binary descriptors (ORB-256 = 32 bytes/row): Hamming distance, popcount of XOR.
 * `bytes` must be a multiple of 4.  Replaces main.cpp:46 for BASELINE config C4.  32-byte
 * descriptors with k <= 2 (16-byte-aligned device buffers) run on the matrix cores (+-1 expansion on
 * i8 MFMA + popcount refinement), everything else on the integer VALU scan; same output. */
int pm_bf_knn_hamming_u8(pm_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                         int bytes, int k, pm_match* out);
int pm_bf_knn_hamming_u8_dev(pm_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                             int bytes, int k, pm_match* d_out);

/* ---- strong-match filters (the slot of main.cpp:49-69) -------------------------------------
 * Host-side, O(n).  `out` must hold n (resp. nq) records; survivors keep query order. */

/* Literal main.cpp:49-69: minMatch starts at 1, maxMatch at 0 (main.cpp:49-50); keep i iff
 * (double)distance < minMatch + (maxMatch - minMatch) / 2  (strict, in double, main.cpp:65).
 * min_out/max_out receive the two values the reference prints at main.cpp:58-59. */
int pm_filter_midpoint(const pm_match* m, int n, double* min_out, double* max_out,
                       pm_match* out, int* n_out);
/* Ratio test on k-NN rows (k >= 2): keep row i iff knn[i*k+1].trainIdx >= 0 and
 * knn[i*k].distance < ratio * knn[i*k+1].distance (float multiply, strict).  Emits knn[i*k]. */
int pm_filter_ratio(const pm_match* knn, int nq, int k, float ratio, pm_match* out, int* n_out);
/* Device-resident fusion of pm_filter_ratio + pm_match_indices + pm_gather_points (main.cpp:49-69,
 * :77-78, :89-91) for batches that stay in HBM between the matcher and RANSAC: stable compaction
 * in query order.  d_good: nq records; d_xy1/d_xy2: nq x 2 floats (may be NULL together with the
 * keypoint arrays when only the match list is wanted); *d_n_good: survivor count (device int). */
int pm_filter_ratio_gather_dev(pm_ctx* ctx, const pm_match* d_knn, int nq, int k, float ratio,
                               const float* d_kp1_xy, const float* d_kp2_xy, pm_match* d_good,
                               float* d_xy1, float* d_xy2, int32_t* d_n_good);
/* The reference's own strong-match rule (main.cpp:49-69, what pm_filter_midpoint does on the host)
 * in the same device-resident form: record i of d_m is d_m[i*k] (k = 1 for a 1-NN match list);
 * d_minmax (may be NULL) receives minMatch / maxMatch as two doubles (main.cpp:58-59). */
int pm_filter_midpoint_gather_dev(pm_ctx* ctx, const pm_match* d_m, int n, int k,
                                  const float* d_kp1_xy, const float* d_kp2_xy, pm_match* d_good,
                                  float* d_xy1, float* d_xy2, int32_t* d_n_good, double* d_minmax);
/* Multi-GPU glue: concatenates `parts` padded blocks of `stride` points (d_counts[p] valid in
 * block p), e.g. the all-gathered per-rank survivors of a query-row-sharded matcher, into one
 * contiguous correspondence array in part order; *d_n_total = sum of counts. */
int pm_concat_points_dev(pm_ctx* ctx, const float* d_xy1_parts, const float* d_xy2_parts,
                         const int32_t* d_counts, int parts, int stride, float* d_xy1,
                         float* d_xy2, int32_t* d_n_total);

/* ---- match list + gather (main.cpp:71-79, :89-91) ------------------------------------------
 * pm_match_indices: pointIndexes1/2 of main.cpp:77-78.
 * pm_gather_points: KeyPoint::convert(keyPoint, selPoints, pointIndexes) of main.cpp:90-91;
 *   kp_xy is the keypoints' .pt as interleaved (x,y) floats; out_xy[i] = kp_xy[idx[i]].
 *   Returns PM_E_INVALID if an index is outside [0, n_kp). */
int pm_match_indices(const pm_match* m, int n, int32_t* query_idx, int32_t* train_idx);
int pm_gather_points(const float* kp_xy, int n_kp, const int32_t* idx, int n, float* out_xy);
/* Formats the stdout block of main.cpp:73-76 into buf (NUL-terminated, truncated at cap);
 * returns the number of bytes that the full text needs (like snprintf). */
long pm_format_match_list(const pm_match* m, int n, char* buf, size_t cap);

/* ---- robust fundamental matrix (replaces main.cpp:95-98 cv::findFundamentalMat) ------------ */
enum { PM_ERR_SAMPSON = 0, PM_ERR_SYM_EPIPOLAR = 1 };

typedef struct pm_ransac_params {
    int64_t  hyp_begin;     /* hypothesis ids [hyp_begin, hyp_end) are evaluated; ids < 2^32  */
    int64_t  hyp_end;       /* single GPU: 0 .. iters.  Multi-GPU: this rank's shard.          */
    uint64_t seed;          /* counter-based sampler key (SPEC S6): sample(h) depends only on  */
                            /* (seed, h, n), never on the shard or the device                   */
    float    thresh_px;     /* inlier threshold tau in pixels; test is num^2 <= tau^2 * den     */
    int32_t  error_kind;    /* PM_ERR_SAMPSON | PM_ERR_SYM_EPIPOLAR                             */
} pm_ransac_params;

/* xy1/xy2: n x 2 interleaved float pixel coordinates in match order (selPoints1/2 of
 * main.cpp:89-91).  Every hypothesis h in the range: sample 8 correspondences, Hartley-
 * normalised 8-point solve with rank-2 enforcement (fp64), score ALL n correspondences (fp32),
 * count inliers.  Winner: most inliers, ties -> lowest h.
 *   best_key : (inliers << 32) | (0xFFFFFFFF - h); 0 = no valid model in the range.  This is
 *              the value a multi-GPU caller max-reduces (one 8-byte all-reduce).
 *   F        : 3x3 row-major, x2^T F x1 = 0, unit Frobenius norm, F[8] >= 0 (may be NULL)
 *   mask     : n bytes 0/1 (may be NULL);  n_inliers: may be NULL.
 * n < 8 -> PM_E_TOO_FEW.  All-degenerate range -> PM_E_NO_MODEL with F = 0, mask = 0. */
int pm_ransac_fundamental(pm_ctx* ctx, const float* xy1, const float* xy2, int n,
                          const pm_ransac_params* p, double F[9], uint8_t* mask,
                          int* n_inliers, uint64_t* best_key);
/* Device-resident form used by the bench and the multi-GPU path: scores the shard and leaves
 * the shard's best key in *d_best_key (device uint64).  No model is materialised. */
int pm_ransac_score_dev(pm_ctx* ctx, const float* d_xy1, const float* d_xy2, int n,
                        const pm_ransac_params* p, uint64_t* d_best_key);
/* Same, with the correspondence count read on the device: n = min(*d_n, n_max).  Lets a batch
 * flow matcher -> pm_filter_ratio_gather_dev -> RANSAC without a host round trip.  *d_n < 8
 * leaves *d_best_key = 0. */
int pm_ransac_score_devn(pm_ctx* ctx, const float* d_xy1, const float* d_xy2, int n_max,
                         const int32_t* d_n, const pm_ransac_params* p, uint64_t* d_best_key);
/* Whole single-shard run on the device: solve + score [hyp_begin, hyp_end), pick the winner and
 * publish its key, F (9 doubles), mask (n_max bytes, zero beyond n) and inlier count — the
 * device-resident equivalent of pm_ransac_fundamental.  d_n may be NULL (n = n_max). */
int pm_ransac_run_dev(pm_ctx* ctx, const float* d_xy1, const float* d_xy2, int n_max,
                      const int32_t* d_n, const pm_ransac_params* p, uint64_t* d_best_key,
                      double* d_F, uint8_t* d_mask, int32_t* d_n_inliers);
/* Device-resident finalisation: F (9 doubles), mask (n_max bytes, zero beyond n) and inlier
 * count of the hypothesis encoded in *d_key (e.g. the all-reduced winner).  d_n may be NULL
 * (then n = n_max).  A zero key / n < 8 gives F = 0, mask = 0, count 0. */
int pm_ransac_model_from_key_dev(pm_ctx* ctx, const float* d_xy1, const float* d_xy2, int n_max,
                                 const int32_t* d_n, const pm_ransac_params* p,
                                 const uint64_t* d_key, double* d_F, uint8_t* d_mask,
                                 int32_t* d_n_inliers);
/* Re-derives F + mask of ONE hypothesis id (every rank calls this with the reduced winner:
 * no model broadcast is needed).  hyp is the id, not the key. */
int pm_ransac_model_from_hyp(pm_ctx* ctx, const float* xy1, const float* xy2, int n,
                             const pm_ransac_params* p, int64_t hyp, double F[9],
                             uint8_t* mask, int* n_inliers);
/* ---- one-launch and sharded forms over a correspondence VIEW --------------------------------------
 * A view names the correspondences without copying them: `parts` blocks of up to `cap` points, part p
 * starting at xy1 + p*pitch_xy / xy2 + p*pitch_xy (floats) and holding counts[p*pitch_cnt] points
 * (device-side, clamped to [0, cap]; counts == NULL: every part is full).  Correspondence i of the run is
 * the i-th point in part order.  parts = 1 is a plain array, optionally with a device-side count.  This is
 * how the all-gathered survivor blocks of a query-row-sharded matcher feed RANSAC on every
 * rank without a concatenation pass. */
#define PM_MAX_PARTS 64
typedef struct pm_points_view {
    const float*   xy1;
    const float*   xy2;
    const int32_t* counts;
    int32_t parts;
    int32_t cap;
    int64_t pitch_xy;
    int32_t pitch_cnt;
    int32_t reserved;
} pm_points_view;
/* What a shard contributes to the multi-GPU exchange: its best key and the fp64 model behind it.  The
 * global winner is the record with the largest key — an arg-max all-reduce, carried as one 80-byte
 * all-gather (RCCL has no user-defined reduction); no rank re-solves anything. */
typedef struct pm_ransac_record {
    uint64_t key;       /* pm_ransac_key of the shard's winner, 0 if it has no valid model */
    double   F[9];
} pm_ransac_record;
/* Sharded run, ONE launch: sample + solve + score ids [hyp_begin, hyp_end) over the view, write the
 * shard's record to *d_record (device). */
int pm_ransac_shard_parts_dev(pm_ctx* ctx, const pm_points_view* view, const pm_ransac_params* p,
                              pm_ransac_record* d_record);
/* After the exchange: winner among d_records[0..n_records), its F (9 doubles, may be NULL), inlier mask
 * over the view (d_mask[0..mask_len), zero beyond the correspondence count), inlier count and (optional)
 * the correspondence count itself.  Only thresh_px and error_kind of *p are used. */
int pm_ransac_finish_parts_dev(pm_ctx* ctx, const pm_points_view* view, const pm_ransac_params* p,
                               const pm_ransac_record* d_records, int n_records, uint64_t* d_key, double* d_F,
                               uint8_t* d_mask, int mask_len, int32_t* d_n_inliers, int32_t* d_n_total);
/* key helpers */
static inline uint64_t pm_ransac_key(uint32_t inliers, uint32_t hyp) {
    return ((uint64_t)inliers << 32) | (uint64_t)(0xFFFFFFFFu - hyp);
}
static inline uint32_t pm_ransac_key_hyp(uint64_t key)     { return 0xFFFFFFFFu - (uint32_t)key; }
static inline uint32_t pm_ransac_key_inliers(uint64_t key) { return (uint32_t)(key >> 32); }

/* ---- 7-point + LMedS: what the reference's call literally selects -----------------
 * cv::findFundamentalMat(..., CV_FM_7POINT) with more than 7 points runs OpenCV 2.4's least-median
 * loop over 7-point minimal solves (main.cpp:95-98) [recalled].  Arithmetic: docs/SPEC.md S13-S15.
 * Hypothesis h in [hyp_begin, hyp_end) samples 7 correspondences and yields up to three models
 * (ids 3h, 3h+1, 3h+2); the model with the smallest median symmetric-epipolar residual wins (ties
 * -> lowest id); inliers are the correspondences within the robust sigma derived from that median.
 * F: row-major, x2^T F x1 = 0, unit Frobenius norm, F[8] >= 0 (pm_f_scale_f33 gives OpenCV's
 * F[8] = 1).  n <= 32768.  pm_lmeds_default_iters: OpenCV's iteration count
 * round(log(1 - confidence) / log(1 - (1 - outlier_ratio)^7)) (300 for 0.99 / 0.45). */
typedef struct pm_lmeds_params {
    int64_t  hyp_begin, hyp_end;
    uint64_t seed;
} pm_lmeds_params;
int pm_lmeds_fundamental(pm_ctx* ctx, const float* xy1, const float* xy2, int n,
                         const pm_lmeds_params* p, double F[9], uint8_t* mask, int* n_inliers,
                         int64_t* best_model, double* median);
int pm_lmeds_fundamental_dev(pm_ctx* ctx, const float* d_xy1, const float* d_xy2, int n,
                             const pm_lmeds_params* p, double* d_F, uint8_t* d_mask,
                             int32_t* d_n_inliers, int64_t* d_best_model, double* d_median);
int pm_lmeds_default_iters(double confidence, double outlier_ratio);
/* The adaptive-iteration RANSAC variant of the same family (OpenCV 2.4 CV_FM_RANSAC [recalled];
 * docs/SPEC.md S16): 7-point models, a correspondence is an inlier when its symmetric-epipolar
 * residual is <= thresh_px^2; hypotheses are visited in id order, a model with more inliers than
 * any before becomes the best and shrinks the iteration budget to
 * log(1 - confidence) / log(1 - (inliers/n)^7) (never above max_iters; OpenCV's defaults: 2000, 0.99,
 * 3 px).  The result equals OpenCV's sequential loop on these models; the device solves and counts
 * ahead in batches.  *iters_run = hypotheses visited. */
typedef struct pm_adaptive_params {
    int64_t  max_iters;
    double   confidence;
    float    thresh_px;
    int32_t  reserved;
    uint64_t seed;
} pm_adaptive_params;
int pm_ransac7_adaptive(pm_ctx* ctx, const float* xy1, const float* xy2, int n,
                        const pm_adaptive_params* p, double F[9], uint8_t* mask, int* n_inliers,
                        int64_t* best_model, int* iters_run);

/* ---- batch of independent image pairs (BASELINE config C5) ----------------------------------
 * One pass of main.cpp:46 -> :49-69 (ratio form) -> :89-91 -> :95-98 per pair, streamed: the batch
 * owns `n_lanes` contexts (stream + scratch + device buffers each); pair j runs on lane
 * j % n_lanes, so H2D of the next pair, the kernels of this one and D2H of the previous one
 * overlap.  Descriptors are f32 rows (L2 matcher, k = 2, `knn_flags` as pm_bf_knn_l2_f32);
 * every pair uses the same ratio and RANSAC parameters.  Host pointers in the jobs should be
 * page-locked (pm_host_register, or any hipHostMalloc'd / torch pinned buffer): pageable memory
 * works but serialises the copies.  Blocking: returns when every result is in `results`
 * (and, when non-NULL, `good`: n_jobs x max_n1 records, first n_good valid per pair; `masks`:
 * n_jobs x max_n1 bytes, inlier flag per surviving match).  results[j].status is PM_OK,
 * PM_E_TOO_FEW (fewer than 8 survivors) or PM_E_NO_MODEL. */
typedef struct pm_pair_job {
    const float* desc1;   /* n1 x dim, image-1 descriptors (query side of main.cpp:46) */
    const float* desc2;   /* n2 x dim */
    const float* kp1_xy;  /* n1 x 2 keypoint pixel coordinates */
    const float* kp2_xy;  /* n2 x 2 */
    int32_t n1, n2;
} pm_pair_job;
typedef struct pm_pair_result {
    double   F[9];
    uint64_t best_key;    /* pm_ransac_key of the winner, 0 if none */
    int32_t  n_good;      /* survivors of the ratio test = correspondences given to RANSAC */
    int32_t  n_inliers;
    int32_t  status;
    int32_t  reserved;
} pm_pair_result;
typedef struct pm_batch pm_batch;   /* opaque */
int pm_batch_create(int device, int n_lanes, int max_n1, int max_n2, int dim, pm_batch** out);
int pm_batch_destroy(pm_batch* b);
int pm_batch_set_option(pm_batch* b, int option, int value);     /* pm_ctx_set_option on every lane's context */
/* Descriptor rows of the jobs: 0 = float32 (default), 1 = uint8 (desc1 / desc2 then point at n x dim BYTES; matched by
 * pm_bf_knn_l2_u8_ratio_dev: same results as the float32 rows of the same values, a quarter of the bytes on the link).
 * One-block jobs: when a job's four arrays lie in ONE host allocation in the order desc1 | desc2 | kp1_xy | kp2_xy
 * (desc2 16-byte, keypoints 8-byte aligned relative to desc1, at most 768 bytes of padding in all), the pair is sent
 * in one copy instead of four — detected per job, nothing to declare. */
int pm_batch_set_desc_type(pm_batch* b, int desc_u8);
/* Host threads that enqueue the pairs of one pm_batch_run call: 0 = automatic (two when the batch has >= 4 lanes: each
 * thread owns half of the lanes and every second job), 1 = the calling thread only, 2.  Results are in job order either way. */
int pm_batch_set_host_threads(pm_batch* b, int n);
int pm_batch_run(pm_batch* b, const pm_pair_job* jobs, int n_jobs, float ratio, int knn_flags,
                 const pm_ransac_params* p, pm_pair_result* results, pm_match* good, uint8_t* masks);
/* Page-lock / release a caller-owned host buffer (hipHostRegister) so the batch copies overlap. */
int pm_host_register(void* ptr, size_t bytes);
int pm_host_unregister(void* ptr);

/* ---- the path over the GPUs of one node (SURVEY.md 8b `pm_ransac_reduce`, 8e) ----------------------------
 * One context per (device, lane), one RCCL communicator set per lane (ncclCommInitAll over xGMI), ONE HOST THREAD PER
 * DEVICE that enqueues that device's launches and collectives; RCCL is bound at run time, so single-GPU users never
 * load it.  `devices`: HIP ordinals (NULL: 0 .. n_dev-1).
 *   pm_mgpu_ransac_fundamental  main.cpp:95-98 with the hypothesis ids cut into n_dev contiguous ranges over
 *       replicated correspondences; the exchange is ONE all-gather of the 80-byte pm_ransac_record per device
 *       (arg-max all-reduce with its payload); every device finishes from the winning record, device 0's answer
 *       is returned.  Same bits as pm_ransac_fundamental for any n_dev (tests assert it).
 *   pm_mgpu_match_ransac        main.cpp:46 -> :49-69 (ratio form) -> :89-91 -> :95-98 (BASELINE config C4): query
 *       rows cut into n_dev contiguous blocks (train set replicated), all-gather #1 of the survivor blocks, RANSAC
 *       as above over the gathered view, all-gather #2 of the records.  desc1/desc2: n x dim float32 rows
 *       (binary == 0, L2, knn_flags as pm_bf_knn_l2_f32) or n x dim bytes (binary != 0, Hamming).  Outputs: `good`
 *       (n1 records, first *n_good valid, query order, queryIdx = row of desc1), F, mask (per good match), counts.
 *       PM_E_TOO_FEW / PM_E_NO_MODEL as pm_ransac_fundamental (good / *n_good are valid either way).  Host inputs go
 *       through ONE pinned staging copy that all devices' copy engines read concurrently.
 * Streamed form (round 3) — what a caller with many image pairs against one train image uses:
 *   pm_mgpu_set_lanes           1 .. 4 lanes per device (stream + communicator + buffers each); pair j runs on lane
 *                               j mod L, so pair j+1's matcher overlaps pair j's two all-gathers.  Default 1.
 *   pm_mgpu_set_train[_dev]     the replicated train side (main.cpp:36-40, image 2) made RESIDENT on every device: uploaded
 *                               once from host memory, or adopted from per-device device pointers the caller keeps alive.
 *   pm_mgpu_submit_dev          one image pair from DEVICE pointers: d_desc1[g] / d_kp1_xy[g] = device g's block of
 *                               rows[g] query rows (blocks in device order = query order; equal blocks, the last possibly
 *                               short, when match records with global queryIdx are wanted).  Returns at once with a ticket.
 *                               A lane holds one pair's results until they are collected: at most `lanes` pairs are in flight.
 *   pm_mgpu_collect             blocks for one ticket (any order): result as pm_pair_result (status
 *                               PM_OK / PM_E_TOO_FEW / PM_E_NO_MODEL), optional match records (sum of rows) and mask.
 *   pm_mgpu_allgather_latency   the collective by itself: microseconds per all-gather of bytes_per_device.
 *   pm_mgpu_batch_run           BASELINE config C5 behind the ABI: pm_batch_run with pair p on device p mod n_dev, one host
 *                               thread per device, results in job order (arguments as pm_batch_create + pm_batch_run). */
typedef struct pm_mgpu pm_mgpu;   /* opaque */
int pm_mgpu_create(int n_dev, const int* devices, pm_mgpu** out);
int pm_mgpu_destroy(pm_mgpu* mg);
int pm_mgpu_size(const pm_mgpu* mg);
pm_ctx* pm_mgpu_ctx(pm_mgpu* mg, int i);        /* device i's lane-0 context (options, timing); owned by mg */
pm_ctx* pm_mgpu_lane_ctx(pm_mgpu* mg, int i, int lane);
int pm_mgpu_ransac_fundamental(pm_mgpu* mg, const float* xy1, const float* xy2, int n, const pm_ransac_params* p,
                               double F[9], uint8_t* mask, int* n_inliers, uint64_t* best_key);
int pm_mgpu_match_ransac(pm_mgpu* mg, const void* desc1, int n1, const void* desc2, int n2, int dim, int binary,
                         const float* kp1_xy, const float* kp2_xy, float ratio, int knn_flags,
                         const pm_ransac_params* p, pm_match* good, int* n_good, double F[9], uint8_t* mask,
                         int* n_inliers, uint64_t* best_key);
int pm_mgpu_set_lanes(pm_mgpu* mg, int n_lanes);
int pm_mgpu_set_train(pm_mgpu* mg, const void* desc2, int n2, int dim, int binary, const float* kp2_xy);
int pm_mgpu_set_train_dev(pm_mgpu* mg, const void* const* d_desc2, int n2, int dim, int binary, const float* const* d_kp2_xy);
int pm_mgpu_submit_dev(pm_mgpu* mg, const void* const* d_desc1, const int32_t* rows, const float* const* d_kp1_xy,
                       float ratio, int knn_flags, const pm_ransac_params* p, int* ticket);
int pm_mgpu_collect(pm_mgpu* mg, int ticket, pm_pair_result* result, pm_match* good, uint8_t* mask);
int pm_mgpu_allgather_latency(pm_mgpu* mg, int bytes_per_device, int reps, double* us_per_collective);
int pm_mgpu_batch_run(pm_mgpu* mg, int n_lanes, int max_n1, int max_n2, int dim, const pm_pair_job* jobs, int n_jobs,
                      float ratio, int knn_flags, const pm_ransac_params* p, pm_pair_result* results, pm_match* good,
                      uint8_t* masks);
int pm_mgpu_batch_set_option(pm_mgpu* mg, int option, int value);   /* pm_ctx_set_option on every context of mg */

/* ---- residual report (main.cpp:103-123) -----------------------------------------------------
 * r[i] = [xa ya 1] * F * [xb yb 1]^T in fp64.  transposed != 0 reproduces the reference
 * literally ((xa,ya) = image-1 point, (xb,yb) = image-2 point: x1^T F x2, main.cpp:110-117);
 * transposed == 0 evaluates x2^T F x1.  mean_abs = sum|r| / n (main.cpp:120,123).
 * pm_f_scale_f33: rescales F so F[8] == 1 when |F[8]| > DBL_EPSILON (OpenCV's output scale,
 * which the magnitudes printed at main.cpp:119 depend on). */
int pm_epipolar_residuals(const float* xy1, const float* xy2, int n, const double F[9],
                          int transposed, double* r, double* mean_abs);
int pm_f_scale_f33(double F[9]);

/* ---- epipolar lines (main.cpp:127-142) ----------------------------------------
 * cv::computeCorrespondEpilines(pts, which_image, F, lines): l = F*x (which_image==1) or
 * F^T*x (==2), scaled so a^2+b^2 = 1; lines: n x 3 floats.  pm_epiline_endpoints: the two
 * cv::Point arguments of main.cpp:138-140 for an image `cols` wide (float->int truncation). */
int pm_epilines(const float* xy, int n, int which_image, const double F[9], float* lines);
int pm_epiline_endpoints(const float* lines, int n, int cols, int32_t* xyxy);

#ifdef __cplusplus
}
#endif
#endif /* PM_H_ */
