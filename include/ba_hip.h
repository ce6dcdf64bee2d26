/*
 * ba_hip.h -- C ABI of the MI355X (gfx950) bundle-adjustment engine.
 *
 * Drop-in boundary for the global / final bundle adjustment of the reference
 * project (AhaanB29/Structure_from_Motion, src/BundleAdjustment.py).  The
 * reference has no FFI: its operator API is the Python module
 * `BundleAdjustment` (7 public functions).  The Python shim
 * `structure_from_motion_amd/BundleAdjustment.py` keeps those signatures and
 * binds the entry points below with ctypes (INTEGRATION.md shows the stub).
 * Every entry point is plain C: pointers + sizes, no exceptions, no torch types;
 * the return value is 0 (BA_OK) or a negative BA_ERR_* code, and
 * ba_last_error() returns the message of the last failure on the calling thread.
 *
 * Data layout (structure of arrays, fp64 / int32) -- the reference's parameter
 * vector and observation list (BundleAdjustment.py:46-65, 129-146, 243-272):
 *   cam_pose  [n_cams*6]   rvec(3), t(3) per camera, camera order = list(cameras.keys()) (:30)
 *   cam_intr  [n_cams*4]   fx, fy, cx, cy: the entries of K that cv2.projectPoints reads (:136,:142)
 *   cam_fixed [n_cams]     1 = gauge-anchored camera, kept constant (:34-38, :53)
 *   points    [n_points*3] xyz of EVERY point, observed or not (:62-63)
 *   obs_cam   [n_obs]      camera index of each observation (position in the camera order)
 *   obs_pt    [n_obs]      point index (:131)
 *   obs_xy    [n_obs*2]    observed pixel (:132)
 * Residuals are interleaved (du, dv) in observation order (:145-152).
 */
#ifndef BA_HIP_H
#define BA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BA_OK 0
#define BA_ERR_INVALID (-1)     /* bad argument (null pointer, index out of range, ...) */
#define BA_ERR_HIP (-2)         /* a HIP runtime call failed / no device */
#define BA_ERR_UNSUPPORTED (-3) /* e.g. more than 2e9 observations per handle */
#define BA_ERR_NOMEM (-4)
#define BA_ERR_NUMERIC (-5)     /* non-finite initial cost */
#define BA_ERR_COMM (-6)        /* the all-reduce callback reported a failure */

#define BA_MAX_TRACK 256 /* longest track on the fused chunk path; longer tracks ("heavy points", any length) are linearised
                            by a workgroup each and their pairs accumulated from global per-block lists (csrc/ba_rows.h) */

/* loss kinds = what scipy.optimize.least_squares accepts (scipy _lsq/least_squares.py:208-209);
 * the reference only uses 'huber' (BundleAdjustment.py:332, :367). */
enum { BA_LOSS_LINEAR = 0, BA_LOSS_SOFT_L1 = 1, BA_LOSS_HUBER = 2, BA_LOSS_CAUCHY = 3, BA_LOSS_ARCTAN = 4 };
enum { BA_SOLVER_AUTO = 0, BA_SOLVER_PCG = 1, BA_SOLVER_DENSE_CHOLESKY = 2, BA_SOLVER_BANDED_CHOLESKY = 3,
       BA_SOLVER_CYCLIC_REDUCTION = 4 };
enum { BA_MEM_HOST = 0, BA_MEM_DEVICE = 1 };
enum { BA_REDUCE_SUM = 0, BA_REDUCE_MAX = 1, BA_BROADCAST0 = 2 };

/* The problem: replaces the (cameras, points_3d, observations, visited_ids) arguments of
 * global_bundle_adjustment (BundleAdjustment.py:7-9) after marshalling to arrays. */
typedef struct BaProblem {
    int32_t n_cams;
    int32_t n_points;
    int64_t n_obs;
    double *cam_pose;         /* in/out */
    const double *cam_intr;
    const uint8_t *cam_fixed;
    double *points;           /* in/out */
    const int32_t *obs_cam;
    const int32_t *obs_pt;
    const double *obs_xy;
    int32_t memory;           /* BA_MEM_HOST or BA_MEM_DEVICE: where ALL the pointers above live */
    int32_t reserved;
} BaProblem;

/* Solver options: the keyword arguments the reference passes to least_squares
 * (BundleAdjustment.py:210-220) plus the engine's own knobs. */
typedef struct BaOptions {
    int32_t loss;          /* BA_LOSS_*            <- robust_loss (:213) */
    int32_t max_iter;      /* LM trial steps       <- max_nfev (:215; counts trial steps, scipy least_squares.py:482) */
    double f_scale;        /*                      <- loss_f_scale (:214) */
    double ftol, xtol, gtol; /*                    <- 1e-8 each (:216-218) */
    double lambda0;        /* initial LM damping (default 1e-4) */
    double pcg_tol;        /* relative residual at which the ITERATIVE solves of the reduced camera system stop (PCG, preconditioned CG;
                              the direct solvers ignore it).  > 0: that tolerance at every step.  0 (default): adaptive, an inexact-Newton
                              forcing term eta = 1e-2 min(0.5, |g|inf) -- scipy's own lsmr_tol = 'auto' rule (_lsq/dogbox.py); the
                              reference's TRF runs lsmr at its default atol = btol = 1e-6 (_lsq/trf.py) */
    int32_t pcg_max_iter;  /* default 500 */
    int32_t linear_solver; /* BA_SOLVER_*: AUTO = dense Cholesky up to 128 free cameras; block cyclic reduction when
                              co-visible free cameras are at most 21 apart along a chain or a closed ring (in camera order,
                              or in an order the planner finds from the co-visibility graph when the numbering is not
                              along the trajectory);
                              a chain / ring plus a remainder of far-apart pairs (revisits): CG preconditioned by the
                              cyclic reduction of the in-band part;
                              a (nearly) full pattern up to 2048 free cameras: blocked dense Cholesky or block-Jacobi PCG,
                              whichever the first two steps time as faster; block-Jacobi PCG otherwise.  BANDED_CHOLESKY (chains, at most 16 apart): on request, and under AUTO only
                              for a chain that the cyclic reduction cannot plan (fewer than two blocks) */
    int32_t verbose;       /* 0 silent; 1 one summary line; 2 (<- verbose=2, :219) scipy's own progress table and closing lines, character
                              for character (scipy/optimize/_lsq/common.py print_header_nonlinear / print_iteration_nonlinear: the
                              starting point and every accepted step; then the termination message and "Function evaluations ...");
                              3 the engine's table instead: every trial step, damping, PCG iterations, solver choices */
    int32_t device;        /* HIP device ordinal; -1 = current device */
    int32_t deterministic; /* 1: no floating-point sum whose order depends on timing (no fp64 atomics in LDS or HBM: owner-lane
                              gathers, one slab row per partial result, fixed-order segmented sums; csrc/ba_det.h) -- the
                              same inputs give the same bits, run after run, as the reference's single-threaded solve
                              does (BundleAdjustment.py:210-220).  Slower build pass; belongs to the analysed structure
                              (cannot change on a handle).  0 (default): the atomic kernels */
    int32_t schur_mode;    /* how the Schur complement is accumulated.  1 = chunks: inside the chunk's workgroup, pair groups reduced
                              in registers, one flush per chunk (streets, rings: neighbouring points share their cameras);
                              2 = rows: per-observation rows in HBM and one owner per block of S walking a global entry list,
                              every block stored once, no atomics (csrc/ba_rows.h: unstructured co-visibility, where a chunk's
                              pair groups hold hardly more than one entry each); 0 (default) = AUTO: chunks, unless the analysis
                              finds fewer than three entries per group on average (BaAnalysis.auto_rows_mode).  Belongs to the
                              analysed structure like `deterministic`; with deterministic = 1 the rows mode is taken as it is (its
                              kernels have no atomics) */
    double curv_floor;     /* Hessian weight of a residual component = max(rho' + 2 rho'' z, curv_floor * rho');
                              1.0 = plain IRLS, default 0.1 (in (0, 1]) */
} BaOptions;

/* Result: the fields of scipy's OptimizeResult that the reference reads (:223-230). */
typedef struct BaSummary {
    int32_t status;        /* scipy codes: 0 max_iter, 1 gtol, 2 ftol, 3 xtol, 4 ftol+xtol (scipy _lsq/common.py:705-717); <0 BA_ERR_* */
    int32_t iterations;    /* LM trial steps taken (= nfev - 1) */
    int32_t accepted;      /* accepted steps */
    int32_t pcg_iterations;/* total PCG iterations */
    double initial_cost;   /* 0.5 * sum rho, scipy least_squares.py:220-227 */
    double final_cost;
    double grad_inf;       /* ||J^T (rho' f)||_inf = 'optimality' */
    double residual_sum;   /* signed sum of the final residuals (the reference's success quirk, :223-224) */
    double lambda_final;
    double t_setup_ms, t_solve_ms;
    int64_t n_chunks, nnz_blocks;
} BaSummary;

/* Structure statistics of the host-side symbolic analysis (no GPU needed). */
typedef struct BaAnalysis {
    int64_t n_chunks, n_pair_entries, nnz_blocks_upper, nnz_blocks_full;
    int64_t n_direct_pairs, n_direct_cam_obs; /* contributions that bypass the LDS tiles */
    int32_t n_free_cams, n_observed_points, max_track, max_chunk_cams, max_chunk_pairs, n_heavy_points;
    double mean_track;
    int64_t n_groups;        /* groups of eight pair entries over all chunks (chunk mode) */
    int32_t auto_rows_mode;  /* 1: BaOptions.schur_mode = 0 takes the rows mode on this structure */
    int32_t reserved;
} BaAnalysis;

typedef struct BaHandle BaHandle;

/* Collective hook for point-sharded multi-GPU solves: called with a DEVICE
 * pointer after the library has synchronised its stream; must return 0 on success and only
 * after the result is visible to later work on any stream.  op = BA_REDUCE_* / BA_BROADCAST0. */
typedef int (*BaCollectiveFn)(void *dev_ptr, int64_t count_f64, int32_t op, void *user);

int ba_version(void);
const char *ba_last_error(void);
void ba_default_options(BaOptions *opt);

/* ---- one-shot entry points --------------------------------------------------------- */
/* Replaces the least_squares(...) call + pack/unpack of global_bundle_adjustment
 * (BundleAdjustment.py:200-226).  On status >= 0 cam_pose / points are overwritten with the
 * optimised values (free cameras and all points); on error they are left untouched. */
int ba_solve(BaProblem *prob, const BaOptions *opt, BaSummary *out);
/* Replaces residuals(params) (BundleAdjustment.py:102-152): out[2*n_obs], caller's order. */
int ba_eval_residuals(const BaProblem *prob, double *out_residuals);
/* Replaces the objective scipy builds from it (least_squares.py:220-227). */
int ba_eval_cost(const BaProblem *prob, int32_t loss, double f_scale, double *out_cost);
/* Replaces the finite-difference Jacobian over jac_sparsity (BundleAdjustment.py:154-198, :212):
 * the non-zero blocks, jc[n_obs*12] = d(du,dv)/d(rvec,t) (2x6 row-major), jp[n_obs*6] (2x3). */
int ba_eval_jacobian_blocks(const BaProblem *prob, double *out_jc, double *out_jp);
/* Replaces compute_ba_reprojection_errors + utils.ReprojectionError (BundleAdjustment.py:388-423,
 * utils.py:271-276): mean L2 error per camera with the FULL 3x3 K (cam_k[n_cams*9]); cameras
 * without observations get NaN. */
int ba_eval_reprojection_errors(const BaProblem *prob, const double *cam_k, double *out_err);
/* The same with the poses exactly as the reference holds them in image_data (BundleAdjustment.py:404-405): utils.py:273
 * multiplies by the 3x3 R AS STORED, so a slightly non-orthonormal or float32-derived R (a PnP pose before any BA) must
 * not go through R -> rvec -> R.  cam_rt[n_cams*12] = R (9, row-major), t (3) per camera; prob->cam_pose and
 * prob->cam_intr are not read (may be NULL).  This is what compute_ba_reprojection_errors of the shim calls. */
int ba_eval_reprojection_errors_rt(const BaProblem *prob, const double *cam_rt, const double *cam_k, double *out_err);

/* The one entry point that allocates nothing.  It has NO counterpart in the reference: it exists for callers that
 * cannot allocate -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It is
 * ba_eval_cost (same kernels, same fixed-order reduction, bit-identical cost) with the library's temporaries -- the
 * per-camera constants and the per-block partial sums -- in a static device array instead of hipMalloc'd buffers:
 *   - every pointer of prob must be a DEVICE pointer (prob->memory == BA_MEM_DEVICE; host arrays would need staging
 *     buffers) and so is out_cost_dev (one double);
 *   - limits of the static array: n_cams <= 16384, n_obs <= 2^24 (16 777 216); beyond them BA_ERR_UNSUPPORTED;
 *   - the work is only ENQUEUED on `stream` (a hipStream_t; NULL = the default stream): no hipMalloc / hipFree, no
 *     host synchronisation; the result is in *out_cost_dev once the stream reaches that point;
 *   - the indices are never seen by the host, so they are checked on the device: an out-of-range obs_cam / obs_pt
 *     makes the result NaN (no fault);
 *   - the FIRST call on a device looks the static array's address up (hipGetSymbolAddress, once per device): make one
 *     call per device outside any graph capture before capturing calls;
 *   - every call on one GPU uses that GPU's one static array (about 4.4 MB per GPU, held until the process ends):
 *     calls that overlap on one GPU -- different streams or threads -- race on it.  Calls in stream order are fine. */
int ba_eval_cost_noalloc(const BaProblem *prob, int32_t loss, double f_scale, double *out_cost_dev, void *stream);

/* Optional: one tiny solve, so that the costs a fresh process pays once -- HIP runtime, code objects, the page-locked
 * staging buffer: some 50 ms -- do not land on the caller's first BA call (the reference's driver calls inter_ba after
 * its second registered image, Reconstruction.py:253-255).  BA_ERR_HIP without a device.  (No reference counterpart.) */
int ba_warm_up(void);

/* Host-only structure analysis (runs without a GPU). */
int ba_analyze(const BaProblem *prob, BaAnalysis *out);

/* Host-only (runs without a GPU): the block-cyclic-reduction schedule (csrc/ba_bcr.h) the engine would plan for a
 * reduced-camera-system pattern of n_free cameras -- pairs: 2*n_pairs int32 free-camera indices (i <= j), the diagonal
 * pairs included.  Serialised as int32 words into out (capacity words; *needed = words required, call with out = NULL
 * to size it):  [ok, cyclic | 2 * reordered, m, nt, nb, n_slots, n_init_slots, levels] | blk_start[m+1] | pos_f[m*(nb/6)]
 * (free camera at (block, offset) or -1: the cameras are taken in index order, else in reverse Cuthill-McKee order) |
 * src[(m+n_init_slots)*(nb/6)^2]
 * | lvl_elim[levels+1] | lvl_op[levels+1] | elims[5 each: k, gu, gv, u, v] | ops[9 each: kind, dst, a0, b0, a1, b1, k0,
 * k1, init] | n_rest | pend_ptr[m+1] | pend[2 * pend_ptr[m]] (per block the right-hand-side updates it has received by the
 * time it is eliminated, as (coupling slot, source block) pairs: the forward substitution of a NEW right-hand side through
 * the stored factors) | lvl_fwd[levels+1] | fwd[6 each: block, a0, k0, a1, k1, flags] (the same sweep level by level as
 * the GPU runs it: csrc/ba_bcr.h, BcrPlan::fwd).  ok = 0 (only the header is written): the pattern is neither a chain nor a ring of bandwidth <= 21
 * cameras.  What the tests execute with NumPy to check the schedule against a dense solve.  (No reference counterpart:
 * the reference leaves the linear algebra to scipy.) */
int ba_bcr_plan(int32_t n_free, const int32_t *pairs, int64_t n_pairs, int32_t *out, int64_t capacity, int64_t *needed);
/* The same, but a pattern that is a chain / ring PLUS a remainder (a revisit, a few long tracks: pairs of cameras further
 * apart than any block holds) gets a schedule for its IN-BAND part when that part holds at least half of the off-diagonal
 * blocks: header flag 4 is set and n_rest = the upper blocks left out (they appear in no src entry).  The engine uses
 * that factorisation as the preconditioner of CG on the whole system (AUTO / CYCLIC_REDUCTION on such a pattern). */
int ba_bcr_plan_rest(int32_t n_free, const int32_t *pairs, int64_t n_pairs, int32_t *out, int64_t capacity, int64_t *needed);

/* ---- handle API: problem resident on the device across calls ------------------------ */
BaHandle *ba_create(const BaProblem *prob, const BaOptions *opt);
void ba_destroy(BaHandle *h);
/* Extra co-visibility pairs (free-camera index pairs i<=j, 2*n_pairs int32) to add to the
 * reduced-camera-system pattern: ranks of a sharded solve must share ONE pattern.  Must be
 * called before the first build, and again after a ba_append on a sharded handle. ba_get_pattern returns this rank's own
 * pairs together with the extra ones it was given before. */
int ba_get_pattern(BaHandle *h, int32_t *out_pairs, int64_t capacity, int64_t *n_pairs);
int ba_set_pattern(BaHandle *h, const int32_t *pairs, int64_t n_pairs);
int ba_set_collective(BaHandle *h, BaCollectiveFn fn, void *user, int32_t rank, int32_t world);
/* How the callback is ordered against the library's work.  0 (default): the library synchronises its stream before
 * the call, and the callback returns when the collective is complete.  1 (stream-ordered): no host synchronisation;
 * the callback ENQUEUES the collective on the handle's HIP stream (ba_get_stream) -- after the work already queued
 * there -- and may return at once; the library queues its next kernels behind it.  (No reference counterpart: the
 * reference is single-process.) */
int ba_set_collective_mode(BaHandle *h, int32_t stream_ordered);
int ba_get_stream(BaHandle *h, void **stream); /* the hipStream_t every kernel of this handle runs on */
/* Built-in collectives: RCCL bound inside the library (librccl.so is opened at run time), so that a sharded solve has
 * no host code -- no Python callback -- between its kernels and its collectives: ncclAllReduce / ncclBroadcast are
 * enqueued in place on the handle's stream.  Rank 0 calls ba_rccl_unique_id (128 bytes) and hands the id to the other
 * ranks by whatever channel the launcher has (torch.distributed, MPI, a file); every rank then calls ba_rccl_init on its
 * handle, collectively, after ba_set_pattern.  The communicator belongs to the handle.  A later ba_set_collective
 * replaces it with the callback again.  (No reference counterpart: the reference is single-process.) */
int ba_rccl_unique_id(void *out_id128);
int ba_rccl_init(BaHandle *h, const void *id128, int32_t rank, int32_t world);
/* One residual + Jacobian + Schur-complement pass at the current state with damping lambda:
 * (x, lambda) -> cost, reduced camera system S, reduced gradient.  The benchmark unit. */
int ba_build(BaHandle *h, double lambda);
/* Timed loop of ba_build with HIP events on the library's stream: average ms per pass and per
 * launch of the dominant (fused linearise+Schur) kernel. */
int ba_build_bench(BaHandle *h, double lambda, int32_t warmup, int32_t reps, double *ms_per_pass,
                   double *ms_fused_kernel);
/* New values for the SAME structure (f1: device-resident problem reuse).  prob must have the sizes, cam_fixed and
 * (obs_cam, obs_pt) of the problem the handle was created from (checked; BA_ERR_INVALID otherwise); cam_pose,
 * cam_intr, points and obs_xy are uploaded again, opt (NULL = keep) replaces loss / tolerances / damping, and the
 * LM state starts afresh.  Skips the symbolic analysis, every allocation and the index uploads of ba_create: what
 * the reference's repeated inter_ba calls over an unchanged track set (Reconstruction.py:253-255) would re-pay. */
int ba_set_values(BaHandle *h, const BaProblem *prob, const BaOptions *opt);
/* f1, the incremental driver's pattern (inter_ba every second image on a GROWING problem, Reconstruction.py:244-255,
 * marshalling BundleAdjustment.py:243-306): `grown` is the handle's problem with cameras, points and observations
 * added at the END of its arrays (the old ones are a prefix: same gauge, same (camera, point) per old observation --
 * checked); all values are taken from `grown`, the LM state starts afresh, opt (NULL = keep) as in ba_set_values.
 *   *mode = 1  appended: the new observations only see points nobody saw before (what the reference's driver records:
 *              a new image's points are its own).  Only those points are analysed; their chunks, pair entries and index
 *              arrays are appended behind the old ones on the host and in HBM (the resident arrays grow in place,
 *              re-allocated geometrically), old block slots keep their numbers.
 *   *mode = 0  rebuilt inside the handle: an old track grew, a long-track ("heavy") point is involved, or the prefix
 *              rule does not hold (e.g. the gauge changed at 15 images, BundleAdjustment.py:38).  Same result, full analysis.
 * A shrunken problem is BA_ERR_INVALID.  mode may be NULL.
 * On a sharded handle (ba_set_collective / ba_rccl_init) every rank appends its own shard's growth -- the cameras are
 * replicated, so every rank passes the same grown camera arrays -- and the ranks then agree on ONE pattern again:
 * ba_get_pattern, union over the ranks, ba_set_pattern (collectively, as at set-up); until then ba_build / ba_run on the
 * handle return BA_ERR_INVALID.
 * If the call fails after the structure has begun to change (out of memory, a HIP error), the handle is marked
 * unusable: every other entry point returns BA_ERR_INVALID for it until a later ba_append succeeds (it then rebuilds
 * the whole structure) or the handle is destroyed. */
int ba_append(BaHandle *h, const BaProblem *grown, const BaOptions *opt, int32_t *mode);
/* Up to max_steps LM trial steps (continues from the current state); fills *out. */
int ba_run(BaHandle *h, int32_t max_steps, BaSummary *out);
int ba_get_state(BaHandle *h, double *cam_pose, double *points); /* host pointers, caller's order */
int ba_get_cost(BaHandle *h, double *cost);
/* compute_ba_reprojection_errors (BundleAdjustment.py:388-423, called right after every BA,
 * Reconstruction.py:263) from the handle's RESIDENT state: the points and the observations stay where they are in HBM
 * (nothing O(points) or O(observations) is marshalled or uploaded again).  Per camera, host pointers, both optional:
 * cam_rt [n_cams*12] = R (row-major), t exactly as the caller stores them (utils.py:273 multiplies by R as stored;
 * NULL = the handle's current poses), cam_k [n_cams*9] full K (NULL = from the handle's fx, fy, cx, cy).
 * out_err: host, [n_cams], NaN for cameras without observations.  Not for sharded handles (BA_ERR_UNSUPPORTED: a rank
 * only holds its shard's observations). */
int ba_get_reprojection_errors(BaHandle *h, const double *cam_rt, const double *cam_k, double *out_err);
/* Parity probes (host pointers, any may be NULL): dense reduced system s[(6Nf)^2] row-major,
 * g_red[6Nf], per-camera u[n_cams*36] and g_c[n_cams*6], per-point v[n_points*9] (undamped)
 * and g_p[n_points*3] of the last ba_build. */
int ba_get_reduced_system(BaHandle *h, double *s, double *g_red, double *u, double *g_c, double *v,
                          double *g_p);
/* The same reduced camera system as it is stored: upper 6x6 blocks (row-major), keys[k] = (i << 32) | j with
 * i <= j free-camera indices.  keys == blocks == NULL only returns the count.  For problems whose dense S
 * would not fit (full BAL sizes). */
int ba_get_reduced_blocks(BaHandle *h, int64_t *keys, double *blocks, int64_t capacity, int64_t *n_blocks);
int ba_synchronize(BaHandle *h);

#ifdef __cplusplus
}
#endif
#endif /* BA_HIP_H */
