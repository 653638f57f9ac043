/*
 * neo_planner.h -- C ABI of libneo_planner_hip.so (MI355X / gfx950).
 *
 * The reference (Amos-Chen98/neo-planner) has no FFI: its boundary is the Python
 * object protocol between ros_node/traj_planner_node.py and
 * traj_planner/expert_planner.py:MinJerkPlanner, and between MinJerkPlanner and
 * map_server/esdf.py:ESDF.  This header is the boundary the MI355X path puts
 * underneath that protocol; neo_planner_amd/planner.py binds it with ctypes and
 * keeps the reference's method names.  Every entry point cites the reference code
 * it replaces (paths relative to src/planner/scripts/).
 *
 * Conventions: plain C, int status return (0 = NEO_OK), caller-owned buffers,
 * opaque context, no exceptions.  All arrays are C-contiguous.  A context owns one
 * HIP stream; calls on one context are serialised, different contexts are
 * independent.  Pointers are HOST pointers unless the argument is documented as a
 * device pointer (the *_dev entry points take device pointers and are asynchronous
 * on the context's stream).
 *
 * Decision vector layout (expert_planner.py:211, :540-541):
 *   x[n] = [ int_wpts row-major (D, M-1) ; tau (M) ],  n = D*(M-1) + M.
 * Boundary states: head[3][D], tail[3][D] = position, velocity, acceleration
 *   (expert_planner.py:170-181).
 */
#ifndef NEO_PLANNER_H
#define NEO_PLANNER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEO_ABI_VERSION 1
#define NEO_MAX_PIECES 64 /* M: one lane per piece */
#define NEO_MAX_DIM 3     /* D */
#define NEO_LBFGS_M 10    /* maxcor (expert_planner.py:221) */

typedef struct neo_ctx neo_ctx; /* opaque */

/* status codes: per call (return value) and per trajectory (status[] arrays) */
enum {
  NEO_OK = 0,
  NEO_ERR_INVALID = 1,      /* bad argument */
  NEO_ERR_HIP = 2,          /* HIP runtime failure, see neo_last_error */
  NEO_ERR_NO_MAP = 3,       /* scene id has no ESDF uploaded */
  NEO_ERR_UNSUPPORTED = 4,
};

/* per-trajectory termination codes (neo_optimize_*): the ctypes host maps them to
 * the reference's exceptions (expert_planner.py:236-237, :481). */
enum {
  NEO_TRAJ_CONVERGED_GRAD = 0,   /* max|g| <= gtol                 (L-BFGS-B "NORM OF PROJECTED GRADIENT") */
  NEO_TRAJ_CONVERGED_F = 1,      /* rel. reduction of f <= ftol    (L-BFGS-B "REL_REDUCTION_OF_F")          */
  NEO_TRAJ_ABNORMAL = 2,         /* line search failed with empty memory (L-BFGS-B "ABNORMAL")             */
  NEO_TRAJ_MAXITER = 3,          /* iteration / evaluation cap                                                */
  NEO_TRAJ_NUMERIC_RANGE = 4,    /* exp(-tau) overflow: the reference raises OverflowError (:481)            */
  NEO_TRAJ_NONFINITE = 5,        /* NaN/Inf objective                                                         */
  NEO_TRAJ_BAD_SCENE = 6,        /* its map-table slot is outside the table (neo_optimize_batch_dev): left untouched */
  NEO_TRAJ_SUSPENDED = 7,        /* out of its launch's evaluation budget (neo_optimize_batch_budget_dev): resumable */
};
/* OR-ed into the code above when weighted collision cost > collision_cost_tol
 * (expert_planner.py:235-237 raises ValueError("collision cost too large")). */
#define NEO_TRAJ_FLAG_COLLISION 0x100

/* ESDF lookup mode */
enum {
  NEO_INTERP_NEAREST_2D_REF = 0, /* esdf.py:53-82: nearest cell, int() truncation, gradient in m/cell */
  NEO_INTERP_TRILINEAR_3D = 1,   /* north-star mode: trilinear distance + analytic gradient (m/m)     */
};

/* element type of an uploaded distance field / arithmetic of the sampling phase */
enum { NEO_F64 = 0, NEO_F32 = 1, NEO_F16 = 2 };

/* voxel order of a 3-D field in HBM */
enum {
  NEO_LAYOUT_LINEAR = 0, /* [z][y][x] */
  NEO_LAYOUT_YZ4 = 1,    /* yz-quads: voxel (x,y,z) stores d(y,z), d(y+1,z), d(y,z+1), d(y+1,z+1) contiguously, records in
                            [z][y][x] order: the 8 corners of a cell are 2 adjacent records (32 contiguous bytes), one
                            cache line per lookup and x-adjacent cells share half their bytes (4x the memory) */
  NEO_LAYOUT_CELL8 = 2,  /* cell-packed: the 2x2x2 corners of every interpolation cell contiguous (8x the
                            memory; one aligned 32-byte read per lookup instead of four gathers) */
  NEO_LAYOUT_BRICK = 3,  /* corner bricks: one 128-byte line per block of 2 x 2 x 2 cells holding the block's 27 corners
                            (fp32; fp16: 4 x 2 x 2 cells, 45 corners) -- a lookup reads one line, and a path stays on it
                            for two cells in every direction (4x the memory in fp32, like NEO_LAYOUT_YZ4) */
};

/* planner parameters: DefaultConfig / PlannerConfig fields (expert_planner.py:12-25,
 * ros_node/traj_planner_node.py:32-46); real values launch/config/planner_config.yaml:2-13 */
typedef struct neo_params {
  double v_max;
  double T_min;
  double T_max;
  double safe_dis;
  double delta_t;
  double weights[4]; /* energy, time, feasibility, collision */
  double collision_cost_tol;
  /* L-BFGS-B options, expert_planner.py:213-225 (tol=1e-4 -> ftol = gtol = 1e-4) */
  double ftol;
  double gtol;
  int32_t maxls;   /* 20 */
  int32_t maxiter; /* 15000 */
  int32_t maxfun;  /* 15000 */
  int32_t bugcompat_stale_T; /* 1 = reproduce expert_planner.py:528-533 */
  int32_t sample_dtype;      /* NEO_F64 | NEO_F32: arithmetic of the sampled cost terms */
  int32_t flags;             /* NEO_FLAG_* below; 0 = defaults */
} neo_params;

/* neo_params.flags.  The optimiser kernel exists in two register allocations with bit-identical results:
 * one wavefront per SIMD (shortest evaluation; default below 1024 trajectories per call) and two per SIMD
 * (slower evaluations, higher throughput once the trajectories queue for the SIMDs: large calls, or several
 * calls in flight on several streams; n <= 128 variables in the fp64 mode and on the 2-D map with D = 2, n <= 256
 * on 3-D fields with fp32 sampling). */
#define NEO_FLAG_ONE_WAVE_PER_SIMD 32
#define NEO_FLAG_TWO_WAVES_PER_SIMD 64
/* small problems (n <= 32 variables, M <= 16, one scene: a 3-D field with fp32 sampling, or the 2-D nearest-cell map
 * with D = 2 in either arithmetic -- the reference's own M = 3 shape): eight trajectories per wavefront, 8
 * lanes each, for n <= 16; four of 16 lanes beyond.  Opt-in: a piece's samples are strided over fewer lanes, so sums associate differently
 * and results agree with the default kernel to fp32 rounding, not bit for bit. */
#define NEO_FLAG_LANE_GROUPS 128
/* all-fp32 evaluation (fp32 sampling; 3-D fields and the 2-D reference map): the coefficient solve, the adjoint pass and the optimiser's vectors
 * and stored pairs in fp32 too -- for n <= 128 variables also the optimiser's scalars and the line search --, three
 * wavefronts per SIMD for n <= 128, two beyond.  Per evaluation the cost and gradient then agree with the
 * fp64 solve to ~1e-5 instead of 2e-6; the optimiser's statistics (evaluations, final costs) are those of the default
 * mode.  Opt-in throughput mode. */
#define NEO_FLAG_F32_SOLVE 2048
/* bits 1..16 switch phases off for timing experiments (tools/), 512 forces lane = piece, 4096 makes the all-fp32
 * adjoint pass reduce the transposed joint system itself instead of reusing the forward reduction's multipliers
 * (comparison runs): leave them 0 */

/* ---- lifetime ------------------------------------------------------------- */
int neo_abi_version(void);
/* device_id: HIP device ordinal.  stream: a hipStream_t to run on, or NULL to let the
 * context create its own. */
int neo_ctx_create(int device_id, void *stream, neo_ctx **out);
int neo_ctx_destroy(neo_ctx *ctx);
const char *neo_last_error(neo_ctx *ctx);
/* fills p with the ROS YAML defaults */
int neo_params_default(neo_params *p);
int neo_params_set(neo_ctx *ctx, const neo_params *p);
int neo_ctx_synchronize(neo_ctx *ctx);
/* stream of the calls that follow (NULL = back to the stream the context was created with).  The `_dev`
 * entry points are asynchronous and only read the context's maps, so a caller may keep several batches in
 * flight on several streams of one context; ordering between the batches' streams is the caller's business.
 * Map updates (neo_esdf_upload_*, neo_esdf_build_*, neo_esdf_drop) wait for ALL work in flight on the device
 * before they rewrite or free a scene's buffer, so a batch launched before the update reads the old map and one
 * launched after it the new map; map-table slots (neo_scene_slot) must be re-read after any update. */
int neo_ctx_set_stream(neo_ctx *ctx, void *stream);

/* ---- maps (map_server/esdf.py) ------------------------------------------- */
/* replaces ESDF.esdf_map / esdf_grad_x / esdf_grad_y (esdf.py:29-33) as looked up by
 * get_edt_dis / get_edt_grad (esdf.py:53-82).  Arrays are [height][width] float64. */
int neo_esdf_upload_2d(neo_ctx *ctx, int scene_id, const double *dist, const double *grad_x,
                       const double *grad_y, int width, int height, double resolution,
                       double origin_x, double origin_y);
/* replaces ESDF.occupancy_map_cb (esdf.py:11-33): int8 occupancy (100 = occupied) ->
 * exact EDT * resolution -> np.gradient, all on the device.  Optionally copies the three
 * arrays back (any of the out pointers may be NULL). */
int neo_esdf_build_2d(neo_ctx *ctx, int scene_id, const int8_t *occupancy, int width, int height,
                      double resolution, double origin_x, double origin_y, double *out_dist,
                      double *out_grad_x, double *out_grad_y);
/* 3-D distance field for NEO_INTERP_TRILINEAR_3D.  dist is [nz][ny][nx] of src_dtype
 * (host pointer, or device pointer when src_is_device != 0); it is stored on the device
 * as store_dtype in `layout`. */
int neo_esdf_upload_3d(neo_ctx *ctx, int scene_id, const void *dist, int src_dtype,
                       int src_is_device, int nx, int ny, int nz, double resolution,
                       const double origin[3], int store_dtype, int layout);
/* 3-D counterpart of neo_esdf_build_2d (esdf.py:23-29 in three dimensions): uint8 occupancy
 * [nz][ny][nx] (non-zero = occupied; host pointer, or device pointer when occ_is_device != 0) ->
 * exact Euclidean distance * resolution on the device, stored as for neo_esdf_upload_3d.
 * out_dist: optional HOST buffer [nz][ny][nx] float32 receiving the distances.
 * Device memory: besides the stored field the build needs 10 bytes per voxel of intermediates (NEO_ERR_HIP if they do not
 * fit); they are kept in the context for the next build while they are at most 512 MB and released otherwise.
 * At most 4096 voxels per axis.  Volumes whose squared diagonal leaves room in 31 bits (all of BASELINE.json's) take a
 * faster form of the line passes and, for rows of 4-byte aligned length up to 1024, of the x pass;
 * neo_esdf_build_config(ctx, NEO_EDT_GENERIC_LINES) forces the general form (same results; the tests run both).
 * 300^3 on one MI355X: 0.5 ms. */
int neo_esdf_build_3d(neo_ctx *ctx, int scene_id, const uint8_t *occupancy, int occ_is_device, int nx,
                      int ny, int nz, double resolution, const double origin[3], int store_dtype,
                      int layout, float *out_dist);
/* per-context switches of neo_esdf_build_3d (0 = defaults): comparison runs and tests */
#define NEO_EDT_GENERIC_LINES 1 /* the general form of the y / z line passes for every volume */
int neo_esdf_build_config(neo_ctx *ctx, int flags);
int neo_esdf_drop(neo_ctx *ctx, int scene_id);
/* point queries, replaces get_edt_dis / get_edt_grad called from Python
 * (astar_planner.py:134, traj_planner_node.py:474).  pts[n][D_map], grad[n][D_map]. */
int neo_esdf_query(neo_ctx *ctx, int scene_id, int n, const double *pts, double *dist, double *grad);

/* ---- cost / gradient (expert_planner.py:539-585) --------------------------
 * One evaluation of get_cost(x) and get_grad(x) for B trajectories of one scene.
 *   x[B][n], head[B][3][D], tail[B][3][D]
 *   cost[B]      = dot(costs, weights)                       (:558)
 *   costs4[B][4] = unweighted [energy, time, feasibility, collision] (:549-552)
 *   grad[B][n]                                               (:579)
 *   coeffs[B][6M][D]  polynomial coefficients (:336)         (may be NULL)
 *   status[B]    NEO_TRAJ_NUMERIC_RANGE or 0                 (may be NULL) */
int neo_cost_grad_batch(neo_ctx *ctx, int scene_id, int B, int M, int D, const double *x,
                        const double *head, const double *tail, double *cost, double *costs4,
                        double *grad, double *coeffs, int32_t *status);
/* same, device pointers, asynchronous on the context stream */
int neo_cost_grad_batch_dev(neo_ctx *ctx, int scene_id, int B, int M, int D, const double *x,
                            const double *head, const double *tail, double *cost, double *costs4,
                            double *grad, double *coeffs, int32_t *status);

/* ---- sampled terms alone (expert_planner.py:392-466: add_sampled_cost + add_sampled_grad_CT) ----
 * The ESDF-lookup kernel on its own, as the reference uses it after get_coeffs()
 * (all_planner_demo.py:46-51).  coeffs[B][6M][D], ts[B][M]  ->
 *   costs2[B][2]      unweighted feasibility and collision cost            (:413, :422)
 *   grad_C[B][6M][D]  weighted partials w.r.t. the coefficients            (:450, :465)
 *   grad_T[B][M]      weighted partials w.r.t. the durations               (:451, :466)
 * _dev: coeffs and grad_C must be 16-byte aligned (the kernel moves them as pairs of doubles). */
int neo_sampled_terms_batch(neo_ctx *ctx, int scene_id, int B, int M, int D, const double *coeffs,
                            const double *ts, double *costs2, double *grad_C, double *grad_T);
int neo_sampled_terms_batch_dev(neo_ctx *ctx, int scene_id, int B, int M, int D, const double *coeffs,
                                const double *ts, double *costs2, double *grad_C, double *grad_T);
/* the same with fp32 coefficient and partials buffers (round 6; sample_dtype NEO_F32 only, NEO_ERR_INVALID otherwise):
 * half the operand bytes of the fp32 sampling path and no conversions in the kernel.  coeffs[B][6M][D] and
 * grad_C[B][6M][D], grad_T[B][M] are floats; ts and costs2 stay doubles (int(T / delta_t) of :401 is taken in fp64).
 * _dev: coeffs and grad_C must be 8-byte aligned. */
int neo_sampled_terms_batch_f32(neo_ctx *ctx, int scene_id, int B, int M, int D, const float *coeffs,
                                const double *ts, double *costs2, float *grad_C, float *grad_T);
int neo_sampled_terms_batch_f32_dev(neo_ctx *ctx, int scene_id, int B, int M, int D, const float *coeffs,
                                    const double *ts, double *costs2, float *grad_C, float *grad_T);

/* ---- optimiser (expert_planner.py:205-237: plan_once) ----------------------
 * Runs L-BFGS-B(maxcor 10, no bounds) from x to termination for every trajectory,
 * entirely on the device.  scene_ids[B] selects the map per trajectory (NULL = all
 * use `scene_id`); all maps of one call must be of the same kind (2-D / 3-D), element type
 * and layout: NEO_ERR_INVALID otherwise.
 * The *_dev variant takes a DEVICE array of map-table slots (neo_scene_slot) in
 * place of scene ids, and `scene_id` then only names the kind of map; since the slots cannot
 * be inspected from the host, every map of that kind held by the context must then share
 * `scene_id`'s element type and layout (NEO_ERR_INVALID otherwise), and a slot outside the
 * table ends that trajectory with NEO_TRAJ_BAD_SCENE.
 *   x[B][n]         in: x0, out: final x (res.x)
 *   costs4[B][4]    unweighted costs at the final x
 *   costs4_last[B][4] unweighted costs at the LAST EVALUATED x -- what the reference
 *                   reports as weighted_cost / final_cost (:233-234)   (may be NULL)
 *   nit[B], nfev[B] L-BFGS-B iteration / evaluation counts (res.nit, res.nfev)
 *   status[B]       NEO_TRAJ_* | NEO_TRAJ_FLAG_COLLISION */
int neo_optimize_batch(neo_ctx *ctx, int scene_id, const int32_t *scene_ids, int B, int M, int D,
                       double *x, const double *head, const double *tail, double *costs4,
                       double *costs4_last, int32_t *nit, int32_t *nfev, int32_t *status);
int neo_optimize_batch_dev(neo_ctx *ctx, int scene_id, const int32_t *scene_ids, int B, int M,
                           int D, double *x, const double *head, const double *tail,
                           double *costs4, double *costs4_last, int32_t *nit, int32_t *nfev,
                           int32_t *status);
/* the same with separate start points: x0[B][n] is only read, the results go to x[B][n] (x0 == x is the in-place
 * form above).  A caller that optimises the same requests again (benchmarks, re-planning from a stored guess) keeps
 * x0 resident and needs no copy per launch. */
int neo_optimize_batch_from_dev(neo_ctx *ctx, int scene_id, const int32_t *scene_ids, int B, int M,
                                int D, const double *x0, double *x, const double *head,
                                const double *tail, double *costs4, double *costs4_last, int32_t *nit,
                                int32_t *nfev, int32_t *status);
/* ---- launches with an evaluation budget (round 5) ----------------------------
 * The duration of one launch is the duration of its LONGEST run (cfg2: 633 evaluations against a mean of 135), which the
 * other trajectories' results wait for.  neo_optimize_batch_budget_dev is neo_optimize_batch_from_dev for ONE scene with
 * a cap on the evaluations a trajectory may make IN THIS LAUNCH: a run that needs more is suspended -- status
 * NEO_TRAJ_SUSPENDED, its complete optimiser state (iterate, gradient, direction, line-search interval, the stored
 * pairs) in state[b] -- and a later launch with resume != 0 continues it exactly where it stopped: the finished run is
 * bit for bit the run of an unbudgeted launch (tests/test_gpu_budget.py).  The reference's own caps keep their meaning:
 * maxiter / maxfun of expert_planner.py:213-225 count over all launches of a run (NEO_TRAJ_MAXITER).
 *   state      DEVICE buffer of B * neo_optimize_state_bytes(M, D) bytes, the caller's, kept between the launches of a run
 *   subset     optional DEVICE array of n_subset trajectory indices: only these are launched -- the compacted re-launch of
 *              the stragglers; NULL = all B.  Arrays are always indexed by trajectory, never by position; an index
 *              outside 0 .. B - 1 is skipped
 *   resume     0: the launched trajectories start from x0; 1: those among them with status NEO_TRAJ_SUSPENDED continue
 *              from state, the others are left untouched
 * A suspended trajectory's x holds the point it evaluates next, costs4 the terms at its last iterate, nit / nfev its
 * counts so far.  3-D fp32 fields in the linear or brick layout, n <= 128 variables, every arithmetic mode. */
size_t neo_optimize_state_bytes(int M, int D);
int neo_optimize_batch_budget_dev(neo_ctx *ctx, int scene_id, int B, int M, int D, const double *x0, double *x,
                                  const double *head, const double *tail, double *costs4, double *costs4_last,
                                  int32_t *nit, int32_t *nfev, int32_t *status, void *state, int eval_budget,
                                  const int32_t *subset, int n_subset, int resume);
/* slot of a scene in the device-side map table, -1 if it has no map.  Slots change
 * whenever a map is uploaded or dropped. */
int neo_scene_slot(neo_ctx *ctx, int scene_id);
/* bytes of device (HBM) workspace neo_optimize_batch_dev keeps for B trajectories.  Currently 0:
 * the L-BFGS history (2 * maxcor * n doubles per trajectory) lives in LDS. */
size_t neo_optimize_workspace_bytes(int B, int M, int D);

/* ---- trajectory evaluation (traj_utils.py:85-222) --------------------------
 * state[B][K][3][D] = position, velocity, acceleration at t_k = k / hz, k < K; rows with
 * t_k >= sum(ts) are left zero and count[B] returns the valid number
 * (= len(np.arange(0, sum(ts), 1/hz)), traj_utils.py:185).  x as above. */
int neo_eval_traj_batch(neo_ctx *ctx, int B, int M, int D, const double *x, const double *head,
                        const double *tail, double hz, int K, double *state, int32_t *count);

/* ---- timing of the device work (bench.py) ----------------------------------
 * When enabled, every kernel launch of the named family is bracketed by HIP events on
 * the context stream; neo_profile_read returns launches and summed milliseconds. */
enum { NEO_KERNEL_EVAL = 0, NEO_KERNEL_OPTIMIZE = 1, NEO_KERNEL_ESDF_BUILD = 2, NEO_KERNEL_ESDF_SAMPLE = 3, NEO_KERNEL_COUNT = 4 };
int neo_profile_enable(neo_ctx *ctx, int on);
/* optional DEVICE array [B] that the next neo_optimize_batch_dev launches fill with the number of
 * quadrature samples (ESDF lookups) each trajectory evaluated; NULL switches it off. */
int neo_optimize_sample_counter(neo_ctx *ctx, int64_t *dev_counts);
/* results as they finish (round 6).  One launch lasts as long as its LONGEST run (cfg2: 527 evaluations against a mean of
 * 135) while 80 % of its trajectories are complete after ~2/3 of that time.  `counter` is a device-accessible int32 (device
 * memory, or pinned host memory the device can add to) that the caller zeroes; every later neo_optimize_batch_dev /
 * _from_dev launch on this context then adds 1 to it -- a system-scope release, after the trajectory's x, cost terms,
 * counts and status are stored -- for each trajectory it completes.  A host that sees the counter reach k may copy the result
 * arrays on another stream: the k finished trajectories are final (preset status[] to -1 to tell them apart, and copy status[]
 * FIRST: a trajectory marked finished in that copy is final in every array copied after it), bit for bit what the
 * completed launch leaves.  NULL switches it off.  Plain launches of optimize_kernel only (not the
 * lane-group kernel, not budgeted launches -- which exist to END a launch early instead). */
int neo_optimize_progress_counter(neo_ctx *ctx, int32_t *counter);
/* diagnostics: optional DEVICE array [B][cap][4] that the next neo_optimize_batch[_dev] launches fill with one record
 * per counted evaluation of every trajectory -- (f, line-search step, quadrature samples, iteration) -- so that a run
 * can be laid beside the CPU optimiser's evaluation by evaluation (tools/classify_divergence.py); NULL switches it off.
 * Not supported by the lane-group kernel. */
int neo_optimize_trace(neo_ctx *ctx, double *dev_trace, int cap);
/* diagnostics: optional DEVICE array [B][cap][2][n] that receives, per counted evaluation, the evaluated point x_k
 * and its gradient g_k (as doubles, whatever arithmetic the kernel ran in).  With neo_optimize_trace's (f, step, ...)
 * records this is everything needed to re-evaluate a device run point by point on the CPU oracle and to re-derive
 * every line-search / restart decision on the host (tests/test_gpu_replay.py).  `cap` must equal neo_optimize_trace's
 * when both are on; NULL switches it off.  Not supported by the lane-group kernel. */
int neo_optimize_trace_xg(neo_ctx *ctx, double *dev_xg, int cap);
/* optional DEVICE permutation [B] for the next neo_optimize_batch_dev launches: workgroup i works on
 * trajectory order[i].  Results stay in the caller's order.  Workgroups start in index order, so
 * putting the runs expected to be long first shortens the launch (a late long run is its tail);
 * NULL = identity; it must be a permutation of 0..B-1 and is ignored by launches of another batch size.
 * neo_planner_amd.BatchPlanner sorts by time slack (sum(ts) * v_max / distance).
 * LIFETIME: the array is the caller's and is read by every later neo_optimize_batch[_dev | _from_dev] launch of B
 * trajectories on this context until another order (or NULL) is set: keep it allocated until those launches have
 * completed.  Only the optimiser kernels read it. */
int neo_optimize_dispatch_order(neo_ctx *ctx, const int32_t *dev_order, int B);
/* the expected-effort order computed ON THE DEVICE from a batch's resident start points (round 6): key = time slack of the
 * guess, sum(T) v_max / |goal - start|, largest first, ties by index -- what neo_planner_amd.BatchPlanner.expected_effort_order
 * computes on the host: a keys kernel and a stable descending radix sort (rocPRIM) on the context's stream.  `scratch`
 * (neo_effort_order_scratch_bytes(B) bytes, 256-byte aligned; the B keys stay in its first B doubles) and order[B] are the
 * caller's device buffers (several batches in flight on several streams: one pair per batch).  Hand `order` to
 * neo_optimize_dispatch_order. */
size_t neo_effort_order_scratch_bytes(int B);
int neo_effort_order_dev(neo_ctx *ctx, int B, int M, int D, const double *x0, const double *head, const double *tail,
                         void *scratch, int32_t *order);
/* the same from a HOST permutation (copied into a context-owned device buffer); NULL or B = 0 resets.
 * Either way the permutation only applies to launches of exactly B trajectories. */
int neo_optimize_dispatch_order_host(neo_ctx *ctx, const int32_t *host_order, int B);
/* results of a batch as fp32 rows [x (n) | weighted total cost | 4 cost terms] -- what the ranks of a scene-sharded job
 * gather (neo_planner_amd/sharding.py): one launch on the context's stream, device pointers, weights4 on
 * the host.  out[B][n + 5]. */
int neo_pack_results_dev(neo_ctx *ctx, int B, int n, const double *x, const double *costs4, const double *weights4,
                         float *out);
/* the ESDF-lookup kernel's own permutation (neo_sampled_terms_batch[_dev] launches of exactly B trajectories; results stay
 * in the caller's order, bit-identical): there the lever is locality -- workgroup i runs on XCD i mod 8, each XCD has its
 * own L2, and BatchPlanner.spatial_order deals requests that fly through the same part of the field to the same XCD.
 * `order` is a host (on_device = 0) or device (on_device = 1) array of B ints; it is COPIED into a context-owned buffer
 * before the call returns (the call synchronises the context's stream), so the caller may free it right away.  NULL or
 * B = 0 resets.  Independent of neo_optimize_dispatch_order. */
int neo_sampled_terms_dispatch_order(neo_ctx *ctx, const int32_t *order, int on_device, int B);
int neo_profile_read(neo_ctx *ctx, int kernel, int64_t *launches, double *total_ms);
int neo_profile_reset(neo_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* NEO_PLANNER_H */
