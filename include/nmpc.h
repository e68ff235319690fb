/*
 * nmpc.h -- C ABI of the MI355X batched path-tracking NMPC solver (libnmpc_hip.so).
 *
 * Drop-in boundary for the reference's solver path.  The reference binds the
 * acados-generated C API (opaque capsule + ocp_nlp_* setters/getters), one OCP
 * instance per capsule; those structs are not reproducible (acados headers are
 * not part of the reference tree), so this ABI offers the same operations on a
 * BATCH of independent instances held by one handle.  Each entry point names the
 * reference call it replaces.  All functions return 0 (NMPC_OK) or a negative
 * NMPC_E* code; none throws.  A handle is externally synchronised (one caller at
 * a time, like the reference's process-wide static solver,
 * src/PathTrackingSys.cpp:59); distinct handles are independent.
 *
 * Host arrays are caller-owned, row-major, double unless stated:
 *   state  x0[B][6]   = x, y, psi, vel, delta, throttle         (include/mpc_local_planner/types.h:39-41)
 *   params p [B][6]   = c0, c1, c2, c3, pitch, v_ref            (AcadosPathTracking.h:31-34)
 *   x     [B][N+1][6], u [B][N][2] = deltaDot, throttleDot      (types.h:80)
 * Per-instance status uses acados numbering: 0 success, 1 NaN, 2 max iterations
 * (SQP mode), 4 QP failure.
 */
#ifndef NMPC_H
#define NMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMPC_OK 0
#define NMPC_EINVAL (-1)   /* bad argument (null pointer, B > capacity, N out of range ...) */
#define NMPC_EHIP (-2)     /* a HIP runtime call failed; see nmpc_last_error() */
#define NMPC_ENOMEM (-3)

#define NMPC_MODE_RTI 0    /* nlp_solver_type SQP_RTI (config/mpc_params.yaml:38) */
#define NMPC_MODE_SQP 1    /* nlp_solver_type SQP */

typedef struct nmpc_config {
    int N;              /* mpc_N   (config/mpc_params.yaml:35); 1 <= N <= 127 */
    double dt;          /* mpc_dt  (:41) */
    double wheelbase;   /* (:59) */
    double W[7];        /* diag of ocp.cost.W (acados/mpcPathTracking.py:103) */
    double lbu[2], ubu[2];  /* steering_angle_dot_range, throttle_dot_range (:56,:50) */
    double lbx[2], ubx[2];  /* steering_angle_range, throttle_range (:53,:47), states 4,5 */
    /* interior-point QP solver (replaces the HPIPM options of mpcPathTracking.py:123-127) */
    int ipm_max_iter;
    double ipm_mu0, ipm_thr0, ipm_tau, ipm_tol_mu, ipm_tol_lin;
    double ipm_tol_comp;  /* every slack/multiplier pair: min(t_i, lam_i) <= this at termination */
    /* early failure: an iteration whose infeasibility scale (product of 1 - step length so far) is still
     * above ipm_stall_lin after ipm_stall_iter iterations is given up as QP failure (status 4) instead of
     * running to ipm_max_iter: measured on warm config-2 sequences, every QP that failed at the limit had a
     * scale >= 0.36 at iteration 20, every QP that converged <= 0.18.  ipm_stall_iter = 0 disables it. */
    int ipm_stall_iter;
    double ipm_stall_lin;
    /* Active-set polish of the QP.  The interior-point iteration identifies the active set long before it
     * has converged (complementarity 1e-3 against a termination at 1e-10); from there the QP is finished
     * exactly: the bounds classified active are imposed as equalities, the others dropped, and one Riccati
     * recursion solves that equality-constrained QP.  The result is accepted if it satisfies the KKT
     * conditions of the QP (the minimiser is then exact to round-off, the same point a converged HPIPM
     * returns to its tolerance: the QP is strictly convex); otherwise the set is corrected by the
     * primal-dual active-set rule, and after polish_tries corrections the interior-point iteration resumes.
     * Warm calls first try the active set their iterate sits on (bounded quantities on their bounds) when
     * the previous call of the instance ended on nearly the set it started from.  The reported iteration
     * count is the number of factorisations (interior-point iterations + active-set solves).
     * polish = 0 runs the interior-point iteration to its own termination test, as rounds 1-2 did. */
    int polish;
    double polish_mu;     /* first attempt once the average complementarity is below this (1e-3) */
    int polish_tries;     /* active-set corrections per attempt (2) */
    double polish_tol;    /* KKT tolerance of acceptance: multiplier signs, feasibility of dropped bounds (1e-10) */
    int warm_tries;       /* corrections of a warm call's first attempt (1); < 0: no such attempt */
    int warm_gate;        /* ... made when the previous call changed at most this many pairs of the set (2) */
    double warm_eps;      /* iterate within this of a bound = guessed active (1e-9) */
} nmpc_config;

typedef struct nmpc_handle nmpc_handle;

/* Reference defaults for everything but N and dt. */
int nmpc_default_config(nmpc_config *cfg, int N, double dt);

/* Run-time parameters of the same file that are not part of the OCP. */
typedef struct nmpc_run_params {
    int mode;                  /* nlp_solver_type (config/mpc_params.yaml:38): NMPC_MODE_RTI or NMPC_MODE_SQP */
    double loop_hz;            /* loop_Hz (:9; src/main.cpp:14), default 30 */
    double path_tracking_vel;  /* (:44; src/RosMpc.cpp:20), default 5.0 */
    double steering_ratio;     /* (:62; src/RosMpc.cpp:34), default 1.0 */
} nmpc_run_params;

/* Replaces the code-generation-time read of the ROS parameter file (acados/mpcPathTracking.py:83-92,
 * 105-108,126; CMakeLists.txt:29-40) and the run-time reads of src/RosMpc.cpp:20,34: fills cfg (and run,
 * if not NULL) from a file shaped like config/mpc_params.yaml.  Keys the reference's code generation
 * indexes (mpc_N, mpc_dt, wheelbase, the four *_range lists, nlp_solver_type) are required, as there;
 * W and the interior-point options keep the values of nmpc_default_config.  Host only, needs no GPU.
 * NMPC_EINVAL with the reason in nmpc_last_error() for unreadable files, missing or malformed keys. */
int nmpc_config_from_file(const char *path, nmpc_config *cfg, nmpc_run_params *run);

/* Replaces path_tracking_acados_create_capsule + _create_with_discretization
 * (src/AcadosPathTracking.cpp:28-49) for `capacity` instances on HIP device `device`.
 * All device memory, the stream and the iterate buffers are owned by the handle.
 * Every instance starts "cold" (see nmpc_reset). */
int nmpc_create(const nmpc_config *cfg, int capacity, int device, nmpc_handle **out);

/* Replaces path_tracking_acados_free + _free_capsule (src/AcadosPathTracking.cpp:51-64). */
int nmpc_destroy(nmpc_handle *h);

/* Replaces Solver::setInitCondition: lbx0 = ubx0 = x0 (src/AcadosSolver.cpp:15-26).
 * nmpc_set_x0 / nmpc_set_params copy the caller's rows into pinned staging memory of the handle and enqueue the
 * transfer on the handle's stream: the caller's buffer may be reused as soon as the call returns, every later
 * call on the handle sees the new values, and no call waits for the device (a consumer on ANOTHER stream
 * must order itself behind nmpc_stream(h), or call nmpc_sync first). */
int nmpc_set_x0(nmpc_handle *h, const double *x0, int B);

/* Replaces PathTracking::setParams / path_tracking_acados_update_params for stages 0..N
 * (src/AcadosPathTracking.cpp:17-24); the same p is used on every stage. */
int nmpc_set_params(nmpc_handle *h, const double *p, int B);

/* Same two inputs already resident in device memory (no copy; the pointers must stay valid
 * until the solves that use them have completed).  No reference counterpart: the reference
 * is host-only. */
int nmpc_set_inputs_device(nmpc_handle *h, const double *d_x0, const double *d_p, int B);

/* Control records of the following solves go to the caller's device buffer d_ctrl[capacity][6]
 * (NULL: back to the handle's own buffer); nmpc_get_controls, nmpc_closed_loop and
 * nmpc_device_ptr(h, 0) follow.  Lets a consumer alternate between two buffers, e.g. to all-gather
 * the records of one solve (RCCL) while the next solve runs.  No reference counterpart (host-only). */
int nmpc_set_controls_device(nmpc_handle *h, double *d_ctrl);

/* Per-instance cost weights W[B][7] / bounds bnd[B][8] = lbu0,lbu1,lbx4,lbx5,ubu0,ubu1,ubx4,ubx5
 * (the reference bakes one set into generated code, mpcPathTracking.py:103-115; the generic
 * acados setters ocp_nlp_cost_model_set / ocp_nlp_constraints_model_set are the counterpart).
 * NULL restores the handle's config values for all instances.  Weights must be positive and finite and
 * every lower bound below its upper bound (NMPC_EINVAL with the offending entry in nmpc_last_error()
 * otherwise; nmpc_create checks the config the same way). */
int nmpc_set_weights(nmpc_handle *h, const double *W, int B);
int nmpc_set_bounds(nmpc_handle *h, const double *bnd, int B);

/* Horizon sweep inside one launch: instance i solves the OCP with horizon N[i] (1 <= N[i] <= cfg.N, the
 * handle's horizon, which sizes the buffers); dt is the handle's.  NULL: every instance uses cfg.N again.
 * The reference fixes N at code generation (config/mpc_params.yaml:35, acados/mpcPathTracking.py:9), the
 * counterpart is one generated solver per N.  An instance with horizon n uses nodes 0..n of its iterate
 * and of nmpc_get's x / u rows (later nodes are left as they are, except that input row n carries the
 * solver's warm-start hint, not an input); its result is bit-identical to that of
 * a handle created with cfg.N = n.  Changing an instance's horizon makes its old iterate meaningless:
 * reset it (nmpc_reset).  Solved by the one-instance-per-workgroup kernel. */
int nmpc_set_horizons(nmpc_handle *h, const int32_t *N, int B);

/* Replaces Solver::setInitGuess / reInit (src/AcadosSolver.cpp:8-13,78-92): instances with
 * mask[i] != 0 (all if mask == NULL) start their next solve from x_k = x0, u_k = 0.
 * Instances whose last solve did not return status 0 (NaN, QP failure, SQP iteration limit) are reset
 * automatically, as the reference does for every status != ACADOS_SUCCESS (src/AcadosSolver.cpp:56-59).
 * A reset instance also forgets its iteration history (the launch order of the next solve). */
int nmpc_reset(nmpc_handle *h, const uint8_t *mask, int B);

/* Replaces the ocp_nlp_out_set "x"/"u" loop of Solver::setInitGuess (src/AcadosSolver.cpp:86-91)
 * with an arbitrary iterate: x[B][N+1][6], u[B][N][2].  The instances become "warm". */
int nmpc_set_iterate(nmpc_handle *h, const double *x, const double *u, int B);

/* Replaces path_tracking_acados_solve (src/AcadosPathTracking.cpp:26) for instances 0..B-1:
 * NMPC_MODE_RTI = one SQP-RTI iteration (rti_phase 0, src/AcadosSolver.cpp:42-48);
 * NMPC_MODE_SQP = up to max_iter iterations, stopping as acados' ocp_nlp_sqp does: when the NLP's KKT
 *   residuals at the current iterate with the last QP's multipliers (stationarity, shooting gaps, bound
 *   violation, complementarity; infinity norms) are all <= tol (acados default 1e-6); status 2 after
 *   max_iter QPs otherwise.
 * Asynchronous on the handle's stream; iterates stay on the device (warm start, no shift,
 * src/PathTrackingSys.cpp:58-61). */
int nmpc_solve(nmpc_handle *h, int B, int mode, int max_iter, double tol);

/* Wait for all work queued on the handle. */
int nmpc_sync(nmpc_handle *h);

/* Replaces ocp_nlp_out_get "x"/"u" (src/AcadosSolver.cpp:51-54).  Any output may be NULL.
 * Synchronises the handle's stream.  (x / u are split on the device and staged through pinned host memory
 * that the handle allocates at the first call asking for them; later calls allocate nothing.) */
int nmpc_get(nmpc_handle *h, int B, double *x, double *u, int32_t *status, double *cost,
             int32_t *ipm_iters, int32_t *sqp_iters);

/* The same trajectories without the copy into caller memory: *x and *u point at x[B][N+1][6] / u[B][N][2] in
 * pinned host memory owned by the handle, valid until the next nmpc_get / nmpc_get_view / nmpc_destroy of this
 * handle.  Replaces the same ocp_nlp_out_get loop (src/AcadosSolver.cpp:51-54) for callers that read the
 * horizon once (the MPCReturn packing of src/AcadosSolver.cpp:64-70 does).  x and u are split on the device
 * and arrive by one DMA each.  Synchronises the handle's stream. */
int nmpc_get_view(nmpc_handle *h, int B, const double **x, const double **u);

/* Per-instance control record [B][6]: u0.deltaDot, u0.throttleDot, x[1].delta, x[1].throttle
 * (what src/RosMpc.cpp:82-86 sends to the car), cost, status + 16*ipm_iters. */
int nmpc_get_controls(nmpc_handle *h, int B, double *ctrl);

/* ---- path front-end and closed loop (the caller side of the solve: SURVEY.md 8(f) rows f1, f2) ----
 *
 * Replaces PathTrackingSys's track (src/PathTrackingSys.cpp:13, PathTrackingSys.h setTrack): xy[n][2]
 * in the map frame, shared by all instances of the handle.  n >= 5. */
int nmpc_set_track(nmpc_handle *h, const double *xy, int n_points);

/* Replaces PathTrackingSys::calcCoeffs + the frame transform of PathTrackingSys::solve(state, pitch,
 * vRef) (src/PathTrackingSys.cpp:17-25,63-136) for B instances: world_state[B][6] =
 * x,y,psi,vel,delta,throttle in the map frame, pitch_vref[B][2].  Leaves the solver-frame state
 * (0,0,rotation,vel,delta,throttle) and the parameters (c0..c3,pitch,vRef) in the handle's device
 * input buffers, i.e. the next nmpc_solve uses them. */
int nmpc_frontend(nmpc_handle *h, const double *world_state, const double *pitch_vref, int B);

/* What the front-end produced: x0_solver[B][6], params[B][6], rotation[B] (any may be NULL). */
int nmpc_get_frontend(nmpc_handle *h, int B, double *x0_solver, double *params, double *rotation);

/* Replaces the un-rotation loop of PathTrackingSys::solve (src/PathTrackingSys.cpp:27-41): predicted
 * positions of the last solve in the map frame, xy[B][N+1][2] (psi is not un-rotated there either). */
int nmpc_get_world_path(nmpc_handle *h, int B, double *xy);

/* Closed loop on the device, `steps` control ticks of: front-end -> one SQP-RTI iteration (warm
 * start, no shift) -> actuation rule of src/RosMpc.cpp:81-90 (delta <- x[1].delta, throttle <-
 * x[1].throttle) -> plant (RK4 of x,y,psi,vel over dt_plant seconds with the commands held;
 * src/main.cpp:14 runs the loop at 30 Hz).  world_state[B][6] in/out, pitch_vref[B][2].
 * trace (optional) receives the world states after every tick, [steps][B][6]; n_failed (optional)
 * the number of (instance, tick) solves with status != 0 (asking for it adds a host sync per tick). */
int nmpc_closed_loop(nmpc_handle *h, int B, int steps, double dt_plant, double steering_ratio,
                     double *world_state, const double *pitch_vref, double *trace, int32_t *n_failed);

/* The same loop with the other branch of the reference's actuation rule, use_actual_steering_topic = true
 * (src/RosMpc.cpp:25,88-90,150): the steering angle of the state is not the commanded one fed back but a
 * measurement -- steer_meas[steps][B] are the steering-WHEEL angles of the actual-steering topic as they
 * arrive before each tick's solve (state delta = angle / steering_ratio, actualSteeringCallback); the car
 * moves with that angle during the tick.  Throttle is fed back as before (prevThrottle, src/RosMpc.cpp:84).
 * commands (optional) receives what src/RosMpc.cpp:82-86 publishes per tick, [steps][B][2] =
 * x[1].delta * steering_ratio, x[1].throttle -- the input of the caller's steering actuator. */
int nmpc_closed_loop_measured(nmpc_handle *h, int B, int steps, double dt_plant, double steering_ratio,
                              double *world_state, const double *pitch_vref, const double *steer_meas,
                              double *commands, double *trace, int32_t *n_failed);

/* One SQP-RTI iteration of instances 0..B-1 (what nmpc_solve does in NMPC_MODE_RTI) WITHOUT a handle and
 * without any allocation.  No counterpart in the reference: it exists for callers that cannot allocate
 * device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * Calls no hipMalloc / hipFree / allocating library routine; the solver's own workspace (launch order,
 * cold-start flags, iteration history, one row of default weights and bounds) is a static device array
 * of 344 KiB that every GPU the library runs on holds until the process ends.
 *   cfg, device : as nmpc_create; stream : hipStream_t (NULL = default stream); asynchronous on it
 *   B           : 1 .. NMPC_STATIC_MAX_BATCH; larger batches are rejected with NMPC_EINVAL and a message
 *   cold != 0   : start from x_k = x0, u_k = 0 (nmpc_reset); cold == 0: warm start from d_traj as it is
 *   inputs  (device): d_x0[B][6], d_p[B][6], d_W[B][7] or NULL, d_bnd[B][8] or NULL (NULL = cfg's values)
 *   in/out  (device): d_traj[B][N+1][8] iterate per node X,Y,psi,v,delta,thr,u0,u1 (nmpc_device_ptr(h, 1)
 *                     layout; node N has no input: its u0 slot carries the solver's warm-start hint from one
 *                     call to the next -- leave it as the last call wrote it, or zero it); d_ipm_iters int32[B] (read when cold == 0: previous counts order the launch)
 *                     d_status int32[B] (read when cold == 0: an instance whose previous status is not 0
 *                     starts cold, as the handle's own flags make it in nmpc_solve and as the reference
 *                     re-initialises after every failure, src/AcadosSolver.cpp:56-59; written with the new status)
 *   outputs (device): d_ctrl[B][6] control records (nmpc_get_controls layout)
 * Results are bit-identical to nmpc_solve on the same inputs (a warm call: given the statuses nmpc_solve left).  All calls on one GPU share that GPU's
 * workspace: calls that overlap on one GPU (different streams or threads) race -- serialise them. */
#define NMPC_STATIC_MAX_BATCH 16384
int nmpc_solve_static(const nmpc_config *cfg, int device, void *stream, int B, int cold, const double *d_x0,
                      const double *d_p, const double *d_W, const double *d_bnd, double *d_traj, double *d_ctrl,
                      int32_t *d_status, int32_t *d_ipm_iters);

/* Multi-GPU (one process per GPU, instances sharded by rank, no collective on the data path): all-gather
 * of the control records of instances 0..B-1 of every rank into d_all[world][B][6] on every rank, over
 * RCCL / xGMI, enqueued on the handle's stream behind the solve (asynchronous; nmpc_sync waits for it).
 * nccl_comm is the caller's ncclComm_t (ncclCommInitRank with its own rank / world size); every rank
 * must call with the same B.  RCCL is loaded on first use (dlopen of librccl.so), there is no link-time
 * dependency.  No reference counterpart: the reference is one process, one solver
 * (src/PathTrackingSys.cpp:58-61); this is what lets its caller hold the whole fleet's controls on
 * every GPU.  To overlap the gather with the next solve, alternate two buffers with
 * nmpc_set_controls_device and run the collective on a second stream. */
int nmpc_gather_controls(nmpc_handle *h, void *nccl_comm, int B, double *d_all);

/* Which instances of a fleet of `total` rank `rank` of `world` owns: the contiguous block
 * [*begin, *begin + *count); blocks differ by at most one instance, lower ranks take the remainder.
 * Host-only arithmetic (no GPU call), the same rule as mpc_local_planner_amd/sharding.py::shard_range,
 * so that C++ and Python callers cut a fleet the same way.  No reference counterpart (one solver there,
 * src/PathTrackingSys.cpp:58-61).  NMPC_EINVAL for a rank outside [0, world) or a negative total. */
int nmpc_shard_range(int64_t total, int rank, int world, int64_t *begin, int64_t *count);

/* Device-side views for zero-copy consumers (RCCL all-gather of the control records,
 * timing with HIP events).  which: 0 control records [cap][6], 1 iterate [cap][N+1][8]
 * (X,Y,psi,v,delta,thr,u0,u1 per node; the u0 slot of node N is not an input but the warm-start hint of
 * the QP polish, see nmpc_config.polish), 2 status int32[cap]. */
void *nmpc_device_ptr(nmpc_handle *h, int which);
void *nmpc_stream(nmpc_handle *h);       /* hipStream_t */
int nmpc_capacity(const nmpc_handle *h);
int nmpc_horizon(const nmpc_handle *h);

/* Text of the last HIP error seen by this thread ("" if none). */
const char *nmpc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* NMPC_H */
