/*
 * ctvio.h -- C ABI of libctvio.so: the MI355X (gfx950) sliding-window continuous-time VIO solve.
 *
 * Drop-in boundary for the Ceres build-and-solve behind the reference's
 * `ctrlvio::TrajectoryEstimator` (reference src/estimator/trajectory_estimator.h:61-206).  The reference
 * has no FFI; its seam is that C++ class, driven by TrajectoryManager::UpdateTrajectory
 * (src/estimator/trajectory_manager.cpp:317-483).  Every entry point below cites the reference
 * interface it replaces.  include/ctvio_estimator.hpp is the header-only C++ adaptor with the
 * reference's method names that a maintainer compiles src/estimator against.
 *
 * Conventions: plain C, caller-owned buffers, fp64 at the ABI regardless of device precision,
 * int status codes (no exceptions, like the reference: trajectory_estimator.cpp has no error paths),
 * one solver handle per host thread / HIP stream.  Parameters are addressed by INDEX (knot k,
 * bias state f, landmark l), not by pointer identity as in Ceres.
 *
 * Unknown ordering of every dense quantity:
 *   knot k : rot 6k..6k+2, pos 6k+3..6k+5 ; bias f : bg 6K+6f.., ba 6K+6f+3.. ; line delay 6K+6F ;
 *   P = 6K+6F+1 ; inverse depth l : P+l ; N = P+L.
 */
#ifndef CTVIO_H_
#define CTVIO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ctvio_solver ctvio_solver;

enum ctvio_status {
  CTVIO_OK = 0,
  CTVIO_ERR_INVALID = 1,    /* bad argument / inconsistent sizes / time outside the spline */
  CTVIO_ERR_NO_DEVICE = 2,  /* no HIP device: the product path has NO CPU fallback */
  CTVIO_ERR_HIP = 3,        /* a HIP runtime call failed (ctvio_last_error has the text) */
  CTVIO_ERR_STATE = 4,      /* call order violated (e.g. solve before upload) */
  CTVIO_ERR_INTERNAL = 5    /* a device-side consistency check failed (an evaluation left the knot span planned for its landmark): the
                               results of THIS call are not to be trusted; the counter is cleared, the batch stays usable */
};

/* CTVIO_FP64 (the only mode): every residual, Jacobian, product and factorisation in fp64, like the reference (Ceres / Eigen
 * are all double); reproduces the fp64 CPU reference's iterates, final state to ~1e-9.  The mixed fp32 mode of rounds 1-2
 * (CTVIO_FP32 = 0) missed the 1e-4 contract on one window in four and was removed: ctvio_create rejects it. */
enum ctvio_precision { CTVIO_FP32_REMOVED = 0, CTVIO_FP64 = 1 };

/* Kinds of parameter blocks kept by a marginalisation prior (reference
 * marginalization_factor.h:115-129 keep_block_*; sizes 4->local 3, 3, 3, 3, 1). */
enum { CTVIO_PK_ROT = 0, CTVIO_PK_POS = 1, CTVIO_PK_BG = 2, CTVIO_PK_BA = 3, CTVIO_PK_LD = 4 };

/* Replaces TrajectoryEstimatorOptions (trajectory_estimator_options.h:34-68) + the ceres::Solver::Options
 * set in TrajectoryEstimator::Solve (trajectory_estimator.cpp:371-398).  Zero-initialise then call
 * ctvio_default_options.                                                                        */
typedef struct ctvio_options {
  int32_t device;               /* HIP device ordinal */
  int32_t precision;            /* CTVIO_FP64 */
  int32_t use_mfma;             /* 1 (default): products on v_mfma_f64_16x16x4_f64; 2: as 1 with every IMU group evaluated by the
                                   general body (the one groups with knot-to-knot rotations >= 0.5 rad or anisotropic accelerometer
                                   weights take): cross-check of the specialised body.  0 (the vector-ALU cross-check kernels of rounds
                                   1-5) was removed: ctvio_create rejects it */
  int32_t check_every;          /* host polls "all windows terminated" every n LM iterations */
  double function_tolerance;    /* 1e-6  Ceres defaults, see SURVEY.md Appendix A */
  double gradient_tolerance;    /* 1e-10 */
  double parameter_tolerance;   /* 1e-8 */
  double initial_radius;        /* 1e4 */
  double max_radius, min_radius;/* 1e16, 1e-32 */
  double min_relative_decrease; /* 1e-3 */
  double min_lm_diagonal, max_lm_diagonal; /* 1e-6, 1e32 */
  int32_t max_consecutive_invalid_steps;   /* 5 */
  int32_t deterministic;        /* 1: order-fixed accumulation everywhere (no floating-point atomics): two runs of the same batch
                                   are bitwise equal.  It exists for batches whose every window has K <= 25 (packed Hessian in LDS):
                                   ctvio_upload / ctvio_set_batch return CTVIO_ERR_INVALID for any other batch
                                   instead of silently accumulating with atomics.  -1 (default): on for batches of <= 64 windows
                                   where it applies, the atomic path (run-to-run differences ~1e-13 in the state) otherwise; 0: off */
  int32_t host_threads;         /* host threads that validate / pack a batch (ctvio_set_batch, ctvio_upload); 0 = min(cores, 16) */
  int32_t use_graph;            /* 1 (default): the launch sequence of one LM pass is captured into a hipGraph and replayed */
  int32_t line_search;          /* 1 (default): Ceres' projected Armijo line search of bounds-constrained problems
                                   (a free line delay has bounds: trajectory_estimator.cpp:311-318 => Minimizer is_constrained) */
} ctvio_options;

/* One sliding window = what TrajectoryManager::UpdateTrajectory feeds a fresh TrajectoryEstimator
 * (trajectory_manager.cpp:331-451).  All pointers are read during ctvio_add_window only.          */
typedef struct ctvio_window {
  int32_t K, F, L, M, NB, V;    /* knots, bias states (frames), landmarks, IMU, bias-chain, visual blocks */
  int32_t pn, pnb;              /* prior residual dim / kept blocks (0: no prior) */
  int64_t t0_ns, dt_ns;         /* Se3Spline(time_interval_ns, start_time_ns), se3_spline.h:108-111 */
  const double *quat;           /* K*4 (x,y,z,w): getKnotSO3(i).data(), se3_spline.h:271 */
  const double *pos;            /* K*3: getKnotPos(i).data(), se3_spline.h:283 */
  const double *bias;           /* F*6 (bg, ba): para_bg_vec / para_ba_vec, trajectory_manager.cpp:332-342 */
  const double *rho;            /* L: para_Feature[l][0], trajectory_manager.h:96 */
  double ld, ld_lo, ld_hi;      /* trajectory_->line_delay, ld_lower, ld_upper (trajectory.h:55-62,99-103) */
  int32_t fix_ld;               /* trajectory_->fix_ld (trajectory_estimator.cpp:312-318) */
  int32_t lock_bg, lock_ba;     /* options.lock_wb / lock_ab (trajectory_estimator.cpp:236-245) */
  int32_t fixed_upto;           /* SetFixedIndex(idx) / lock_traj (trajectory_estimator.cpp:134-138); -1 none (see also knot_const) */
  double q_CI[4], p_CI[3];      /* ImageFeatureDelayFactor::S_CtoI / p_CinI (image_feature_factor.h:273-274) */
  double gravity[3];            /* AddIMUMeasurementAnalytic gravity argument */
  double imu_w[6];              /* info_vec (opt_weight.h:124-126) */
  double img_w;                 /* ImageFeatureDelayFactor::sqrt_info = img_w*I2 (trajectory_manager.cpp:57) */
  double cauchy_a;              /* ceres::CauchyLoss(a) of every visual block without an entry in v_cauchy; <= 0 : none */
  /* AddIMUMeasurementAnalytic (trajectory_estimator.h:102-106) x M */
  const int64_t *imu_t; const double *imu_gyro, *imu_acc; const int32_t *imu_bias;
  /* AddBiasFactor (trajectory_estimator.h:109-112) x NB : r = w .* (b_j - b_i), dt = 1 */
  const int32_t *bc_i, *bc_j; const double *bc_w;
  /* AddImageFeatureDelayAnalytic (trajectory_estimator.h:128-131) x V */
  const int32_t *v_lm; const int64_t *v_ti, *v_tj; const int32_t *v_rowi, *v_rowj;
  const double *v_pi, *v_pj;    /* V*2 (x, y), z = 1 */
  /* AddMarginalizationFactor (trajectory_estimator.h:146-148): r = r0 + J0*dx over the kept blocks */
  const double *pJ0;            /* pn*pn COLUMN-major (Eigen default of linearized_jacobians) */
  const double *pr0;            /* pn */
  const int32_t *p_kind, *p_index, *p_off; /* pnb: block kind, knot/frame index, column offset (keep_block_idx - m) */
  const double *p_x0;           /* pnb*4: keep_block_data (quaternion x,y,z,w or 3-vector / scalar, zero padded) */
  /* per residual block: the reference creates one loss per AddImageFeatureDelayAnalytic call, CauchyLoss(marg_this_feature ?
   * 1 : 2) (trajectory_estimator.cpp:320-323).  V entries (<= 0: no loss for that block) or NULL: cauchy_a for every block. */
  const double *v_cauchy;
  /* per knot: constancy is decided per AddControlPoints call (trajectory_estimator.cpp:134-138), so the constant knots need not
   * be a prefix.  K flags (non-zero: SetParameterBlockConstant) or NULL; applied on top of fixed_upto. */
  const uint8_t *knot_const;
} ctvio_window;

/* Replaces ceres::Solver::Summary (callers only print BriefReport(): trajectory_manager.cpp:314,455). */
typedef struct ctvio_summary {
  int32_t iterations;           /* Ceres-style iteration counter at exit */
  int32_t num_successful, num_unsuccessful;
  int32_t termination;          /* 0 max-iterations 1 gradient 2 parameter 3 function 4 min-radius 5 failure */
  double initial_cost, final_cost, final_radius;
  int32_t num_line_search_steps;   /* ceres::Solver::Summary::num_line_search_steps: Armijo iterations beyond the first trial */
  int32_t num_line_search_reduced; /* LM iterations whose step was shortened by the projected line search */
} ctvio_summary;

void ctvio_default_options(ctvio_options *opt);
const char *ctvio_status_string(int32_t status);
const char *ctvio_last_error(void);
int32_t ctvio_device_count(void);

/* new TrajectoryEstimator(trajectory, option) / ~TrajectoryEstimator (trajectory_estimator.h:76-84):
 * one handle owns a HIP stream and the device buffers of a BATCH of independent windows.            */
int32_t ctvio_create(const ctvio_options *opt, ctvio_solver **out);
void ctvio_destroy(ctvio_solver *s);

/* Drop all windows of the batch (a fresh estimator per solve in the reference: trajectory_manager.cpp:350). */
int32_t ctvio_clear(ctvio_solver *s);
/* The Add*Factor calls of one UpdateTrajectory, recorded as one window; returns the window id in *id. */
int32_t ctvio_add_window(ctvio_solver *s, const ctvio_window *w, int32_t *id);
/* Pack (sort IMU samples into (segment,bias) groups, visual blocks landmark by landmark, prior J0^T J0, ...), PLAN THE SPARSITY of the
 * reduced system -- what the reference leaves to Ceres' SPARSE_NORMAL_CHOLESKY (trajectory_estimator.cpp:371-384): every landmark's knot span
 * over the box [ld_lo, ld_hi] of the line delay, the row order of Hpl, the envelope of the Schur complement -- and copy to HBM.
 * Checked here (CTVIO_ERR_INVALID otherwise): P = 6K + 6F + 1 <= ~600; at most 64 visual blocks per landmark; finite observations and line
 * delay; ld_lo <= ld_hi for a free line delay; every row time t + row * ld inside the spline for ld at both ends of the box. */
int32_t ctvio_upload(ctvio_solver *s);
/* ctvio_clear + n x ctvio_add_window + ctvio_upload in one call, without the intermediate host copy: the n windows are
 * validated and packed straight from the caller's buffers (read during this call only) by opt.host_threads threads into a
 * pinned staging arena, which reaches HBM with a single asynchronous copy on the solver's stream.  Window ids = 0..n-1. */
int32_t ctvio_set_batch(ctvio_solver *s, int32_t n, const ctvio_window *wins);
int32_t ctvio_num_windows(const ctvio_solver *s);

/* TrajectoryEstimator::Solve(max_iterations) (trajectory_estimator.cpp:367-408) for every window of the
 * batch, device-resident LM with Ceres 1.14 trust-region semantics.  out: n_windows summaries (may be NULL). */
int32_t ctvio_solve(ctvio_solver *s, int32_t max_iterations, ctvio_summary *out);

/* Results back to the caller's live state (Ceres updates the double* in place; here explicit). */
int32_t ctvio_get_state(ctvio_solver *s, int32_t id, double *quat, double *pos, double *bias, double *rho, double *ld);
/* The state of EVERY window of the batch with one device-to-host copy: quat (sum K x 4), pos (sum K x 3), bias (sum F x 6),
 * rho (sum L), ld (n), each concatenated in window order.  Any pointer may be NULL. */
int32_t ctvio_get_batch_state(ctvio_solver *s, double *quat, double *pos, double *bias, double *rho, double *ld);
/* Overwrite the state of an uploaded window (re-solve the same factors from another initial guess).  A free line delay is projected on
 * [ld_lo, ld_hi]; the line delay of a fix_ld window must equal the uploaded one (the landmarks' knot spans were planned for it). */
int32_t ctvio_set_state(ctvio_solver *s, int32_t id, const double *quat, const double *pos, const double *bias,
                        const double *rho, double ld);

/* Keep / bring back a device-side copy of the whole batch state: re-solve the same windows from the same initial
 * guess without touching the host (the reference rebuilds its estimator from live state every frame instead). */
int32_t ctvio_snapshot_state(ctvio_solver *s);
int32_t ctvio_restore_state(ctvio_solver *s);

/* ---- diagnostics used by the per-kernel parity tests (ResidualSummary analogue,
 *      trajectory_estimator.h:37-59,168-171) ---- */
/* Linearise window id at its current state: Hpp (P*P row-major, symmetric filled), W (P*L row-major,
 * Hpl block), Hll (L), g (N), cost.  Any pointer may be NULL. */
int32_t ctvio_linearize(ctvio_solver *s, int32_t id, double *Hpp, double *W, double *Hll, double *g, double *cost);
/* Cost only (residual kernels). */
int32_t ctvio_cost(ctvio_solver *s, int32_t id, double *cost);
/* ResidualSummary (trajectory_estimator.h:37-59, AddResidualInfo at trajectory_estimator.cpp:36-67): for window id at its
 * current state, the sum over all blocks of |r_i| for every residual component of each factor type (the cost functions' own
 * whitened residuals, before any robust loss) and the block counts.  sums: IMU [6], bias [6], image [2], prior [pn]
 * (14 + pn doubles); counts4 = {M, NB, V, prior present}; err_ave of PrintSummary = sums / count. */
int32_t ctvio_residual_summary(ctvio_solver *s, int32_t id, double *sums, int32_t *counts4);
/* One LM step for radius mu at the current state (Jacobi scaling from this same point): delta (N),
 * model cost change; the state is not modified. */
int32_t ctvio_lm_step(ctvio_solver *s, int32_t id, double mu, double *delta, double *model_cost_change);

/* Batched trajectory query on the device: Se3Spline::poseNs / transVelWorld / rotVelBody / transAccelWorld
 * (se3_spline.h:361-399).  pose7 = (px,py,pz,qx,qy,qz,qw).  Any output may be NULL. */
int32_t ctvio_spline_eval(ctvio_solver *s, int32_t id, int32_t n, const int64_t *t_ns, double *pose7, double *vel3,
                          double *omega3, double *acc3);

/* The same queries for ANY windows of the batch in ONE launch (the per-row poses of rolling-shutter frames,
 * 11 x 640 row times per window, for a whole batch): query i belongs to window win[i] (0 .. ctvio_num_windows - 1) at absolute time
 * t_ns[i]; outputs in query order, laid out as in ctvio_spline_eval, any may be NULL.  Reference consumers: Se3Spline::poseNs /
 * transVelWorld / rotVelBody (se3_spline.h:361-399), Trajectory::GetSensorPose (trajectory.cpp:39-56) called once per time.
 * kernel_ms (may be NULL) receives the device time of the evaluation kernel alone (HIP events on the solver's stream); the call itself
 * also pays for the H2D copy of the queries and the D2H copy of the results. */
int32_t ctvio_spline_eval_batch(ctvio_solver *s, int64_t n, const int32_t *win, const int64_t *t_ns, double *pose7, double *vel3,
                                double *omega3, double *acc3, double *kernel_ms);

/* Sensor pose on the device: Trajectory::GetSensorPose (src/spline/trajectory.cpp:39-56; used for the camera by
 * visual_odometry.cpp:197-221): pose_S_to_G(t) = poseNs(t) * T_StoI with T_StoI = (q_SI = (x,y,z,w), p_SI).  pose7 as above.
 * A query outside [minTimeNs, maxTimeNs) is an error (the reference asserts). */
int32_t ctvio_sensor_pose(ctvio_solver *s, int32_t id, int32_t n, const int64_t *t_ns, const double *q_SI, const double *p_SI, double *pose7);

/* Prior construction for the next window (reference MarginalizationInfo::preMarginalize / marginalize,
 * src/estimator/factor/analytic_diff/marginalization_factor.cpp:106-265, driven by TrajectoryEstimator::
 * PrepareMarginalizationInfo / SaveMarginalizationInfo, trajectory_estimator.cpp:143-204).  Window `id` holds the factors
 * to be dropped (IMU samples, visual blocks with cauchy_a = 1 as in the reference, bias chain, the previous prior) at the
 * linearisation point; its normal equations A = sum J~^T J~, b = sum J~^T r~ are assembled on the device by the linearise
 * kernels.  role[N] (N = 6K + 6F + 1 + L, the library's unknown order): 1 = marginalise, 0 = keep, -1 = not involved.
 * Output: n_keep, kept[n_keep] (unknown indices, ascending), J0 (n_keep x n_keep row-major) and r0 (n_keep) such that the
 * prior residual of the next window is r0 + J0 dx (ctvio_window.pJ0 / pr0, column j of J0 = unknown kept[j]).
 * kept / J0 / r0 must have room for N, N*N and N entries.  eps = 1e-8 in the reference (:26).  Use a CTVIO_FP64 solver when
 * the prior has to match an fp64 reference tightly: the elimination amplifies the fp32 noise of the default path. */
int32_t ctvio_marginalize(ctvio_solver *s, int32_t id, const int8_t *role, double eps, int32_t *n_keep, int32_t *kept, double *J0, double *r0);

/* The same for EVERY window of the batch in one launch (one workgroup per window: Schur elimination + two parallel-Jacobi
 * eigendecompositions in LDS).  role: the windows' role arrays concatenated (sum of N_i entries, window order); outputs:
 * n_keep[n_windows]; kept: window i's kept unknowns at offset sum_{k<i} N_k; J0 / r0 packed tightly in window order
 * (offsets sum_{k<i} n_keep_k^2 / sum_{k<i} n_keep_k).  Every window needs <= 180 marginalised and <= 180 kept unknowns
 * (ctvio_marginalize falls back to a host factorisation beyond that). */
int32_t ctvio_marginalize_batch(ctvio_solver *s, const int8_t *role, double eps, int32_t *n_keep, int32_t *kept, double *J0, double *r0);
/* Where the LAST ctvio_marginalize / ctvio_marginalize_batch call of this handle factored: 0 = on the device (the product path), 1 = the
 * eigen-decompositions ran on the HOST cores (csrc/marginalize.hpp: a window with more than 180 marginalised or kept unknowns, or one whose
 * in-LDS Jacobi sweeps stalled; ctvio_marginalize only -- the batch entry fails instead).  The normal equations come from the device kernels
 * either way.  Reference counterpart of the host leg: MarginalizationInfo::marginalize, marginalization_factor.cpp:178-265. */
int32_t ctvio_marginalize_ran_on_host(const ctvio_solver *s);

/* 4-DoF gauge restore after a solve, on the device, for n windows of the batch at once (reference
 * TrajectoryManager::double2vector, src/estimator/trajectory_manager.cpp:485-516, called at :467 right after Solve):
 * for window ids[i], the rigid transform that puts the yaw (full rotation near the Euler singularity) and the position
 * of knot knot[i] back to the pre-solve pose q0[i] = (x,y,z,w), t0[i] is applied to knots knot[i] .. K-1.
 * knot[i] = computeTIndexNs(timestamps[0]).second - index of the window's first knot (reference :324-329). */
int32_t ctvio_gauge_restore(ctvio_solver *s, int32_t n, const int32_t *ids, const int32_t *knot, const double *q0, const double *t0);

/* ---- every GPU of the node from one C / C++ process (windows are independent units; device g owns the windows
 * {w : w mod G = g}, one host thread + one solver handle / HIP stream per device, no inter-GPU traffic during the solves) ----
 * ctvio_solve_sharded: ctvio_set_batch + ctvio_solve + ctvio_get_batch_state of the n windows spread over n_devices devices
 * (0: all visible ones; device ordinals 0 .. n_devices - 1; opt->device is ignored).  out (n summaries) and the state arrays are in
 * the caller's WINDOW order, laid out like ctvio_get_batch_state (quat: sum K x 4, pos: sum K x 3, bias: sum F x 6, rho: sum L,
 * ld: n); any of them may be NULL.  The per-device handles and one persistent host thread per shard are created on first use and kept
 * for the next call (grow-only arenas, no thread creation per call); ctvio_sharded_release frees them.  One sharded solve at a time
 * per process (concurrent callers serialise on a mutex: use one solver handle per thread for concurrent batches).  Python ranks use torch.distributed + ctrl-vio_amd/sharding.py for the same rule. */
int32_t ctvio_solve_sharded(const ctvio_options *opt, int32_t n_devices, int32_t n, const ctvio_window *wins, int32_t max_iterations,
                            ctvio_summary *out, double *quat, double *pos, double *bias, double *rho, double *ld);
void ctvio_sharded_release(void);
/* The partition itself (no device needed): owner of window w, and how many of n windows device g gets -- for G shards.  The G a
 * call of ctvio_solve_sharded really uses is ctvio_shards_used(n_devices, n) = min(n_devices or all, devices present, n): pass THAT
 * to ctvio_shard_of / ctvio_shard_count when predicting owners.  Options: a call whose *opt differs from the one a kept handle was
 * created with gets a fresh handle (tolerances, deterministic, use_mfma, line_search ... take effect immediately).  Failures inside a
 * shard's host thread (HIP errors, std::bad_alloc) come back as that shard's status; nothing is thrown across the ABI. */
int32_t ctvio_shard_of(int32_t window_id, int32_t n_devices);
int32_t ctvio_shard_count(int32_t n, int32_t device, int32_t n_devices);
int32_t ctvio_shards_used(int32_t n_devices, int32_t n);

/* Kernel timing of the next ctvio_solve calls with HIP events recorded on the solver's stream around every
 * launch group (adds a few microseconds per launch: use a dedicated profiling solve, not the timed one). */
int32_t ctvio_set_profiling(ctvio_solver *s, int32_t on);
/* Timing of the last ctvio_solve.  ms8: accumulated device time [ms] per launch group
 *   0 k_imu_linearize  1 k_vis_eval (linearise)  2 k_assemble_vis  3 zero + k_assemble_imu + k_misc + k_post_linearize
 *   4 Schur SYRK (k_schur_window / k_schur_mfma)  5 k_cholesky_solve  6 everything else (damping, rhs, backsub, update, cost, control)
 *   7 whole solve (always measured).  launches8: number of launches of each group, [7] = LM passes launched.
 * Groups 0..6 are zero unless profiling was on. */
int32_t ctvio_last_timing(ctvio_solver *s, double *ms8, int32_t *launches8);
/* The HIP stream every kernel of this solver is launched on (hipStream_t), for external event timing. */
void *ctvio_stream(ctvio_solver *s);
/* How many times this handle captured its LM pass into a hipGraph so far (diagnostic: a stream of equally shaped batches captures
 * once; the launch sequence the reference replaces is ceres::Solve's per-iteration Evaluate loop, trajectory_estimator.cpp:399). */
int32_t ctvio_graph_captures(const ctvio_solver *s);

/* ---- Diagnostic switches.  libctvio.so reads a fixed set of environment variables ONCE per handle, inside ctvio_create (csrc/ctvio.hip:
 * DebugSwitches, the only getenv of the library), and never again: kernel selection cannot change between ctvio_upload and ctvio_solve.
 * They exist for A/B measurements and for the tests' cross-checks; production callers set none of them.
 *   CTVIO_DENSE=1 (dense sparsity plan)   CTVIO_CHOL_TILES=0|1|3   CTVIO_SCHUR_TILE2=0|1   CTVIO_SCHUR_TILES=1   CTVIO_SCHUR_COPY_PLAIN=1
 *   CTVIO_STORE_PATH=0|1   CTVIO_SPLIT_LINEARIZE=1   CTVIO_MERGE_LINEARIZE=0|1   CTVIO_ZERO_KERNEL=1   CTVIO_NO_IMU_BAND=1   CTVIO_IMU_WAVES=n
 *   CTVIO_IMU_GENERAL=1   CTVIO_MARG_HOST=1   CTVIO_MARG_DEBUG=1   CTVIO_DEBUG_STAMPS=1
 *   CTVIO_SHARD_OVERSUBSCRIBE=1 (test only; read by ctvio_shards_used / ctvio_solve_sharded at call time) */

#ifdef __cplusplus
}
#endif
#endif /* CTVIO_H_ */
