/*
 * ctvo.h -- CPU fp64 ORACLE for the Ctrl-VIO sliding-window solve.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ctrl-vio_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (APRIL-ZJU/Ctrl-VIO) ships no tests, golden vectors or
 * fixtures for this path, and it cannot be built here (needs Eigen3 + Ceres 1.14 + glog +
 * ROS, none present).  This file restates the reference's arithmetic from its sources
 * (file:line cited per function, relative to /root/reference) and restates Ceres 1.14's
 * documented Levenberg-Marquardt loop.  It is pinned instead by (i) an independent NumPy
 * restatement + central finite differences + scipy.optimize.least_squares
 * (oracle/np_oracle.py -> tests/golden/), (ii) property tests.
 *
 * Plain C99, no dependencies.
 */
#ifndef CTVO_H_
#define CTVO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Kinds of parameter blocks kept by a marginalisation prior. */
enum { CTVO_PK_ROT = 0, CTVO_PK_POS = 1, CTVO_PK_BG = 2, CTVO_PK_BA = 3, CTVO_PK_LD = 4 };

/* One sliding window: state + factors.  All arrays caller-owned.
 * Unknown ordering used by every dense quantity below:
 *   knot k : rot 6k..6k+2, pos 6k+3..6k+5 ; bias f : bg 6K+6f.., ba 6K+6f+3.. ;
 *   line delay 6K+6F ; P = 6K+6F+1 ; inverse depth l : P+l ; N = P+L.            */
typedef struct ctvo_window {
  /* sizes */
  int32_t K, F, L, M, NB, V;
  int32_t pn, pnb;          /* prior: residual dim, number of kept blocks (0 = no prior) */
  /* spline (src/spline/se3_spline.h:108-111) */
  int64_t t0_ns, dt_ns;
  double *quat;             /* K*4 (x,y,z,w)   Sophus storage order, so3.hpp */
  double *pos;              /* K*3 */
  double *bias;             /* F*6 (bg, ba) */
  double *rho;              /* L inverse depths */
  double ld, ld_lo, ld_hi;  /* line delay [s] + box (trajectory.h:55-62) */
  int32_t fix_ld, lock_bg, lock_ba;
  int32_t fixed_upto;       /* knots with index <= fixed_upto are constant (-1: none) */
  /* calibration / weights */
  double q_CI[4], p_CI[3];  /* camera->IMU extrinsic (image_feature_factor.h:273-274) */
  double gravity[3];
  double imu_w[6];          /* info_vec (trajectory_value_factor.h:171) */
  double img_w;             /* sqrt_info = img_w * I2 (trajectory_manager.cpp:57) */
  double cauchy_a;          /* CauchyLoss(a) (trajectory_estimator.cpp:321-322); <=0: no loss */
  /* IMU factors */
  int64_t *imu_t; double *imu_gyro, *imu_acc; int32_t *imu_bias;
  /* bias random-walk chain */
  int32_t *bc_i, *bc_j; double *bc_w;   /* NB, NB, NB*6 */
  /* visual factors */
  int32_t *v_lm; int64_t *v_ti, *v_tj; int32_t *v_rowi, *v_rowj;
  double *v_pi, *v_pj;      /* V*2 normalised image coords (z = 1) */
  /* prior r = r0 + J0*dx (marginalization_factor.cpp:326-373) */
  double *pJ0;              /* pn*pn column-major */
  double *pr0;              /* pn */
  int32_t *p_kind, *p_index, *p_off;  /* pnb */
  double *p_x0;             /* pnb*4 */
  /* per residual block: CauchyLoss(marg_this_feature ? 1 : 2), one loss object per AddImageFeatureDelayAnalytic call
   * (trajectory_estimator.cpp:320-323).  V entries (<= 0: none) or NULL: cauchy_a for every block. */
  double *v_cauchy;
  /* per knot: SetParameterBlockConstant is decided per AddControlPoints call (trajectory_estimator.cpp:134-138);
   * K flags or NULL, on top of fixed_upto. */
  uint8_t *knot_const;
} ctvo_window;

typedef struct ctvo_summary {
  int32_t iterations;       /* Ceres iteration counter at exit */
  int32_t num_successful, num_unsuccessful;
  int32_t termination;      /* 0 no-convergence(max iters) 1 gradient 2 parameter 3 function 4 min-radius 5 failure */
  double initial_cost, final_cost;
  double final_radius;
  double cost_hist[64];     /* x_cost after each iteration (index = iteration) */
  int32_t num_line_search_steps;   /* Ceres Summary::num_line_search_steps: Armijo iterations beyond the first trial (alpha = 1) */
  int32_t num_line_search_reduced; /* LM iterations whose step was shortened by the projected line search (alpha < 1) */
} ctvo_summary;

/* --- Lie primitives (src/sophus_lib/so3.hpp:220-262,534-569; src/utils/sophus_utils.hpp:166-242) */
void ctvo_so3_exp(const double w[3], double q[4]);
void ctvo_so3_log(const double q[4], double w[3]);
void ctvo_so3_Jr(const double phi[3], double J[9]);
void ctvo_so3_Jr_inv(const double phi[3], double J[9]);
void ctvo_quat_to_R(const double q[4], double R[9]);

/* --- residual blocks (raw, i.e. before the robust corrector) ---
 * IMU  : r[6], J[6*30]  local cols: rot k0..k3 (12) | pos k0..k3 (12) | bg (3) | ba (3); *s = first knot.
 * visual: r[2], J[2*50] local cols: rot_i (12) | pos_i (12) | rot_j (12) | pos_j (12) | rho | ld.
 * J may be NULL (residual only).                                                      */
void ctvo_imu_block(const ctvo_window *w, int m, double *r, double *J, int32_t *s);
void ctvo_visual_block(const ctvo_window *w, int v, double *r, double *J, int32_t *si, int32_t *sj);
void ctvo_bias_block(const ctvo_window *w, int b, double *r, double *Jdiag /*6: +-diag*/);
void ctvo_prior_residual(const ctvo_window *w, double *r /*pn*/, double *dx /*pn*/);

/* Total cost 1/2 sum rho_b(|r_b|^2). */
double ctvo_cost(const ctvo_window *w);
/* Dense normal equations at the current state with robust-corrected blocks:
 * H (N*N row-major, full symmetric), g (N) = J^T r, returns cost.  N = 6K+6F+1+L. */
double ctvo_build_normal(const ctvo_window *w, double *H, double *g);
/* active[N]: 1 if the unknown is in the reduced program (referenced and not constant). */
void ctvo_active_mask(const ctvo_window *w, uint8_t *active);

/* LM solve with Ceres-1.14 semantics.  Updates the state in place.
 * use_schur: 1 = eliminate landmarks then dense P*P Cholesky; 0 = dense Cholesky on all N. */
int ctvo_solve(ctvo_window *w, int max_iters, int use_schur, ctvo_summary *out);
/* Test hooks: projected line search on/off (default on, as Ceres does for bounded problems); the interpolation step. */
void ctvo_set_line_search(int on);
double ctvo_ls_interpolate(int ns, const double *x, const double *value, const double *gradient, double x_min, double x_max);
/* Test hook: override function/gradient/parameter tolerances (Ceres defaults 1e-6, 1e-10, 1e-8). */
void ctvo_set_tolerances(double ftol, double gtol, double ptol);

/* One LM linear step from the current state (no update): delta[N] for radius mu, with the
 * Jacobi scaling computed at this same point.  Returns model_cost_change. */
double ctvo_lm_step(const ctvo_window *w, double mu, int use_schur, double *delta);

/* Retraction x (+) delta with projection on the line-delay box (ceres_local_param.h:137-145). */
void ctvo_plus(ctvo_window *w, const double *delta);

/* Trajectory query (se3_spline.h:361-399, so3_spline.h:240-322, rd_spline.h:229-259):
 * pose7 = (px,py,pz,qx,qy,qz,qw); vel3 world; omega3 body; acc3 world. Any out may be NULL. */
void ctvo_spline_eval(const ctvo_window *w, int n, const int64_t *t_ns, double *pose7,
                      double *vel3, double *omega3, double *acc3);

#ifdef __cplusplus
}
#endif

/* 4-DoF gauge restore after a solve (reference TrajectoryManager::double2vector, trajectory_manager.cpp:485-516, with
 * Utility::R2ypr / ypr2R, visual_odometry/utility.h:74-113): the yaw and the position of knot `knot` are put back to
 * their pre-solve values (q0 = x,y,z,w; t0) by one rigid transform applied to knots knot..K-1.  Near the Euler
 * singularity (|pitch| within 1 degree of 90) the full rotation difference R0 R00^T is used instead of the yaw. */
void ctvo_gauge_restore(int K, double *quat, double *pos, int knot, const double q0[4], const double t0[3]);

/* Jacobian-precision sensitivity study hook (not used by the parity tests); 0, 0 = off. */
void ctvo_set_jacobian_noise(double imu_rel, double vis_rel);
void ctvo_set_product_rounding(int on);

/* Prior construction = marginalisation of a window that holds the dropped factors (reference
 * MarginalizationInfo::marginalize, factor/analytic_diff/marginalization_factor.cpp:189-265; the factor evaluation with
 * the robust correction, :39-67, is ctvo_build_normal).  role[N]: 1 = marginalise, 0 = keep, -1 = not involved (its row
 * and column of A are ignored).  A = H (all factors of w), b = g, reordered [marginalised | kept];
 *   Amm^+ by symmetric eigendecomposition with eigenvalues <= eps dropped; A' = Arr - Arm Amm^+ Amr, b' = br - Arm Amm^+ bm;
 *   A' = V S V^T (eigenvalues <= eps -> 0):  J0 = sqrt(S) V^T,  r0 = S^-1/2 V^T b'.
 * kept[n] receives the unknown indices of the kept set in ascending order; J0 is n x n row-major (row i = sqrt(S_i) v_i^T,
 * eigenvalues ascending), r0 has n entries.  Returns n (<= 0: nothing to keep). */
int ctvo_marginalize(const ctvo_window *w, const int8_t *role, double eps, int32_t *kept, double *J0, double *r0);
/* cyclic Jacobi eigen-solver of a symmetric n x n matrix (row-major, destroyed): eigenvalues ascending in ev,
 * eigenvectors in the COLUMNS of V (row-major n x n). */
void ctvo_sym_eig(int n, double *A, double *ev, double *V);

#endif
