// device_types.hpp -- POD descriptors shared by host packer and kernels (HBM layout of a batch of windows).
//
// A batch is the concatenation of independent windows.  Every per-window array lives at an offset
// recorded in WinMeta; kernels are launched over (element, window) grids and exit early for windows
// whose LM loop has terminated (Lm::status != 0) -- the LM control flow never returns to the host.
#pragma once
#include <stdint.h>

namespace ctv {

struct WinMeta {
  int32_t K, F, L, M, NB, V, P, N;
  int32_t pn, pnb;
  int32_t knot0, bias0, lm0;   // offsets into state arrays (knots, bias states, landmarks)
  int32_t imu0, grp0, ngrp;    // IMU samples (sorted by group) / groups
  int32_t vis0, bc0;           // visual blocks (LANDMARK-major, padded: see Vp) / bias-chain links
  int32_t anc0, A;             // anchors: distinct (landmark, t_i, row_i, p_i) of the window's visual blocks, landmark-major
  int32_t vitem0, nvitem;      // visual work items (<= CH blocks of one frame pair, listed in Dev::vblk)
  int32_t vis_lds, Vp;         // 1: the packed visual Hessian fits in LDS (k_assemble_vis); Vp: block slots of the window incl. padding
                               // (a multiple of 64; a landmark's blocks are consecutive and never straddle a group of 64)
  int32_t u0, p0;              // offsets into per-unknown (sum N) and per-pose-unknown (sum P) arrays
  int32_t ldw, Lpad;           // W is [Lpad][ldw] (landmark-major, zero padded; ldw % 32 == 0, Lpad % 2 == 0)
  int32_t pv0, pblk0;          // prior vectors (sum pn) / prior blocks
  int32_t fix_ld, lock_bg, lock_ba, fixed_upto;
  int64_t H0;                  // offset of the window's [P][ldh] block in Hpp / S (doubles)
  int32_t ldh;                 // row stride of Hpp / S: P rounded up to 16 doubles, so every row starts on a 128-byte line
  int64_t W0;                  // offset of W (elements)
  int64_t pH0;                 // offset of the prior's J0^T J0 (pn*pn doubles)
  int32_t tr0, ntr;            // sparsity plan (host_pack.hpp: plan_sparsity): offset of the window's per-tile-row entries in Dev::env_first / tl_beg / tl_end;
                               // ntr = P / 16 + 1 tile rows (the rhs row P rides along as a row of the reduced system)
  int32_t Lobs, pad_sp;        // rows of W that can be non-zero (landmarks with observations sort first)
  int64_t dt_ns;
  double inv_dt;               // 1e9 / dt_ns  (reference spline_segment.h:58)
  double q_CI[4], p_CI[3], gravity[3], imu_w[6], img_w, cauchy_a, ld_lo, ld_hi;
};

struct VisItem { int32_t start, count; };  // blocks Dev::vblk[start .. start + count): one frame pair, frame-pair order

// A run of IMU samples sharing the same 4 active knots (segment s) and the same bias state.
struct ImuGroup {
  int32_t win, s, bias, start, count;
  int32_t kabs, babs, iabs;   // knot0 + s, bias0 + bias, imu0 + start: the group's own data can be requested before the window's record arrives
};

// Per-window Levenberg-Marquardt state (Ceres 1.14 TrustRegionMinimizer variables).
// sensor-to-IMU extrinsic applied by k_spline_eval when on != 0 (reference ExtrinsicParam::se3)
constexpr int VT_ROWS = 40;   // doubles per block record in Dev::Jt (factors.hpp: VB_*)
struct SensorExt { double q[4]; double p[3]; int on; };

struct Lm {
  double cost, cand_cost, initial_cost;
  double mu, nu;               // trust-region radius, decrease factor
  double model_change;
  double step2, xnorm2, cand_xnorm2;
  unsigned long long gmax_bits; // max-norm of x - Plus(x, -g) as the bit pattern of a non-negative double
  int32_t iter, invalid, status; // status 0 = running, else 1 + termination code
  int32_t cur;                   // which of the two normal-equation sets holds the linearisation at the CURRENT state (the other one
                                 // receives the speculative linearisation at the candidate and becomes current on acceptance)
  int32_t scaled, last_ok, step_valid, chol_fail, accept;
  int32_t nsucc, nunsucc, have_grad;   // have_grad: the candidate of this pass was linearised (not just costed)
  unsigned long long cand_gmax_bits;   // gradient max-norm at the candidate (becomes gmax_bits on acceptance)
  double cand_gd;                      // g(candidate) . delta: directional derivative at the trial point of the line search
  // Ceres' projected Armijo line search of bounds-constrained problems (TrustRegionMinimizer::DoLineSearch, line_search.cc):
  // ls_on = the reduced program has a bounded parameter (free line delay); ls_active 0 = not searching, 1 = searching (alpha holds
  // the next trial step), 2 = search failed, the full step is being re-evaluated, 3 = the first trial failed Armijo on a pass that
  // only costed the candidate (last iteration): the same trial is linearised next pass to get its directional derivative.
  // alpha = step size of the candidate x (+) alpha * delta.
  int32_t ls_on, ls_active, ls_iters, ls_prev_valid, ls_cur_valid, nls_steps, nls_reduced, pad2;
  double alpha, ls_gd0, ls_dmax;          // g . delta at x (initial directional derivative), max-norm of delta
  double ls_cur_x, ls_cur_v, ls_cur_g, ls_prev_x, ls_prev_v, ls_prev_g;
};

struct LmParams {
  double ftol, gtol, ptol, max_radius, min_radius, min_rel_dec, min_diag, max_diag;
  int32_t max_invalid, max_iters;
};

// Device pointers of one batch (all arithmetic is fp64, like the reference's).
struct Dev {
  int32_t nwin, Ktot, Ftot, Ltot, Mtot, Gtot, Vtot, NBtot, Utot, maxN, maxP, maxPn;
  int32_t schur_plain_in_H;   // k_schur_window_f64 left the 16 x 16 tiles without Schur products unwritten: k_cholesky_tiles forms them from Hpp + D itself
  const WinMeta *wins;
  // state, fp64 master copies: current and candidate
  double *quat, *pos, *bias, *rho, *ld;
  double *cquat, *cpos, *cbias, *crho, *cld;
  const int32_t *knot_win, *bias_win, *lm_win;
  // per consecutive knot pair (k, k+1) of the state about to be linearised (the initial state: k_knot_prep; every candidate:
  // k_step_finish): d = log(R_k^-1 R_k+1) and Jr^-1(d) -- shared by all residual blocks of the window
  double *lkd;           // [Ktot][3]
  double *kjri;               // [Ktot][9]
  // IMU factors (sorted by group)
  const ImuGroup *groups;
  const int32_t *imu_grp;
  const double *imu_u;        // [Mtot] normalised time in the segment
  const double *imu_meas;     // [6][Mtot] gyro xyz, accel xyz
  double *imu_tiles;          // [Gtot][32*32]  A^T A of the group, A = [J | r] (6n x 31)
  double *imu_cost;      // [Gtot] 1/2 |r|^2 of the group's samples; vis_cost [Vtot / 64] robustified cost of the wave's blocks;
  double *vis_cost;      // misc_cost [nwin] bias chain + prior: summed per window in a fixed order by k_lm_control (no atomics)
  double *misc_cost;
  // visual factors.  Anchors = the i ends: all blocks of a feature share (t_i, row_i, p_i) in the reference (trajectory_manager.cpp:
  // 367-383); the host finds the distinct ones (any caller-given blocks are handled: a landmark may own several anchors), k_vis_anchor
  // evaluates each once per linearisation into a record (factors.hpp: AREC doubles), the blocks evaluate only their j end.
  const int32_t *a_win, *a_lm;  // [Atot] window / landmark (window-local) of the anchor
  const int64_t *a_t;           // [Atot] t_i relative to the window's t0
  const int32_t *a_row;         // [Atot] row_i
  const double *a_obs;          // [2][Atot] p_i
  double *arec;                 // [Atot][AREC] records of the state being linearised
  int32_t *a_s;                 // [Atot] first active knot of the anchor end
  int32_t Atot, pad_at;
  const int32_t *v_win, *v_lm, *v_anc;   // [Vtot] window (-1: padding slot), landmark, anchor (absolute) of the block slot
  const int32_t *vb_win;                 // [Vtot / 64] the window of every group of 64 block slots (a window's slots start on a group boundary)
  const int64_t *v_tj;          // relative to the window's t0
  const int32_t *v_rowj;
  const double *v_obs;        // [2][Vtot] pjx, pjy
  const double *v_cauchy; // [Vtot] width a of the block's ceres::CauchyLoss(a) (<= 0: no loss): the reference picks it per residual
                         // block (trajectory_estimator.cpp:320-323: 1 when the feature is being marginalised, else 2)
  double *Jt;                 // robust-corrected block records, block-major [Vtot][VT_ROWS] (factors.hpp VB_*: rotation columns of the j end,
                         // inverse-depth and line-delay columns, residual, A~ (2 x 3) and the j end's blending coefficients -- the i-end
                         // columns are A~ times the anchor record and are rebuilt by the assembly): the assembly gathers the blocks of an
                         // item (frame-pair order) from the landmark-major block order, 320 contiguous bytes each; a wave of k_vis_eval
                         // writes the 64 x VT_ROWS entries of its blocks as ONE contiguous 20 KB region (from LDS).
  int32_t *vsj;          // [Vtot] first active knot of the j end
  const VisItem *vitems;
  const int32_t *vblk;   // [Vtot] block slots in frame-pair order, window by window (VisItem::start indexes it)
  const int32_t *vblk_anc; // [Vtot] the anchors (absolute) of those blocks
  int32_t maxL, maxLdw;
  // ---- sparsity plan of the reduced system (the reference solves with SPARSE_NORMAL_CHOLESKY, trajectory_estimator.cpp:371-384: a visual block
  // touches <= 2 x (4-5) knots, :293-309).  The rows of W are stored in SORTED landmark order (by first, then last touched knot): row r of
  // window w belongs to landmark lm_at[lm0 + r]; lm_pos is the inverse.  A row is non-zero only in the knot columns [6 lm_klo, 6 lm_khi + 6),
  // the line-delay column P - 1 (and g_rho, which rides as column P): conservative over the line delay's box (the segment of an observation moves
  // with row * line delay).  Everything else of the row is never written and stays zero from the upload's memset.
  const int32_t *lm_pos, *lm_at;   // [Ltot]
  const int32_t *lm_klo, *lm_khi;  // [Ltot] by ROW (sorted): first / last knot (khi < klo: the landmark has no observation, its row is zero)
  // per 16-column tile c of the window (tr0 + c, c < ntr): the rows [tl_beg, tl_end) of W that can be non-zero in the tile's columns (bias-only
  // tiles: empty), and env_first: the first tile column of tile row c inside the ENVELOPE of the reduced system S (and of its Cholesky factor:
  // fill stays inside the row envelope) -- tiles (r, c < env_first[r]) are structurally zero: not formed, not stored, not multiplied.
  const int32_t *tl_beg, *tl_end, *env_first;
  const int32_t *env_tile;         // the raw (unaligned) tile envelope, also for batches whose env_first is the whole triangle: k_cholesky_tiles skips empty tiles' products
  double *grs;                     // [Ltot] g_rho by ROW (written with dinv by begin_iteration): the Schur kernels stream rows, not landmarks
  int32_t *span_viol;              // device counter: an evaluation fell outside its landmark's planned span (never, unless the plan is wrong)
  int32_t max_span6, pad_ms;       // 6 x the widest landmark span of the batch (knot columns): LDS row width of k_vis_eval
  // bias chain
  const int32_t *bc_win, *bc_i, *bc_j;
  const double *bc_w;    // [NBtot][6]
  // prior (J0^T J0, J0^T r0, r0^T r0 precomputed on the host in fp64)
  const double *pH, *pb0, *pc0;
  const double *pJ0, *pr0;   // the prior as given: J0 (column-major n x n at pH0) and r0 (at pv0) -- residual summary only
  const int32_t *pcol;   // [sum pn] unknown index of each prior dimension
  const int32_t *pinv;   // [sum P] the inverse map: prior dimension of each pose unknown (-1: not in the prior)
  double *pgrad;         // [sum pn] J0^T r0 + (J0^T J0) dx at the state being linearised (k_misc, store mode)
  // per bias state: the IMU groups that carry it -- bgl_off [Ftot + nwin] (window w's F + 1 offsets start at bias0 + w), bgl [Gtot] group ids
  const int32_t *bgl_off, *bgl;
  double *Hpart;         // [nwin][parts][npart_stride] packed partial Hessians + gradients of the multi-part assembly
  int32_t npart_stride, pad_np;
  const int32_t *p_kind, *p_index, *p_off;
  const double *p_x0;
  // normal equations, two sets (Lm::cur): linearisation at the current state / speculative linearisation at the candidate
  double *HppS[2];       // [sum P*ldh] lower triangle used
  double *WS[2];              // Hpl^T, landmark-major
  double *HllS[2], *gS[2];   // [Ltot], [Utot]
  double *S, *rhs;       // Schur complement (lower) and its right-hand side [sum P]
  double *chol_inv;      // [nwin][chol_nblk][32][32] inverses of the diagonal blocks of the Cholesky factor (row-major)
  int32_t chol_nblk;
  double *dd, *dinv;     // LM damping per unknown [Utot]; 1/(Hll + dd) [Ltot] by ROW of W (sorted landmark order)
  double *cscale;        // Jacobi scaling [Utot]
  double *delta;         // step [Utot]
  const uint8_t *active; // [Utot] unknown is in the reduced program
  Lm *lm;
  int32_t *n_active;
  long long *dbg;        // optional clock64() stamps (profiling aid), may be null
  int32_t line_search;   // 1: restate Ceres' projected line search (all-fp64 product path)
  LmParams prm;
};

}  // namespace ctv
