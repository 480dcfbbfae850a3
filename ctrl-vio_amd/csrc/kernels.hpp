// kernels.hpp -- gfx950 kernels of the sliding-window solve (all fp64).  One launch covers a whole batch of windows.  This header holds
// what the kernels share (linearisation modes, column maps, knot loading) and includes the sections in order:
//
//   kernels_control.hpp   k_lm_init, k_initial_cost, k_pass_end (gradient norm, cost, Ceres 1.14 accept / reject / terminate / Armijo, set
//                         swap, next iteration's damping), k_zero_normal (only for batches with an IMU-less window), k_knot_prep (d =
//                         log(R_k^-1 R_k+1) and Jr^-1(d) of every knot pair, shared by all blocks; the candidate's table is made by k_step_finish)
//   kernels_imu.hpp       k_imu_linearize_f64 (waves walking the IMU groups: rows through LDS, per-group A^T A on the fp64 matrix cores; it
//                         also clears the accumulated parts of the normal equations), k_imu_linearize_rest (groups the specialised body
//                         leaves out), assemble_imu_window (run by k_misc)
//   kernels_visual.hpp    k_vis_anchor (one record per anchor end), k_vis_eval (landmark-major: block records via LDS, the rows of W, Hll,
//                         g_rho formed in the same kernel), k_linearize_f64 (IMU + visual in one launch for small batches)
//   kernels_assemble.hpp  k_assemble_vis_mfma (MFMA + fp64 LDS Hessian, or global atomics for K > 25; STORE = the order-fixed tail of the
//                         deterministic mode with k_reduce_finalize / k_bias_rows), k_misc
//                         (IMU tiles' bias rows + bias chain + prior), k_post_linearize (first linearisation only)
//   kernels_solve.hpp     k_begin_iter, k_schur_window_f64 (large batches) / k_schur_tile_f64 (small; both also produce the reduced rhs),
//                         k_cholesky_tiles (register-resident 16 x 16 tiles; k_cholesky_solve =
//                         panel kernel for P > 223), k_step_finish (back-substitution, candidate x (+) alpha delta, its knot-pair table)
//   kernels_query.hpp     k_gauge_restore (double2vector), k_residual_summary, k_spline_eval (trajectory queries)
#pragma once
#include <utility>

#include "device_types.hpp"
#include "factors.hpp"

namespace ctv {

// SPECULATIVE LINEARISATION.  Every pass evaluates the candidate x (+) alpha delta exactly once -- residuals, Jacobians and the
// normal equations together, into the normal-equation set that is NOT the current one (Lm::cur).  The cost at the candidate is
// a by-product (per-group / per-wave partial sums, added up in a fixed order by k_pass_end); on acceptance the sets swap and the
// next iteration starts from a finished linearisation; on rejection the current set is still intact.  The trial points of
// Ceres' projected line search need value and gradient anyway.  Only the last allowed iteration (nothing can follow it) is
// costed without Jacobians.  Modes of the linearisation kernels:
enum { LIN_AT_X = 0,      // the current state into the current set (the first linearisation of a solve, diagnostics)
       LIN_SPEC = 1,      // the candidate into the other set (every pass)
       COST_AT_X = 2 };   // residuals of the current state only (ctvio_cost)
__device__ __forceinline__ bool lin_run(const Lm &lm, int mode) { return lm.status == 0 && (mode != LIN_SPEC || lm.step_valid != 0); }
__device__ __forceinline__ bool lin_cost_only(const Lm &lm, int mode, const LmParams &p) {
  return mode == COST_AT_X || (mode == LIN_SPEC && lm.iter >= p.max_iters && lm.ls_active == 0);
}
__device__ __forceinline__ int lin_target(const Lm &lm, int mode) { return mode == LIN_SPEC ? 1 - lm.cur : lm.cur; }

// (defined with the Schur / step kernels, used by k_pass_end)
__device__ __forceinline__ double grad_norm_entry(const Dev &d, const WinMeta &m, int w, int j, const double *g, bool at_cand);
__device__ __forceinline__ void begin_iteration(const Dev &d, int w, int *s_go);

// a lane's double as a wave-uniform value (two v_readlane_b32)
__device__ __forceinline__ double readlane_d(double x, int lane) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// local column -> unknown index maps
__device__ __forceinline__ int imu_col(int c, int s, int K, int bias) {
  if (c < 12) return 6 * (s + c / 3) + c % 3;
  if (c < 24) return 6 * (s + (c - 12) / 3) + 3 + (c - 12) % 3;
  return 6 * K + 6 * bias + (c - 24);
}
__device__ __forceinline__ int vis_col(int c, int si, int sj, int P) {
  if (c < 12) return 6 * (si + c / 3) + c % 3;
  if (c < 24) return 6 * (si + (c - 12) / 3) + 3 + (c - 12) % 3;
  if (c < 36) return 6 * (sj + (c - 24) / 3) + (c - 24) % 3;
  if (c < 48) return 6 * (sj + (c - 36) / 3) + 3 + (c - 36) % 3;
  if (c == 48) return -1;  // inverse depth: landmark block
  return P - 1;            // line delay
}

__device__ __forceinline__ void load_knots(const double *quat, const double *pos, int k0, const double *origin,
                                                            Knots4 &k) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double *q = quat + 4 * (k0 + i), *p = pos + 3 * (k0 + i);
    k.q[i] = qmk((double)q[0], (double)q[1], (double)q[2], (double)q[3]);
    k.p[i] = mk((double)(p[0] - origin[0]), (double)(p[1] - origin[1]), (double)(p[2] - origin[2]));
  }
}

// Knots k0..k0+3 in the local frame of the reference knot `kref` (fp64 arithmetic, then cast):
//   q'_k = q_ref^-1 q_k (near identity), p'_k = R_ref^T (p_k - p_ref).
struct LocalFrame {
  Q4 qref_inv;
  M3 RT;      // R_ref^T
  double o[3];
  __device__ __forceinline__ void init(const double *quat, const double *pos, int kref) {
    const double *q = quat + 4 * kref, *p = pos + 3 * kref;
    qref_inv = qmk(-q[0], -q[1], -q[2], q[3]);
    const M3 R = q2R(qmk(q[0], q[1], q[2], q[3]));
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) RT.m[3 * i + j] = R.m[3 * j + i];
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
  }
  __device__ __forceinline__ void load(const double *quat, const double *pos, int k0, Knots4 &k) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double *q = quat + 4 * (k0 + i), *p = pos + 3 * (k0 + i);
      const Q4 ql = qmul_raw(qref_inv, qmk(q[0], q[1], q[2], q[3]));   // unit x unit: no renormalisation in fp64
      const V3 pl = mul(RT, mk(p[0] - o[0], p[1] - o[1], p[2] - o[2]));
      k.q[i] = qmk((double)ql.x, (double)ql.y, (double)ql.z, (double)ql.w);
      k.p[i] = mk((double)pl.x, (double)pl.y, (double)pl.z);
    }
  }
  __device__ __forceinline__ V3 rotate(const double *v) const {  // R_ref^T v
    const V3 r = mul(RT, mk(v[0], v[1], v[2]));
    return mk((double)r.x, (double)r.y, (double)r.z);
  }
  __device__ __forceinline__ M3 RrefT() const {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = (double)RT.m[i];
    return r;
  }
};

}  // namespace ctv

#include "kernels_control.hpp"
#include "kernels_imu.hpp"
#include "kernels_visual.hpp"
#include "kernels_assemble.hpp"
#include "kernels_solve.hpp"
#include "kernels_query.hpp"
