// ctvio_estimator.hpp -- header-only C++ adaptor with the call surface of the reference's
// `ctrlvio::TrajectoryEstimator` (reference src/estimator/trajectory_estimator.h:61-206) on top of the C ABI
// (include/ctvio.h).  It lets code written like TrajectoryManager::UpdateTrajectory
// (src/estimator/trajectory_manager.cpp:317-483) keep its shape: construct an estimator from a trajectory,
// Add*Factor with raw `double*` parameters, Solve(max_iterations), results appear in place.
//
// What changes underneath: the reference hands pointers to ceres::Problem and parameter identity = pointer
// identity (trajectory_estimator.cpp:114-141, marginalization_factor.cpp:97-102).  Here the adaptor turns
// pointers into INDICES (knot k, bias state f, landmark l) by looking them up in the storage they came from,
// records the factors in flat arrays, and ships one ctvio_window to the GPU.  No Eigen/Sophus/Ceres/glog types:
// the reference's Eigen::Vector3d arguments become `const double*` (Eigen users pass v.data()).
//
// Not provided (reference methods that are declared but never defined, trajectory_estimator.h:87-143, or that
// belong to prior construction -- SURVEY.md section 8f-1): AddPoseMeasurementAnalytic, AddStartTimePose,
// AddStaticSegment, AddPreIntegrationAnalytic, AddImageFeatureAnalytic, AddDelayAnalytic, SetKeyScanConstant,
// PrepareMarginalizationInfo / SaveMarginalizationInfo.
#pragma once

#include <cmath>
#include <cstdint>
#include <deque>
#include <array>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "ctvio.h"

namespace ctvio {

// ---- Se3Spline<4>/Trajectory storage surface (reference src/spline/se3_spline.h:108-356, trajectory.h:38-116).
// Knots live in two deques (stable addresses under push_back, like so3_spline.h:410 / rd_spline.h:317).
class Trajectory {
 public:
  Trajectory(int64_t dt_ns, int64_t t0_ns = 0) : dt_ns_(dt_ns), t0_ns_(t0_ns) {}
  int64_t getDtNs() const { return dt_ns_; }
  int64_t minTimeNs() const { return t0_ns_; }
  int64_t maxTimeNs() const { return t0_ns_ + ((int64_t)numKnots() - 3) * dt_ns_; }  // rd_spline.h maxTimeNs
  size_t numKnots() const { return so3_.size(); }
  int cpnum() const { return (int)so3_.size(); }
  void knots_push_back(const double q_xyzw[4], const double p[3]) {
    so3_.push_back({q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3]});
    pos_.push_back({p[0], p[1], p[2]});
  }
  // extendKnotsTo (se3_spline.h:188-207): repeat the given knot until maxTime >= t
  void extendKnotsTo(int64_t t_ns, const double q_xyzw[4], const double p[3]) {
    while (numKnots() < 4 || maxTimeNs() < t_ns) knots_push_back(q_xyzw, p);
  }
  std::array<double, 4> &getKnotSO3(size_t i) { return so3_.at(i); }   // .data() = (x,y,z,w), se3_spline.h:271
  std::array<double, 3> &getKnotPos(size_t i) { return pos_.at(i); }   // se3_spline.h:283
  // computeTIndexNs (se3_spline.h:458-461 -> rd_spline.h:117-133): (u, first active knot)
  std::pair<double, size_t> computeTIndexNs(int64_t t_ns) const {
    const int64_t st = t_ns - t0_ns_;
    return {double(st % dt_ns_) / double(dt_ns_), size_t(st / dt_ns_)};
  }
  // trajectory.h:55-62,99-103
  void SetLineDelay(double ld_init, bool fix, double lo, double hi) { line_delay = ld_init; fix_ld = fix; ld_lower = lo; ld_upper = hi; }
  double line_delay = 0.0, ld_lower = 0.0, ld_upper = 3.5e-5;
  bool fix_ld = false;
  // camera -> IMU extrinsic (trajectory.h:64-74; consumed by ImageFeatureDelayFactor::S_CtoI/p_CinI)
  double q_CI[4] = {0, 0, 0, 1}, p_CI[3] = {0, 0, 0};

 private:
  friend class TrajectoryEstimator;
  int64_t dt_ns_, t0_ns_;
  std::deque<std::array<double, 4>> so3_;
  std::deque<std::array<double, 3>> pos_;
};

// reference src/utils/parameter_struct.h:58-65
struct IMUData {
  int64_t timestamp;
  double gyro[3];
  double accel[3];
};

// reference src/estimator/trajectory_estimator_options.h:34-68 (fields the solve reads)
struct TrajectoryEstimatorOptions {
  bool lock_traj = false, lock_ab = false, lock_wb = false;
  double image_weight = 800.0;  // ImageFeatureDelayFactor::sqrt_info (trajectory_manager.cpp:55-61)
  int precision = CTVIO_FP32;
  int device = 0;
};

// What MarginalizationInfo exposes to MarginalizationFactor (marginalization_factor.h:115-129).
struct MarginalizationInfo {
  int n = 0;                                   // residual dimension
  std::vector<double> linearized_jacobians;    // n*n column-major
  std::vector<double> linearized_residuals;    // n
  std::vector<int> keep_block_size;            // 4 / 3 / 1 (global sizes)
  std::vector<int> keep_block_idx;             // column offset of each kept block (already minus m)
  std::vector<std::array<double, 4>> keep_block_data;  // linearisation point of each block
};

struct SolveSummary {
  ctvio_summary s{};
  std::string BriefReport() const {  // the only thing the reference's callers use (trajectory_manager.cpp:314,455)
    static const char *term[] = {"NO_CONVERGENCE", "CONVERGENCE (gradient)", "CONVERGENCE (parameter)", "CONVERGENCE (function)",
                                 "CONVERGENCE (min radius)", "FAILURE"};
    return "ctvio: iterations " + std::to_string(s.iterations) + ", initial cost " + std::to_string(s.initial_cost) + ", final cost " +
           std::to_string(s.final_cost) + ", " + term[s.termination < 0 || s.termination > 5 ? 5 : s.termination];
  }
};

class TrajectoryEstimator {
 public:
  // TrajectoryEstimator(Trajectory::Ptr, TrajectoryEstimatorOptions&)  trajectory_estimator.h:76-77
  TrajectoryEstimator(Trajectory *trajectory, const TrajectoryEstimatorOptions &option) : traj_(trajectory), opt_(option) {
    for (size_t k = 0; k < traj_->numKnots(); ++k) knot_of_[traj_->so3_[k].data()] = (int)k;
  }
  // void SetFixedIndex(int idx)  trajectory_estimator.h:90
  void SetFixedIndex(int idx) { fixed_upto_ = idx; }

  // trajectory_estimator.h:102-106 / .cpp:219-263.  gyro_bias / accel_bias: pointers to 3 doubles; identical pointers
  // denote the same bias state (one per keyframe interval, trajectory_manager.cpp:332-342).
  void AddIMUMeasurementAnalytic(const IMUData &imu, const double gravity[3], double *gyro_bias, double *accel_bias,
                                 const double info_vec[6], bool /*marg_this_factor*/ = false) {
    imu_t_.push_back(imu.timestamp);
    for (int c = 0; c < 3; ++c) { imu_gyro_.push_back(imu.gyro[c]); imu_acc_.push_back(imu.accel[c]); gravity_[c] = gravity[c]; }
    for (int c = 0; c < 6; ++c) imu_w_[c] = info_vec[c];
    imu_bias_.push_back(bias_index(gyro_bias, accel_bias));
  }
  // trajectory_estimator.h:109-112 / .cpp:265-291
  void AddBiasFactor(double *bg_i, double *bg_j, double *ba_i, double *ba_j, double dt, const double info_vec[6], bool /*marg*/ = false) {
    bc_i_.push_back(bias_index(bg_i, ba_i));
    bc_j_.push_back(bias_index(bg_j, ba_j));
    const double s = 1.0 / std::sqrt(dt);  // BiasFactor: sqrt_info / sqrt(dt), trajectory_value_factor.h:39-44
    for (int c = 0; c < 6; ++c) bc_w_.push_back(info_vec[c] * s);
  }
  // trajectory_estimator.h:128-131 / .cpp:293-332.  pi / pj: normalised image points (x, y, 1).
  void AddImageFeatureDelayAnalytic(int64_t ti, int rowi, const double pi[3], int64_t tj, int rowj, const double pj[3],
                                    double *inv_depth, double *line_delay, bool /*fixed_depth*/ = false, bool /*marg*/ = false) {
    if (line_delay != &traj_->line_delay) throw std::invalid_argument("line_delay must be &trajectory->line_delay");
    auto it = lm_of_.find(inv_depth);
    int l;
    if (it == lm_of_.end()) { l = (int)lm_ptr_.size(); lm_of_[inv_depth] = l; lm_ptr_.push_back(inv_depth); }
    else l = it->second;
    v_lm_.push_back(l); v_ti_.push_back(ti); v_tj_.push_back(tj); v_rowi_.push_back(rowi); v_rowj_.push_back(rowj);
    v_pi_.push_back(pi[0] / pi[2]); v_pi_.push_back(pi[1] / pi[2]);
    v_pj_.push_back(pj[0] / pj[2]); v_pj_.push_back(pj[1] / pj[2]);
  }
  // trajectory_estimator.h:146-148 / .cpp:334-348: the kept parameter blocks are identified by address.
  void AddMarginalizationFactor(const MarginalizationInfo *info, const std::vector<double *> &parameter_blocks) {
    prior_ = info;
    prior_blocks_ = parameter_blocks;
  }

  // ceres::Solver::Summary Solve(int max_iterations = 50, ...)  trajectory_estimator.h:154-155 / .cpp:367-408
  SolveSummary Solve(int max_iterations = 50, bool /*progress*/ = false, int /*num_threads*/ = -1) {
    const int K = (int)traj_->numKnots(), F = (int)bias_ptr_.size(), L = (int)lm_ptr_.size();
    std::vector<double> quat(4 * (size_t)K), pos(3 * (size_t)K), bias(6 * (size_t)F), rho(L);
    for (int k = 0; k < K; ++k) {
      for (int c = 0; c < 4; ++c) quat[4 * k + c] = traj_->so3_[k][c];
      for (int c = 0; c < 3; ++c) pos[3 * k + c] = traj_->pos_[k][c];
    }
    for (int f = 0; f < F; ++f)
      for (int c = 0; c < 3; ++c) { bias[6 * f + c] = bias_ptr_[f].first[c]; bias[6 * f + 3 + c] = bias_ptr_[f].second[c]; }
    for (int l = 0; l < L; ++l) rho[l] = *lm_ptr_[l];
    ctvio_window w{};
    w.K = K; w.F = F; w.L = L; w.M = (int)imu_t_.size(); w.NB = (int)bc_i_.size(); w.V = (int)v_lm_.size();
    w.t0_ns = traj_->minTimeNs(); w.dt_ns = traj_->getDtNs();
    w.quat = quat.data(); w.pos = pos.data(); w.bias = bias.data(); w.rho = rho.data();
    w.ld = traj_->line_delay; w.ld_lo = traj_->ld_lower; w.ld_hi = traj_->ld_upper; w.fix_ld = traj_->fix_ld;
    w.lock_bg = opt_.lock_wb; w.lock_ba = opt_.lock_ab; w.fixed_upto = opt_.lock_traj ? K - 1 : fixed_upto_;
    for (int c = 0; c < 4; ++c) w.q_CI[c] = traj_->q_CI[c];
    for (int c = 0; c < 3; ++c) { w.p_CI[c] = traj_->p_CI[c]; w.gravity[c] = gravity_[c]; }
    for (int c = 0; c < 6; ++c) w.imu_w[c] = imu_w_[c];
    w.img_w = opt_.image_weight; w.cauchy_a = 2.0;  // CauchyLoss(2), trajectory_estimator.cpp:321-322
    w.imu_t = imu_t_.data(); w.imu_gyro = imu_gyro_.data(); w.imu_acc = imu_acc_.data(); w.imu_bias = imu_bias_.data();
    w.bc_i = bc_i_.data(); w.bc_j = bc_j_.data(); w.bc_w = bc_w_.data();
    w.v_lm = v_lm_.data(); w.v_ti = v_ti_.data(); w.v_tj = v_tj_.data(); w.v_rowi = v_rowi_.data(); w.v_rowj = v_rowj_.data();
    w.v_pi = v_pi_.data(); w.v_pj = v_pj_.data();
    std::vector<int32_t> p_kind, p_index, p_off;
    std::vector<double> p_x0;
    if (prior_ && prior_->n > 0) {
      for (size_t b = 0; b < prior_blocks_.size(); ++b) {
        int kind, index;
        classify(prior_blocks_[b], prior_->keep_block_size[b], kind, index);
        p_kind.push_back(kind); p_index.push_back(index); p_off.push_back(prior_->keep_block_idx[b]);
        for (int c = 0; c < 4; ++c) p_x0.push_back(prior_->keep_block_data[b][c]);
      }
      w.pn = prior_->n; w.pnb = (int)p_kind.size();
      w.pJ0 = prior_->linearized_jacobians.data(); w.pr0 = prior_->linearized_residuals.data();
      w.p_kind = p_kind.data(); w.p_index = p_index.data(); w.p_off = p_off.data(); w.p_x0 = p_x0.data();
    }
    ctvio_options o;
    ctvio_default_options(&o);
    o.precision = opt_.precision; o.device = opt_.device;
    ctvio_solver *s = nullptr;
    check(ctvio_create(&o, &s));
    SolveSummary sum;
    int32_t id = 0;
    int rc = ctvio_add_window(s, &w, &id);
    if (!rc) rc = ctvio_upload(s);
    if (!rc) rc = ctvio_solve(s, max_iterations, &sum.s);
    double ld = traj_->line_delay;
    if (!rc) rc = ctvio_get_state(s, id, quat.data(), pos.data(), bias.data(), rho.data(), &ld);
    const std::string err = rc ? std::string(ctvio_last_error()) : std::string();
    ctvio_destroy(s);
    if (rc) throw std::runtime_error(std::string(ctvio_status_string(rc)) + ": " + err);
    // results back in place, like Ceres writing through the double* (trajectory_manager.cpp:457-463)
    for (int k = 0; k < K; ++k) {
      for (int c = 0; c < 4; ++c) traj_->so3_[k][c] = quat[4 * k + c];
      for (int c = 0; c < 3; ++c) traj_->pos_[k][c] = pos[3 * k + c];
    }
    for (int f = 0; f < F; ++f)
      for (int c = 0; c < 3; ++c) { bias_ptr_[f].first[c] = bias[6 * f + c]; bias_ptr_[f].second[c] = bias[6 * f + 3 + c]; }
    for (int l = 0; l < L; ++l) *lm_ptr_[l] = rho[l];
    traj_->line_delay = ld;
    return sum;
  }

 private:
  static void check(int rc) {
    if (rc) throw std::runtime_error(std::string(ctvio_status_string(rc)) + ": " + ctvio_last_error());
  }
  int bias_index(double *bg, double *ba) {
    auto it = bias_of_.find(bg);
    if (it != bias_of_.end()) return it->second;
    const int f = (int)bias_ptr_.size();
    bias_of_[bg] = f; bias_of_[ba] = f;
    bias_ptr_.push_back({bg, ba});
    return f;
  }
  void classify(double *p, int size, int &kind, int &index) {
    if (size == 4) { kind = CTVIO_PK_ROT; index = knot_of_.at(p); return; }
    if (size == 1) { kind = CTVIO_PK_LD; index = 0; return; }
    for (size_t k = 0; k < traj_->numKnots(); ++k)
      if (traj_->pos_[k].data() == p) { kind = CTVIO_PK_POS; index = (int)k; return; }
    auto it = bias_of_.find(p);
    if (it == bias_of_.end()) throw std::invalid_argument("prior parameter block is not a knot / bias of this window");
    index = it->second;
    kind = (bias_ptr_[index].first == p) ? CTVIO_PK_BG : CTVIO_PK_BA;
  }

  Trajectory *traj_;
  TrajectoryEstimatorOptions opt_;
  int fixed_upto_ = -1;
  std::unordered_map<const double *, int> knot_of_, bias_of_, lm_of_;
  std::vector<std::pair<double *, double *>> bias_ptr_;
  std::vector<double *> lm_ptr_;
  double gravity_[3] = {0, 0, 9.80766}, imu_w_[6] = {250, 250, 250, 12.5, 12.5, 12.5};
  std::vector<int64_t> imu_t_, v_ti_, v_tj_;
  std::vector<double> imu_gyro_, imu_acc_, bc_w_, v_pi_, v_pj_;
  std::vector<int32_t> imu_bias_, bc_i_, bc_j_, v_lm_, v_rowi_, v_rowj_;
  const MarginalizationInfo *prior_ = nullptr;
  std::vector<double *> prior_blocks_;
};

}  // namespace ctvio
