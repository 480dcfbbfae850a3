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
// Marginalisation side-channel: marg_this_factor flags, PrepareMarginalizationInfo (previous prior + drop set) and
// SaveMarginalizationInfo (-> ctvio_marginalize) build the next window's prior; GetResidualSummary -> ctvio_residual_summary.
// Only the knots the factors touch are shipped (the trajectory may be arbitrarily long); one solver handle per thread is
// kept alive across estimators (SolverCache).
// Not provided (reference methods that are declared but never defined, trajectory_estimator.h:87-143):
// AddPoseMeasurementAnalytic, AddStartTimePose, AddStaticSegment, AddPreIntegrationAnalytic, AddImageFeatureAnalytic,
// AddDelayAnalytic, SetKeyScanConstant; AddCallback (debug hook, never called).
#pragma once

#include <algorithm>
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

// Sophus::SE3d stand-in of this Eigen-free boundary: unit quaternion (x, y, z, w) + translation, with the group product.
struct SE3 {
  double q[4] = {0, 0, 0, 1};
  double p[3] = {0, 0, 0};
  void rotate(const double v[3], double out[3]) const {   // R(q) v
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * (y * v[2] - z * v[1]), ty = 2 * (z * v[0] - x * v[2]), tz = 2 * (x * v[1] - y * v[0]);
    out[0] = v[0] + w * tx + (y * tz - z * ty); out[1] = v[1] + w * ty + (z * tx - x * tz); out[2] = v[2] + w * tz + (x * ty - y * tx);
  }
};
inline SE3 operator*(const SE3 &a, const SE3 &b) {
  SE3 r;
  r.q[0] = a.q[3] * b.q[0] + a.q[0] * b.q[3] + a.q[1] * b.q[2] - a.q[2] * b.q[1];
  r.q[1] = a.q[3] * b.q[1] - a.q[0] * b.q[2] + a.q[1] * b.q[3] + a.q[2] * b.q[0];
  r.q[2] = a.q[3] * b.q[2] + a.q[0] * b.q[1] - a.q[1] * b.q[0] + a.q[2] * b.q[3];
  r.q[3] = a.q[3] * b.q[3] - a.q[0] * b.q[0] - a.q[1] * b.q[1] - a.q[2] * b.q[2];
  const double n = std::sqrt(r.q[0] * r.q[0] + r.q[1] * r.q[1] + r.q[2] * r.q[2] + r.q[3] * r.q[3]);
  for (double &c : r.q) c /= n;
  double t[3];
  a.rotate(b.p, t);
  for (int i = 0; i < 3; ++i) r.p[i] = t[i] + a.p[i];
  return r;
}
// reference src/utils/parameter_struct.h (ExtrinsicParam: so3 / p / q / se3 / t_offset) and IMUState (:67-78)
struct ExtrinsicParam {
  double q[4] = {0, 0, 0, 1};   // sensor -> IMU rotation (x, y, z, w)
  double p[3] = {0, 0, 0};      // sensor origin in the IMU frame
  double t_offset = 0.0;
  SE3 se3() const { SE3 s; for (int i = 0; i < 4; ++i) s.q[i] = q[i]; for (int i = 0; i < 3; ++i) s.p[i] = p[i]; return s; }
};
enum SensorType { IMUSensor = 0, CameraSensor };   // trajectory.h:30-34
struct IMUState {
  int64_t timestamp = 0;
  double p[3] = {0, 0, 0}, v[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1};
};

class SolverCache;

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
  const std::array<double, 4> &getKnotSO3(size_t i) const { return so3_.at(i); }
  const std::array<double, 3> &getKnotPos(size_t i) const { return pos_.at(i); }
  // se3_spline.h:128-144, 212-216, 248-265 (SE3 overloads used at trajectory_manager.cpp:114, 514; odometry_manager.cpp:443)
  SE3 getKnot(size_t i) const { SE3 s; for (int c = 0; c < 4; ++c) s.q[c] = so3_.at(i)[c]; for (int c = 0; c < 3; ++c) s.p[c] = pos_.at(i)[c]; return s; }
  SE3 getLastKnot() const { return getKnot(numKnots() - 1); }
  void setKnot(const SE3 &pose, int i) { setKnotSO3(pose.q, i); setKnotPos(pose.p, i); }
  void setKnotSO3(const double q_xyzw[4], int i) { for (int c = 0; c < 4; ++c) so3_.at(i)[c] = q_xyzw[c]; }
  void setKnotPos(const double p[3], int i) { for (int c = 0; c < 3; ++c) pos_.at(i)[c] = p[c]; }
  void knots_push_back(const SE3 &knot) { knots_push_back(knot.q, knot.p); }
  void extendKnotsTo(int64_t t_ns, const SE3 &initial_knot) { extendKnotsTo(t_ns, initial_knot.q, initial_knot.p); }
  // trajectory.h:64-74: sensor -> IMU extrinsics by sensor type; the camera's also feeds ImageFeatureDelayFactor::S_CtoI / p_CinI
  // (trajectory_manager.cpp:42,51-62), i.e. q_CI / p_CI below
  void SetSensorExtrinsics(SensorType type, const ExtrinsicParam &EP_StoI) {
    EP_StoI_[type] = EP_StoI;
    if (type == CameraSensor) { for (int c = 0; c < 4; ++c) q_CI[c] = EP_StoI.q[c]; for (int c = 0; c < 3; ++c) p_CI[c] = EP_StoI.p[c]; }
  }
  ExtrinsicParam &GetSensorEP(SensorType type) { return EP_StoI_.at(type); }
  std::map<SensorType, ExtrinsicParam> &GetSensorEPs() { return EP_StoI_; }
  void SetDataStartTime(int64_t time) { data_start_time_ = time; }   // trajectory.h:94-96
  int64_t GetDataStartTime() const { return data_start_time_; }
  // ---- trajectory queries (se3_spline.h:361-399, trajectory.cpp:27-55), evaluated ON THE DEVICE: the knots the query times touch are
  //      shipped as a factor-free window to this thread's cached solver handle (ctvio_spline_eval / ctvio_sensor_pose).  A time
  //      outside [minTimeNs, maxTimeNs) throws (the reference asserts).  The batched forms take n times at once.
  void QueryNs(int n, const int64_t *t_ns, double *pose7, double *vel3, double *omega3, double *acc3, const ExtrinsicParam *EP_StoI = nullptr) const;
  SE3 poseNs(int64_t t_ns) const { double o[7]; QueryNs(1, &t_ns, o, nullptr, nullptr, nullptr); return from7(o); }
  void transVelWorld(int64_t t_ns, double v[3]) const { QueryNs(1, &t_ns, nullptr, v, nullptr, nullptr); }
  void rotVelBody(int64_t t_ns, double w[3]) const { QueryNs(1, &t_ns, nullptr, nullptr, w, nullptr); }
  void transAccelWorld(int64_t t_ns, double a[3]) const { QueryNs(1, &t_ns, nullptr, nullptr, nullptr, a); }
  void GetIMUState(int64_t time, IMUState &imu_state) const {   // trajectory.cpp:27-37
    double o[7];
    QueryNs(1, &time, o, imu_state.v, nullptr, nullptr);
    imu_state.timestamp = time;
    for (int c = 0; c < 3; ++c) imu_state.p[c] = o[c];
    for (int c = 0; c < 4; ++c) imu_state.q[c] = o[3 + c];
  }
  SE3 GetSensorPose(int64_t time_ns, const ExtrinsicParam &EP_StoI) const {   // trajectory.cpp:39-56: pose_I_to_G * EP_StoI.se3
    double o[7];
    QueryNs(1, &time_ns, o, nullptr, nullptr, nullptr, &EP_StoI);
    return from7(o);
  }
  SE3 GetCameraPose(int64_t timestamp) const { return GetSensorPose(timestamp, EP_StoI_.at(CameraSensor)); }   // trajectory.h:87-90
  int device = 0;   // HIP device of the queries
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
  static SE3 from7(const double o[7]) { SE3 s; for (int c = 0; c < 3; ++c) s.p[c] = o[c]; for (int c = 0; c < 4; ++c) s.q[c] = o[3 + c]; return s; }
  int64_t dt_ns_, t0_ns_, data_start_time_ = -1;
  std::map<SensorType, ExtrinsicParam> EP_StoI_;
  std::deque<std::array<double, 4>> so3_;
  std::deque<std::array<double, 3>> pos_;
};

// reference src/utils/parameter_struct.h:58-65
struct IMUData {
  int64_t timestamp;
  double gyro[3];
  double accel[3];
};

// reference src/estimator/trajectory_estimator_options.h:34-68 (fields the solve / the marginalisation read)
struct TrajectoryEstimatorOptions {
  bool lock_traj = false, lock_ab = false, lock_wb = false;
  bool is_marg_state = false;                       // :58  the estimator collects a MarginalizationInfo
  int ctrl_to_be_opt_now = 0, ctrl_to_be_opt_later = 0;   // :60-64  knots with index < ctrl_to_be_opt_later are marginalised
  bool show_residual_summary = false;               // :66
  double image_weight = 800.0;  // ImageFeatureDelayFactor::sqrt_info (trajectory_manager.cpp:55-61)
  int precision = CTVIO_FP64;   // all-fp64, like the reference (the only mode)
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
  bool ran_on_host = false;                    // the eigen-decompositions of this prior ran on the host cores (ctvio_marginalize_ran_on_host)
};

// reference trajectory_estimator.h:37-59: per residual type, sum of |r_i| per component and the number of blocks
struct ResidualSummary {
  std::vector<double> imu_sum = std::vector<double>(6, 0.0), bias_sum = std::vector<double>(6, 0.0), image_sum = std::vector<double>(2, 0.0),
                      prior_sum;
  int imu_num = 0, bias_num = 0, image_num = 0, prior_num = 0;
  std::string descri_info;
  std::string PrintSummary() const {
    std::string o = "ResidualSummary :" + descri_info + "\n";
    auto line = [&](const char *name, int num, const std::vector<double> &sum) {
      if (num <= 0) return;
      o += std::string("\t- ") + name + ": num = " + std::to_string(num) + "; err_ave = ";
      for (double v : sum) o += std::to_string(v / num) + ", ";
      o += "\n";
    };
    line("IMU", imu_num, imu_sum); line("Bias", bias_num, bias_sum); line("Image", image_num, image_sum); line("Prior", prior_num, prior_sum);
    return o;
  }
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

// One solver handle (HIP stream + grow-only device / pinned arenas) per (device, precision) and host thread, kept alive
// across estimators: the reference builds a fresh TrajectoryEstimator for every solve (trajectory_manager.cpp:350); here
// that costs no device allocation after the first one.
class SolverCache {
 public:
  static ctvio_solver *get(int device, int precision) {
    thread_local SolverCache cache;
    for (auto &e : cache.items_) if (e.device == device && e.precision == precision) return e.s;
    ctvio_options o;
    ctvio_default_options(&o);
    o.precision = precision; o.device = device; o.host_threads = 1;   // a single window: no packing threads
    ctvio_solver *s = nullptr;
    const int rc = ctvio_create(&o, &s);
    if (rc) throw std::runtime_error(std::string(ctvio_status_string(rc)) + ": " + ctvio_last_error());
    cache.items_.push_back({device, precision, s});
    return s;
  }
  ~SolverCache() { for (auto &e : items_) ctvio_destroy(e.s); }
 private:
  struct Item { int device, precision; ctvio_solver *s; };
  std::vector<Item> items_;
};

inline void Trajectory::QueryNs(int n, const int64_t *t_ns, double *pose7, double *vel3, double *omega3, double *acc3,
                                const ExtrinsicParam *EP_StoI) const {
  if (n <= 0) return;
  int64_t lo = t_ns[0], hi = t_ns[0];
  for (int i = 1; i < n; ++i) { lo = std::min(lo, t_ns[i]); hi = std::max(hi, t_ns[i]); }
  if (lo < minTimeNs() || hi >= maxTimeNs()) throw std::out_of_range("trajectory query time not in [minTimeNs, maxTimeNs)");
  const int k0 = (int)((lo - t0_ns_) / dt_ns_), k1 = (int)((hi - t0_ns_) / dt_ns_) + 3, K = k1 - k0 + 1;
  std::vector<double> quat(4 * (size_t)K), pos(3 * (size_t)K);
  for (int k = 0; k < K; ++k) {
    for (int c = 0; c < 4; ++c) quat[4 * k + c] = so3_[k0 + k][c];
    for (int c = 0; c < 3; ++c) pos[3 * k + c] = pos_[k0 + k][c];
  }
  const double bias0[6] = {0, 0, 0, 0, 0, 0};
  ctvio_window w{};
  w.K = K; w.F = 1; w.t0_ns = t0_ns_ + (int64_t)k0 * dt_ns_; w.dt_ns = dt_ns_;
  w.quat = quat.data(); w.pos = pos.data(); w.bias = bias0; w.fix_ld = 1; w.fixed_upto = -1;
  w.q_CI[3] = 1.0;
  ctvio_solver *s = SolverCache::get(device, CTVIO_FP64);
  auto chk = [](int rc) { if (rc) throw std::runtime_error(std::string(ctvio_status_string(rc)) + ": " + ctvio_last_error()); };
  chk(ctvio_set_batch(s, 1, &w));
  if (EP_StoI) {
    if (vel3 || omega3 || acc3) throw std::invalid_argument("sensor-pose queries return the pose only");
    chk(ctvio_sensor_pose(s, 0, n, t_ns, EP_StoI->q, EP_StoI->p, pose7));
  } else {
    chk(ctvio_spline_eval(s, 0, n, t_ns, pose7, vel3, omega3, acc3));
  }
}

class TrajectoryEstimator {
 public:
  // TrajectoryEstimator(Trajectory::Ptr, TrajectoryEstimatorOptions&)  trajectory_estimator.h:76-77
  TrajectoryEstimator(Trajectory *trajectory, const TrajectoryEstimatorOptions &option) : traj_(trajectory), opt_(option) {
    for (size_t k = 0; k < traj_->numKnots(); ++k) { knot_of_[traj_->so3_[k].data()] = (int)k; knot_of_[traj_->pos_[k].data()] = (int)k; }
    const int64_t ld_ns = (int64_t)((traj_->fix_ld ? traj_->line_delay : std::max(traj_->line_delay, traj_->ld_upper)) * 1e9);
    (void)ld_ns;
  }
  // void SetFixedIndex(int idx)  trajectory_estimator.h:90.  As in the reference, constancy is decided when a factor adds its
  // knots (AddControlPoints, trajectory_estimator.cpp:134-138: SetParameterBlockConstant for knots <= idx, never undone): a knot
  // is constant iff SOME factor touched it while the index (or lock_traj) covered it -- not necessarily a prefix.
  void SetFixedIndex(int idx) { fixed_idx_ = idx; }

  // trajectory_estimator.h:102-106 / .cpp:219-263.  gyro_bias / accel_bias: pointers to 3 doubles; identical pointers
  // denote the same bias state (one per keyframe interval, trajectory_manager.cpp:332-342).
  void AddIMUMeasurementAnalytic(const IMUData &imu, const double gravity[3], double *gyro_bias, double *accel_bias,
                                 const double info_vec[6], bool marg_this_factor = false) {
    imu_t_.push_back(imu.timestamp);
    for (int c = 0; c < 3; ++c) { imu_gyro_.push_back(imu.gyro[c]); imu_acc_.push_back(imu.accel[c]); gravity_[c] = gravity[c]; }
    for (int c = 0; c < 6; ++c) imu_w_[c] = info_vec[c];
    imu_bias_.push_back(bias_index(gyro_bias, accel_bias));
    imu_marg_.push_back(opt_.is_marg_state && marg_this_factor);
    const int s = (int)traj_->computeTIndexNs(imu.timestamp).second;     // CaculateSplineMeta({{t, t}}): 4 knots
    touch(s, s + 3);
  }
  // trajectory_estimator.h:109-112 / .cpp:265-291
  void AddBiasFactor(double *bg_i, double *bg_j, double *ba_i, double *ba_j, double dt, const double info_vec[6], bool marg_this_factor = false) {
    bc_i_.push_back(bias_index(bg_i, ba_i));
    bc_j_.push_back(bias_index(bg_j, ba_j));
    const double s = 1.0 / std::sqrt(dt);  // BiasFactor: sqrt_info / sqrt(dt), trajectory_value_factor.h:39-44
    for (int c = 0; c < 6; ++c) bc_w_.push_back(info_vec[c] * s);
    bc_marg_.push_back(opt_.is_marg_state && marg_this_factor);
  }
  // trajectory_estimator.h:128-131 / .cpp:293-332.  pi / pj: normalised image points (x, y, 1).
  void AddImageFeatureDelayAnalytic(int64_t ti, int rowi, const double pi[3], int64_t tj, int rowj, const double pj[3],
                                    double *inv_depth, double *line_delay, bool /*fixed_depth*/ = false, bool marg_this_feature = false) {
    if (line_delay != &traj_->line_delay) throw std::invalid_argument("line_delay must be &trajectory->line_delay");
    auto it = lm_of_.find(inv_depth);
    int l;
    if (it == lm_of_.end()) { l = (int)lm_ptr_.size(); lm_of_[inv_depth] = l; lm_ptr_.push_back(inv_depth); }
    else l = it->second;
    v_lm_.push_back(l); v_ti_.push_back(ti); v_tj_.push_back(tj); v_rowi_.push_back(rowi); v_rowj_.push_back(rowj);
    v_pi_.push_back(pi[0] / pi[2]); v_pi_.push_back(pi[1] / pi[2]);
    v_pj_.push_back(pj[0] / pj[2]); v_pj_.push_back(pj[1] / pj[2]);
    v_marg_.push_back(opt_.is_marg_state && marg_this_feature);
    v_cauchy_.push_back(marg_this_feature ? 1.0 : 2.0);   // one CauchyLoss per residual block (trajectory_estimator.cpp:320-323)
    const int64_t pad = (int64_t)(0.039 * 1e9);   // spans [t, t + 0.039 s] (trajectory_estimator.cpp:299)
    for (int64_t t : {ti, tj}) touch((int)traj_->computeTIndexNs(t).second, (int)traj_->computeTIndexNs(t + pad).second + 3);
  }
  // trajectory_estimator.h:146-148 / .cpp:334-348: the kept parameter blocks are identified by address.
  void AddMarginalizationFactor(const MarginalizationInfo *info, const std::vector<double *> &parameter_blocks) {
    prior_ = info;
    prior_blocks_ = parameter_blocks;
    touch_prior(parameter_blocks, info);
  }
  // trajectory_estimator.h:158-162 (the RType_Prior overload used by TrajectoryManager::UpdateVIOPrior,
  // trajectory_manager.cpp:161-200): the previous prior enters the marginalisation with `drop_set` = indices into
  // parameter_blocks of the blocks to marginalise.
  void PrepareMarginalizationInfo(const MarginalizationInfo *last_info, const std::vector<double *> &parameter_blocks,
                                  const std::vector<int> &drop_set) {
    marg_prior_ = last_info;
    marg_prior_blocks_ = parameter_blocks;
    marg_prior_drop_ = drop_set;
    touch_prior(parameter_blocks, last_info);
  }

  // ceres::Solver::Summary Solve(int max_iterations = 50, ...)  trajectory_estimator.h:154-155 / .cpp:367-408
  SolveSummary Solve(int max_iterations = 50, bool /*progress*/ = false, int /*num_threads*/ = -1) {
    Packed pk;
    pack(pk, /*marg_only=*/false);
    ctvio_solver *s = SolverCache::get(opt_.device, opt_.precision);
    check(ctvio_set_batch(s, 1, &pk.w));
    SolveSummary sum;
    check(ctvio_solve(s, max_iterations, &sum.s));
    double ld = traj_->line_delay;
    check(ctvio_get_state(s, 0, pk.quat.data(), pk.pos.data(), pk.bias.data(), pk.rho.data(), &ld));
    // results back in place, like Ceres writing through the double* (trajectory_manager.cpp:457-463)
    for (int k = 0; k < pk.w.K; ++k) {
      for (int c = 0; c < 4; ++c) traj_->so3_[pk.kmin + k][c] = pk.quat[4 * k + c];
      for (int c = 0; c < 3; ++c) traj_->pos_[pk.kmin + k][c] = pk.pos[3 * k + c];
    }
    for (int f = 0; f < pk.w.F; ++f)
      for (int c = 0; c < 3; ++c) { bias_ptr_[f].first[c] = pk.bias[6 * f + c]; bias_ptr_[f].second[c] = pk.bias[6 * f + 3 + c]; }
    for (int l = 0; l < pk.w.L; ++l) *lm_ptr_[l] = pk.rho[l];
    traj_->line_delay = ld;
    return sum;
  }

  // trajectory_estimator.h:165-166 / .cpp:178-204: preMarginalize + marginalize of the factors added with
  // marg_this_factor (and the prior given to PrepareMarginalizationInfo).  Dropped: knots below ctrl_to_be_opt_later
  // (PrepareMarginalizationInfo, .cpp:153-176), the bias blocks of the marginalised IMU factors and the first pair of the
  // marginalised bias factors (.cpp:248-257, 281-285), the inverse depths of the marginalised features (.cpp:325-331), the
  // drop_set of the previous prior.  Returns false (marg_info_out untouched, blocks cleared) when nothing is kept.
  bool SaveMarginalizationInfo(MarginalizationInfo &marg_info_out, std::vector<double *> &marg_param_blocks_out) {
    Packed pk;
    pack(pk, /*marg_only=*/true);
    const int K = pk.w.K, F = pk.w.F, L = pk.w.L, P = 6 * K + 6 * F + 1, N = P + L;
    // involved = every parameter block of the collected residual blocks (constant or not); role 1 = drop, 0 = keep
    std::vector<int8_t> role((size_t)N, -1);
    auto involve = [&](int u, int n) { for (int c = 0; c < n; ++c) if (role[u + c] < 0) role[u + c] = 0; };
    auto drop = [&](int u, int n) { for (int c = 0; c < n; ++c) role[u + c] = 1; };
    const int64_t pad = (int64_t)(0.039 * 1e9);
    for (size_t m = 0; m < imu_t_.size(); ++m) {
      if (!imu_marg_[m]) continue;
      const int s0 = (int)traj_->computeTIndexNs(imu_t_[m]).second - pk.kmin;
      involve(6 * s0, 24);
      drop(6 * K + 6 * pk.bias_map[imu_bias_[m]], 6);
    }
    for (size_t b = 0; b < bc_i_.size(); ++b) {
      if (!bc_marg_[b]) continue;
      drop(6 * K + 6 * pk.bias_map[bc_i_[b]], 6);
      involve(6 * K + 6 * pk.bias_map[bc_j_[b]], 6);
    }
    for (size_t v = 0; v < v_lm_.size(); ++v) {
      if (!v_marg_[v]) continue;
      for (int64_t t : {v_ti_[v], v_tj_[v]}) {
        const int s0 = (int)traj_->computeTIndexNs(t).second - pk.kmin, s1 = (int)traj_->computeTIndexNs(t + pad).second - pk.kmin;
        involve(6 * s0, 6 * (s1 + 4 - s0));
      }
      drop(P + pk.lm_map[v_lm_[v]], 1);
      involve(P - 1, 1);
    }
    if (marg_prior_) {
      for (size_t b = 0; b < marg_prior_blocks_.size(); ++b) {
        int kind, index;
        classify(marg_prior_blocks_[b], marg_prior_->keep_block_size[b], kind, index);
        const int u = unknown_of(kind, index, pk), n = kind == CTVIO_PK_LD ? 1 : 3;
        const bool dropped = std::find(marg_prior_drop_.begin(), marg_prior_drop_.end(), (int)b) != marg_prior_drop_.end();
        if (dropped) drop(u, n); else involve(u, n);
      }
    }
    if (opt_.ctrl_to_be_opt_later > opt_.ctrl_to_be_opt_now)
      for (int k = 0; k < K && pk.kmin + k < opt_.ctrl_to_be_opt_later; ++k)
        for (int c = 0; c < 6; ++c) if (role[6 * k + c] >= 0) role[6 * k + c] = 1;
    ctvio_solver *s = SolverCache::get(opt_.device, opt_.precision);
    check(ctvio_set_batch(s, 1, &pk.w));
    std::vector<int32_t> kept((size_t)N);
    std::vector<double> J0((size_t)N * N), r0((size_t)N);
    int32_t n = 0;
    check(ctvio_marginalize(s, 0, role.data(), 1e-8, &n, kept.data(), J0.data(), r0.data()));
    marg_param_blocks_out.clear();
    if (n <= 0) return false;
    MarginalizationInfo out;
    out.n = n;
    out.ran_on_host = ctvio_marginalize_ran_on_host(s) != 0;
    out.linearized_jacobians.resize((size_t)n * n);
    out.linearized_residuals.assign(r0.begin(), r0.begin() + n);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) out.linearized_jacobians[(size_t)j * n + i] = J0[(size_t)i * n + j];   // column-major (Eigen)
    // group the kept unknowns into parameter blocks; the linearisation point is the current value (marginalization_factor.cpp:292-311)
    for (int j = 0; j < n;) {
      const int u = kept[j];
      double *ptr; int size; std::array<double, 4> x0{0, 0, 0, 0};
      if (u == P - 1) { ptr = &traj_->line_delay; size = 1; x0[0] = traj_->line_delay; }
      else if (u < 6 * K) {
        const int k = u / 6 + pk.kmin;
        if (u % 6 == 0) { ptr = traj_->so3_[k].data(); size = 4; for (int c = 0; c < 4; ++c) x0[c] = traj_->so3_[k][c]; }
        else { ptr = traj_->pos_[k].data(); size = 3; for (int c = 0; c < 3; ++c) x0[c] = traj_->pos_[k][c]; }
      } else {
        const int f = pk.bias_unmap[(u - 6 * K) / 6];
        const bool gyro = (u - 6 * K) % 6 == 0;
        ptr = gyro ? bias_ptr_[f].first : bias_ptr_[f].second; size = 3;
        for (int c = 0; c < 3; ++c) x0[c] = ptr[c];
      }
      out.keep_block_size.push_back(size);
      out.keep_block_idx.push_back(j);
      out.keep_block_data.push_back(x0);
      marg_param_blocks_out.push_back(ptr);
      j += (size == 1) ? 1 : 3;
    }
    marg_info_out = std::move(out);
    return true;
  }

  // trajectory_estimator.h:168-171: residuals of ALL added factors at the current parameter values
  ResidualSummary GetResidualSummary(const std::string &descri = "") {
    Packed pk;
    pack(pk, false, /*all_factors=*/true);
    ctvio_solver *s = SolverCache::get(opt_.device, opt_.precision);
    check(ctvio_set_batch(s, 1, &pk.w));
    std::vector<double> sums((size_t)14 + pk.w.pn);
    int32_t cnt[4];
    check(ctvio_residual_summary(s, 0, sums.data(), cnt));
    ResidualSummary r;
    r.descri_info = descri;
    r.imu_sum.assign(sums.begin(), sums.begin() + 6); r.bias_sum.assign(sums.begin() + 6, sums.begin() + 12);
    r.image_sum.assign(sums.begin() + 12, sums.begin() + 14); r.prior_sum.assign(sums.begin() + 14, sums.end());
    r.imu_num = cnt[0]; r.bias_num = cnt[1]; r.image_num = cnt[2]; r.prior_num = cnt[3];
    return r;
  }

 private:
  // One packed window: only the knots the factors touch (the reference registers exactly those with Ceres,
  // trajectory_estimator.cpp:114-141; the trajectory itself keeps growing), biases / landmarks that are referenced.
  struct Packed {
    ctvio_window w{};
    int kmin = 0;
    std::vector<double> quat, pos, bias, rho, imu_gyro, imu_acc, bc_w, v_pi, v_pj, p_x0, v_cauchy;
    std::vector<uint8_t> knot_const;
    std::vector<int64_t> imu_t, v_ti, v_tj;
    std::vector<int32_t> imu_bias, bc_i, bc_j, v_lm, v_rowi, v_rowj, p_kind, p_index, p_off;
    std::vector<int> bias_map, bias_unmap, lm_map;   // adaptor bias / landmark index -> index in this window (-1: absent), and back
  };
  static void check(int rc) {
    if (rc) throw std::runtime_error(std::string(ctvio_status_string(rc)) + ": " + ctvio_last_error());
  }
  // AddControlPoints (trajectory_estimator.cpp:114-141) for knots k0..k1
  void touch(int k0, int k1) {
    kmin_ = std::min(kmin_, k0); kmax_ = std::max(kmax_, k1);
    const int last = std::min(k1, (int)traj_->numKnots() - 1);
    for (int k = std::max(k0, 0); k <= last; ++k)
      if (opt_.lock_traj || (fixed_idx_ >= 0 && k <= fixed_idx_)) {
        if ((int)knot_const_.size() <= k) knot_const_.resize((size_t)k + 1, 0);
        knot_const_[k] = 1;
      }
  }
  // the prior's blocks are handed to AddResidualBlock directly (trajectory_estimator.cpp:334-348): registered, never set constant
  void touch_prior(const std::vector<double *> &blocks, const MarginalizationInfo *info) {
    for (size_t b = 0; b < blocks.size(); ++b)
      if (info->keep_block_size[b] == 4 || knot_of_.count(blocks[b])) {
        const int k = knot_of_.at(blocks[b]);
        kmin_ = std::min(kmin_, k); kmax_ = std::max(kmax_, k);
      }
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
    auto kt = knot_of_.find(p);
    if (kt != knot_of_.end()) { kind = CTVIO_PK_POS; index = kt->second; return; }
    auto it = bias_of_.find(p);
    if (it == bias_of_.end()) throw std::invalid_argument("prior parameter block is not a knot / bias of this window");
    index = it->second;
    kind = (bias_ptr_[index].first == p) ? CTVIO_PK_BG : CTVIO_PK_BA;
  }
  static int unknown_of(int kind, int index, const Packed &pk) {   // index: global knot / adaptor bias index
    const int K = pk.w.K, F = pk.w.F;
    switch (kind) {
      case CTVIO_PK_ROT: return 6 * (index - pk.kmin);
      case CTVIO_PK_POS: return 6 * (index - pk.kmin) + 3;
      case CTVIO_PK_BG: return 6 * K + 6 * pk.bias_map[index];
      case CTVIO_PK_BA: return 6 * K + 6 * pk.bias_map[index] + 3;
      default: return 6 * K + 6 * F;
    }
  }
  // marg_only: the factors flagged marg_this_factor + the prior of PrepareMarginalizationInfo, CauchyLoss(1)
  // (every visual block with the CauchyLoss width it was added with: 1 when marg_this_feature, else 2, trajectory_estimator.cpp:320-323);
  // otherwise every factor + the prior of AddMarginalizationFactor.
  void pack(Packed &pk, bool marg_only, bool all_factors = false) {
    if (kmin_ > kmax_) throw std::logic_error("no factor touches the trajectory");
    const int kmin = std::max(0, kmin_), kmax = std::min((int)traj_->numKnots() - 1, std::max(kmax_, kmin + 3));
    const int K = kmax - kmin + 1;
    pk.kmin = kmin;
    auto want = [&](bool flag) { return all_factors || (marg_only ? flag : true); };
    // bias states and landmarks referenced by the selected factors
    pk.bias_map.assign(bias_ptr_.size(), -1);
    pk.lm_map.assign(lm_ptr_.size(), -1);
    auto use_bias = [&](int f) { if (pk.bias_map[f] < 0) { pk.bias_map[f] = (int)pk.bias_unmap.size(); pk.bias_unmap.push_back(f); } };
    const MarginalizationInfo *prior = marg_only ? marg_prior_ : prior_;
    const std::vector<double *> &pblocks = marg_only ? marg_prior_blocks_ : prior_blocks_;
    if (!marg_only) for (size_t f = 0; f < bias_ptr_.size(); ++f) use_bias((int)f);   // frame order = registration order
    for (size_t b = 0; b < bc_i_.size(); ++b) if (want(bc_marg_[b])) { use_bias(bc_i_[b]); use_bias(bc_j_[b]); }
    for (size_t m = 0; m < imu_t_.size(); ++m) if (want(imu_marg_[m])) use_bias(imu_bias_[m]);
    if (prior)
      for (size_t b = 0; b < pblocks.size(); ++b) {
        int kind, index; classify(pblocks[b], prior->keep_block_size[b], kind, index);
        if (kind == CTVIO_PK_BG || kind == CTVIO_PK_BA) use_bias(index);
      }
    if (pk.bias_unmap.empty()) use_bias(0 < (int)bias_ptr_.size() ? 0 : throw std::logic_error("no bias state registered"));
    for (size_t v = 0; v < v_lm_.size(); ++v)
      if (want(v_marg_[v]) && pk.lm_map[v_lm_[v]] < 0) { pk.lm_map[v_lm_[v]] = (int)pk.rho.size(); pk.rho.push_back(*lm_ptr_[v_lm_[v]]); }
    const int F = (int)pk.bias_unmap.size(), L = (int)pk.rho.size();
    pk.quat.resize(4 * (size_t)K); pk.pos.resize(3 * (size_t)K); pk.bias.resize(6 * (size_t)F);
    for (int k = 0; k < K; ++k) {
      for (int c = 0; c < 4; ++c) pk.quat[4 * k + c] = traj_->so3_[kmin + k][c];
      for (int c = 0; c < 3; ++c) pk.pos[3 * k + c] = traj_->pos_[kmin + k][c];
    }
    for (int f = 0; f < F; ++f)
      for (int c = 0; c < 3; ++c) { pk.bias[6 * f + c] = bias_ptr_[pk.bias_unmap[f]].first[c]; pk.bias[6 * f + 3 + c] = bias_ptr_[pk.bias_unmap[f]].second[c]; }
    for (size_t m = 0; m < imu_t_.size(); ++m) {
      if (!want(imu_marg_[m])) continue;
      pk.imu_t.push_back(imu_t_[m]); pk.imu_bias.push_back(pk.bias_map[imu_bias_[m]]);
      for (int c = 0; c < 3; ++c) { pk.imu_gyro.push_back(imu_gyro_[3 * m + c]); pk.imu_acc.push_back(imu_acc_[3 * m + c]); }
    }
    for (size_t b = 0; b < bc_i_.size(); ++b) {
      if (!want(bc_marg_[b])) continue;
      pk.bc_i.push_back(pk.bias_map[bc_i_[b]]); pk.bc_j.push_back(pk.bias_map[bc_j_[b]]);
      for (int c = 0; c < 6; ++c) pk.bc_w.push_back(bc_w_[6 * b + c]);
    }
    for (size_t v = 0; v < v_lm_.size(); ++v) {
      if (!want(v_marg_[v])) continue;
      pk.v_lm.push_back(pk.lm_map[v_lm_[v]]); pk.v_ti.push_back(v_ti_[v]); pk.v_tj.push_back(v_tj_[v]);
      pk.v_rowi.push_back(v_rowi_[v]); pk.v_rowj.push_back(v_rowj_[v]);
      for (int c = 0; c < 2; ++c) { pk.v_pi.push_back(v_pi_[2 * v + c]); pk.v_pj.push_back(v_pj_[2 * v + c]); }
      pk.v_cauchy.push_back(v_cauchy_[v]);
    }
    ctvio_window &w = pk.w;
    w.K = K; w.F = F; w.L = L; w.M = (int)pk.imu_t.size(); w.NB = (int)pk.bc_i.size(); w.V = (int)pk.v_lm.size();
    w.t0_ns = traj_->minTimeNs() + (int64_t)kmin * traj_->getDtNs(); w.dt_ns = traj_->getDtNs();
    w.quat = pk.quat.data(); w.pos = pk.pos.data(); w.bias = pk.bias.data(); w.rho = pk.rho.data();
    w.ld = traj_->line_delay; w.ld_lo = traj_->ld_lower; w.ld_hi = traj_->ld_upper; w.fix_ld = traj_->fix_ld;
    w.lock_bg = opt_.lock_wb; w.lock_ba = opt_.lock_ab;
    // constant knots: per-knot flags as AddControlPoints set them (any set, not only a prefix)
    w.fixed_upto = -1;
    pk.knot_const.assign((size_t)K, 0);
    for (int k = 0; k < K; ++k) pk.knot_const[k] = (kmin + k < (int)knot_const_.size()) ? knot_const_[kmin + k] : 0;
    w.knot_const = pk.knot_const.data();
    for (int c = 0; c < 4; ++c) w.q_CI[c] = traj_->q_CI[c];
    for (int c = 0; c < 3; ++c) { w.p_CI[c] = traj_->p_CI[c]; w.gravity[c] = gravity_[c]; }
    for (int c = 0; c < 6; ++c) w.imu_w[c] = imu_w_[c];
    w.img_w = opt_.image_weight;
    w.cauchy_a = 2.0;
    w.v_cauchy = pk.v_cauchy.data();   // CauchyLoss(marg_this_feature ? 1 : 2) per block, trajectory_estimator.cpp:320-323
    w.imu_t = pk.imu_t.data(); w.imu_gyro = pk.imu_gyro.data(); w.imu_acc = pk.imu_acc.data(); w.imu_bias = pk.imu_bias.data();
    w.bc_i = pk.bc_i.data(); w.bc_j = pk.bc_j.data(); w.bc_w = pk.bc_w.data();
    w.v_lm = pk.v_lm.data(); w.v_ti = pk.v_ti.data(); w.v_tj = pk.v_tj.data(); w.v_rowi = pk.v_rowi.data(); w.v_rowj = pk.v_rowj.data();
    w.v_pi = pk.v_pi.data(); w.v_pj = pk.v_pj.data();
    if (prior && prior->n > 0) {
      for (size_t b = 0; b < pblocks.size(); ++b) {
        int kind, index;
        classify(pblocks[b], prior->keep_block_size[b], kind, index);
        pk.p_kind.push_back(kind);
        pk.p_index.push_back(kind <= CTVIO_PK_POS ? index - kmin : (kind <= CTVIO_PK_BA ? pk.bias_map[index] : 0));
        pk.p_off.push_back(prior->keep_block_idx[b]);
        for (int c = 0; c < 4; ++c) pk.p_x0.push_back(prior->keep_block_data[b][c]);
      }
      w.pn = prior->n; w.pnb = (int)pk.p_kind.size();
      w.pJ0 = prior->linearized_jacobians.data(); w.pr0 = prior->linearized_residuals.data();
      w.p_kind = pk.p_kind.data(); w.p_index = pk.p_index.data(); w.p_off = pk.p_off.data(); w.p_x0 = pk.p_x0.data();
    }
  }

  Trajectory *traj_;
  TrajectoryEstimatorOptions opt_;
  int fixed_idx_ = -1;
  std::vector<uint8_t> knot_const_;   // per global knot index: SetParameterBlockConstant was called for it
  int kmin_ = 1 << 30, kmax_ = -1;
  std::unordered_map<const double *, int> knot_of_, bias_of_, lm_of_;
  std::vector<std::pair<double *, double *>> bias_ptr_;
  std::vector<double *> lm_ptr_;
  double gravity_[3] = {0, 0, 9.80766}, imu_w_[6] = {250, 250, 250, 12.5, 12.5, 12.5};
  std::vector<int64_t> imu_t_, v_ti_, v_tj_;
  std::vector<double> imu_gyro_, imu_acc_, bc_w_, v_pi_, v_pj_, v_cauchy_;
  std::vector<int32_t> imu_bias_, bc_i_, bc_j_, v_lm_, v_rowi_, v_rowj_;
  std::vector<char> imu_marg_, bc_marg_, v_marg_;
  const MarginalizationInfo *prior_ = nullptr, *marg_prior_ = nullptr;
  std::vector<double *> prior_blocks_, marg_prior_blocks_;
  std::vector<int> marg_prior_drop_;
};

}  // namespace ctvio
