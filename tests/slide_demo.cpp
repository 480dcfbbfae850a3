// slide_demo.cpp -- three consecutive sliding windows driven through the reference-shaped C++ adaptor
// (include/ctvio_estimator.hpp) exactly the way the reference's TrajectoryManager drives TrajectoryEstimator:
//   UpdateTrajectory   (src/estimator/trajectory_manager.cpp:317-483): prior + image + IMU + bias factors, Solve(15),
//   double2vector      (:485-516): 4-DoF gauge restore,
//   UpdateVIOPrior     (:122-286, MARGIN_OLD): PrepareMarginalizationInfo / marg_this_factor / SaveMarginalizationInfo.
// Input: the 13-frame world dumped by tests/test_gpu_slide.py (same text layout as tests/test_gpu_adaptor.py); output: the
// live state after the third window.  The protocol is the one of tests/slide_helpers.py, which runs it with the CPU oracle.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "ctvio_estimator.hpp"
#include "ctvio_packer.hpp"

int main(int argc, char **argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s world.txt out.txt\n", argv[0]); return 2; }
  std::ifstream in(argv[1]);
  int K, F, L, M, NB, V, pn, pnb;
  long long t0, dt;
  in >> K >> F >> L >> M >> NB >> V >> pn >> pnb >> t0 >> dt;
  ctvio::Trajectory traj(dt, t0);
  for (int k = 0; k < K; ++k) { double q[4], p[3]; in >> q[0] >> q[1] >> q[2] >> q[3] >> p[0] >> p[1] >> p[2]; traj.knots_push_back(q, p); }
  std::vector<std::array<double, 3>> bg(F), ba(F);   // all_imu_bias_ (trajectory_manager.h:105)
  for (int f = 0; f < F; ++f) in >> bg[f][0] >> bg[f][1] >> bg[f][2] >> ba[f][0] >> ba[f][1] >> ba[f][2];
  std::vector<double> para_Feature(L);
  for (int l = 0; l < L; ++l) in >> para_Feature[l];
  double ld, ld_lo, ld_hi; int fix_ld;
  in >> ld >> ld_lo >> ld_hi >> fix_ld;
  traj.SetLineDelay(ld, fix_ld != 0, ld_lo, ld_hi);
  for (int c = 0; c < 4; ++c) in >> traj.q_CI[c];
  for (int c = 0; c < 3; ++c) in >> traj.p_CI[c];
  double gravity[3], imu_w[6], img_w;
  for (int c = 0; c < 3; ++c) in >> gravity[c];
  for (int c = 0; c < 6; ++c) in >> imu_w[c];
  in >> img_w;
  // the initial (synthetic gauge) prior, blocks by pointer
  ctvio::MarginalizationInfo last_marginalization_info;
  std::vector<double *> last_marginalization_parameter_blocks;
  bool have_prior = pn > 0;
  if (have_prior) {
    auto &mi = last_marginalization_info;
    mi.n = pn; mi.linearized_jacobians.resize((size_t)pn * pn); mi.linearized_residuals.resize(pn);
    for (auto &v : mi.linearized_jacobians) in >> v;
    for (auto &v : mi.linearized_residuals) in >> v;
    for (int b = 0; b < pnb; ++b) {
      int kind, index, off; std::array<double, 4> x0;
      in >> kind >> index >> off >> x0[0] >> x0[1] >> x0[2] >> x0[3];
      mi.keep_block_size.push_back(kind == 0 ? 4 : (kind == 4 ? 1 : 3));
      mi.keep_block_idx.push_back(off);
      mi.keep_block_data.push_back(x0);
      last_marginalization_parameter_blocks.push_back(kind == 0 ? traj.getKnotSO3(index).data() : kind == 1 ? traj.getKnotPos(index).data()
                                                      : kind == 2 ? bg[index].data() : kind == 3 ? ba[index].data() : &traj.line_delay);
    }
  }
  std::vector<ctvio::IMUData> imu(M);
  for (int m = 0; m < M; ++m) {
    long long t; int bias_unused; in >> t; imu[m].timestamp = t;
    in >> imu[m].gyro[0] >> imu[m].gyro[1] >> imu[m].gyro[2] >> imu[m].accel[0] >> imu[m].accel[1] >> imu[m].accel[2] >> bias_unused;
  }
  for (int b = 0; b < NB; ++b) { int i, j; double w6; in >> i >> j; for (int c = 0; c < 6; ++c) in >> w6; }   // recomputed per window below
  struct Obs { int lm, rowi, rowj; long long ti, tj; double pi[3], pj[3]; };
  std::vector<Obs> obs(V);
  std::vector<long long> anchor_t(L, -1);
  for (int v = 0; v < V; ++v) {
    Obs &o = obs[v]; o.pi[2] = o.pj[2] = 1.0;
    in >> o.lm >> o.ti >> o.tj >> o.rowi >> o.rowj >> o.pi[0] >> o.pi[1] >> o.pj[0] >> o.pj[1];
    anchor_t[o.lm] = o.ti;
  }
  const long long FRAME_DT = 100000000LL;
  const int WIN = 11, WINDOW_SIZE = 10, NWIN = 3;

  for (int k = 0; k < NWIN; ++k) {
    int64_t timestamps[WIN];
    std::vector<int64_t> frame_t(WIN);
    for (int i = 0; i < WIN; ++i) timestamps[i] = frame_t[i] = (int64_t)(k + i) * FRAME_DT;
    const int min_idx = (int)traj.computeTIndexNs(timestamps[0]).second;
    const int64_t opt_min_time = (int64_t)min_idx * traj.getDtNs(), opt_max_time = timestamps[WIN - 1];
    const int last_knot = std::min((int)traj.numKnots() - 1, (int)traj.computeTIndexNs(timestamps[WIN - 1] + (int64_t)(0.039 * 1e9)).second + 3);
    std::vector<int64_t> imu_t_win;
    for (const auto &v : imu) if (ctvio::imu_in_window(v.timestamp, opt_min_time, opt_max_time)) imu_t_win.push_back(v.timestamp);
    const std::vector<int32_t> bias_idx = ctvio::imu_bias_index(imu_t_win, frame_t);
    const std::vector<double> sqrt_info_bias = ctvio::bias_chain_sqrt_info(imu_t_win, frame_t, 2.0e-5, 4.0e-4);
    double *para_bg[WIN], *para_ba[WIN];
    for (int i = 0; i < WIN; ++i) { para_bg[i] = bg[k + i].data(); para_ba[i] = ba[k + i].data(); }
    auto candidate = [&](int lm) { const int a = (int)(anchor_t[lm] / FRAME_DT); return a >= k && a - k < WINDOW_SIZE - 2; };

    // ---------------- UpdateTrajectory
    double q0[4], p0[3];
    for (int c = 0; c < 4; ++c) q0[c] = traj.getKnotSO3(min_idx)[c];
    for (int c = 0; c < 3; ++c) p0[c] = traj.getKnotPos(min_idx)[c];
    {
      ctvio::TrajectoryEstimatorOptions option;
      option.image_weight = img_w;
      ctvio::TrajectoryEstimator estimator(&traj, option);
      for (int i = 0; i + 1 < WIN; ++i)   // bias factors first: registers the bias states in frame order
        estimator.AddBiasFactor(para_bg[i], para_bg[i + 1], para_ba[i], para_ba[i + 1], 1.0, &sqrt_info_bias[6 * i]);
      if (have_prior) estimator.AddMarginalizationFactor(&last_marginalization_info, last_marginalization_parameter_blocks);
      for (const Obs &o : obs)
        if (candidate(o.lm) && o.tj <= timestamps[WIN - 1])
          estimator.AddImageFeatureDelayAnalytic(o.ti, o.rowi, o.pi, o.tj, o.rowj, o.pj, &para_Feature[o.lm], &traj.line_delay, false);
      size_t n = 0;
      for (const auto &v : imu)
        if (ctvio::imu_in_window(v.timestamp, opt_min_time, opt_max_time)) {
          estimator.AddIMUMeasurementAnalytic(v, gravity, para_bg[bias_idx[n]], para_ba[bias_idx[n]], imu_w);
          ++n;
        }
      const ctvio::SolveSummary summary = estimator.Solve(15, false);
      std::cout << "window " << k << ": " << summary.BriefReport() << std::endl;
      if (k == 0) std::cout << estimator.GetResidualSummary("after solve").PrintSummary();
    }
    ctvio::gauge_restore_4dof(traj, min_idx, last_knot, q0, p0);   // double2vector
    if (k + 1 == NWIN) break;

    // ---------------- UpdateVIOPrior(MARGIN_OLD)
    {
      ctvio::TrajectoryEstimatorOptions option;
      option.image_weight = img_w;
      option.is_marg_state = true;
      // the selection rules of UpdateVIOPrior(MARGIN_OLD) come from include/ctvio_packer.hpp
      const ctvio::MargOldSelection sel = ctvio::marg_old_selection(traj, timestamps[0], timestamps[1], para_bg[0], para_ba[0],
                                                                    last_marginalization_parameter_blocks);
      option.ctrl_to_be_opt_now = sel.ctrl_to_be_opt_now;
      option.ctrl_to_be_opt_later = sel.ctrl_to_be_opt_later;
      ctvio::TrajectoryEstimator estimator(&traj, option);
      if (have_prior && !sel.drop_set.empty())   // [1] prior: drop the knots in [now, later) and the oldest bias
        estimator.PrepareMarginalizationInfo(&last_marginalization_info, last_marginalization_parameter_blocks, sel.drop_set);
      for (const Obs &o : obs) {   // [2] image: features anchored at the oldest frame are marginalised
        if (!candidate(o.lm) || o.tj > timestamps[WIN - 1]) continue;
        const bool marg_this_factor = ctvio::marg_this_feature((int)(anchor_t[o.lm] / FRAME_DT) - k, para_Feature[o.lm]);
        estimator.AddImageFeatureDelayAnalytic(o.ti, o.rowi, o.pi, o.tj, o.rowj, o.pj, &para_Feature[o.lm], &traj.line_delay, false, marg_this_factor);
      }
      for (const auto &v : imu)   // [3] IMU before the second keyframe
        if (ctvio::imu_in_marg_old(v.timestamp, opt_min_time, timestamps[1]))
          estimator.AddIMUMeasurementAnalytic(v, gravity, para_bg[0], para_ba[0], imu_w, true);
      estimator.AddBiasFactor(para_bg[0], para_bg[1], para_ba[0], para_ba[1], 1.0, &sqrt_info_bias[0], true);   // [4]
      have_prior = estimator.SaveMarginalizationInfo(last_marginalization_info, last_marginalization_parameter_blocks);
      std::cout << "prior " << k << ": n = " << last_marginalization_info.n << ", blocks = " << last_marginalization_parameter_blocks.size() << std::endl;
    }
  }

  std::ofstream out(argv[2]);
  out.precision(17);
  for (int k = 0; k < K; ++k) {
    for (double v : traj.getKnotSO3(k)) out << v << " ";
    for (double v : traj.getKnotPos(k)) out << v << " ";
    out << "\n";
  }
  for (int f = 0; f < F; ++f) out << bg[f][0] << " " << bg[f][1] << " " << bg[f][2] << " " << ba[f][0] << " " << ba[f][1] << " " << ba[f][2] << "\n";
  for (int l = 0; l < L; ++l) out << para_Feature[l] << "\n";
  out << traj.line_delay << "\n";
  // the consumers' side of the solve (trajectory_manager.cpp:108-120, odometry_manager.cpp:287): queries on the optimised spline,
  // evaluated on the device through ctvio::Trajectory
  ctvio::ExtrinsicParam EP_CtoI;
  for (int c = 0; c < 4; ++c) EP_CtoI.q[c] = traj.q_CI[c];
  for (int c = 0; c < 3; ++c) EP_CtoI.p[c] = traj.p_CI[c];
  traj.SetSensorExtrinsics(ctvio::CameraSensor, EP_CtoI);
  traj.SetDataStartTime(12345);
  const int64_t tq = traj.maxTimeNs() - (int64_t)(0.05 * 1e9);
  const ctvio::SE3 cam = traj.GetCameraPose(tq);
  ctvio::IMUState ist;
  traj.GetIMUState(tq, ist);
  const ctvio::SE3 last = traj.getLastKnot();
  traj.setKnot(traj.getKnot(traj.numKnots() - 1), (int)traj.numKnots() - 1);
  out << tq << " " << traj.GetDataStartTime() << "\n";
  for (double v : cam.p) out << v << " ";
  for (double v : cam.q) out << v << " ";
  out << "\n";
  for (double v : ist.p) out << v << " ";
  for (double v : ist.q) out << v << " ";
  for (double v : ist.v) out << v << " ";
  out << "\n";
  for (double v : last.p) out << v << " ";
  for (double v : last.q) out << v << " ";
  out << "\n";
  return 0;
}
