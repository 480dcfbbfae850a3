// estimator_demo.cpp -- drives libctvio.so through the reference-shaped C++ adaptor (include/ctvio_estimator.hpp)
// the way TrajectoryManager::UpdateTrajectory drives TrajectoryEstimator (reference
// src/estimator/trajectory_manager.cpp:350-463).  Input: a window dumped as text by tests/test_gpu_adaptor.py;
// output: the solved state, same text layout.  Built and run by the GPU test only.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <vector>

#include "ctvio_estimator.hpp"

int main(int argc, char **argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s in.txt out.txt max_iters [fp64]\n", argv[0]); return 2; }
  std::ifstream in(argv[1]);
  int K, F, L, M, NB, V, pn, pnb;
  long long t0, dt;
  in >> K >> F >> L >> M >> NB >> V >> pn >> pnb >> t0 >> dt;
  ctvio::Trajectory traj(dt, t0);
  for (int k = 0; k < K; ++k) { double q[4], p[3]; in >> q[0] >> q[1] >> q[2] >> q[3] >> p[0] >> p[1] >> p[2]; traj.knots_push_back(q, p); }
  std::vector<std::array<double, 3>> bg(F), ba(F);   // the reference keeps these in a std::map<int64_t, IMUBias>
  for (int f = 0; f < F; ++f) in >> bg[f][0] >> bg[f][1] >> bg[f][2] >> ba[f][0] >> ba[f][1] >> ba[f][2];
  std::vector<double> para_Feature(L);               // para_Feature[NUM_OF_F][1], trajectory_manager.h:96
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

  ctvio::TrajectoryEstimatorOptions option;
  option.lock_ab = false; option.lock_wb = false; option.image_weight = img_w;
  if (argc > 4) option.precision = CTVIO_FP64;
  ctvio::TrajectoryEstimator estimator(&traj, option);

  // [1] prior
  ctvio::MarginalizationInfo marg;
  std::vector<double *> marg_blocks;
  if (pn > 0) {
    marg.n = pn;
    marg.linearized_jacobians.resize((size_t)pn * pn);
    marg.linearized_residuals.resize(pn);
    for (auto &v : marg.linearized_jacobians) in >> v;   // column-major
    for (auto &v : marg.linearized_residuals) in >> v;
    for (int b = 0; b < pnb; ++b) {
      int kind, index, off; std::array<double, 4> x0;
      in >> kind >> index >> off >> x0[0] >> x0[1] >> x0[2] >> x0[3];
      marg.keep_block_size.push_back(kind == 0 ? 4 : (kind == 4 ? 1 : 3));
      marg.keep_block_idx.push_back(off);
      marg.keep_block_data.push_back(x0);
      double *p = kind == 0 ? traj.getKnotSO3(index).data() : kind == 1 ? traj.getKnotPos(index).data()
                 : kind == 2 ? bg[index].data() : kind == 3 ? ba[index].data() : &traj.line_delay;
      marg_blocks.push_back(p);
    }
  }
  // [3] IMU (bias pointers must be registered in frame order so that indices follow the frames)
  std::vector<ctvio::IMUData> imu(M); std::vector<int> imu_bias(M);
  for (int m = 0; m < M; ++m) {
    long long t; in >> t; imu[m].timestamp = t;
    in >> imu[m].gyro[0] >> imu[m].gyro[1] >> imu[m].gyro[2] >> imu[m].accel[0] >> imu[m].accel[1] >> imu[m].accel[2] >> imu_bias[m];
  }
  // [4] bias chain first: registers bias states 0..F-1 in order (AddBiasFactor(bg_i, bg_j, ba_i, ba_j, 1, sqrt_info))
  for (int b = 0; b < NB; ++b) {
    int i, j; double w6[6];
    in >> i >> j; for (int c = 0; c < 6; ++c) in >> w6[c];
    estimator.AddBiasFactor(bg[i].data(), bg[j].data(), ba[i].data(), ba[j].data(), 1.0, w6);
  }
  if (pn > 0) estimator.AddMarginalizationFactor(&marg, marg_blocks);
  for (int m = 0; m < M; ++m) estimator.AddIMUMeasurementAnalytic(imu[m], gravity, bg[imu_bias[m]].data(), ba[imu_bias[m]].data(), imu_w);
  // [2] image
  for (int v = 0; v < V; ++v) {
    int lm, rowi, rowj; long long ti, tj; double pi[3] = {0, 0, 1}, pj[3] = {0, 0, 1};
    in >> lm >> ti >> tj >> rowi >> rowj >> pi[0] >> pi[1] >> pj[0] >> pj[1];
    estimator.AddImageFeatureDelayAnalytic(ti, rowi, pi, tj, rowj, pj, &para_Feature[lm], &traj.line_delay, false);
  }
  ctvio::SolveSummary summary = estimator.Solve(std::atoi(argv[3]), false);
  std::cout << summary.BriefReport() << std::endl;

  std::ofstream out(argv[2]);
  out.precision(17);
  for (int k = 0; k < K; ++k) {
    for (double v : traj.getKnotSO3(k)) out << v << " ";
    for (double v : traj.getKnotPos(k)) out << v << " ";
    out << "\n";
  }
  for (int f = 0; f < F; ++f) out << bg[f][0] << " " << bg[f][1] << " " << bg[f][2] << " " << ba[f][0] << " " << ba[f][1] << " " << ba[f][2] << "\n";
  for (int l = 0; l < L; ++l) out << para_Feature[l] << "\n";
  out << traj.line_delay << "\n" << summary.s.iterations << " " << summary.s.final_cost << "\n";
  return 0;
}
