// Test-only shim: include/ctvio_packer.hpp behind a C interface for tests/test_packer.py (compared with ctrl-vio_amd/packer.py).
#include "../include/ctvio_packer.hpp"
#include <cstring>
extern "C" {
void hp_bias_index(int M, const int64_t *imu_t, int F, const int64_t *frame_t, int32_t *out) {
  auto r = ctvio::imu_bias_index(std::vector<int64_t>(imu_t, imu_t + M), std::vector<int64_t>(frame_t, frame_t + F));
  std::memcpy(out, r.data(), sizeof(int32_t) * M);
}
void hp_bias_chain(int M, const int64_t *imu_t, int F, const int64_t *frame_t, double sbg, double sba, double *out) {
  auto r = ctvio::bias_chain_sqrt_info(std::vector<int64_t>(imu_t, imu_t + M), std::vector<int64_t>(frame_t, frame_t + F), sbg, sba);
  std::memcpy(out, r.data(), sizeof(double) * r.size());
}
int64_t hp_opt_min_time(int64_t t, int64_t t0, int64_t dt) { return ctvio::opt_min_time(t, t0, dt); }
// tracks flattened: n_obs[T], start[T], depth[T], points (sum n_obs x 3), uv (sum n_obs x 2); outputs sized by the caller
int hp_pack_visual(int T, const int32_t *n_obs, const int32_t *start, const double *depth, const double *points, const double *uv, int W,
                   const int64_t *timestamps, int32_t *v_lm, int64_t *v_ti, int64_t *v_tj, int32_t *v_rowi, int32_t *v_rowj, double *v_pi,
                   double *v_pj, double *rho, int32_t *owner, int32_t *n_lm) {
  std::vector<ctvio::FeatureTrack> tr(T);
  size_t o = 0;
  for (int t = 0; t < T; ++t) {
    tr[t].start_frame = start[t]; tr[t].depth = depth[t];
    for (int k = 0; k < n_obs[t]; ++k, ++o) {
      tr[t].points.push_back({points[3 * o], points[3 * o + 1], points[3 * o + 2]});
      tr[t].uv.push_back({uv[2 * o], uv[2 * o + 1]});
    }
  }
  auto r = ctvio::pack_visual(tr, std::vector<int64_t>(timestamps, timestamps + W + 1), W);
  const size_t V = r.v_lm.size();
  std::memcpy(v_lm, r.v_lm.data(), 4 * V); std::memcpy(v_ti, r.v_ti.data(), 8 * V); std::memcpy(v_tj, r.v_tj.data(), 8 * V);
  std::memcpy(v_rowi, r.v_rowi.data(), 4 * V); std::memcpy(v_rowj, r.v_rowj.data(), 4 * V);
  std::memcpy(v_pi, r.v_pi.data(), 16 * V); std::memcpy(v_pj, r.v_pj.data(), 16 * V);
  std::memcpy(rho, r.rho.data(), 8 * r.rho.size()); std::memcpy(owner, r.track_of_landmark.data(), 4 * r.rho.size());
  *n_lm = (int32_t)r.rho.size();
  return (int)V;
}
}
