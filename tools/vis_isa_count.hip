// tools/vis_isa_count.hip -- the per-anchor / per-block evaluation bodies (factors.hpp) as stand-alone kernels, compiled to ISA only, so
// that their fp64 instruction counts can be read off (tools/vis_isa_count.sh -> DESIGN.md section 4, bench.py: algorithmic_flops).
#include "../ctrl-vio_amd/csrc/factors.hpp"
using namespace ctv;
struct Sink { double *J; __device__ void put(int e, double v) { J[e] = v; } };
extern "C" __global__ void isa_vis_block_small(const double *rec, const double *kn, const double *kd, const double *kj, const double *cal, double *out) {
  const int i = threadIdx.x;
  SegConstLazy sc;
  seg_const_lazy(kd + 3 * i, kj + 9 * i, sc);
  V3 p[4];
  for (int k = 0; k < 4; ++k) p[k] = mk(kn[64 * (4 + 3 * k) + i], kn[64 * (5 + 3 * k) + i], kn[64 * (6 + 3 * k) + i]);
  M3 R;
  for (int e = 0; e < 9; ++e) R.m[e] = cal[e];
  double r[2];
  Sink s{out + 41 * i};
  const double c = vis_block_eval<true>(rec + 50 * i, qmk(kn[i], kn[64 + i], kn[128 + i], kn[192 + i]), p, sc, cal[20 + i], cal[9], R,
                                        mk(cal[10], cal[11], cal[12]), cal[13], cal[14], cal[15], cal[16], cal[17], r, true, s);
  out[41 * i + 40] = c + r[0] + r[1];
}
extern "C" __global__ void isa_vis_anchor_small(const double *kn, const double *kd, const double *kj, const double *cal, double *rec) {
  const int i = threadIdx.x;
  SegConstLazy sc;
  seg_const_lazy(kd + 3 * i, kj + 9 * i, sc);
  V3 p[4];
  for (int k = 0; k < 4; ++k) p[k] = mk(kn[64 * (4 + 3 * k) + i], kn[64 * (5 + 3 * k) + i], kn[64 * (6 + 3 * k) + i]);
  vis_anchor_eval<true>(qmk(kn[i], kn[64 + i], kn[128 + i], kn[192 + i]), p, sc, cal[20 + i], cal[9], qmk(cal[0], cal[1], cal[2], cal[3]),
                        mk(cal[10], cal[11], cal[12]), cal[13], cal[14], cal[15], cal[16], true, rec + 51 * i);
}
