// Test shim (CPU): the host-side planning of a window's visual blocks (ctrl-vio_amd/csrc/host_pack.hpp: plan_window) compiled
// with g++ against the HIP headers (no device code, nothing is launched).  Exposes the slot layout so that
// tests/test_host_plan.py can check its invariants.
#define __HIP_PLATFORM_AMD__ 1
#include "../ctrl-vio_amd/csrc/host_pack.hpp"

extern "C" {
// in: V blocks (landmark, ti, tj, rowi, rowj, pi), L landmarks, items of <= vch blocks.  out: Vp, lord[Vp_cap], vpos[V], vord[V], nvitem,
// the number of anchors, anc_of[V] (anchor of every block) and anc_rep[V] (first A entries: a block carrying each anchor).
// returns 0, or 1 when the plan is rejected (err_out gets the message).
int hp_plan(int V, int L, const int32_t *v_lm, const int64_t *v_ti, const int64_t *v_tj, const int32_t *v_rowi, const int32_t *v_rowj,
            const double *v_pi, int vch, int Vp_cap, int32_t *Vp, int32_t *lord, int32_t *vpos, int32_t *vord, int32_t *nvitem, int32_t *A,
            int32_t *anc_of, int32_t *anc_rep, char *err_out, int err_cap) {
  ctvio_window w{};
  w.V = V; w.L = L; w.M = 0;
  w.v_lm = v_lm; w.v_ti = v_ti; w.v_tj = v_tj; w.v_rowi = v_rowi; w.v_rowj = v_rowj; w.v_pi = v_pi;
  w.dt_ns = 1; w.t0_ns = 0;
  ctv::PackTmp t;
  ctv::plan_window(&w, vch, t);
  if (!t.err.empty()) { std::snprintf(err_out, (size_t)err_cap, "%s", t.err.c_str()); return 1; }
  *Vp = t.Vp; *nvitem = t.nvitem;
  if (t.Vp > Vp_cap) return 2;
  for (int i = 0; i < t.Vp; ++i) lord[i] = t.lord[i];
  for (int i = 0; i < V; ++i) { vpos[i] = t.vpos[i]; vord[i] = t.vord[i]; anc_of[i] = t.anc_of[i]; }
  *A = t.A;
  for (int a = 0; a < t.A; ++a) anc_rep[a] = t.anc_rep[a];
  return 0;
}
}

// The sparsity plan of a whole window (host_pack.hpp: plan_sparsity, after plan_window): rows of W in sorted landmark order with their knot
// spans, per-tile row ranges, envelope of the reduced system.  `w` is the C-ABI window as the caller hands it over.
extern "C" int hp_sparsity(const ctvio_window *w, int dense, int full_ranges, int32_t *lm_pos, int32_t *lm_at, int32_t *row_klo, int32_t *row_khi,
                           int32_t *tl_beg, int32_t *tl_end, int32_t *env_first, int32_t *env_tile, int32_t *Lobs, int32_t *max_span, int32_t *ntr, char *err_out, int err_cap) {
  std::string err;
  if (!ctv::validate_window(w, err)) { std::snprintf(err_out, (size_t)err_cap, "%s", err.c_str()); return 1; }
  ctv::PackTmp t;
  ctv::plan_window(w, 8, t);
  if (!t.err.empty()) { std::snprintf(err_out, (size_t)err_cap, "%s", t.err.c_str()); return 1; }
  ctv::plan_sparsity(w, dense != 0, full_ranges != 0, t);
  for (int l = 0; l < w->L; ++l) { lm_pos[l] = t.lm_pos[l]; lm_at[l] = t.lm_at[l]; row_klo[l] = t.row_klo[l]; row_khi[l] = t.row_khi[l]; }
  for (int r = 0; r < t.ntr; ++r) { tl_beg[r] = t.tl_beg[r]; tl_end[r] = t.tl_end[r]; env_first[r] = t.env_first[r]; env_tile[r] = t.env_tile[r]; }
  *Lobs = t.Lobs; *max_span = t.max_span; *ntr = t.ntr;
  return 0;
}
