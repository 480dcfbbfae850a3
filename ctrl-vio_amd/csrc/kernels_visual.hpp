// kernels_visual.hpp -- Visual evaluation, factored through the anchor end: k_vis_anchor (one record per anchor), k_vis_eval (blocks, rows of W), and
// k_linearize_f64 (IMU groups and visual waves in one launch, for batches smaller than the chip).
// Part of kernels.hpp (included from there, in order; not a stand-alone header).
#pragma once

namespace ctv {

// ------------------------------------------------------------------------------------------------ visual
// Entry (row = 2 * local column + residual row, < 100; 100 / 101 = the residual) of the robust-corrected 2 x 50 Jacobian of the block
// in slot v with anchor `anc`, rebuilt from the block record and the anchor record (factors.hpp): the cross-check assembly's input.
__device__ __forceinline__ double vis_J_entry(const Dev &d, int row, unsigned v, unsigned anc) {
  const double *J = d.Jt + (size_t)v * VT_ROWS;
  if (row >= 100) return J[VB_RES + row - 100];
  const int col = row >> 1, rr = row & 1;
  if (col >= 48) return J[(col == 48 ? VB_RHO : VB_LD) + rr];
  const double *rec = d.arec + (size_t)anc * AREC;
  if (col < 12) return J[VB_AT + rr] * rec[AR_GR + 3 * col] + J[VB_AT + 2 + rr] * rec[AR_GR + 3 * col + 1] + J[VB_AT + 4 + rr] * rec[AR_GR + 3 * col + 2];
  if (col < 24) { const int c = col - 12; return rec[AR_CP0 + c / 3] * J[VB_AT + 2 * (c % 3) + rr]; }
  if (col < 36) return J[VB_JROT + 2 * (col - 24) + rr];
  const int c = col - 36;
  return -(J[VB_CP1 + c / 3] * J[VB_AT + 2 * (c % 3) + rr]);
}

// time -> (first active knot, u) in integer ns (reference spline_segment.h:83-85); the line delay is
// truncated to integer ns exactly as image_feature_factor.h:72.
__device__ __forceinline__ void vis_times(const WinMeta &m, long long t_rel, int row, double ld, int &s, double &u) {
  const long long ld_ns = (long long)(ld * 1e9);
  const long long tau = t_rel + (long long)row * ld_ns;
  s = (int)(tau / m.dt_ns);
  u = (double)(tau % m.dt_ns) / (double)m.dt_ns;
}

// One lane per ANCHOR (the i end shared by a feature's blocks: factors.hpp): the record of the state being linearised.  The usual wave --
// every knot-pair log of its anchors below 0.5 rad (a ballot) -- takes the series-only evaluation (no branch, no closed-form code on the
// path), the others the general one; both in the global frame.
constexpr int AREC_LD = AREC + 1;   // odd LDS stride of the staged records
__device__ __forceinline__ void vis_anchor_body(const Dev &d, int mode, double *srec /* LDS [64][AREC_LD] */, int block) {
  // the records of the wave's 64 anchors are staged in LDS and written as ONE contiguous region with 16-byte stores (a lane writing
  // its own 400-byte record entry by entry costs 50 scattered partial-line stores: 229 MB of write traffic for 164 MB of records)
  const int a = block * 64 + threadIdx.x;
  bool run = false;
  if (a < d.Atot) run = lin_run(d.lm[d.a_win[a]], mode);
  const unsigned long long run_mask = __ballot(run);
  if (run_mask == 0) return;                   // (wave-uniform)
  if (run) {
  const int w = d.a_win[a];
  const Lm &lm = d.lm[w];
  const WinMeta &m = d.wins[w];
  const bool jac = !lin_cost_only(lm, mode, d.prm);
  const bool at_cand = mode == LIN_SPEC;
  const double *quat = at_cand ? d.cquat : d.quat, *pos = at_cand ? d.cpos : d.pos, *rho = at_cand ? d.crho : d.rho, *ldp = at_cand ? d.cld : d.ld;
  int si;
  double ui;
  const int rowi = d.a_row[a];
  vis_times(m, d.a_t[a], rowi, ldp[w], si, ui);
  si = max(0, min(si, m.K - 4));   // host validated the worst case; clamp keeps loads in range regardless
  SegConstLazy sc;   // Jr^-1 of the knot pairs stays in the table until the streamed Jacobians need it
  seg_const_lazy(d.lkd + 3 * (m.knot0 + si), d.kjri + 9 * (m.knot0 + si), sc);
  double dmax = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) dmax = fmax(dmax, dot(sc.d[i], sc.d[i]));
  const bool small = __ballot(dmax >= 0.25) == 0ull;
  const double *qi = quat + 4 * (m.knot0 + si), *pi = pos + 3 * (m.knot0 + si);
  const Q4 q0 = qmk(qi[0], qi[1], qi[2], qi[3]);
  V3 p[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = mk(pi[3 * i], pi[3 * i + 1], pi[3 * i + 2]);
  const Q4 q_CI = qmk(m.q_CI[0], m.q_CI[1], m.q_CI[2], m.q_CI[3]);
  const V3 p_CI = mk(m.p_CI[0], m.p_CI[1], m.p_CI[2]);
  double *rec = srec + AREC_LD * threadIdx.x;
  if (!jac)                                    // (a costed record carries p_G alone; the rest goes out as zeros, not as stale LDS)
    for (int e = 0; e < AREC; ++e) rec[e] = 0.0;
  const double pix = d.a_obs[a], piy = d.a_obs[(size_t)d.Atot + a], d_inv = rho[m.lm0 + d.a_lm[a]];
  if (small) vis_anchor_eval<true>(q0, p, sc, ui, m.inv_dt, q_CI, p_CI, pix, piy, (double)rowi, d_inv, jac, rec);
  else vis_anchor_eval<false>(q0, p, sc, ui, m.inv_dt, q_CI, p_CI, pix, piy, (double)rowi, d_inv, jac, rec);
  d.a_s[a] = si;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);          // (one wave per workgroup: the wave's own LDS writes have completed)
  __builtin_amdgcn_wave_barrier();
  {
    const int lane = threadIdx.x;
    double *dst = d.arec + (size_t)(block * 64) * AREC;
    constexpr int HP = AREC / 2;                 // pairs per record
#pragma unroll 5
    for (int k = 0; k < HP; ++k) {               // 64 * HP pairs, 64 per store (cost-only records carry p_G alone: the rest is never read)
      const int i = k * 64 + lane, bl = i / HP, r = 2 * (i - bl * HP);
      VecN<double, 2> pr;
      pr.v[0] = srec[AREC_LD * bl + r];
      pr.v[1] = srec[AREC_LD * bl + r + 1];
      if ((run_mask >> bl) & 1ull) *reinterpret_cast<VecN<double, 2> *>(dst + (size_t)bl * AREC + r) = pr;
    }
  }
}
__global__ __launch_bounds__(64) void k_vis_anchor(Dev d, int mode) {
  __shared__ __attribute__((aligned(16))) double srec[64 * AREC_LD];
  vis_anchor_body(d, mode, srec, blockIdx.x);
}

// k_vis_eval<LIN> stages the records of its 64 blocks in LDS, block-major like the copy in HBM ([64][VT_LD]: this lane's block starts
// at J[0]; the odd stride spreads the lanes over the banks) and forms the landmark rows from it after the evaluation.
constexpr int VT_LD = VT_ROWS + 1;
struct VisRecSink {
  double *J;
  __device__ __forceinline__ void put(int e, double v) { J[e] = v; }
};
struct VisNullSink {
  __device__ __forceinline__ void put(int, double) {}
};

// One lane per visual block, landmark-major: evaluate the block's own (j) end against its anchor's record -- r~ and the record of J~
// (robust-corrected), materialised block-major -- and form the rows of W, Hll, g_rho of the wave's landmarks into the normal-equation
// set the mode selects.  The wave's share of the cost goes to Dev::vis_cost (a window's block slots start on a wave boundary: one window
// per wave).  A window on its last allowed iteration is only costed (residuals, no Jacobians, nothing else written).
constexpr int VIS_LDS_BYTES = 64 * VT_LD * 8;   // the records of a wave's 64 blocks, afterwards the fp64 rows of W of the wave's landmarks
__device__ __forceinline__ void vis_eval_body(const Dev &d, int mode, unsigned char *smt, long long *rowoff, int *rlm, int *rspan, int vblock) {
  const int v = vblock * 64 + threadIdx.x;
  const long long t_entry = d.dbg ? clock64() : 0ll;
  constexpr int LDS_BYTES = VIS_LDS_BYTES;
  double *wcs = reinterpret_cast<double *>(smt);
  const bool at_cand = mode == LIN_SPEC;
  const double *quat = at_cand ? d.cquat : d.quat, *pos = at_cand ? d.cpos : d.pos, *ldp = at_cand ? d.cld : d.ld;
  double c = 0.0;
  int ksj = 0, my_lm = -1, my_anc = -1;
  bool on = false, costed = false;
  // The wave's window is known from the block index alone (a window's slots start on a wave boundary: Dev::vb_win): its descriptor, LM state and
  // line delay arrive through the scalar unit while the per-block inputs are on their way -- "slot arrays, then the window, then the knots"
  // used to be three dependent round trips, the window one a vector load of a uniform address per field.
  const int wu = d.vb_win[vblock];
  const Lm &lmw = d.lm[wu];
  const WinMeta &m = d.wins[wu];
  if (!lin_run(lmw, mode)) return;               // (wave-uniform: nothing of this window is evaluated on this pass)
  const bool jac = !lin_cost_only(lmw, mode, d.prm);
  const int tgw = lin_target(lmw, mode);
  if (v < d.Vtot && d.v_win[v] >= 0) {           // (padding slots carry window -1)
    {
      int sj;
      double uj;
      const int rowj = d.v_rowj[v];
      vis_times(m, d.v_tj[v], rowj, ldp[wu], sj, uj);
      sj = max(0, min(sj, m.K - 4));  // host validated the worst case; clamp keeps loads in range regardless
      SegConstLazy scj;   // Jr^-1 of the knot pairs stays in the table until the streamed Jacobians need it
      seg_const_lazy(d.lkd + 3 * (m.knot0 + sj), d.kjri + 9 * (m.knot0 + sj), scj);
      // The usual wave: every knot-pair log of its blocks below 0.5 rad -> series-only evaluation (uniform choice: a ballot over the
      // running lanes); otherwise the general form.  Global frame, absolute positions (fp64).
      double dmax = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) dmax = fmax(dmax, dot(scj.d[i], scj.d[i]));
      const bool small = __ballot(dmax >= 0.25) == 0ull;
      const double *qj = quat + 4 * (m.knot0 + sj), *pj = pos + 3 * (m.knot0 + sj);
      const Q4 q0 = qmk(qj[0], qj[1], qj[2], qj[3]);
      V3 p[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = mk(pj[3 * i], pj[3 * i + 1], pj[3 * i + 2]);
      const int anc = d.v_anc[v];
      const double *rec = reinterpret_cast<const double *>(__builtin_assume_aligned(d.arec + (size_t)anc * AREC, 16));
      M3 RCIT;
      {
        const M3 R = q2R(qmk(m.q_CI[0], m.q_CI[1], m.q_CI[2], m.q_CI[3]));
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) RCIT.m[3 * a + b] = R.m[3 * b + a];
      }
      const V3 p_CI = mk(m.p_CI[0], m.p_CI[1], m.p_CI[2]);
      const double pjx = d.v_obs[v], pjy = d.v_obs[(size_t)d.Vtot + v], ca = d.v_cauchy[v];
      double r[2];
      if (jac) {
        VisRecSink sink{wcs + VT_LD * threadIdx.x};
        on = true;
        my_lm = d.v_lm[v];
        my_anc = anc;
        if (small) c = vis_block_eval<true>(rec, q0, p, scj, uj, m.inv_dt, RCIT, p_CI, m.img_w, ca, pjx, pjy, (double)rowj, r, true, sink);
        else c = vis_block_eval<false>(rec, q0, p, scj, uj, m.inv_dt, RCIT, p_CI, m.img_w, ca, pjx, pjy, (double)rowj, r, true, sink);
        sink.J[VB_RES] = r[0]; sink.J[VB_RES + 1] = r[1];
        d.vsj[v] = sj;
        ksj = sj;
      } else {
        VisNullSink nsink;
        costed = true;
        c = vis_block_eval<false>(rec, q0, p, scj, uj, m.inv_dt, RCIT, p_CI, m.img_w, ca, pjx, pjy, (double)rowj, r, false, nsink);
      }
    }
  }
  {
    const int lane = threadIdx.x;
    const unsigned long long on_mask = __ballot(on);
    if (on_mask != 0 || __any(costed)) {       // the wave's share of the cost: fixed-order sum over the lanes, one store
      double cs = c;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) cs += __shfl_xor(cs, off);
      if (lane == 0) d.vis_cost[vblock] = cs;
    }
    if (on_mask == 0) return;                  // (wave-uniform)
    // the wave's window and its normal-equation set
    const int P = m.P, ldw = m.ldw, lm0 = m.lm0, u0 = m.u0;
    const long long W0 = m.W0;
    double *Wset = d.WS[tgw];
    double *Hllset = d.HllS[tgw], *gset = d.gS[tgw];
    // One wave per workgroup: LDS hand-overs only need the wave's own LDS operations to have completed.  (__syncthreads() also
    // waits for vmcnt(0), i.e. for the J~ and W stores in flight to be acknowledged -- ~5 us per barrier here, measured.)
#define LDS_SYNC() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); } while (0)
    long long *dbg = (d.dbg && vblock == 1000) ? d.dbg + 32 : nullptr;   // (profiling aid: clock stamps of one wave)
    if (dbg && lane == 0) { dbg[-1] = t_entry; dbg[0] = clock64() + (long long)(c * 0); }
    LDS_SYNC();
    // ---- this lane's contributions to its landmark's row of W.  With jr = J~_rho (2) and n3 = A~^T jr (3): the columns of the block's own
    //      (j) end are jr^T J~_rot and -cp1[k] n3; the anchor end's are (sum over the anchor's blocks of n3)^T [GR | cp0 (x) I] -- formed
    //      once per anchor from the record; line delay, Hll, g_rho ride with that sum.
    double wj[24], s6[6];
    {
      const double *Jl = wcs + VT_LD * lane;
      const double jr0 = Jl[VB_RHO], jr1 = Jl[VB_RHO + 1];
#pragma unroll
      for (int cc = 0; cc < 12; ++cc) wj[cc] = jr0 * Jl[VB_JROT + 2 * cc] + jr1 * Jl[VB_JROT + 2 * cc + 1];
#pragma unroll
      for (int b = 0; b < 3; ++b) s6[b] = jr0 * Jl[VB_AT + 2 * b] + jr1 * Jl[VB_AT + 2 * b + 1];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double cp1 = Jl[VB_CP1 + k];
#pragma unroll
        for (int b = 0; b < 3; ++b) wj[12 + 3 * k + b] = -(cp1 * s6[b]);
      }
      s6[3] = jr0 * Jl[VB_LD] + jr1 * Jl[VB_LD + 1];
      s6[4] = jr0 * jr0 + jr1 * jr1;
      s6[5] = jr0 * Jl[VB_RES] + jr1 * Jl[VB_RES + 1];
    }
    // the anchor record's GR and cp0 again (every lane asks for its own anchor's: same lines as during the evaluation; only the head
    // lane of an anchor uses them) -- requested here, consumed after the copy-out below, which hides the round trip
    double hg[40];
    int ksi, my_row, my_klo, my_khi;
    {
      const double *rec = reinterpret_cast<const double *>(__builtin_assume_aligned(d.arec + (size_t)max(my_anc, 0) * AREC, 16));
#pragma unroll
      for (int e = 0; e < 40; ++e) hg[e] = rec[AR_GR + e];     // GR[12][3], cp0[4]: entries 3 .. 42
      ksi = d.a_s[max(my_anc, 0)];
      // the landmark's row of W (sorted landmark order) and its planned knot span (host_pack.hpp: plan_sparsity): the row is formed, and
      // written, over the span's columns only
      my_row = d.lm_pos[lm0 + max(my_lm, 0)];
      my_klo = d.lm_klo[lm0 + my_row]; my_khi = d.lm_khi[lm0 + my_row];
    }
    // ---- rows of W.  A landmark's blocks are consecutive lanes (the host keeps a landmark inside one wave), anchor by anchor.
    const int prev_lm = __shfl_up(my_lm, 1), prev_anc = __shfl_up(my_anc, 1);
    const bool head = on && (lane == 0 || prev_lm != my_lm), head_a = on && (lane == 0 || prev_anc != my_anc);
    const unsigned long long heads = __ballot(head), heads_a = __ballot(head_a);
    const int ord = __popcll(heads & ((2ull << lane) - 1ull)) - 1;     // ordinal of this lane's landmark in the wave
    const int nlm = __popcll(heads);
    const int ha = 63 - __clzll((long long)(heads_a & ((2ull << lane) - 1ull)));       // head lane of this lane's anchor
    int maxlen = on ? lane - ha + 1 : 0;                                               // longest anchor of the wave (uniform)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    for (int off = 1; off < maxlen; off <<= 1) {     // segmented sums over the lanes of an anchor (an LDS atomic of several lanes on ONE
      const int oh = __shfl_down(on ? ha : -1, off);  // address costs ~64 cycles per lane)
      const bool take = on && (lane + off < 64) && oh == ha;
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) { const double o = __shfl_down(s6[cc], off); s6[cc] += take ? o : 0.0; }
    }
    if (dbg && lane == 0) dbg[1] = clock64() + (long long)(s6[0] * 0);
    // ---- the records go out block-major: the 64 x VT_ROWS entries of the wave's blocks are one contiguous region, written as pairs
    //      of entries (16 bytes per lane, 1 KiB per store: under load a store costs ~100 cycles whatever its width, measured)
    {
      double *dst = d.Jt + (size_t)(vblock * 64) * VT_ROWS;
      constexpr int HP = VT_ROWS / 2;            // pairs per block
#pragma unroll 4
      for (int k = 0; k < HP; ++k) {             // 64 * HP pairs, 64 per store
        const int i = k * 64 + lane, bl = i / HP, r = 2 * (i - bl * HP);
        VecN<double, 2> pr;
        pr.v[0] = wcs[VT_LD * bl + r];
        pr.v[1] = wcs[VT_LD * bl + r + 1];
        if ((on_mask >> bl) & 1ull) *reinterpret_cast<VecN<double, 2> *>(dst + (size_t)bl * VT_ROWS + r) = pr;
      }
    }
    if (dbg && lane == 0) dbg[2] = clock64();
    // the anchor end's 24 columns from the segmented sum (used by the head lane of the anchor)
    double wi[24];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int cc = 3 * k + b;
        wi[cc] = s6[0] * hg[3 * cc] + s6[1] * hg[3 * cc + 1] + s6[2] * hg[3 * cc + 2];
        wi[12 + cc] = hg[36 + k] * s6[b];
      }
    // The buffer becomes NR fp64 rows ([0, SPW) the knot columns FROM THE LANDMARK'S FIRST KNOT on -- SPW = 6 x the widest span of the batch --,
    // SPW line delay, SPW + 1 Hll, SPW + 2 g_rho); every lane adds the 24 values of its own end into the row of its landmark (LDS atomics: the
    // ends of different blocks may share knots), the head lane of every anchor the anchor end's 24 + 3; NR landmarks per sweep; then the span's
    // knot columns and the line-delay column of every row, Hll and g_rho are written: W is complete when this kernel ends (the columns
    // outside a landmark's span are never written by anybody: zero since the upload).
    double *rows = reinterpret_cast<double *>(smt);
    const int SPW = d.max_span6;
    const int RS = SPW + 3;                                            // odd row stride (SPW is even)
    const int NR = max(1, min(nlm, ((int)(LDS_BYTES / 8) - 1) / RS));
    if (head) { rowoff[ord] = W0 + (long long)my_row * ldw; rlm[ord] = my_lm; rspan[ord] = (my_klo << 16) | max(my_khi - my_klo + 1, 0); }
    // an end outside the planned span would corrupt a neighbour's row: counted (ctvio_solve fails loudly) and clamped
    int ri = ksi - my_klo, rj = ksj - my_klo;
    if (on && (rj < 0 || ksj + 3 > my_khi || (head_a && (ri < 0 || ksi + 3 > my_khi)))) atomicAdd(d.span_viol, 1);
    ri = max(0, min(ri, SPW / 6 - 4)); rj = max(0, min(rj, SPW / 6 - 4));
    LDS_SYNC();   // every lane has read its record, the copy-out has read them all
    for (int c0 = 0; c0 < nlm; c0 += NR) {
      const int nr = min(NR, nlm - c0);
      for (int i = 2 * lane; i < nr * RS; i += 128) *reinterpret_cast<VecN<double, 2> *>(rows + i) = VecN<double, 2>{{0.0, 0.0}};   // (NR RS + 1 doubles fit)
      LDS_SYNC();
      if (on && ord >= c0 && ord < c0 + nr) {
        double *row = rows + (size_t)(ord - c0) * RS;
        if (head_a) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
              atomicAdd(&row[6 * (ri + k) + b], wi[3 * k + b]);
              atomicAdd(&row[6 * (ri + k) + 3 + b], wi[12 + 3 * k + b]);
            }
          atomicAdd(&row[SPW], s6[3]);
          atomicAdd(&row[SPW + 1], s6[4]);
          atomicAdd(&row[SPW + 2], s6[5]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            atomicAdd(&row[6 * (rj + k) + b], wj[3 * k + b]);
            atomicAdd(&row[6 * (rj + k) + 3 + b], wj[12 + 3 * k + b]);
          }
      }
      LDS_SYNC();
      if (dbg && lane == 0) dbg[3 + 2 * (c0 / NR)] = clock64();
      // write-out: a row's span columns as 16-byte column pairs (a span starts on a 48-byte boundary of a 256-byte aligned row), four rows
      // per store instruction -- one row per 16-lane group
      {
        for (int rb = 0; rb < nr; rb += 4) {
          const int q = rb + (lane >> 4);
          if (q < nr) {
            const int sp = rspan[c0 + q], k0 = sp >> 16, npair = 3 * (sp & 0xffff);
            double *Wr = Wset + rowoff[c0 + q] + 6 * k0;
            const double *row = rows + (size_t)q * RS;
            for (int cp = lane & 15; cp < npair; cp += 16) {
              VecN<double, 2> rv;
              rv.v[0] = row[2 * cp]; rv.v[1] = row[2 * cp + 1];
              *reinterpret_cast<VecN<double, 2> *>(Wr + 2 * cp) = rv;
            }
          }
        }
        if (lane < nr) {                                       // the line-delay column of row `lane`, its Hll and g_rho
          const double *row = rows + (size_t)lane * RS;
          const int l = rlm[c0 + lane];
          Wset[rowoff[c0 + lane] + P - 1] = row[SPW];
          Hllset[lm0 + l] = row[SPW + 1];
          gset[u0 + P + l] = row[SPW + 2];
        }
      }
      LDS_SYNC();
      if (dbg && lane == 0) { dbg[4 + 2 * (c0 / NR)] = clock64(); dbg[10] = nlm * 1000000ll; dbg[11] = NR * 1000; }
    }
#undef LDS_SYNC
  }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_vis_eval(Dev d, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char smt[VIS_LDS_BYTES];
  __shared__ long long rowoff[64];
  __shared__ int rlm[64], rspan[64];
  vis_eval_body(d, mode, smt, rowoff, rlm, rspan, blockIdx.x);
}

// Both evaluations in ONE launch: workgroups [0, Gtot) take an IMU group each, the others a wave of 64 visual block slots.  The two are
// independent; for a batch smaller than the chip their single-wave latencies (23 us each for one window) overlap instead of adding
// up, and large batches lose nothing.  The IMU rows use the head of the visual kernel's LDS buffer.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_linearize_f64(Dev d, int mode, int general_only, int zero_mode) {
  static_assert(VIS_LDS_BYTES >= (72 * 33 + 64) * 8, "the IMU rows use the head of the visual body's LDS buffer");
  __shared__ __attribute__((aligned(32))) unsigned char smt[VIS_LDS_BYTES];
  __shared__ long long rowoff[64];
  __shared__ int rlm[64], rspan[64];
  if ((int)blockIdx.x < d.Gtot) {
    if (!general_only) imu_linearize_f64_fast(d, mode, reinterpret_cast<double *>(smt), blockIdx.x, d.Gtot, zero_mode);   // (one group per wave here)
  } else vis_eval_body(d, mode, smt, rowoff, rlm, rspan, blockIdx.x - d.Gtot);
}

}  // namespace ctv
