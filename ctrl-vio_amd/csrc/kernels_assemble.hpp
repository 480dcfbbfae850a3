// kernels_assemble.hpp -- Assembly of the pose block: k_assemble_vis_mfma (fp64 matrix cores, LDS Hessian; STORE = order-fixed tail with k_reduce_finalize /
// k_bias_rows), k_misc (bias chain + prior), k_post_linearize.
// Part of kernels.hpp (included from there, in order; not a stand-alone header).
#pragma once

namespace ctv {

// ---- store-semantics tail of the assembly (product path, windows whose packed Hessian is LDS resident).  Every entry of Hpp / g is
// formed completely by ONE thread and written with a plain store -- the knot x knot block and the line-delay row from the packed
// (LDS or summed-partials) Hessian, the bias rows by a gather over the IMU group tiles of that bias state (fixed order), the bias
// chain and the prior (J0^T J0 looked up through the inverse column map): no pre-zeroing pass, no floating-point atomics, and the
// value does not depend on any execution order.  Entries no factor reaches are zeroed once at upload and never written.
__device__ __forceinline__ double prior_H(const Dev &d, const WinMeta &m, int ga, int gb) {
  if (m.pn <= 0) return 0.0;
  const int pi = d.pinv[m.p0 + ga], pj = d.pinv[m.p0 + gb];
  return (pi >= 0 && pj >= 0) ? d.pH[m.pH0 + (size_t)pi * m.pn + pj] : 0.0;
}
__device__ __forceinline__ double prior_g(const Dev &d, const WinMeta &m, int u) {
  if (m.pn <= 0) return 0.0;
  const int pi = d.pinv[m.p0 + u];
  return pi >= 0 ? d.pgrad[m.pv0 + pi] : 0.0;
}
// packed index i of the knot block / line-delay row -> (row, column) unknowns
__device__ __forceinline__ void packed_decode(int i, int tri, int K6, int P, int &ga, int &gb) {
  if (i < tri) {
    ga = (int)((sqrtf(8.0f * (float)i + 1.0f) - 1.0f) * 0.5f);
    ga += ((ga + 1) * (ga + 2) / 2 <= i) ? 1 : 0;      // the float estimate is off by at most one either way
    ga -= (ga * (ga + 1) / 2 > i) ? 1 : 0;
    gb = i - ga * (ga + 1) / 2;
  } else {
    ga = P - 1;
    gb = (i - tri) < K6 ? (i - tri) : P - 1;
  }
}
// Bias rows (and the bias columns of the line-delay row) of Hpp and the bias entries of g of window w: item e of [0, nitems)
// handled by thread e of a grid-stride loop.  `bias` = the linearisation state (candidate or current).
__device__ __forceinline__ void bias_rows_store(const Dev &d, const WinMeta &m, int tg, const double *bias, int first, int stride) {
  const int K = m.K, F = m.F, P = m.P, K6 = 6 * K, ldh = m.ldh, nbr = 6 * F;
  double *Hg = d.HppS[tg] + m.H0, *g = d.gS[tg] + m.u0;
  const int32_t *boff = d.bgl_off + m.bias0 + (int)(&m - d.wins);   // F + 1 offsets of this window's per-bias group lists
  // items: rows r = K6 .. P - 2 with all columns c <= r (triangle over the bias rows, rectangle over the knot columns), then the
  // line-delay row's bias columns, then the bias entries of g
  const int n_rect = nbr * K6, n_tri = nbr * (nbr + 1) / 2, n_ld = nbr, n_g = nbr;
  for (int e = first; e < n_rect + n_tri + n_ld + n_g; e += stride) {
    if (e < n_rect + n_tri) {
      int rb, c;   // rb: bias row index (0 .. 6F), c: column unknown
      if (e < n_rect) { rb = e / K6; c = e - rb * K6; }
      else {
        const int t = e - n_rect;
        int i2 = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        i2 += ((i2 + 1) * (i2 + 2) / 2 <= t) ? 1 : 0;
        i2 -= (i2 * (i2 + 1) / 2 > t) ? 1 : 0;
        rb = i2; c = K6 + t - i2 * (i2 + 1) / 2;
      }
      const int f = rb / 6, a = rb - 6 * f, r = K6 + rb;
      double v = 0.0;
      const int g0 = boff[f], g1 = boff[f + 1];
      if (c < K6) {
        const int k = c / 6, cc = c - 6 * k;
        for (int q = g0; q < g1; ++q) {
          const int gi = d.bgl[q];
          const int sg = d.groups[gi].s;
          if (k >= sg && k <= sg + 3) v += d.imu_tiles[(size_t)gi * 1024 + (24 + a) * 32 + (cc < 3 ? 3 * (k - sg) + cc : 12 + 3 * (k - sg) + cc - 3)];
        }
      } else {
        const int cb = c - K6, f2 = cb / 6, a2 = cb - 6 * f2;
        if (f2 == f)
          for (int q = g0; q < g1; ++q) v += d.imu_tiles[(size_t)d.bgl[q] * 1024 + (24 + a) * 32 + 24 + a2];
        if (a2 == a)
          for (int b = 0; b < m.NB; ++b) {
            const int bi = d.bc_i[m.bc0 + b], bj = d.bc_j[m.bc0 + b];
            const double wv = d.bc_w[(size_t)(m.bc0 + b) * 6 + a];
            if (f2 == f) { if (bi == f) v += wv * wv; if (bj == f) v += wv * wv; }
            else if ((bi == f2 && bj == f) || (bi == f && bj == f2)) v -= wv * wv;
          }
      }
      Hg[(long long)r * ldh + c] = v + prior_H(d, m, r, c);
    } else if (e < n_rect + n_tri + n_ld) {
      const int c = K6 + e - n_rect - n_tri;
      Hg[(long long)(P - 1) * ldh + c] = prior_H(d, m, P - 1, c);
    } else {
      const int rb = e - n_rect - n_tri - n_ld, f = rb / 6, a = rb - 6 * f, r = K6 + rb;
      double v = 0.0;
      for (int q = boff[f]; q < boff[f + 1]; ++q) v += d.imu_tiles[(size_t)d.bgl[q] * 1024 + (24 + a) * 32 + 30];
      for (int b = 0; b < m.NB; ++b) {
        const int bi = d.bc_i[m.bc0 + b], bj = d.bc_j[m.bc0 + b];
        if (bi != f && bj != f) continue;
        const double wv = d.bc_w[(size_t)(m.bc0 + b) * 6 + a];
        const double rr = wv * (bias[6 * (m.bias0 + bj) + a] - bias[6 * (m.bias0 + bi) + a]);
        if (bi == f) v -= wv * rr;
        if (bj == f) v += wv * rr;
      }
      g[r] = v + prior_g(d, m, r);
    }
  }
}

// Visual assembly on the fp64 matrix cores: gridDim.y workgroups (parts) per window.  The host sorted the visual blocks by frame pair and cut
// them into items of <= CH = 8 blocks; blocks of an item that evaluate on the same knot quadruples (si, sj) form a run, whose product is added
// into an LDS-resident copy of the window's visual Hessian (packed lower triangle over the 6K knot unknowns + the line-delay row).  The two ends
// of a block may share knots (reference image_feature_factor.h:165-180,215,233): every ordered column pair whose unknowns satisfy g(a) >= g(b)
// is added, so shared knots sum correctly.  Landmark terms (W row, Hll, g_rho) are formed by k_vis_eval.  Windows whose packed Hessian does
// not fit in LDS (vis_lds = 0) add straight into Hpp.
//   * the run's [J~_pose]^T [J~_pose] (48 x 48 = 3 x 3 tiles of 16; the lower 6 tiles) is formed with v_mfma_f64_16x16x4_f64: K = 4 is two
//     blocks x two residual rows, the operands are plain LDS reads of the staged item ([116][CH + 2], row = 2 * column + residual row),
//     the A and B operand of a tile pair are the same registers; the line-delay column and the residual ride along as plain FMAs;
//   * the staging area of a wave is private, so there is no workgroup barrier inside the item loop, and the next item's records (15
//     values per lane) are requested before the current item is processed: their latency hides under the products.
// D register r of lane l = D[(l / 16) + 4 r][l % 16] (tools/mfma_f64_layout.hip); eight fp64 staging areas fit beside the packed Hessian.
// LDS staging of one item: [VIS_SROWS][CH + 1] doubles per wave + 2 CH ints of keys.  The host sizes the dynamic LDS request -- and decides
// whether a window keeps its packed Hessian in LDS -- from these same constants (ctvio.hip: SolverImpl::vis_stage_bytes).
constexpr int VIS_SROWS = 116;
constexpr int vis_stage_stride(int ch) { return ch + 1; }
constexpr size_t vis_stage_bytes(int nw, int ch) { return (size_t)nw * VIS_SROWS * vis_stage_stride(ch) * sizeof(double) + (size_t)nw * 2 * ch * sizeof(int); }
__device__ __forceinline__ f64x4 mfma16(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
// NW = waves per workgroup: 8, or 1 in the deterministic mode (all LDS additions of a partial Hessian then come from one wave, in program
// order).  STORE (LDS-resident windows of the product path): store-semantics tail -- with one part the workgroup finishes the window
// itself (prior added on the way out, bias rows gathered); with several parts every part writes its packed partial Hessian to
// Dev::Hpart and k_reduce_finalize sums them in part order.
template <int CH, bool LDSH, int NW = 8, bool STORE = false> __global__ __launch_bounds__(64 * NW) void k_assemble_vis_mfma(Dev d, int mode) {
  typedef f64x4 acc_t;
  // Row stride of a staged item: CH + 1 = 9 doubles.  The MFMA operand reads are ds_read_b64 of lanes (l15, k-row): row 2 (16 I + l15) + rr, so
  // consecutive l15 are 2 x 9 x 2 = 36 four-byte banks apart -- 16 distinct bank pairs, the rr = 1 half on the odd pairs: conflict-free.  With the
  // stride 10 of rounds 1-4 (40 banks apart: period 8) lanes l15 and l15 + 8 met on one bank and every operand read took twice its cycles.
  constexpr int CHP = vis_stage_stride(CH), RPP = 64 / CH, NT = 64 * NW;
  static_assert(!STORE || LDSH, "the store-semantics tail needs the LDS-resident Hessian");
  // staged rows per item: 0..95 pose columns (row = 2 * column + residual row), 98/99 line delay, 100/101 residual, 102..107 A~,
  // 108..111 cp0, 112..115 cp1.  38 of them come from the block records in HBM (rows 48..71 = the j end's rotation columns, 98..107,
  // 112..115) and 4 from the anchor records (cp0); the anchor end's rotation rows 0..23 are A~ times the anchor's GR (9 record entries
  // per lane, held in registers), the 48 position rows (24..47, 72..95) are rebuilt in LDS from rows 102..115; the inverse-depth
  // column is not needed here.
  static_assert(CH == 8, "the staging pattern is written for items of 8 blocks");
  constexpr int SROWS = VIS_SROWS, NEXP = 48 / RPP;
  const long long t_begin = d.dbg ? clock64() : 0;
  const int w = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
  if (!lin_run(d.lm[w], mode) || lin_cost_only(d.lm[w], mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  const int tgset = lin_target(d.lm[w], mode);
  const int P = m.P, K = m.K, nvitem = m.nvitem, vitem0 = m.vitem0, ngrp = m.ngrp, grp0 = m.grp0, u0 = m.u0, ldh = m.ldh;
  if ((m.vis_lds != 0) != LDSH) return;   // the host launches both variants; each window is handled by one of them
  if (!LDSH && m.V == 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smv[];
  // fp64 accumulators: ds_add_f64 sustains ~8 cycles per wave instruction on gfx950, ds_add_f32 ~190 (measured,
  // tools/lds_atomic_bench.hip) -- and the fp64 sums do not depend on the order of the additions to ~1e-16
  double *Hs = reinterpret_cast<double *>(smv);
  const int K6 = 6 * K, tri = K6 * (K6 + 1) / 2;
  const int nHh = LDSH ? tri + K6 + 1 : 0;     // packed Hessian entries: knot x knot lower triangle, line-delay row
  const int nH = nHh + K6 + 1;                 // + gradient of the pose columns (knots, line delay)
  double *gs = Hs + nHh;
  double *stage = reinterpret_cast<double *>(Hs + ((nH + 3) & ~3));     // [NW][SROWS][CHP]
  int *keys = reinterpret_cast<int *>(stage + NW * SROWS * CHP);        // [NW][2][CH]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int per_round = NW * nparts;
  // IMU group tiles (knot x knot part, 24 x 24 per group, overlapping between consecutive segments): the loads of this wave's
  // first NGI groups are issued before the LDS Hessian is zeroed and added right after -- at the end of the kernel they
  // were three exposed memory round trips (24 k of 243 k cycles, measured)
  constexpr int NGI = 3;
  double tv[NGI][9], tgv[NGI];   // tgv: the group's gradient entries of the knot rows (tile column 30), store-semantics tail only
  int gs_[NGI], gb_[NGI];
  if (LDSH) {
#pragma unroll
    for (int u = 0; u < NGI; ++u) {
      const int gi = min(part * NW + wave + u * per_round, max(ngrp - 1, 0));
      const ImuGroup grp = d.groups[grp0 + gi];
      gs_[u] = grp.s; gb_[u] = grp.bias;
      const double *tile = d.imu_tiles + (size_t)(grp0 + gi) * 1024;
#pragma unroll
      for (int q = 0; q < 9; ++q) { const int e = lane + 64 * q; tv[u][q] = ngrp > 0 ? tile[(e / 24) * 32 + e % 24] : 0.0; }
      tgv[u] = (STORE && ngrp > 0) ? tile[min(lane, 23) * 32 + 30] : 0.0;
    }
  }
  for (int i = 2 * tid; i < ((nH + 3) & ~3); i += 2 * NT) *reinterpret_cast<double2 *>(Hs + i) = double2{0.0, 0.0};
  __syncthreads();
  if (LDSH) {
#pragma unroll
    for (int u = 0; u < NGI; ++u) {
      if (part * NW + wave + u * per_round >= ngrp) continue;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int e = lane + 64 * q, a = e / 24, b = e % 24;
        const int ga = imu_col(a, gs_[u], K, gb_[u]), gb = imu_col(b, gs_[u], K, gb_[u]);
        if (ga >= gb) atomicAdd(&Hs[ga * (ga + 1) / 2 + gb], (double)tv[u][q]);
      }
      if (STORE && lane < 24) atomicAdd(&Hs[nHh + imu_col(lane, gs_[u], K, gb_[u])], (double)tgv[u]);   // (gs = Hs + nHh)
    }
  }
  double *Js = stage + wave * SROWS * CHP;
  int *ks = keys + wave * 2 * CH;
  const size_t V = (size_t)d.Vtot;
  // every wave owns a contiguous range of items: a run that continues into the wave's next item keeps its accumulators
  // and is scattered once (the scatter costs as much as the products of an item: ~5 k cycles, measured)
  const int it0 = (int)((long long)nvitem * (part * NW + wave) / per_round);
  const int rounds = (int)((long long)nvitem * (part * NW + wave + 1) / per_round) - it0;
  double *Hg = d.HppS[tgset] + m.H0;
  const int q4 = lane >> 4, l15 = lane & 15, bsel = q4 >> 1, rr = q4 & 1;   // MFMA k index = 2 * (block of the pair) + residual row
  const int sc = lane % CH, srr = lane / CH;                               // staging: column (block) and row parity of this lane
  long long *dbg = (d.dbg && w == (d.nwin > 1000 ? 1000 : 0) && part == 0) ? d.dbg + 48 : nullptr;
  int dbi = 0;
#define CTV_STAMP() do { if (dbg && tid == 0 && dbi < 15) dbg[dbi++] = clock64(); } while (0)
  if (dbg && tid == 0) dbg[dbi++] = t_begin;
  CTV_STAMP();
  // MFMA operand row offsets of this lane: tile row I -> knot column 16 I + l15, at k = q4 (block bsel of the pair, residual
  // row rr).  The line-delay column (staged rows 98, 99) and the residual (rows 100, 101) ride on the same operand
  // values with plain FMAs: every lane multiplies its three J entries by J_ld[k] and r[k] of its own k; the four k
  // groups (lanes l15 + 16 q4) are summed with two shuffles per value at the end of the run.
  int orow[3];
#pragma unroll
  for (int I = 0; I < 3; ++I) orow[I] = (2 * (16 * I + l15) + rr) * CHP;
  const int ldrow = (98 + rr) * CHP, rrow = (100 + rr) * CHP;
  double tj[5], tg9[9], tc0 = 0.0;
  int n = 0, v0 = 0, key_i = 0, key_j = 0;
  // the (start, count) of this wave's items: lane r holds item r, read once -- a per-item load of the descriptor would put a
  // full memory round trip in front of every item's J~ request
  int my_start, my_count;
  {
    const int it = it0 + lane;
    const VisItem I = d.vitems[vitem0 + min(it, max(nvitem - 1, 0))];   // clamped: always a valid descriptor
    my_start = I.start;
    my_count = (lane < rounds && it < nvitem) ? I.count : 0;
  }
  auto item_desc = [&](int r, int &istart, int &icount) {
    if (r < 64) { istart = __shfl(my_start, r); icount = __shfl(my_count, r); }
    else {
      const int it = it0 + r;
      const VisItem I = d.vitems[vitem0 + min(it, nvitem - 1)];
      istart = I.start; icount = (r < rounds && it < nvitem) ? I.count : 0;
    }
    if (r >= rounds) icount = 0;
  };
  // The blocks of an item are slots of the landmark-major evaluation order, listed in Dev::vblk: the slot of this lane's block
  // (c = sc) and of the block whose keys it reads (lane) are requested one item ahead, so that the J~ loads of an item do not
  // wait for its slot list.
  int idn = 0, idkn = 0, ian = 0, iakn = 0;
  auto load_ids = [&](int r) {
    int istart, icount;
    item_desc(r, istart, icount);
    idn = d.vblk[istart + (sc < icount ? sc : 0)];
    ian = d.vblk_anc[istart + (sc < icount ? sc : 0)];
    idkn = d.vblk[istart + (lane < icount ? lane : 0)];
    iakn = d.vblk_anc[istart + (lane < icount ? lane : 0)];
  };
  auto fetch = [&](int r) {   // request item r of this wave: unconditional loads on clamped addresses, masked when staged
    int istart, icount;
    item_desc(r, istart, icount);
    n = icount;
    v0 = istart;
    // The block records are block-major ([slot][VT_ROWS]); lane (sc, srr) takes entries srr + 8 i of its block: 0..23 the j end's
    // rotation columns, 26..33 line delay / residual / A~[0..3], 34..39 A~[4, 5] and cp1 (the inverse-depth entries 24, 25 are
    // skipped) -- the RPP lanes of a block read RPP consecutive entries.  From the anchor record: the three factors GR[c][0..2] of the
    // lane's three anchor-end rotation entries e = srr + 8 i (column c = e / 2) and one of the four cp0.
    const unsigned jb = (unsigned)idn * (unsigned)VT_ROWS + (unsigned)srr;
#pragma unroll
    for (int i = 0; i < 3; ++i) tj[i] = d.Jt[jb + (unsigned)(8 * i)];
    tj[3] = d.Jt[jb + 26u];
    tj[4] = d.Jt[(unsigned)idn * (unsigned)VT_ROWS + (unsigned)min(34 + srr, VT_ROWS - 1)];
    const double *rec = d.arec + (size_t)ian * AREC;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int mm = 0; mm < 3; ++mm) tg9[3 * i + mm] = rec[AR_GR + 3 * ((srr + 8 * i) >> 1) + mm];
    tc0 = rec[AR_CP0 + (srr & 3)];
    key_i = d.a_s[iakn];
    key_j = d.vsj[idkn];
    load_ids(r + 1);
  };
  if (nvitem > 0) load_ids(0);
  if (nvitem > 0) fetch(0);
  // accumulators of the open run (asi, asj): 3 x 3 lower tiles of the 48 x 48 pose block, line-delay column, residual
  acc_t acc[6];
  double pl[3], pr[3], pll, prl;
  int asi = -1, asj = -1;
  auto reset_acc = [&]() {
#pragma unroll
    for (int q = 0; q < 6; ++q) acc[q] = acc_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int I = 0; I < 3; ++I) { pl[I] = 0.0; pr[I] = 0.0; }
    pll = 0.0; prl = 0.0;
  };
  // ---- scatter: local column -> unknown, each unordered local pair once; pairs of different local columns that map
  //      to the same unknown (ends sharing a knot) count twice on the diagonal
  auto scatter = [&](int si, int sj) {
#pragma unroll
    for (int I = 0; I < 3; ++I) {
      pl[I] += __shfl_xor(pl[I], 16); pl[I] += __shfl_xor(pl[I], 32);
      pr[I] += __shfl_xor(pr[I], 16); pr[I] += __shfl_xor(pr[I], 32);
    }
    pll += __shfl_xor(pll, 16); pll += __shfl_xor(pll, 32);
    prl += __shfl_xor(prl, 16); prl += __shfl_xor(prl, 32);
    // unknown index and triangular row offset g (g + 1) / 2 of this lane's 3 tile columns and 12 tile rows, once per run
    int gcol[3], tcol[3], grow[3][4], trow[3][4];
#pragma unroll
    for (int J = 0; J < 3; ++J) {
      gcol[J] = vis_col(16 * J + l15, si, sj, P);
      tcol[J] = gcol[J] * (gcol[J] + 1) / 2;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        grow[J][rg] = vis_col(16 * J + (q4 + 4 * rg), si, sj, P);
        trow[J][rg] = grow[J][rg] * (grow[J][rg] + 1) / 2;
      }
    }
    int q = 0;
#pragma unroll
    for (int I = 0; I < 3; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J, ++q) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int ca = 16 * I + (q4 + 4 * rg), cb = 16 * J + l15;
          if (I == J && ca < cb) continue;
          const int gA = grow[I][rg], gB = gcol[J];
          double hv = acc[q][rg];
          if (gA == gB && ca != cb) hv *= 2.0;
          const bool ge = gA >= gB;
          if (LDSH) atomicAdd(&Hs[(ge ? trow[I][rg] : tcol[J]) + (ge ? gB : gA)], (double)hv);
          else atomicAdd(&Hg[(long long)(ge ? gA : gB) * ldh + (ge ? gB : gA)], (double)hv);
        }
      }
    // line-delay row of the Hessian and the pose gradient (every k group holds the totals; group q4 = 0 adds them)
    if (q4 == 0) {
#pragma unroll
      for (int J = 0; J < 3; ++J) {
        if (LDSH) atomicAdd(&Hs[tri + gcol[J]], (double)pl[J]);
        else atomicAdd(&Hg[(long long)(P - 1) * ldh + gcol[J]], (double)pl[J]);
        atomicAdd(&gs[gcol[J]], (double)pr[J]);
      }
      if (l15 == 0) {   // (ld, ld) and r . J_ld
        if (LDSH) atomicAdd(&Hs[tri + K6], (double)pll);
        else atomicAdd(&Hg[(long long)(P - 1) * ldh + (P - 1)], (double)pll);
        atomicAdd(&gs[K6], (double)prl);
      }
    }
  };
  for (int r = 0; r < rounds && nvitem > 0; ++r) {
    // ---- stage the fetched item (LDS operations of one wave are ordered: no barrier), then request the next one
    const int ncur = n;
    {
      const bool in = sc < ncur;
#pragma unroll
      for (int i = 0; i < 3; ++i) Js[(48 + srr + 8 * i) * CHP + sc] = in ? tj[i] : 0.0;
      Js[(98 + srr) * CHP + sc] = in ? tj[3] : 0.0;                     // rows 98..105: line delay, residual, A~[0..3]
      if (srr < 2) Js[(106 + srr) * CHP + sc] = in ? tj[4] : 0.0;       // A~[4, 5]
      else if (srr < 6) Js[(110 + srr) * CHP + sc] = in ? tj[4] : 0.0;  // cp1 (entries 36..39 -> rows 112..115)
      if (srr < 4) Js[(108 + srr) * CHP + sc] = in ? tc0 : 0.0;         // cp0 (anchor record)
    }
    if (lane < CH) { ks[lane] = key_i; ks[CH + lane] = key_j; }
    __builtin_amdgcn_wave_barrier();
    // anchor end's rotation rows: entry e = srr + 8 i (= 2 * column + residual row) = A~[rr][0..2] . GR[column][0..2]
    {
      const int rr2 = srr & 1;
      const double a0 = Js[(102 + rr2) * CHP + sc], a1 = Js[(104 + rr2) * CHP + sc], a2 = Js[(106 + rr2) * CHP + sc];   // (zero for sc >= ncur)
#pragma unroll
      for (int i = 0; i < 3; ++i) Js[(srr + 8 * i) * CHP + sc] = a0 * tg9[3 * i] + a1 * tg9[3 * i + 1] + a2 * tg9[3 * i + 2];
    }
    // position rows: column 12 + 3 k + b (i end) = cp0[k] P~[b], column 36 + 3 k + b (j end) = -cp1[k] P~[b]
#pragma unroll
    for (int e = 0; e < NEXP; ++e) {
      constexpr int HALF = 24 / RPP;
      const int side = e / HALF, rem = (e % HALF) * RPP + srr;    // rem = 2 * (3 k + b) + residual row
      const int pc = rem >> 1, kk = pc / 3, b = pc - 3 * kk;
      const double pv = Js[(102 + 2 * b + (rem & 1)) * CHP + sc], cv = Js[(108 + 4 * side + kk) * CHP + sc];
      Js[(24 + 48 * side + rem) * CHP + sc] = side ? -(cv * pv) : cv * pv;
    }
    __builtin_amdgcn_wave_barrier();
    if (r < 2) CTV_STAMP();
    fetch(r + 1);
    int start = 0;
    while (start < ncur) {
      const int si = ks[start], sj = ks[CH + start];
      const bool diff = (lane > start && lane < ncur) && (ks[lane] != si || ks[CH + lane] != sj);
      const unsigned long long mask = __ballot(diff);
      const int end = mask ? (__ffsll((long long)mask) - 1) : ncur;
      if (asi != si || asj != sj) {     // a run that continues from the previous item keeps accumulating
        if (asi >= 0) scatter(asi, asj);
        reset_acc();
        asi = si; asj = sj;
      }
      // 4 K-steps (8 blocks) per trip: the operand reads first, then the products -- one LDS latency per trip
      for (int v8 = start; v8 < end; v8 += 8) {
        double a[4][3], ldv[4], rv[4];
        // (the usual item is one run that ends with the item: the columns past it were staged as zeros, nothing to mask)
        const bool nomask = end == ncur && v8 + 8 <= CH;     // uniform
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int blk = v8 + 2 * s + bsel;
          const int bc = min(blk, CH);       // a valid LDS address even when past the run (value discarded)
#pragma unroll
          for (int I = 0; I < 3; ++I) a[s][I] = Js[orow[I] + bc];
          ldv[s] = Js[ldrow + bc];
          rv[s] = Js[rrow + bc];
        }
        if (!nomask) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bool in = v8 + 2 * s + bsel < end;
#pragma unroll
            for (int I = 0; I < 3; ++I) a[s][I] = in ? a[s][I] : 0.0;
            ldv[s] = in ? ldv[s] : 0.0;
            rv[s] = in ? rv[s] : 0.0;
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (v8 + 2 * s >= end) break;     // uniform
          int q = 0;
#pragma unroll
          for (int I = 0; I < 3; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J, ++q) acc[q] = mfma16(a[s][I], a[s][J], acc[q]);
#pragma unroll
          for (int I = 0; I < 3; ++I) { pl[I] += a[s][I] * ldv[s]; pr[I] += a[s][I] * rv[s]; }
          pll += ldv[s] * ldv[s];
          prl += rv[s] * ldv[s];
        }
      }
      if (r < 2) CTV_STAMP();
      start = end;
    }
    if (r < 2) CTV_STAMP();
  }
  if (asi >= 0) scatter(asi, asj);
  __syncthreads();
  CTV_STAMP();
  // IMU group tiles: the knot x knot part (24 x 24 per group, overlapping between consecutive segments); without the LDS
  // Hessian k_assemble_imu adds them
  for (int gi = part * NW + wave + NGI * per_round; LDSH && gi < ngrp; gi += per_round) {   // groups beyond the prefetched ones
    const ImuGroup grp = d.groups[grp0 + gi];
    const double *tile = d.imu_tiles + (size_t)(grp0 + gi) * 1024;
    double tv[9];   // 24 x 24 = 9 x 64 entries: all loads in flight together
#pragma unroll
    for (int u = 0; u < 9; ++u) { const int e = lane + 64 * u; tv[u] = tile[(e / 24) * 32 + e % 24]; }
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int e = lane + 64 * u, a = e / 24, b = e % 24;
      const int ga = imu_col(a, grp.s, K, grp.bias), gb = imu_col(b, grp.s, K, grp.bias);
      if (ga >= gb) atomicAdd(&Hs[ga * (ga + 1) / 2 + gb], (double)tv[u]);
    }
    if (STORE && lane < 24) atomicAdd(&gs[imu_col(lane, grp.s, K, grp.bias)], (double)tile[lane * 32 + 30]);
  }
  __syncthreads();
  CTV_STAMP();
  if constexpr (STORE) {
    if (nparts > 1) {   // this part's packed Hessian + gradient: summed with the others, in part order, by k_reduce_finalize
      double *dst = d.Hpart + ((size_t)w * nparts + part) * d.npart_stride;
      for (int i = tid; i < nH; i += NT) dst[i] = Hs[i];
      return;
    }
    double *gq = d.gS[tgset] + u0;
    for (int i0 = tid; i0 < nHh; i0 += 4 * NT) {     // 4 entries per trip: the LDS reads first, then decode + prior + store
      double hv4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) hv4[u] = Hs[min(i0 + NT * u, nHh - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + NT * u;
        if (i >= nHh) continue;
        int ga, gb;
        packed_decode(i, tri, K6, P, ga, gb);
        Hg[(long long)ga * ldh + gb] = hv4[u] + prior_H(d, m, ga, gb);
      }
    }
    for (int i = tid; i < K6 + 1; i += NT) { const int uu = i < K6 ? i : P - 1; gq[uu] = gs[i] + prior_g(d, m, uu); }
    return;   // (the bias rows: k_bias_rows -- a gather with dependent loads wants more waves per CU than this kernel's LDS allows)
  }
  for (int i0 = tid; LDSH && i0 < nHh; i0 += 4 * NT) {     // 4 entries per trip: the LDS reads first, then decode + store
    double hv4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) hv4[u] = Hs[min(i0 + NT * u, nHh - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + NT * u;
      const double hv = hv4[u];
      if (i >= nHh || (nparts > 1 && hv == 0.0)) continue;
      int ga, gb;
      packed_decode(i, tri, K6, P, ga, gb);
      if (nparts > 1) atomicAdd(&Hg[(long long)ga * ldh + gb], hv);
      else Hg[(long long)ga * ldh + gb] = hv;  // first writer after k_zero_normal; later kernels add atomically
    }
  }
  CTV_STAMP();
  for (int i = tid; i < K6 + 1; i += NT) {
    const double gv = gs[i];
    if (gv != 0.0) atomicAdd(&d.gS[tgset][u0 + (i < K6 ? i : P - 1)], gv);
  }
  CTV_STAMP();
#undef CTV_STAMP
}

// Bias rows of the single-part store-semantics assembly: grid (blocks, windows).
__global__ __launch_bounds__(256) void k_bias_rows(Dev d, int mode) {
  const int w = blockIdx.y;
  if (!lin_run(d.lm[w], mode) || lin_cost_only(d.lm[w], mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  if (!m.vis_lds) return;
  bias_rows_store(d, m, lin_target(d.lm[w], mode), mode == LIN_SPEC ? d.cbias : d.bias, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// Several parts per window (batches smaller than the chip; the deterministic mode): sum of the parts' packed Hessians in part
// order, prior added, plain stores; the bias rows by gather.  Grid (blocks, windows).
__global__ __launch_bounds__(256) void k_reduce_finalize(Dev d, int mode, int nparts) {
  const int w = blockIdx.y;
  if (!lin_run(d.lm[w], mode) || lin_cost_only(d.lm[w], mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  if (!m.vis_lds) return;
  const int tg = lin_target(d.lm[w], mode);
  const int P = m.P, K6 = 6 * m.K, tri = K6 * (K6 + 1) / 2, nHh = tri + K6 + 1, nH = nHh + K6 + 1, ldh = m.ldh;
  double *Hg = d.HppS[tg] + m.H0, *g = d.gS[tg] + m.u0;
  const double *src = d.Hpart + (size_t)w * nparts * d.npart_stride;
  const int first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (int i = first; i < nH; i += stride) {
    double v = 0.0;
    int p = 0;
    for (; p + 8 <= nparts; p += 8) {   // eight loads in flight, added in part order
      double t8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t8[q] = src[(size_t)(p + q) * d.npart_stride + i];
#pragma unroll
      for (int q = 0; q < 8; ++q) v += t8[q];
    }
    for (; p < nparts; ++p) v += src[(size_t)p * d.npart_stride + i];
    if (i < nHh) {
      int ga, gb;
      packed_decode(i, tri, K6, P, ga, gb);
      Hg[(long long)ga * ldh + gb] = v + prior_H(d, m, ga, gb);
    } else {
      const int uu = (i - nHh) < K6 ? (i - nHh) : P - 1;
      g[uu] = v + prior_g(d, m, uu);
    }
  }
  bias_rows_store(d, m, tg, mode == LIN_SPEC ? d.cbias : d.bias, first, stride);
}

// ------------------------------------------------------------------------------------------------ bias chain + prior
__device__ __forceinline__ const double *prior_block_ptr(const WinMeta &m, int kind, int idx, const double *quat, const double *pos,
                                                         const double *bias, const double *ldp, int w) {
  switch (kind) {
    case 0: return quat + 4 * (m.knot0 + idx);
    case 1: return pos + 3 * (m.knot0 + idx);
    case 2: return bias + 6 * (m.bias0 + idx);
    case 3: return bias + 6 * (m.bias0 + idx) + 3;
    default: return ldp + w;
  }
}

// BiasFactor (trajectory_value_factor.h:45-99) and MarginalizationFactor (marginalization_factor.cpp:326-373), fp64.
// With the prior written r = r0 + J0 dx:  J^T r = J0^T r0 + (J0^T J0) dx,  |r|^2 = r0^T r0 + 2 b0.dx + dx^T (J0^T J0) dx.
// Adds to Hpp / g of the set the mode selects (not on a cost-only pass) and stores the window's cost share (Dev::misc_cost).
// store != 0 (store-semantics assembly tail): nothing is added here -- the prior's gradient J0^T r0 + (J0^T J0) dx goes to Dev::pgrad,
// and the assembly looks the prior and the chain up when it writes each entry.
// with_imu != 0: the window's IMU group tiles are scattered first (assemble_imu_window: the accumulate path's k_assemble_imu, fused).
// NT threads per workgroup: 256 (k_misc), or 64 when the store-semantics part (prior gradient + cost share, with_imu = 0) rides in k_pre_linearize
template <int NT> __device__ __forceinline__ void misc_body(const Dev &d, int mode, int store, int with_imu, int w, double *smd /* LDS [pn (+ band)] */, double *red /* LDS [NT] */) {
  // with_imu == 2: the launch carries 144 maxK doubles of LDS behind the prior's dx for the band of the IMU knot blocks (windows with K > 24)
  if (with_imu) { assemble_imu_window(d, mode, w, with_imu == 2 ? smd + ((max(d.maxPn, 1) + 1) & ~1) : nullptr); __syncthreads(); }
  const Lm &lm = d.lm[w];
  if (!lin_run(lm, mode)) return;
  const bool LIN = !lin_cost_only(lm, mode, d.prm) && !store;
  const WinMeta &m = d.wins[w];
  const bool at_cand = mode == LIN_SPEC;
  const double *quat = at_cand ? d.cquat : d.quat, *pos = at_cand ? d.cpos : d.pos, *bias = at_cand ? d.cbias : d.bias, *ldp = at_cand ? d.cld : d.ld;
  const int tg = lin_target(lm, mode);
  double *Hpp = d.HppS[tg] + m.H0, *g = d.gS[tg] + m.u0;
  double *dx = smd;                 // [pn]
  const int tid = threadIdx.x;
  double cost = 0.0;
  for (int e = tid; e < m.NB * 6; e += NT) {
    const int b = e / 6, k = e % 6;
    const int bi = d.bc_i[m.bc0 + b], bj = d.bc_j[m.bc0 + b];
    const double wv = d.bc_w[(size_t)(m.bc0 + b) * 6 + k];
    const double r = wv * (bias[6 * (m.bias0 + bj) + k] - bias[6 * (m.bias0 + bi) + k]);
    cost += 0.5 * r * r;
    if (LIN) {
      const int ii = 6 * m.K + 6 * bi + k, jj = 6 * m.K + 6 * bj + k;
      atomicAdd(&g[ii], -wv * r);
      atomicAdd(&g[jj], wv * r);
      atomicAdd(&Hpp[(long long)ii * m.ldh + ii], wv * wv);
      atomicAdd(&Hpp[(long long)jj * m.ldh + jj], wv * wv);
      const int hi = max(ii, jj), lo = min(ii, jj);
      atomicAdd(&Hpp[(long long)hi * m.ldh + lo], -wv * wv);
    }
  }
  const int n = m.pn;
  if (n > 0) {
    for (int i = tid; i < n; i += NT) dx[i] = 0.0;
    __syncthreads();
    for (int b = tid; b < m.pnb; b += NT) {
      const int kind = d.p_kind[m.pblk0 + b], idx = d.p_index[m.pblk0 + b], off = d.p_off[m.pblk0 + b];
      const double *x = prior_block_ptr(m, kind, idx, quat, pos, bias, ldp, w);
      const double *x0 = d.p_x0 + 4 * (size_t)(m.pblk0 + b);
      if (kind == 0) {  // dx = 2 vec(q0^-1 q), sign-fixed (marginalization_factor.cpp:344-350)
        const Q4 dq = qmul_raw(qmk(-x0[0], -x0[1], -x0[2], x0[3]), qmk(x[0], x[1], x[2], x[3]));
        const double sg = (dq.w >= 0) ? 2.0 : -2.0;
        dx[off] = sg * dq.x; dx[off + 1] = sg * dq.y; dx[off + 2] = sg * dq.z;
      } else {
        const int sz = (kind == 4) ? 1 : 3;
        for (int k = 0; k < sz; ++k) dx[off + k] = x[k] - x0[k];
      }
    }
    __syncthreads();
    const double *pH = d.pH + m.pH0, *b0 = d.pb0 + m.pv0;
    const int *pcol = d.pcol + m.pv0;
    for (int i = tid; i < n; i += NT) {
      double hd = 0.0;
      for (int j = 0; j < n; ++j) hd += pH[(size_t)j * n + i] * dx[j];   // (J0^T J0 is symmetric: column i, coalesced over the threads)
      cost += dx[i] * (b0[i] + 0.5 * hd);
      if (store) d.pgrad[m.pv0 + i] = b0[i] + hd;
      if (LIN && pcol[i] >= 0) atomicAdd(&g[pcol[i]], b0[i] + hd);
    }
    if (tid == 0) cost += 0.5 * d.pc0[w];
    if (LIN) {
      for (int e = tid; e < n * n; e += NT) {
        const int i = e / n, j = e % n;
        const int ci = pcol[i], cj = pcol[j];
        if (ci >= 0 && cj >= 0 && ci >= cj) atomicAdd(&Hpp[(long long)ci * m.ldh + cj], pH[e]);
      }
    }
  }
  red[tid] = cost;
  __syncthreads();
  for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
  if (tid == 0) {
    d.misc_cost[w] = red[0];
    if (store && !lin_cost_only(lm, mode, d.prm)) {   // (the generic path resets these in k_zero_normal)
      if (mode == LIN_SPEC) d.lm[w].cand_gmax_bits = 0ull; else d.lm[w].gmax_bits = 0ull;
    }
  }
}
__global__ __launch_bounds__(256) void k_misc(Dev d, int mode, int store, int with_imu) {
  extern __shared__ __attribute__((aligned(16))) double smd[];
  __shared__ double red[256];
  misc_body<256>(d, mode, store, with_imu, blockIdx.x, smd, red);
}
// Small batches (the merged linearisation, <= 128 windows): everything that must precede k_linearize_f64 or is independent of it in ONE launch of
// 64-thread workgroups -- [0, nA) the anchors' records (k_vis_anchor), [nA, nA + nwin) the IMU groups the specialised body leaves out
// (k_imu_linearize_rest; nothing to do as a rule), and, on the store-semantics path, [nA + nwin, nA + 2 nwin) the prior gradient + cost share
// (k_misc with store = 1).  For one window these were three launches of 5 - 7 us each in a pass of 180 us (the reference's operating mode:
// one UpdateTrajectory per image, odometry_manager.cpp:268-277); their latencies now overlap.  The general IMU body sets the register
// allocation (one wave per SIMD), which a batch smaller than the chip does not feel.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_pre_linearize(Dev d, int mode, int general_only, int zero_mode, int n_anchor_blocks, int with_misc) {
  __shared__ __attribute__((aligned(32))) double srec[64 * AREC_LD];
  static_assert(64 * AREC_LD >= 64 * 33, "the general IMU body's row buffer uses the anchor records' staging area");
  const int b = blockIdx.x;
  if (b < n_anchor_blocks) { vis_anchor_body(d, mode, srec, b); return; }
  if (b < n_anchor_blocks + d.nwin) { imu_rest_body(d, mode, general_only, zero_mode, reinterpret_cast<unsigned char *>(srec), b - n_anchor_blocks); return; }
  if (with_misc) misc_body<64>(d, mode, 1, 0, b - n_anchor_blocks - d.nwin, srec, srec + 64 * AREC_LD - 64);   // (dx: pn <= ~600 doubles, far below the reduction cells)
}

// Jacobi scaling (computed once, at iteration 0: Ceres jacobi_scaling), gradient max-norm of
// x - Plus(x, -g) (Ceres gradient_max_norm) and |x|^2 of the reduced program.
// |x - Plus(x, -g)| of unknown j (Ceres gradient_max_norm: ambient difference for a rotation block, the box of the line delay)
__device__ __forceinline__ double grad_norm_entry(const Dev &d, const WinMeta &m, int w, int j, const double *g, bool at_cand) {
  const double *squat = at_cand ? d.cquat : d.quat, *sld = at_cand ? d.cld : d.ld;
  const int K6 = 6 * m.K;
  if (j < K6) {
    const int k = j / 6, c = j % 6;
    if (c == 0) {  // rotation block: ambient difference q - q*exp(-g)
      const double *q = squat + 4 * (m.knot0 + k);
      const Q4 q0 = qmk(q[0], q[1], q[2], q[3]);
      const Q4 q1 = qmul(q0, so3_exp(mk(-g[j], -g[j + 1], -g[j + 2])));
      return fmax(fmax(fabs(q0.x - q1.x), fabs(q0.y - q1.y)), fmax(fabs(q0.z - q1.z), fabs(q0.w - q1.w)));
    }
    return c >= 3 ? fabs(g[j]) : 0.0;
  }
  if (j == m.P - 1) {
    const double ld = sld[w];
    double nl = ld - g[j];
    if (!m.fix_ld) nl = fmin(fmax(nl, m.ld_lo), m.ld_hi);
    return fabs(ld - nl);
  }
  return fabs(g[j]);
}

// After the first linearisation of a solve (LIN_AT_X): Jacobi scaling (computed once, at iteration 0: Ceres jacobi_scaling) and the
// gradient max-norm of the initial state.  (Every later pass: k_pass_end.)
__global__ void k_post_linearize(Dev d, int mode) {
  const int w = blockIdx.y;
  Lm &lm = d.lm[w];
  if (!lin_run(lm, mode) || lin_cost_only(lm, mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m.N) return;
  const bool at_cand = mode == LIN_SPEC;
  const int tg = lin_target(lm, mode);
  const bool act = d.active[m.u0 + j] != 0;
  if (!at_cand && !lm.scaled) {
    const double h = (j < m.P) ? d.HppS[tg][m.H0 + (long long)j * m.ldh + j] : d.HllS[tg][m.lm0 + j - m.P];
    d.cscale[m.u0 + j] = act ? 1.0 / (1.0 + sqrt(fmax(h, 0.0))) : 1.0;
  }
  if (!act) return;
  const double gm = grad_norm_entry(d, m, w, j, d.gS[tg] + m.u0, at_cand);
  if (gm > 0.0) atomicMax(at_cand ? &lm.cand_gmax_bits : &lm.gmax_bits, (unsigned long long)__double_as_longlong(gm));
}

}  // namespace ctv
