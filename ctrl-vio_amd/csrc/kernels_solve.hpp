// kernels_solve.hpp -- The step: begin_iteration / k_begin_iter, Schur complement (k_schur_window_f64, k_schur_tile_f64, k_schur_tile2_f64), Cholesky
// (k_cholesky_tiles register-resident, k_cholesky_solve panel kernel), k_step_finish (back-substitution, candidate, its pair table).
// Part of kernels.hpp (included from there, in order; not a stand-alone header).
#pragma once

namespace ctv {

// ------------------------------------------------------------------------------------------------ Schur + solve
// Start of an iteration, by the threads of one workgroup.  Thread 0: FinalizeIterationAndCheckIfMinimizerCanContinue (windows inside
// the line search only report that they are still running).  Then, for the windows that start an iteration: the LM diagonal
// D^2 = clamp(diag(J^T J), min, max) / mu on the Jacobi-scaled system (Ceres LevenbergMarquardtStrategy::ComputeStep), expressed for
// the unscaled system: dd_j = clamp(c_j^2 H_jj) / (mu c_j^2), and 1 / (Hll + dd) of the landmarks.
__device__ __forceinline__ void begin_iteration(const Dev &d, int w, int *s_go) {
  Lm &lm = d.lm[w];
  if (threadIdx.x == 0) {
    int go = 0;
    if (!lm.status) {
      if (lm.ls_active) atomicAdd(d.n_active, 1);   // inside the line search: no new LM iteration
      else if (lm.iter >= d.prm.max_iters) lm.status = 1 + 0;
      else if (lm.last_ok && __longlong_as_double((long long)lm.gmax_bits) <= d.prm.gtol) lm.status = 1 + 1;
      else if (lm.mu <= d.prm.min_radius) lm.status = 1 + 4;
      else {
        lm.iter += 1;
        lm.accept = 0; lm.step_valid = 0; lm.chol_fail = 0; lm.alpha = 1.0;
        atomicAdd(d.n_active, 1);
        go = 1;
      }
    }
    *s_go = go;
  }
  __syncthreads();
  if (!*s_go) return;
  const WinMeta &m = d.wins[w];
  const double mu = lm.mu;
  const double *Hd = d.HppS[lm.cur] + m.H0, *Hl = d.HllS[lm.cur] + m.lm0;
  for (int j = threadIdx.x; j < m.N; j += blockDim.x) {
    const bool act = d.active[m.u0 + j] != 0;
    const double c = d.cscale[m.u0 + j];
    const double h = (j < m.P) ? Hd[(long long)j * m.ldh + j] : Hl[j - m.P];
    const double sc = fmin(fmax(c * c * h, d.prm.min_diag), d.prm.max_diag);
    const double dd = act ? sc / (mu * c * c) : 0.0;
    d.dd[m.u0 + j] = dd;
    if (j >= m.P) {   // by ROW of W (sorted landmark order, Dev::lm_pos): the Schur kernels and the back-substitution stream rows
      const int row = m.lm0 + d.lm_pos[m.lm0 + j - m.P];
      d.dinv[row] = (act && (h + dd) > 0.0) ? 1.0 / (h + dd) : 0.0;
      d.grs[row] = d.gS[lm.cur][m.u0 + j];
    }
  }
}
// The first iteration of a solve (every later one starts at the end of the previous pass: k_pass_end).
__global__ __launch_bounds__(256) void k_begin_iter(Dev d) {
  __shared__ int s_go;
  begin_iteration(d, blockIdx.x, &s_go);
}

__device__ __forceinline__ void tile_decode(int t, int &bi, int &bj) {  // t -> (bi >= bj), row-major over the lower triangle
  bi = 0;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  bj = t - bi * (bi + 1) / 2;
}

// Rows of W (sorted landmark order) that can be non-zero in BOTH the columns of tile row bi and those of tile column bj: the intersection of the
// host's per-tile ranges (host_pack.hpp: plan_sparsity); the tile row that holds index P carries g_rho on its A side for EVERY observed row.
// Tiles over bias-only columns come out empty.
__device__ __forceinline__ void schur_row_range(const Dev &d, const WinMeta &m, int bi, int bj, int &lbeg, int &lend) {
  int rb = d.tl_beg[m.tr0 + bi], re = d.tl_end[m.tr0 + bi];
  if (16 * bi <= m.P && m.P < 16 * bi + 16) { rb = 0; re = m.Lobs; }
  lbeg = max(rb, d.tl_beg[m.tr0 + bj]);
  lend = min(re, d.tl_end[m.tr0 + bj]);
}

// fp64 product path, large batches: the window kernel on the fp64 matrix cores.  One workgroup (8 waves) per window; W is read
// from HBM once, staged through LDS in double-buffered chunks of 16 landmarks (masked by the active flags, g_rho appended as
// column P so that the tile row holding index P also produces the reduced right-hand side: no k_rhs pass).  Output tiles are
// 16 x 16 (v_mfma_f64_16x16x4_f64, K = 4 landmarks per instruction); tile t of the lower triangle belongs to wave t % 8, which
// keeps its <= NTQ accumulators in registers over the whole landmark loop; tiles over bias-only columns have no products.
// NPRE = compact chunk elements per thread (16 (6K + 2) / 512 rounded up); NTQ = ceil(tiles with products / 8).
template <int NPRE, int NTQ> __global__ __launch_bounds__(512, NTQ <= 7 ? 4 : 2) void k_schur_window_f64(Dev d) {
  const int w = blockIdx.x;
  if (d.lm[w].status || d.lm[w].ls_active) return;
  const WinMeta &m = d.wins[w];
  const int P = m.P, L = m.L, ldw = m.ldw, u0 = m.u0, K6 = 6 * m.K, ldh = m.ldh;
  const int nt = ldw >> 4, ntile = nt * (nt + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) double smd64[];
  // LDS row stride of a staged chunk: ldw + 16 doubles.  ds_read_b64 serves lanes 0-31 in one cycle if they hit 64 distinct 4-byte banks: the 16
  // lanes of a k-row read 128 contiguous bytes, and the next k-row (lanes 16-31) must start 128 bytes (mod 256) away -- with the row stride ldw
  // (a multiple of 32 doubles = 256 bytes) both halves fell on the same 32 banks and every operand read took twice its cycles.
  const int ldl = ldw + 16;
  double *Wb = smd64;                    // [2][16][ldl]
  double *acts = Wb + 2 * 16 * ldl;      // [ldw] 1 / 0 (0 beyond P)      } per-column vectors of the epilogue, staged once: no global
  double *ddv = acts + ldw;              // [ldw] D of the column            } round trip per tile there (the activity of a STAGED column
  double *gv = ddv + ldw;                // [ldw] gradient                   } travels in pre_lc)
  double *dch = gv + ldw;                // [2][16] 1 / (Hll + D) of the chunk's landmarks (0 beyond L)
  int *tlist = reinterpret_cast<int *>(dch + 32);   // [8 NTQ] tiles with products (bi << 8 | bj), any order
  int &tcount = tlist[8 * NTQ];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q4 = lane >> 4, l15 = lane & 15;
  const double *Wp = d.WS[d.lm[w].cur] + m.W0;
  const double *dinv = d.dinv + m.lm0, *gl = d.grs + m.lm0;   // (by row of W: sorted landmark order)
  for (int c = tid; c < ldw; c += 512) {
    const int cc = min(c, P - 1);
    acts[c] = (c < P && d.active[u0 + cc]) ? 1.0 : 0.0; ddv[c] = d.dd[u0 + cc]; gv[c] = d.gS[d.lm[w].cur][u0 + cc];
  }
  if (tid == 0) tcount = 0;
  // W is non-zero only in the knot columns [0, 6K) and the line-delay column P - 1 (plus the rhs row P): a tile has products
  // when its row tile and its column tile both hold such a column.  Those tiles (55 of 105 at K = 24) are listed and dealt to
  // the waves; the others only need the epilogue (S = Hpp + D).
  auto nz_row = [&](int b) { return (16 * b < K6) || (P >= 16 * b && P - 1 < 16 * b + 16); };
  auto nz_col = [&](int b) { return (16 * b < K6) || (P - 1 >= 16 * b && P - 1 < 16 * b + 16); };
  __syncthreads();   // tcount
  for (int t = tid; t < ntile; t += 512) {
    int ti, tj;
    tile_decode(t, ti, tj);
    if (nz_row(ti) && nz_col(tj)) { const int pos = atomicAdd(&tcount, 1); if (pos < 8 * NTQ) tlist[pos] = (ti << 8) | tj; }
  }
  // Only the knot columns [0, 6K), the line-delay column P - 1 and the appended g_rho column P are fetched and staged (NC
  // compact columns per landmark); every other column of the two LDS buffers is zeroed once and stays zero.
  // SPARSITY: the rows of W are sorted by knot span (host_pack.hpp: plan_sparsity); rows past Lobs are zero, and a tile multiplies only the
  // chunks that overlap the row range of its two column tiles (cbeg / cend below).
  const int nchunk = (m.Lobs + 15) >> 4, nel = 16 * ldl, NC = K6 + 2, nelc = 16 * NC;
  for (int e = tid; e < 2 * nel; e += 512) Wb[e] = 0.0;
  double pre[NPRE];
  double pre_d = 0.0;
  int pre_lc[NPRE];     // chunk row << 16 | window column of this thread's elements (the same for every chunk)
  double keep[NPRE];    // (read together, after the barrier below)
#pragma unroll
  for (int k = 0; k < NPRE; ++k) {   // bit 30: the element is stored as it is (active column, or g_rho); otherwise as zero.  (As a factor read
    // from LDS at every stash -- acts[c] -- the compiler gave each element its own branch, ds_read and s_waitcnt: five serial round trips per chunk.)
    const int e = min(tid + 512 * k, nelc - 1), cc = e % NC, c = cc < K6 ? cc : P - 1 + (cc - K6);
    pre_lc[k] = ((e / NC) << 16) | c | ((tid + 512 * k < nelc && c == P) ? 1 << 30 : 0);
  }
  auto fetch = [&](int ch) {     // unconditional loads on clamped rows; masked when stored
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int l = min(16 * ch + ((pre_lc[k] >> 16) & 0xff), L - 1), c = pre_lc[k] & 0xffff;
      pre[k] = (c == P) ? gl[l] : Wp[(long long)l * ldw + c];
    }
    if (tid < 16) pre_d = dinv[min(16 * ch + tid, L - 1)];
  };
  auto stash = [&](int ch, int buf) {
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int lr = (pre_lc[k] >> 16) & 0xff, c = pre_lc[k] & 0xffff;
      const bool lv = 16 * ch + lr < L && (pre_lc[k] >> 30) != 0;
      if (tid + 512 * k < nelc) Wb[buf * nel + lr * ldl + c] = lv ? pre[k] : 0.0;
    }
    if (tid < 16) dch[16 * buf + tid] = (16 * ch + tid < L) ? pre_d : 0.0;
  };
  __syncthreads();   // zeroed buffers, tile list, column vectors
#pragma unroll
  for (int k = 0; k < NPRE; ++k) keep[k] = acts[min(pre_lc[k] & 0xffff, ldw - 1)];
#pragma unroll
  for (int k = 0; k < NPRE; ++k) pre_lc[k] |= (tid + 512 * k < nelc && keep[k] != 0.0) ? 1 << 30 : 0;
  const int nact = min(tcount, 8 * NTQ);
  long long *dbg = (d.dbg && w == (d.nwin > 1000 ? 1000 : 0)) ? d.dbg + 96 : nullptr;   // CTVIO_DEBUG_STAMPS: clock64 of thread 0 at the phase boundaries
  int dbi = 0;
#define CTV_STAMP() do { if (dbg && tid == 0 && dbi < 30) dbg[dbi++] = clock64(); } while (0)
  CTV_STAMP();
  // this wave's tiles: slot q holds list entry wave + 8 q; slots past the end repeat the wave's first tile (products computed,
  // result dropped) so that the tile loop below has no branches and the operand reads of a tile overlap the previous products
  int tij[NTQ];
#pragma unroll
  for (int q = 0; q < NTQ; ++q) tij[q] = __builtin_amdgcn_readfirstlane(tlist[(wave + 8 * q < nact) ? wave + 8 * q : min(wave, max(nact - 1, 0))]);   // SGPRs
  // the chunks a tile has products with ride in bits 16-23 (first) and 24-31 (end; 255 = no upper limit) of its SGPR -- separate registers
  // sent the kernel's scalar file over the edge
#pragma unroll
  for (int q = 0; q < NTQ; ++q) {
    int lb, le;
    schur_row_range(d, m, tij[q] >> 8, tij[q] & 255, lb, le);
    const bool real = wave + 8 * q < nact && le > lb;   // (a slot past the end of the list repeats the wave's first tile: no products for it)
    const int cb = real ? min(lb >> 4, 254) : 0, ce = real ? min((le + 15) >> 4, 255) : 0;
    tij[q] = __builtin_amdgcn_readfirstlane(tij[q] | (cb << 16) | (ce << 24));
  }
  // The accumulators start at -Hpp (rows beyond the unknowns -- the rhs row -- at 0): the tile's Hpp entries arrive with the first chunk of W
  // instead of costing the epilogue a global round trip per tile, and S = -(acc) + D needs no second operand there.
  const double *H = d.HppS[d.lm[w].cur] + m.H0;
  f64x4 acc[NTQ];
  int opq;   // (a zero the compiler cannot see through: otherwise the row / column indices computed here are kept -- spilled -- for the epilogue)
  asm volatile("s_mov_b32 %0, 0" : "=s"(opq));
#pragma unroll
  for (int q = 0; q < NTQ; ++q) {
    const int tq = (tij[q] & 0xffff) + opq, jc = min(16 * (tq & 255) + l15, P - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ii = 16 * (tq >> 8) + q4 + 4 * r, ic = min(ii, P - 1);
      const double h = H[(unsigned)(ic * ldh + min(jc, ic))];   // (32-bit offset from a uniform base: one address register per load)
      acc[q][r] = ii < P ? -h : 0.0;
    }
  }
  if (nchunk > 0) { fetch(0); stash(0, 0); }
  __syncthreads();
  CTV_STAMP();
  for (int ch = 0; ch < nchunk && nact > 0; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunk) fetch(ch + 1);
    const double *B = Wb + buf * nel + q4 * ldl + l15;
    double dl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) dl[s] = dch[16 * buf + 4 * s + q4];
    // The eight operand reads of a tile are issued together, and the scheduler may not move anything across the fences: left to itself it
    // issued every ds_read right before the v_mfma that consumes it -- 28 serial LDS round trips per chunk and wave (5.5 k - 13 k clocks
    // against 0.9 k of matrix-core time).  One round trip per tile is hidden by the other three waves of the SIMD.
#pragma unroll
    for (int q = 0; q < NTQ; ++q) {
      { const int cb = (tij[q] >> 16) & 255, ce = (tij[q] >> 24) & 255; if (ch < cb || (ch >= ce && ce != 255)) continue; }   // (uniform) no row of this chunk reaches both column tiles
      double a[4], b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[s] = B[4 * s * ldl + 16 * ((tij[q] >> 8) & 255)];
        b[s] = B[4 * s * ldl + 16 * (tij[q] & 255)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s] * dl[s], acc[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch + 1 < nchunk) stash(ch + 1, buf ^ 1);
    __syncthreads();
  }
  CTV_STAMP();
  // epilogue: S = Hpp - W^T Hll^-1 W + D on the active lower triangle, identity rows for fixed unknowns; rhs row.  The per-column vectors
  // come from LDS, read before any branch (pin) so that the compiler does not sink each read into a branch of its own.
  double *S = d.S + m.H0, *rhs = d.rhs + m.p0;
  auto pin = [](double &x) { asm volatile("" : "+v"(x)); };
  // hs[r] = the tile's entry without the damping: -acc of a product tile, Hpp of a plain one; bs[r] = its rhs-row entry before -g
  auto store_tile = [&](int ti, int tj, const double (&hs)[4], const double (&bs)[4]) {
    const int jj = 16 * tj + l15, jc = min(jj, P - 1);
    double a_j = acts[jc], dd_j = ddv[jc], g_j = gv[jc], a_i[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a_i[r] = acts[min(16 * ti + q4 + 4 * r, P - 1)];
    pin(a_j); pin(dd_j); pin(g_j);
#pragma unroll
    for (int r = 0; r < 4; ++r) pin(a_i[r]);
    const bool act_j = a_j != 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ii = 16 * ti + q4 + 4 * r;
      if (ii < P && jj <= ii) {
        const bool on = a_i[r] != 0.0 && act_j;
        S[(long long)ii * ldh + jj] = on ? hs[r] + (ii == jj ? dd_j : 0.0) : (ii == jj ? 1.0 : 0.0);
      } else if (ii == P && jj < P) {
        rhs[jj] = act_j ? bs[r] - g_j : 0.0;
      }
    }
  };
#pragma unroll
  for (int q = 0; q < NTQ; ++q) {
    if (wave + 8 * q >= nact) continue;
    double hs[4], bs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { hs[r] = -acc[q][r]; bs[r] = acc[q][r]; }
    store_tile((tij[q] >> 8) & 255, tij[q] & 255, hs, bs);
  }
  CTV_STAMP();
  // tiles without products (S = Hpp + D): this wave's list first (scalar), then the Hpp entries of four tiles requested together -- every tile used
  // to be a global round trip of its own
  auto next_plain = [&](int t) {   // the next tile without products of this wave at or after t (wave-uniform)
    for (; t < ntile; t += 8) {
      int ti, tj;
      tile_decode(t, ti, tj);
      if (!(nz_row(ti) && nz_col(tj))) break;
    }
    return t;
  };
  if (d.schur_plain_in_H) {   // the Cholesky kernel reads Hpp itself (half of this kernel's Hpp reads and S writes were copies); only the
    for (int c = tid; c < P; c += 512)   // rhs entries of the columns that no product tile covers are left to do
      if (!nz_col(c >> 4)) rhs[c] = acts[c] != 0.0 ? -gv[c] : 0.0;
  } else
  for (int tp = next_plain(wave); tp < ntile;) {
    int pi[4], pj[4];   // (-1: none)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      pi[u] = -1; pj[u] = 0;
      if (tp < ntile) { tile_decode(tp, pi[u], pj[u]); tp = next_plain(tp + 8); }
    }
    double hv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // unconditional loads (a missing tile repeats tile (0, 0))
      const int ti = max(pi[u], 0), jc = min(16 * pj[u] + l15, P - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ic = min(16 * ti + q4 + 4 * r, P - 1);
        hv[u][r] = H[(unsigned)(ic * ldh + min(jc, ic))];
      }
    }
    const double zero[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (pi[u] >= 0) store_tile(pi[u], pj[u], hv[u], zero);
  }
  CTV_STAMP();
#undef CTV_STAMP
}

// fp64 path: the same SYRK on the fp64 matrix cores, one wave per 16 x 16 tile of the lower triangle
// (v_mfma_f64_16x16x4_f64: A operand lane l = X[k = l/16][i = l%16], B operand lane l = Y[k = l/16][j = l%16],
// D register r of lane l = D[(l/16) + 4r][l%16]; measured with tools/mfma_f64_layout.hip).  Operands straight from W,
// 16 rows (4 products) per trip with all loads of a trip in flight; the reduced rhs rides along as row P.  SPARSITY: a tile outside the window's
// envelope (Dev::env_first) is not formed at all; inside it the tile multiplies only the rows of W whose knot span meets both its column ranges.
__global__ __launch_bounds__(64) void k_schur_tile_f64(Dev d, int ntile_max) {
  // XCD-aware tile -> workgroup map: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the tiles of one
  // window get ids that are congruent mod 8: they all run on one XCD and the window's W (re-read by every tile) comes out of
  // that L2 instead of being fetched 8 times over the fabric.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int w = (slot / ntile_max) * 8 + xcd, tile = slot % ntile_max;
  if (w >= d.nwin) return;
  if (d.lm[w].status || d.lm[w].ls_active) return;
  const WinMeta &m = d.wins[w];
  const int P = m.P, L = m.L, ldw = m.ldw, u0 = m.u0, ldh = m.ldh;
  const int nt = P / 16 + 1;   // tile rows up to index P: the rhs rides along as row P (g_rho on the A side), so the tile row that
  if (tile >= nt * (nt + 1) / 2) return;   // holds it also produces W^T diag(dinv) g_rho -- no separate k_rhs pass
  int bi, bj;
  tile_decode(tile, bi, bj);
  if (bj < d.env_first[m.tr0 + bi]) return;   // structurally zero: never read by the factorisation
  const int lane = threadIdx.x, q4 = lane >> 4, l15 = lane & 15;
  const int i = min(16 * bi + l15, ldw - 1), j = min(16 * bj + l15, ldw - 1);
  const bool rhs_lane = 16 * bi + l15 == P;
  const double ai = (16 * bi + l15 < P && d.active[u0 + min(i, P - 1)]) ? 1.0 : 0.0;
  const double aj = (16 * bj + l15 < P && d.active[u0 + min(j, P - 1)]) ? 1.0 : 0.0;
  const double *Wp = d.WS[d.lm[w].cur] + m.W0;
  const double *dinv = d.dinv + m.lm0, *gl = d.grs + m.lm0;
  int lbeg, lend;
  schur_row_range(d, m, bi, bj, lbeg, lend);   // (empty: the loop -- and its clamped row -- is skipped)
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int l0 = lbeg; l0 < lend; l0 += 16) {
    double wa[4], wb[4], dv[4], gv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {   // unconditional loads on clamped rows, masked below
      const int lc = min(l0 + 4 * s + q4, L - 1);
      wa[s] = Wp[(long long)lc * ldw + i];
      wb[s] = Wp[(long long)lc * ldw + j];
      dv[s] = dinv[lc];
      gv[s] = gl[lc];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) wa[s] = rhs_lane ? gv[s] : wa[s] * ai;   // (unconditional loads, selected afterwards)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool lv = l0 + 4 * s + q4 < lend;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[s], lv ? wb[s] * aj * dv[s] : 0.0, acc, 0, 0, 0);
    }
  }
  double *S = d.S + m.H0, *rhs = d.rhs + m.p0;
  const double *H = d.HppS[d.lm[w].cur] + m.H0;
  const int jj = 16 * bj + l15, jc = min(jj, P - 1);
  const bool act_j = d.active[u0 + jc] != 0;
  const double dd_j = d.dd[u0 + jc], g_j = d.gS[d.lm[w].cur][u0 + jc];
  double hv[4];
  unsigned char act_i[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ic = min(16 * bi + q4 + 4 * r, P - 1);
    act_i[r] = d.active[u0 + ic];
    hv[r] = H[(long long)ic * ldh + min(jc, ic)];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ii = 16 * bi + q4 + 4 * r;
    if (ii < P && jj <= ii) {
      const bool on = act_i[r] && act_j;
      S[(long long)ii * ldh + jj] = on ? hv[r] - acc[r] + (ii == jj ? dd_j : 0.0) : (ii == jj ? 1.0 : 0.0);
    } else if (ii == P && jj < P) {
      rhs[jj] = act_j ? acc[r] - g_j : 0.0;   // reduced right-hand side: -g_p + W^T diag(dinv) g_rho
    }
  }
}

// The same with 2 x 2 REGISTER BLOCKING: a wave owns the four tiles (2 Bi + a, 2 Bj + b) of a 32 x 32 block of the lower triangle and
// feeds four MFMAs from four operand loads per K-step (the one-tile form: two loads per MFMA).  With operands straight from W the
// one-tile kernel is bound by L2 bandwidth once there are enough waves to fill the chip (config 5, P = 571: 666 tiles x 128 windows, every
// tile wave re-reading 2 x 16 columns of its window's 4.9 MB W: 0.19 of the fp64 matrix peak); half the loads per product.  For batches
// whose tile count fills the chip; a single small window keeps the one-tile form (more waves in flight, shorter latency).
// SPARSITY: a block whose four tiles all lie outside the envelope exits at once; the others multiply the union of their tiles' row ranges (the
// rows of W are sorted by knot span, so the union is a short interval: config 5, K = 64 -- ~60 of 1000 rows per block instead of all).
__global__ __launch_bounds__(64) void k_schur_tile2_f64(Dev d, int nblk_max) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;     // (XCD-aware map as above: a window's blocks share one L2)
  const int w = (slot / nblk_max) * 8 + xcd, blk = slot % nblk_max;
  if (w >= d.nwin) return;
  if (d.lm[w].status || d.lm[w].ls_active) return;
  const WinMeta &m = d.wins[w];
  const int P = m.P, L = m.L, ldw = m.ldw, u0 = m.u0, ldh = m.ldh;
  const int nt = P / 16 + 1, nb = (nt + 1) / 2;   // tile rows up to index P (the rhs row rides along as row P); 32-row blocks
  if (blk >= nb * (nb + 1) / 2) return;
  int Bi, Bj;
  tile_decode(blk, Bi, Bj);
  const int lane = threadIdx.x, q4 = lane >> 4, l15 = lane & 15;
  const int cur = d.lm[w].cur;
  const double *Wp = d.WS[cur] + m.W0;
  const double *dinv = d.dinv + m.lm0, *gl = d.grs + m.lm0;
  int ci[2], cj[2];
  bool rhs_lane[2];
  double ai[2], aj[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int bi = 2 * Bi + a, bj = 2 * Bj + a;
    ci[a] = min(16 * bi + l15, ldw - 1); cj[a] = min(16 * bj + l15, ldw - 1);
    rhs_lane[a] = 16 * bi + l15 == P;
    ai[a] = (16 * bi + l15 < P && d.active[u0 + min(ci[a], P - 1)]) ? 1.0 : 0.0;
    aj[a] = (16 * bj + l15 < P && d.active[u0 + min(cj[a], P - 1)]) ? 1.0 : 0.0;
  }
  // which of the four tiles exist (inside the triangle, inside the envelope), and the union of their row ranges (all uniform)
  bool live[2][2];
  int lbeg = L, lend = 0;
  bool any = false;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int bi = 2 * Bi + a, bj = 2 * Bj + b;
      live[a][b] = bi < nt && bj <= bi && bj >= d.env_first[m.tr0 + min(bi, nt - 1)];
      if (live[a][b]) {
        any = true;
        int rb, re;
        schur_row_range(d, m, bi, bj, rb, re);
        if (re > rb) { lbeg = min(lbeg, rb); lend = max(lend, re); }
      }
    }
  if (!any) return;
  if (lend <= lbeg) { lbeg = 0; lend = 0; }
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  // Two chunks of 16 rows in flight: the 24 operand loads of the next chunk are requested before the 16 products of the current one
  // (a trip used to be "load, wait, multiply": the matrix cores idle for a memory round trip per chunk, 0.34 of the fp64 peak at three
  // waves per SIMD).  Fences keep the scheduler from moving the requests back behind the products.
  struct Chunk { double wa[2][4], wb[2][4], dv[4], gv[4]; };
  auto fetch = [&](int l0, Chunk &c) {   // unconditional loads on clamped rows, masked in `products`
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int lc = max(0, min(l0 + 4 * s + q4, L - 1));
      const double *row = Wp + (long long)lc * ldw;
      c.wa[0][s] = row[ci[0]]; c.wa[1][s] = row[ci[1]];
      c.wb[0][s] = row[cj[0]]; c.wb[1][s] = row[cj[1]];
      c.dv[s] = dinv[lc];
      c.gv[s] = gl[lc];
    }
  };
  const bool upper = Bi != Bj;   // (uniform; on a diagonal block tile (0, 1) is above the diagonal)
  auto products = [&](int l0, const Chunk &c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool lv = l0 + 4 * s + q4 < lend;
      const double dvs = lv ? c.dv[s] : 0.0;
      const double a0 = rhs_lane[0] ? c.gv[s] : c.wa[0][s] * ai[0], a1 = rhs_lane[1] ? c.gv[s] : c.wa[1][s] * ai[1];
      const double b0 = c.wb[0][s] * aj[0] * dvs, b1 = c.wb[1][s] * aj[1] * dvs;
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
      if (upper) acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    }
  };
  // (The requests are unconditional -- past the end they re-read the last rows: a request behind a uniform branch makes the compiler wait,
  // at the join, as if the OLDER chunk were the newest one: vmcnt(23) instead of vmcnt(47).)
  if (L > 0 && lend > lbeg) {
    Chunk ca, cb;
    fetch(lbeg, ca);
    for (int l0 = lbeg; l0 < lend; l0 += 32) {
      fetch(l0 + 16, cb);
      __builtin_amdgcn_sched_barrier(0);
      products(l0, ca);
      __builtin_amdgcn_sched_barrier(0);
      fetch(l0 + 32, ca);
      __builtin_amdgcn_sched_barrier(0);
      if (l0 + 16 < lend) products(l0 + 16, cb);   // (uniform)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double *S = d.S + m.H0, *rhs = d.rhs + m.p0;
  const double *H = d.HppS[cur] + m.H0, *gp = d.gS[cur] + u0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int bi = 2 * Bi + a, bj = 2 * Bj + b;
      if (!live[a][b]) continue;   // (uniform)
      const int jj = 16 * bj + l15, jc = min(jj, P - 1);
      const bool act_j = d.active[u0 + jc] != 0;
      const double dd_j = d.dd[u0 + jc], g_j = gp[jc];
      double hv[4];          // requested together, before any branch (inside the branch each was a round trip of its own)
      unsigned char av[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ic = min(16 * bi + q4 + 4 * r, P - 1);
        hv[r] = H[(long long)ic * ldh + min(jc, ic)];
        av[r] = d.active[u0 + ic];
      }
      asm volatile("" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = 16 * bi + q4 + 4 * r;
        if (ii < P && jj <= ii) {
          const bool on = av[r] && act_j;
          S[(long long)ii * ldh + jj] = on ? hv[r] - acc[a][b][r] + (ii == jj ? dd_j : 0.0) : (ii == jj ? 1.0 : 0.0);
        } else if (ii == P && jj < P) {
          rhs[jj] = act_j ? acc[a][b][r] - g_j : 0.0;   // reduced right-hand side: -g_p + W^T diag(dinv) g_rho
        }
      }
    }
}

// Dense fp64 Cholesky of the P x P reduced system + solve, one workgroup (4 waves) per window, right-looking with
// 32-column panels, the matrix products on the fp64 matrix cores (v_mfma_f64_16x16x4_f64):
//   1. wave 0 factors the 32 x 32 diagonal block, one row per lane in registers, with v_readlane broadcasts (no LDS,
//      no barriers inside the 32 pivot steps) and forms L11^-1 in the same sweep (lane = column of the inverse).
//      Meanwhile waves 1-3 stage the panel rows A21 (and the rhs row) into LDS, k-major.
//   2. L21 = A21 L11^-T as an MFMA product, in place in the LDS panel (a 16-row tile is owned by one wave);
//   3. trailing update A22 -= L21 L21^T: one 16 x 16 tile per wave at a time, 8 MFMAs, read-modify-write of S.
// The right-hand side rides along as an extra matrix row (Cholesky of [S b; b^T .]), so y = L^-1 b needs no
// separate forward substitution; only the block back-substitution L^T x = y remains.  Result in delta[0..P).
// MFMA register layout (measured, tools/mfma_f64_layout.hip): A operand lane l = A[l%16][l/16], B operand lane l =
// B[l/16][l%16], D register r of lane l = D[(l/16) + 4r][l%16].

// Diagonal block of k_cholesky_solve: factorisation fused with the inversion, on ONE register array.  Lanes 0-31 hold the rows
// of the block (v[c] = A[lane][c]), lanes 32-63 the columns of X = L11^-1 in the making (v[c] = X[c][lane - 32], identity at the
// start).  The rank-1 update of pivot J, a_c -= a_J s with s = L[C][J] = v[J] of lane C, is also the substitution step
// x_c -= x_J s of the inverse: one v_readlane pair and ONE v_fma per (J, C) serve both halves of the wave.
// One update as an asm block so that the broadcast value lives for exactly these instructions (left to the compiler, every
// broadcast was spilled and reloaded).
template <int C> __device__ __forceinline__ void chol_bcast_update(double &vc, double vj, int vj_lo, int vj_hi) {
  asm("v_readlane_b32 s96, %2, %4\n\tv_readlane_b32 s97, %3, %4\n\ts_nop 1\n\t"
      "v_fma_f64 %0, -%1, s[96:97], %0"
      : "+v"(vc)
      : "v"(vj), "v"(vj_lo), "v"(vj_hi), "n"(C)
      : "s96", "s97");
}
// The same for the first column after the pivot, whose result feeds the next pivot's v_readlane straight away: gfx950 needs a
// wait state between a VALU write of a VGPR and a v_readlane of it (and between the compiler's scaling of v[J] and the first
// v_readlane here); the hazard recogniser cannot see into an asm block, so the s_nops are spelled out.
template <int C> __device__ __forceinline__ void chol_bcast_update_first(double &vc, double vj, int vj_lo, int vj_hi) {
  asm("s_nop 1\n\tv_readlane_b32 s96, %2, %4\n\tv_readlane_b32 s97, %3, %4\n\ts_nop 1\n\t"
      "v_fma_f64 %0, -%1, s[96:97], %0\n\ts_nop 1"
      : "+v"(vc)
      : "v"(vj), "v"(vj_lo), "v"(vj_hi), "n"(C)
      : "s96", "s97");
}
// 1 / sqrt(p) of the pivot: hardware estimate + two Newton steps (short dependent chain instead of sqrt + divide)
__device__ __forceinline__ double chol_pivot_rsqrt(double pj, int &bad) {
  const bool ok = (pj > 0.0) && isfinite(pj);
  if (!ok) bad = 1;
  const double ps = ok ? pj : 1.0;
  double di = __builtin_amdgcn_rsq(ps);
  const double hp = 0.5 * ps;
  di = di * (1.5 - hp * di * di);
  di = di * (1.5 - hp * di * di);
  return di;
}
// Four columns at once, each broadcast in its own SGPR pair: with a single pair every update waited for the previous FMA to
// release it (~42 cycles per update, measured: 25 k cycles per 32 x 32 block); here the eight v_readlane run ahead of the
// four FMAs, which also puts the two wait states gfx950 wants between a VALU write of an SGPR and its VALU read in between.
template <int C> __device__ __forceinline__ void chol_bcast_update4(double &v0, double &v1, double &v2, double &v3, double vj, int vj_lo, int vj_hi) {
  asm("v_readlane_b32 s92, %5, %7\n\tv_readlane_b32 s93, %6, %7\n\t"
      "v_readlane_b32 s94, %5, %8\n\tv_readlane_b32 s95, %6, %8\n\t"
      "v_readlane_b32 s96, %5, %9\n\tv_readlane_b32 s97, %6, %9\n\t"
      "v_readlane_b32 s98, %5, %10\n\tv_readlane_b32 s99, %6, %10\n\t"
      "v_fma_f64 %0, -%4, s[92:93], %0\n\tv_fma_f64 %1, -%4, s[94:95], %1\n\t"
      "v_fma_f64 %2, -%4, s[96:97], %2\n\tv_fma_f64 %3, -%4, s[98:99], %3"
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)
      : "v"(vj), "v"(vj_lo), "v"(vj_hi), "n"(C), "n"(C + 1), "n"(C + 2), "n"(C + 3)
      : "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
}
// columns C .. 31 of pivot J
template <int J, int C> __device__ __forceinline__ void chol_row_updates(double (&v)[32], int lo, int hi) {
  if constexpr (C + 3 <= 31) {
    chol_bcast_update4<C>(v[C], v[C + 1], v[C + 2], v[C + 3], v[J], lo, hi);
    chol_row_updates<J, C + 4>(v, lo, hi);
  } else if constexpr (C <= 31) {
    chol_bcast_update<C>(v[C], v[J], lo, hi);
    chol_row_updates<J, C + 1>(v, lo, hi);
  }
}
// Pivot J with its 1 / sqrt already known (di): scale column J, update column J + 1 first, start the NEXT pivot's reciprocal
// square root from it (its dependent chain of ~10 fp64 operations then overlaps the remaining updates), update the rest.
template <int J> __device__ __forceinline__ double chol_diag_step(double (&v)[32], double di, int &bad) {
  v[J] *= di;   // lanes < 32: lane J sqrt(p_J), lanes > J L[i][J]; lanes >= 32: X[J][.], final (every k < J has been eliminated)
  const int lo = __double2loint(v[J]), hi = __double2hiint(v[J]);
  double di_next = 0.0;
  if constexpr (J < 31) {
    chol_bcast_update_first<J + 1>(v[J + 1], v[J], lo, hi);
    di_next = chol_pivot_rsqrt(readlane_d(v[J + 1], J + 1), bad);
    if constexpr (J < 30) chol_row_updates<J, J + 2>(v, lo, hi);
  }
  return di_next;
}
template <int J> __device__ __forceinline__ void chol_diag_from(double (&v)[32], double di, int &bad) {
  const double dn = chol_diag_step<J>(v, di, bad);
  if constexpr (J < 31) chol_diag_from<J + 1>(v, dn, bad);
}
__device__ __forceinline__ void chol_diag_all(double (&v)[32], int &bad) {
  chol_diag_from<0>(v, chol_pivot_rsqrt(readlane_d(v[0], 0), bad), bad);
}
// NW waves per window: 4 for large batches (two windows share a CU), 8 when there are fewer windows than CUs (the parallel
// phases -- L21, trailing update, staging -- go twice as fast; the diagonal blocks hide behind the trailing updates).
// SPARSITY (the reference factors with SPARSE_NORMAL_CHOLESKY, trajectory_estimator.cpp:371-384): the kernel works inside the window's ENVELOPE
// (Dev::env_first, host_pack.hpp: plan_sparsity -- per 16-row tile the first tile column that can be non-zero; fill stays inside the row envelope).
// A 16-row tile R below panel jb TAKES PART in it iff env_first[R] <= jb / 16 + 1; only those tiles are staged, solved against L11 and updated,
// and a trailing tile (Ri, Rj) is touched only when both rows take part.  Config 5 (K = 64, P = 571): 1.9 k of 7.8 k tile products.  The host
// aligns the envelope to the 32-column panels and makes the next diagonal block take part in every panel (its look-ahead below).
template <int NW> __global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_cholesky_solve(Dev d) {
  constexpr int NT = 64 * NW;
  const int w = blockIdx.x;
  Lm &lm = d.lm[w];
  if (lm.status || lm.ls_active) return;
  const WinMeta &m = d.wins[w];
  const int P = m.P, ldh = m.ldh, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  extern __shared__ __attribute__((aligned(16))) double smc[];
  double *Lb = smc;                 // [32][34] NEXT diagonal block (row-major), deposited by the trailing update of the current panel
  double *LiT = Lb + 32 * 34;       // [32][34] L11^-1 transposed: LiT[k][j] = Linv[j][k]
  double *dinvs = LiT + 32 * 34;    // [32] 1 / L_jj
  double *yb = dinvs + 32;          // [32]
  int &s_fail = *reinterpret_cast<int *>(yb + 32);
  int &s_trip = reinterpret_cast<int *>(yb + 32)[1];   // next unclaimed tile of the trailing update
  int *plist = reinterpret_cast<int *>(yb + 34);       // [64] the local 16-row tiles that take part in the current panel, ascending
  double *LpT = yb + 34 + 32;       // [32][RS] panel (+ rhs row) k-major: LpT[k][r]
  double *S = d.S + m.H0;
  double *y = d.rhs + m.p0;         // augmented row; becomes L^-1 rhs
  double *x = d.delta + m.u0;
  const int32_t *ef = d.env_first + m.tr0;   // [P / 16 + 1]
  if (tid == 0) s_fail = 0;
  for (int e = tid; e < 32 * 32; e += NT) {   // first diagonal block -> LDS (rows / columns clamped; masked when read)
    const int r = e >> 5, c = e & 31;
    Lb[r * 34 + c] = S[(long long)min(r, P - 1) * ldh + min(c, P - 1)];
  }
  __syncthreads();
  long long *dbg = (d.dbg && w == 0) ? d.dbg : nullptr;
  int dbi = 0;
#define CTV_STAMP() do { if (dbg && tid == 0 && dbi < 30) dbg[dbi++] = clock64(); } while (0)
  CTV_STAMP();
  const int q4 = lane >> 4, l15 = lane & 15;
  // ---- diagonal block at column jb (one wave): lanes 0-31 the rows (lanes >= nb of the last, partial block carry identity
  //      rows), lanes 32-63 the columns of the inverse (identity).  The block is in LDS (Lb): the first one staged above, the
  //      later ones left there by the trailing update.  Result: LiT (LDS) and chol_inv (HBM, for the back-substitution); L11
  //      itself is not written back, nothing reads it.
  auto diag_block = [&](int jb) {
    const int nb = min(32, P - jb);
    double v[32];
#pragma unroll
    for (int c = 0; c < 32; c += 2) {
      const VecN<double, 2> v2 = *reinterpret_cast<const VecN<double, 2> *>(Lb + (lane & 31) * 34 + c);
      v[c] = v2.v[0]; v[c + 1] = v2.v[1];
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const bool in = lane < nb && c < nb && c <= lane;
      v[c] = in ? v[c] : ((c == (lane & 31)) ? 1.0 : 0.0);
    }
    int bad = 0;
    chol_diag_all(v, bad);
    if (lane >= 32) {
      const int col = lane - 32;
      double *gi = d.chol_inv + ((size_t)w * d.chol_nblk + (jb >> 5)) * 1024;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        LiT[col * 34 + i] = v[i];   // LiT[k = col][j = i] = Linv[i][col]
        gi[i * 32 + col] = v[i];    // row-major Linv[i][col]
      }
    }
    if (lane == 0 && bad) s_fail = 1;
  };
  for (int jb = 0; jb < P; jb += 32) {
    const int nb = min(32, P - jb), r0 = jb + nb, nt = P - r0, ntr = nt + 1;  // ntr: trailing rows incl. the rhs row
    const int RS = (ntr + 15) & ~15, ntile = RS >> 4;
    // ---- the tiles that take part (every wave forms the same mask: one ballot; <= 36 local tiles).  The last wave lists them in LDS.
    const int R0 = r0 >> 4;
    const bool mine = lane < ntile && ef[min(R0 + lane, P / 16)] <= (jb >> 4) + 1;
    const unsigned long long pmask = __ballot(mine);
    const int np = __popcll(pmask);
    if (wave == NW - 1 && mine) plist[__popcll(pmask & ((1ull << lane) - 1ull))] = lane;
    // ---- panel rows (and the rhs row) into the LDS panel, LpT[k][r].  First panel: wave 0 factors the diagonal block
    //      meanwhile; the later diagonal blocks were factored during the previous trailing update (look-ahead, below).
    {
      const int first = jb == 0 ? 64 : 0, nthr = NT - first;
      if (jb == 0 && wave == 0) diag_block(0);
      for (int r = tid - first; r >= 0 && r < RS; r += nthr) {
        if (!((pmask >> (r >> 4)) & 1ull)) continue;   // a tile outside the panel's envelope: nothing of it is read below
        const double *src = (r < nt) ? S + (long long)(r0 + r) * ldh + jb : y + jb;
        double tmp[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) tmp[k] = src[min(k, nb - 1)];   // unconditional: 32 loads in flight
        const bool live = r < ntr;
#pragma unroll
        for (int k = 0; k < 32; ++k) LpT[k * RS + r] = (live && k < nb) ? tmp[k] : 0.0;
      }
    }
    if (tid == 0) s_trip = 4;   // wave 0 starts with tiles 0-3 (they hold the next diagonal block)
    // LDS-only barrier: what the next phase reads (LiT, LpT, plist) is in LDS; wave 0's global stores of the block inverse may
    // stay in flight (a full __syncthreads would wait for them; they are read after later full barriers only)
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    CTV_STAMP();
    // ---- L21 = A21 L11^-T, in place: Linv is lower triangular, so output columns 0..15 need k < 16 only
    for (int it = wave; it < np; it += NW) {
      const int tr = plist[it];
      f64x4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
      const double *pa = LpT + 16 * tr + l15;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int k = 4 * kk + q4;
        const double av = pa[k * RS];
        if (kk < 4) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, LiT[k * 34 + l15], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, LiT[k * 34 + 16 + l15], c1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tr + q4 + 4 * r;
        LpT[l15 * RS + row] = c0[r];
        LpT[(16 + l15) * RS + row] = c1[r];
        if (row < ntr) {
          double *dst = (row < nt) ? S + (long long)(r0 + row) * ldh + jb : y + jb;
          if (l15 < nb) dst[l15] = c0[r];
          if (16 + l15 < nb) dst[16 + l15] = c1[r];
        }
      }
    }
    __syncthreads();
    CTV_STAMP();
    // ---- trailing update A22 -= L21 L21^T on the lower triangle (16 x 16 tiles) and the rhs row, over the PAIRS of tiles that take part.
    //      Trips of 4 consecutive pairs are claimed from an LDS counter.  Wave 0 takes pairs 0-3 first -- (0,0), (1,0), (1,1) are the next
    //      diagonal block (local tiles 0 and 1 always take part), left in Lb -- then factors that block (LOOK-AHEAD: 21 k cycles on one wave
    //      that used to sit between the panels with three waves idle) while the other waves work through the rest, then joins them.
    const int ntt = nt > 0 ? np * (np + 1) / 2 : 0;   // last panel: nothing left to update
    // (requesting the next trip's S values before the current products was tried: no gain, it is bandwidth not latency)
    bool first_trip = wave == 0;
    while (true) {   // 4 tiles per trip: 16 loads in flight, 32 MFMAs, 16 stores
      int tb;
      if (first_trip) tb = 0;
      else tb = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(&s_trip, 4) : 0);
      if (tb >= ntt) break;
      double sv[4][4];
      int ti4[4], tj4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int pa_, pb_;
        tile_decode(min(tb + u, ntt - 1), pa_, pb_);
        ti4[u] = plist[pa_]; tj4[u] = plist[pb_];
        const int col = 16 * tj4[u] + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ti4[u] + q4 + 4 * r;
          const double *src = (row < nt) ? S + (long long)(r0 + row) * ldh + r0 : y + r0;
          sv[u][r] = src[min(col, nt - 1)];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f64x4 c = {0.0, 0.0, 0.0, 0.0};
        const double *pa = LpT + 16 * ti4[u] + l15, *pb = LpT + 16 * tj4[u] + l15;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int k = 4 * kk + q4;
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k * RS], pb[k * RS], c, 0, 0, 0);
        }
        const int col = 16 * tj4[u] + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ti4[u] + q4 + 4 * r;
          if (tb + u < ntt && col < nt && ((row < nt && col <= row) || row == nt)) {
            double *dst = (row < nt) ? S + (long long)(r0 + row) * ldh + r0 : y + r0;
            const double nv = sv[u][r] - c[r];
            dst[col] = nv;
            if (row < 32 && row < nt) Lb[row * 34 + col] = nv;   // tiles (0,0), (1,0), (1,1): the next diagonal block
          }
        }
      }
      if (first_trip) {
        first_trip = false;
        __builtin_amdgcn_s_waitcnt(0xc07f);   // this wave's Lb writes
        diag_block(r0);
      }
    }
    __syncthreads();
    CTV_STAMP();
  }
  // ---- block back-substitution L^T x = y with the stored block inverses: x_b = Linv_b^T t_b, then t_j -= L[b][j]^T x_b
  //      for the rows above.  x lives in LDS; per block the loads of Linv_b (wave 0) and of the panel rows (everyone) do
  //      not depend on x and are issued together, before the block solve.  Block row b reaches back to column 16 env_first only: the two
  //      16-row tiles of the block have their own starts (ca <= cb or cb <= ca), columns before a tile's start are not stored at all.
  double *xs = LpT;   // the panel is no longer needed
  for (int i = tid; i < P; i += NT) xs[i] = y[i];
  __syncthreads();
  const int nblk = (P + 31) / 32;
  for (int b = nblk - 1; b >= 0; --b) {
    const int jb = 32 * b, nb = min(32, P - jb);
    const int ca = 16 * ef[2 * b], cb = 16 * ef[min(2 * b + 1, P / 16)], c0 = min(ca, cb);
    double lv[32];   // column j of the panel rows of this block: L[jb + ii][j]
    const int j0 = c0 + tid;
    const bool upd = j0 < jb;
    {
      const int j = min(j0, max(jb - 1, 0)), ja = max(j, ca), jc = max(j, cb);   // (clamped into each tile's stored range; masked below)
#pragma unroll
      for (int ii = 0; ii < 32; ++ii) lv[ii] = S[(long long)(jb + min(ii, nb - 1)) * ldh + (ii < 16 ? ja : jc)];
    }
    if (wave == 0) {
      const double *gi = d.chol_inv + ((size_t)w * d.chol_nblk + b) * 1024;
      double li[32];
      const int l31 = lane & 31;
#pragma unroll
      for (int i = 0; i < 32; ++i) li[i] = gi[i * 32 + l31];
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < 32; ++i) acc += li[i] * ((i < nb) ? xs[jb + i] : 0.0);   // Linv is lower triangular: rows i >= lane
      __builtin_amdgcn_wave_barrier();
      if (lane < nb) { xs[jb + lane] = acc; yb[lane] = acc; }
    }
    __syncthreads();
    if (upd) {
      double sacc = 0.0;
      const bool ma = j0 >= ca, mb = j0 >= cb;
#pragma unroll
      for (int ii = 0; ii < 32; ++ii) sacc += ((ii < 16 ? ma : mb) && ii < nb) ? lv[ii] * yb[ii] : 0.0;
      xs[j0] -= sacc;
    }
    for (int j = j0 + NT; j < jb; j += NT) {   // remaining columns of a wide block row
      double sacc = 0.0;
      for (int ii = 0; ii < nb; ++ii) sacc += (j >= (ii < 16 ? ca : cb)) ? S[(long long)(jb + ii) * ldh + j] * yb[ii] : 0.0;
      xs[j] -= sacc;
    }
    __syncthreads();
  }
  for (int i = tid; i < P; i += NT) x[i] = xs[i];
  CTV_STAMP();
  if (tid == 0) lm.chol_fail = s_fail;
#undef CTV_STAMP
}

// ---- Register-resident tile Cholesky (windows with P <= 223): the whole lower triangle of the reduced system lives in the
// VGPRs of ONE workgroup as 16 x 16 tiles in the accumulator layout of v_mfma_f64_16x16x4_f64 (tile t = i (i + 1) / 2 + j, i >= j,
// belongs to wave t % NW, slot t / NW: 105 tiles at P = 211 -> 7 slots x 4 registers per lane on 16 waves).  S is read from HBM
// exactly once and never written back; the right-hand side rides along as row P (so y = L^-1 b falls out of the factorisation).
// Per 16-column panel k:
//   A. the owner of the diagonal tile moves it through LDS into row-per-lane form and factors it with 16 pivots whose column updates are single
//      v_fmac_f64_dpp row_newbcast instructions (chol16_dpp: even 16-lane rows of the wave the tile's rows, odd rows the columns of the inverse),
//      leaves L_kk^-1 in LDS; three of its resident tiles wait in LDS meanwhile;
//   C. the owners of the tiles below it form L_ik = A_ik L_kk^-T (4 MFMAs; the accumulator -> operand transposition goes through
//      the tile's slice of the LDS panel) and publish L_ik there;
//   E. every owner of a trailing tile (i, j), j > k, subtracts L_ik L_jk^T (4 MFMAs, operands from the LDS panel).
// Back-substitution L^T x = y runs over the tiles still in registers: x_b = L_bb^-T t_b, then t_j -= L_bj^T x_b by the single
// owner of tile (b, j) -- no atomics anywhere, the summation order is fixed (bitwise reproducible); one barrier per block.
// With Dev::schur_plain_in_H the tiles without Schur products are read from Hpp (damping and fixed unknowns applied here), the others from S.
// Pivots with index >= P (the rhs row, padding rows) are forced to 1 and never flagged.
template <int J, int C> __device__ __forceinline__ void chol16_row_updates(double (&v)[16], int lo, int hi) {
  if constexpr (C + 3 <= 15) {
    chol_bcast_update4<C>(v[C], v[C + 1], v[C + 2], v[C + 3], v[J], lo, hi);
    chol16_row_updates<J, C + 4>(v, lo, hi);
  } else if constexpr (C <= 15) {
    chol_bcast_update<C>(v[C], v[J], lo, hi);
    chol16_row_updates<J, C + 1>(v, lo, hi);
  }
}
template <int J> __device__ __forceinline__ void chol16_from(double (&v)[16], double di, int nreal, int &bad) {
  v[J] *= di;
  const int lo = __double2loint(v[J]), hi = __double2hiint(v[J]);
  if constexpr (J < 15) {
    chol_bcast_update_first<J + 1>(v[J + 1], v[J], lo, hi);
    double di_next = 1.0;
    if (J + 1 < nreal) di_next = chol_pivot_rsqrt(readlane_d(v[J + 1], J + 1), bad);   // (uniform branch; without it -- the 16 pivots as one
    // basic block, so that the scheduler may put the row updates of pivot J into the bubbles of pivot J + 1's rsq / Newton chain -- the
    // diagonal tile took 8.2 k cycles instead of 7.6 k: measured, not kept)
    if constexpr (J < 14) chol16_row_updates<J, J + 2>(v, lo, hi);
    chol16_from<J + 1>(v, di_next, nreal, bad);
  }
}
// ---- The same diagonal tile with the broadcasts done by the data-parallel path of the fp64 ALU (v_fmac_f64_dpp row_newbcast:C: every lane of
// a 16-lane row reads lane C of ITS row; gfx90a+ allows exactly this DPP control on 64-bit operations) -- ONE instruction per column update instead
// of two v_readlane and an FMA.  A lone wave issues one vector instruction per ~5.3 clocks whether or not it depends on the one before
// (tools/fp64_latency_probe.hip), so the tile's time is its instruction count and nothing else.  Even rows of the wave (lanes 0-15, 32-47) hold
// the rows of the tile, odd rows (16-31, 48-63) the columns of the inverse; the odd rows need the multipliers L[C][J] of the EVEN row beside them,
// which v_permlane16_swap_b32 (gfx950) copies across once per pivot (m).  1 / sqrt(p) = v_rsq_f64 + one third-order step (2^-23 -> below 2^-60).
// A pivot that is not positive and finite turns everything after it into NaN (no select on the way): the last diagonal entry tells.
template <int C> __device__ __forceinline__ void chol_dpp_update(double &vc, double m, double vj) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(vc) : "v"(m), "v"(vj), "n"(C));
}
template <int C> __device__ __forceinline__ void chol_dpp_update4(double &v0, double &v1, double &v2, double &v3, double m, double vj) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%4, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, -%4, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, -%4, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, -%4, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)
      : "v"(m), "v"(vj), "n"(C), "n"(C + 1), "n"(C + 2), "n"(C + 3));
}
template <int J, int C> __device__ __forceinline__ void chol16_dpp_rest(double (&v)[16], double m) {
  if constexpr (C + 3 <= 15) {
    chol_dpp_update4<C>(v[C], v[C + 1], v[C + 2], v[C + 3], m, v[J]);
    chol16_dpp_rest<J, C + 4>(v, m);
  } else if constexpr (C <= 15) {
    chol_dpp_update<C>(v[C], m, v[J]);
    chol16_dpp_rest<J, C + 1>(v, m);
  }
}
__device__ __forceinline__ double chol_pivot_rsqrt3(double p) {
  const double y = __builtin_amdgcn_rsq(p);
  const double e = __builtin_fma(-(p * y), y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}
// FULL: all sixteen pivots are unknowns of the window (every tile but the last); otherwise pivots >= nreal are forced to 1
template <int J, bool FULL> __device__ __forceinline__ void chol16_dpp_from(double (&v)[16], double d, int nreal) {
  v[J] *= d;
  if constexpr (J < 15) {
    const unsigned lo = (unsigned)__double2loint(v[J]), hi = (unsigned)__double2hiint(v[J]);
    const auto slo = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);   // [0]: odd rows <- the even rows beside them
    const auto shi = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double m = __hiloint2double((int)shi[0], (int)slo[0]);
    chol_dpp_update<J + 1>(v[J + 1], m, v[J]);
    double dn = chol_pivot_rsqrt3(readlane_d(v[J + 1], J + 1));
    if constexpr (!FULL) dn = (J + 1 < nreal) ? dn : 1.0;
    if constexpr (J < 14) chol16_dpp_rest<J, J + 2>(v, m);
    chol16_dpp_from<J + 1, FULL>(v, dn, nreal);
  }
}
__device__ __forceinline__ void chol16_dpp(double (&v)[16], int nreal, int &bad) {
  const double d0 = chol_pivot_rsqrt3(readlane_d(v[0], 0));
  if (nreal >= 16) chol16_dpp_from<0, true>(v, d0, nreal);
  else chol16_dpp_from<0, false>(v, nreal > 0 ? d0 : 1.0, nreal);
  const double last = readlane_d(v[15], 15);   // every column after a bad pivot is NaN (0 x NaN included), so is this one
  if (!(last - last == 0.0)) bad = 1;
}

// the sum of a value over the four 16-lane row groups of the wave, in every lane (the tree of two __shfl_xor steps, without the LDS crossbar)
__device__ __forceinline__ double rowgroup_sum(double p) {
  unsigned lo = (unsigned)__double2loint(p), hi = (unsigned)__double2hiint(p);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);   // [0] = {g0, g0, g2, g2}, [1] = {g1, g1, g3, g3}
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double s = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
  lo = (unsigned)__double2loint(s); hi = (unsigned)__double2hiint(s);
  const auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);   // [0] = {lower half, lower half}, [1] = {upper, upper}
  const auto e = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)e[0], (int)c[0]) + __hiloint2double((int)e[1], (int)c[1]);
}
// acc += sum_k t[lane k of this lane's row] * lk[k], k = K .. 15
template <int K> __device__ __forceinline__ void dpp_dot16(double &acc, double t, const double (&lk)[16]) {
  if constexpr (K < 16) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(lk[K]), "n"(K));
    dpp_dot16<K + 1>(acc, t, lk);
  }
}
__device__ __forceinline__ double f64x4_get(const f64x4 &a, int r) { return r == 0 ? a[0] : (r == 1 ? a[1] : (r == 2 ? a[2] : a[3])); }

// (History of the diagonal tile.  With v_readlane broadcasts -- chol16_from above, still the cross-check in tools/chol16_probe.hip -- a BLOCKED
//  variant (four blocks of four pivots, rank-4 updates as two v_mfma_f64_16x16x4) and a reciprocal-based pivot chain were built early in round 4
//  and found no faster: 8.7 k / 9.0 k vs 8.3 k cycles per tile.  The explanation given then -- "300 dependent cycles per pivot" -- was wrong: a
//  lone wave issues one fp64 instruction per ~5.3 clocks whether it depends on the previous one or not (tools/fp64_latency_probe.hip), the tile
//  was 760 instructions, and 2.2 k of its clocks were the tile's LOAD, compiled into 16 branches with a ds_read and an s_waitcnt each.
//  chol16_dpp + the branch-free load: 3.2 k clocks per tile.)
template <int NW, int NS> __global__ __launch_bounds__(64 * NW) void k_cholesky_tiles(Dev d) {
  constexpr int NT = 64 * NW, TS = 16 * 17;    // a 16 x 16 block in LDS: row stride 17
  const int w = blockIdx.x;
  Lm &lm = d.lm[w];
  if (lm.status || lm.ls_active) return;
  const WinMeta &m = d.wins[w];
  const int P = m.P, ldh = m.ldh, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int NTR = P / 16 + 1, ntiles = NTR * (NTR + 1) / 2, ip = P / 16, rp = P % 16;   // the rhs row P sits in tile row ip, local row rp
  extern __shared__ __attribute__((aligned(16))) double smt[];
  double *Id = smt;                    // [TS] a 16 x 16 identity: the diagonal tile's inverse lanes start from it
  double *Li = Id + TS;                // [NTR][TS] inverses of the diagonal blocks, Li[b][j * 17 + k] = Linv_b[j][k]
  double *Pn = Li + NTR * TS;          // [NTR][TS] panel: Pn[i][m * 17 + c] = L_ik[m][c] of the current panel
  double *tv = Pn + NTR * TS;          // [16 NTR] y, then the running right-hand side of the back-substitution
  double *xs = tv + 16 * NTR;          // [16 NTR] solution
  int &s_fail = *reinterpret_cast<int *>(xs + 16 * NTR);
  double *park = xs + 16 * NTR + 2;    // [12][64] three tiles of the wave that factors a diagonal tile wait here meanwhile (see step A)
  const double *S = d.S + m.H0, *y = d.rhs + m.p0;
  const double *Hc = d.HppS[lm.cur] + m.H0;
  const bool from_h = d.schur_plain_in_H != 0;
  const int K6 = 6 * m.K;
  // (the same tile classification as k_schur_window_f64: W is non-zero in the knot columns, the line-delay column and the rhs row)
  auto nz_row = [&](int b) { return (16 * b < K6) || (P >= 16 * b && P - 1 < 16 * b + 16); };
  auto nz_col = [&](int b) { return (16 * b < K6) || (P - 1 >= 16 * b && P - 1 < 16 * b + 16); };
  if (tid == 0) s_fail = 0;
  for (int i = tid; i < 16 * NTR; i += NT) tv[i] = 0.0;
  // activity of the unknowns as four 64-bit masks in SGPRs (each wave builds its own: four byte loads per lane, no LDS, no barrier)
  unsigned long long amask[4] = {0ull, 0ull, 0ull, 0ull};
  if (from_h) {
    unsigned char ab[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ab[k] = d.active[m.u0 + min(lane + 64 * k, P - 1)];
#pragma unroll
    for (int k = 0; k < 4; ++k) amask[k] = __ballot(lane + 64 * k < P && ab[k] != 0);
  }
  auto active_bit = [&](int i) {   // (i < 256; lane-variant)
    const unsigned long long wlo = (i & 128) ? amask[2] : amask[0], whi = (i & 128) ? amask[3] : amask[1];
    return (int)((((i & 64) ? whi : wlo) >> (i & 63)) & 1ull);
  };
  for (int i = tid; i < TS; i += NT) Id[i] = (i / 17 == i % 17) ? 1.0 : 0.0;
  // ---- this wave's tiles (SGPRs) and their contents
  // SPARSITY: tile (i, c) of the factor is empty for c < env_tile[i] (host_pack.hpp: plan_sparsity; fill stays inside the row envelope), so
  // panel k neither solves nor updates with a tile whose row starts after it: ek = the first panel either row of the tile takes part in.
  // (The tiles themselves are all resident -- the empty ones hold exact zeros -- so loads, layout and the back-substitution do not change.)
  int ti[NS], tj[NS], ek[NS];
  f64x4 acc[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const int t = wave + NW * q;
    int a, b;
    tile_decode(min(t, ntiles - 1), a, b);
    ti[q] = __builtin_amdgcn_readfirstlane(t < ntiles ? a : -1);
    tj[q] = __builtin_amdgcn_readfirstlane(t < ntiles ? b : 1 << 20);   // (never equal to a panel, never a trailing tile: ti < tj)
    ek[q] = __builtin_amdgcn_readfirstlane(max(d.env_tile[m.tr0 + a], d.env_tile[m.tr0 + b]));
    // unconditional loads on clamped addresses straight into the tile registers; fixed up below
    const bool plain = from_h && !(nz_row(a) && nz_col(b));   // (wave-uniform)
    const double *src = plain ? Hc : S;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rc = min(16 * a + q4 + 4 * r, P - 1);
      acc[q][r] = src[(long long)rc * ldh + min(16 * b + l15, rc)];
    }
  }
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    if (ti[q] < 0) continue;
    const int col = 16 * tj[q] + l15;
    if (from_h && !(nz_row(ti[q]) && nz_col(tj[q]))) {   // a tile without Schur products, straight from Hpp: damping and fixed unknowns here
      const int a_j = active_bit(col);
      double ddiag = 0.0;   // (a diagonal tile among them: a few bias-bias blocks per window; one L2 round trip for its wave)
      if (ti[q] == tj[q]) ddiag = d.dd[m.u0 + min(col, P - 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti[q] + q4 + 4 * r;
        acc[q][r] = (active_bit(row) & a_j) ? acc[q][r] + (row == col ? ddiag : 0.0) : (row == col ? 1.0 : 0.0);
      }
    }
    if (ti[q] == tj[q]) {             // diagonal tile: the upper half is not stored in S
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] = (col <= 16 * ti[q] + q4 + 4 * r) ? acc[q][r] : 0.0;
    }
    if (ti[q] == ip) {                // tile row of the rhs row P; identity beyond it
      const double yv = y[min(col, P - 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ip + q4 + 4 * r;
        acc[q][r] = row < P ? acc[q][r] : (row == P ? (col < P ? yv : 0.0) : (row == col ? 1.0 : 0.0));
      }
    }
  }
  __syncthreads();
  long long *dbg = (d.dbg && w == 0) ? d.dbg : nullptr;   // CTVIO_DEBUG_STAMPS: clock64 of thread 0 at the phase boundaries
  int dbi = 0;
#define CTV_STAMP() do { if (dbg && tid == 0 && dbi < 30) dbg[dbi++] = clock64(); } while (0)
  CTV_STAMP();
  for (int k = 0; k < NTR; ++k) {
    int opq;   // (a zero the compiler cannot see through: the per-slot LDS addresses of steps C and E are recomputed each panel -- one add each --
    asm volatile("s_mov_b32 %0, 0" : "=s"(opq));   // instead of being kept as 14 loop-invariant registers, which no longer fit beside the tile)
    double *Pnk = Pn + opq;
    // ---- A. diagonal tile (k, k)
    const int td = k * (k + 1) / 2 + k, od = td % NW, sd = td / NW;
    if (wave == od) {
      double *Dg = Li + k * TS;
#pragma unroll
      for (int q = 0; q < NS; ++q)
        if (q == sd) {
#pragma unroll
          for (int r = 0; r < 4; ++r) Dg[(q4 + 4 * r) * 17 + l15] = acc[q][r];
        }
      // The tile's 16 columns, the multipliers and the pivot chain do not fit beside seven resident tiles in 128 registers: three tiles wait in
      // LDS meanwhile (left to the compiler they went to scratch: a dozen scratch round trips per panel on the critical path).
      if constexpr (NS >= 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { park[(3 * r) * 64 + lane] = acc[0][r]; park[(3 * r + 1) * 64 + lane] = acc[1][r]; park[(3 * r + 2) * 64 + lane] = acc[2][r]; }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      double v[16];
      int opaque0;   // a zero the compiler cannot see through: without it the 16 identity columns below are hoisted out of the panel
      asm volatile("s_mov_b32 %0, 0" : "=s"(opaque0));   // loop as loop invariants and, for lack of registers, kept in scratch
      // even rows of the wave: the tile's rows (whole rows: the factorisation never reads the upper half); odd rows: the identity, from LDS as
      // well.  (With selects -- lane < 16 ? (c <= lane ? a : 0) : (c == lane) -- the compiler sank each of the 16 LDS reads into its own branch
      // with its own s_waitcnt: 2.2 k clocks per tile for the load alone, tools/chol16_probe.hip.)
      const double *src = ((lane & 16) ? Id : Dg) + (l15 + opaque0) * 17;
#pragma unroll
      for (int c = 0; c < 16; ++c) v[c] = src[c];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();      // every lane has read its row before the block is overwritten with the inverse
      const int nreal = P - 16 * k;         // pivots below this are real; the rhs row and the padding rows are not factored
      int bad = 0;
      chol16_dpp(v, nreal, bad);
      if (lane >= 16 && lane < 32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) Dg[i * 17 + l15] = v[i];   // Linv[i][column l15]
      }
      if (k == ip && lane == rp) {          // the part of y inside the last diagonal tile: L[P][16 ip + c], c < rp
#pragma unroll
        for (int c = 0; c < 16; ++c) if (c < rp) tv[16 * ip + c] = v[c];
      }
      if (lane == 0 && bad) s_fail = 1;
      if constexpr (NS >= 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[0][r] = park[(3 * r) * 64 + lane]; acc[1][r] = park[(3 * r + 1) * 64 + lane]; acc[2][r] = park[(3 * r + 2) * 64 + lane]; }
      }
    }
    if (k < 4) CTV_STAMP();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    if (k < 4) CTV_STAMP();
    // ---- C. L_ik = A_ik L_kk^-T for the tiles below the diagonal one
    const double *Lk = Li + k * TS;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      if (tj[q] != k || ti[q] <= k || k < ek[q]) continue;   // (uniform; an empty tile stays zero and publishes nothing)
      double *blk = Pnk + ti[q] * TS;
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[(q4 + 4 * r) * 17 + l15] = acc[q][r];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      double a[4], b[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) { a[s4] = blk[l15 * 17 + 4 * s4 + q4]; b[s4] = Lk[l15 * 17 + 4 * s4 + q4]; }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();          // operands are in registers before the slice is overwritten
      f64x4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], c, 0, 0, 0);
      acc[q] = c;
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[(q4 + 4 * r) * 17 + l15] = c[r];
      if (ti[q] == ip && q4 == (rp & 3)) tv[16 * k + l15] = f64x4_get(c, rp >> 2);   // y: row P of L
    }
    if (k < 4) CTV_STAMP();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    if (k < 4) CTV_STAMP();
    // ---- E. trailing tiles (i, j), j > k: A_ij -= L_ik L_jk^T
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      if (ti[q] < 0 || tj[q] <= k || tj[q] >= (1 << 20) || k < ek[q]) continue;   // (uniform; L_ik or L_jk is empty)
      const double *pa = Pnk + ti[q] * TS + l15 * 17 + q4, *pb = Pnk + tj[q] * TS + l15 * 17 + q4;
      double a[4], b[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) { a[s4] = -pa[4 * s4]; b[s4] = pb[4 * s4]; }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], acc[q], 0, 0, 0);
    }
    // (K-step outermost, so that consecutive MFMAs go to different tiles, was measured slower: 5.7 k cycles for the first panel's
    //  updates either way, and the diagonal tiles waited longer.  LOOK-AHEAD -- the owner of tile (k + 1, k + 1) updates that tile
    //  first and factors it while the other waves do their updates, the LDS panel double buffered, its own remaining updates
    //  deferred to the next panel -- was built and measured slower too: 1.63 vs 1.30 ms per 16 single-window factorisations,
    //  13.5 vs 10.6 ms per 2048-window solve.  The pivot chain takes ~12 k cycles instead of ~7 k when the other 15 waves are
    //  busy on the same SIMDs / LDS, s_setprio 3 does not change that, and step C grows by the pending updates.)
    //  ROUND 6: the same overlap with the pivot chain on a wave AND SIMD of its own (wave 0 factors, waves 4 / 8 / 12 -- its SIMD mates -- exit,
    //  12 update waves x 9 tiles on the other three SIMDs, the owner of (k + 1, k) also updating (k + 1, k + 1)): parity-green, and no faster --
    //  69 vs 71 us per factorisation, 45.8 vs 46.4 ms per 2048-window solve.  Early panels become bound by the updates on three SIMDs, late
    //  panels by A -> L_(k+1,k) -> update of (k+1,k+1) -> A with the two middle steps serial in one wave (6.9 k cycles per panel against
    //  ~5.7 k here): profiles/r06_chol_chain_experiment.txt.)
    // (the next panel's step C overwrites the LDS panel only after the barrier that follows its step A)
    if (k < 4) CTV_STAMP();
  }
  CTV_STAMP();
  __syncthreads();
  CTV_STAMP();
  // ---- back-substitution L^T x = y over the tiles in registers: ONE barrier per block.  After x_b is known, the only contribution t_{b-1}
  // still lacks is that of tile (b, b - 1): its owner finishes t_{b-1} in registers and forms x_{b-1} = L_{b-1,b-1}^-T t_{b-1} at once (the sixteen
  // t[k] read across the 16-lane rows by v_fmac_f64_dpp row_newbcast, the sum over the four row groups by v_permlane16/32_swap -- no LDS round
  // trip on the chain); the owners of the other tiles (b, j) subtract their parts from t_j in LDS meanwhile.  (x_b by one wave, barrier, the
  // updates, barrier: 19.7 k of a factorisation's 143 k cycles.)
  if (wave == ((NTR - 1) % NW)) {   // x_b[j] = sum_k Linv[k][j] t[k] of the last block: lane (q4, j = l15) sums k = 4 q4 .. 4 q4 + 3
    const int bl = NTR - 1;
    const double *Lb = Li + bl * TS;
    double xa = 0.0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xa += Lb[(4 * q4 + kk) * 17 + l15] * tv[16 * bl + 4 * q4 + kk];
    xa += __shfl_xor(xa, 16);
    xa += __shfl_xor(xa, 32);
    if (q4 == 0) xs[16 * bl + l15] = xa;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();
  for (int b = NTR - 1; b >= 1; --b) {
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      if (ti[q] != b || tj[q] >= b) continue;   // tiles (b, j), j < b: t_j -= L_bj^T x_b
      double xb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[r] = xs[16 * b + q4 + 4 * r];
      if (tj[q] == b - 1) {                     // (uniform) the chain
        const double *Lb = Li + (b - 1) * TS;
        double lk[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) lk[kk] = Lb[kk * 17 + l15];
        const double tb = tv[16 * (b - 1) + l15];
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) part += acc[q][r] * xb[r];
        const double t = tb - rowgroup_sum(part);   // t_{b-1}[l15], in every row group
        double xa = 0.0;
        dpp_dot16<0>(xa, t, lk);
        if (q4 == 0) xs[16 * (b - 1) + l15] = xa;
      } else {
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) part += acc[q][r] * xb[r];
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (q4 == 0) tv[16 * tj[q] + l15] -= part;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
  }
  CTV_STAMP();
  double *x = d.delta + m.u0;
  for (int i = tid; i < P; i += NT) x[i] = xs[i];
  if (tid == 0) lm.chol_fail = s_fail;
#undef CTV_STAMP
}

// ---- The same factorisation as a DATA-FLOW of waves (round 6): the critical path of a tile Cholesky is
//      factor (k, k)  ->  L_(k+1,k) = A_(k+1,k) L_kk^-T  ->  A_(k+1,k+1) -= L_(k+1,k) L_(k+1,k)^T  ->  factor (k + 1, k + 1)  -> ...
// and k_cholesky_tiles puts two workgroup barriers and every other tile's work into each of its fourteen links (clock stamps of one window,
// CTVIO_DEBUG_STAMPS: 3.8 k + 1.7 k + 3.5 k .. 0.4 k cycles per panel, 63 % of the wave cycles parked).  Here ONE wave -- the CHAIN wave, wave 0 --
// runs that path and nothing else (5.0 k cycles per link: 1.45 k for the two products, 3.5 k for the tile), with both tiles of a link handed to it
// through LDS; the other fifteen waves (seven tile slots each, dealt in column-major order) are UPDATE waves that form the L_ik of the other rows
// and apply the trailing updates, in panel order, BEHIND the chain.  There is no workgroup barrier inside the factorisation: every hand-over is a
// flag in LDS that the consumer polls (s_sleep between polls, every loop bounded -- a bound trips the window's chol_fail instead of hanging):
//   F_inv[k]   L_kk^-1 is in Li[k]                                   chain wave   -> update waves (their L_ik of panel k)
//   F_row[i]   = c + 1: L_ic of panel c is in Pn[c % 3][i]            whoever formed it (chain wave for i = c + 1) -> every trailing update
//   F_park[k]  = 2: tiles (k, k - 1) and (k, k), with every update up to panel k - 2 applied, are parked in Ls[k] / Li[k]
//                                                                      their owners -> chain wave
//   done_E[c]  = 15: every update wave is through with panel c       update waves -> whoever overwrites that panel's buffer (panel c + 3)
// The three panel buffers bound the skew between waves to three panels.  Tile (k, k - 1) comes back from the chain wave as L_(k,k-1) in Ls[k], where
// its owner picks it up for the back-substitution (unchanged, one barrier per block, the chain wave gone by then).  Every tile has one owner and
// receives its updates in panel order: the arithmetic -- and so the bitwise reproducibility of the deterministic mode -- does not depend on timing.
// Measured (tools/r6_chol.sh, profiles/r06_chol_flow.txt): 64 us per factorisation of one window against 70 (k_cholesky_tiles), one window's
// solve 2.72 - 2.76 ms against 2.81; 2048 windows per launch 45.5 - 45.9 ms per solve against 45.8 - 46.0.  The chain wave still waits 1 - 6 k cycles
// per panel for its tiles: in-order update waves with blocking waits form convoys.  Tried on the way: the chain wave alone on its SIMD (waves 4 /
// 8 / 12 exit, twelve update waves x nine tiles): the updates on three SIMDs cannot keep up (70 -> 78 us); barriers instead of flags with the
// chain overlapped (profiles/r06_chol_chain_experiment.txt): 69 - 71 us.
// Also tried: the three tiles (r, r - 2), (r, r - 1), (r, r) of a row with ONE owner who applies panel r - 2 to the last two AHEAD of its other work
// and parks them at once (the chain never waited in panels 1 - 3, and 5 - 6 k cycles in every later one: 84 us per factorisation).
constexpr int CHOL_NU = 15, CHOL_NS = 7;   // update waves, tile slots per wave (105 tiles at P = 211)
struct CholMap { signed char ti[15][CHOL_NU][CHOL_NS], tj[15][CHOL_NU][CHOL_NS]; };   // [tile rows NTR][update wave][slot]: tile (ti, tj), -1 = empty slot
constexpr CholMap make_chol_map() {
  CholMap mp{};
  for (int n = 0; n < 15; ++n) {
    for (int o = 0; o < CHOL_NU; ++o)
      for (int q = 0; q < CHOL_NS; ++q) { mp.ti[n][o][q] = -1; mp.tj[n][o][q] = -1; }
    int t = 0;
    for (int j = 0; j < n; ++j)          // column-major: a wave meets its tiles in the order the panels need them
      for (int i = j; i < n; ++i, ++t) { mp.ti[n][t % CHOL_NU][t / CHOL_NU] = (signed char)i; mp.tj[n][t % CHOL_NU][t / CHOL_NU] = (signed char)j; }
  }
  return mp;
}
constexpr bool chol_map_complete(const CholMap &mp) {   // every tile of every size exactly once
  for (int n = 0; n < 15; ++n) {
    int seen = 0;
    for (int o = 0; o < CHOL_NU; ++o)
      for (int q = 0; q < CHOL_NS; ++q) {
        const int i = mp.ti[n][o][q], j = mp.tj[n][o][q];
        if (i < 0) continue;
        if (j < 0 || j > i || i >= n) return false;
        seen += 1;
      }
    if (seen != n * (n + 1) / 2) return false;
  }
  return true;
}
static_assert(chol_map_complete(make_chol_map()), "CHOL_MAP must hold every tile once, within seven slots per update wave");
__constant__ const CholMap CHOL_MAP = make_chol_map();

// poll an LDS flag until it reaches `target` (wave-uniform); a bound instead of a hang
__device__ __forceinline__ void chol_wait(volatile int *f, int target, int &fail) {
  int n = 0;
  while (__builtin_amdgcn_readfirstlane(*f) < target) {
    if (++n > (1 << 20)) { fail = 1; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
// everything this wave wrote to LDS is there; then the flag
__device__ __forceinline__ void chol_post(volatile int *f, int value, int lane) {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) *f = value;
}
__device__ __forceinline__ void chol_count(int *f, int lane) {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) atomicAdd(f, 1);
}

__global__ __launch_bounds__(1024) void k_cholesky_flow(Dev d) {
  constexpr int NU = CHOL_NU, NS = CHOL_NS, NTU = 64 * NU, TS = 16 * 17, NPB = 3;    // update waves, tile slots per wave; a 16 x 16 block in LDS: row stride 17
  const int w = blockIdx.x;
  Lm &lm = d.lm[w];
  if (lm.status || lm.ls_active) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool chain = wave == 0;
  const int uw = wave - 1;                      // update wave index 0 .. 14
  const WinMeta &m = d.wins[w];
  const int P = m.P, ldh = m.ldh;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int NTR = P / 16 + 1, ip = P / 16, rp = P % 16;   // the rhs row P sits in tile row ip, local row rp
  extern __shared__ __attribute__((aligned(16))) double smt[];
  double *Id = smt;                    // [TS] a 16 x 16 identity: the diagonal tile's inverse lanes start from it
  double *Li = Id + TS;                // [NTR][TS] diagonal blocks on their way to the chain wave, then their inverses: Li[b][j * 17 + k] = Linv_b[j][k]
  double *Ls = Li + NTR * TS;          // [NTR][TS] tile (b, b - 1) on its way to the chain wave, then L_(b,b-1) (row major) for the back-substitution
  double *Pn = Ls + NTR * TS;          // [NPB][NTR][TS] panels: Pn[c % NPB][i][m * 17 + cc] = L_ic[m][cc]
  double *tv = Pn + NPB * NTR * TS;      // [16 NTR] y, then the running right-hand side of the back-substitution
  double *xs = tv + 16 * NTR;          // [16 NTR] solution
  int *fl = reinterpret_cast<int *>(xs + 16 * NTR);
  int &s_fail = fl[0];
  volatile int *F_inv = fl + 16, *F_row = fl + 32;
  int *F_park = fl + 48, *done_E = fl + 64;
  long long *dbg = (d.dbg && w == 0) ? d.dbg : nullptr;   // CTVIO_DEBUG_STAMPS: clock64 of the chain wave at its steps
#define CTV_BAR() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); } while (0)
  if (chain) {
    // ================================================================ the chain wave
    int dbi = 0, fail = 0;
#define CTV_STAMP() do { if (dbg && lane == 0 && dbi < 30) dbg[dbi++] = clock64(); } while (0)
    for (int i = lane; i < 80; i += 64) fl[i] = 0;
    for (int i = lane; i < TS; i += 64) Id[i] = (i / 17 == i % 17) ? 1.0 : 0.0;
    for (int i = lane; i < 16 * NTR; i += 64) tv[i] = 0.0;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    CTV_BAR();                                   // tile (0, 0) is in LDS, the flags are clear
    CTV_STAMP();
    for (int k = 0; k < NTR; ++k) {
      double *Dg = Li + k * TS;
      if (k > 0) {
        chol_wait(reinterpret_cast<volatile int *>(F_park + k), 2, fail);
        if (k < 4) CTV_STAMP();
        // ---- L_(k,k-1) = A_(k,k-1) L_(k-1,k-1)^-T: both operands are in LDS in row-major form (no accumulator -> operand round trip)
        double *blk = Ls + k * TS;
        const double *Lk = Li + (k - 1) * TS;
        double a[4], b[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { a[s4] = blk[l15 * 17 + 4 * s4 + q4]; b[s4] = Lk[l15 * 17 + 4 * s4 + q4]; }
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Dg[(q4 + 4 * r) * 17 + l15];     // tile (k, k), requested with the operands
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        f64x4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], c, 0, 0, 0);
        if (k > NPB) chol_wait(reinterpret_cast<volatile int *>(done_E + (k - 1 - NPB)), NU, fail);   // the buffer of panel k - 1 - NPB is free
        double *pub = Pn + ((k - 1) % NPB) * NTR * TS + k * TS;
#pragma unroll
        for (int r = 0; r < 4; ++r) { blk[(q4 + 4 * r) * 17 + l15] = c[r]; pub[(q4 + 4 * r) * 17 + l15] = c[r]; }
        if (k == ip && q4 == (rp & 3)) tv[16 * (k - 1) + l15] = f64x4_get(c, rp >> 2);   // y: row P of L
        chol_post(F_row + k, k, lane);           // L_(k,k-1) is published: panel k - 1 can be applied to the tiles of row / column k
        // ---- A_kk -= L_(k,k-1) L_(k,k-1)^T
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { b[s4] = blk[l15 * 17 + 4 * s4 + q4]; a[s4] = -b[s4]; }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Dg[(q4 + 4 * r) * 17 + l15] = acc[r];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (k < 4) CTV_STAMP();
      }
      double v[16];
      int opaque0;   // a zero the compiler cannot see through (the 16 identity columns would be hoisted out of the panel loop otherwise)
      asm volatile("s_mov_b32 %0, 0" : "=s"(opaque0));
      // even rows of the wave: the tile's rows (whole rows: the factorisation never reads the upper half); odd rows: the identity, from LDS too
      const double *src = ((lane & 16) ? Id : Dg) + (l15 + opaque0) * 17;
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) v[cc] = src[cc];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();          // every lane has read its row before the block is overwritten with the inverse
      const int nreal = P - 16 * k;             // pivots below this are real; the rhs row and the padding rows are not factored
      int bad = 0;
      chol16_dpp(v, nreal, bad);
      if (lane >= 16 && lane < 32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) Dg[i * 17 + l15] = v[i];   // Linv[i][column l15]
      }
      if (k == ip && lane == rp) {              // the part of y inside the last diagonal tile: L[P][16 ip + c], c < rp
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) if (cc < rp) tv[16 * ip + cc] = v[cc];
      }
      if (bad) fail = 1;
      chol_post(F_inv + k, 1, lane);
      CTV_STAMP();
    }
    if (lane == 0 && fail) s_fail = 1;
    // back-substitution: x of the last block, x_b[j] = sum_k Linv[k][j] t[k]: lane (q4, j = l15) sums k = 4 q4 .. 4 q4 + 3
    {
      const int bl = NTR - 1;
      const double *Lb = Li + bl * TS;
      double xa = 0.0;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) xa += Lb[(4 * q4 + kk) * 17 + l15] * tv[16 * bl + 4 * q4 + kk];
      xa += __shfl_xor(xa, 16);
      xa += __shfl_xor(xa, 32);
      if (q4 == 0) xs[16 * bl + l15] = xa;
    }
    CTV_STAMP();
    CTV_BAR();                                   // the factorisation is complete (the update waves arrive here when they are through)
    return;                                      // (the update waves finish the back-substitution among themselves)
#undef CTV_STAMP
  }
  // ================================================================== update waves
  const int utid = 64 * uw + lane;
  int fail = 0;
  const double *S = d.S + m.H0, *y = d.rhs + m.p0;
  const double *Hc = d.HppS[lm.cur] + m.H0;
  const bool from_h = d.schur_plain_in_H != 0;
  const int K6 = 6 * m.K;
  // (the same tile classification as k_schur_window_f64: W is non-zero in the knot columns, the line-delay column and the rhs row)
  auto nz_row = [&](int b) { return (16 * b < K6) || (P >= 16 * b && P - 1 < 16 * b + 16); };
  auto nz_col = [&](int b) { return (16 * b < K6) || (P - 1 >= 16 * b && P - 1 < 16 * b + 16); };
  // activity of the unknowns as four 64-bit masks in SGPRs (each wave builds its own: four byte loads per lane, no LDS, no barrier)
  unsigned long long amask[4] = {0ull, 0ull, 0ull, 0ull};
  if (from_h) {
    unsigned char ab[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ab[k] = d.active[m.u0 + min(lane + 64 * k, P - 1)];
#pragma unroll
    for (int k = 0; k < 4; ++k) amask[k] = __ballot(lane + 64 * k < P && ab[k] != 0);
  }
  auto active_bit = [&](int i) {   // (i < 256; lane-variant)
    const unsigned long long wlo = (i & 128) ? amask[2] : amask[0], whi = (i & 128) ? amask[3] : amask[1];
    return (int)((((i & 64) ? whi : wlo) >> (i & 63)) & 1ull);
  };
  // ---- this wave's tiles (SGPRs) and their contents.  SPARSITY: tile (i, c) of the factor is empty for c < env_tile[i] (host_pack.hpp:
  // plan_sparsity; fill stays inside the row envelope), so panel k neither solves nor updates with a tile whose row starts after it: ek = the
  // first panel either row of the tile takes part in.  (The tiles are all resident -- the empty ones hold exact zeros.)
  int ti[NS], tj[NS], ek[NS];
  f64x4 acc[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const int a0 = CHOL_MAP.ti[NTR][uw][q], b0 = CHOL_MAP.tj[NTR][uw][q];
    const int a = max(a0, 0), b = max(b0, 0);
    ti[q] = __builtin_amdgcn_readfirstlane(a0);
    tj[q] = __builtin_amdgcn_readfirstlane(a0 >= 0 ? b0 : 1 << 20);   // (never equal to a panel, never a trailing tile: ti < tj)
    ek[q] = __builtin_amdgcn_readfirstlane(max(d.env_tile[m.tr0 + a], d.env_tile[m.tr0 + b]));
    // unconditional loads on clamped addresses straight into the tile registers; fixed up below
    const bool plain = from_h && !(nz_row(a) && nz_col(b));   // (wave-uniform)
    const double *src = plain ? Hc : S;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rc = min(16 * a + q4 + 4 * r, P - 1);
      acc[q][r] = src[(long long)rc * ldh + min(16 * b + l15, rc)];
    }
  }
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    if (ti[q] < 0) continue;
    const int col = 16 * tj[q] + l15;
    if (from_h && !(nz_row(ti[q]) && nz_col(tj[q]))) {   // a tile without Schur products, straight from Hpp: damping and fixed unknowns here
      const int a_j = active_bit(col);
      double ddiag = 0.0;   // (a diagonal tile among them: a few bias-bias blocks per window; one L2 round trip for its wave)
      if (ti[q] == tj[q]) ddiag = d.dd[m.u0 + min(col, P - 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti[q] + q4 + 4 * r;
        acc[q][r] = (active_bit(row) & a_j) ? acc[q][r] + (row == col ? ddiag : 0.0) : (row == col ? 1.0 : 0.0);
      }
    }
    if (ti[q] == tj[q]) {             // diagonal tile: the upper half is not stored in S
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] = (col <= 16 * ti[q] + q4 + 4 * r) ? acc[q][r] : 0.0;
    }
    if (ti[q] == ip) {                // tile row of the rhs row P; identity beyond it
      const double yv = y[min(col, P - 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ip + q4 + 4 * r;
        acc[q][r] = row < P ? acc[q][r] : (row == P ? (col < P ? yv : 0.0) : (row == col ? 1.0 : 0.0));
      }
    }
    if (ti[q] == 0) {                 // tile (0, 0): to the chain wave as it is (row 1 follows in iteration 0 of the loop below)
      double *dst = Li;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(q4 + 4 * r) * 17 + l15] = acc[q][r];
    }
  }
  CTV_BAR();
  // Iteration c of an update wave: panel c - 1 applied to its NEAR trailing tiles (columns c and c + 1: the tiles whose L_ic the next panel needs, and
  // the two tiles of row c + 1 that go to the chain wave), then -- as soon as the chain wave has L_cc^-1 -- the L_ic of its tiles of column c, and
  // panel c - 1 applied to the rest, which fills the wait for L_cc^-1 when that is not there yet.
  for (int c = 0; c < NTR; ++c) {
    int opq;   // (a zero the compiler cannot see through: the per-slot LDS addresses are recomputed each panel -- one add each -- instead of
    asm volatile("s_mov_b32 %0, 0" : "=s"(opq));   // being kept as loop-invariant registers beside the tiles)
    double *Pnc = Pn + (c % NPB) * NTR * TS + opq;                      // panel c: written here
    const double *Pnp = Pn + ((c + NPB - 1) % NPB) * NTR * TS + opq;    // panel c - 1: applied here
    unsigned seen = 0;     // rows whose L_(i,c-1) this wave has already found published
    // A_ij -= L_(i,c-1) L_(j,c-1)^T for this wave's tiles (i, j), j >= c, i >= c + 1: part 0 = the near columns (c, c + 1), parts 1 / 2 = the far
    // columns in the first / second half of the wave's slots
    auto apply_prev = [&](int part) {
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        if (ti[q] < c + 1 || tj[q] < c || tj[q] >= (1 << 20)) continue;   // (uniform)
        if ((tj[q] <= c + 1 ? 0 : (q < (NS + 1) / 2 ? 1 : 2)) != part) continue;
        if (c - 1 >= ek[q]) {
          if (!((seen >> ti[q]) & 1u)) { chol_wait(F_row + ti[q], c, fail); seen |= 1u << ti[q]; }
          if (!((seen >> tj[q]) & 1u)) { chol_wait(F_row + tj[q], c, fail); seen |= 1u << tj[q]; }
          const double *pa = Pnp + ti[q] * TS + l15 * 17 + q4, *pb = Pnp + tj[q] * TS + l15 * 17 + q4;
          double a[4], b[4];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) { a[s4] = -pa[4 * s4]; b[s4] = pb[4 * s4]; }
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], acc[q], 0, 0, 0);
        }
        if (ti[q] == c + 1) {   // (uniform) tiles (c + 1, c) and (c + 1, c + 1): every update up to panel c - 1 is in; to the chain wave
          double *dst = (tj[q] == c + 1 ? Li : Ls) + (c + 1) * TS;
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(q4 + 4 * r) * 17 + l15] = acc[q][r];
          chol_count(F_park + c + 1, lane);
        }
      }
    };
    auto form_column = [&]() {   // L_ic = A_ic L_cc^-T for this wave's tiles of column c, rows >= c + 2 (row c + 1 is the chain wave's), through the
      const double *Lc = Li + c * TS;   // tile's slice of the panel buffer (accumulator -> operand transposition), published there
      bool first = true;
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        if (tj[q] != c || ti[q] < c + 2 || c < ek[q]) continue;   // (uniform; an empty tile stays zero and publishes nothing)
        if (first) {
          chol_wait(F_inv + c, 1, fail);
          if (c >= NPB) chol_wait(reinterpret_cast<volatile int *>(done_E + (c - NPB)), NU, fail);   // nobody reads panel c - NPB any more
          first = false;
        }
        double *blk = Pnc + ti[q] * TS;
#pragma unroll
        for (int r = 0; r < 4; ++r) blk[(q4 + 4 * r) * 17 + l15] = acc[q][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        double a[4], b[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { a[s4] = blk[l15 * 17 + 4 * s4 + q4]; b[s4] = Lc[l15 * 17 + 4 * s4 + q4]; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();          // operands are in registers before the slice is overwritten
        f64x4 cc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) cc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], cc, 0, 0, 0);
        acc[q] = cc;
#pragma unroll
        for (int r = 0; r < 4; ++r) blk[(q4 + 4 * r) * 17 + l15] = cc[r];
        if (ti[q] == ip && q4 == (rp & 3)) tv[16 * c + l15] = f64x4_get(cc, rp >> 2);   // y: row P of L
        chol_post(F_row + ti[q], c + 1, lane);
      }
    };
    apply_prev(0);        // (c = 0: nothing to apply -- tiles (1, 0) and (1, 1) go to the chain wave as they are)
    bool formed = false;  // the column is formed as soon as L_cc^-1 is seen -- checked before each part of the far work, which fills the wait
    auto try_form = [&](bool must) {
      if (formed) return;
      if (!must && __builtin_amdgcn_readfirstlane(F_inv[c]) == 0) return;
      asm volatile("" ::: "memory");
      form_column();
      formed = true;
    };
    try_form(false);
    if (c > 0) apply_prev(1);
    try_form(false);
    if (c > 0) apply_prev(2);
    try_form(true);
    if (c > 0) chol_count(done_E + (c - 1), lane);   // this wave is through with panel c - 1
  }
  if (lane == 0 && fail) s_fail = 1;
  CTV_BAR();                                      // the factorisation is complete; x of the last block (chain wave)
  // the tiles (b, b - 1) come back from the chain wave as L_(b,b-1)
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    if (ti[q] < 1 || tj[q] != ti[q] - 1) continue;
    const double *src = Ls + ti[q] * TS;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = src[(q4 + 4 * r) * 17 + l15];
  }
  // ---- back-substitution L^T x = y over the tiles in registers: ONE barrier per block.  After x_b is known, the only contribution t_{b-1}
  // still lacks is that of tile (b, b - 1): its owner finishes t_{b-1} in registers and forms x_{b-1} = L_{b-1,b-1}^-T t_{b-1} at once (the sixteen
  // t[k] read across the 16-lane rows by v_fmac_f64_dpp row_newbcast, the sum over the four row groups by v_permlane16/32_swap -- no LDS round
  // trip on the chain); the owners of the other tiles (b, j) subtract their parts from t_j in LDS meanwhile.
  for (int b = NTR - 1; b >= 1; --b) {
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      if (ti[q] != b || tj[q] >= b) continue;   // tiles (b, j), j < b: t_j -= L_bj^T x_b
      double xb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[r] = xs[16 * b + q4 + 4 * r];
      if (tj[q] == b - 1) {                     // (uniform) the chain
        const double *Lb = Li + (b - 1) * TS;
        double lk[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) lk[kk] = Lb[kk * 17 + l15];
        const double tb = tv[16 * (b - 1) + l15];
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) part += acc[q][r] * xb[r];
        const double t = tb - rowgroup_sum(part);   // t_{b-1}[l15], in every row group
        double xa = 0.0;
        dpp_dot16<0>(xa, t, lk);
        if (q4 == 0) xs[16 * (b - 1) + l15] = xa;
      } else {
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) part += acc[q][r] * xb[r];
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (q4 == 0) tv[16 * tj[q] + l15] -= part;
      }
    }
    CTV_BAR();
  }
  double *x = d.delta + m.u0;
  for (int i = utid; i < P; i += NTU) x[i] = xs[i];
  if (utid == 0) lm.chol_fail = s_fail;
#undef CTV_BAR
}

// (Fusing this kernel into k_cholesky_tiles -- same workgroup, the pose step straight from LDS -- was built and measured: no gain for
//  one window (3.09 vs 3.05 ms per solve) and slower for 2048 (15.4 vs 13.6 ms for the two phases): the fused kernel spills, and
//  the landmark back-substitution wants more workgroups per CU than the tile kernel's registers allow.  Kept apart.)
// delta_l = dinv_l (-g_l - W_l . delta_p), one wave per landmark (coalesced over the row of W);
// model_cost_change = 1/2 delta^T (D^2 delta - g)  (equals Ceres' -(J y)^T (r + J y / 2) when
// (H + D^2) delta = -g);  then ComputeTrustRegionStep validity / HandleInvalidStep.
// Then, in the same workgroup (one per window): the candidate x (+) alpha delta of this pass (Plus: q <- q exp(d),
// ceres_local_param.h:137-145; additive elsewhere; the line delay projected on its box, trajectory_estimator.cpp:316-317), |step|^2 and
// |x|^2 of the reduced program (fixed-order block reductions, no atomics) and the knot-pair constants of the candidate for the
// linearisation that follows.  Windows inside the line search skip the solve part: their step is the same, only alpha changed.
template <int NWV> __global__ __launch_bounds__(64 * NWV) void k_step_finish(Dev d) {
  constexpr int NT = 64 * NWV;
  const int w = blockIdx.x;
  // the pass's "windows that start another pass" counter (k_pass_end adds to it, several launches later): cleared here instead of by a
  // memset node of its own
  if (w == 0 && threadIdx.x == 0) *d.n_active = 0;
  Lm &lm = d.lm[w];
  if (lm.status) return;
  const WinMeta &m = d.wins[w];
  const int P = m.P, L = m.L, N = m.N, u0 = m.u0, lm0 = m.lm0, ldw = m.ldw;
  extern __shared__ __attribute__((aligned(16))) double xs[];   // [P] pose step
  __shared__ double red[NWV], red_gd[NWV], red_dm[NWV];
  __shared__ int bad, s_go;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (lm.ls_active) {
    if (tid == 0) s_go = lm.step_valid;
  } else {
  double *x = d.delta + u0;
  const double *g = d.gS[d.lm[w].cur] + u0, *dd = d.dd + u0;
  const double *Wp = d.WS[d.lm[w].cur] + m.W0;
  for (int i = tid; i < P; i += NT) xs[i] = x[i];
  if (tid == 0) bad = 0;
  __syncthreads();
  // delta_rho = -(g_rho + W_r . delta_p) / (Hll + D) of the landmark of row r: a wave takes 8 ROWS of W per pass.  The rows are sorted by knot
  // span (host_pack.hpp: plan_sparsity), so the 8 rows of a pass share most of their columns: only the union of their spans' knot columns and the
  // line-delay column are read (config 2: ~100 of 145 columns; K = 64: ~110 of 385) -- the rest of a row is zero and never touched.
  const int32_t *klo = d.lm_klo + lm0, *khi = d.lm_khi + lm0, *lat = d.lm_at + lm0;
  for (int l0 = 8 * wave; l0 < L; l0 += 8 * NWV) {
    double acc8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc8[u] = 0.0;
    // the row this lane will finish (see the reduction below): its landmark's g, 1/(Hll + D) and active flag travel with the W loads
    const int lrow = min(l0 + (lane >> 3), L - 1);
    const int lmk = lat[lrow];
    const double g_l = g[P + lmk], dinv_l = d.dinv[lm0 + lrow];
    const bool act_l = d.active[u0 + P + lmk] != 0;
    int kmin = klo[lrow], kmax = khi[lrow];      // (a row without observations: klo = K, khi = -1 -- it widens nothing)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) { kmin = min(kmin, __shfl_xor(kmin, off)); kmax = max(kmax, __shfl_xor(kmax, off)); }
    const int c_lo = __builtin_amdgcn_readfirstlane(6 * kmin), nkc = __builtin_amdgcn_readfirstlane(max(6 * (kmax + 1) - 6 * kmin, 0));
    const int NCB = nkc + 1;                     // compact columns: the knot columns of the union, then the line delay
    for (int i0 = 0; i0 < NCB; i0 += 128) {
      double wv[8][2];
      int col[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) { const int cc = min(i0 + lane + 64 * k, NCB - 1); col[k] = cc < nkc ? c_lo + cc : P - 1; }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          // clamped, unconditional loads (a predicated load compiles to branch + load + s_waitcnt: one round trip EACH);
          // out-of-range columns are masked through xi below, out-of-range rows are never written
          const int l = min(l0 + u, L - 1);
          wv[u][k] = Wp[(long long)l * ldw + col[k]];
        }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const double xi = (i0 + lane + 64 * k < NCB) ? xs[col[k]] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc8[u] += (double)wv[u][k] * xi;
      }
    }
    // 8 row sums over 64 lanes with 10 shuffles: each butterfly step halves the rows a lane carries (bit 5 of the lane
    // picks rows 0-3 / 4-7, bit 4 the pair, bit 3 the row), then three plain steps; row u = lane >> 3 ends up in lane 8u
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    double v4[4], v2[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) v4[q] = (b5 ? acc8[4 + q] : acc8[q]) + __shfl_xor(b5 ? acc8[q] : acc8[4 + q], 32);
#pragma unroll
    for (int q = 0; q < 2; ++q) v2[q] = (b4 ? v4[2 + q] : v4[q]) + __shfl_xor(b4 ? v4[q] : v4[2 + q], 16);
    double v1 = (b3 ? v2[1] : v2[0]) + __shfl_xor(b3 ? v2[0] : v2[1], 8);
    v1 += __shfl_xor(v1, 4);
    v1 += __shfl_xor(v1, 2);
    v1 += __shfl_xor(v1, 1);
    if ((lane & 7) == 0 && l0 + (lane >> 3) < L) x[P + lmk] = act_l ? (-g_l - v1) * dinv_l : 0.0;
  }
  __syncthreads();
  double mc = 0.0, gd = 0.0, dm = 0.0;   // model change; g . delta and |delta|_inf for the projected line search
  for (int j = tid; j < N; j += NT) {
    const double dj = x[j];
    if (!isfinite(dj)) bad = 1;
    if (d.active[u0 + j]) { mc += 0.5 * dj * (dd[j] * dj - g[j]); gd += g[j] * dj; dm = fmax(dm, fabs(dj)); }
  }
  for (int off = 32; off > 0; off >>= 1) { mc += __shfl_down(mc, off); gd += __shfl_down(gd, off); dm = fmax(dm, __shfl_down(dm, off)); }
  if (lane == 0) { red[wave] = mc; red_gd[wave] = gd; red_dm[wave] = dm; }
  __syncthreads();
  if (tid == 0) {
    double mc_t = 0.0, gd_t = 0.0, dm_t = 0.0;
    for (int q = 0; q < NWV; ++q) { mc_t += red[q]; gd_t += red_gd[q]; dm_t = fmax(dm_t, red_dm[q]); }   // fixed order
    lm.model_change = mc_t;
    lm.ls_gd0 = gd_t;
    lm.ls_dmax = dm_t;
    const bool valid = !lm.chol_fail && !bad && (mc_t > 0.0);
    if (valid) { lm.step_valid = 1; lm.invalid = 0; }
    else {
      lm.step_valid = 0;
      if (++lm.invalid >= d.prm.max_invalid) lm.status = 1 + 5;
      else { lm.mu /= lm.nu; lm.nu *= 2.0; lm.last_ok = 0; lm.nunsucc += 1; }
    }
    s_go = valid ? 1 : 0;
  }
  }   // (solve part)
  __syncthreads();
  if (!s_go) return;
  // ---- candidate = Plus(x, alpha delta)
  {
    const double al = lm.alpha;   // 1, or the trial step size of the projected line search
    const double *dl = d.delta + u0;
    const uint8_t *act = d.active + u0;
    double step2 = 0.0, x2 = 0.0;
    const int nst = m.K + m.F + L + 1;
    for (int t = tid; t < nst; t += NT) {
      if (t < m.K) {
        const int gk = m.knot0 + t;
        const bool ar = act[6 * t] != 0, ap = act[6 * t + 3] != 0;
        const Q4 q0 = qmk(d.quat[4 * gk], d.quat[4 * gk + 1], d.quat[4 * gk + 2], d.quat[4 * gk + 3]);
        Q4 q1 = q0;
        if (ar) q1 = qmul(q0, so3_exp(mk(al * dl[6 * t], al * dl[6 * t + 1], al * dl[6 * t + 2])));
        d.cquat[4 * gk] = q1.x; d.cquat[4 * gk + 1] = q1.y; d.cquat[4 * gk + 2] = q1.z; d.cquat[4 * gk + 3] = q1.w;
        if (ar) {
          step2 += (q1.x - q0.x) * (q1.x - q0.x) + (q1.y - q0.y) * (q1.y - q0.y) + (q1.z - q0.z) * (q1.z - q0.z) + (q1.w - q0.w) * (q1.w - q0.w);
          x2 += q1.x * q1.x + q1.y * q1.y + q1.z * q1.z + q1.w * q1.w;
        }
        for (int c = 0; c < 3; ++c) {
          const double p0 = d.pos[3 * gk + c], p1 = ap ? p0 + al * dl[6 * t + 3 + c] : p0;
          d.cpos[3 * gk + c] = p1;
          if (ap) { step2 += (p1 - p0) * (p1 - p0); x2 += p1 * p1; }
        }
      } else if (t < m.K + m.F) {
        const int f = t - m.K, gf = m.bias0 + f, u = 6 * m.K + 6 * f;
        for (int c = 0; c < 6; ++c) {
          const bool a = act[u + c] != 0;
          const double b0 = d.bias[6 * gf + c], b1 = a ? b0 + al * dl[u + c] : b0;
          d.cbias[6 * gf + c] = b1;
          if (a) { step2 += (b1 - b0) * (b1 - b0); x2 += b1 * b1; }
        }
      } else if (t < m.K + m.F + L) {
        const int l = t - m.K - m.F;
        const bool a = act[P + l] != 0;
        const double r0 = d.rho[lm0 + l], r1 = a ? r0 + al * dl[P + l] : r0;
        d.crho[lm0 + l] = r1;
        if (a) { step2 += (r1 - r0) * (r1 - r0); x2 += r1 * r1; }
      } else {
        const bool a = act[P - 1] != 0;
        const double l0 = d.ld[w];
        double l1 = a ? l0 + al * dl[P - 1] : l0;
        if (a && !m.fix_ld) l1 = fmin(fmax(l1, m.ld_lo), m.ld_hi);
        d.cld[w] = l1;
        if (a) { step2 += (l1 - l0) * (l1 - l0); x2 += l1 * l1; }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { step2 += __shfl_xor(step2, off); x2 += __shfl_xor(x2, off); }
    __syncthreads();   // (red / red_gd of the solve part have been consumed)
    if (lane == 0) { red[wave] = step2; red_gd[wave] = x2; }
    __syncthreads();   // also: the candidate knots are visible to the whole workgroup
    if (tid == 0) {
      double s2 = 0.0, x2t = 0.0;
      for (int q = 0; q < NWV; ++q) { s2 += red[q]; x2t += red_gd[q]; }   // fixed order
      lm.step2 = s2;
      lm.cand_xnorm2 = x2t;
    }
  }
  // ---- knot-pair constants of the candidate (shared by all residual blocks of the linearisation that follows)
  for (int t = tid; t < m.K - 1; t += NT) {
    const int gk = m.knot0 + t;
    knot_pair_const(d.cquat + 4 * gk, d.cquat + 4 * gk + 4, d.lkd + 3 * gk, d.kjri + 9 * gk);
  }
}

}  // namespace ctv
