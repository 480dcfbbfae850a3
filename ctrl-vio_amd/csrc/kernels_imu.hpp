// kernels_imu.hpp -- IMU linearisation: k_imu_linearize_f64 (the product path's walk over groups, fast body), the general body / k_imu_linearize_rest,
// assemble_imu_window (the bias rows of the group tiles: run by k_misc).
// Part of kernels.hpp (included from there, in order; not a stand-alone header).
#pragma once

namespace ctv {

// ------------------------------------------------------------------------------------------------ IMU
template <class T, int N> struct alignas(N * sizeof(T)) VecN { T v[N]; };

typedef double f64x4 __attribute__((ext_vector_type(4)));

// The zeroing of the normal equations' accumulated parts, done by the IMU groups of the window instead of a pass of its own (k_zero_normal
// is HBM-bound, ~100 us per 2048 windows; the stores cost this compute-bound kernel nothing): group gi of the window clears its
// share of Hpp -- only the bias rows and the line-delay row when the single visual-assembly part overwrites the knot x knot block with plain
// stores (zero_mode 1), everything otherwise (zero_mode 2) -- and the first group the gradient and the max-norm cell.  Every accumulating
// kernel (k_assemble_vis*, k_assemble_imu, k_misc) is launched after the linearisation kernels.  zero_mode 0: k_zero_normal did it.
__device__ __forceinline__ void imu_zero_share(const Dev &d, int mode, const ImuGroup &grp, int gidx, int zero_mode) {
  if (!zero_mode) return;
  const int w = grp.win, lane = threadIdx.x;
  const WinMeta &m = d.wins[w];
  const int tg = lin_target(d.lm[w], mode);
  double *Hpp = d.HppS[tg] + m.H0, *g = d.gS[tg] + m.u0;
  const int nH = m.P * m.ldh, first = (m.vis_lds && zero_mode == 1) ? 6 * m.K * m.ldh : 0;   // (ldh is a multiple of 16: both even)
  const int gi = gidx - m.grp0, per = (((nH - first) / 2 + m.ngrp - 1) / m.ngrp) * 2;
  const int lo = first + gi * per, hi = min(lo + per, nH);
  for (int i = lo + 2 * lane; i < hi; i += 128) *reinterpret_cast<double2 *>(Hpp + i) = double2{0.0, 0.0};
  if (gi == 0) {
    for (int i = lane; i < m.P; i += 64) g[i] = 0.0;
    if (lane == 0) { if (mode == LIN_SPEC) d.lm[w].cand_gmax_bits = 0ull; else d.lm[w].gmax_bits = 0ull; }
  }
}

// All-fp64 product path: one wave per IMU group, 64 samples per pass (one per lane), A^T A on the fp64 matrix cores.
// The 6 x 30 Jacobian of a sample stays in REGISTERS in factored form (ImuJac); its six rows are streamed through LDS one
// row index at a time -- phase a: row a of all 64 samples ([64][33] doubles = 16.9 KB, so 8 waves fit a CU and every lane
// evaluates a sample), then 16 K-steps of v_mfma_f64_16x16x4_f64 per output tile.  Accelerometer rows feed the three lower
// 16 x 16 tiles of the 32-column space, gyro rows (non-zero in rotation, gyro-bias and residual columns only) one 16 x 16
// tile on compacted columns.  MFMA operand layout (measured, tools/mfma_f64_layout.hip): A lane l = X[k = l/16][i = l%16],
// B lane l = Y[k = l/16][j = l%16], D register r of lane l = D[(l/16) + 4r][l%16].
__device__ __forceinline__ void imu_linearize_f64_body(const Dev &d, int mode, double *A /* LDS [64][33] */, int gidx, int zero_mode) {
  const ImuGroup grp = d.groups[gidx];
  const int w = grp.win;
  if (!lin_run(d.lm[w], mode)) return;
  const bool jac = !lin_cost_only(d.lm[w], mode, d.prm);   // (uniform) the last allowed iteration only costs its candidate
  if (jac) imu_zero_share(d, mode, grp, gidx, zero_mode);
  const WinMeta &m = d.wins[w];
  const int lane = threadIdx.x, q4 = lane >> 4, l15 = lane & 15;
  const bool at_cand = mode == LIN_SPEC;
  double csum = 0.0;
  const double *s_quat = at_cand ? d.cquat : d.quat, *s_pos = at_cand ? d.cpos : d.pos, *s_bias = at_cand ? d.cbias : d.bias;
  Knots4 k;
  LocalFrame lf;
  lf.init(s_quat, s_pos, m.knot0 + grp.s);
  lf.load(s_quat, s_pos, m.knot0 + grp.s, k);
  const M3 RrefT = lf.RrefT();
  SegConstLazy sc;   // Jr^-1 of the three knot pairs: fetched from the table where it is used
  seg_const_lazy(d.lkd + 3 * (m.knot0 + grp.s), d.kjri + 9 * (m.knot0 + grp.s), sc);
  double bias[6], wgt[6];
  const double *bp = s_bias + 6 * (m.bias0 + grp.bias);
#pragma unroll
  for (int i = 0; i < 6; ++i) { bias[i] = bp[i]; wgt[i] = m.imu_w[i]; }
  const V3 grav = lf.rotate(m.gravity);
  const double idt = m.inv_dt;
  f64x4 acc00 = {0.0, 0.0, 0.0, 0.0}, acc10 = {0.0, 0.0, 0.0, 0.0}, acc11 = {0.0, 0.0, 0.0, 0.0}, gacc = {0.0, 0.0, 0.0, 0.0};
  const size_t Mt = (size_t)d.Mtot;
  if (!jac) {   // residuals only (a separate, small code path: the full one below keeps its compile-time `want_jac = true`)
    for (int c0 = 0; c0 < grp.count; c0 += 64) {
      const bool live = c0 + lane < grp.count;
      const int idx = m.imu0 + grp.start + min(c0 + lane, grp.count - 1);
      double gy[3], ac[3], r[6], wl[6];
#pragma unroll
      for (int i = 0; i < 3; ++i) { gy[i] = d.imu_meas[(size_t)i * Mt + idx]; ac[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
#pragma unroll
      for (int i = 0; i < 6; ++i) wl[i] = live ? wgt[i] : 0.0;
      ImuJac J;
      imu_eval_core(k, sc, d.imu_u[idx], idt, grav, bias, gy, ac, wl, RrefT, r, false, J);
#pragma unroll
      for (int i = 0; i < 6; ++i) csum += 0.5 * r[i] * r[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off);
    if (lane == 0) d.imu_cost[gidx] = csum;
    return;
  }
  for (int c0 = 0; c0 < grp.count; c0 += 64) {
    const int nval = min(64, grp.count - c0);
    const bool live = lane < nval;
    const int idx = m.imu0 + grp.start + min(c0 + lane, grp.count - 1);   // clamped: every lane evaluates (uniform control flow around the MFMAs)
    double gy[3], ac[3], r[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) { gy[i] = d.imu_meas[(size_t)i * Mt + idx]; ac[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
    // lanes past the end of the group evaluate a clamped sample with ZERO weights: every row of w .* [J | r] is then exactly
    // zero (one select per weight instead of one per stored entry: 288 v_cndmask per pass)
    double wl[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) wl[i] = live ? wgt[i] : 0.0;
    const int kmax = (nval + 3) & ~3;
    ImuJac J;
    imu_eval_core(k, sc, d.imu_u[idx], idt, grav, bias, gy, ac, wl, RrefT, r, true, J);
#pragma unroll
    for (int i = 0; i < 6; ++i) csum += 0.5 * r[i] * r[i];   // (dead lanes: zero weights, zero residual)
    // ---- accelerometer rows: 32 columns, tiles (0,0), (1,0), (1,1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {   // unrolled: the row index must be static (a dynamic index would push ImuJac to scratch)
      double row[32];
      imu_row_accel(J, wl, r, a, row);
      __builtin_amdgcn_wave_barrier();   // the previous phase's operand reads are complete (consumed by its MFMAs)
#pragma unroll
      for (int c = 0; c < 32; ++c) A[lane * 33 + c] = row[c];
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the rows are in LDS
      __builtin_amdgcn_wave_barrier();
      for (int k0 = 0; k0 < kmax; k0 += 4) {
        const double lo = A[(k0 + q4) * 33 + l15], hi = A[(k0 + q4) * 33 + 16 + l15];
        acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(lo, lo, acc00, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi, lo, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi, hi, acc11, 0, 0, 0);
      }
    }
    // ---- gyro rows: 16 compacted columns, one tile
#pragma unroll
    for (int a = 0; a < 3; ++a) {   // unrolled: the row index must be static (a dynamic index would push ImuJac to scratch)
      double row[16];
      imu_row_gyro(J, wl, r, a, row);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < 16; ++c) A[lane * 17 + c] = row[c];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      for (int k0 = 0; k0 < kmax; k0 += 4) {
        const double v = A[(k0 + q4) * 17 + l15];
        gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, gacc, 0, 0, 0);
      }
    }
  }
  // ---- the group's share of the cost: fixed-order sum over the lanes (butterfly), one store
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off);
  if (lane == 0) d.imu_cost[gidx] = csum;
  // ---- combine in LDS into the full symmetric 32 x 32 tile, then one coalesced store
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = q4 + 4 * r, col = l15;
    A[row * 32 + col] = acc00[r];
    A[(16 + row) * 32 + 16 + col] = acc11[r];
    A[(16 + row) * 32 + col] = acc10[r];
    A[col * 32 + 16 + row] = acc10[r];     // mirror of the off-diagonal tile
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  {
    const int tc = l15 < 12 ? l15 : (l15 < 15 ? l15 + 12 : 30);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int grow = q4 + 4 * r;
      const int tr = grow < 12 ? grow : (grow < 15 ? grow + 12 : 30);
      A[tr * 32 + tc] += gacc[r];
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  double *tile = d.imu_tiles + (size_t)gidx * 1024;
#pragma unroll
  for (int i = 0; i < 16; ++i) tile[i * 64 + lane] = A[i * 64 + lane];
}

// The MFMA chains of one row phase, two K-steps per trip: the operands of step k + 1 are requested before the MFMAs of step k are issued
// (clock stamps of one group, 3 gyro + 3 accelerometer row phases of a full pass: 6980 + 11372 cycles with the read of step k issued right
// before its MFMA, 6364 + 10384 like this; three steps ahead -- four steps per trip, K rounded to 16 rows -- 6332 + 10828 and the partial
// passes lose to the rounding: not kept; a second gyro accumulator changes nothing either: the chain is not waiting for its own results).
// K is rounded up to a multiple of 8 rows: the rows past the last sample are zero rows (dead lanes write zeros), the buffer has 8 spare rows
// for the last prefetch.
// (The operand fetches and their waits are inline assembly: left to itself the compiler re-loads the carried operand at the top of the
//  next trip -- one ds_read2, one wait, two MFMAs, the very serialisation this removes; volatile loads become flat loads with a full wait
//  each.  The waits carry the operand as an in/out so that the MFMA that consumes it stays behind them; a final lgkmcnt(0) leaves nothing
//  in flight that the compiler's own wait counting does not know about.)
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void imu_chain_gyro(const double *A, int q4, int l15, int kmax, f64x4 &gacc) {
  unsigned addr = (unsigned)(size_t)(A + q4 * 17 + l15);
  double v0, v1;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v0) : "v"(addr));
  for (int k0 = 0; k0 < kmax; k0 += 8) {
    asm volatile("ds_read_b64 %0, %1 offset:544" : "=v"(v1) : "v"(addr));          // step k0 + 4 (4 rows of 17 doubles on)
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(v0));
    gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, gacc, 0, 0, 0);
    addr += 8 * 17 * 8;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v0) : "v"(addr));                     // step k0 + 8
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(v1));
    gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v1, gacc, 0, 0, 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0));
}
__device__ __forceinline__ void imu_chain_accel(const double *A, int q4, int l15, int kmax, f64x4 &acc00, f64x4 &acc10) {
  unsigned addr = (unsigned)(size_t)(A + q4 * 33 + l15);
  f64x2 o0, o1;   // (lo, hi) = columns l15 and 16 + l15 of four rows
  asm volatile("ds_read2_b64 %0, %1 offset1:16" : "=v"(o0) : "v"(addr));
  for (int k0 = 0; k0 < kmax; k0 += 8) {
    asm volatile("ds_read2_b64 %0, %1 offset0:132 offset1:148" : "=v"(o1) : "v"(addr));   // step k0 + 4 (4 rows of 33 doubles on)
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(o0));
    acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(o0[0], o0[0], acc00, 0, 0, 0);
    acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(o0[1], o0[0], acc10, 0, 0, 0);
    addr += 8 * 33 * 8;
    asm volatile("ds_read2_b64 %0, %1 offset1:16" : "=v"(o0) : "v"(addr));                   // step k0 + 8
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(o1));
    acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(o1[0], o1[0], acc00, 0, 0, 0);
    acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(o1[1], o1[0], acc10, 0, 0, 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o0));
}

// The fast body's groups: every knot-pair log below 0.5 rad, isotropic accelerometer weights.  Asked in two places (the fast body about its
// own group, k_imu_linearize_rest about every group of its window) that must agree to the bit: the operations are spelled out (no
// contraction choices left to the compiler).
__device__ __forceinline__ bool imu_fast_pred(const double kd[9], const double *imu_w) {
  double mx = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) mx = fmax(mx, __fma_rn(kd[3 * i + 2], kd[3 * i + 2], __fma_rn(kd[3 * i + 1], kd[3 * i + 1], __dmul_rn(kd[3 * i], kd[3 * i]))));
  return mx < 0.25 && imu_w[3] == imu_w[4] && imu_w[3] == imu_w[5];
}
// ---- The product path's body for the usual group (imu_group_fast: knot-pair logs below 0.5 rad, isotropic accelerometer weights).
// Same wave-per-group scheme and row streaming as the general body above, with
//   * the evaluation in stages (factors.hpp, staged form): values, gyro Jacobians -> three row phases, accelerometer Jacobians -> three row
//     phases, so the two 36-entry Jacobians are never live together; small-angle series, no branch in the loop;
//   * global frame (the local frame of the general body is an fp32 device), Jr^-1 of the three knot pairs and their logs in SGPRs;
//   * the accelerometer rows in two 16-column tiles T0 = [rot 12 | ba 3 | r], T1 = [pos 12]: T0^T T0 and T1^T T0 on the matrix cores,
//     T1^T T1 = w^2 sum_s lamA_k lamA_k' I3 from ten per-lane sums (R(t)^T W^2 R(t) = w^2 I): 2 MFMAs per K-step instead of 3;
//   * the next pass's measurements requested before the current pass is evaluated.
// (fp64 MFMA and fp64 VALU instructions share one datapath on gfx950 -- tools/mfma_valu_overlap.hip: one wave's MFMAs and FMAs add up,
//  two waves on a SIMD do not overlap them either -- so the kernel's time is the SUM of its vector and matrix work: both are cut here.)
// A wave WALKS its groups g0, g0 + stride, ... (k_imu_linearize_f64: 2048 waves for the whole batch) and everything the NEXT group's
// record locates -- pair logs and Jr^-1, first knot's rotation, knot positions, bias, gravity, weights, 1 / dt, the window's LM flags, one
// element per lane -- is requested while the CURRENT group is evaluated, and the record after that is on its way as well; the next group's
// first 64 samples are requested by the current group's last pass.  A group's own prologue (three dependent round trips group -> window ->
// data at one wave per SIMD: ~10 k of a group's 55 k cycles, measured) shrinks to a few dozen v_readlane.
struct ImuPre { double pc, kq; int fl; };
__device__ __forceinline__ void imu_prefetch(const Dev &d, int mode, const ImuGroup &g, int lane, ImuPre &o) {
  const bool at_cand = mode == LIN_SPEC;
  const double *s_quat = at_cand ? d.cquat : d.quat, *s_pos = at_cand ? d.cpos : d.pos, *s_bias = at_cand ? d.cbias : d.bias;
  const WinMeta &m = d.wins[g.win];
  const Lm &lm = d.lm[g.win];
  const double *kd = d.lkd + 3 * g.kabs, *kj = d.kjri + 9 * g.kabs;
  const double *q = s_quat + 4 * g.kabs, *pp = s_pos + 3 * g.kabs, *bp = s_bias + 6 * g.babs;
  // pc: lanes 0..8 the pair logs, 9..35 Jr^-1 (row major per pair)
  o.pc = *(lane < 9 ? kd + lane : kj + (min(lane, 35) - 9));
  // kq: 0..3 q_0 | 4..15 the four knot positions | 21..23 gravity | 24..29 bias | 30..35 weights | 36 1 / dt   (unconditional loads on valid addresses)
  const double *src = lane < 4 ? q + lane : lane < 16 ? pp + (lane - 4) : lane < 21 ? q : lane < 24 ? m.gravity + (lane - 21)
                      : lane < 30 ? bp + (lane - 24) : lane < 36 ? m.imu_w + (lane - 30) : &m.inv_dt;
  o.kq = *src;
  // fl: lanes 0..3 the window's LM flags (not written by any linearisation kernel)
  const int32_t *fp = lane == 0 ? &lm.status : lane == 1 ? &lm.step_valid : lane == 2 ? &lm.iter : &lm.ls_active;
  o.fl = *fp;
}
__device__ __forceinline__ void imu_linearize_f64_fast(const Dev &d, int mode, double *A /* LDS [72][33] + 64 */, int g0, int stride, int zero_mode) {
  const int lane = threadIdx.x, q4 = lane >> 4, l15 = lane & 15;
  const size_t Mt = (size_t)d.Mtot;
  int gidx = g0;
  ImuGroup grp = d.groups[gidx];
  ImuPre cur;
  imu_prefetch(d, mode, grp, lane, cur);                 // (the walk's first group: its round trips are exposed once)
  double gyn[3], acn[3], un;   // the next pass's measurements, in flight while the current pass is evaluated
  {
    const int idx = grp.iabs + min(lane, grp.count - 1);
#pragma unroll
    for (int i = 0; i < 3; ++i) { gyn[i] = d.imu_meas[(size_t)i * Mt + idx]; acn[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
    un = d.imu_u[idx];
  }
  bool has_next = gidx + stride < d.Gtot;
  ImuGroup grpn = d.groups[has_next ? gidx + stride : gidx];
  for (;;) {
  // ---- the group after the next one's record and the next one's constants: on their way during this group
  const bool has_next2 = has_next && gidx + 2 * stride < d.Gtot;
  const ImuGroup grpn2 = d.groups[has_next2 ? gidx + 2 * stride : gidx];
  ImuPre nxt;
  imu_prefetch(d, mode, grpn, lane, nxt);
  bool nmeas = false;          // the next group's first pass has been requested (by this group's last pass)
  do {
  const int w = grp.win;
  const int base = grp.iabs;
  // the group's constants: knot-pair logs and Jr^-1 (used a dozen times per pass) in scalar registers, the rest (used once or twice per
  // pass) in LDS behind the row buffer -- [0..11] knot positions relative to knot 0, [12..20] R_0^T, [21..23] gravity, [24..29] bias,
  // [30..35] weights.  (All of them in scalar registers overflow the SGPR file: 250 v_readlane per pass to fetch them back.)
  SegConstS sc;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    sc.d[i] = mk(readlane_d(cur.pc, 3 * i), readlane_d(cur.pc, 3 * i + 1), readlane_d(cur.pc, 3 * i + 2));
#pragma unroll
    for (int e = 0; e < 9; ++e) sc.JrI[i].m[e] = readlane_d(cur.pc, 9 + 9 * i + e);
  }
  // ---- the window: LM state
  const int f_status = __builtin_amdgcn_readlane(cur.fl, 0), f_valid = __builtin_amdgcn_readlane(cur.fl, 1), f_iter = __builtin_amdgcn_readlane(cur.fl, 2),
            f_ls = __builtin_amdgcn_readlane(cur.fl, 3);
  if (!(f_status == 0 && (mode != LIN_SPEC || f_valid != 0))) break;                                   // lin_run
  const bool jac = !(mode == COST_AT_X || (mode == LIN_SPEC && f_iter >= d.prm.max_iters && f_ls == 0));   // lin_cost_only: (uniform) the last allowed iteration only costs its candidate
  // is this group the fast body's?  (imu_group_fast, decided HERE from the pair logs and the weights already in registers)
  {
    const double kd9[9] = {sc.d[0].x, sc.d[0].y, sc.d[0].z, sc.d[1].x, sc.d[1].y, sc.d[1].z, sc.d[2].x, sc.d[2].y, sc.d[2].z};
    const double w6[6] = {0.0, 0.0, 0.0, readlane_d(cur.kq, 33), readlane_d(cur.kq, 34), readlane_d(cur.kq, 35)};
    if (!imu_fast_pred(kd9, w6)) break;   // (uniform) left to k_imu_linearize_rest
  }
  long long *dbg = (d.dbg && gidx == 5000 && jac) ? d.dbg + 64 : nullptr;   // CTVIO_DEBUG_STAMPS: clock64 of lane 0 at the phase boundaries
  int dbi = 0;
#define CTV_ISTAMP(x) do { if (dbg && lane == 0 && dbi < 16) dbg[dbi++] = clock64() + (long long)((x) * 0.0); } while (0)
  CTV_ISTAMP(0.0);
  double *gc = A + 72 * 33;   // (8 spare rows behind the 64: the chains' last prefetch)
  {
    double gcv = cur.kq;      // lanes 21..35: gravity, bias, weights as requested
    const M3 R0 = q2R(qmk(readlane_d(cur.kq, 0), readlane_d(cur.kq, 1), readlane_d(cur.kq, 2), readlane_d(cur.kq, 3)));
    const double pk = __shfl(cur.kq, 4 + min(lane, 11)), p0 = __shfl(cur.kq, 4 + min(lane, 11) % 3);
    if (lane < 12) gcv = pk - p0;
    else if (lane < 21) {   // R_0^T, row major (a select chain: no dynamically indexed register array)
      const int e = lane - 12, src = 3 * (e % 3) + e / 3;
#pragma unroll
      for (int i = 0; i < 9; ++i) gcv = src == i ? R0.m[i] : gcv;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) gc[lane] = gcv;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  const double idt = readlane_d(cur.kq, 36);
  double csum = 0.0;
  if (!jac) {   // residuals only
    for (int c0 = 0; c0 < grp.count; c0 += 64) {
      const bool live = c0 + lane < grp.count;
      const int idx = base + min(c0 + lane, grp.count - 1);
      double gy[3], ac[3], r[6], wl[6];
#pragma unroll
      for (int i = 0; i < 3; ++i) { gy[i] = d.imu_meas[(size_t)i * Mt + idx]; ac[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
#pragma unroll
      for (int i = 0; i < 6; ++i) wl[i] = live ? gc[30 + i] : 0.0;
      ImuMid3 md;
      imu_eval_values3(gc, sc, d.imu_u[idx], idt, gy, ac, wl, r, md);
#pragma unroll
      for (int i = 0; i < 6; ++i) csum += 0.5 * r[i] * r[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off);
    if (lane == 0) d.imu_cost[gidx] = csum;
    break;
  }
  f64x4 acc00 = {0.0, 0.0, 0.0, 0.0}, acc10 = {0.0, 0.0, 0.0, 0.0}, gacc = {0.0, 0.0, 0.0, 0.0};
  double spp[10] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int c0 = 0; c0 < grp.count; c0 += 64) {
    CTV_ISTAMP(csum);
    const int nval = min(64, grp.count - c0);
    const bool live = lane < nval;
    double gy[3], ac[3], r[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) { gy[i] = gyn[i]; ac[i] = acn[i]; }
    const double u = un;
    {
      // the next pass's samples -- after the group's last pass the NEXT GROUP's first ones (without one, a valid sample that is dropped)
      const bool last = c0 + 64 >= grp.count;
      const int idx = (last && has_next) ? grpn.iabs + min(lane, grpn.count - 1) : base + min(c0 + 64 + lane, grp.count - 1);
      nmeas = last;
#pragma unroll
      for (int i = 0; i < 3; ++i) { gyn[i] = d.imu_meas[(size_t)i * Mt + idx]; acn[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
      un = d.imu_u[idx];
    }
    // lanes past the end of the group evaluate a clamped sample with ZERO weights: every row of w .* [J | r] is then exactly zero
    double wl[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) wl[i] = live ? gc[30 + i] : 0.0;
    const int kmax = (nval + 3) & ~3;
    ImuMid3 md;
    imu_eval_values3(gc, sc, u, idt, gy, ac, wl, r, md);
#pragma unroll
    for (int i = 0; i < 6; ++i) csum += 0.5 * r[i] * r[i];   // (dead lanes: zero weights, zero residual)
    CTV_ISTAMP(csum);
    {
      M3 Jw[4];
      imu_jac_gyro3(md, sc, Jw);
      CTV_ISTAMP(Jw[3].m[8]);
#pragma unroll
      for (int a = 0; a < 3; ++a) {   // unrolled: the row index must be static
        double row[16];
        imu_row_gyro2(Jw, wl, r, a, row);
        __builtin_amdgcn_wave_barrier();   // the previous phase's operand reads are complete (consumed by its MFMAs)
#pragma unroll
        for (int c = 0; c < 16; ++c) A[lane * 17 + c] = row[c];
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the rows are in LDS
        __builtin_amdgcn_wave_barrier();
        imu_chain_gyro(A, q4, l15, kmax, gacc);
      }
    }
    CTV_ISTAMP(gacc[0]);
    {
      M3 Ja[4], Rinv_g;
      imu_jac_accel3(md, sc, gc, Ja, Rinv_g);
      CTV_ISTAMP(Ja[3].m[8] + Rinv_g.m[8]);
      {
        double la[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) la[kk] = wl[3] * md.lamA[kk];
        int e = 0;
#pragma unroll
        for (int ka = 0; ka < 4; ++ka)
#pragma unroll
          for (int kb = 0; kb <= ka; ++kb) { spp[e] += la[ka] * la[kb]; ++e; }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        double row[28];
        imu_row_accel3(Ja, md.lamA, Rinv_g, wl, r, a, row);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 28; ++c) A[lane * 33 + c] = row[c];   // (columns 28..31 feed accumulator rows nobody reads)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        imu_chain_accel(A, q4, l15, kmax, acc00, acc10);
      }
    }
  }
  CTV_ISTAMP(acc00[0] + acc10[0]);
  // ---- combine in LDS into the full symmetric 32 x 32 tile in the local column order [rot 12 | pos 12 | bg 3 | ba 3 | r | -]
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 16; ++i) A[i * 64 + lane] = 0.0;
  {
    // eleven sums over the 64 lanes in a fixed order -- the ten of the pos x pos block and the group's share of the cost -- through the
    // free half of the buffer: lane (e, part) adds 16 lanes' values, two butterfly steps join the four parts (one LDS round trip for all
    // of them instead of a six-step butterfly per value)
    double *S = A + 1024;
#pragma unroll
    for (int e = 0; e < 10; ++e) S[e * 64 + lane] = spp[e];
    S[10 * 64 + lane] = csum;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int e = min(lane >> 2, 10), part = lane & 3;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += S[e * 64 + part * 16 + i];
    t += __shfl_xor(t, 1);
    t += __shfl_xor(t, 2);
    __builtin_amdgcn_wave_barrier();
    if (lane < 40 && part == 0) S[704 + e] = t;
    if (lane == 40) d.imu_cost[gidx] = t;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  {
    // T0 (accelerometer rows: rot 12 | ba 3 | r) and the gyro tile (rot 12 | bg 3 | r) share the accumulator layout: where neither index
    // is a bias one the two land on the same entry and are added in registers; a bias index sends them to the ba / bg columns
    const int c0 = l15 < 12 ? l15 : (l15 < 15 ? l15 + 15 : 30);   // T0 index -> local column (ba at 27..29)
    const int cg = l15 < 12 ? l15 : (l15 < 15 ? l15 + 12 : 30);   // gyro tile index -> local column (bg at 24..26)
    const bool cb = l15 >= 12 && l15 < 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = q4 + 4 * r;
      const int r0 = t < 12 ? t : (t < 15 ? t + 15 : 30), rg = t < 12 ? t : (t < 15 ? t + 12 : 30);
      const bool shared = !cb && !(t >= 12 && t < 15);
      A[r0 * 32 + c0] = shared ? acc00[r] + gacc[r] : acc00[r];
      if (!shared) A[rg * 32 + cg] = gacc[r];
      if (t < 12) { A[(12 + t) * 32 + c0] = acc10[r]; A[c0 * 32 + 12 + t] = acc10[r]; }
    }
    // lane (ka, kb, b) < 48 places one entry of the pos x pos block
    const int ka = lane / 12, kb = (lane / 3) & 3, b = lane % 3;
    const int hi = max(ka, kb), lo = min(ka, kb);
    if (lane < 48) A[(12 + 3 * ka + b) * 32 + 12 + 3 * kb + b] = A[1024 + 704 + hi * (hi + 1) / 2 + lo];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  double *tile = d.imu_tiles + (size_t)gidx * 1024;
#pragma unroll
  for (int i = 0; i < 8; ++i)   // 16 bytes per lane: 8 stores of 1 KiB (under load a store costs ~100 cycles whatever its width)
    *reinterpret_cast<double2 *>(tile + i * 128 + 2 * lane) = *reinterpret_cast<const double2 *>(A + i * 128 + 2 * lane);
  // (last: the memory counter is in-order, a load issued after these stores would wait for their acknowledgement)
  imu_zero_share(d, mode, grp, gidx, zero_mode);
  CTV_ISTAMP(0.0);
#undef CTV_ISTAMP
  } while (false);
  // ---- on to the wave's next group
  if (!has_next) break;
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();          // (the tile copy-out has read the LDS buffer before the next group writes its constants)
  if (!nmeas) {                             // this group left early: the next one's first pass has not been asked for yet
    const int idx = grpn.iabs + min(lane, grpn.count - 1);
#pragma unroll
    for (int i = 0; i < 3; ++i) { gyn[i] = d.imu_meas[(size_t)i * Mt + idx]; acn[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
    un = d.imu_u[idx];
  }
  gidx += stride;
  grp = grpn; grpn = grpn2; cur = nxt;
  has_next = has_next2;
  }
}

// One wave per SIMD: the evaluation needs ~430 fp64-pair registers; with a 512-register budget the overflow lives in AGPRs.
// (Two waves per SIMD with the overflow spilled to scratch was measured 3x slower: 1690 vs 540 us per 1024 windows.)
// The fast body evaluates the small-angle series only: a group whose knot-pair logs reach 0.5 rad (28.6 degrees between two knots 50 ms
// apart) takes the general body.  It also takes the pos x pos block from R(t)^T W^2 R(t) = w^2 I: isotropic accelerometer weights (the
// reference's: one scalar per sensor) -- any other weighting takes the general body as well.
__device__ __forceinline__ bool imu_group_fast(const Dev &d, int gidx) {
  const ImuGroup grp = d.groups[gidx];
  const double *kd = d.lkd + 3 * grp.kabs;
  const double kd9[9] = {kd[0], kd[1], kd[2], kd[3], kd[4], kd[5], kd[6], kd[7], kd[8]};
  return imu_fast_pred(kd9, d.wins[grp.win].imu_w);
}
// The groups the fast body leaves out are picked up by k_imu_linearize_rest (one wave per WINDOW: its lanes look at the window's groups,
// the wave then takes the flagged ones in turn -- 12 us per launch when there is nothing to do, which is the rule): the two bodies in
// one kernel cost the fast one registers.
// (general_only: every group through the general body -- ctvio_options.use_mfma = 2 / CTVIO_IMU_GENERAL=1, the tests' way into it)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_imu_linearize_f64(Dev d, int mode, int general_only, int zero_mode) {
  extern __shared__ __attribute__((aligned(32))) unsigned char smraw[];
  if (!general_only) imu_linearize_f64_fast(d, mode, reinterpret_cast<double *>(smraw), blockIdx.x, gridDim.x, zero_mode);   // (skips the groups that are not its own)
}
__device__ __forceinline__ void imu_rest_body(const Dev &d, int mode, int general_only, int zero_mode, unsigned char *smraw /* LDS [64][33] doubles */, int w) {
  const WinMeta &m = d.wins[w];
  for (int g0 = 0; g0 < m.ngrp; g0 += 64) {
    const int gl = g0 + (int)threadIdx.x;
    const bool need = gl < m.ngrp && (general_only || !imu_group_fast(d, m.grp0 + gl));
    unsigned long long todo = __ballot(need);
    while (todo) {
      const int b = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      imu_linearize_f64_body(d, mode, reinterpret_cast<double *>(smraw), m.grp0 + g0 + b, zero_mode);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  }
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_imu_linearize_rest(Dev d, int mode, int general_only, int zero_mode) {
  extern __shared__ __attribute__((aligned(32))) unsigned char smraw[];
  imu_rest_body(d, mode, general_only, zero_mode, smraw, blockIdx.x);
}

// Scatter the group tiles into Hpp (lower triangle, fp64) and g.
// One workgroup per WINDOW walking its groups (one per group was 43 k workgroups of 195 useful threads: dispatch-bound).
// (a device function: k_misc runs it in front of the bias chain and the prior -- the two were separate launches of the same shape, one
//  256-thread workgroup per window, each a chain of dependent loads followed by atomics)
__device__ __forceinline__ void assemble_imu_window(const Dev &d, int mode, int w, double *band /* LDS [144 K] or null */) {
  if (!lin_run(d.lm[w], mode) || lin_cost_only(d.lm[w], mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  const int K = m.K, ldh = m.ldh, tg = lin_target(d.lm[w], mode);
  double *Hpp = d.HppS[tg] + m.H0, *g = d.gS[tg] + m.u0;
  if (m.vis_lds) {
    // the knot x knot part is accumulated in LDS by k_assemble_vis; what is left is the bias rows (6 x 24 against the
    // knots, the 6 x 6 lower triangle) and the gradient: 195 entries per group -- one load, one atomic each
    for (int i = threadIdx.x; i < 195 * m.ngrp; i += 256) {
      const int gi = i / 195, t = i - 195 * gi;
      const ImuGroup grp = d.groups[m.grp0 + gi];
      const double *tile = d.imu_tiles + (size_t)(m.grp0 + gi) * 1024;
      int a, b;
      if (t < 144) { a = 24 + t / 24; b = t % 24; }
      else if (t < 165) {
        const int q = t - 144;                     // lower triangle of the bias block, row-major
        const int r = q < 1 ? 0 : q < 3 ? 1 : q < 6 ? 2 : q < 10 ? 3 : q < 15 ? 4 : 5;
        a = 24 + r; b = 24 + q - r * (r + 1) / 2;
      } else { a = t - 165; b = 30; }
      // (the tile is symmetric: the gradient column is read as row 30, next to the bias rows -- 7 consecutive rows of the tile instead of a
      //  cache line of every row)
      const double v = (double)(b == 30 ? tile[30 * 32 + a] : tile[a * 32 + b]);
      const int ga = imu_col(a, grp.s, K, grp.bias);
      if (b == 30) { atomicAdd(&g[ga], v); continue; }
      const int gb = imu_col(b, grp.s, K, grp.bias);
      atomicAdd(&Hpp[(long long)max(ga, gb) * ldh + min(ga, gb)], v);
    }
    return;
  }
  // Windows whose packed Hessian does not fit in LDS (K > 24).  The 24 x 24 knot blocks of consecutive segments overlap in three of their four
  // knots (and two groups of one segment coincide): added one by one with global atomics they cost 300 atomics per group -- config 5 (K = 64):
  // 27 k per window, 0.45 ms per 512 windows.  They are summed first in an LDS BAND [K][4 knots][6][6] (a knot couples with itself and the three
  // before it: 74 KB at K = 64) and every band entry goes out once; the bias rows and the gradient (195 per group) stay as they were.
  if (band) {
    const int nb = 144 * K;
    for (int i = threadIdx.x; i < nb; i += 256) band[i] = 0.0;
    __syncthreads();
    for (int i = threadIdx.x; i < 300 * m.ngrp; i += 256) {
      const int gi = i / 300, t = i - 300 * gi;
      int a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);     // t -> (a >= b), row-major over the lower triangle of 24 x 24
      a += ((a + 1) * (a + 2) / 2 <= t) ? 1 : 0;
      a -= (a * (a + 1) / 2 > t) ? 1 : 0;
      const int b = t - a * (a + 1) / 2;
      const ImuGroup grp = d.groups[m.grp0 + gi];
      const double v = d.imu_tiles[(size_t)(m.grp0 + gi) * 1024 + a * 32 + b];
      int ga = imu_col(a, grp.s, K, grp.bias), gb = imu_col(b, grp.s, K, grp.bias);
      if (ga < gb) { const int tmp = ga; ga = gb; gb = tmp; }            // (local order [rot | pos]: a >= b does not order the unknowns)
      const int k1 = ga / 6, k2 = gb / 6;
      atomicAdd(&band[((k2 * 4 + (k1 - k2)) * 6 + (ga - 6 * k1)) * 6 + (gb - 6 * k2)], v);   // (every unordered local pair once)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 256) {
      const double v = band[i];
      if (v == 0.0) continue;
      const int r2 = i % 6, r1 = (i / 6) % 6, dk = (i / 36) % 4, k2 = i / 144;
      const int ga = 6 * (k2 + dk) + r1, gb = 6 * k2 + r2;
      atomicAdd(&Hpp[(long long)ga * ldh + gb], v);
    }
    for (int i = threadIdx.x; i < 195 * m.ngrp; i += 256) {              // bias rows, bias block, gradient: as in the LDS-resident case above
      const int gi = i / 195, t = i - 195 * gi;
      const ImuGroup grp = d.groups[m.grp0 + gi];
      const double *tile = d.imu_tiles + (size_t)(m.grp0 + gi) * 1024;
      int a, b;
      if (t < 144) { a = 24 + t / 24; b = t % 24; }
      else if (t < 165) {
        const int q = t - 144;
        const int r = q < 1 ? 0 : q < 3 ? 1 : q < 6 ? 2 : q < 10 ? 3 : q < 15 ? 4 : 5;
        a = 24 + r; b = 24 + q - r * (r + 1) / 2;
      } else { a = t - 165; b = 30; }
      const double v = (double)(b == 30 ? tile[30 * 32 + a] : tile[a * 32 + b]);
      const int ga = imu_col(a, grp.s, K, grp.bias);
      if (b == 30) { atomicAdd(&g[ga], v); continue; }
      const int gb = imu_col(b, grp.s, K, grp.bias);
      atomicAdd(&Hpp[(long long)max(ga, gb) * ldh + min(ga, gb)], v);
    }
    return;
  }
  for (int i = threadIdx.x; i < 31 * 30 * m.ngrp; i += 256) {
    const int gi = i / 930, e = i - 930 * gi;
    const ImuGroup grp = d.groups[m.grp0 + gi];
    const double *tile = d.imu_tiles + (size_t)(m.grp0 + gi) * 1024;
    const int b = e / 30, a = e % 30;  // a < 30 : unknown row; b <= 30
    const double v = (double)tile[a * 32 + b];
    const int ga = imu_col(a, grp.s, K, grp.bias);
    if (b == 30) { atomicAdd(&g[ga], v); continue; }
    const int gb = imu_col(b, grp.s, K, grp.bias);
    if (ga >= gb) atomicAdd(&Hpp[(long long)ga * ldh + gb], v);
  }
}

}  // namespace ctv
