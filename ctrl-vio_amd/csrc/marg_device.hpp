// marg_device.hpp -- prior construction entirely on the device (reference
// MarginalizationInfo::marginalize, src/estimator/factor/analytic_diff/marginalization_factor.cpp:189-265): from the
// normal equations of a window that holds the dropped factors (assembled by the linearise kernels), eliminate the
// marginalised unknowns with the eigen pseudo-inverse of Amm (eigenvalues <= eps dropped) and factor the remainder
// A' = V S V^T into J0 = sqrt(S) V^T, r0 = S^-1/2 V^T b'.  One workgroup per window, a whole batch per launch.
//
// Both symmetric eigen-problems (Amm: m x m, A': n x n; m, n <= MARG_MAXD) run as a PARALLEL CYCLIC JACOBI with the
// matrix resident in LDS as a packed lower triangle (n = 150: 91 KB): per step the n/2 disjoint pivot pairs of a
// round-robin tournament are rotated at once -- every 2 x 2 block (pair I, pair J) of the matrix is owned by one thread,
// so no entry is touched twice in a step.  The rotation angles are recorded in HBM scratch; the eigenvectors are then
// rebuilt by replaying them on row slabs of the identity that fit in the same LDS (V itself would not: 180 KB).
#pragma once
#include "device_types.hpp"

namespace ctv {

constexpr int MARG_MAXD = 180;        // largest eigen-problem: packed lower triangle = 130 KB of LDS
constexpr int MARG_MAX_SWEEPS = 24;   // cyclic Jacobi converges quadratically; 8-11 sweeps are typical at n = 150

struct MargMeta {
  int32_t N, P, m, n;                 // unknowns, pose unknowns, marginalised, kept
  int32_t idx0;                       // offset of [im (m) | ik (n)] in the int scratch
  int32_t status;                     // out: 0 ok, 1 = Jacobi did not converge
  int64_t A0, V0, X0, Y0, rot0, b0;   // offsets (doubles) into the scratch: A full N x N | Vm m x m | X, Y m x (n+1) | rotations | b' n
  int64_t J0, r0;                     // offsets (doubles) into the outputs
  int32_t sweeps_m, sweeps_n;         // out: Jacobi sweeps of the two eigen-problems
  double trace[2 * 26];               // out: off / diagonal mass before every sweep (diagnostics)
};

__device__ __forceinline__ int pk_idx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// round-robin tournament over np players (np even), step s in [0, np-1), pair i in [0, np/2): (p < q)
__device__ __forceinline__ void rr_pair(int np, int s, int i, int &p, int &q) {
  const int r = np - 1;
  int a, b;
  if (i == 0) { a = r; b = s; }
  else { a = (s + i) % r; b = (s + r - i) % r; }
  p = min(a, b); q = max(a, b);
}

// Cyclic parallel Jacobi on the packed symmetric matrix Apk (dimension nd) in LDS; (c, s) of every rotation goes to rot
// [sweep][step][pair].  Returns the number of sweeps done (eigenvalues are left on the diagonal), or -1 if not converged.
__device__ inline int jacobi_packed(double *Apk, int nd, double *rot, double *cs, int *pq, double *red, double *trace) {
  const int tid = threadIdx.x, np = nd + (nd & 1), half = np / 2, steps = np - 1;
  const int nblk = half * (half + 1) / 2;
  double prev_off = 1e300;
  for (int sweep = 0; sweep < MARG_MAX_SWEEPS; ++sweep) {
    // convergence: off-diagonal mass against the diagonal (the oracle's test)
    double offs = 0.0, dia = 0.0;   // summed separately: off = total - diagonal would cancel (16 orders of magnitude apart)
    for (int i = tid; i < nd; i += 256) {
      const double *row = Apk + i * (i + 1) / 2;
      for (int j = 0; j < i; ++j) offs += row[j] * row[j];
      dia += row[i] * row[i];
    }
    red[tid] = offs; red[256 + tid] = dia;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) { red[tid] += red[tid + st]; red[256 + tid] += red[256 + tid + st]; } __syncthreads(); }
    const double off = red[0], d2 = red[256];
    if (tid == 0 && sweep < 26) trace[sweep] = off / d2;
    __syncthreads();
    // converged: the oracle's test (one full sweep beyond ~1e-29 is what resolves the noise-level eigenvalues of a rank-deficient
    // A', which decide what falls under eps); stagnation just above it after many sweeps is accepted as the rounding floor
    // (the floor of the off-diagonal mass is ~ n^2 eps^2 d2: at n = 180 that is 1.6e-27 d2, above a fixed 1e-28)
    const double floor_rel = fmax(1e-28, 4.0 * (double)nd * (double)nd * 4.93e-32);
    if (off <= 1e-60 || off <= 1e-32 * d2 || (sweep >= 12 && off <= floor_rel * d2 && off > 0.25 * prev_off)) return sweep;
    prev_off = off;
    for (int s = 0; s < steps; ++s) {
      if (tid < half) {
        int p, q;
        rr_pair(np, s, tid, p, q);
        double c = 1.0, sn = 0.0;
        if (q < nd) {
          const double apq = Apk[pk_idx(q, p)];
          if (apq != 0.0) {
            const double app = Apk[pk_idx(p, p)], aqq = Apk[pk_idx(q, q)];
            const double theta = (aqq - app) / (2.0 * apq);
            const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            c = 1.0 / sqrt(tt * tt + 1.0); sn = tt * c;
          }
        }
        cs[2 * tid] = c; cs[2 * tid + 1] = sn; pq[2 * tid] = p; pq[2 * tid + 1] = q;
        rot[((size_t)sweep * steps + s) * half * 2 + 2 * tid] = c;
        rot[((size_t)sweep * steps + s) * half * 2 + 2 * tid + 1] = sn;
      }
      __syncthreads();
      for (int e = tid; e < nblk; e += 256) {
        int I = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while ((I + 1) * (I + 2) / 2 <= e) ++I;
        while (I * (I + 1) / 2 > e) --I;
        const int J = e - I * (I + 1) / 2;
        const int p1 = pq[2 * I], q1 = pq[2 * I + 1], p2 = pq[2 * J], q2 = pq[2 * J + 1];
        const double c1 = cs[2 * I], s1 = cs[2 * I + 1], c2 = cs[2 * J], s2 = cs[2 * J + 1];
        if (I == J) {
          if (q1 >= nd) continue;
          const double app = Apk[pk_idx(p1, p1)], aqq = Apk[pk_idx(q1, q1)], apq = Apk[pk_idx(q1, p1)];
          Apk[pk_idx(p1, p1)] = c1 * c1 * app - 2.0 * c1 * s1 * apq + s1 * s1 * aqq;
          Apk[pk_idx(q1, q1)] = s1 * s1 * app + 2.0 * c1 * s1 * apq + c1 * c1 * aqq;
          Apk[pk_idx(q1, p1)] = 0.0;
          continue;
        }
        const bool vq1 = q1 < nd, vq2 = q2 < nd;   // a dummy player (odd nd) has no row / column
        const double a_pp = Apk[pk_idx(p1, p2)], a_pq = vq2 ? Apk[pk_idx(p1, q2)] : 0.0;
        const double a_qp = vq1 ? Apk[pk_idx(q1, p2)] : 0.0, a_qq = (vq1 && vq2) ? Apk[pk_idx(q1, q2)] : 0.0;
        const double t_pp = c2 * a_pp - s2 * a_pq, t_pq = s2 * a_pp + c2 * a_pq;   // columns (pair J)
        const double t_qp = c2 * a_qp - s2 * a_qq, t_qq = s2 * a_qp + c2 * a_qq;
        Apk[pk_idx(p1, p2)] = c1 * t_pp - s1 * t_qp;                                // rows (pair I)
        if (vq2) Apk[pk_idx(p1, q2)] = c1 * t_pq - s1 * t_qq;
        if (vq1) Apk[pk_idx(q1, p2)] = s1 * t_pp + c1 * t_qp;
        if (vq1 && vq2) Apk[pk_idx(q1, q2)] = s1 * t_pq + c1 * t_qq;
      }
      __syncthreads();
    }
  }
  return -1;
}

// Replays the recorded rotations on rows [r0, r0 + nr) of the identity: slab[row][col], nr * nd doubles in LDS.
__device__ inline void jacobi_replay(double *slab, int nd, int r0, int nr, const double *rot, int sweeps, double *cs) {
  const int tid = threadIdx.x, np = nd + (nd & 1), half = np / 2, steps = np - 1;
  for (int e = tid; e < nr * nd; e += 256) slab[e] = (e % nd == r0 + e / nd) ? 1.0 : 0.0;
  double pc = 1.0, ps = 0.0;
  if (tid < half && sweeps > 0) { pc = rot[2 * tid]; ps = rot[2 * tid + 1]; }
  __syncthreads();
  const int total = sweeps * steps;
  for (int it = 0; it < total; ++it) {
    if (tid < half) { cs[2 * tid] = pc; cs[2 * tid + 1] = ps; }
    __syncthreads();
    if (tid < half && it + 1 < total) { pc = rot[(size_t)(it + 1) * half * 2 + 2 * tid]; ps = rot[(size_t)(it + 1) * half * 2 + 2 * tid + 1]; }   // next step's angles in flight
    const int s = it % steps;
    for (int e = tid; e < nr * half; e += 256) {
      const int row = e / half, i = e % half;
      int p, q;
      rr_pair(np, s, i, p, q);
      if (q >= nd) continue;
      const double c = cs[2 * i], sn = cs[2 * i + 1];
      const double vp = slab[row * nd + p], vq = slab[row * nd + q];
      slab[row * nd + p] = c * vp - sn * vq;
      slab[row * nd + q] = sn * vp + c * vq;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_marginalize(Dev d, MargMeta *metas, const int32_t *iscr, double *scr, double *out, double eps) {
  const int w = blockIdx.x, tid = threadIdx.x;
  MargMeta &mm = metas[w];
  const WinMeta &wm = d.wins[w];
  const int N = mm.N, P = mm.P, m = mm.m, n = mm.n;
  if (n <= 0) return;
  extern __shared__ __attribute__((aligned(16))) double sml[];
  constexpr int NPK = MARG_MAXD * (MARG_MAXD + 1) / 2;
  double *Apk = sml;                   // packed matrix / eigenvector slab
  double *ev = Apk + NPK;              // [MARG_MAXD] eigenvalues
  double *cs = ev + MARG_MAXD;         // [MARG_MAXD] (c, s) of the current step
  double *red = cs + MARG_MAXD;        // [512]
  double *racc = red + 512;            // [MARG_MAXD] r0 accumulators
  int *pq = reinterpret_cast<int *>(racc + MARG_MAXD);   // [MARG_MAXD]
  int *rank = pq + MARG_MAXD;          // [MARG_MAXD]
  const int32_t *im = iscr + mm.idx0, *ik = im + m;
  double *A = scr + mm.A0, *Vm = scr + mm.V0, *X = scr + mm.X0, *Y = scr + mm.Y0, *rot = scr + mm.rot0, *bp = scr + mm.b0;
  const int cset = d.lm[blockIdx.x].cur;   // the normal-equation set that holds the linearisation at the current state
  const double *g = d.gS[cset] + wm.u0;
  // ---- dense symmetric A (N x N) from the structured normal equations: [Hpp W^T; W diag(Hll)]
  {
    const double *H = d.HppS[cset] + wm.H0;
    const double *Wp = d.WS[cset] + wm.W0;
    for (long long e = tid; e < (long long)N * N; e += 256) {
      const int i = (int)(e / N), j = (int)(e % N);
      double v;
      if (i < P && j < P) v = H[(long long)max(i, j) * wm.ldh + min(i, j)];
      else if (i >= P && j >= P) v = (i == j) ? d.HllS[cset][wm.lm0 + i - P] : 0.0;
      else v = (double)Wp[(long long)d.lm_pos[wm.lm0 + max(i, j) - P] * wm.ldw + min(i, j)];   // (rows of W: sorted landmark order)
      A[e] = v;
    }
  }
  __syncthreads();
  int status = 0;
  // ---- Amm = Vm diag(em) Vm^T
  if (m > 0) {
    for (int e = tid; e < m * (m + 1) / 2; e += 256) {
      int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      while (i * (i + 1) / 2 > e) --i;
      const int j = e - i * (i + 1) / 2;
      Apk[e] = A[(long long)im[i] * N + im[j]];
    }
    __syncthreads();
    const int sweeps = jacobi_packed(Apk, m, rot, cs, pq, red, mm.trace);
    if (tid == 0) mm.sweeps_m = sweeps;
    if (sweeps < 0) status = 1;
    for (int i = tid; i < m; i += 256) ev[i] = Apk[i * (i + 3) / 2];
    __syncthreads();
    const int SL = max(1, min(m, NPK / m));
    for (int r0 = 0; r0 < m; r0 += SL) {
      const int nr = min(SL, m - r0);
      jacobi_replay(Apk, m, r0, nr, rot, sweeps < 0 ? MARG_MAX_SWEEPS : sweeps, cs);
      for (int e = tid; e < nr * m; e += 256) Vm[(long long)(r0 + e / m) * m + e % m] = Apk[e];
      __syncthreads();
    }
    // Y = diag(1 / em) Vm^T [Amr | bm], X = Vm Y   (pseudo-inverse: eigenvalues <= eps dropped)
    for (int e = tid; e < m * (n + 1); e += 256) {
      const int a = e / (n + 1), c = e % (n + 1);
      double s = 0.0;
      if (ev[a] > eps) {
        for (int i = 0; i < m; ++i) s += Vm[(long long)i * m + a] * (c < n ? A[(long long)im[i] * N + ik[c]] : g[im[i]]);
        s /= ev[a];
      }
      Y[e] = s;
    }
    __threadfence_block();
    __syncthreads();
    for (int e = tid; e < m * (n + 1); e += 256) {
      const int i = e / (n + 1), c = e % (n + 1);
      double s = 0.0;
      for (int a = 0; a < m; ++a) s += Vm[(long long)i * m + a] * Y[(long long)a * (n + 1) + c];
      X[e] = s;
    }
    __threadfence_block();
    __syncthreads();
  }
  // ---- A' = Arr - Arm X (symmetrised, packed into LDS), b' = br - Arm x_b
  for (int e = tid; e < n * (n + 1) / 2; e += 256) {
    int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    while ((r + 1) * (r + 2) / 2 <= e) ++r;
    while (r * (r + 1) / 2 > e) --r;
    const int c = e - r * (r + 1) / 2;
    double s1 = A[(long long)ik[r] * N + ik[c]], s2 = A[(long long)ik[c] * N + ik[r]];
    for (int i = 0; i < m; ++i) {
      s1 -= A[(long long)ik[r] * N + im[i]] * X[(long long)i * (n + 1) + c];
      s2 -= A[(long long)ik[c] * N + im[i]] * X[(long long)i * (n + 1) + r];
    }
    Apk[e] = 0.5 * (s1 + s2);
  }
  for (int r = tid; r < n; r += 256) {
    double s = g[ik[r]];
    for (int i = 0; i < m; ++i) s -= A[(long long)ik[r] * N + im[i]] * X[(long long)i * (n + 1) + n];
    bp[r] = s;
  }
  __threadfence_block();
  __syncthreads();
  // ---- A' = V S V^T
  const int sweeps = jacobi_packed(Apk, n, rot, cs, pq, red, mm.trace + 26);
  if (tid == 0) mm.sweeps_n = sweeps;
  if (sweeps < 0) status = 1;
  for (int i = tid; i < n; i += 256) { ev[i] = Apk[i * (i + 3) / 2]; racc[i] = 0.0; }
  __syncthreads();
  for (int a = tid; a < n; a += 256) {   // ascending order of the eigenvalues (ties by index), like the references' sorted output
    int rk = 0;
    for (int b = 0; b < n; ++b) rk += (ev[b] < ev[a] || (ev[b] == ev[a] && b < a)) ? 1 : 0;
    rank[a] = rk;
  }
  __syncthreads();
  double *J0 = out + mm.J0, *r0v = out + mm.r0;
  const int nsw = sweeps < 0 ? MARG_MAX_SWEEPS : sweeps;
  const int SL = max(1, min(n, NPK / n));
  for (int rb = 0; rb < n; rb += SL) {
    const int nr = min(SL, n - rb);
    jacobi_replay(Apk, n, rb, nr, rot, nsw, cs);
    // J0[rank a][i] = sqrt(S_a) V[i][a];  r0[rank a] += V[i][a] b'_i
    for (int e = tid; e < nr * n; e += 256) {
      const int row = e / n, a = e % n;
      const double S = ev[a] > eps ? ev[a] : 0.0;
      J0[(long long)rank[a] * n + rb + row] = sqrt(S) * Apk[e];
    }
    for (int a = tid; a < n; a += 256) {
      double s = 0.0;
      for (int row = 0; row < nr; ++row) s += Apk[row * n + a] * bp[rb + row];
      racc[a] += s;
    }
    __syncthreads();
  }
  for (int a = tid; a < n; a += 256) {
    const double S = ev[a] > eps ? ev[a] : 0.0;
    r0v[rank[a]] = S > 0.0 ? racc[a] / sqrt(S) : 0.0;
  }
  if (tid == 0) mm.status = status;
}

}  // namespace ctv
