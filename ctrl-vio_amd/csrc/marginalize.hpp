// marginalize.hpp -- host side of the prior construction: from the normal equations A x = b of a
// window that holds the dropped factors (assembled ON THE DEVICE by the linearise kernels), eliminate the marginalised
// unknowns and factor the result into the prior (J0, r0) the next window consumes.
// Reference: MarginalizationInfo::marginalize, src/estimator/factor/analytic_diff/marginalization_factor.cpp:189-265
// (Eigen::SelfAdjointEigenSolver twice: pseudo-inverse of Amm with eigenvalues <= eps dropped; A' = V S V^T ->
// J0 = sqrt(S) V^T, r0 = S^-1/2 V^T b').  The dense algebra is O(N^3) on N <= a few hundred unknowns and runs on the host:
// Householder tridiagonalisation + implicit QL (the classic tred2 / tql2 pair), not the oracle's Jacobi iteration.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace ctv {

// Symmetric eigendecomposition: A (n x n row-major, symmetric) -> eigenvalues d ascending, eigenvectors in the COLUMNS of V.
inline void sym_eig_ql(int n, const double *A, std::vector<double> &d, std::vector<double> &V) {
  d.assign(n, 0.0);
  V.assign(A, A + (size_t)n * n);
  if (n == 0) return;
  std::vector<double> e(n, 0.0);
  auto v = [&](int i, int j) -> double & { return V[(size_t)i * n + j]; };
  // ---- Householder reduction to tridiagonal form
  for (int j = 0; j < n; ++j) d[j] = v(n - 1, j);
  for (int i = n - 1; i > 0; --i) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) { d[j] = v(i - 1, j); v(i, j) = 0.0; v(j, i) = 0.0; }
    } else {
      for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
      double f = d[i - 1], g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0.0;
      for (int j = 0; j < i; ++j) {
        f = d[j];
        v(j, i) = f;
        g = e[j] + v(j, j) * f;
        for (int k = j + 1; k <= i - 1; ++k) { g += v(k, j) * d[k]; e[k] += v(k, j) * f; }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; ++k) v(k, j) -= (f * e[k] + g * d[k]);
        d[j] = v(i - 1, j);
        v(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; ++i) {   // accumulate the transformations
    v(n - 1, i) = v(i, i);
    v(i, i) = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; ++k) d[k] = v(k, i + 1) / h;
      for (int j = 0; j <= i; ++j) {
        double g = 0.0;
        for (int k = 0; k <= i; ++k) g += v(k, i + 1) * v(k, j);
        for (int k = 0; k <= i; ++k) v(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; ++k) v(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; ++j) { d[j] = v(n - 1, j); v(n - 1, j) = 0.0; }
  v(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // ---- implicit QL on the tridiagonal matrix
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = std::ldexp(1.0, -52);
  for (int l = 0; l < n; ++l) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) { if (std::fabs(e[m]) <= eps * tst1) break; ++m; }
    if (m > l) {
      int iter = 0;
      do {
        ++iter;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
        const double el1 = e[l + 1];
        for (int i = m - 1; i >= l; --i) {
          c3 = c2; c2 = c; s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; ++k) {
            h = v(k, i + 1);
            v(k, i + 1) = s * v(k, i) + c * h;
            v(k, i) = c * v(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
    }
    d[l] += f;
    e[l] = 0.0;
  }
  for (int i = 0; i < n - 1; ++i) {   // ascending order
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < n; ++j) if (d[j] < p) { k = j; p = d[j]; }
    if (k != i) {
      d[k] = d[i];
      d[i] = p;
      for (int j = 0; j < n; ++j) std::swap(v(j, i), v(j, k));
    }
  }
}

// role[N]: 1 = marginalise, 0 = keep, -1 = not involved.  H is N x N row-major (full symmetric), g has N entries.
// Outputs: kept (ascending unknown indices, n of them), J0 (n x n row-major, row i = sqrt(S_i) v_i^T, S ascending), r0 (n).
inline int marginalize_dense(int N, const double *H, const double *g, const int8_t *role, double eps, std::vector<int32_t> &kept,
                             std::vector<double> &J0, std::vector<double> &r0) {
  std::vector<int> im, ik;
  for (int i = 0; i < N; ++i) { if (role[i] == 1) im.push_back(i); else if (role[i] == 0) ik.push_back(i); }
  const int m = (int)im.size(), n = (int)ik.size();
  kept.assign(ik.begin(), ik.end());
  J0.assign((size_t)n * n, 0.0);
  r0.assign(n, 0.0);
  if (n == 0) return 0;
  // X = Amm^+ [Amr | bm]   (m x (n + 1))
  std::vector<double> X((size_t)std::max(m, 1) * (n + 1), 0.0);
  if (m > 0) {
    std::vector<double> Amm((size_t)m * m), em, Vm;
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) Amm[(size_t)i * m + j] = 0.5 * (H[(size_t)im[i] * N + im[j]] + H[(size_t)im[j] * N + im[i]]);
    sym_eig_ql(m, Amm.data(), em, Vm);
    std::vector<double> Y((size_t)m * (n + 1));
    for (int a = 0; a < m; ++a)
      for (int c = 0; c <= n; ++c) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += Vm[(size_t)i * m + a] * (c < n ? H[(size_t)im[i] * N + ik[c]] : g[im[i]]);
        Y[(size_t)a * (n + 1) + c] = em[a] > eps ? s / em[a] : 0.0;
      }
    for (int i = 0; i < m; ++i)
      for (int c = 0; c <= n; ++c) {
        double s = 0.0;
        for (int a = 0; a < m; ++a) s += Vm[(size_t)i * m + a] * Y[(size_t)a * (n + 1) + c];
        X[(size_t)i * (n + 1) + c] = s;
      }
  }
  std::vector<double> Ar((size_t)n * n), br(n);
  for (int r = 0; r < n; ++r) {
    for (int c = 0; c < n; ++c) {
      double s = H[(size_t)ik[r] * N + ik[c]];
      for (int i = 0; i < m; ++i) s -= H[(size_t)ik[r] * N + im[i]] * X[(size_t)i * (n + 1) + c];
      Ar[(size_t)r * n + c] = s;
    }
    double s = g[ik[r]];
    for (int i = 0; i < m; ++i) s -= H[(size_t)ik[r] * N + im[i]] * X[(size_t)i * (n + 1) + n];
    br[r] = s;
  }
  for (int r = 0; r < n; ++r)
    for (int c = r + 1; c < n; ++c) { const double s = 0.5 * (Ar[(size_t)r * n + c] + Ar[(size_t)c * n + r]); Ar[(size_t)r * n + c] = s; Ar[(size_t)c * n + r] = s; }
  std::vector<double> er, Vr;
  sym_eig_ql(n, Ar.data(), er, Vr);
  for (int a = 0; a < n; ++a) {
    const double S = er[a] > eps ? er[a] : 0.0, sq = std::sqrt(S), isq = S > 0 ? 1.0 / std::sqrt(S) : 0.0;
    double s = 0.0;
    for (int i = 0; i < n; ++i) { J0[(size_t)a * n + i] = sq * Vr[(size_t)i * n + a]; s += Vr[(size_t)i * n + a] * br[i]; }
    r0[a] = isq * s;
  }
  return n;
}

}  // namespace ctv
