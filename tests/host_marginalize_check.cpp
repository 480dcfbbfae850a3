// Test-only shim: exposes the library's HOST marginalisation algebra (ctrl-vio_amd/csrc/marginalize.hpp) to ctypes so that
// tests/test_marginalize_host.py can compare it with the oracle without a GPU.
#include "../ctrl-vio_amd/csrc/marginalize.hpp"
#include <cstring>
extern "C" int hm_marginalize_dense(int N, const double *H, const double *g, const int8_t *role, double eps, int32_t *kept, double *J0, double *r0) {
  std::vector<int32_t> k; std::vector<double> J, r;
  const int n = ctv::marginalize_dense(N, H, g, role, eps, k, J, r);
  std::memcpy(kept, k.data(), sizeof(int32_t) * n);
  std::memcpy(J0, J.data(), sizeof(double) * (size_t)n * n);
  std::memcpy(r0, r.data(), sizeof(double) * n);
  return n;
}
extern "C" void hm_sym_eig(int n, const double *A, double *d, double *V) {
  std::vector<double> dd, VV;
  ctv::sym_eig_ql(n, A, dd, VV);
  std::memcpy(d, dd.data(), sizeof(double) * n);
  std::memcpy(V, VV.data(), sizeof(double) * (size_t)n * n);
}
