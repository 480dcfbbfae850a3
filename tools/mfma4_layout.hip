#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_layout(double *out) {   // block b = (la, lb): one-hot A at lane la, one-hot B at lane lb (value 2 and 3 -> product 6)
  const int l = threadIdx.x, la = blockIdx.x / 64, lb = blockIdx.x % 64;
  double a = l == la ? 2.0 : 0.0, b = l == lb ? 3.0 : 0.0, c = 0.0;
  asm volatile("s_nop 4\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(c) : "v"(a), "v"(b));
  out[(size_t)blockIdx.x * 64 + l] = c;
}
int main() {
  double *d; static double h[64 * 64 * 64];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k_layout, dim3(64 * 64), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int n = 0;
  for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l) {
    const double v = h[((size_t)la * 64 + lb) * 64 + l];
    if (v != 0.0) { printf("%d %d %d %g\n", la, lb, l, v); ++n; }
  }
  fprintf(stderr, "nonzeros %d\n", n);
  return 0;
}
