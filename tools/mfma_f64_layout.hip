// Empirical register layout of v_mfma_f64_16x16x4_f64 on gfx950 (used by k_cholesky_solve).
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_layout.hip -o /tmp/l && /tmp/l
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *out) {   // A[16][4], B[4][16] row-major
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16];      // hypothesis: lane holds A[i = l%16][k = l/16]
  const double b = B[(l / 16) * 16 + l % 16];     // hypothesis: lane holds B[k = l/16][j = l%16]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  double A[64], B[64], D[256], out[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i + 17 * k;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 3 + 5 * j + 1000 * k;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; D[i * 16 + j] = s; }
  double *dA, *dB, *dO;
  hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dO, sizeof out);
  hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dO);
  hipMemcpy(out, dO, sizeof out, hipMemcpyDeviceToHost);
  int h1 = 1, h2 = 1, found = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const double v = out[l * 4 + r];
    if (v != D[(4 * (l / 16) + r) * 16 + l % 16]) h1 = 0;
    if (v != D[((l / 16) + 4 * r) * 16 + l % 16]) h2 = 0;
    int hit = -1;
    for (int e = 0; e < 256; ++e) if (D[e] == v) hit = e;
    if (hit < 0) found = 0;
    if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> D[%2d][%2d]\n", l, r, hit / 16, hit % 16);
  }
  printf("operand hypothesis consistent: %d\nH1 row=4*(l/16)+r col=l%%16 : %d\nH2 row=(l/16)+4*r col=l%%16 : %d\n", found, h1, h2);
  return 0;
}
