// mfma_f64_4x4_probe.hip -- is v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 products per instruction: 512 flop) worth using for the
// SYMMETRIC tiles (J^T J of 16 columns: 10 of 16 4 x 4 blocks are needed, 3 instructions instead of one 16 x 16 x 4)?  Only if it runs at the
// 16 x 16 x 4 instruction's flop rate.  Measured here: clocks per instruction for a lone wave, dependent and 3-way independent, and the layout
// (which lane holds which operand / result element) by feeding unit vectors.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_4x4_probe.hip -o /tmp/mfma4 && /tmp/mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k_rate(long long *out, double *sink, int mode) {
  double a = 1.0 + threadIdx.x * 1e-3, kk = 0.999, c0 = 0.0, c1 = 0.0, c2 = 0.0;
  f64x4 acc = {0, 0, 0, 0};
  __syncthreads();
  long long t0 = clock64();
  if (mode == 0) asm volatile(REP256("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\t") : "+v"(c0) : "v"(a), "v"(kk));
  if (mode == 1) asm volatile(REP64("v_mfma_f64_4x4x4_4b_f64 %0, %3, %4, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %3, %4, %1\n\tv_mfma_f64_4x4x4_4b_f64 %2, %3, %4, %2\n\tv_mfma_f64_4x4x4_4b_f64 %0, %3, %4, %0\n\t") : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(a), "v"(kk));
  if (mode == 2) asm volatile(REP256("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\t") : "+v"(acc) : "v"(a), "v"(kk));
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  sink[threadIdx.x] = c0 + c1 + c2 + acc[0];
}
// layout: A operand = 1 in lane la only, B operand = 1 in lane lb only; which lanes of D become 1?
__global__ void k_layout(double *out, int la, int lb) {
  const int l = threadIdx.x;
  double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0, c = 0.0;
  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7" : "+v"(c) : "v"(a), "v"(b));
  out[l] = c;
}
int main() {
  long long *out, h[16];
  double *sink, *lay, hl[64];
  hipMalloc(&out, sizeof h); hipMalloc(&sink, 1024 * 8); hipMalloc(&lay, 64 * 8);
  const char *names[] = {"v_mfma_f64_4x4x4_4b dependent", "v_mfma_f64_4x4x4_4b 3 chains", "v_mfma_f64_16x16x4 dependent"};
  for (int mode = 0; mode < 3; ++mode)
    for (int waves : {1, 4, 8}) {
      for (int pass = 0; pass < 2; ++pass) { hipLaunchKernelGGL(k_rate, dim3(1), dim3(64 * waves), 0, 0, out, sink, mode); hipDeviceSynchronize(); }
      hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
      long long mx = 0;
      for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
      const int per_simd = waves < 4 ? 1 : waves / 4;
      printf("%-32s waves/SIMD %d: %6.2f clocks per instruction per SIMD  (%s flop / clock / SIMD = %.1f)\n", names[mode], per_simd, mx / (256.0 * per_simd),
             mode < 2 ? "512" : "2048", (mode < 2 ? 512.0 : 2048.0) * 256.0 * per_simd / mx);
    }
  // layout table: for each (la, lb) pair that gives a non-zero, print the D lanes
  printf("layout: A lane la x B lane lb -> D lanes (value 1)\n");
  for (int la = 0; la < 64; la += 1) {
    for (int lb = 0; lb < 64; ++lb) {
      hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, lay, la, lb);
      hipMemcpy(hl, lay, sizeof hl, hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; ++l) if (hl[l] != 0.0 && (la < 8 || la % 16 == 0) ) printf("  A[%2d] B[%2d] -> D[%2d]\n", la, lb, l);
    }
    if (la >= 20) break;
  }
  return 0;
}
