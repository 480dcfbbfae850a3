// issue_rate_probe.hip -- does a SIMD issue fp64 vector instructions faster with more than one wave resident?  W waves per SIMD each run a chain of
// N dependent (or 4-way independent) v_fma_f64 / v_mfma_f64_16x16x4_f64; reported: clocks per instruction PER SIMD (wall clocks of the slowest wave / (W N)).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_rate_probe.hip -o /tmp/issue_rate && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(long long *out, double *sink, int mode) {
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5, c = 2.0, e = 3.0, kk = 0.999;
  f64x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  __syncthreads();
  long long t0 = clock64();
  if (mode == 0) asm volatile(REP256("v_fma_f64 %0, %0, %4, %4\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(e) : "v"(kk));
  if (mode == 1) asm volatile(REP64("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(e) : "v"(kk));
  if (mode == 2) asm volatile(REP256("v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\t") : "+v"(acc), "+v"(acc2) : "v"(a), "v"(kk));
  if (mode == 3) asm volatile(REP64("v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n\tv_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n\t") : "+v"(acc), "+v"(acc2) : "v"(a), "v"(kk));
  if (mode == 4) asm volatile(REP64("v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_fma_f64 %4, %4, %3, %3\n\tv_fma_f64 %5, %5, %3, %3\n\tv_fma_f64 %4, %4, %3, %3\n\t") : "+v"(acc), "+v"(acc2) : "v"(a), "v"(kk), "v"(b), "v"(c));   // 1 MFMA + 3 FMA
  if (mode == 5) asm volatile(REP256("v_mul_f32 %0, %0, %1\n\t") : "+v"(*(float *)&a) : "v"(*(float *)&kk));
  long long t1 = clock64();
  __syncthreads();
  long long t2 = clock64();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t1 - t0; out[2 * (threadIdx.x >> 6) + 1] = t2 - t0; }
  sink[threadIdx.x] = a + b + c + e + acc[0] + acc2[1];
}
int main() {
  long long *out, h[32];
  double *sink;
  hipMalloc(&out, sizeof(h)); hipMalloc(&sink, 1024 * 8);
  const char *names[] = {"v_fma_f64 dependent", "v_fma_f64 4 chains", "mfma_f64_16x16x4 dependent", "mfma_f64_16x16x4 2 chains", "1 mfma + 3 fma", "v_mul_f32 dependent"};
  for (int mode = 0; mode < 6; ++mode)
    for (int waves : {1, 4, 8, 16}) {   // 4 SIMDs per CU: 4 waves = 1 per SIMD, 8 = 2 per SIMD, 16 = 4 per SIMD
      for (int pass = 0; pass < 2; ++pass) { hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, sink, mode); hipDeviceSynchronize(); }
      hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      long long mx = 0, own = 0;
      for (int w = 0; w < waves; ++w) { if (h[2 * w + 1] > mx) mx = h[2 * w + 1]; own += h[2 * w]; }
      const int per_simd = waves < 4 ? 1 : waves / 4;
      printf("%-28s waves/SIMD %d: a wave's own 256 instr %6.0f clocks;  all done after %6lld clocks = %5.2f clocks per instruction per SIMD\n", names[mode], per_simd,
             own / (double)waves, mx, mx / (256.0 * per_simd));
    }
  return 0;
}
