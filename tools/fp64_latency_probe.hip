// fp64_latency_probe.hip -- issue and dependent-issue cost (core clocks, one wave alone on a CU) of the instructions the serial parts of the
// solver are made of: v_fma_f64, v_mul_f64, v_rsq_f64, v_rcp_f64, v_fmac_f64_dpp row_newbcast, v_readlane_b32 -> VALU, v_permlane16_swap_b32,
// v_cndmask.  Each test runs N copies of a pattern inside s_memtime brackets; "dep" = every instruction consumes the previous result.
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_latency_probe.hip -o /tmp/fp64_lat && /tmp/fp64_lat
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define TEST(ID, N, BODY)                                                              \
  if (which == ID) {                                                                   \
    long long t0 = clock64();                                                          \
    asm volatile(BODY : "+v"(a), "+v"(b), "+v"(c), "+v"(e), "+v"(f), "+v"(x), "+v"(y) : "v"(k) : "s90", "s91", "s92", "s93", "vcc"); \
    long long t1 = clock64();                                                          \
    if (threadIdx.x == 0) out[ID] = (double)(t1 - t0) / (N);                           \
  }

__global__ void k_lat(double *out, double *sink, int which) {
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4, c = 2.0, e = 3.0, f = 4.0, k = 0.999;
  int x = threadIdx.x, y = 64 - threadIdx.x;
  TEST(0, 64, REP64("v_fma_f64 %0, %0, %7, %7\n\t"))                                                       // dependent fma
  TEST(1, 64, REP16("v_fma_f64 %0, %0, %7, %7\n\tv_fma_f64 %1, %1, %7, %7\n\tv_fma_f64 %2, %2, %7, %7\n\tv_fma_f64 %3, %3, %7, %7\n\t"))   // 4 independent chains
  TEST(2, 64, REP64("v_mul_f64 %0, %0, %7\n\t"))
  TEST(3, 64, REP64("v_rsq_f64 %0, %0\n\t"))                                                               // dependent rsq
  TEST(4, 64, REP16("v_rsq_f64 %0, %7\n\tv_rsq_f64 %1, %7\n\tv_rsq_f64 %2, %7\n\tv_rsq_f64 %3, %7\n\t"))   // independent rsq
  TEST(5, 64, REP64("v_rsq_f64 %0, %0\n\ts_nop 0\n\tv_mul_f64 %0, %0, %7\n\t"))                            // rsq -> mul pairs (2 instr per N)
  TEST(6, 64, REP64("v_fmac_f64_dpp %0, %7, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"))            // dpp fmac dependent through src1/dst only
  TEST(7, 64, REP16("v_fmac_f64_dpp %0, %7, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %7, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f64_dpp %2, %7, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %7, %7 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"))   // independent dpp fmac
  TEST(8, 64, REP64("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"))  // dpp source = previous result (2 wait states)
  TEST(9, 64, REP64("v_readlane_b32 s90, %5, 3\n\tv_readlane_b32 s91, %6, 3\n\ts_nop 1\n\tv_fma_f64 %0, s[90:91], %7, %7\n\tv_cvt_i32_f64 %5, %0\n\ts_nop 0\n\t"))   // readlane pair -> fma -> readlane
  TEST(10, 64, REP16("v_readlane_b32 s90, %5, 3\n\tv_readlane_b32 s91, %5, 4\n\tv_readlane_b32 s92, %6, 5\n\tv_readlane_b32 s93, %6, 6\n\t"))  // independent readlanes (4 per N)
  TEST(11, 64, REP64("s_nop 1\n\tv_permlane16_swap_b32 %5, %6\n\t"))                                       // swap chain
  TEST(12, 64, REP64("v_cmp_gt_f64 vcc, %0, %7\n\tv_cndmask_b32 %5, %5, %6, vcc\n\tv_cvt_f64_i32 %0, %5\n\t"))                    // compare + select on the chain (lo dword only)
  TEST(13, 64, REP64("v_mov_b32 %5, %5\n\t"))                                                               // 32-bit dependent mov
  TEST(14, 64, REP64("v_add_f64 %0, %0, %7\n\t"))
  TEST(15, 64, REP64("v_rcp_f64 %0, %0\n\t"))
  TEST(16, 64, REP16("v_fma_f64 %0, %0, %7, %7\n\tv_fmac_f64_dpp %1, %7, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %7, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %3, %7, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"))          // a dependent fma chain with three independent dpp fmacs between links (per 4 instr)
  TEST(17, 64, REP16("v_fma_f64 %0, %0, %7, %7\n\tv_fma_f64 %1, %7, %7, %1\n\tv_fma_f64 %2, %7, %7, %2\n\tv_fma_f64 %3, %7, %7, %3\n\t"))
  sink[threadIdx.x] = a + b + c + e + f + x + y;
}

int main() {
  const char *names[] = {"fma dep", "fma 4 chains", "mul dep", "rsq dep", "rsq indep", "rsq, nop, mul (per triple)", "fmac_dpp dep via acc", "fmac_dpp indep",
                         "nop 1 + fmac_dpp on its own result", "readlane x2, nop, fma, cvt, nop (per group)", "readlane indep", "nop 1 + permlane16_swap dep", "cmp, cndmask, cvt (per triple)", "v_mov_b32 dep", "add dep", "rcp dep",
                         "fma dep + 3 dpp between", "fma dep + 3 fma between"};
  double *out, *sink, h[18];
  hipMalloc(&out, sizeof(h));
  hipMalloc(&sink, 64 * 8);
  hipMemset(out, 0, sizeof(h));
  for (int pass = 0; pass < 2; ++pass)
    for (int w = 0; w < 18; ++w) { hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, sink, w); hipDeviceSynchronize(); }
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  for (int w = 0; w < 18; ++w) printf("%-44s %7.1f clocks per instruction (per listed group where the pattern is one)\n", names[w], h[w]);
  return 0;
}
