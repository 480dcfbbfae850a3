// Where do the 16 waves of a 1024-thread workgroup land?  Prints HW_REG_HW_ID fields (gfx9: wave_id [3:0], simd_id [5:4], pipe [7:6], cu [11:8],
// sh [12], se [15:13]) of every wave of a few workgroups.  k_cholesky_chain assumes waves go to the SIMDs round robin (wave w on SIMD w mod 4), so
// that waves 4, 8, 12 share the chain wave's SIMD; it stays correct under any placement, but only that one isolates the pivot chain.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/simd_probe tools/simd_probe.hip && /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned *out) {
  const int wave = threadIdx.x >> 6;
  const unsigned id = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = id;
}
int main() {
  const int nb = 600;
  unsigned *d, h[600 * 16];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(nb), dim3(1024), 100 * 1024, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < nb; ++b) {
    const unsigned s0 = (h[b * 16] >> 4) & 3;
    for (int w = 0; w < 16; ++w) {
      const unsigned simd = (h[b * 16 + w] >> 4) & 3;
      if (simd != ((s0 + w) & 3)) ++bad;
    }
    if (b < 4 || b == 300) {
      std::printf("block %3d: simd of waves 0..15 =", b);
      for (int w = 0; w < 16; ++w) std::printf(" %u", (h[b * 16 + w] >> 4) & 3);
      std::printf("   cu %u se %u\n", (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
    }
  }
  std::printf("waves NOT at (simd of wave 0 + w) mod 4: %d of %d\n", bad, nb * 16);
  return 0;
}
