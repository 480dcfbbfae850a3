// LDS atomic throughput on gfx950: float add vs u32 add vs plain read-add-write, 8 waves per workgroup, conflict-free
// addresses (lane -> distinct bank) and a 4-way conflicting pattern.  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int STRIDE> __global__ __launch_bounds__(512) void k(float *out, int iters, long long *cyc) {
  __shared__ float buf[16384];
  unsigned *ub = reinterpret_cast<unsigned *>(buf);
  for (int i = threadIdx.x; i < 16384; i += 512) buf[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = ((lane * STRIDE) + 67 * u + 131 * wave + 17 * it) & 16383;
      if (MODE == 0) atomicAdd(&buf[idx], 1.0f);
      else if (MODE == 1) atomicAdd(&ub[idx], 1u);
      else if (MODE == 2) buf[idx] += 1.0f;
      else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long *>(buf) + (idx >> 1), 1ull);
      else atomicAdd(reinterpret_cast<double *>(buf) + (idx >> 1), 1.0);
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = buf[threadIdx.x];
}
template <int MODE, int STRIDE> void run(const char *name) {
  float *out; long long *cyc, h;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  const int iters = 1000;
  hipLaunchKernelGGL((k<MODE, STRIDE>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
  hipLaunchKernelGGL((k<MODE, STRIDE>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s stride %2d: %.1f cycles per wave-instruction (8 waves/CU each issuing)\n", name, STRIDE, (double)h / (iters * 8));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 1>("ds_add_f32"); run<0, 4>("ds_add_f32"); run<0, 32>("ds_add_f32");
  run<1, 1>("ds_add_u32"); run<1, 4>("ds_add_u32"); run<1, 32>("ds_add_u32");
  run<2, 1>("read+add+write (non-atomic)"); run<2, 4>("read+add+write (non-atomic)");
  run<3, 2>("ds_add_u64"); run<3, 8>("ds_add_u64"); run<4, 2>("ds_add_f64"); run<4, 8>("ds_add_f64");
  return 0;
}
