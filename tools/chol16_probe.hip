// chol16_probe.hip -- the 16 x 16 diagonal tile of k_cholesky_tiles on its own: the v_readlane variant (chol16_from) against the DPP variant
// (chol16_dpp), one wave, checked against a host Cholesky + inverse and timed with s_memtime over REP tiles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ctrl-vio_amd/csrc tools/chol16_probe.hip -o /tmp/chol16_probe && /tmp/chol16_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.hpp"

using namespace ctv;

template <int VARIANT> __global__ __launch_bounds__(1024) void k_probe(const double *tiles, double *out, long long *cycles, int rep, int nreal) {
  __shared__ double Dg[16 * 17], Id[16 * 17];
  for (int i = threadIdx.x; i < 16 * 17; i += blockDim.x) Id[i] = (i / 17 == i % 17) ? 1.0 : 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, l15 = lane & 15;
  if (threadIdx.x >= 64) return;
  long long t0 = 0, acc = 0, accl = 0, accf = 0;
  int bad = 0;
  for (int it = 0; it < rep; ++it) {
    const double *A = tiles + 256 * it;
    for (int i = lane; i < 256; i += 64) Dg[(i / 16) * 17 + i % 16] = A[i];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    t0 = clock64();
    double v[16];
    int opaque0;
    asm volatile("s_mov_b32 %0, 0" : "=s"(opaque0));
    const int lz = l15 + opaque0;
    if (VARIANT == 0) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double a = Dg[l15 * 17 + c];
        v[c] = lane < 16 ? (c <= lz ? a : 0.0) : (c == lz ? 1.0 : 0.0);
      }
    } else {   // even rows: the tile's rows (whole rows: the upper half is never read); odd rows: the identity, from LDS as well -- no selects
      const double *src = ((lane & 16) ? Id : Dg) + lz * 17;
#pragma unroll
      for (int c = 0; c < 16; ++c) v[c] = src[c];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const long long t1 = clock64();
    if (VARIANT == 0) {
      double di0 = 1.0;
      if (nreal > 0) di0 = chol_pivot_rsqrt(readlane_d(v[0], 0), bad);
      chol16_from<0>(v, di0, nreal, bad);
    } else {
      chol16_dpp(v, nreal, bad);
    }
    const long long t2 = clock64();
    if (lane >= 16 && lane < 32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) Dg[i * 17 + l15] = v[i];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    acc += clock64() - t0; accl += t1 - t0; accf += t2 - t1;
    // out: [it][0..255] = L (row-major, lower), [it][256..511] = Linv (row-major)
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 16; ++c) out[512 * it + 16 * lane + c] = v[c];
    }
    for (int i = lane; i < 256; i += 64) out[512 * it + 256 + i] = Dg[(i / 16) * 17 + i % 16];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) { cycles[0] = acc; cycles[1] = bad; cycles[2] = accl; cycles[3] = accf; }
}

static void host_ref(const double *A, int nreal, double *L, double *Li) {   // rows >= nreal: pivot forced to 1 (as the kernel)
  for (int i = 0; i < 256; ++i) L[i] = 0.0;
  std::vector<double> W(A, A + 256);
  for (int j = 0; j < 16; ++j) {
    const double d = j < nreal ? 1.0 / std::sqrt(W[16 * j + j]) : 1.0;
    for (int i = j; i < 16; ++i) L[16 * i + j] = W[16 * i + j] * d;
    for (int c = j + 1; c < 16; ++c)
      for (int i = c; i < 16; ++i) W[16 * i + c] -= L[16 * i + j] * L[16 * c + j];
  }
  // the kernel's "inverse" follows the same recurrence on the identity: X = (unit pivots where forced) forward substitution
  for (int col = 0; col < 16; ++col) {
    double x[16];
    for (int c = 0; c < 16; ++c) x[c] = c == col ? 1.0 : 0.0;
    for (int j = 0; j < 16; ++j) {
      const double d = j < nreal ? 1.0 / L[16 * j + j] : 1.0;
      x[j] *= d;
      for (int c = j + 1; c < 16; ++c) x[c] -= x[j] * L[16 * c + j];
    }
    for (int c = 0; c < 16; ++c) Li[16 * c + col] = x[c];
  }
}

int main() {
  const int rep = 64;
  std::vector<double> tiles(256 * rep);
  srand(7);
  for (int it = 0; it < rep; ++it) {
    double B[16][16];
    for (auto &r : B) for (double &x : r) x = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = i == j ? 0.5 : 0.0;
        for (int k = 0; k < 16; ++k) s += B[i][k] * B[j][k];
        tiles[256 * it + 16 * i + j] = s;
      }
  }
  double *dt, *dout;
  long long *dc;
  hipMalloc(&dt, tiles.size() * 8);
  hipMalloc(&dout, 512 * rep * 8);
  hipMalloc(&dc, 32);
  hipMemcpy(dt, tiles.data(), tiles.size() * 8, hipMemcpyHostToDevice);
  int fail = 0;
  for (int nreal : {16, 11}) {
    for (int variant = 0; variant < 2; ++variant) {
      std::vector<double> out(512 * rep);
      long long cyc[4];
      for (int pass = 0; pass < 2; ++pass) {
        hipMemset(dout, 0, 512 * rep * 8);
        if (variant == 0) hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(1024), 0, 0, dt, dout, dc, rep, nreal);
        else hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(1024), 0, 0, dt, dout, dc, rep, nreal);
        hipDeviceSynchronize();
      }
      hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
      hipMemcpy(cyc, dc, 32, hipMemcpyDeviceToHost);
      double eL = 0.0, eI = 0.0;
      for (int it = 0; it < rep; ++it) {
        double L[256], Li[256];
        host_ref(&tiles[256 * it], nreal, L, Li);
        for (int i = 0; i < 16; ++i)
          for (int j = 0; j <= i; ++j) {
            eL = std::fmax(eL, std::fabs(out[512 * it + 16 * i + j] - L[16 * i + j]) / (1.0 + std::fabs(L[16 * i + j])));
            eI = std::fmax(eI, std::fabs(out[512 * it + 256 + 16 * i + j] - Li[16 * i + j]) / (1.0 + std::fabs(Li[16 * i + j])));
          }
      }
      const bool ok = eL < 1e-12 && eI < 1e-10 && cyc[1] == 0;
      fail |= !ok;
      printf("nreal %2d  %-9s  cycles/tile %7.0f (load %5.0f, factor %5.0f)  max err L %.2e  Linv %.2e  bad %lld  %s\n", nreal, variant ? "dpp" : "readlane", cyc[0] / (double)rep, cyc[2] / (double)rep, cyc[3] / (double)rep, eL, eI,
             cyc[1], ok ? "ok" : "MISMATCH");
    }
  }
  // a tile that is not positive definite (one diagonal entry negated): both variants must flag it
  {
    std::vector<double> bad_tile(tiles.begin(), tiles.begin() + 256);
    bad_tile[16 * 9 + 9] = -bad_tile[16 * 9 + 9];
    hipMemcpy(dt, bad_tile.data(), 256 * 8, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; ++variant) {
      long long cyc[4];
      if (variant == 0) hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(1024), 0, 0, dt, dout, dc, 1, 16);
      else hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(1024), 0, 0, dt, dout, dc, 1, 16);
      hipDeviceSynchronize();
      hipMemcpy(cyc, dc, 32, hipMemcpyDeviceToHost);
      printf("indefinite tile, %-9s: flagged %lld  %s\n", variant ? "dpp" : "readlane", cyc[1], cyc[1] ? "ok" : "NOT FLAGGED");
      fail |= cyc[1] == 0;
    }
  }
  return fail;
}
