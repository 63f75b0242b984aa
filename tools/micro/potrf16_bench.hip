// Cost of one potrf_inv16 call (16 x 16 Cholesky + inverse by one wavefront, the serial core of every
// factorisation kernel): one workgroup repeats it on a fresh copy of an SPD block.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scratch/micro/potrf16_bench scratch/micro/potrf16_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../include/smcp_amd.h"
#include "../../smcp_amd/csrc/context.hpp"
#include "../../smcp_amd/csrc/front_generic.hip"
#include "../../smcp_amd/csrc/front_mfma.hip"
using namespace smcp;

__global__ void __launch_bounds__(256) k_bench(const double* A, double* out, int reps, int variant, int w) {
  __shared__ double D[16 * 17], Dinv[256];
  int f = 0;
  for (int r = 0; r < reps; ++r) {
    for (int e = threadIdx.x; e < 256; e += blockDim.x) D[(e & 15) + (e >> 4) * 17] = A[e];
    f |= variant ? potrf_inv16(D, 17, w, Dinv) : potrf_inv16_readlane(D, 17, w, Dinv);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 256; e += blockDim.x) { out[e] = D[(e & 15) + (e >> 4) * 17]; out[256 + e] = Dinv[e]; }
  if (threadIdx.x == 0) out[512] = f;
}

int main() {
  std::vector<double> A(256), L(256, 0.0);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24) - 0.5; };
  for (int j = 0; j < 16; ++j) for (int i = j; i < 16; ++i) L[i + 16 * j] = (i == j) ? 1.0 + fabs(rnd()) : rnd();
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double v = 0; for (int k = 0; k < 16; ++k) v += L[i + 16 * k] * L[j + 16 * k]; A[i + 16 * j] = v; }
  double *dA, *dout;
  hipMalloc(&dA, 256 * 8); hipMalloc(&dout, 520 * 8);
  hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
  for (int w : {16, 13, 5, 4})
  for (int variant = 0; variant < 2; ++variant) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 2000;
    hipLaunchKernelGGL(k_bench, dim3(1), dim3(256), 0, 0, dA, dout, 10, variant, w);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_bench, dim3(1), dim3(256), 0, 0, dA, dout, reps, variant, w);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> o(520);
    hipMemcpy(o.data(), dout, 520 * 8, hipMemcpyDeviceToHost);
    double eL = 0, eI = 0;
    for (int j = 0; j < w; ++j) for (int i = j; i < w; ++i) eL = fmax(eL, fabs(o[i + 16 * j] - L[i + 16 * j]));
    // Dinv[j + i * 16] = (L^-1)(i, j)?  check L * X = I with X(i, j) = Dinv[j + 16 i] transposed storage as the kernel writes it
    for (int i = 0; i < w; ++i) for (int j = 0; j < w; ++j) {
      double v = 0; for (int k = 0; k < w; ++k) v += L[i + 16 * k] * o[256 + k + 16 * j];
      eI = fmax(eI, fabs(v - (i == j)));
    }
    double ez = 0;    // outside the leading w x w block the inverse must read as zero
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (i >= w || j >= w || i < j) ez = fmax(ez, fabs(o[256 + i + 16 * j]));
    printf("w %2d variant %d: %.3f us per call  (fail %g, |L - Lref| %.2e, |L X - I| %.2e, outside %.1e)\n", w, variant, 1e3 * ms / reps, o[512], eL, eI, ez);
  }
  return 0;
}
