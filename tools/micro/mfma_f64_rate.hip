// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: cycles per instruction for C independent accumulator chains per
// wave and W waves per SIMD (one workgroup per CU).  build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int C>
__global__ void k(double* out, unsigned long long* cyc, int n) {
  d4 acc[C];
  for (int c = 0; c < C; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4;
  __syncthreads();
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  unsigned long long t1 = clock64();
  double s = 0.0;
  for (int c = 0; c < C; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int C>
void run(int waves_per_simd, int nblk) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(double) * nblk * 1024);
  hipMalloc(&cyc, sizeof(unsigned long long) * nblk);
  const int n = 65536, thr = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<C>, dim3(nblk), dim3(thr), 0, 0, out, cyc, n);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<C>, dim3(nblk), dim3(thr), 0, 0, out, cyc, n);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double nm = (double)nblk * waves_per_simd * 4 * n * C;
  printf("chains %d, waves/SIMD %d, blocks %d: %.1f clock64 ticks per MFMA per wave; %.3f ms -> %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", C, waves_per_simd, nblk,
         (double)h[0] / ((double)n * C), ms, nm * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)n * C * waves_per_simd));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int nblk : {256}) {
    run<1>(1, nblk); run<2>(1, nblk); run<4>(1, nblk);
    run<1>(2, nblk); run<2>(2, nblk); run<1>(3, nblk);
  }
  return 0;
}
