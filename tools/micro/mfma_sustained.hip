// Sustained rate of v_mfma_f64_16x16x4_f64 over tens of milliseconds: same operands every instruction (MODE 0), operands
// rotating over four register pairs with static indices (MODE 1), operand values of very different bit patterns (MODE 2).
// hipcc --offload-arch=gfx950 -O3 -o mfma_sustained mfma_sustained.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(double* out, const double* in, int n) {
  d4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  double a[4], b[4];
  for (int c = 0; c < 4; ++c) { a[c] = in[threadIdx.x * 8 + c]; b[c] = in[threadIdx.x * 8 + 4 + c]; }
  if (MODE == 0) for (int c = 1; c < 4; ++c) { a[c] = a[0]; b[c] = b[0]; }
  for (int i = 0; i < n; i += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[MODE ? ((c + u) & 3) : 0], b[MODE ? c : 0], acc[c], 0, 0, 0);
  }
  double s = 0.0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(double* out, const double* in, int n, int w) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * w), 0, 0, out, in, 1024);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * w), 0, 0, out, in, n);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double nm = 256.0 * w * 4 * n * 4;
  printf("mode %d n %d waves/SIMD %d: %.2f ms -> %.1f TFLOP/s\n", MODE, n, w, ms, nm * 2048 / (ms * 1e-3) / 1e12);
}
int main() {
  double *out, *in;
  (void)hipMalloc(&out, sizeof(double) * 256 * 1024);
  (void)hipMalloc(&in, sizeof(double) * 1024 * 8);
  static double h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = 1e-3 * ((i * 2654435761u) % 1000) - 0.5;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int n : {16384, 262144})
    for (int w : {1, 2}) { run<0>(out, in, n, w); run<1>(out, in, n, w); }
  for (int i = 0; i < 8192; ++i) h[i] = (i & 1) ? 0.0 : 1.0;           // trivial bit patterns
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<1>(out, in, 262144, 2);
  return 0;
}
