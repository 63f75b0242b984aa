// Microbenchmark: throughput of ds_add_f64 (LDS atomic add, double) against a plain LDS read-modify-write and
// plain LDS stores, one workgroup of 1024 threads per CU, conflict-free addresses (lane-consecutive doubles).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NT = 1024, ITER = 256, LDSN = 16384;   // 128 KB of doubles
template <int MODE>
__global__ void __launch_bounds__(NT) k(double* out, long long* cyc) {
  __shared__ double T[LDSN];
  for (int e = threadIdx.x; e < LDSN; e += NT) T[e] = 0.0;
  __syncthreads();
  const long long t0 = clock64();
  double v = 1.0 + threadIdx.x;
  for (int it = 0; it < ITER; ++it) {
    const int idx = (threadIdx.x + it * 1031) & (LDSN - 1);        // consecutive lanes -> consecutive doubles
    if (MODE == 0) unsafeAtomicAdd(&T[idx], v);
    else if (MODE == 1) T[idx] += v;                               // not atomic across waves: throughput only
    else T[idx] = v;
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  double s = 0.0;
  for (int e = threadIdx.x; e < LDSN; e += NT) s += T[e];
  out[blockIdx.x * NT + threadIdx.x] = s;
}
int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  double* out; long long* cyc;
  const int nb = 256;
  CK(hipMalloc(&out, sizeof(double) * nb * NT)); CK(hipMalloc(&cyc, sizeof(long long) * nb));
  long long h[nb];
  const char* names[3] = {"ds_add_f64 (unsafeAtomicAdd)", "plain LDS read-modify-write", "plain LDS store"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) k<0><<<nb, NT>>>(out, cyc);
      else if (mode == 1) k<1><<<nb, NT>>>(out, cyc);
      else k<2><<<nb, NT>>>(out, cyc);
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < nb; ++i) avg += h[i]; avg /= nb;
    printf("%-32s %9.0f clock64 ticks per workgroup for %d lane-ops -> %.3f ticks per lane-op (clock64 = 100 MHz ticks if constant clock)\n",
           names[mode], avg, NT * ITER, avg / (NT * ITER));
  }
  // wall-clock version
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 20; ++rep) {
      if (mode == 0) k<0><<<nb, NT>>>(out, cyc);
      else if (mode == 1) k<1><<<nb, NT>>>(out, cyc);
      else k<2><<<nb, NT>>>(out, cyc);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-32s %.1f us per launch: %.2f lane-ops per ns per CU\n", names[mode], 1e3 * ms / 20, (double)NT * ITER / (1e6 * ms / 20));
  }
  return 0;
}
