// Tile product of the 4096 front (config 2, k_lf_up2): lower tiles of  C = Z Li^T + Li Z^T,  Z and Li lower triangular,
// inner range [0, n0 + 64) per tile.  v0 = the product kernel's loop as it stands (gemm_tile64<1>: element loaders, guards,
// two barriers per slice); v1 = both operands staged row-contiguous with 16-byte loads, pointer increments, no guards,
// double-buffered LDS (one barrier per slice); v2 = v1 with 32-wide slices.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/tile_gemm.hip -o /tmp/tile_gemm && /tmp/tile_gemm [n] [nrhs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int LT = 64, LKC = 16, LSA = LT + 1, LSB = LKC + 1;

__device__ inline void lower_pair(int t, int& tm, int& tn) { tm = 0; while (t > tm) { t -= tm + 1; ++tm; } tn = t; }
__device__ inline void lower_pair_wide_first(int t, int nt, int& tm, int& tn) { int q = 0; while (t > q) { t -= q + 1; ++q; } tn = nt - 1 - q; tm = tn + t; }

template <class LA, class LB>
__device__ inline void gemm_v0(d4 (&acc)[2][2], int M, int N, int Kd, int m0, int n0, LA la, LB lb, double* sA, double* sB) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  double va[4], vb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      const int i = e & 63, kk = e >> 6;
      va[u] = (m0 + i < M && k0 + kk < Kd) ? la(m0 + i, k0 + kk) : 0.0;
      const int kb = e & 15, j = e >> 4;
      vb[u] = (n0 + j < N && k0 + kb < Kd) ? lb(k0 + kb, n0 + j) : 0.0;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < Kd; k0 += LKC) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      sA[(e & 63) + (e >> 6) * LSA] = va[u];
      sB[(e & 15) + (e >> 4) * LSB] = vb[u];
    }
    __syncthreads();
    if (k0 + LKC < Kd) fetch(k0 + LKC);
#pragma unroll
    for (int ks = 0; ks < LKC / 4; ++ks) {
      const int kk = 4 * ks + kq;
      const double a0 = sA[(32 * wm + l15) + kk * LSA], a1 = sA[(32 * wm + 16 + l15) + kk * LSA];
      const double b0 = sB[kk + (32 * wn + l15) * LSB], b1 = sB[kk + (32 * wn + 16 + l15) * LSB];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
    }
  }
}
__device__ inline void store_lower(const d4 (&acc)[2][2], int m0, int n0, double* C, int64_t ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 32 * (wave & 1) + 16 * a + l15, n = n0 + 32 * (wave >> 1) + 16 * b + kq + 4 * r;
        if (m >= n) C[m + (int64_t)n * ld] = acc[a][b][r];
      }
}
template <int ORDER>
__global__ void __launch_bounds__(256) k_v0(const double* Z, const double* Li, double* C, int n) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const int nt = n / LT;
  int tm, tn;
  if (ORDER) lower_pair_wide_first(blockIdx.x, nt, tm, tn); else lower_pair(blockIdx.x, tm, tn);
  const int m0 = tm * LT, n0 = tn * LT;
  const int64_t ld = n;
  C += (int64_t)blockIdx.z * n * n;
  d4 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = d4{0, 0, 0, 0};
  gemm_v0(acc, n, n, min(n, n0 + LT), m0, n0, [=](int m, int kk) { return Z[m + (int64_t)kk * ld]; }, [=](int kk, int nn_) { return Li[nn_ + (int64_t)kk * ld]; }, sA, sB);
  gemm_v0(acc, n, n, min(n, n0 + LT), m0, n0, [=](int m, int kk) { return Li[m + (int64_t)kk * ld]; }, [=](int kk, int nn_) { return Z[nn_ + (int64_t)kk * ld]; }, sA, sB);
  store_lower(acc, m0, n0, C, ld);
}

// ---- v1 / v2: row-contiguous staging of both operands, 16-byte loads, double-buffered LDS
template <int KC>
struct V1 {
  static constexpr int LDS = 80;                 // leading dimension of a staged slice (rows + 16: k and k + 1 land 32 banks apart)
  static constexpr int NLD = KC / 8;             // 16-byte loads per thread, operand and slice: 64 rows x KC / 2 / 256
  const double* pa; const double* pb;            // this thread's first row pair, current slice
  int64_t step;                                  // KC * ld
  d2 ra[NLD], rb[NLD];
  __device__ void init(const double* A, const double* B, int m0, int n0, int64_t ld) {
    const int tid = threadIdx.x, r2 = tid & 31, kk = tid >> 5;          // kk 0..7; load u covers k = kk + 8 u
    pa = A + m0 + 2 * r2 + (int64_t)kk * ld;
    pb = B + n0 + 2 * r2 + (int64_t)kk * ld;
    step = (int64_t)KC * ld;
  }
  __device__ void fetch(int64_t ld) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      ra[u] = *reinterpret_cast<const d2*>(pa + (int64_t)(8 * u) * ld);
      rb[u] = *reinterpret_cast<const d2*>(pb + (int64_t)(8 * u) * ld);
    }
    pa += step; pb += step;
  }
  __device__ void stage(double* sA, double* sB) {
    const int tid = threadIdx.x, r2 = tid & 31, kk = tid >> 5;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      *reinterpret_cast<d2*>(sA + 2 * r2 + (kk + 8 * u) * LDS) = ra[u];
      *reinterpret_cast<d2*>(sB + 2 * r2 + (kk + 8 * u) * LDS) = rb[u];
    }
  }
};
template <int KC>
__device__ inline void gemm_v1(d4 (&acc)[2][2], const double* A, const double* B, int64_t ld, int Kd, int m0, int n0, double* smem) {
  constexpr int LDS = V1<KC>::LDS, SL = KC * LDS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  V1<KC> ld_;
  ld_.init(A, B, m0, n0, ld);
  const int ns = Kd / KC;
  ld_.fetch(ld);
  __syncthreads();                               // (the previous product may still read the buffers)
  ld_.stage(smem, smem + 2 * SL);
  __syncthreads();
  for (int s = 0; s < ns; ++s) {
    const double* sA = smem + (s & 1) * SL;
    const double* sB = smem + 2 * SL + (s & 1) * SL;
    if (s + 1 < ns) ld_.fetch(ld);
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      const int kk = 4 * ks + kq;
      const double a0 = sA[(32 * wm + l15) + kk * LDS], a1 = sA[(32 * wm + 16 + l15) + kk * LDS];
      const double b0 = sB[(32 * wn + l15) + kk * LDS], b1 = sB[(32 * wn + 16 + l15) + kk * LDS];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
    }
    if (s + 1 < ns) ld_.stage(smem + ((s + 1) & 1) * SL, smem + 2 * SL + ((s + 1) & 1) * SL);
    __syncthreads();
  }
}
template <int KC, int ORDER>
__global__ void __launch_bounds__(256) k_v1(const double* Z, const double* Li, double* C, int n) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int nt = n / LT;
  int tm, tn;
  if (ORDER) lower_pair_wide_first(blockIdx.x, nt, tm, tn); else lower_pair(blockIdx.x, tm, tn);
  const int m0 = tm * LT, n0 = tn * LT;
  const int64_t ld = n;
  C += (int64_t)blockIdx.z * n * n;
  d4 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = d4{0, 0, 0, 0};
  const int Kd = min(n, n0 + LT);
  gemm_v1<KC>(acc, Z, Li, ld, Kd, m0, n0, smem);
  gemm_v1<KC>(acc, Li, Z, ld, Kd, m0, n0, smem);
  store_lower(acc, m0, n0, C, ld);
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096, nrhs = argc > 2 ? atoi(argv[2]) : 1;
  if (n % 64) { printf("n must be a multiple of 64\n"); return 1; }
  const int64_t nn = (int64_t)n * n;
  std::vector<double> hz(nn, 0.0), hl(nn, 0.0);
  unsigned long long s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)((s >> 33) & 0xffff) / 65536.0 - 0.5; };
  for (int j = 0; j < n; ++j) for (int i = j; i < n; ++i) { hz[i + (int64_t)j * n] = rnd(); hl[i + (int64_t)j * n] = rnd(); }
  double *Z, *Li, *C0, *C1;
  CHK(hipMalloc(&Z, nn * 8)); CHK(hipMalloc(&Li, nn * 8)); CHK(hipMalloc(&C0, nn * 8 * nrhs)); CHK(hipMalloc(&C1, nn * 8 * nrhs));
  CHK(hipMemcpy(Z, hz.data(), nn * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(Li, hl.data(), nn * 8, hipMemcpyHostToDevice));
  CHK(hipMemset(C0, 0, nn * 8 * nrhs)); CHK(hipMemset(C1, 0, nn * 8 * nrhs));
  const int nt = n / 64, tiles = nt * (nt + 1) / 2;
  double flops = 0;
  for (int tn = 0; tn < nt; ++tn) flops += (double)(nt - tn) * 2.0 * 2.0 * 64 * 64 * (64.0 * (tn + 1));
  flops *= nrhs;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch, double* out) -> int {
    launch(out);
    CHK(hipDeviceSynchronize());
    float best = 1e30f, tot = 0;
    const int reps = 6;
    for (int r = 0; r < reps; ++r) {
      CHK(hipEventRecord(e0, 0));
      launch(out);
      CHK(hipEventRecord(e1, 0));
      CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best; tot += ms;
    }
    printf("%-28s best %8.3f ms  mean %8.3f ms  %6.1f TFLOP/s (best)  %6.1f (mean)\n", name, best, tot / reps, flops / best / 1e9, flops / (tot / reps) / 1e9);
    return 0;
  };
  const dim3 grid(tiles, 1, nrhs);
  CHK(hipFuncSetAttribute((const void*)k_v1<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  CHK(hipFuncSetAttribute((const void*)k_v1<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  CHK(hipFuncSetAttribute((const void*)k_v1<32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  if (timeit("v0 row-major order", [&](double* o) { hipLaunchKernelGGL((k_v0<0>), grid, dim3(256), 0, 0, Z, Li, o, n); }, C0)) return 1;
  if (timeit("v0 wide tiles first", [&](double* o) { hipLaunchKernelGGL((k_v0<1>), grid, dim3(256), 0, 0, Z, Li, o, n); }, C0)) return 1;
  if (timeit("v1 (KC 16) row-major", [&](double* o) { hipLaunchKernelGGL((k_v1<16, 0>), grid, dim3(256), 4 * 16 * 80 * 8, 0, Z, Li, o, n); }, C1)) return 1;
  if (timeit("v1 (KC 16) wide first", [&](double* o) { hipLaunchKernelGGL((k_v1<16, 1>), grid, dim3(256), 4 * 16 * 80 * 8, 0, Z, Li, o, n); }, C1)) return 1;
  if (timeit("v2 (KC 32) wide first", [&](double* o) { hipLaunchKernelGGL((k_v1<32, 1>), grid, dim3(256), 4 * 32 * 80 * 8, 0, Z, Li, o, n); }, C1)) return 1;
  std::vector<double> h0(nn), h1(nn);
  CHK(hipMemcpy(h0.data(), C0, nn * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(h1.data(), C1, nn * 8, hipMemcpyDeviceToHost));
  double worst = 0, big = 0;
  for (int64_t e = 0; e < nn; ++e) { worst = fmax(worst, fabs(h0[e] - h1[e])); big = fmax(big, fabs(h0[e])); }
  printf("max |v0 - v2| = %.3e  (max |v0| = %.3e)\n", worst, big);
  return 0;
}
