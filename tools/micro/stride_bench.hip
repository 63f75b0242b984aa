// Microbenchmark: does the RHS-major layout (each workgroup jumps `stride` bytes per pass) cost bandwidth
// against a clique-major layout (each workgroup streams a contiguous range)?  Mimics the mid-level
// launch of k_hess_up_n16: NWG workgroups, each R passes; per pass it reads NSEG segments of SEG bytes and a panel,
// writes a panel and an update.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// layout 0: addr(unit u, pass r) = r * ustride + u * usz      (RHS-major)
// layout 1: addr(unit u, pass r) = u * (R * usz) + r * usz    (clique-major)
template <int PF>
__global__ void __launch_bounds__(512) k_stream(const double* __restrict__ in, double* __restrict__ out, int nunit, int R, int usz_in,
                                                int usz_out, int layout, int upw) {
  // each workgroup owns `upw` consecutive units (cliques); loops passes
  double acc = 0.0;
  for (int uu = 0; uu < upw; ++uu) {
    const int64_t u = (int64_t)blockIdx.x * upw + uu;
    if (u >= nunit) break;
    for (int r = 0; r < R; ++r) {
      const int64_t ai = layout ? (u * R + r) * (int64_t)usz_in : ((int64_t)r * nunit + u) * usz_in;
      const int64_t ao = layout ? (u * R + r) * (int64_t)usz_out : ((int64_t)r * nunit + u) * usz_out;
      double v = 0.0;
      for (int e = threadIdx.x; e < usz_in; e += 512) v += in[ai + e];
      acc += v;
      for (int e = threadIdx.x; e < usz_out; e += 512) out[ao + e] = v + e;
    }
  }
  if (acc == 123.456) out[0] = acc;
}

int main() {
  const int R = 100;
  const int nunit = 896;                 // mid cliques
  const int usz_in = 8 * 496 + 1185;     // doubles read per (clique, rhs): 8 children's packed updates + panel
  const int usz_out = 2080 + 1185;       // doubles written: packed update + panel
  size_t nin = (size_t)R * nunit * usz_in, nout = (size_t)R * 7168 * (180 + 496) + 1024;   // out also serves the leaf-like case
  setvbuf(stdout, nullptr, _IOLBF, 0);
  double *in, *out;
  CK(hipMalloc(&in, nin * 8)); CK(hipMalloc(&out, nout * 8));
  CK(hipMemset(in, 0, nin * 8)); CK(hipMemset(out, 0, nout * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int layout = 0; layout < 2; ++layout)
    for (int split = 1; split <= 4; split *= 2) {   // workgroups per clique (rhs split)
      // emulate rhs split by launching nunit*split workgroups each doing R/split passes: use units = nunit*split, R/split
      int nu = nunit * split, Rs = R / split;
      for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        k_stream<0><<<nu, 512>>>(in, out, nu, Rs, usz_in, usz_out, layout, 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 2) printf("layout %d (%s) wgs %d passes %d: %.3f ms  %.2f TB/s\n", layout, layout ? "clique-major" : "rhs-major", nu, Rs, ms,
                            (double)nu * Rs * (usz_in + usz_out) * 8 / ms / 1e9);
      }
    }
  // leaf-like: 7168 units, small segments
  {
    const int nl = 7168, uin = 180, uout = 180 + 496;
    if ((size_t)nl * R * uin > nin || (size_t)nl * R * uout > nout) { printf("size check failed\n"); return 1; }
    for (int layout = 0; layout < 2; ++layout)
      for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        k_stream<0><<<nl / 4, 512>>>(in, out, nl, R, uin, uout, layout, 4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 2) printf("leaf-like layout %d: %.3f ms  %.2f TB/s\n", layout, ms, (double)nl * R * (uin + uout) * 8 / ms / 1e9);
      }
  }
  return 0;
}
