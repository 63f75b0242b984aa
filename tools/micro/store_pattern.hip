// How fast can the chip WRITE the output of the family sweep?  896 families x 100 right-hand sides; per (family, rhs)
// 2625 doubles of panels and 2080 doubles of packed update.  Variants:
//   0: rhs-major stack (ustack[r][blkval], 20 MB between right-hand sides), linear coalesced stores
//   1: the same bytes in a family-major layout ([family][rhs][entries]: a workgroup streams one contiguous region)
//   2: rhs-major, every lane of a store in a different line (worst case)
//   3: rhs-major, the stores of k_fam_sparse: children's panels linear (8 x 180), the parent's Q tiles and the packed
//      update tiles straight from the MFMA accumulator layout (four 16-lane = 128 B pieces per store instruction)
// build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int NF = 896, NR = 100, PAN = 2625, UPD = 2080, GY = 2;
__global__ void __launch_bounds__(512) k_store(double* u, double* up, int64_t ldu, int64_t ldup, int variant, int nwaves) {
  const int f = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  if (wave >= nwaves) return;
  const int nthr = nwaves * 64;
  for (int r = blockIdx.y; r < NR; r += GY) {
    double* P; double* U;
    if (variant == 1) { P = u + ((int64_t)f * NR + r) * PAN; U = up + ((int64_t)f * NR + r) * UPD; }
    else { P = u + (int64_t)r * ldu + (int64_t)f * PAN; U = up + (int64_t)r * ldup + (int64_t)f * UPD; }
    if (variant == 2) {
      const int l15 = lane & 15, kq = lane >> 4;
      for (int t = wave; t < PAN / 64 + 1; t += nwaves)
        for (int rr = 0; rr < 4; ++rr) { const int e = t * 64 + (kq + 4 * rr) * 16 + l15; if (e < PAN) P[(e * 79) % PAN] = 1.0; }
      for (int t = wave; t < UPD / 64 + 1; t += nwaves)
        for (int rr = 0; rr < 4; ++rr) { const int e = t * 64 + (kq + 4 * rr) * 16 + l15; if (e < UPD) U[e] = 1.0; }
    } else if (variant == 3) {
      const int l15 = lane & 15, kq = lane >> 4, gw = wave & 3, nn = 15, na = 64, nf = 79;
      if ((wave >> 2) == ((r / GY) & 1)) {      // two groups alternate right-hand sides
        for (int c = gw; c < 8; c += 4) for (int e = lane; e < 180; e += 64) P[c * 180 + e] = 1.0;
        double* PP = P + 1440;
        const int m = 16 * gw + l15;
        for (int rr = 0; rr < 4; ++rr) { const int n = kq + 4 * rr; if (n < nn) PP[(nn + m) + n * nf] = 1.0; }
        if (gw == 0) for (int rr = 0; rr < 4; ++rr) { const int n = kq + 4 * rr; if (l15 < nn && n <= l15) PP[l15 + n * nf] = 1.0; }
        for (int tn = 0; tn <= gw; ++tn)
          for (int rr = 0; rr < 4; ++rr) { const int n = 16 * tn + kq + 4 * rr; if (m >= n) U[n * na - ((n * (n - 1)) >> 1) + (m - n)] = 1.0; }
      }
    } else {
      for (int e = tid; e < PAN; e += nthr) P[e] = 1.0;
      for (int e = tid; e < UPD; e += nthr) U[e] = 1.0;
    }
  }
}
int main() {
  const int64_t ldu = 2493568, ldup = (int64_t)NF * UPD + 100000;
  double *u, *up;
  hipMalloc(&u, sizeof(double) * ldu * NR);
  hipMalloc(&up, sizeof(double) * ldup * NR);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = 8.0 * NF * NR * (PAN + UPD) / 1e9;
  for (int nw : {8, 4})
    for (int v = 0; v < 4; ++v) {
      float best = 1e9;
      for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_store, dim3(NF, GY), dim3(512), 0, 0, u, up, ldu, ldup, v, nw);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("waves %d variant %d: %.3f ms  %.2f GB -> %.2f TB/s\n", nw, v, best, gb, gb / best);
    }
  return 0;
}
