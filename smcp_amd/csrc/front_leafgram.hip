// Closed-form Gram contribution of the childless small cliques (the children of the families of front_fam2.hip) to the
// Schur complement H = G(A)^T W G(A) (Gram formulation, solvers.py:414-420 / 479-487).
//
// The swept panel of such a clique is a sum of a few closed-form terms, one per entry of the constraint (front_fam2.hip,
// fact 1).  Its block of the Gram matrix is therefore a bilinear form in the constraint ENTRIES and needs neither the
// panels in HBM (1.03 GB written by the family kernel and read again by the Gram kernel on synth50k) nor the matrix
// pipe.  With the front-local matrices (rows / columns: supernode first, then separator)
//     Omega = [Li^T Li, 0; 0, 0],   Psi = [K^T Y_AA K, -K^T Y_AA; -Y_AA K, Y_AA],   Y_front = Omega + Psi,
// the panel of a front matrix F is (G_NN, R^T G_AN) = rows of [Li 0; -K I] F [Li 0]^T, and for two front matrices
//     <G(F), G(F')>_W = tr(F Omega F' Omega) + 2 tr(F Psi F' Omega).
// For two symmetric entries F = w (e_p e_q^T + e_q e_p^T), F' = w' (e_r e_s^T + e_s e_r^T) (q, s < nn: the columns of
// the panel; w = v / 2 on the diagonal) this is
//     2 w w' [ Omega_qr Y_ps + Psi_qr Omega_ps + Omega_qs Y_pr + Psi_qs Omega_pr ]
// -- eight table look-ups, no cancellation (the same products the panel route forms, summed in another order).
// One wave per clique: the tables of the clique in LDS, the entries of all constraints staged in LDS in constraint
// order, the pairs (e >= f) dealt over the lanes, ds_add_f64 into the workgroup's packed lower triangle of H, which is
// written out once per workgroup; k_gram_reduce adds these partial triangles to the partial tiles of the Gram kernel
// in a fixed order.
#include <hip/hip_runtime.h>

namespace smcp {

struct LeafGramArgs {
  const CliqueDesc* cl;
  const int32_t* list;      // the cliques (childless, nn <= 16, na <= 32)
  int cnt;
  const double* LK;         // inverse-form factor (blkval layout)
  const double* yaa;        // Y_AA blocks, lower triangles, update layout
  const int32_t* kc_ptr; const int32_t* kc_ij; const double* kc_val; const int32_t* ids;
  int kc_stride;            // m + 1
  int nr;                   // constraints of the sweep: order of H
  int nfmax, nnmax, namax;  // LDS sizing over the list
  int ecap;                 // entries of one clique over all constraints (host-checked)
  double* part;             // gridDim.x packed lower triangles (row-major: (i, j <= i) at i (i + 1) / 2 + j)
  int* info;
};

// per-wave doubles of LDS: Psi (nf x nf) | Omega panel (nf x nn) | K (na x nn) | Li (nn x nn) | entry values | entry words
__host__ __device__ inline int leafgram_wave_doubles(int nfmax, int nnmax, int namax, int ecap) {
  return nfmax * nfmax + nfmax * nnmax + namax * nnmax + nnmax * nnmax + ecap + (ecap + 1) / 2;
}

__device__ inline void lg_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(512) k_leaf_gram(LeafGramArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)blockDim.x >> 6;
  const int np = a.nr * (a.nr + 1) / 2;
  double* const sH = smem;
  const int wd = leafgram_wave_doubles(a.nfmax, a.nnmax, a.namax, a.ecap);
  double* const wb = smem + ((np + 1) & ~1) + wave * wd;
  double* const sPsi = wb;
  double* const sOm = sPsi + a.nfmax * a.nfmax;
  double* const sK = sOm + a.nfmax * a.nnmax;
  double* const sLi = sK + a.namax * a.nnmax;
  double* const sW = sLi + a.nnmax * a.nnmax;
  int* const sPk = reinterpret_cast<int*>(sW + a.ecap);
  for (int e = tid; e < np; e += (int)blockDim.x) sH[e] = 0.0;
  __syncthreads();

  for (int x = (int)blockIdx.x * nw + wave; x < a.cnt; x += (int)gridDim.x * nw) {
    const int k = a.list[x];
    const CliqueDesc d = a.cl[k];
    const int nn = d.nn, na = d.na, nf = nn + na;
    const double* const lk = a.LK + d.blk;
    const double* const ya = a.yaa + d.upd;
    // ---- tables
    for (int e = lane; e < nf * nn; e += 64) sOm[e] = 0.0;
    for (int e = lane; e < na * nn; e += 64) { const int r = e % na, c = e / na; sK[e] = lk[(nn + r) + (int64_t)c * nf]; }
    for (int e = lane; e < nn * nn; e += 64) { const int i = e % nn, j = e / nn; sLi[e] = i >= j ? lk[i + (int64_t)j * nf] : 0.0; }
    for (int e = lane; e < na * na; e += 64) {
      const int i = e % na, j = e / na;
      sPsi[(nn + i) + (nn + j) * nf] = ya[max(i, j) + (int64_t)min(i, j) * na];
    }
    // ---- entries of all constraints, in constraint order: word = row | column << 8 | constraint << 16
    const int32_t* const kp = a.kc_ptr + (int64_t)k * a.kc_stride;
    int E = 0;
    for (int c0 = 0; c0 < a.nr; c0 += 64) {
      const int r = c0 + lane;
      int beg = 0, cn = 0;
      if (r < a.nr) { const int j = a.ids ? a.ids[r] : r; beg = kp[j]; cn = kp[j + 1] - beg; }
      int incl = cn;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
      const int off = E + incl - cn;
      for (int t = 0; t < cn; ++t)
        if (off + t < a.ecap) {
          const int ij = a.kc_ij[beg + t];
          const int i = ij & 0xffff, jc = ij >> 16;
          const double v = a.kc_val[beg + t];
          sPk[off + t] = i | (jc << 8) | (r << 16);
          sW[off + t] = i == jc ? 0.5 * v : v;
        }
      E += __shfl(incl, 63);
    }
    if (E > a.ecap) { if (lane == 0) atomicCAS(a.info, 0, -8); E = a.ecap; }     // host sizing error: shows as a failed solve
    lg_wave_sync();
    // ---- Psi_AN = -(Y_AA K) (both triangles of Psi are kept), Omega_NN = Li^T Li
    for (int e = lane; e < na * nn; e += 64) {
      const int r = e % na, c = e / na;
      double s = 0.0;
      for (int q = 0; q < na; ++q) s += sPsi[(nn + r) + (nn + q) * nf] * sK[q + c * na];
      sPsi[(nn + r) + c * nf] = -s;
      sPsi[c + (nn + r) * nf] = -s;
    }
    for (int e = lane; e < nn * nn; e += 64) {
      const int i = e % nn, j = e / nn;
      double s = 0.0;
      for (int t = max(i, j); t < nn; ++t) s += sLi[t + i * nn] * sLi[t + j * nn];
      sOm[i + j * nf] = s;
    }
    lg_wave_sync();
    // ---- Psi_NN = K^T Y_AA K = -K^T Psi_AN
    for (int e = lane; e < nn * nn; e += 64) {
      const int i = e % nn, c = e / nn;
      if (i < c) continue;
      double s = 0.0;
      for (int r = 0; r < na; ++r) s -= sK[r + i * na] * sPsi[(nn + r) + c * nf];
      sPsi[i + c * nf] = s;
      sPsi[c + i * nf] = s;
    }
    lg_wave_sync();
    // ---- pairs (e >= f), flattened: t = e (e + 1) / 2 + f
    const int npair = E * (E + 1) / 2;
    for (int t0 = 0; t0 < npair; t0 += 64) {
      const int t = t0 + lane;
      const bool on = t < npair;
      const int tt = on ? t : 0;
      int e = (int)((__fsqrt_rn(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
      if (e * (e + 1) / 2 > tt) --e;
      else if ((e + 1) * (e + 2) / 2 <= tt) ++e;
      const int f = tt - e * (e + 1) / 2;
      const int pe = sPk[e], pf = sPk[f];
      const double we = sW[e], wf = sW[f];
      const int p = pe & 0xff, q = (pe >> 8) & 0xff, ce = pe >> 16;
      const int r = pf & 0xff, s = (pf >> 8) & 0xff, cf = pf >> 16;
      const double om_qr = sOm[r + q * nf], om_ps = sOm[p + s * nf], om_qs = sOm[q + s * nf];
      const double om_pr = sOm[p + min(r, nn - 1) * nf];
      const double ps_ps = sPsi[p + s * nf], ps_qr = sPsi[r + q * nf], ps_pr = sPsi[p + r * nf], ps_qs = sPsi[q + s * nf];
      const double opr = r < nn ? om_pr : 0.0;
      const double term = om_qr * (ps_ps + om_ps) + ps_qr * om_ps + om_qs * (ps_pr + opr) + ps_qs * opr;
      const double mult = e == f ? 2.0 : (ce == cf ? 4.0 : 2.0);
      if (on) unsafeAtomicAdd(&sH[ce * (ce + 1) / 2 + cf], mult * we * wf * term);
    }
    lg_wave_sync();
  }
  __syncthreads();
  double* const out = a.part + (int64_t)blockIdx.x * np;
  for (int e = tid; e < np; e += (int)blockDim.x) out[e] = sH[e];
}

}  // namespace smcp
