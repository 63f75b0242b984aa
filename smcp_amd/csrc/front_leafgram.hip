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
//
// Two kernels per Schur complement:
//   k_leaf_tables  one wave per clique, little LDS, many resident waves: Psi and Omega of every listed clique from the
//                  inverse-form factor and the Y_AA block -> global table (one fixed-size record per clique);
//   k_leaf_pairs   one workgroup per CU, a wave per clique at a time: the record and the clique's entries over ALL
//                  constraints (a static array in constraint order, built by kkt_set_constraints) are PREFETCHED into
//                  registers while the pairs of the previous clique are formed, then copied to LDS; the pairs (e >= f)
//                  are dealt over the lanes, ds_add_f64 into the workgroup's packed lower triangle of H, which is
//                  written out once per workgroup.
// k_gram_reduce adds these partial triangles to the partial tiles of the Gram kernel in a fixed order.
#include <hip/hip_runtime.h>

namespace smcp {

struct LeafGramArgs {
  const CliqueDesc* cl;
  const int32_t* list;      // the cliques (childless, nn <= 16, na <= 32) ...
  const int32_t* slot;      // ... and their index among ALL family children (records and entry lists are indexed by it)
  int cnt;
  const double* LK;         // inverse-form factor (blkval layout)
  const double* yaa;        // Y_AA blocks, lower triangles, update layout
  double* tab;              // records: Psi (nf x nf, ld nf) then the Omega panel (nf x nn), rec doubles each
  int rec;
  const int32_t* eptr;      // family child g -> first entry (lg_children + 1)
  const int32_t* epk;       // row | column << 8 | constraint << 16
  const double* ew;         // value (halved on the diagonal of the supernode block)
  const int32_t* remap;     // constraint -> row of H, -1: not part of this Gram block (null: identity)
  int nr;                   // order of H
  int ecap;                 // entries of one clique over all constraints (host-checked)
  double* part;             // gridDim.x slots of partial tiles (the layout of k_gram_diag128's partials)
  int* info;
  int skip;                 // timing studies only (SMCP_LGSKIP): 1 = no atomics, 2 = no table look-ups, 4 = no pairs at all
};

__device__ inline void lg_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// doubles of LDS of k_leaf_tables: Psi | Omega panel | K | Li
__host__ __device__ inline int leaftab_doubles(int nfmax, int nnmax, int namax) {
  return nfmax * nfmax + nfmax * nnmax + namax * nnmax + nnmax * nnmax;
}

__global__ void __launch_bounds__(64) k_leaf_tables(LeafGramArgs a, int nfmax, int nnmax, int namax) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int x = blockIdx.x;
  const int k = a.list[x];
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  double* const sPsi = smem;
  double* const sOm = sPsi + nfmax * nfmax;
  double* const sK = sOm + nfmax * nnmax;
  double* const sLi = sK + namax * nnmax;
  const double* const lk = a.LK + d.blk;
  const double* const ya = a.yaa + d.upd;
  for (int e = lane; e < nf * nn; e += 64) sOm[e] = 0.0;
  // (eight loads in flight per lane: as plain copy loops these compiled to one load, s_waitcnt vmcnt(0), one LDS store per
  // trip -- fifteen dependent round trips for the 31 x 31 block of a synth50k leaf)
  batched_loop<4>(lane, na * nn, 64, [=](int e) { const int r = e % na, c = e / na; return lk[(nn + r) + (int64_t)c * nf]; },
                  [=](int e, double v) { sK[e] = v; });
  for (int e = lane; e < nn * nn; e += 64) { const int i = e % nn, j = e / nn; sLi[e] = i >= j ? lk[i + (int64_t)j * nf] : 0.0; }
  batched_loop<8>(lane, na * na, 64, [=](int e) { const int i = e % na, j = e / na; return ya[max(i, j) + (int64_t)min(i, j) * na]; },
                  [=](int e, double v) { const int i = e % na, j = e / na; sPsi[(nn + i) + (nn + j) * nf] = v; });
  lg_wave_sync();
  // Psi_AN = -(Y_AA K) (both triangles of Psi are kept), Omega_NN = Li^T Li
  for (int e = lane; e < na * ((nn + 3) / 4); e += 64) {     // row r, four columns at a time: K is read by broadcast
    const int r = e % na, c0 = 4 * (e / na);
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < na; ++q) {
      const double y = sPsi[(nn + r) + (nn + q) * nf];
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] += y * sK[q + min(c0 + u, nn - 1) * na];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c0 + u < nn) { sPsi[(nn + r) + (c0 + u) * nf] = -s4[u]; sPsi[(c0 + u) + (nn + r) * nf] = -s4[u]; }
  }
  for (int e = lane; e < nn * nn; e += 64) {
    const int i = e % nn, j = e / nn;
    double s = 0.0;
    for (int t = max(i, j); t < nn; ++t) s += sLi[t + i * nn] * sLi[t + j * nn];
    sOm[i + j * nf] = s;
  }
  lg_wave_sync();
  // Psi_NN = K^T Y_AA K = -K^T Psi_AN
  for (int e = lane; e < nn * nn; e += 64) {
    const int i = e % nn, c = e / nn;
    if (i < c) continue;
    double s = 0.0;
    for (int r = 0; r < na; ++r) s -= sK[r + i * na] * sPsi[(nn + r) + c * nf];
    sPsi[i + c * nf] = s;
    sPsi[c + i * nf] = s;
  }
  lg_wave_sync();
  double* const out = a.tab + (int64_t)a.slot[x] * a.rec;
  for (int e = lane; e < nf * nf; e += 64) out[e] = sPsi[e];
  for (int e = lane; e < nf * nn; e += 64) out[nf * nf + e] = sOm[e];
}

// per-wave doubles of LDS of k_leaf_pairs: the record | entry values | entry words
__host__ __device__ inline int leafgram_wave_doubles(int rec, int ecap) { return rec + ecap + (ecap + 1) / 2; }

// RT = ceil(rec / 64) and ET = ceil(ecap / 64): register images of the prefetched record and entry list
template <int RT, int ET>
__global__ void __launch_bounds__(512) k_leaf_pairs(LeafGramArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)blockDim.x >> 6;
  const int np = a.nr * (a.nr + 1) / 2;
  double* const sH = smem;
  const int wd = leafgram_wave_doubles(a.rec, a.ecap);
  double* const sT = smem + ((np + 1) & ~1) + wave * wd;
  double* const sW = sT + a.rec;
  int* const sPk = reinterpret_cast<int*>(sW + a.ecap);
  for (int e = tid; e < np; e += (int)blockDim.x) sH[e] = 0.0;
  __syncthreads();

  double rt[RT], rw[ET];
  int rp[ET];
  int nxt_E = 0, nxt_nn = 1, nxt_nf = 1;
  auto prefetch = [&](int x) {           // record and entries of list position x -> registers
    if (x >= a.cnt) return;
    const CliqueDesc d = a.cl[a.list[x]];
    nxt_nn = d.nn; nxt_nf = d.nn + d.na;
    const int g = a.slot[x];
    const int e0 = a.eptr[g];
    nxt_E = min(a.eptr[g + 1] - e0, a.ecap);
    const double* const tb = a.tab + (int64_t)g * a.rec;
#pragma unroll
    for (int i = 0; i < RT; ++i) { const int e = lane + 64 * i; rt[i] = e < a.rec ? tb[e] : 0.0; }
#pragma unroll
    for (int i = 0; i < ET; ++i) {
      const int e = lane + 64 * i;
      rp[i] = e < nxt_E ? a.epk[e0 + e] : 0;
      rw[i] = e < nxt_E ? a.ew[e0 + e] : 0.0;
    }
  };
  const int stride = (int)gridDim.x * nw;
  int x = (int)blockIdx.x * nw + wave;
  prefetch(x);
  for (; x < a.cnt; x += stride) {
    const int E = nxt_E, nn = nxt_nn, nf = nxt_nf;
#pragma unroll
    for (int i = 0; i < RT; ++i) { const int e = lane + 64 * i; if (e < a.rec) sT[e] = rt[i]; }
#pragma unroll
    for (int i = 0; i < ET; ++i) {
      const int e = lane + 64 * i;
      if (e < E) {
        int pk = rp[i];
        double w = rw[i];
        if (a.remap) {                   // a Gram block over a subset of the constraints
          const int cm = a.remap[pk >> 16];
          if (cm < 0) w = 0.0;
          pk = (pk & 0xffff) | (max(cm, 0) << 16);
        }
        sPk[e] = pk;
        sW[e] = w;
      }
    }
    prefetch(x + stride);                // in flight while the pairs below are formed
    lg_wave_sync();
    const double* const sPsi = sT;
    const double* const sOm = sT + nf * nf;
    // ---- pairs (e >= f), flattened: t = e (e + 1) / 2 + f; four pairs per lane and step, so that the LDS round trips
    // of the four dependent chains (entry words -> table addresses -> products -> atomic) overlap (two waves per SIMD
    // cannot hide them: one pair per step ran at ~18 cycles per instruction)
    const int npair = (a.skip & 4) ? 0 : E * (E + 1) / 2;
    for (int t0 = 0; t0 < npair; t0 += 256) {
      int pe[4], pf[4];
      double we[4], wf[4];
      bool on[4], dg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + lane + 64 * u;
        on[u] = t < npair;
        const int tt = on[u] ? t : 0;
        int e = (int)((__fsqrt_rn(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
        if (e * (e + 1) / 2 > tt) --e;
        else if ((e + 1) * (e + 2) / 2 <= tt) ++e;
        const int f = tt - e * (e + 1) / 2;
        dg[u] = e == f;
        pe[u] = sPk[e]; pf[u] = sPk[f];
        we[u] = sW[e]; wf[u] = sW[f];
      }
      double val[4];
      int hidx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = pe[u] & 0xff, q = (pe[u] >> 8) & 0xff, ce = pe[u] >> 16;
        const int r = pf[u] & 0xff, s = (pf[u] >> 8) & 0xff, cf = pf[u] >> 16;
        double term;
        if (a.skip & 2) term = (double)(p + q + r + s);
        else {
          const double om_qr = sOm[r + q * nf], om_ps = sOm[p + s * nf], om_qs = sOm[q + s * nf];
          const double om_pr = sOm[p + min(r, nn - 1) * nf];
          const double ps_ps = sPsi[p + s * nf], ps_qr = sPsi[r + q * nf], ps_pr = sPsi[r + p * nf], ps_qs = sPsi[q + s * nf];
          const double opr = r < nn ? om_pr : 0.0;
          term = om_qr * (ps_ps + om_ps) + ps_qr * om_ps + om_qs * (ps_pr + opr) + ps_qs * opr;
        }
        const double mult = dg[u] ? 2.0 : (ce == cf ? 4.0 : 2.0);
        const int hi = max(ce, cf), lo = min(ce, cf);      // (a remapped subset need not be in ascending order)
        val[u] = mult * we[u] * wf[u] * term;
        hidx[u] = hi * (hi + 1) / 2 + lo;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (on[u]) { if (a.skip & 1) { if (val[u] == 1.2345e-300) sH[0] = 1.0; } else unsafeAtomicAdd(&sH[hidx[u]], val[u]); }
    }
    lg_wave_sync();
  }
  __syncthreads();
  // write-out in the layout of the Gram kernel's partial tiles (slot = 64 tiles of 16 x 16, tile (tm, tn <= tm) at
  // tm + tn * mti, element (row, col) at col * 16 + row): k_gram_reduce then sums both kinds of partials alike
  double* const out = a.part + (int64_t)blockIdx.x * (64 * 256);
  const int mti = (a.nr + 15) >> 4;
  for (int e = tid; e < mti * mti * 256; e += (int)blockDim.x) {
    const int t = e >> 8, idx = e & 255;
    const int tm = t % mti, tn = t / mti;
    if (tm < tn) continue;
    const int i = tm * 16 + (idx & 15), j = tn * 16 + (idx >> 4);
    const int hi = max(i, j), lo = min(i, j);
    out[e] = hi < a.nr ? sH[hi * (hi + 1) / 2 + lo] : 0.0;
  }
}

}  // namespace smcp
