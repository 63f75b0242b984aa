// MFMA paths of the remaining tree operations the interior-point drivers call every iteration:
// inverse Hessian factors (hessian(..., inv=True): kkt_res, Newton decrements, esd updates) and the
// maximum-determinant completion (line searches in primal scaling).  Same structure as the sweeps in
// front_mfma.hip: products on v_mfma_f64_16x16x4 through wg_mma(), working set in LDS when it fits
// (LDS = true) or in HBM/L2 scratch (LDS = false), formulas of SURVEY.md App. A.4 / A.5:
//   G^-adj : Q = Z_AN L_NN + Z_AA L_AN ;  G_NN = L_NN^T Z_NN L_NN + L_AN^T Q'' + Q''^T L_AN,  Q'' = Z_AN L_NN + Z_AA L_AN / 2
//   G^-1   : V = G_AN + L_AN G_NN / 2 ;  Upd = children - V L_AN^T - L_AN V^T ;  F_AN = (2V - G_AN) L_NN^T ;
//            F_NN = L_NN G_NN L_NN^T ;  panel = F - children
//   Y_AA^-1 is applied through the explicit inverse Ri of its Cholesky factor (cached next to it).
#include <hip/hip_runtime.h>

namespace smcp {

// blocked in-place Cholesky of the n x n lower matrix A (LDS or HBM); D16: 256 doubles of LDS.
__device__ inline int wg_potrf_blocked(int n, double* A, int64_t lda, double* D16) {
  for (int jb = 0; jb < n; jb += 16) {
    const int bw = min(16, n - jb);
    int f = potrf_inv16(A + jb + jb * lda, (int)lda, bw, D16);
    if (f) return jb + f;
    const int mrem = n - jb - bw;
    if (mrem > 0) {
      double* Pj = A + (jb + bw) + jb * lda;
      wg_mma(mrem, bw, bw, [=](int m, int kk) { return Pj[m + kk * lda]; },
             [=](int kk, int nn_) { return D16[nn_ + kk * 16]; },
             [=](int m, int nn_, double acc) { Pj[m + nn_ * lda] = acc; });
      __syncthreads();
      double* Tr = A + (jb + bw) + (jb + bw) * lda;
      wg_mma(mrem, mrem, bw, [=](int m, int kk) { return Pj[m + kk * lda]; },
             [=](int kk, int nn_) { return Pj[nn_ + kk * lda]; },
             [=](int m, int nn_, double acc) { if (m >= nn_) Tr[m + nn_ * lda] -= acc; }, true);
      __syncthreads();
    }
  }
  return 0;
}
// Li (n x n, ldi, zeros above the diagonal) <- inverse of the lower-triangular L (ldl); D16: 256, S: 16 x 128 doubles of LDS
__device__ inline void wg_tri_inverse(int n, const double* L, int64_t ldl, double* Li, int64_t ldi, double* D16, double* S) {
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    int i = e % n, j = e / n;
    if (i < j) Li[i + j * ldi] = 0.0;
  }
  for (int ib = 0; ib < n; ib += 16) {
    const int bw = min(16, n - ib);
    tri_inv16(L + ib + ib * ldl, (int)ldl, bw, D16);
    for (int e = threadIdx.x; e < bw * bw; e += blockDim.x) {
      int i = e % bw, j = e / bw;
      if (i >= j) Li[(ib + i) + (ib + j) * ldi] = D16[i + j * 16];
    }
    for (int c0 = 0; c0 < ib; c0 += 128) {
      const int cw = min(128, ib - c0);
      __syncthreads();
      wg_mma(bw, cw, ib - c0, [=](int m, int kk) { return L[(ib + m) + (c0 + kk) * ldl]; },
             [=](int kk, int nn_) { return kk >= nn_ ? Li[(c0 + kk) + (c0 + nn_) * ldi] : 0.0; },
             [=](int m, int nn_, double acc) { S[m + nn_ * 16] = acc; });
      __syncthreads();
      wg_mma(bw, cw, bw, [=](int m, int kk) { return D16[m + kk * 16]; },
             [=](int kk, int nn_) { return S[kk + nn_ * 16]; },
             [=](int m, int nn_, double acc) { Li[(ib + m) + (c0 + nn_) * ldi] = -acc; });
    }
    __syncthreads();
  }
}

// faci[k] <- inverse of the lower-triangular fac[k] (update-matrix layout), one workgroup per clique
// Ri = R^-1 of the cached Cholesky factor of Y_AA for separators of at most LDM - 1 rows: triangle in LDS, the 16 x 16
// diagonal blocks inverted side by side and the rest by recursive doubling (tri_inv64_rd), where k_factor_inverse below
// walks block rows against HBM (0.64 ms per call on the 8064 small fronts of synth50k, once per iteration of the
// interior-point drivers; this one: see DESIGN.md section 4).  LDM = 65: 256 threads, dynamic LDS (2 * 65 * 64 + 1024) doubles;
// LDM = 33: 128 threads (two diagonal blocks, one product), (2 * 33 * 32 + 256) doubles.
template <int LDM>
__global__ void __launch_bounds__(256) k_factor_inverse_lds(MfmaArgs a, const double* fac, double* faci) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NB = LDM - 1;
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int na = d.na;
  if (!na) return;
  double* const D = smem;
  double* const Di = smem + NB * LDM;
  double* const s16 = Di + NB * LDM;
  const double* src = fac + d.upd;
  double* dst = faci + d.upd;
  for (int e = threadIdx.x; e < NB * LDM; e += blockDim.x) Di[e] = 0.0;
  batched_loop<8>(threadIdx.x, na * na, blockDim.x, [=](int e) { return (e % na) >= (e / na) ? src[e] : 0.0; },
                  [=](int e, double v) { D[(e % na) + (e / na) * LDM] = v; });
  __syncthreads();
  tri_inv64_rd<LDM>(D, na, Di, s16);
  for (int e = threadIdx.x; e < na * na; e += blockDim.x) dst[e] = Di[(e % na) + (e / na) * LDM];
}

__global__ void __launch_bounds__(256) k_factor_inverse(TreeArgs t, const double* fac, double* faci) {
  __shared__ double D16[256], S[16 * 128];
  const int k = t.lev[blockIdx.x];
  const CliqueDesc d = t.cl[k];
  if (!d.na) return;
  wg_tri_inverse(d.na, fac + d.upd, d.na, faci + d.upd, d.na, D16, S);
}

// ---------------------------------------------------------------- G^-adj (and the Y_AA^-1 / R^-1 scaling after it)
// a.LK = the factor L itself (blkval layout), a.ysc = faci (R^-1), a.ymode: 0 none, 1 apply Y_AA^-1 = Ri^T Ri, 2 apply Ri
template <bool LDS>
__global__ void k_hess_down_inv_mfma(MfmaArgs a, double* u, int64_t ldu) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  Work w = make_work<LDS>(a, d, smem, k, blockIdx.y);
  if (LDS) load_consts(a, d, w, a.ymode != 0);
  const int ymode = a.ymode;
  __syncthreads();
  for (int r = blockIdx.y; r < a.nrhs; r += gridDim.y) {
    double* ur = u + (int64_t)r * ldu;
    double* P = ur + d.blk;
    // Z_AA of every clique was gathered from the input (all cliques, before any is overwritten) into upd
    double* UkG = a.t.upd + (int64_t)r * a.t.updlen + d.upd;
    if (LDS) {
      double* Fl = w.F; const int ldfl = w.ldf;
      batched_loop<8>(threadIdx.x, nf * nn, blockDim.x, [=](int e) { return P[e]; },
                      [=](int e, double v) { Fl[(e % nf) + (e / nf) * ldfl] = v; });
      double* Ul = w.U; const int ldul = w.ldu;
      batched_loop<8>(threadIdx.x, na * na, blockDim.x, [=](int e) { return (e % na) >= (e / na) ? UkG[e] : 0.0; },
                      [=](int e, double v) { Ul[(e % na) + (e / na) * ldul] = v; });
    } else {
      w.F = P;
      w.U = UkG;
    }
    __syncthreads();
    const Work v = w;
    // phase 1: E = Z_AN L_NN ; G = Z_AA L_AN ; T = Z_NN L_NN
    wg_mma(na, nn, nn, [=](int m, int kk) { return v.F[nn + m + kk * v.ldf]; },
           [=](int kk, int n) { return v.Li[kk + n * v.ldl]; },
           [=](int m, int n, double acc) { v.E[m + n * v.lde] = acc; });
    wg_mma(na, nn, na, [=](int m, int kk) { return m >= kk ? v.U[m + kk * v.ldu] : v.U[kk + m * v.ldu]; },
           [=](int kk, int n) { return v.K[kk + n * v.ldk]; },
           [=](int m, int n, double acc) { v.G[m + n * v.ldg] = acc; }, false, 2);
    wg_mma(nn, nn, nn, [=](int m, int kk) { return m >= kk ? v.F[m + kk * v.ldf] : v.F[kk + m * v.ldf]; },
           [=](int kk, int n) { return v.Li[kk + n * v.ldl]; },
           [=](int m, int n, double acc) { v.T[m + n * v.ldt] = acc; }, false, 5);
    __syncthreads();
    // phase 2: Q = E + G (into the panel) ; Q'' = E + G / 2 (into E)
    for (int e = threadIdx.x; e < na * nn; e += blockDim.x) {
      int i = e % na, j = e / na;
      const double ee = v.E[i + j * v.lde], gg = v.G[i + j * v.ldg];
      v.F[nn + i + j * v.ldf] = ee + gg;
      v.E[i + j * v.lde] = ee + 0.5 * gg;
    }
    __syncthreads();
    // phase 3: G_NN = L_NN^T T + L_AN^T Q'' + Q''^T L_AN (lower) ; G = Ri Q
    wg_mma(nn, nn, nn + 2 * na,
           [=](int m, int kk) {
             return kk < nn ? v.Li[kk + m * v.ldl] : (kk < nn + na ? v.K[(kk - nn) + m * v.ldk] : v.E[(kk - nn - na) + m * v.lde]);
           },
           [=](int kk, int n) {
             return kk < nn ? v.T[kk + n * v.ldt] : (kk < nn + na ? v.E[(kk - nn) + n * v.lde] : v.K[(kk - nn - na) + n * v.ldk]);
           },
           [=](int m, int n, double acc) { if (m >= n) v.F[m + n * v.ldf] = acc; });
    if (ymode) {
      wg_mma(na, nn, na, [=](int m, int kk) { return m >= kk ? v.Y[m + kk * v.ldy] : 0.0; },
             [=](int kk, int n) { return v.F[nn + kk + n * v.ldf]; },
             [=](int m, int n, double acc) { v.G[m + n * v.ldg] = acc; }, false, 3);
      __syncthreads();
      // phase 4: G_AN = Ri^T G (ymode 1) or G (ymode 2)
      if (ymode == 1)
        wg_mma(na, nn, na, [=](int m, int kk) { return kk >= m ? v.Y[kk + m * v.ldy] : 0.0; },
               [=](int kk, int n) { return v.G[kk + n * v.ldg]; },
               [=](int m, int n, double acc) { v.F[nn + m + n * v.ldf] = acc; });
      else
        for (int e = threadIdx.x; e < na * nn; e += blockDim.x) v.F[nn + (e % na) + (e / na) * v.ldf] = v.G[(e % na) + (e / na) * v.ldg];
    }
    __syncthreads();
    if (LDS) {
      for (int e = threadIdx.x; e < nf * nn; e += blockDim.x) {
        int i = e % nf, j = e / nf;
        if (i >= j) P[e] = v.F[i + j * v.ldf];
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------- G^-1 (leaves -> root), optionally preceded by G_AN = Ri^T Ghat_AN (ymode 3)
template <bool LDS>
__global__ void k_hess_up_inv_mfma(MfmaArgs a, double* u, int64_t ldu) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  Work w = make_work<LDS>(a, d, smem, k, blockIdx.y);
  if (LDS) load_consts(a, d, w, a.ymode != 0);
  const int ymode = a.ymode;
  for (int r = blockIdx.y; r < a.nrhs; r += gridDim.y) {
    double* P = u + (int64_t)r * ldu + d.blk;
    const double* ubp = a.t.updp + (int64_t)r * a.t.updplen;
    double* UkG = a.t.upd + (int64_t)r * a.t.updlen + d.upd;
    double* UkP = a.t.updp + (int64_t)r * a.t.updplen + d.updp;
    if (LDS) {
      double* Fl = w.F; const int ldfl = w.ldf;
      batched_loop<8>(threadIdx.x, nf * nn, blockDim.x, [=](int e) { return P[e]; },
                      [=](int e, double v) { Fl[(e % nf) + (e / nf) * ldfl] = v; });
    } else {
      w.F = P;
      w.U = UkG;
    }
    __syncthreads();
    const Work v = w;
    if (ymode == 3) {
      wg_mma(na, nn, na, [=](int m, int kk) { return kk >= m ? v.Y[kk + m * v.ldy] : 0.0; },
             [=](int kk, int n) { return v.F[nn + kk + n * v.ldf]; },
             [=](int m, int n, double acc) { v.G[m + n * v.ldg] = acc; });
      __syncthreads();
      for (int e = threadIdx.x; e < na * nn; e += blockDim.x) v.F[nn + (e % na) + (e / na) * v.ldf] = v.G[(e % na) + (e / na) * v.ldg];
      __syncthreads();
    }
    auto gsym = [=](int i, int j) { return i >= j ? v.F[i + j * v.ldf] : v.F[j + i * v.ldf]; };
    // phase 1: V = G_AN + L_AN G_NN / 2 (into E) ; T = G_NN L_NN^T
    wg_mma(na, nn, nn, [=](int m, int kk) { return v.K[m + kk * v.ldk]; }, [=](int kk, int n) { return gsym(kk, n); },
           [=](int m, int n, double acc) { v.E[m + n * v.lde] = v.F[nn + m + n * v.ldf] + 0.5 * acc; });
    wg_mma(nn, nn, nn, [=](int m, int kk) { return gsym(m, kk); }, [=](int kk, int n) { return v.Li[n + kk * v.ldl]; },
           [=](int m, int n, double acc) { v.T[m + n * v.ldt] = acc; }, false, 4);
    __syncthreads();
    // phase 2: U = -(V L_AN^T + L_AN V^T) (lower) ; G = (2V - G_AN) L_NN^T
    wg_mma(na, na, 2 * nn,
           [=](int m, int kk) { return kk < nn ? v.E[m + kk * v.lde] : v.K[m + (kk - nn) * v.ldk]; },
           [=](int kk, int n) { return kk < nn ? v.K[n + kk * v.ldk] : v.E[n + (kk - nn) * v.lde]; },
           [=](int m, int n, double acc) { if (m >= n) v.U[m + n * v.ldu] = -acc; }, true);
    wg_mma(na, nn, nn, [=](int m, int kk) { return 2.0 * v.E[m + kk * v.lde] - v.F[nn + m + kk * v.ldf]; },
           [=](int kk, int n) { return v.Li[n + kk * v.ldl]; },
           [=](int m, int n, double acc) { v.G[m + n * v.ldg] = acc; }, false, 3);
    __syncthreads();
    // phase 3: F_NN = L_NN T (lower, into the panel) ; F_AN = G
    wg_mma(nn, nn, nn, [=](int m, int kk) { return v.Li[m + kk * v.ldl]; }, [=](int kk, int n) { return v.T[kk + n * v.ldt]; },
           [=](int m, int n, double acc) { if (m >= n) v.F[m + n * v.ldf] = acc; });
    for (int e = threadIdx.x; e < na * nn; e += blockDim.x) v.F[nn + (e % na) + (e / na) * v.ldf] = v.G[(e % na) + (e / na) * v.ldg];
    __syncthreads();
    // U = F - children on the panel,  G_AA = children - (...) on the update block
    add_children_front(a.t, d, ubp, v.F, v.ldf, v.U, v.ldu, -1.0, 1.0);
    if (LDS) {
      for (int e = threadIdx.x; e < nf * nn; e += blockDim.x) {
        int i = e % nf, j = e / nf;
        if (i >= j) P[e] = v.F[i + j * v.ldf];
      }
    }
    for (int e = threadIdx.x; e < na * na; e += blockDim.x) {
      int i = e % na, j = e / na;
      if (i >= j) UkP[pk_idx(i, j, na)] = v.U[i + j * v.ldu];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- completion (clique-local once chol(X_AA) and its inverse are cached)
// a.ysc = faci = R^-1 with R R^T = X_AA;  in: panel (X_NN, X_AN);  out: (L_NN, L_AN) with P_V((L L^T)^-1) = X
template <bool LDS>
__global__ void k_completion_mfma(MfmaArgs a, double* x) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // row-block scratch of the triangular inversion (supernodes of more than 16 columns only): LDS = true: behind the
  // compact working set, and only when the class has such supernodes (completion_lds_doubles); LDS = false: static
  __shared__ double Sstat[LDS ? 1 : 16 * 128];
  double* const S = LDS ? smem + mfma_lds_doubles_for(WK_COMPL, a.nnmax, a.namax) : Sstat;
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  Work w = make_work<LDS, WK_COMPL>(a, d, smem, k, 0);
  double* P = x + d.blk;
  double* D16 = w.D16;
  if (LDS) {
    load_consts(a, d, w, true);
    double* Fl = w.F; const int ldfl = w.ldf;
    batched_loop<8>(threadIdx.x, nf * nn, blockDim.x, [=](int e) { return P[e]; },
                    [=](int e, double v) { Fl[(e % nf) + (e / nf) * ldfl] = v; });
  } else {
    w.F = P;
  }
  __syncthreads();
  const Work v = w;
  // E = Ri X_AN ; G = Ri^T E  (= X_AA^-1 X_AN)
  wg_mma(na, nn, na, [=](int m, int kk) { return m >= kk ? v.Y[m + kk * v.ldy] : 0.0; },
         [=](int kk, int n) { return v.F[nn + kk + n * v.ldf]; },
         [=](int m, int n, double acc) { v.E[m + n * v.lde] = acc; });
  __syncthreads();
  wg_mma(na, nn, na, [=](int m, int kk) { return kk >= m ? v.Y[kk + m * v.ldy] : 0.0; },
         [=](int kk, int n) { return v.E[kk + n * v.lde]; },
         [=](int m, int n, double acc) { v.G[m + n * v.ldg] = acc; });
  __syncthreads();
  // T = reversed(Sigma), Sigma = X_NN - X_AN^T G
  wg_mma(nn, nn, na, [=](int m, int kk) { return v.F[nn + kk + m * v.ldf]; }, [=](int kk, int n) { return v.G[kk + n * v.ldg]; },
         [=](int m, int n, double acc) {
           const double f = m >= n ? v.F[m + n * v.ldf] : v.F[n + m * v.ldf];
           v.T[(nn - 1 - m) + (nn - 1 - n) * v.ldt] = f - acc;
         });
  __syncthreads();
  int f = wg_potrf_blocked(nn, v.T, v.ldt, D16);
  if (f) { if (threadIdx.x == 0) atomicCAS(info_of(a.t, k), 0, info_val(a.t, k)); return; }
  wg_tri_inverse(nn, v.T, v.ldt, v.Fnn, v.ldn, D16, S);       // Mi = M^-1 (lower)
  // L_NN[i][j] = Mi[nn-1-j][nn-1-i] (i >= j), zeros above
  for (int e = threadIdx.x; e < nn * nn; e += blockDim.x) {
    int i = e % nn, j = e / nn;
    v.F[i + j * v.ldf] = i >= j ? v.Fnn[(nn - 1 - j) + (nn - 1 - i) * v.ldn] : 0.0;
  }
  __syncthreads();
  // L_AN = -G L_NN
  wg_mma(na, nn, nn, [=](int m, int kk) { return v.G[m + kk * v.ldg]; },
         [=](int kk, int n) { return kk >= n ? v.F[kk + n * v.ldf] : 0.0; },
         [=](int m, int n, double acc) { v.E[m + n * v.lde] = -acc; });
  __syncthreads();
  for (int e = threadIdx.x; e < nf * nn; e += blockDim.x) {
    int i = e % nf, j = e / nf;
    if (i >= nn) P[e] = v.E[(i - nn) + j * v.lde];
    else if (LDS) { if (i >= j) P[e] = v.F[i + j * v.ldf]; }
    else if (i < j) P[e] = 0.0;
  }
}

// ---------------------------------------------------------------- llt: X = L L^T on V (leaves -> root)
// X_NN = L_NN L_NN^T + children ; X_AN = L_AN L_NN^T + children ; update = L_AN L_AN^T + children
template <bool LDS>
__global__ void k_llt_mfma(MfmaArgs a, double* x) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  Work w = make_work<LDS>(a, d, smem, k, 0);
  double* P = x + d.blk;
  double* UkG = a.t.upd + d.upd;
  double* UkP = a.t.updp + d.updp;
  if (LDS) {
    double* Fl = w.F; const int ldfl = w.ldf;
    batched_loop<8>(threadIdx.x, nf * nn, blockDim.x, [=](int e) { return P[e]; },
                    [=](int e, double v) { Fl[(e % nf) + (e / nf) * ldfl] = v; });
  } else {
    w.F = P;
    w.U = UkG;
  }
  __syncthreads();
  const Work v = w;
  auto lnnT = [=](int kk, int n) { return n >= kk ? v.F[n + kk * v.ldf] : 0.0; };     // (L_NN^T)[kk][n]
  wg_mma(nn, nn, nn, [=](int m, int kk) { return m >= kk ? v.F[m + kk * v.ldf] : 0.0; }, lnnT,
         [=](int m, int n, double acc) { if (m >= n) v.T[m + n * v.ldt] = acc; }, true);
  wg_mma(na, nn, nn, [=](int m, int kk) { return v.F[nn + m + kk * v.ldf]; }, lnnT,
         [=](int m, int n, double acc) { v.E[m + n * v.lde] = acc; }, false, 2);
  wg_mma(na, na, nn, [=](int m, int kk) { return v.F[nn + m + kk * v.ldf]; },
         [=](int kk, int n) { return v.F[nn + n + kk * v.ldf]; },
         [=](int m, int n, double acc) { if (m >= n) v.U[m + n * v.ldu] = acc; }, true, 5);
  __syncthreads();
  for (int e = threadIdx.x; e < nf * nn; e += blockDim.x) {
    const int i = e % nf, j = e / nf;
    if (i >= nn) v.F[i + j * v.ldf] = v.E[(i - nn) + j * v.lde];
    else if (i >= j) v.F[i + j * v.ldf] = v.T[i + j * v.ldt];
  }
  __syncthreads();
  add_children_front(a.t, d, a.t.updp, v.F, v.ldf, v.U, v.ldu, 1.0, 1.0);
  if (LDS) {
    for (int e = threadIdx.x; e < nf * nn; e += blockDim.x) {
      const int i = e % nf, j = e / nf;
      if (i >= j) P[e] = v.F[i + j * v.ldf];
    }
  }
  for (int e = threadIdx.x; e < na * na; e += blockDim.x) {
    const int i = e % na, j = e / na;
    if (i >= j) UkP[pk_idx(i, j, na)] = v.U[i + j * v.ldu];
  }
}

}  // namespace smcp
