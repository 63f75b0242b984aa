// Generic-size clique kernels: one workgroup per (clique, right-hand side), operands in HBM/L2.
//
// These implement every tree operation of the Newton-KKT path for ANY front size; the
// LDS-resident / MFMA specialisations in front_lds.hip take over where a front fits on chip.
// Math per clique follows SURVEY.md App. A (A.2 cholesky, A.3 projected_inverse,
// A.4 completion, A.5 hessian, A.7 llt/trsm/dot), i.e. the semantics of the CHOMPACK calls
// made at src/python/solvers.py:82-97.
#include <hip/hip_runtime.h>

#include "context.hpp"
#include "wgblas.hpp"

namespace smcp {

using namespace wg;

struct TreeArgs {
  const CliqueDesc* cl;
  const int32_t* relidx;
  const int32_t* chidx;
  const int32_t* lev;  // cliques of this launch
  int64_t updlen;      // stride between right-hand sides in the update workspace
  int64_t tmplen;      // stride between right-hand sides in the scratch workspace
  const int64_t* tmpptr;
  double* upd;
  double* updp;        // packed child->parent exchange buffer (fast kernels)
  int64_t updplen;
  double* tmp;
  int* info;           // failure flags: one per copy of the pattern (a replicated context factors ntrial matrices at once)
  int nsn1;            // cliques per copy (= nsn of an ordinary context): clique k belongs to copy k / nsn1
  // extend-add gather plan (null = not available)
  const int64_t* gp_tptr;
  const int32_t* gp_tgt;
  const int64_t* gp_cptr;
  const int32_t* gp_src;
};

// failure flag of the copy clique k belongs to, and the value a failure in clique k leaves there
__device__ inline int* info_of(const TreeArgs& t, int k) { return t.info + k / t.nsn1; }
__device__ inline int info_val(const TreeArgs& t, int k) { return k % t.nsn1 + 1; }

// ---- extend-add / gather -------------------------------------------------------------
__device__ inline void add_children(const TreeArgs& a, const CliqueDesc& d, const double* updbase,
                                    double* P, double* Uk, double sgn_panel, double sgn_upd) {
  const int nn = d.nn, nf = d.nn + d.na, na = d.na;
  for (int q = d.chbeg; q < d.chend; ++q) {
    const CliqueDesc c = a.cl[a.chidx[q]];
    const int nac = c.na;
    const int32_t* rel = a.relidx + c.rel;
    const double* Uc = updbase + c.upd;
    for (int e = SMCP_TID; e < nac * nac; e += SMCP_NT) {
      int i = e % nac, j = e / nac;
      if (i < j) continue;
      int ri = rel[i], rj = rel[j];
      double v = Uc[i + (int64_t)j * nac];
      if (rj < nn) P[ri + (int64_t)rj * nf] += sgn_panel * v;
      else Uk[(ri - nn) + (int64_t)(rj - nn) * na] += sgn_upd * v;
    }
    __syncthreads();
  }
}
// Uk <- front(parent)[rel, rel]  (lower), parent front = [Pp | Up]
__device__ inline void gather_sep(const TreeArgs& a, const CliqueDesc& d, const double* xbase,
                                  const double* updbase, double* Uk, int part = 0, int nparts = 1) {
  if (d.parent < 0 || d.na == 0) return;
  const CliqueDesc p = a.cl[d.parent];
  const int na = d.na, nnp = p.nn, nfp = p.nn + p.na, nap = p.na;
  const int32_t* rel = a.relidx + d.rel;
  const double* Pp = xbase + p.blk;
  const double* Up = updbase + p.upd;
  for (int e = part * SMCP_NT + SMCP_TID; e < na * na; e += nparts * SMCP_NT) {
    int i = e % na, j = e / na;
    if (i < j) continue;
    int ri = rel[i], rj = rel[j];
    Uk[i + (int64_t)j * na] = (rj < nnp) ? Pp[ri + (int64_t)rj * nfp] : Up[(ri - nnp) + (int64_t)(rj - nnp) * nap];
  }
  __syncthreads();
}

__global__ void k_gather_level(TreeArgs a, const double* x, int64_t ldx, double* updbase) {
  const CliqueDesc d = a.cl[a.lev[blockIdx.x]];
  const int r = blockIdx.y;
  double* ub = updbase + (int64_t)r * a.updlen;
  gather_sep(a, d, x + (int64_t)r * ldx, ub, ub + d.upd, blockIdx.z, gridDim.z);
}

// ---- cholesky ------------------------------------------------------------------------
__global__ void k_chol_level(TreeArgs a, double* x) {
  const int k = a.lev[blockIdx.x];
  if (*info_of(a, k)) return;
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  double* P = x + d.blk;
  double* Uk = a.upd + d.upd;
  zero((int64_t)na * na, Uk);
  add_children(a, d, a.upd, P, Uk, 1.0, 1.0);
  int f = potrf(nn, P, nf);
  if (f) {
    if (SMCP_TID == 0) atomicCAS(info_of(a, k), 0, info_val(a, k));
    return;
  }
  if (na) {
    trsm_rlT(na, nn, P, nf, P + nn, nf);
    gemm(na, na, nn, -1.0, Mat{P + nn, nf}, MatT{P + nn, nf}, 1.0, Uk, na, true);
  }
}

// ---- llt -----------------------------------------------------------------------------
__global__ void k_llt_level(TreeArgs a, double* x) {
  const int k = a.lev[blockIdx.x];
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  double* P = x + d.blk;
  double* Uk = a.upd + d.upd;
  double* T = a.tmp + a.tmpptr[k];  // nf x nn scratch
  if (na) gemm(na, na, nn, 1.0, Mat{P + nn, nf}, MatT{P + nn, nf}, 0.0, Uk, na, true);
  gemm(nf, nn, nn, 1.0, LowT{P, nf}, LowTT{P, nf}, 0.0, T, nf);  // [L_NN; L_AN] L_NN^T
  // LowT on the AN rows: rows >= nn always satisfy i >= j, so the accessor is exact there
  for (int e = SMCP_TID; e < nf * nn; e += SMCP_NT) {
    int i = e % nf, j = e / nf;
    if (i >= j) P[i + (int64_t)j * nf] = T[i + (int64_t)j * nf];
  }
  __syncthreads();
  add_children(a, d, a.upd, P, Uk, 1.0, 1.0);
}

// ---- projected inverse ---------------------------------------------------------------
__global__ void k_pinv_level(TreeArgs a, double* x) {
  const int k = a.lev[blockIdx.x];
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  double* P = x + d.blk;
  double* Uk = a.upd + d.upd;
  double* Li = a.tmp + a.tmpptr[k];        // nn x nn
  double* K = Li + (int64_t)nn * nn;       // na x nn
  double* T = K + (int64_t)na * nn;        // na x nn
  gather_sep(a, d, x, a.upd, Uk);
  set_identity(nn, Li, nn);
  trsm_llN(nn, nn, P, nf, Li, nn);
  if (na) {
    gemm(na, nn, nn, 1.0, Mat{P + nn, nf}, LowT{Li, nn}, 0.0, K, na);
    gemm(na, nn, na, 1.0, SymL{Uk, na}, Mat{K, na}, 0.0, T, na);
  }
  // Y_NN = Li^T Li + K^T T (lower)
  for (int e = SMCP_TID; e < nn * nn; e += SMCP_NT) {
    int i = e % nn, j = e / nn;
    if (i < j) continue;
    double acc = 0.0;
    for (int p = i; p < nn; ++p) acc += Li[p + (int64_t)i * nn] * Li[p + (int64_t)j * nn];
    for (int q = 0; q < na; ++q) acc += K[q + (int64_t)i * na] * T[q + (int64_t)j * na];
    P[i + (int64_t)j * nf] = acc;
  }
  if (na) copy(na, nn, T, na, P + nn, nf, -1.0);
  else __syncthreads();
}

// ---- completion (clique-local once X_AA has been gathered) --------------------------
__global__ void k_completion_all(TreeArgs a, double* x) {
  const int k = blockIdx.x;
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  double* P = x + d.blk;
  double* R = a.upd + d.upd;               // X_AA -> its Cholesky factor
  double* Sg = a.tmp + a.tmpptr[k];        // nn x nn
  double* Mi = Sg + (int64_t)nn * nn;      // nn x nn
  double* T = Mi + (int64_t)nn * nn;       // na x nn
  if (na) {
    int f = potrf(na, R, na);
    if (f) { if (SMCP_TID == 0) atomicCAS(info_of(a, k), 0, info_val(a, k)); return; }
    trsm_llN(na, nn, R, na, P + nn, nf);  // Z = R^-1 X_AN
  }
  // Sigma = X_NN - Z^T Z, stored reversed: Sg[i,j] = Sigma[nn-1-i, nn-1-j]
  for (int e = SMCP_TID; e < nn * nn; e += SMCP_NT) {
    int i = e % nn, j = e / nn;
    int ii = nn - 1 - i, jj = nn - 1 - j;
    double acc = ii >= jj ? P[ii + (int64_t)jj * nf] : P[jj + (int64_t)ii * nf];
    for (int q = 0; q < na; ++q) acc -= P[nn + q + (int64_t)ii * nf] * P[nn + q + (int64_t)jj * nf];
    Sg[i + (int64_t)j * nn] = acc;
  }
  __syncthreads();
  int f = potrf(nn, Sg, nn);
  if (f) { if (SMCP_TID == 0) atomicCAS(info_of(a, k), 0, info_val(a, k)); return; }
  set_identity(nn, Mi, nn);
  trsm_llN(nn, nn, Sg, nn, Mi, nn);  // Mi = M^-1 (lower)
  // L_NN[i,j] = Mi[nn-1-j, nn-1-i]
  for (int e = SMCP_TID; e < nn * nn; e += SMCP_NT) {
    int i = e % nn, j = e / nn;
    if (i >= j) P[i + (int64_t)j * nf] = Mi[(nn - 1 - j) + (int64_t)(nn - 1 - i) * nn];
  }
  __syncthreads();
  if (na) {
    trsm_llT(na, nn, R, na, P + nn, nf);
    gemm(na, nn, nn, -1.0, Mat{P + nn, nf}, LowT{P, nf}, 0.0, T, na);
    copy(na, nn, T, na, P + nn, nf);
  }
}

// ---- Hessian: leaves -> root half ----------------------------------------------------
__global__ void k_hess_up_level(TreeArgs a, const double* L, double* u, int64_t ldu) {
  const int k = a.lev[blockIdx.x];
  const int r = blockIdx.y;
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const double* Lk = L + d.blk;
  double* P = u + (int64_t)r * ldu + d.blk;
  double* ub = a.upd + (int64_t)r * a.updlen;
  double* Uk = ub + d.upd;
  double* T1 = a.tmp + (int64_t)r * a.tmplen + a.tmpptr[k];  // nn x nn
  double* Pm = T1 + (int64_t)nn * nn;                         // na x nn
  zero((int64_t)na * na, Uk);
  add_children(a, d, ub, P, Uk, 1.0, 1.0);
  symfull(nn, P, nf, T1);
  trsm_llN(nn, nn, Lk, nf, T1, nn);
  trsm_rlT(nn, nn, Lk, nf, T1, nn);
  if (na) {
    trsm_rlT(na, nn, Lk, nf, P + nn, nf);                                // W
    gemm(na, nn, nn, 1.0, Mat{Lk + nn, nf}, Mat{T1, nn}, 0.0, Pm, na);   // L_AN T1
    for (int e = SMCP_TID; e < na * nn; e += SMCP_NT) {
      int i = e % na, j = e / na;
      double w = P[nn + i + (int64_t)j * nf], pm = Pm[i + (int64_t)j * na];
      Pm[i + (int64_t)j * na] = w - 0.5 * pm;   // W'
      P[nn + i + (int64_t)j * nf] = w - pm;     // G_AN
    }
    __syncthreads();
    // Uk -= L_AN W'^T + W' L_AN^T (lower)
    for (int e = SMCP_TID; e < na * na; e += SMCP_NT) {
      int i = e % na, j = e / na;
      if (i < j) continue;
      double acc = 0.0;
      for (int p = 0; p < nn; ++p)
        acc += Lk[nn + i + (int64_t)p * nf] * Pm[j + (int64_t)p * na] + Pm[i + (int64_t)p * na] * Lk[nn + j + (int64_t)p * nf];
      Uk[i + (int64_t)j * na] -= acc;
    }
    __syncthreads();
  }
  copy_lower(nn, T1, nn, P, nf);
}

// ---- Hessian: root -> leaves half ----------------------------------------------------
__global__ void k_hess_down_level(TreeArgs a, const double* L, double* u, int64_t ldu) {
  const int k = a.lev[blockIdx.x];
  const int r = blockIdx.y;
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const double* Lk = L + d.blk;
  double* ur = u + (int64_t)r * ldu;
  double* P = ur + d.blk;
  double* ub = a.upd + (int64_t)r * a.updlen;
  double* Uk = ub + d.upd;
  double* M = a.tmp + (int64_t)r * a.tmplen + a.tmpptr[k];  // nn x nn
  double* Qp = M + (int64_t)nn * nn;                         // na x nn
  gather_sep(a, d, ur, ub, Uk);
  if (na) {
    // Q' = Q - Z_AA L_AN / 2
    for (int e = SMCP_TID; e < na * nn; e += SMCP_NT) {
      int i = e % na, j = e / na;
      double acc = 0.0;
      for (int q = 0; q < na; ++q) acc += (i >= q ? Uk[i + (int64_t)q * na] : Uk[q + (int64_t)i * na]) * Lk[nn + q + (int64_t)j * nf];
      Qp[i + (int64_t)j * na] = P[nn + i + (int64_t)j * nf] - 0.5 * acc;
    }
    __syncthreads();
  }
  for (int e = SMCP_TID; e < nn * nn; e += SMCP_NT) {
    int i = e % nn, j = e / nn;
    double acc = i >= j ? P[i + (int64_t)j * nf] : P[j + (int64_t)i * nf];
    for (int q = 0; q < na; ++q)
      acc -= Lk[nn + q + (int64_t)i * nf] * Qp[q + (int64_t)j * na] + Qp[q + (int64_t)i * na] * Lk[nn + q + (int64_t)j * nf];
    M[i + (int64_t)j * nn] = acc;
  }
  __syncthreads();
  if (na) {
    for (int e = SMCP_TID; e < na * nn; e += SMCP_NT) {
      int i = e % na, j = e / na;
      P[nn + i + (int64_t)j * nf] = 2.0 * Qp[i + (int64_t)j * na] - P[nn + i + (int64_t)j * nf];
    }
    __syncthreads();
    trsm_rlN(na, nn, Lk, nf, P + nn, nf);
  }
  trsm_llT(nn, nn, Lk, nf, M, nn);
  trsm_rlN(nn, nn, Lk, nf, M, nn);
  copy_lower(nn, M, nn, P, nf);
}

// ---- inverse of the root -> leaves half (clique-local after gathering Z_AA) ----------
__global__ void k_hess_down_inv_all(TreeArgs a, const double* L, double* u, int64_t ldu) {
  const int k = blockIdx.x;
  const int r = blockIdx.y;
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const double* Lk = L + d.blk;
  double* P = u + (int64_t)r * ldu + d.blk;
  const double* Uk = a.upd + (int64_t)r * a.updlen + d.upd;
  double* M = a.tmp + (int64_t)r * a.tmplen + a.tmpptr[k];  // nn x nn
  double* T = M + (int64_t)nn * nn;                          // nn x nn
  double* Qpp = T + (int64_t)nn * nn;                        // na x nn
  double* Q = Qpp + (int64_t)na * nn;                        // na x nn
  // M = L^T Z_NN L
  gemm(nn, nn, nn, 1.0, SymL{P, nf}, LowT{Lk, nf}, 0.0, T, nn);
  gemm(nn, nn, nn, 1.0, LowTT{Lk, nf}, Mat{T, nn}, 0.0, M, nn);
  if (na) {
    // Q = Z_AN L_NN + Z_AA L_AN ;  Q'' = Z_AN L_NN + Z_AA L_AN / 2
    for (int e = SMCP_TID; e < na * nn; e += SMCP_NT) {
      int i = e % na, j = e / na;
      double a1 = 0.0, a2 = 0.0;
      for (int p = j; p < nn; ++p) a1 += P[nn + i + (int64_t)p * nf] * Lk[p + (int64_t)j * nf];
      for (int q = 0; q < na; ++q)
        a2 += (i >= q ? Uk[i + (int64_t)q * na] : Uk[q + (int64_t)i * na]) * Lk[nn + q + (int64_t)j * nf];
      Qpp[i + (int64_t)j * na] = a1 + 0.5 * a2;
      Q[i + (int64_t)j * na] = a1 + a2;
    }
    __syncthreads();
    // M += L_AN^T Q'' + Q''^T L_AN
    for (int e = SMCP_TID; e < nn * nn; e += SMCP_NT) {
      int i = e % nn, j = e / nn;
      double acc = 0.0;
      for (int q = 0; q < na; ++q)
        acc += Lk[nn + q + (int64_t)i * nf] * Qpp[q + (int64_t)j * na] + Qpp[q + (int64_t)i * na] * Lk[nn + q + (int64_t)j * nf];
      M[i + (int64_t)j * nn] += acc;
    }
    __syncthreads();
    copy(na, nn, Q, na, P + nn, nf);
  }
  copy_lower(nn, M, nn, P, nf);
}

// ---- inverse of the leaves -> root half ---------------------------------------------
__global__ void k_hess_up_inv_level(TreeArgs a, const double* L, double* u, int64_t ldu) {
  const int k = a.lev[blockIdx.x];
  const int r = blockIdx.y;
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const double* Lk = L + d.blk;
  double* P = u + (int64_t)r * ldu + d.blk;
  double* ub = a.upd + (int64_t)r * a.updlen;
  double* Uk = ub + d.upd;
  double* G = a.tmp + (int64_t)r * a.tmplen + a.tmpptr[k];  // nn x nn full
  double* Vv = G + (int64_t)nn * nn;                         // na x nn
  double* T = Vv + (int64_t)na * nn;                         // nf x nn
  symfull(nn, P, nf, G);
  if (na) {
    // V = G_AN + L_AN G_NN / 2
    for (int e = SMCP_TID; e < na * nn; e += SMCP_NT) {
      int i = e % na, j = e / na;
      double acc = 0.0;
      for (int p = 0; p < nn; ++p) acc += Lk[nn + i + (int64_t)p * nf] * G[p + (int64_t)j * nn];
      Vv[i + (int64_t)j * na] = P[nn + i + (int64_t)j * nf] + 0.5 * acc;
    }
    __syncthreads();
    for (int e = SMCP_TID; e < na * na; e += SMCP_NT) {
      int i = e % na, j = e / na;
      if (i < j) continue;
      double acc = 0.0;
      for (int p = 0; p < nn; ++p)
        acc += Vv[i + (int64_t)p * na] * Lk[nn + j + (int64_t)p * nf] + Lk[nn + i + (int64_t)p * nf] * Vv[j + (int64_t)p * na];
      Uk[i + (int64_t)j * na] = -acc;
    }
    // F_AN = (2V - G_AN) L_NN^T  -> T rows nn..nf
    for (int e = SMCP_TID; e < na * nn; e += SMCP_NT) {
      int i = e % na, j = e / na;
      double acc = 0.0;
      for (int p = 0; p <= j; ++p)
        acc += (2.0 * Vv[i + (int64_t)p * na] - P[nn + i + (int64_t)p * nf]) * Lk[j + (int64_t)p * nf];
      T[nn + i + (int64_t)j * nf] = acc;
    }
  }
  // F_NN = L G L^T -> T rows 0..nn ; first X = L G into Vv-free scratch? reuse: two-step via rows
  __syncthreads();
  {
    // X = G L^T (nn x nn) stored in place of G column by column is unsafe; use T's top block as X
    for (int e = SMCP_TID; e < nn * nn; e += SMCP_NT) {
      int i = e % nn, j = e / nn;
      double acc = 0.0;
      for (int p = 0; p <= j; ++p) acc += G[i + (int64_t)p * nn] * Lk[j + (int64_t)p * nf];
      T[i + (int64_t)j * nf] = acc;  // X = G L^T
    }
    __syncthreads();
    for (int e = SMCP_TID; e < nn * nn; e += SMCP_NT) {
      int i = e % nn, j = e / nn;
      double acc = 0.0;
      for (int p = 0; p <= i; ++p) acc += Lk[i + (int64_t)p * nf] * T[p + (int64_t)j * nf];
      G[i + (int64_t)j * nn] = acc;  // L X
    }
    __syncthreads();
  }
  for (int e = SMCP_TID; e < nf * nn; e += SMCP_NT) {
    int i = e % nf, j = e / nf;
    if (i < j) continue;
    P[i + (int64_t)j * nf] = i < nn ? G[i + (int64_t)j * nn] : T[i + (int64_t)j * nf];
  }
  __syncthreads();
  add_children(a, d, ub, P, Uk, -1.0, 1.0);
}

// ---- AN-block scalings used to compose G, G^adj, H and their inverses ----------------
// mode 0: R^T B ; 1: R B ; 2: R^-T B ; 3: R^-1 B ; 4: Y_AA B ; 5: Y_AA^-1 B (two solves with R)
__global__ void k_scale_an(TreeArgs a, const double* yaa, const double* fac, double* u, int64_t ldu,
                           int mode) {
  const int k = blockIdx.x;
  const int r = blockIdx.y;
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  if (!na) return;
  double* B = u + (int64_t)r * ldu + d.blk + nn;
  const double* R = fac + d.upd;
  double* T = a.tmp + (int64_t)r * a.tmplen + a.tmpptr[k];  // na x nn
  switch (mode) {
    case 0: gemm(na, nn, na, 1.0, LowTT{R, na}, Mat{B, nf}, 0.0, T, na); copy(na, nn, T, na, B, nf); break;
    case 1: gemm(na, nn, na, 1.0, LowT{R, na}, Mat{B, nf}, 0.0, T, na); copy(na, nn, T, na, B, nf); break;
    case 2: trsm_llT(na, nn, R, na, B, nf); break;
    case 3: trsm_llN(na, nn, R, na, B, nf); break;
    case 4: gemm(na, nn, na, 1.0, SymL{yaa + d.upd, na}, Mat{B, nf}, 0.0, T, na); copy(na, nn, T, na, B, nf); break;
    case 5: trsm_llN(na, nn, R, na, B, nf); trsm_llT(na, nn, R, na, B, nf); break;
  }
}

// fac[k] <- chol(yaa[k]) for every clique
__global__ void k_factor_yaa(TreeArgs a, const double* yaa, double* fac) {
  const int k = blockIdx.x;
  const CliqueDesc d = a.cl[k];
  const int na = d.na;
  if (!na) return;
  double* R = fac + d.upd;
  copy_lower(na, yaa + d.upd, na, R, na);
  int f = potrf(na, R, na);
  if (f && SMCP_TID == 0) atomicCAS(info_of(a, k), 0, info_val(a, k));
}

// ---- supernodal triangular solves with a dense right-hand side ------------------------
// Update vectors travel through the (na x nrhs) blocks of the update workspace (ld na).
// Right-hand-side columns are independent: blockIdx.y takes a block of TRSM_CB columns of B.
constexpr int TRSM_CB = 16;
__global__ void k_trsm_fwd_level(TreeArgs a, const double* L, double* B, int nrhs, int64_t ldb,
                                 const int32_t* rowidx) {
  const int k = a.lev[blockIdx.x];
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const int c0 = blockIdx.y * TRSM_CB, nc = min(TRSM_CB, nrhs - c0);
  if (nc <= 0) return;
  const double* Lk = L + d.blk;
  double* Bn = B + d.first + (int64_t)c0 * ldb;
  double* Uk = a.tmp + d.rel * nrhs + (int64_t)c0 * na;  // na x nrhs (ld na); d.rel == sepptr[k]
  // gather children's update vectors: rows of the child separator map into this clique's rows
  zero((int64_t)na * nc, Uk);
  for (int q = d.chbeg; q < d.chend; ++q) {
    const CliqueDesc c = a.cl[a.chidx[q]];
    const int32_t* rel = a.relidx + c.rel;
    const double* Uc = a.tmp + c.rel * nrhs + (int64_t)c0 * c.na;
    for (int e = SMCP_TID; e < c.na * nc; e += SMCP_NT) {
      int i = e % c.na, col = e / c.na;
      int ri = rel[i];
      double v = Uc[i + (int64_t)col * c.na];
      if (ri < nn) Bn[ri + (int64_t)col * ldb] += v;
      else Uk[(ri - nn) + (int64_t)col * na] += v;
    }
    __syncthreads();
  }
  trsm_llN(nn, nc, Lk, nf, Bn, ldb);
  if (na) gemm(na, nc, nn, -1.0, Mat{Lk + nn, nf}, Mat{Bn, ldb}, 1.0, Uk, na);
}
__global__ void k_trsm_bwd_level(TreeArgs a, const double* L, double* B, int nrhs, int64_t ldb,
                                 const int32_t* rowidx) {
  const int k = a.lev[blockIdx.x];
  const CliqueDesc d = a.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const int c0 = blockIdx.y * TRSM_CB, nc = min(TRSM_CB, nrhs - c0);
  if (nc <= 0) return;
  const double* Lk = L + d.blk;
  double* Bc = B + (int64_t)c0 * ldb;
  double* Bn = Bc + d.first;
  const int32_t* rows = rowidx + d.rows;
  // B_N <- L_NN^-T (B_N - L_AN^T B_A); B_A rows are final (ancestors processed earlier)
  for (int e = SMCP_TID; e < nn * nc; e += SMCP_NT) {
    int j = e % nn, col = e / nn;
    double acc = 0.0;
    for (int i = 0; i < na; ++i) acc += Lk[nn + i + (int64_t)j * nf] * Bc[rows[nn + i] + (int64_t)col * ldb];
    Bn[j + (int64_t)col * ldb] -= acc;
  }
  __syncthreads();
  trsm_llT(nn, nc, Lk, nf, Bn, ldb);
}

// ---- flat reductions -------------------------------------------------------------------
// mode 0: dot (diag once, others twice) of x,y ; mode 1: sum log diag(x)
__global__ void k_reduce_cliques(const CliqueDesc* cl, int nsn, const double* x, const double* y,
                                 int mode, double* partial) {
  double acc = 0.0;
  for (int k = blockIdx.x; k < nsn; k += gridDim.x) {
    const CliqueDesc d = cl[k];
    const int nn = d.nn, nf = d.nn + d.na;
    const double* a = x + d.blk;
    if (mode == 0) {
      const double* b = y + d.blk;
      for (int e = SMCP_TID; e < nf * nn; e += SMCP_NT) {
        int i = e % nf, j = e / nf;
        if (i < j) continue;
        double v = a[e] * b[e];
        acc += (i == j) ? v : 2.0 * v;
      }
    } else {
      for (int j = SMCP_TID; j < nn; j += SMCP_NT) acc += log(a[j + (int64_t)j * nf]);
    }
  }
  __shared__ double sh[256];
  sh[SMCP_TID] = acc;
  __syncthreads();
  for (int s = SMCP_NT / 2; s > 0; s >>= 1) {
    if (SMCP_TID < s) sh[SMCP_TID] += sh[SMCP_TID + s];
    __syncthreads();
  }
  if (SMCP_TID == 0) partial[blockIdx.x] = sh[0];
}
// the same reductions over the flat blkval with the inner-product weights read from sw (1 diagonal, sqrt 2
// strictly lower, 0 unused upper): every workgroup takes a slice, so one huge front does not serialise
__global__ void k_reduce_flat(int64_t len, const double* sw, const double* x, const double* y, int mode, double* partial) {
  double acc = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * SMCP_NT + SMCP_TID; e < len; e += (int64_t)gridDim.x * SMCP_NT) {
    const double s = sw[e];
    if (mode == 0) { if (s != 0.0) { const double v = x[e] * y[e]; acc += (s == 1.0) ? v : 2.0 * v; } }
    else if (s == 1.0) acc += log(x[e]);
  }
  __shared__ double sh[256];
  sh[SMCP_TID] = acc;
  __syncthreads();
  for (int s2 = SMCP_NT / 2; s2 > 0; s2 >>= 1) {
    if (SMCP_TID < s2) sh[SMCP_TID] += sh[SMCP_TID + s2];
    __syncthreads();
  }
  if (SMCP_TID == 0) partial[blockIdx.x] = sh[0];
}
__global__ void k_reduce_final(const double* partial, int n, double* out) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = SMCP_TID; i < n; i += SMCP_NT) acc += partial[i];
  sh[SMCP_TID] = acc;
  __syncthreads();
  for (int s = SMCP_NT / 2; s > 0; s >>= 1) {
    if (SMCP_TID < s) sh[SMCP_TID] += sh[SMCP_TID + s];
    __syncthreads();
  }
  if (SMCP_TID == 0) *out = sh[0];
}

__global__ void k_axpby(int64_t len, double a, const double* x, double b, double* y) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < len; i += stride) y[i] = (b == 0.0 ? 0.0 : b * y[i]) + (x ? a * x[i] : 0.0);
}

}  // namespace smcp
