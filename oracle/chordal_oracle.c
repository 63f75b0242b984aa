/*
 * chordal_oracle.c -- CPU restatement of the chordal-matrix kernels on the Newton-KKT path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under smcp_amd/ may import, link or call this file;
 * it is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the third-party package CHOMPACK
 * (pinned ">= 2.3.4", /root/reference/pyproject.toml:25-28; imported at
 * /root/reference/src/python/solvers.py:82-97), which is not vendored in the reference and
 * is not installable here, and the reference's only test (tests/test_basic.py:6-22) asserts
 * nothing.  There are therefore no golden vectors.  This file restates the published
 * supernodal multifrontal algorithms (Andersen, Dahl, Vandenberghe, "Logarithmic barriers
 * for sparse matrix cones", 2013) following the reference's own call sites, and is pinned
 * instead by exact dense identities in tests/test_oracle_identities.py:
 *   cholesky            L L^T = X with zero fill           (call sites solvers.py:640,884)
 *   projected_inverse   Y = P_V((L L^T)^-1)                (solvers.py:891,2361)
 *   completion          P_V((L L^T)^-1) = X                (solvers.py:625,874; 392-393)
 *   hessian             adj=None: U <- P_V(S^-1 U S^-1)    (solvers.py:405,483,524,531)
 *                       adj=False/True: factors G, G^adj with H = G^adj o G (solvers.py:403-404,482)
 *   llt                 X = L L^T on V                     (solvers.py:904,1721)
 *   trsm                supernodal triangular solve        (solvers.py:491-492)
 *   dot                 tr(XY) on V                        (solvers.py:399,836)
 *
 * Storage (same flat layout the product uses): clique k owns a dense column-major
 * (nn+na) x nn block [X_NN; X_AN] at blkval + blkptr[k]; only the lower triangle of X_NN is
 * meaningful.  Update matrices (na x na, lower triangle meaningful) live at upd + updptr[k].
 * Cliques are numbered in postorder (children before parents).
 *
 * Plain C, straightforward loops: clique-by-clique in postorder, dense BLAS-3-shaped
 * operations per clique, exactly the structure CHOMPACK's Python/C routines have ([EXT],
 * SURVEY.md App. A).  Every routine is single threaded; orc_schur_columns (bottom of the file)
 * spreads the INDEPENDENT columns of the Schur complement over OpenMP threads so that bench.py's
 * cpu_baseline can use the host's cores.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t n, nsn;
  const int64_t *snptr, *snpar, *rowptr;
  const int32_t *rowidx;
  const int64_t *sepptr;
  const int32_t *relidx;
  const int64_t *blkptr, *updptr, *chptr, *chidx;
} orc_sym;

#define NN(s, k) ((s)->snptr[(k) + 1] - (s)->snptr[k])
#define NF(s, k) ((s)->rowptr[(k) + 1] - (s)->rowptr[k])

/* ------------------------------------------------------------------ dense helpers */

/* Optional host BLAS / LAPACK for the per-clique dense operations (what CHOMPACK calls through CVXOPT [EXT]):
 * oracle.use_blas() hands over the Fortran-interface entry points of the OpenBLAS that scipy bundles
 * (scipy.linalg.cython_blas / cython_lapack).  Used for blocks with a dimension >= blas_min only; the plain loops
 * below remain the default and the parity checker (tests/test_oracle_identities.py runs both). */
typedef void (*dgemm_f)(char *, char *, int *, int *, int *, double *, double *, int *, double *, int *, double *, double *, int *);
typedef void (*dtrsm_f)(char *, char *, char *, char *, int *, int *, double *, double *, int *, double *, int *);
typedef void (*dtrmm_f)(char *, char *, char *, char *, int *, int *, double *, double *, int *, double *, int *);
typedef void (*dsyrk_f)(char *, char *, int *, int *, double *, double *, int *, double *, double *, int *);
typedef void (*dpotrf_f)(char *, int *, double *, int *, int *);
static struct { dgemm_f gemm; dtrsm_f trsm; dtrmm_f trmm; dsyrk_f syrk; dpotrf_f potrf; int min; } blas = {0, 0, 0, 0, 0, 32};
void orc_set_blas(void *gemm_, void *trsm_, void *trmm_, void *syrk_, void *potrf_, int min_dim) {
  blas.gemm = (dgemm_f)gemm_; blas.trsm = (dtrsm_f)trsm_; blas.trmm = (dtrmm_f)trmm_;
  blas.syrk = (dsyrk_f)syrk_; blas.potrf = (dpotrf_f)potrf_; blas.min = min_dim > 0 ? min_dim : 32;
}
int orc_blas_enabled(void) { return blas.gemm != 0; }
static int use_blas(int64_t a, int64_t b) { return blas.gemm && (a >= blas.min || b >= blas.min) && a > 0 && b > 0; }
static void blas_tr(int mm, dtrsm_f f, char side, char trans, int64_t m, int64_t n, const double *L, int64_t ldl,
                    double *B, int64_t ldb) {
  char uplo = 'L', diag = 'N';
  int m_ = (int)m, n_ = (int)n, ldl_ = (int)ldl, ldb_ = (int)ldb;
  double one = 1.0;
  (void)mm;
  f(&side, &uplo, &trans, &diag, &m_, &n_, &one, (double *)L, &ldl_, B, &ldb_);
}

/* in-place lower Cholesky; returns 0 or j+1 of the failing pivot */
static int potrf_l(int64_t n, double *A, int64_t lda) {
  if (blas.potrf && n >= blas.min) {
    char uplo = 'L';
    int n_ = (int)n, lda_ = (int)lda, info = 0;
    blas.potrf(&uplo, &n_, A, &lda_, &info);
    return info > 0 ? info : 0;
  }
  for (int64_t j = 0; j < n; ++j) {
    double d = A[j + j * lda];
    if (!(d > 0.0)) return (int)(j + 1);
    d = sqrt(d);
    A[j + j * lda] = d;
    double r = 1.0 / d;
    for (int64_t i = j + 1; i < n; ++i) A[i + j * lda] *= r;
    for (int64_t k = j + 1; k < n; ++k) {
      double a = A[k + j * lda];
      for (int64_t i = k; i < n; ++i) A[i + k * lda] -= A[i + j * lda] * a;
    }
  }
  return 0;
}
/* B (m x n) <- L^-1 B, L m x m lower */
static void trsm_llN(int64_t m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb) {
  if (use_blas(m, n)) { blas_tr(0, blas.trsm, 'L', 'N', m, n, L, ldl, B, ldb); return; }
  for (int64_t c = 0; c < n; ++c) {
    double *b = B + c * ldb;
    for (int64_t j = 0; j < m; ++j) {
      double x = b[j] / L[j + j * ldl];
      b[j] = x;
      for (int64_t i = j + 1; i < m; ++i) b[i] -= L[i + j * ldl] * x;
    }
  }
}
/* B (m x n) <- L^-T B */
static void trsm_llT(int64_t m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb) {
  if (use_blas(m, n)) { blas_tr(0, blas.trsm, 'L', 'T', m, n, L, ldl, B, ldb); return; }
  for (int64_t c = 0; c < n; ++c) {
    double *b = B + c * ldb;
    for (int64_t j = m - 1; j >= 0; --j) {
      double s = b[j];
      for (int64_t i = j + 1; i < m; ++i) s -= L[i + j * ldl] * b[i];
      b[j] = s / L[j + j * ldl];
    }
  }
}
/* B (m x n) <- B L^-T, L n x n lower:  X L^T = B  => column j of X depends on columns < j */
static void trsm_rlT(int64_t m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb) {
  if (use_blas(m, n)) { blas_tr(0, blas.trsm, 'R', 'T', m, n, L, ldl, B, ldb); return; }
  for (int64_t j = 0; j < n; ++j) {
    double *bj = B + j * ldb;
    for (int64_t k = 0; k < j; ++k) {
      double l = L[j + k * ldl];
      const double *bk = B + k * ldb;
      for (int64_t i = 0; i < m; ++i) bj[i] -= bk[i] * l;
    }
    double r = 1.0 / L[j + j * ldl];
    for (int64_t i = 0; i < m; ++i) bj[i] *= r;
  }
}
/* B (m x n) <- B L^-1:  X L = B => column j of X depends on columns > j */
static void trsm_rlN(int64_t m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb) {
  if (use_blas(m, n)) { blas_tr(0, blas.trsm, 'R', 'N', m, n, L, ldl, B, ldb); return; }
  for (int64_t j = n - 1; j >= 0; --j) {
    double *bj = B + j * ldb;
    for (int64_t k = j + 1; k < n; ++k) {
      double l = L[k + j * ldl];
      const double *bk = B + k * ldb;
      for (int64_t i = 0; i < m; ++i) bj[i] -= bk[i] * l;
    }
    double r = 1.0 / L[j + j * ldl];
    for (int64_t i = 0; i < m; ++i) bj[i] *= r;
  }
}
/* B (m x n) <- B L^T : new col j = sum_{k<=j} B[:,k] L[j,k]; go j descending to stay in place */
static void trmm_rlT(int64_t m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb) {
  if (use_blas(m, n)) { blas_tr(0, (dtrsm_f)blas.trmm, 'R', 'T', m, n, L, ldl, B, ldb); return; }
  for (int64_t j = n - 1; j >= 0; --j) {
    double *bj = B + j * ldb;
    double d = L[j + j * ldl];
    for (int64_t i = 0; i < m; ++i) bj[i] *= d;
    for (int64_t k = 0; k < j; ++k) {
      double l = L[j + k * ldl];
      const double *bk = B + k * ldb;
      for (int64_t i = 0; i < m; ++i) bj[i] += bk[i] * l;
    }
  }
}
/* B (m x n) <- B L : new col j = sum_{k>=j} B[:,k] L[k,j]; go j ascending */
static void trmm_rlN(int64_t m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb) {
  if (use_blas(m, n)) { blas_tr(0, (dtrsm_f)blas.trmm, 'R', 'N', m, n, L, ldl, B, ldb); return; }
  for (int64_t j = 0; j < n; ++j) {
    double *bj = B + j * ldb;
    double d = L[j + j * ldl];
    for (int64_t i = 0; i < m; ++i) bj[i] *= d;
    for (int64_t k = j + 1; k < n; ++k) {
      double l = L[k + j * ldl];
      const double *bk = B + k * ldb;
      for (int64_t i = 0; i < m; ++i) bj[i] += bk[i] * l;
    }
  }
}
/* B (m x n) <- L B, L m x m lower (left, notrans): rows descending in place */
static void trmm_llN(int64_t m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb) {
  if (use_blas(m, n)) { blas_tr(0, (dtrsm_f)blas.trmm, 'L', 'N', m, n, L, ldl, B, ldb); return; }
  for (int64_t c = 0; c < n; ++c) {
    double *b = B + c * ldb;
    for (int64_t i = m - 1; i >= 0; --i) {
      double s = 0.0;
      for (int64_t k = 0; k <= i; ++k) s += L[i + k * ldl] * b[k];
      b[i] = s;
    }
  }
}
/* B (m x n) <- L^T B */
static void trmm_llT(int64_t m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb) {
  if (use_blas(m, n)) { blas_tr(0, (dtrsm_f)blas.trmm, 'L', 'T', m, n, L, ldl, B, ldb); return; }
  for (int64_t c = 0; c < n; ++c) {
    double *b = B + c * ldb;
    for (int64_t i = 0; i < m; ++i) {
      double s = 0.0;
      for (int64_t k = i; k < m; ++k) s += L[k + i * ldl] * b[k];
      b[i] = s;
    }
  }
}
/* C (m x n) = beta C + alpha op(A) op(B); ta/tb: 0 = as is, 1 = transposed. k = inner dim */
static void gemm(int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha, const double *A,
                 int64_t lda, const double *B, int64_t ldb, double beta, double *C, int64_t ldc) {
  if (blas.gemm && m > 0 && n > 0 && k > 0 && (m >= blas.min || n >= blas.min || k >= blas.min)) {
    char ta_ = ta ? 'T' : 'N', tb_ = tb ? 'T' : 'N';
    int m_ = (int)m, n_ = (int)n, k_ = (int)k, lda_ = (int)lda, ldb_ = (int)ldb, ldc_ = (int)ldc;
    blas.gemm(&ta_, &tb_, &m_, &n_, &k_, &alpha, (double *)A, &lda_, (double *)B, &ldb_, &beta, C, &ldc_);
    return;
  }
  for (int64_t j = 0; j < n; ++j) {
    double *c = C + j * ldc;
    if (beta == 0.0) for (int64_t i = 0; i < m; ++i) c[i] = 0.0;
    else if (beta != 1.0) for (int64_t i = 0; i < m; ++i) c[i] *= beta;
    for (int64_t p = 0; p < k; ++p) {
      double b = alpha * (tb ? B[j + p * ldb] : B[p + j * ldb]);
      if (b == 0.0) continue;
      if (!ta) {
        const double *a = A + p * lda;
        for (int64_t i = 0; i < m; ++i) c[i] += a[i] * b;
      } else {
        for (int64_t i = 0; i < m; ++i) c[i] += A[p + i * lda] * b;
      }
    }
  }
}
/* expand the lower triangle of A (n x n, lda) into a full symmetric matrix F (ld n) */
static void symfull(int64_t n, const double *A, int64_t lda, double *F) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = j; i < n; ++i) F[i + j * n] = F[j + i * n] = A[i + j * lda];
}
static void put_lower(int64_t n, const double *F, double *A, int64_t lda) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = j; i < n; ++i) A[i + j * lda] = F[i + j * n];
}

/* ------------------------------------------------------------------ tree plumbing */

/* extend-add the (lower) update matrix of child c into parent's panel P (nf_p x nn_p) and
 * parent's update matrix Up (na_p x na_p); sgn = +1 or -1 */
static void extend_add(const orc_sym *s, int64_t c, const double *Uc, double *P, double *Up,
                       double sgn) {
  int64_t p = s->snpar[c];
  int64_t nac = NF(s, c) - NN(s, c), nnp = NN(s, p), nfp = NF(s, p), nap = nfp - nnp;
  const int32_t *rel = s->relidx + s->sepptr[c];
  for (int64_t j = 0; j < nac; ++j) {
    int64_t rj = rel[j];
    for (int64_t i = j; i < nac; ++i) {
      int64_t ri = rel[i];
      double v = sgn * Uc[i + j * nac];
      if (rj < nnp) P[ri + rj * nfp] += v;
      else Up[(ri - nnp) + (rj - nnp) * nap] += v;
    }
  }
}
/* gather the A_c x A_c block (lower) of the parent's front [P | Up] into Uc */
static void gather_sep(const orc_sym *s, int64_t c, const double *P, const double *Up, double *Uc) {
  int64_t p = s->snpar[c];
  int64_t nac = NF(s, c) - NN(s, c), nnp = NN(s, p), nfp = NF(s, p), nap = nfp - nnp;
  const int32_t *rel = s->relidx + s->sepptr[c];
  for (int64_t j = 0; j < nac; ++j) {
    int64_t rj = rel[j];
    for (int64_t i = j; i < nac; ++i) {
      int64_t ri = rel[i];
      Uc[i + j * nac] = (rj < nnp) ? P[ri + rj * nfp] : Up[(ri - nnp) + (rj - nnp) * nap];
    }
  }
}
/* top-down: upd[k] <- X[A_k, A_k] for every clique, reading the panels of x unchanged */
static void gather_all(const orc_sym *s, const double *x, double *upd) {
  for (int64_t k = s->nsn - 1; k >= 0; --k) {
    int64_t p = s->snpar[k];
    if (p < 0) continue;
    gather_sep(s, k, x + s->blkptr[p], upd + s->updptr[p], upd + s->updptr[k]);
  }
}
static void add_children(const orc_sym *s, int64_t k, const double *upd, double *P, double *Uk,
                         double sgn) {
  for (int64_t q = s->chptr[k]; q < s->chptr[k + 1]; ++q) {
    int64_t c = s->chidx[q];
    extend_add(s, c, upd + s->updptr[c], P, Uk, sgn);
  }
}

/* ------------------------------------------------------------------ public kernels */

/* X -> L, L L^T = X.  upd: workspace of updptr[nsn] doubles.  Returns 0 or k+1 (clique). */
static int cholesky_m(const orc_sym *s, double *x, double *upd, const unsigned char *mask) {
  for (int64_t k = 0; k < s->nsn; ++k) {
    if (mask && !mask[k]) continue;
    int64_t nn = NN(s, k), nf = NF(s, k), na = nf - nn;
    double *P = x + s->blkptr[k], *Uk = upd + s->updptr[k];
    memset(Uk, 0, sizeof(double) * na * na);
    add_children(s, k, upd, P, Uk, 1.0);
    if (potrf_l(nn, P, nf)) return (int)(k + 1);
    if (na) {
      trsm_rlT(na, nn, P, nf, P + nn, nf);
      if (blas.syrk && (na >= blas.min || nn >= blas.min)) {
        char uplo = 'L', tr = 'N';
        int na_ = (int)na, nn_ = (int)nn, nf_ = (int)nf;
        double mone = -1.0, one = 1.0;
        blas.syrk(&uplo, &tr, &na_, &nn_, &mone, P + nn, &nf_, &one, Uk, &na_);
      } else
      for (int64_t j = 0; j < na; ++j)
        for (int64_t p = 0; p < nn; ++p) {
          double b = P[nn + j + p * nf];
          for (int64_t i = j; i < na; ++i) Uk[i + j * na] -= P[nn + i + p * nf] * b;
        }
    }
  }
  return 0;
}
int orc_cholesky(const orc_sym *s, double *x, double *upd) { return cholesky_m(s, x, upd, 0); }

/* L -> X = L L^T restricted to V */
int orc_llt(const orc_sym *s, double *x, double *upd) {
  for (int64_t k = 0; k < s->nsn; ++k) {
    int64_t nn = NN(s, k), nf = NF(s, k), na = nf - nn;
    double *P = x + s->blkptr[k], *Uk = upd + s->updptr[k];
    for (int64_t j = 0; j < na; ++j)
      for (int64_t i = j; i < na; ++i) {
        double v = 0.0;
        for (int64_t p = 0; p < nn; ++p) v += P[nn + i + p * nf] * P[nn + j + p * nf];
        Uk[i + j * na] = v;
      }
    double *T = (double *)malloc(sizeof(double) * nn * nn);
    for (int64_t j = 0; j < nn; ++j)
      for (int64_t i = 0; i < nn; ++i) T[i + j * nn] = (i >= j) ? P[i + j * nf] : 0.0;
    if (na) trmm_rlT(na, nn, T, nn, P + nn, nf);
    for (int64_t j = 0; j < nn; ++j)
      for (int64_t i = j; i < nn; ++i) {
        double v = 0.0;
        for (int64_t p = 0; p <= j; ++p) v += T[i + p * nn] * T[j + p * nn];
        P[i + j * nf] = v;
      }
    free(T);
    add_children(s, k, upd, P, Uk, 1.0);
  }
  return 0;
}

/* L -> Y = P_V((L L^T)^-1) */
static int projected_inverse_m(const orc_sym *s, double *x, double *upd, const unsigned char *mask) {
  for (int64_t k = s->nsn - 1; k >= 0; --k) {
    if (mask && !mask[k]) continue;
    int64_t nn = NN(s, k), nf = NF(s, k), na = nf - nn, p = s->snpar[k];
    double *P = x + s->blkptr[k], *Uk = upd + s->updptr[k];
    if (p >= 0) gather_sep(s, k, x + s->blkptr[p], upd + s->updptr[p], Uk);
    /* Linv^T Linv */
    double *Li = (double *)calloc(nn * nn, sizeof(double));
    for (int64_t i = 0; i < nn; ++i) Li[i + i * nn] = 1.0;
    trsm_llN(nn, nn, P, nf, Li, nn); /* Li = L^-1 (lower) */
    double *Ynn = (double *)malloc(sizeof(double) * nn * nn);
    gemm(1, 0, nn, nn, nn, 1.0, Li, nn, Li, nn, 0.0, Ynn, nn);
    if (na) {
      double *Yf = (double *)malloc(sizeof(double) * na * na);
      double *T = (double *)malloc(sizeof(double) * na * nn);
      symfull(na, Uk, na, Yf);
      trsm_rlN(na, nn, P, nf, P + nn, nf);                        /* K = L_AN L_NN^-1 */
      gemm(0, 0, na, nn, na, 1.0, Yf, na, P + nn, nf, 0.0, T, na); /* T = Y_AA K */
      gemm(1, 0, nn, nn, na, 1.0, P + nn, nf, T, na, 1.0, Ynn, nn); /* += K^T T */
      for (int64_t j = 0; j < nn; ++j)
        for (int64_t i = 0; i < na; ++i) P[nn + i + j * nf] = -T[i + j * na];
      free(Yf);
      free(T);
    }
    put_lower(nn, Ynn, P, nf);
    free(Li);
    free(Ynn);
  }
  return 0;
}
int orc_projected_inverse(const orc_sym *s, double *x, double *upd) { return projected_inverse_m(s, x, upd, 0); }

/* "UL" factorisation S = U U^T (U upper) of a full symmetric nn x nn matrix, then
 * return L = U^-T (lower) in Lout (full storage, zeros above).  0 ok / j+1 on failure */
static int inv_chol_rev(int64_t nn, double *Sg, double *Lout) {
  /* reverse rows/cols: J S J = M M^T, M lower => U = J M J */
  double *M = (double *)malloc(sizeof(double) * nn * nn);
  for (int64_t j = 0; j < nn; ++j)
    for (int64_t i = 0; i < nn; ++i) M[i + j * nn] = Sg[(nn - 1 - i) + (nn - 1 - j) * nn];
  int info = potrf_l(nn, M, nn);
  if (info) { free(M); return info; }
  /* U[i,j] = M[nn-1-i, nn-1-j] (upper).  L = U^-T  <=>  L^T = U^-1  <=> U L^T = I.
   * Equivalently (J M J) L^T = I => M (J L^T J) = I => J L^T J = M^-1 (lower) */
  double *Mi = (double *)calloc(nn * nn, sizeof(double));
  for (int64_t i = 0; i < nn; ++i) Mi[i + i * nn] = 1.0;
  trsm_llN(nn, nn, M, nn, Mi, nn);
  /* L^T = J Mi J  => L[i,j] = (J Mi J)[j,i] = Mi[nn-1-j, nn-1-i] */
  for (int64_t j = 0; j < nn; ++j)
    for (int64_t i = 0; i < nn; ++i) Lout[i + j * nn] = (i >= j) ? Mi[(nn - 1 - j) + (nn - 1 - i) * nn] : 0.0;
  free(M);
  free(Mi);
  return 0;
}

/* X -> L with P_V((L L^T)^-1) = X (factor of the inverse of the max-det completion) */
int orc_completion(const orc_sym *s, double *x, double *upd) {
  gather_all(s, x, upd);
  for (int64_t k = s->nsn - 1; k >= 0; --k) {
    int64_t nn = NN(s, k), nf = NF(s, k), na = nf - nn;
    double *P = x + s->blkptr[k], *Uk = upd + s->updptr[k];
    double *Sg = (double *)malloc(sizeof(double) * nn * nn);
    double *Ln = (double *)malloc(sizeof(double) * nn * nn);
    symfull(nn, P, nf, Sg);
    if (na) {
      if (potrf_l(na, Uk, na)) { free(Sg); free(Ln); return (int)(k + 1); }
      trsm_llN(na, nn, Uk, na, P + nn, nf);                            /* Z = R^-1 X_AN */
      gemm(1, 0, nn, nn, na, -1.0, P + nn, nf, P + nn, nf, 1.0, Sg, nn); /* Sigma = X_NN - Z^T Z */
    }
    if (inv_chol_rev(nn, Sg, Ln)) { free(Sg); free(Ln); return (int)(k + 1); }
    if (na) {
      trsm_llT(na, nn, Uk, na, P + nn, nf); /* R^-T Z = X_AA^-1 X_AN */
      trmm_rlN(na, nn, Ln, nn, P + nn, nf);
      for (int64_t j = 0; j < nn; ++j)
        for (int64_t i = 0; i < na; ++i) P[nn + i + j * nf] = -P[nn + i + j * nf];
    }
    put_lower(nn, Ln, P, nf);
    free(Sg);
    free(Ln);
  }
  return 0;
}

/* yaa[k] <- Y[A_k,A_k] (lower); if fac != NULL also fac[k] <- chol(Y_AA) (lower). */
static int prepare_yaa_m(const orc_sym *s, const double *Y, double *yaa, double *fac, const unsigned char *mask) {
  for (int64_t k = s->nsn - 1; k >= 0; --k) {
    int64_t p = s->snpar[k];
    if (p < 0 || (mask && !mask[k])) continue;
    gather_sep(s, k, Y + s->blkptr[p], yaa + s->updptr[p], yaa + s->updptr[k]);
  }
  if (fac) {
    for (int64_t k = 0; k < s->nsn; ++k) {
      int64_t na = NF(s, k) - NN(s, k);
      if (mask && !mask[k]) continue;
      memcpy(fac + s->updptr[k], yaa + s->updptr[k], sizeof(double) * na * na);
      if (na && potrf_l(na, fac + s->updptr[k], na)) return (int)(k + 1);
    }
  }
  return 0;
}
static int prepare_yaa(const orc_sym *s, const double *Y, double *yaa, double *fac) { return prepare_yaa_m(s, Y, yaa, fac, 0); }

/* leaves->root half of the Hessian: panel <- (G_NN, G_AN) */
static void hess_up_m(const orc_sym *s, const double *L, double *u, double *upd, const unsigned char *mask) {
  for (int64_t k = 0; k < s->nsn; ++k) {
    if (mask && !mask[k]) continue;
    int64_t nn = NN(s, k), nf = NF(s, k), na = nf - nn;
    const double *Lk = L + s->blkptr[k];
    double *P = u + s->blkptr[k], *Uk = upd + s->updptr[k];
    memset(Uk, 0, sizeof(double) * na * na);
    add_children(s, k, upd, P, Uk, 1.0);
    double *T1 = (double *)malloc(sizeof(double) * nn * nn);
    symfull(nn, P, nf, T1);
    trsm_llN(nn, nn, Lk, nf, T1, nn);
    trsm_rlT(nn, nn, Lk, nf, T1, nn);
    if (na) {
      double *Pm = (double *)malloc(sizeof(double) * na * nn);
      trsm_rlT(na, nn, Lk, nf, P + nn, nf);                         /* W */
      gemm(0, 0, na, nn, nn, 1.0, Lk + nn, nf, T1, nn, 0.0, Pm, na); /* L_AN T1 */
      /* Uk -= L_AN W'^T + W' L_AN^T, W' = W - Pm/2 ; then G_AN = W - Pm */
      for (int64_t j = 0; j < nn; ++j)
        for (int64_t i = 0; i < na; ++i) Pm[i + j * na] = P[nn + i + j * nf] - 0.5 * Pm[i + j * na]; /* W' */
      for (int64_t j = 0; j < na; ++j)
        for (int64_t i = j; i < na; ++i) {
          double v = 0.0;
          for (int64_t p = 0; p < nn; ++p)
            v += Lk[nn + i + p * nf] * Pm[j + p * na] + Pm[i + p * na] * Lk[nn + j + p * nf];
          Uk[i + j * na] -= v;
        }
      for (int64_t j = 0; j < nn; ++j)
        for (int64_t i = 0; i < na; ++i) P[nn + i + j * nf] = 2.0 * Pm[i + j * na] - P[nn + i + j * nf]; /* 2W'-W = W-Pm */
      free(Pm);
    }
    put_lower(nn, T1, P, nf);
    free(T1);
  }
}

static void hess_up(const orc_sym *s, const double *L, double *u, double *upd) { hess_up_m(s, L, u, upd, 0); }

/* root->leaves half: panel holds (G_NN, Q); result Z = 𝐋^-T [G_NN Q^T; Q Z_AA] 𝐋^-1 on V */
static void hess_down_m(const orc_sym *s, const double *L, double *u, double *upd, const unsigned char *mask) {
  for (int64_t k = s->nsn - 1; k >= 0; --k) {
    if (mask && !mask[k]) continue;
    int64_t nn = NN(s, k), nf = NF(s, k), na = nf - nn, p = s->snpar[k];
    const double *Lk = L + s->blkptr[k];
    double *P = u + s->blkptr[k], *Uk = upd + s->updptr[k];
    if (p >= 0) gather_sep(s, k, u + s->blkptr[p], upd + s->updptr[p], Uk);
    double *M = (double *)malloc(sizeof(double) * nn * nn);
    symfull(nn, P, nf, M);
    if (na) {
      double *Zf = (double *)malloc(sizeof(double) * na * na);
      double *Qp = (double *)malloc(sizeof(double) * na * nn);
      symfull(na, Uk, na, Zf);
      for (int64_t j = 0; j < nn; ++j)
        for (int64_t i = 0; i < na; ++i) Qp[i + j * na] = P[nn + i + j * nf];
      gemm(0, 0, na, nn, na, -0.5, Zf, na, Lk + nn, nf, 1.0, Qp, na); /* Q' = Q - Z_AA L_AN / 2 */
      gemm(1, 0, nn, nn, na, -1.0, Lk + nn, nf, Qp, na, 1.0, M, nn);
      gemm(1, 0, nn, nn, na, -1.0, Qp, na, Lk + nn, nf, 1.0, M, nn);
      for (int64_t j = 0; j < nn; ++j)
        for (int64_t i = 0; i < na; ++i) P[nn + i + j * nf] = 2.0 * Qp[i + j * na] - P[nn + i + j * nf];
      trsm_rlN(na, nn, Lk, nf, P + nn, nf);
      free(Zf);
      free(Qp);
    }
    trsm_llT(nn, nn, Lk, nf, M, nn);
    trsm_rlN(nn, nn, Lk, nf, M, nn);
    put_lower(nn, M, P, nf);
    free(M);
  }
}
static void hess_down(const orc_sym *s, const double *L, double *u, double *upd) { hess_down_m(s, L, u, upd, 0); }

/* inverse of hess_down: Z (on V) -> (G_NN, Q).  Clique-local once Z_AA has been gathered. */
static void hess_down_inv(const orc_sym *s, const double *L, double *u, double *upd) {
  gather_all(s, u, upd);
  for (int64_t k = 0; k < s->nsn; ++k) {
    int64_t nn = NN(s, k), nf = NF(s, k), na = nf - nn;
    const double *Lk = L + s->blkptr[k];
    double *P = u + s->blkptr[k], *Uk = upd + s->updptr[k];
    double *M = (double *)malloc(sizeof(double) * nn * nn);
    symfull(nn, P, nf, M);
    trmm_llT(nn, nn, Lk, nf, M, nn); /* L^T Z_NN */
    trmm_rlN(nn, nn, Lk, nf, M, nn); /* L^T Z_NN L */
    if (na) {
      double *Zf = (double *)malloc(sizeof(double) * na * na);
      double *Qpp = (double *)malloc(sizeof(double) * na * nn);
      symfull(na, Uk, na, Zf);
      trmm_rlN(na, nn, Lk, nf, P + nn, nf); /* Z_AN L_NN */
      for (int64_t j = 0; j < nn; ++j)
        for (int64_t i = 0; i < na; ++i) Qpp[i + j * na] = P[nn + i + j * nf];
      gemm(0, 0, na, nn, na, 0.5, Zf, na, Lk + nn, nf, 1.0, Qpp, na); /* Q'' */
      gemm(1, 0, nn, nn, na, 1.0, Lk + nn, nf, Qpp, na, 1.0, M, nn);
      gemm(1, 0, nn, nn, na, 1.0, Qpp, na, Lk + nn, nf, 1.0, M, nn);
      for (int64_t j = 0; j < nn; ++j)
        for (int64_t i = 0; i < na; ++i) P[nn + i + j * nf] = 2.0 * Qpp[i + j * na] - P[nn + i + j * nf]; /* Q */
      free(Zf);
      free(Qpp);
    }
    put_lower(nn, M, P, nf);
    free(M);
  }
}

/* inverse of hess_up: (G_NN, G_AN) -> U */
static void hess_up_inv(const orc_sym *s, const double *L, double *u, double *upd) {
  for (int64_t k = 0; k < s->nsn; ++k) {
    int64_t nn = NN(s, k), nf = NF(s, k), na = nf - nn;
    const double *Lk = L + s->blkptr[k];
    double *P = u + s->blkptr[k], *Uk = upd + s->updptr[k];
    double *G = (double *)malloc(sizeof(double) * nn * nn);
    symfull(nn, P, nf, G);
    memset(Uk, 0, sizeof(double) * na * na);
    if (na) {
      double *Vv = (double *)malloc(sizeof(double) * na * nn);
      for (int64_t j = 0; j < nn; ++j)
        for (int64_t i = 0; i < na; ++i) Vv[i + j * na] = P[nn + i + j * nf];
      gemm(0, 0, na, nn, nn, 0.5, Lk + nn, nf, G, nn, 1.0, Vv, na); /* V = G_AN + L_AN G_NN / 2 */
      for (int64_t j = 0; j < na; ++j)
        for (int64_t i = j; i < na; ++i) {
          double v = 0.0;
          for (int64_t p = 0; p < nn; ++p)
            v += Vv[i + p * na] * Lk[nn + j + p * nf] + Lk[nn + i + p * nf] * Vv[j + p * na];
          Uk[i + j * na] = -v;
        }
      /* F_AN = (L_AN G_NN + G_AN) L_NN^T = (2V - G_AN) L_NN^T */
      for (int64_t j = 0; j < nn; ++j)
        for (int64_t i = 0; i < na; ++i) P[nn + i + j * nf] = 2.0 * Vv[i + j * na] - P[nn + i + j * nf];
      trmm_rlT(na, nn, Lk, nf, P + nn, nf);
      free(Vv);
    }
    trmm_llN(nn, nn, Lk, nf, G, nn);
    trmm_rlT(nn, nn, Lk, nf, G, nn);
    put_lower(nn, G, P, nf);
    free(G);
    /* U = F - children,  G_AA = children_AA - (...) */
    add_children(s, k, upd, P, Uk, 1.0);
    /* the children's contribution must be SUBTRACTED from the panel but ADDED to Uk:
     * redo the panel part with the right sign */
    for (int64_t q = s->chptr[k]; q < s->chptr[k + 1]; ++q) {
      int64_t c = s->chidx[q];
      int64_t nac = NF(s, c) - NN(s, c);
      const int32_t *rel = s->relidx + s->sepptr[c];
      const double *Uc = upd + s->updptr[c];
      for (int64_t j = 0; j < nac; ++j) {
        int64_t rj = rel[j];
        if (rj >= nn) break;
        for (int64_t i = j; i < nac; ++i) P[rel[i] + rj * nf] -= 2.0 * Uc[i + j * nac];
      }
    }
  }
}

/* scale the AN block of every clique: mode 0: G_AN <- R^T G_AN ; 1: <- R G_AN ; 2: <- R^-T ; 3: <- R^-1;
 * 4: <- Y_AA G_AN (full symmetric, from yaa) ; 5: <- Y_AA^-1 G_AN (via fac) */
static void scale_an_m(const orc_sym *s, double *u, const double *yaa, const double *fac, int mode,
                       const unsigned char *mask) {
  for (int64_t k = 0; k < s->nsn; ++k) {
    if (mask && !mask[k]) continue;
    int64_t nn = NN(s, k), nf = NF(s, k), na = nf - nn;
    if (!na) continue;
    double *B = u + s->blkptr[k] + nn;
    const double *R = fac ? fac + s->updptr[k] : 0;
    switch (mode) {
      case 0: trmm_llT(na, nn, R, na, B, nf); break;
      case 1: trmm_llN(na, nn, R, na, B, nf); break;
      case 2: trsm_llT(na, nn, R, na, B, nf); break;
      case 3: trsm_llN(na, nn, R, na, B, nf); break;
      case 4: {
        double *Yf = (double *)malloc(sizeof(double) * na * na);
        double *T = (double *)malloc(sizeof(double) * na * nn);
        symfull(na, yaa + s->updptr[k], na, Yf);
        gemm(0, 0, na, nn, na, 1.0, Yf, na, B, nf, 0.0, T, na);
        for (int64_t j = 0; j < nn; ++j)
          for (int64_t i = 0; i < na; ++i) B[i + j * nf] = T[i + j * na];
        free(Yf);
        free(T);
      } break;
      case 5: trsm_llN(na, nn, R, na, B, nf); trsm_llT(na, nn, R, na, B, nf); break;
    }
  }
}

static void scale_an(const orc_sym *s, double *u, const double *yaa, const double *fac, int mode) {
  scale_an_m(s, u, yaa, fac, mode, 0);
}

/* hessian(L, Y, U, adj, inv) for one matrix U (in place).
 * adj: 0 = G, 1 = G^adj, 2 = both (None in the reference).  work: 3*updptr[nsn] doubles. */
int orc_hessian(const orc_sym *s, const double *L, const double *Y, double *u, int adj, int inv,
                double *work) {
  int64_t ul = s->updptr[s->nsn];
  double *upd = work, *yaa = work + ul, *fac = work + 2 * ul;
  int need_fac = !(adj == 2 && inv == 0);
  int info = prepare_yaa(s, Y, yaa, need_fac ? fac : 0);
  if (info) return info;
  if (!inv) {
    if (adj == 0) { hess_up(s, L, u, upd); scale_an(s, u, yaa, fac, 0); }
    else if (adj == 1) { scale_an(s, u, yaa, fac, 1); hess_down(s, L, u, upd); }
    else { hess_up(s, L, u, upd); scale_an(s, u, yaa, fac, 4); hess_down(s, L, u, upd); }
  } else {
    if (adj == 0) { scale_an(s, u, yaa, fac, 2); hess_up_inv(s, L, u, upd); }
    else if (adj == 1) { hess_down_inv(s, L, u, upd); scale_an(s, u, yaa, fac, 3); }
    else { hess_down_inv(s, L, u, upd); scale_an(s, u, yaa, fac, 5); hess_up_inv(s, L, u, upd); }
  }
  return 0;
}

/* Pieces of hessian(adj=False) exposed for the multi-GPU host-logic tests (tests/test_distributed.py):
 * orc_prepare_fac: yaa <- Y[A_k,A_k], fac <- chol(yaa);  orc_hess_g_masked: the leaves->root G sweep
 * (including the R^T scaling of the AN blocks) restricted to the cliques with mask[k] != 0.  Running it
 * over a partition of the cliques in any order that respects the tree gives exactly orc_hessian(adj=0). */
int orc_prepare_fac(const orc_sym *s, const double *Y, double *yaa, double *fac) { return prepare_yaa(s, Y, yaa, fac); }
void orc_hess_g_masked(const orc_sym *s, const double *L, const double *fac, double *u, double *upd,
                       const unsigned char *mask) {
  hess_up_m(s, L, u, upd, mask);
  scale_an_m(s, u, 0, fac, 0, mask);
}

/* Masked pieces for the sharded factorisation / solve (tests/test_distributed.py): each acts on the cliques with
 * mask[k] != 0 only and takes the update / separator blocks of the others as they stand in upd (yaa). */
int orc_cholesky_masked(const orc_sym *s, double *x, double *upd, const unsigned char *mask) { return cholesky_m(s, x, upd, mask); }
int orc_projected_inverse_masked(const orc_sym *s, double *x, double *upd, const unsigned char *mask) {
  return projected_inverse_m(s, x, upd, mask);
}
int orc_prepare_fac_masked(const orc_sym *s, const double *Y, double *yaa, double *fac, const unsigned char *mask) {
  return prepare_yaa_m(s, Y, yaa, fac, mask);
}
/* the two halves of hessian(adj=None, inv=False): up (with the Y_AA scaling of the AN blocks) and down */
void orc_hess_up_masked(const orc_sym *s, const double *L, const double *yaa, double *u, double *upd, const unsigned char *mask) {
  hess_up_m(s, L, u, upd, mask);
  scale_an_m(s, u, yaa, 0, 4, mask);
}
void orc_hess_down_masked(const orc_sym *s, const double *L, double *u, double *upd, const unsigned char *mask) {
  hess_down_m(s, L, u, upd, mask);
}

/* supernodal triangular solve with a dense n x nrhs right-hand side in permuted row order.
 * trans = 0: B <- L^-1 B ; trans = 1: B <- L^-T B */
int orc_trsm(const orc_sym *s, const double *L, double *B, int64_t nrhs, int64_t ldb, int trans) {
  if (!trans) {
    for (int64_t k = 0; k < s->nsn; ++k) {
      int64_t nn = NN(s, k), nf = NF(s, k), f = s->snptr[k];
      const double *Lk = L + s->blkptr[k];
      const int32_t *rows = s->rowidx + s->rowptr[k];
      trsm_llN(nn, nrhs, Lk, nf, B + f, ldb);
      for (int64_t c = 0; c < nrhs; ++c)
        for (int64_t j = 0; j < nn; ++j) {
          double x = B[f + j + c * ldb];
          for (int64_t i = nn; i < nf; ++i) B[rows[i] + c * ldb] -= Lk[i + j * nf] * x;
        }
    }
  } else {
    for (int64_t k = s->nsn - 1; k >= 0; --k) {
      int64_t nn = NN(s, k), nf = NF(s, k), f = s->snptr[k];
      const double *Lk = L + s->blkptr[k];
      const int32_t *rows = s->rowidx + s->rowptr[k];
      for (int64_t c = 0; c < nrhs; ++c)
        for (int64_t j = 0; j < nn; ++j) {
          double v = 0.0;
          for (int64_t i = nn; i < nf; ++i) v += Lk[i + j * nf] * B[rows[i] + c * ldb];
          B[f + j + c * ldb] -= v;
        }
      trsm_llT(nn, nrhs, Lk, nf, B + f, ldb);
    }
  }
  return 0;
}

/* tr(XY) over V: diagonal once, every other stored entry twice */
double orc_dot(const orc_sym *s, const double *x, const double *y) {
  double acc = 0.0;
  for (int64_t k = 0; k < s->nsn; ++k) {
    int64_t nn = NN(s, k), nf = NF(s, k);
    const double *a = x + s->blkptr[k], *b = y + s->blkptr[k];
    for (int64_t j = 0; j < nn; ++j) {
      acc += a[j + j * nf] * b[j + j * nf];
      double t = 0.0;
      for (int64_t i = j + 1; i < nf; ++i) t += a[i + j * nf] * b[i + j * nf];
      acc += 2.0 * t;
    }
  }
  return acc;
}

/* sum of log of the diagonal entries (X.diag() + log + sum, solvers.py:395,925,934) */
double orc_logdiagsum(const orc_sym *s, const double *x) {
  double acc = 0.0;
  for (int64_t k = 0; k < s->nsn; ++k) {
    int64_t nn = NN(s, k), nf = NF(s, k);
    for (int64_t j = 0; j < nn; ++j) acc += log(x[s->blkptr[k] + j + j * nf]);
  }
  return acc;
}

/* dense helpers exposed for the Schur-complement leg (lapack.potrf/potrs, solvers.py:501,526) */
int orc_dense_potrf(int64_t n, double *A, int64_t lda) { return potrf_l(n, A, lda); }
void orc_dense_potrs(int64_t n, int64_t nrhs, const double *A, int64_t lda, double *B, int64_t ldb) {
  trsm_llN(n, nrhs, A, lda, B, ldb);
  trsm_llT(n, nrhs, A, lda, B, ldb);
}

/* H[i,j] for i>=j per misc.c:620-663 (SCMcolumn2): A_j column-sparse, V = S^-1[:,K_j] (n x |K_j|,
 * column-major), kl maps a matrix column index to its column in V.  Av is given in
 * coordinate form per constraint: ptr (m+1), row/col (permuted matrix coordinates, row>=col), val. */
void orc_scmcolumn2(int64_t m, int64_t n, double *H, const int64_t *ptr, const int64_t *row,
                    const int64_t *col, const double *val, const double *V, const int64_t *kl,
                    int64_t j) {
  for (int64_t i = j; i < m; ++i) H[j * m + i] = 0.0;
  for (int64_t p = ptr[j]; p < ptr[j + 1]; ++p) {
    double alpha = val[p];
    int64_t r = row[p], c = col[p];
    if (r != c) alpha *= 2;
    r = kl[r];
    c = kl[c];
    for (int64_t i = j; i < m; ++i)
      for (int64_t q = ptr[i]; q < ptr[i + 1]; ++q) {
        double beta = val[q];
        int64_t r1 = row[q], c1 = col[q];
        H[j * m + i] += alpha * beta * V[n * r + r1] * V[n * c + c1];
        if (r1 != c1) H[j * m + i] += alpha * beta * V[n * r + c1] * V[n * c + r1];
      }
  }
}

/* ------------------------------------------------------------------ CPU baseline helper
 * Columns j0..j1-1 of the (unfactored) Schur complement H_ij = <A_i, hessian(L, Y)(A_j)>: one Hessian
 * application per column exactly as the reference's loop does (solvers.py:479-487), the independent columns
 * spread over `nthreads` OpenMP threads, each with its own right-hand side and workspace.  Constraints: CSC over
 * blkval positions (cptr, cidx, cval); w = cval with off-diagonal entries doubled (Amap weights,
 * solvers.py:369-375).  H: m x (j1 - j0), column-major.  Returns 0 or the first nonzero orc_hessian code. */
int orc_schur_columns(const orc_sym *s, const double *L, const double *Y, int64_t m, const int64_t *cptr,
                      const int64_t *cidx, const double *cval, const double *w, int64_t j0, int64_t j1, double *H,
                      int nthreads, double *seconds) {
  const int64_t bl = s->blkptr[s->nsn], ul = s->updptr[s->nsn];
  int rc = 0;
  double t_begin = 0.0, t_end = 0.0;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
  {
    /* per-thread right-hand side and workspace, touched before the clock starts (a long-running solver pays the
     * page faults of its workspaces once, not per column) */
    const size_t nu = (size_t)(bl > 0 ? bl : 1), nw = (size_t)(3 * ul > 0 ? 3 * ul : 1);
    double *u = (double *)malloc(sizeof(double) * nu);
    double *work = (double *)malloc(sizeof(double) * nw);
    memset(u, 0, sizeof(double) * nu);
    memset(work, 0, sizeof(double) * nw);
#pragma omp barrier
#pragma omp master
    t_begin = omp_get_wtime();
#pragma omp for schedule(dynamic, 1)
    for (int64_t j = j0; j < j1; ++j) {
      memset(u, 0, sizeof(double) * (size_t)bl);
      for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) u[cidx[e]] = cval[e];
      int r = orc_hessian(s, L, Y, u, 2, 0, work);
      if (r) {
#pragma omp critical
        if (!rc) rc = r;
      }
      double *h = H + (j - j0) * m;
      for (int64_t i = 0; i < m; ++i) {
        double acc = 0.0;
        for (int64_t e = cptr[i]; e < cptr[i + 1]; ++e) acc += w[e] * u[cidx[e]];
        h[i] = acc;
      }
    }
#pragma omp master
    t_end = omp_get_wtime();
    free(u);
    free(work);
  }
  if (seconds) *seconds = t_end - t_begin;
  return rc;
}
