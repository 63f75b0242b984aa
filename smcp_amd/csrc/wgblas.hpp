// Workgroup-cooperative dense routines on column-major fp64 matrices (generic-size path).
//
// Every routine is called by ALL threads of a workgroup with identical arguments, begins by
// assuming its inputs are visible (caller synchronised) and ends with a __syncthreads(), so
// routines can be chained.  Pointers may refer to HBM or LDS (flat addressing).
// These are the any-size fallbacks; the LDS/MFMA specialisations live in front_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace smcp {
namespace wg {

#define SMCP_TID ((int)threadIdx.x)
#define SMCP_NT ((int)blockDim.x)

// in-place lower Cholesky of the n x n matrix A; returns 0 or j+1 (uniform across the block)
__device__ inline int potrf(int n, double* A, int64_t lda) {
  for (int j = 0; j < n; ++j) {
    double d = A[j + j * lda];
    if (!(d > 0.0)) return j + 1;
    __syncthreads();
    double sd = sqrt(d), r = 1.0 / sd;
    for (int i = j + SMCP_TID; i < n; i += SMCP_NT) A[i + j * lda] = (i == j) ? sd : A[i + j * lda] * r;
    __syncthreads();
    int rem = n - j - 1;  // trailing (rem x rem) lower update
    for (int e = SMCP_TID; e < rem * rem; e += SMCP_NT) {
      int i = e % rem, k = e / rem;
      if (i >= k) A[(j + 1 + i) + (j + 1 + k) * lda] -= A[(j + 1 + i) + j * lda] * A[(j + 1 + k) + j * lda];
    }
    __syncthreads();
  }
  return 0;
}

// B (m x n) <- B L^-T   (L n x n lower): X L^T = B
__device__ inline void trsm_rlT(int m, int n, const double* L, int64_t ldl, double* B, int64_t ldb) {
  for (int j = 0; j < n; ++j) {
    double r = 1.0 / L[j + j * ldl];
    for (int i = SMCP_TID; i < m; i += SMCP_NT) B[i + j * ldb] *= r;
    __syncthreads();
    int rem = n - j - 1;
    for (int e = SMCP_TID; e < m * rem; e += SMCP_NT) {
      int i = e % m, k = j + 1 + e / m;
      B[i + k * ldb] -= B[i + j * ldb] * L[k + j * ldl];
    }
    __syncthreads();
  }
}
// B (m x n) <- B L^-1 : X L = B
__device__ inline void trsm_rlN(int m, int n, const double* L, int64_t ldl, double* B, int64_t ldb) {
  for (int j = n - 1; j >= 0; --j) {
    double r = 1.0 / L[j + j * ldl];
    for (int i = SMCP_TID; i < m; i += SMCP_NT) B[i + j * ldb] *= r;
    __syncthreads();
    for (int e = SMCP_TID; e < m * j; e += SMCP_NT) {
      int i = e % m, k = e / m;
      B[i + k * ldb] -= B[i + j * ldb] * L[j + k * ldl];
    }
    __syncthreads();
  }
}
// B (m x n) <- L^-1 B  (L m x m lower)
__device__ inline void trsm_llN(int m, int n, const double* L, int64_t ldl, double* B, int64_t ldb) {
  for (int j = 0; j < m; ++j) {
    double r = 1.0 / L[j + j * ldl];
    for (int c = SMCP_TID; c < n; c += SMCP_NT) B[j + c * ldb] *= r;
    __syncthreads();
    int rem = m - j - 1;
    for (int e = SMCP_TID; e < rem * n; e += SMCP_NT) {
      int i = j + 1 + e % rem, c = e / rem;
      B[i + c * ldb] -= L[i + j * ldl] * B[j + c * ldb];
    }
    __syncthreads();
  }
}
// B (m x n) <- L^-T B
__device__ inline void trsm_llT(int m, int n, const double* L, int64_t ldl, double* B, int64_t ldb) {
  for (int j = m - 1; j >= 0; --j) {
    double r = 1.0 / L[j + j * ldl];
    for (int c = SMCP_TID; c < n; c += SMCP_NT) B[j + c * ldb] *= r;
    __syncthreads();
    for (int e = SMCP_TID; e < j * n; e += SMCP_NT) {
      int i = e % j, c = e / j;
      B[i + c * ldb] -= L[j + i * ldl] * B[j + c * ldb];
    }
    __syncthreads();
  }
}

// element accessors -------------------------------------------------------------------
struct Mat {  // plain column-major
  const double* p; int64_t ld;
  __device__ double operator()(int i, int j) const { return p[i + j * ld]; }
};
struct MatT {  // transposed view
  const double* p; int64_t ld;
  __device__ double operator()(int i, int j) const { return p[j + i * ld]; }
};
struct SymL {  // symmetric, lower triangle stored
  const double* p; int64_t ld;
  __device__ double operator()(int i, int j) const { return i >= j ? p[i + j * ld] : p[j + i * ld]; }
};
struct LowT {  // lower-triangular matrix (zeros above the diagonal)
  const double* p; int64_t ld;
  __device__ double operator()(int i, int j) const { return i >= j ? p[i + j * ld] : 0.0; }
};
struct LowTT {  // transpose of a lower-triangular matrix
  const double* p; int64_t ld;
  __device__ double operator()(int i, int j) const { return j >= i ? p[j + i * ld] : 0.0; }
};

// C (m x n) = beta*C + alpha * A(m x k) * B(k x n); lower_only: only i>=j is touched
template <class TA, class TB>
__device__ inline void gemm(int m, int n, int k, double alpha, TA A, TB B, double beta, double* C,
                            int64_t ldc, bool lower_only = false) {
  for (int e = SMCP_TID; e < m * n; e += SMCP_NT) {
    int i = e % m, j = e / m;
    if (lower_only && i < j) continue;
    double acc = 0.0;
    for (int p = 0; p < k; ++p) acc += A(i, p) * B(p, j);
    double c = (beta == 0.0) ? 0.0 : beta * C[i + j * ldc];
    C[i + j * ldc] = c + alpha * acc;
  }
  __syncthreads();
}

__device__ inline void copy(int m, int n, const double* A, int64_t lda, double* B, int64_t ldb,
                            double alpha = 1.0) {
  for (int e = SMCP_TID; e < m * n; e += SMCP_NT) {
    int i = e % m, j = e / m;
    B[i + j * ldb] = alpha * A[i + j * lda];
  }
  __syncthreads();
}
__device__ inline void copy_lower(int n, const double* A, int64_t lda, double* B, int64_t ldb) {
  for (int e = SMCP_TID; e < n * n; e += SMCP_NT) {
    int i = e % n, j = e / n;
    if (i >= j) B[i + j * ldb] = A[i + j * lda];
  }
  __syncthreads();
}
// F (n x n full, ld n) <- symmetric expansion of the lower triangle of A
__device__ inline void symfull(int n, const double* A, int64_t lda, double* F) {
  for (int e = SMCP_TID; e < n * n; e += SMCP_NT) {
    int i = e % n, j = e / n;
    F[i + j * n] = i >= j ? A[i + j * lda] : A[j + i * lda];
  }
  __syncthreads();
}
__device__ inline void set_identity(int n, double* A, int64_t lda) {
  for (int e = SMCP_TID; e < n * n; e += SMCP_NT) {
    int i = e % n, j = e / n;
    A[i + j * lda] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
}
__device__ inline void zero(int64_t len, double* A) {
  for (int64_t e = SMCP_TID; e < len; e += SMCP_NT) A[e] = 0.0;
  __syncthreads();
}

}  // namespace wg
}  // namespace smcp
