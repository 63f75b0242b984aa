// kkt_qr: the QR-based KKT solver of the reference (src/python/solvers.py:413-475 and 1843-1905) on the device.
//
// The reference stacks the half-Hessian images svec(G(A_j)) of all m constraints into a |V| x m matrix At, takes its
// Householder QR (lapack.geqrf) and solves the KKT system through Q and R instead of the Cholesky factor of the
// Schur complement H = At^T At, which squares the condition number.  Here the stack is what the Gram formulation of
// kkt_chol already builds (one leaves->root sweep per constraint, D.ustack: row j = G(A_j) in blkval layout), and
// the QR is computed where the chip is fast, as Cholesky-QR iterations on the MFMA SYRK of the Gram path:
//
//     G = W^T W (weighted: H's inner product),  R_p = chol(G)^T,  W <- W R_p^-1,  R <- R_p R      (one pass)
//
// Two passes (CholeskyQR2) give Q orthonormal to rounding when kappa(At) < ~1e8.  When chol(G) breaks down -- the
// regime where kkt_chol fails -- the first pass is repeated with a shifted Gram matrix G + s I (shifted CholeskyQR3,
// Fukaya et al., SIAM J. Sci. Comput. 42 (2020)): the shift makes the factorisation go through, the pass still
// reduces the condition number to where two plain passes finish the job.  Q is kept explicitly (it overwrites the
// stack), R as its lower-triangular transpose; solve_ applies Q^T, R^-T, R^-1 and Q exactly where the reference
// applies ormqr / trtrs (solvers.py:443-459).  Normalisation: the reference scales the diagonal entries by
// 1/sqrt(2) and stores off-diagonal entries once, so its R^T R = H / 2; here the inner product carries the weights
// (1 on the diagonal, 2 below it), R^T R = H, and the 0.5 of solvers.py:453 disappears.

namespace {

constexpr int QR_JB = 8;       // rows of the stack per register block of k_stack_trsm

// W <- W * R^-1 in place BY SUBSTITUTION (multiplying by an explicit inverse would leave Q R = At only to
// kappa(R) * eps, which is exactly what the QR path is there to avoid).  W is the m x len stack (row j at W + j * ldw);
// Rp the upper-triangular factor, row-major with leading dimension ldr = 8 * nblk, reciprocals on its diagonal,
// padded with a unit diagonal.
// One workgroup = 64 P stack positions (P per lane) x 8 waves; the m rows are cut into blocks of 8 dealt cyclically to
// the waves, and every wave keeps ITS blocks of the columns in registers for the whole kernel (NBW blocks = 8 NBW P
// accumulators).  Right-looking: for block J = 0, 1, ... its owner finishes it (8-step substitution with the
// diagonal block, in registers), stores the rows and broadcasts them through LDS; every wave then subtracts their
// contribution from its own later blocks (64 P FMAs per block pair).  The entries of R are wave-uniform scalar loads
// and they are what bounds the kernel: R (m^2 doubles) does not fit the 16 KB scalar cache and the workgroups of a CU
// walk it at different phases, so P positions per lane (P = 4 for m <= 128, else 2) divide that traffic by P and the
// eight-wave workgroups halve the number of phases per CU (1.72 ms -> see DESIGN for one pass on synth50k).
// LDS carries only the 8 finished rows (double buffered: one barrier per block step).
template <int NBW, int P, int QR_NW>
__global__ void __launch_bounds__(64 * QR_NW) k_stack_trsm(double* W, int64_t ldw, int64_t len, int m,
                                                           const double* __restrict__ Rp, int ldr, int fake) {
  __shared__ double sq[2][QR_JB][64 * P];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t p0 = (int64_t)blockIdx.x * (64 * P) + lane;       // this lane's positions: p0 + 64 t
  const int nblk = ldr / QR_JB;
  double acc[NBW][QR_JB][P];
#pragma unroll
  for (int s = 0; s < NBW; ++s) {
    const int K = s * QR_NW + wave;
#pragma unroll
    for (int jj = 0; jj < QR_JB; ++jj) {
      const int row = K * QR_JB + jj;
#pragma unroll
      for (int t = 0; t < P; ++t) acc[s][jj][t] = (row < m && p0 + 64 * t < len) ? W[(int64_t)row * ldw + p0 + 64 * t] : 0.0;
    }
  }
  for (int J = 0; J < nblk; ++J) {
    const int j0 = J * QR_JB, buf = J & 1;
    if ((J & (QR_NW - 1)) == wave) {
      const int sJ = J / QR_NW;
#pragma unroll
      for (int s = 0; s < NBW; ++s)
        if (s == sJ) {
          double q[QR_JB][P];
#pragma unroll
          for (int jj = 0; jj < QR_JB; ++jj) {
            const double rinv = Rp[(int64_t)(j0 + jj) * ldr + j0 + jj];     // the packed diagonal holds reciprocals
#pragma unroll
            for (int t = 0; t < P; ++t) {
              double v = acc[s][jj][t];
#pragma unroll
              for (int ii = 0; ii < jj; ++ii) v -= q[ii][t] * Rp[(int64_t)(j0 + ii) * ldr + j0 + jj];
              q[jj][t] = v * rinv;
            }
          }
#pragma unroll
          for (int jj = 0; jj < QR_JB; ++jj)
#pragma unroll
            for (int t = 0; t < P; ++t) {
              sq[buf][jj][lane + 64 * t] = q[jj][t];
              if (j0 + jj < m && p0 + 64 * t < len) W[(int64_t)(j0 + jj) * ldw + p0 + 64 * t] = q[jj][t];
            }
        }
    }
    __syncthreads();
    if (J + 1 >= nblk) break;
#pragma unroll
    for (int s = 0; s < NBW; ++s) {
      const int K = s * QR_NW + wave;
      if (K > J && K < nblk) {
        const double* r = fake ? Rp : Rp + (int64_t)j0 * ldr + K * QR_JB;
        // two halves of four rows of R: 32 wave-uniform doubles (64 SGPRs) in flight at a time -- all 64 at once
        // overflow the scalar register file and the spills land in the inner loop
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int ii = h * (QR_JB / 2); ii < (h + 1) * (QR_JB / 2); ++ii) {
            double q[P];
#pragma unroll
            for (int t = 0; t < P; ++t) q[t] = sq[buf][ii][lane + 64 * t];
#pragma unroll
            for (int kk = 0; kk < QR_JB; ++kk) {
              const double rv = r[(int64_t)ii * ldr + kk];
#pragma unroll
              for (int t = 0; t < P; ++t) acc[s][kk][t] -= q[t] * rv;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
}
// The same substitution with the block updates on the matrix cores (v_mfma_f64_16x16x4: the same peak as the vector
// fp64 pipe on gfx950, but the entries of R become a VECTOR operand, 16 consecutive doubles of a row per load, pipelined under vmcnt --
// the scalar loads of k_stack_trsm cannot be: they return out of order, so every wait is a wait for all of them).
// One workgroup = 64 stack positions x 4 waves; the rows are cut into blocks of 16 dealt cyclically to the waves;
// a wave keeps its blocks as MFMA accumulators D[position][row] (4 position groups of 16 x NB blocks x 4 registers).
// Block step J: the owner spills its block through LDS into "lane = position" form, every lane runs the 16-step
// substitution of its position in registers (diagonal block of R: wave-uniform loads, reciprocals precomputed),
// stores the finished rows (coalesced) and leaves them in LDS; after the barrier every wave subtracts
// Q_J R[J, K] from its later blocks K: per block pair 4 k-steps x 4 position groups = 16 MFMAs, the R operand of a
// k-step loaded once and shared by the four groups.  Rp: ldr a multiple of 16 here.
constexpr int QR_MB = 16;
template <int NB>
__global__ void __launch_bounds__(256, (NB <= 1 ? 4 : (NB <= 2 ? 3 : 2))) k_stack_trsm_mfma(double* W, int64_t ldw, int64_t len, int m,
                                                         const double* __restrict__ Rp, int ldr, int nblk) {
  __shared__ double sT[QR_MB][64 + 1];          // the owner's block, [row][position]
  __shared__ double sQ[2][QR_MB][64];           // finished rows of the block step, double buffered
  __shared__ double sR[2][QR_MB][QR_MB + 1];    // diagonal block of R of the block step (reciprocals on its diagonal)
  const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t p0 = (int64_t)blockIdx.x * 64;      // nblk: blocks of 16 rows of this call (a panel of a larger factor: Rp points at its
                                                    // diagonal block, ldr stays the stride of the whole packed factor)
  d4 D[NB][4];
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    const int K = s * 4 + wave;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = K * QR_MB + kq + 4 * r;
        const int64_t p = p0 + 16 * g + l15;
        D[s][g][r] = (row < m && p < len) ? W[(int64_t)row * ldw + p] : 0.0;
      }
  }
  // diagonal block of R for block step J: fetched by its owner-to-be one step ahead (vector loads, 4 entries per lane)
  // into sR[J & 1]; written and read by the same wave
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = lane + 64 * t;
      sR[0][e >> 4][e & 15] = Rp[(int64_t)(e >> 4) * ldr + (e & 15)];
    }
  }
  for (int J = 0; J < nblk; ++J) {
    const int j0 = J * QR_MB, buf = J & 1;
    // R operands of this step's updates: they do not depend on the rows being finished, so the loads go out before
    // the owner's substitution and the barrier
    double b[NB][4];
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const int K = s * 4 + wave;
      if (K > J && K < nblk) {
        const double* r = Rp + (int64_t)(j0 + kq) * ldr + K * QR_MB + l15;
#pragma unroll
        for (int st = 0; st < 4; ++st) b[s][st] = r[(int64_t)(4 * st) * ldr];
      } else {
#pragma unroll
        for (int st = 0; st < 4; ++st) b[s][st] = 0.0;
      }
    }
    if ((J & 3) == wave) {
      const int sJ = J >> 2;
#pragma unroll
      for (int s = 0; s < NB; ++s)
        if (s == sJ) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) sT[kq + 4 * r][16 * g + l15] = D[s][g][r];
        }
      // same wave wrote and reads: the LDS operations of a wave are executed in order.  Substitution of this lane's
      // position in registers
      const int64_t p = p0 + lane;
      double v[QR_MB];
#pragma unroll
      for (int jj = 0; jj < QR_MB; ++jj) v[jj] = sT[jj][lane];
#pragma unroll
      for (int jj = 0; jj < QR_MB; ++jj) {          // right-looking: 15 - jj independent updates per finished entry
        const double qv = v[jj] * sR[buf][jj][jj];  // reciprocal
#pragma unroll
        for (int kk = jj + 1; kk < QR_MB; ++kk) v[kk] -= qv * sR[buf][jj][kk];
        sQ[buf][jj][lane] = qv;
        __builtin_amdgcn_sched_barrier(0);          // one row of R in flight at a time
      }
      if (p < len) {
        double* wp = W + (int64_t)j0 * ldw + p;
#pragma unroll 4
        for (int jj = 0; jj < QR_MB; ++jj)
          if (j0 + jj < m) wp[(int64_t)jj * ldw] = sQ[buf][jj][lane];
      }
    }
    __syncthreads();
    if (J + 1 >= nblk) break;
    if (((J + 1) & 3) == wave) {
      const int jn = j0 + QR_MB;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int e = lane + 64 * t;
        sR[(J + 1) & 1][e >> 4][e & 15] = Rp[(int64_t)(jn + (e >> 4)) * ldr + jn + (e & 15)];
      }
    }
    double aq[4][4];                              // -Q_J as the A operand: [position group][k-step]
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int st = 0; st < 4; ++st) aq[g][st] = -sQ[buf][kq + 4 * st][16 * g + l15];
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const int K = s * 4 + wave;
      if (K > J && K < nblk) {
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int g = 0; g < 4; ++g) D[s][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[s][st], aq[g][st], D[s][g], 0, 0, 0);
      }
    }
  }
}
// Panels of a factor with more than 320 columns (k_stack_trsm_mfma keeps a panel's rows as accumulators: at most five
// blocks of 64 rows): before the substitution inside the panel of rows [c0, c0 + 16 ntile) the finished rows above it
// are eliminated,  W[j][p] -= sum_{i < c0} W[i][p] R[i][j]  -- a tall product on the matrix pipe: 64 positions per
// workgroup, wave w the column tiles w, w + 4, ... (NT per wave), every finished row read once per panel.
template <int NT>
__global__ void __launch_bounds__(256) k_stack_gemm_panel(double* W, int64_t ldw, int64_t len, int c0, int ntile, int m,
                                                          const double* __restrict__ Rp, int ldr) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  d4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[t][g] = d4{0.0, 0.0, 0.0, 0.0};
  for (int i0 = 0; i0 < c0; i0 += 8) {            // two k-steps per trip: eight loads of W in flight
    double b[2][4], a[2][NT];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = i0 + 4 * h + kq;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t p = p0 + 16 * g + l15;
        b[h][g] = (i < c0 && p < len) ? W[(int64_t)i * ldw + p] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tile = wave + 4 * t;
        a[h][t] = (i < c0 && tile < ntile) ? Rp[(int64_t)i * ldr + c0 + 16 * tile + l15] : 0.0;
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[t][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[h][t], b[h][g], acc[t][g], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tile = wave + 4 * t;
    if (tile >= ntile) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int col = c0 + 16 * tile + kq + 4 * x;
        const int64_t p = p0 + 16 * g + l15;
        if (col < m && p < len) W[(int64_t)col * ldw + p] -= acc[t][g][x];
      }
  }
}
// Rp (row-major upper, ldr x ldr, RECIPROCALS on the diagonal, unit diagonal in the padding) <- transpose of the lower
// Cholesky factor T (m x m)
__global__ void k_qr_pack(const double* T, int m, int64_t ldt, double* Rp, int ldr) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ldr * ldr; e += gridDim.x * blockDim.x) {
    const int i = e / ldr, j = e % ldr;
    Rp[e] = (i < m && j < m) ? (j > i ? T[j + (int64_t)i * ldt] : (j == i ? 1.0 / T[i + (int64_t)i * ldt] : 0.0)) : (i == j ? 1.0 : 0.0);
  }
}

// r <- sw^2 * r  (the inner-product weights, folded into the vector once)
__global__ void k_weight_vec(const double* __restrict__ sw, double* r, int64_t len) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < len; p += (int64_t)gridDim.x * 256) r[p] *= sw[p] * sw[p];
}
// part[j * nchunk + c] = sum over the c-th chunk of Q[j] * rw for the four rows j = 4 blockIdx.y .. (rw = weighted r:
// inner products of the rows of Q with r; four rows share every load of rw)
constexpr int QR_DROWS = 4;
__global__ void __launch_bounds__(256) k_stack_dots(const double* __restrict__ Q, int64_t ldq, int64_t len, int m,
                                                    const double* __restrict__ rw, int64_t chunk, double* part, int nchunk) {
  __shared__ double red[4][QR_DROWS];
  const int j0 = blockIdx.y * QR_DROWS;
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(len, lo + chunk);
  double acc[QR_DROWS];
#pragma unroll
  for (int t = 0; t < QR_DROWS; ++t) acc[t] = 0.0;
  for (int64_t p = lo + threadIdx.x; p < hi; p += 256) {
    const double w = rw[p];
#pragma unroll
    for (int t = 0; t < QR_DROWS; ++t)
      if (j0 + t < m) acc[t] += w * Q[(int64_t)(j0 + t) * ldq + p];
  }
#pragma unroll
  for (int t = 0; t < QR_DROWS; ++t) {
    double a = acc[t];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = a;
  }
  __syncthreads();
  if (threadIdx.x < QR_DROWS && j0 + threadIdx.x < m)
    part[(int64_t)(j0 + threadIdx.x) * nchunk + blockIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// out[j] = sum_c part[j * nchunk + c]  (fixed order: deterministic)
__global__ void k_rows_sum(const double* part, int nchunk, double* out) {
  const int j = blockIdx.x, lane = threadIdx.x;
  double acc = 0.0;
  for (int c = lane; c < nchunk; c += 64) acc += part[(int64_t)j * nchunk + c];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) out[j] = acc;
}
// out = sum_j v[j] Q[j] - r   (out may alias r)
__global__ void __launch_bounds__(256) k_stack_comb(const double* __restrict__ Q, int64_t ldq, int64_t len, int m,
                                                    const double* __restrict__ v, const double* r, double* out) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < len; p += (int64_t)gridDim.x * 256) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int j = 0;
    for (; j + 4 <= m; j += 4) {
      a0 += v[j] * Q[(int64_t)j * ldq + p];
      a1 += v[j + 1] * Q[(int64_t)(j + 1) * ldq + p];
      a2 += v[j + 2] * Q[(int64_t)(j + 2) * ldq + p];
      a3 += v[j + 3] * Q[(int64_t)(j + 3) * ldq + p];
    }
    for (; j < m; ++j) a0 += v[j] * Q[(int64_t)j * ldq + p];
    out[p] = ((a0 + a1) + (a2 + a3)) - r[p];
  }
}

// ---- m x m helpers (small: one or a few workgroups) ------------------------------------------------------------
// T = A * B for lower-triangular A, B (column-major m x m): T(i, j) = sum_{k = j..i} A(i, k) B(k, j)
__global__ void k_qr_lower_mul(const double* A, const double* B, int m, int64_t ld, double* T) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < m * m; e += gridDim.x * blockDim.x) {
    const int i = e % m, j = e / m;
    double acc = 0.0;
    for (int k = j; k <= i; ++k) acc += A[i + (int64_t)k * ld] * B[k + (int64_t)j * ld];
    T[i + (int64_t)j * ld] = acc;
  }
}
// b <- L^-1 b (trans 0) or L^-T b (trans 1), one workgroup
__global__ void __launch_bounds__(256) k_qr_trsv(const double* L, int m, int64_t ldl, double* b, int trans) {
  if (trans) wg::trsm_llT(m, 1, L, ldl, b, m);
  else wg::trsm_llN(m, 1, L, ldl, b, m);
}
// A += (rel * trace(A)) I, one workgroup
__global__ void __launch_bounds__(256) k_qr_shift(double* A, int m, int64_t lda, double rel) {
  __shared__ double red[4];
  double t = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) t += A[i + (int64_t)i * lda];
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  const double s = rel * ((red[0] + red[1]) + (red[2] + red[3]));
  for (int i = threadIdx.x; i < m; i += 256) A[i + (int64_t)i * lda] += s;
}

// out[0] = max |A - I| over the lower triangle (one workgroup): how far the stack is from orthonormal
__global__ void __launch_bounds__(256) k_qr_deviation(const double* A, int m, int64_t lda, double* out) {
  __shared__ double red[4];
  double t = 0.0;
  for (int e = threadIdx.x; e < m * m; e += 256) {
    const int i = e % m, j = e / m;
    if (i >= j) t = fmax(t, fabs(A[i + (int64_t)j * lda] - (i == j ? 1.0 : 0.0)));
  }
  for (int off = 32; off > 0; off >>= 1) t = fmax(t, __shfl_down(t, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

constexpr int64_t QR_DOT_CHUNK = 16384;

int qr_alloc(csp_ctx* c) {
  DeviceCtx& D = c->D;
  const int64_t m = D.m, bl = c->S.blklen();
  const int64_t ldr = (m + QR_MB - 1) / QR_MB * QR_MB;
  const int64_t nchunk = (bl + QR_DOT_CHUNK - 1) / QR_DOT_CHUNK;
  // G | T | Lc | scratch (m x m each), Rp (ldr x ldr: the packed factor of the pass), r1 (bl), part (m * nchunk), xm, r2 (m each)
  const int64_t need = 4 * m * m + ldr * ldr + bl + m * nchunk + 2 * m + bl;     // ... and the weighted copy of r1 at the end
  if (D.qr_ws && D.qr_len >= need) return 0;
  if (D.qr_ws) { HIPCHK(hipFree(D.qr_ws)); D.bytes -= D.qr_len * 8; D.qr_ws = nullptr; }
  if (int rc = dev_alloc(&D.qr_ws, need, D.bytes)) return rc;
  D.qr_len = need;
  HIPCHK(hipMemset(D.qr_ws, 0, sizeof(double) * need));
  return 0;
}

}  // namespace

extern "C" {

int kkt_qr_factor(csp_ctx* c, const double* L, const double* Y, int64_t* passes_out, double* shift_out, void* stream) {
  if (int rc = ready(c)) return rc;
  DeviceCtx& D = c->D;
  const int64_t m = D.m, bl = c->S.blklen();
  if (!m || use_generic(c)) return SMCP_EINVAL;
  if (D.ns) return SMCP_EINVAL;                 // every constraint must be swept (kkt_set_tnzcols(0) before the constraints)
  if (D.max_rhs < 1 || D.ustack_cols < m) return SMCP_EINVAL;
  // one device holds the whole Q: under a subtree partition over more than one rank (csp_set_partition) this context's
  // factors are valid on its own cliques and the top only, and there are no kkt_qr_*_part forms
  if (c->xr_world > 1) return SMCP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  D.qr_valid = false;
  if (int rc = qr_alloc(c)) return rc;
  // The breakdown of chol(G) below DECIDES control flow (shifted pass, clean-up passes): its flag must be read back at
  // once even when the caller runs with deferred status (csp_lazy_status), where fetch_info only latches and returns 0
  // -- the loop would then go on with a half-factored T.  Eager for the duration of this call; a failure that survives
  // the retries is returned directly.
  struct EagerStatus {
    csp_ctx* c; bool was;
    explicit EagerStatus(csp_ctx* c_) : c(c_), was(c_->lazy_status) { c->lazy_status = false; }
    ~EagerStatus() { c->lazy_status = was; }
  } eager(c);
  const int64_t ldr8 = (m + QR_JB - 1) / QR_JB * QR_JB, ldr16 = (m + QR_MB - 1) / QR_MB * QR_MB;
  double* G = D.qr_ws;
  double* T = G + m * m;
  double* Lc = T + m * m;
  double* S2 = Lc + m * m;
  double* X = S2 + m * m;
  // the stack of half-Hessian images, exactly as the Gram formulation of kkt_chol builds it
  if (int rc = gram_prepare(c, L, Y, st)) return rc;
  for (int64_t jb = 0; jb < m; jb += D.max_rhs) {
    int nr = (int)std::min(D.max_rhs, m - jb);
    hess_up_fast(c, D.ustack + jb * bl, nr, bl, D.fac, 2, st, 0, D.kc_ptr ? jb : -1);
  }
  static int npass_env = -1;
  if (npass_env < 0) { const char* e = sw_str("SMCP_QR_PASSES"); npass_env = e ? std::max(1, atoi(e)) : 0; }
  int npass = npass_env ? npass_env : 2;
  double shift = 0.0;
  const int64_t range[2] = {0, bl};
  for (int pass = 0; pass < npass && pass < 8; ++pass) {
    if (int rc = gram_accumulate(c, 1, range, G, m, st)) return rc;
    if (pass == 0) { if (int f = fetch_info(c, st)) return f; }     // chol(Y_AA) failure inside the sweeps
    if (pass > 0 && pass == npass - 1 && !npass_env && npass < 5) {
      // last planned pass: its Gram matrix tells how orthonormal the previous pass left the stack; this pass squares
      // that deviation, so one more is planned when it would not reach rounding level
      double dev = 0.0;
      launch(c, KID_qr_small, k_qr_deviation, dim3(1), dim3(256), st, (const double*)G, (int)m, m, S2);
      HIPCHK(hipMemcpyAsync(&dev, S2, sizeof(double), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (!(dev < 1e-4)) ++npass;
    }
    int tries = 0;
    for (;;) {
      HIPCHK(hipMemcpyAsync(T, G, sizeof(double) * m * m, hipMemcpyDeviceToDevice, st));
      if (shift > 0.0) launch(c, KID_qr_small, k_qr_shift, dim3(1), dim3(256), st, T, (int)m, m, shift);
      const int rc = dense_potrf(c, T, m, m, stream);
      if (rc == 0) break;
      if (rc < 0) return rc;
      // breakdown: shift (relative to the trace, which bounds ||At||_2^2 from above) and, on the first pass, plan the
      // two clean-up passes of shifted CholeskyQR3
      if (++tries > 8) return rc;
      shift = shift > 0.0 ? shift * 100.0 : 1e-15 * (double)m;
      // a stack that still breaks down after several passes is rank deficient (more constraints than entries, exactly
      // dependent constraints): report it like a failed Cholesky instead of shifting for ever
      if (pass >= 4) return rc;
      if (!npass_env) npass = std::max(npass, pass + 3);
    }
    if (pass == 0 && shift_out) *shift_out = shift;
    shift = 0.0;                     // the next pass sees a better-conditioned stack and starts unshifted again
    static int fake = -1;      // timing experiment only, vector-FMA kernel (SMCP_QR_FAKE=1: every update reads the same block of R)
    if (fake < 0) { const char* e = sw_str("SMCP_QR_FAKE"); fake = e ? atoi(e) : 0; }
    static int vfma = -1;      // SMCP_QR_TRSM=fma: the vector-FMA kernel (k_stack_trsm) instead of the MFMA one
    if (vfma < 0) { const char* e = sw_str("SMCP_QR_TRSM"); vfma = (e && e[0] == 'f') ? 1 : 0; }
    const int64_t ldr = vfma ? ldr8 : ldr16;
    launch(c, KID_qr_small, k_qr_pack, dim3((unsigned)std::min<int64_t>(256, (ldr * ldr + 255) / 256)), dim3(256), st,
           (const double*)T, (int)m, m, X, (int)ldr);
    if (!vfma) {
      const dim3 grid((unsigned)((bl + 63) / 64)), blk(256);
      const int nb = (int)((ldr / QR_MB + 3) / 4);
      // panels of at most 256 rows (320 when the whole factor fits one): rows above a panel are eliminated by a tall
      // product first (k_stack_gemm_panel), then the substitution runs inside the panel
      const int PB = ldr <= 320 ? (int)ldr : 256;
      for (int c0 = 0; c0 < (int)ldr; c0 += PB) {
        const int pc = std::min(PB, (int)ldr - c0), rows = std::max(0, std::min(pc, (int)m - c0));
        if (!rows) break;
        if (c0 > 0) launch(c, KID_qr_rmul, k_stack_gemm_panel<4>, grid, blk, st, D.ustack, bl, bl, c0, pc / QR_MB, (int)m, (const double*)X, (int)ldr);
        double* Wp = D.ustack + (int64_t)c0 * bl;
        const double* Xp = X + (int64_t)c0 * ldr + c0;
        const int nbp = (pc / QR_MB + 3) / 4;
#define SMCP_TRSM_CASE(N) case N: launch(c, KID_qr_rmul, k_stack_trsm_mfma<N>, grid, blk, st, Wp, bl, bl, rows, Xp, (int)ldr, pc / QR_MB); break;
        switch (nbp) {
          SMCP_TRSM_CASE(1) SMCP_TRSM_CASE(2) SMCP_TRSM_CASE(3) SMCP_TRSM_CASE(4) SMCP_TRSM_CASE(5)
          default: return SMCP_ENOMEM;
        }
#undef SMCP_TRSM_CASE
      }
    } else {
      // SMCP_QR_P=2: two positions per lane (halves the scalar loads per FMA; measured no faster).  Eight waves per
      // workgroup and four positions per lane were measured too (1.75-2.4 ms per pass) and are not instantiated any more.
      if (m > 320) return SMCP_ENOMEM;            // the vector-FMA variant keeps all rows of a position in registers
      static int pv = -1;
      const int nwv = 4;
      if (pv < 0) { const char* e = sw_str("SMCP_QR_P"); pv = (e && e[0] == '2') ? 2 : 1; }
      const int nbw = (int)((ldr / QR_JB + nwv - 1) / nwv);
      int P = pv;
      if (nbw * P > 10) P = nbw > 5 ? 1 : 2;        // register budget: 8 NBW P accumulators per lane
      const dim3 grid((unsigned)((bl + 64 * P - 1) / (64 * P))), blk(64 * nwv);
#define SMCP_TRSM_CASE(N, PP, NWV) case (N * 8 + PP) * 16 + NWV: launch(c, KID_qr_rmul, k_stack_trsm<N, PP, NWV>, grid, blk, st, D.ustack, bl, bl, (int)m, (const double*)X, (int)ldr, fake); break;
      switch ((nbw * 8 + P) * 16 + nwv) {
        SMCP_TRSM_CASE(1, 1, 4) SMCP_TRSM_CASE(2, 1, 4) SMCP_TRSM_CASE(3, 1, 4) SMCP_TRSM_CASE(4, 1, 4) SMCP_TRSM_CASE(5, 1, 4)
        SMCP_TRSM_CASE(6, 1, 4) SMCP_TRSM_CASE(7, 1, 4) SMCP_TRSM_CASE(8, 1, 4) SMCP_TRSM_CASE(9, 1, 4) SMCP_TRSM_CASE(10, 1, 4)
        SMCP_TRSM_CASE(1, 2, 4) SMCP_TRSM_CASE(2, 2, 4) SMCP_TRSM_CASE(3, 2, 4) SMCP_TRSM_CASE(4, 2, 4) SMCP_TRSM_CASE(5, 2, 4)
        default: return SMCP_ENOMEM;
      }
#undef SMCP_TRSM_CASE
    }
    if (pass == 0) {
      HIPCHK(hipMemcpyAsync(Lc, T, sizeof(double) * m * m, hipMemcpyDeviceToDevice, st));
    } else {
      launch(c, KID_qr_small, k_qr_lower_mul, dim3((unsigned)std::min<int64_t>(256, (m * m + 255) / 256)), dim3(256), st,
             (const double*)Lc, (const double*)T, (int)m, m, S2);
      HIPCHK(hipMemcpyAsync(Lc, S2, sizeof(double) * m * m, hipMemcpyDeviceToDevice, st));
    }
  }
  HIPCHK(end_call(c));
  if (passes_out) *passes_out = npass;
  D.qr_valid = true;
  D.qr_L = L; D.qr_Y = Y;
  return 0;
}

int kkt_qr_solve(csp_ctx* c, const double* L, const double* Y, double kk, double* bx, double* by, void* stream) {
  if (int rc = ready(c)) return rc;
  DeviceCtx& D = c->D;
  const int64_t m = D.m, bl = c->S.blklen();
  if (!m || !D.qr_valid || D.qr_L != L || D.qr_Y != Y) return SMCP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t ldr = (m + QR_MB - 1) / QR_MB * QR_MB;
  const int nchunk = (int)((bl + QR_DOT_CHUNK - 1) / QR_DOT_CHUNK);
  double* Lc = D.qr_ws + 2 * m * m;
  double* r1 = D.qr_ws + 4 * m * m + ldr * ldr;
  double* part = r1 + bl;
  double* xm = part + m * nchunk;
  double* r2 = xm + m;
  // the half-Hessians below need chol(Y_AA), not only Y_AA (kkt_solve's full Hessian does with the latter): another
  // factorisation between kkt_qr_factor and this call (a line-search completion, say) leaves the cache with its own
  HIPCHK(zero_flag(c, st));
  prepare_yaa(c, Y, true, st);
  if (int rc = prep_lk_cached(c, L, Y, st)) return rc;
  HIPCHK(hipMemcpyAsync(r1, bx, sizeof(double) * bl, hipMemcpyDeviceToDevice, st));
  hessian_impl(c, L, r1, 1, bl, 0, 0, st);                                   // r1 = G(bx)           (solvers.py:444-447)
  // x = Q^T r1 in the weighted inner product (solvers.py:449-450); rw = sw^2 r1 in the partial-sum scratch's tail
  double* rw = D.qr_ws + D.qr_len - bl;
  HIPCHK(hipMemcpyAsync(rw, r1, sizeof(double) * bl, hipMemcpyDeviceToDevice, st));
  launch(c, KID_qr_small, k_weight_vec, dim3((unsigned)std::min<int64_t>(2048, (bl + 255) / 256)), dim3(256), st, (const double*)D.sw, rw, bl);
  launch(c, KID_qr_dots, k_stack_dots, dim3((unsigned)nchunk, (unsigned)((m + QR_DROWS - 1) / QR_DROWS)), dim3(256), st,
         (const double*)D.ustack, bl, bl, (int)m, (const double*)rw, QR_DOT_CHUNK, part, nchunk);
  launch(c, KID_qr_small, k_rows_sum, dim3((unsigned)m), dim3(64), st, (const double*)part, nchunk, xm);   // x = Q^T r1  (449-450)
  HIPCHK(hipMemcpyAsync(r2, by, sizeof(double) * m, hipMemcpyDeviceToDevice, st));
  launch(c, KID_qr_small, k_qr_trsv, dim3(1), dim3(256), st, (const double*)Lc, (int)m, m, r2, 0);        // r2 = R^-T by (452)
  launch(c, KID_vec_axpby, k_vec_axpby, dim3((unsigned)((m + 255) / 256)), dim3(256), st, m, kk, (const double*)r2, 1.0, xm);  // (453)
  HIPCHK(hipMemcpyAsync(by, xm, sizeof(double) * m, hipMemcpyDeviceToDevice, st));
  launch(c, KID_qr_small, k_qr_trsv, dim3(1), dim3(256), st, (const double*)Lc, (int)m, m, by, 1);        // y = R^-1 x  (454-455)
  launch(c, KID_qr_comb, k_stack_comb, dim3((unsigned)std::min<int64_t>(65535, (bl + 255) / 256)), dim3(256), st,
         (const double*)D.ustack, bl, bl, (int)m, (const double*)xm, (const double*)r1, bx);                // Q x - r1    (457-458)
  hessian_impl(c, L, bx, 1, bl, 1, 0, st);                                                                  // G^T         (461)
  launch(c, KID_axpby, k_axpby, dim3(1024), dim3(256), st, bl, 0.0, (const double*)nullptr, 1.0 / kk, bx);
  HIPCHK(end_call(c));
  return 0;
}

// test hook: copies the composite factor (lower m x m, R^T) to the host and the Gram matrix of the current stack
// (Q^T Q in the weighted inner product, = I after a factorisation) into G (device, m x m)
int kkt_qr_inspect(csp_ctx* c, double* Rt_host, double* G_dev, void* stream) {
  if (int rc = ready(c)) return rc;
  DeviceCtx& D = c->D;
  if (!D.m || !D.qr_valid) return SMCP_EINVAL;
  const int64_t m = D.m;
  hipStream_t st = (hipStream_t)stream;
  if (G_dev) {
    const int64_t range[2] = {0, c->S.blklen()};
    if (int rc = gram_accumulate(c, 1, range, G_dev, m, st)) return rc;
  }
  if (Rt_host) {
    HIPCHK(hipMemcpyAsync(Rt_host, D.qr_ws + 2 * m * m, sizeof(double) * m * m, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

}  // extern "C"
