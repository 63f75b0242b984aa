// KKT layer on device: Amap / Aadj, Schur-complement build + factor, solve_ closure.
// Mirrors kkt_chol of src/python/solvers.py:477-541 (and Amap/Aadj 369-386); included by capi.hip.

namespace {

using namespace smcp;

// y[i] = sum_e w_e a_e X[idx_e], one workgroup per (constraint i, rhs r) with four gathers in flight per thread (a
// constraint of synth50k has 11 k entries: one wave per constraint walked them in 178 dependent steps);
// X_r = X + r*ldx; y_r = y + r*ldy.  The partial sums are combined in a fixed order (deterministic).
// gridDim.z > 1: workgroup z takes the z-th share of the constraint's entries and writes its partial sum to y + z * pstride (the
// caller adds the shares in order: k_vec_axpby_parts)
__global__ void __launch_bounds__(1024) k_amap(int64_t m, const int64_t* cptr, const int64_t* cidx, const double* cwval,
                                               const double* X, int64_t ldx, double* y, int64_t ldy, int64_t pstride) {
  __shared__ double part[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nthr = blockDim.x;
  const int64_t i = blockIdx.x;
  const int r = blockIdx.y;
  const double* x = X + (int64_t)r * ldx;
  const int64_t c0 = cptr[i], c1 = cptr[i + 1], share = (c1 - c0 + gridDim.z - 1) / gridDim.z;
  const int64_t e0 = c0 + (int64_t)blockIdx.z * share, e1 = e0 + share < c1 ? e0 + share : c1;
  y += (int64_t)blockIdx.z * pstride;
  double acc = 0.0;
  for (int64_t e = e0 + threadIdx.x; e < e1; e += 4 * nthr) {
    double w[4], v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t ee = e + (int64_t)q * nthr;
      w[q] = ee < e1 ? cwval[ee] : 0.0;
      v[q] = ee < e1 ? x[cidx[ee]] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc += w[q] * v[q];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int q = 0; q < (nthr >> 6); ++q) s += part[q];
    y[i + (int64_t)r * ldy] = s;
  }
}

// X[rpos[q]] = sum over constraints touching that position of y[con] * val   (X pre-zeroed)
__global__ void k_aadj(int64_t rnnz, const int64_t* rpos, const int64_t* rptr, const int32_t* rcon,
                       const double* rval, const double* y, double* X) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= rnnz) return;
  double acc = 0.0;
  for (int64_t e = rptr[q]; e < rptr[q + 1]; ++e) acc += rval[e] * y[rcon[e]];
  X[rpos[q]] = acc;
}

// X[rpos[q]] -= the same sum: X <- X - Aadj(y) touching only the positions that carry constraint entries (solve_ forms
// bx - Aadj(y) in place this way: no cleared scratch vector, no pass over the whole of blkval)
__global__ void k_aadj_sub(int64_t rnnz, const int64_t* rpos, const int64_t* rptr, const int32_t* rcon,
                           const double* rval, const double* y, double* X) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= rnnz) return;
  double acc = 0.0;
  for (int64_t e = rptr[q]; e < rptr[q + 1]; ++e) acc += rval[e] * y[rcon[e]];
  X[rpos[q]] -= acc;
}

// U_r (pre-zeroed) <- A_{j0+r} scattered into blkval coordinates
__global__ void k_scatter_constraints(int64_t j0, const int64_t* cptr, const int64_t* cidx,
                                      const double* cval, double* U, int64_t ldu) {
  const int r = blockIdx.y;
  const int64_t j = j0 + r;
  double* u = U + (int64_t)r * ldu;
  for (int64_t e = cptr[j] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < cptr[j + 1];
       e += (int64_t)gridDim.x * blockDim.x)
    u[cidx[e]] = cval[e];
}

// U_q (pre-zeroed) <- A_{ids[q]} scattered into blkval coordinates
__global__ void k_scatter_constraints_ids(const int32_t* ids, const int64_t* cptr, const int64_t* cidx,
                                          const double* cval, double* U, int64_t ldu) {
  const int64_t j = ids[blockIdx.y];
  double* u = U + (int64_t)blockIdx.y * ldu;
  for (int64_t e = cptr[j] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < cptr[j + 1];
       e += (int64_t)gridDim.x * blockDim.x)
    u[cidx[e]] = cval[e];
}
// H[ids[i], ids[j]] <- Hd[i, j]  (md x md, both triangles)
__global__ void k_scatter_hd(const int32_t* ids, int md, const double* Hd, double* H, int64_t ldh) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)md * md; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e % md), j = (int)(e / md);
    H[ids[i] + (int64_t)ids[j] * ldh] = Hd[e];
  }
}
// V (n x ncols, zeroed) <- unit vectors: column q has a one in row kidx[q]
__global__ void k_unit_columns(const int32_t* kidx, int64_t ncols, double* V, int64_t n) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < ncols) V[kidx[q] + q * n] = 1.0;
}
// misc.SCMcolumn2 (misc.c:620-663): H[i, s] = tr(A_i S^-1 A_s S^-1) from V_s = S^-1[:, K_s], for every
// constraint i and the column-sparse constraints s of this chunk (one wave per pair; H is symmetric, both
// triangles are written).  voff[q]: first column of V_s inside V for the q-th sparse constraint of the chunk.
// owner / me (sharded over ranks by sparse constraint, kkt_schur_gram_part): owner[i] = the rank that computes the columns of
// constraint i, -1 for the swept (dense-class) constraints.  The partial matrices of the ranks are SUMMED, so a pair of
// sparse constraints owned by two different ranks must be written by one of them only: the owner of the smaller index.
__global__ void k_scm_columns(int64_t m, const int64_t* cptr, const int32_t* a_r, const int32_t* a_c, const double* cval,
                              const int32_t* slist, const int64_t* voff, const int32_t* rloc, const int32_t* cloc,
                              const double* V, int64_t n, double* H, int64_t ldh, const int32_t* owner = nullptr, int me = 0) {
  const int lane = threadIdx.x & 63;
  const int64_t i = blockIdx.x;
  const int64_t s = slist[blockIdx.y];
  if (owner && owner[i] >= 0 && owner[i] != me && i < s) return;
  const double* Vs = V + voff[blockIdx.y] * n;
  double acc = 0.0;
  for (int64_t q = cptr[i] + lane; q < cptr[i + 1]; q += 64) {
    const int r1 = a_r[q], c1 = a_c[q];
    const double beta = cval[q];
    double t = 0.0;
    for (int64_t p = cptr[s]; p < cptr[s + 1]; ++p) {
      const double alpha = (a_r[p] != a_c[p]) ? 2.0 * cval[p] : cval[p];
      const double* Vr = Vs + (int64_t)rloc[p] * n;
      const double* Vc = Vs + (int64_t)cloc[p] * n;
      double w = Vr[r1] * Vc[c1];
      if (r1 != c1) w += Vr[c1] * Vc[r1];
      t += alpha * w;
    }
    acc += beta * t;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) { H[i + s * ldh] = acc; H[s + i * ldh] = acc; }
}

// single-workgroup dense Cholesky / triangular solves (generic path)
__global__ void k_dense_potrf(double* A, int n, int64_t lda, int* info) {
  int f = wg::potrf(n, A, lda);
  if (f && threadIdx.x == 0) *info = f;
}
__global__ void k_dense_potrs(const double* A, int n, int64_t lda, double* B, int nrhs, int64_t ldb) {
  wg::trsm_llN(n, nrhs, A, lda, B, ldb);
  wg::trsm_llT(n, nrhs, A, lda, B, ldb);
}
// threads of the one-workgroup factorisation of H (SMCP_POTRF_THREADS, timing studies)
static dim3 potrf_blk() {
  static int t = 0;
  if (!t) { const char* e = sw_str("SMCP_POTRF_THREADS"); t = e ? atoi(e) : 1024; if (t < 64 || t > 1024 || (t & 63)) t = 1024; }
  return dim3(t);
}
// m <= 128: the whole factorisation in the LDS of one workgroup -- 16-wide block columns, diagonal blocks factored
// and inverted by one wavefront (potrf_inv16), panel and trailing updates on MFMA (the scheme of k_factor_yaa_lds).
// The inverses of the diagonal blocks are kept (dinv: 256 doubles per block) for k_dense_potrs_small.
__global__ void __launch_bounds__(1024) k_dense_potrf_small(double* A, int n, int64_t lda, int* info, double* dinv) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int ld = n | 1;
  double* const M = smem;
  double* const D16 = smem + (int64_t)ld * n;
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    const int i = e % n, j = e / n;
    M[i + j * ld] = (i >= j) ? A[i + (int64_t)j * lda] : 0.0;
  }
  for (int jb = 0; jb < n; jb += 16) {
    const int bw = min(16, n - jb);
    const int f = potrf_inv16(M + jb + jb * ld, ld, bw, D16);
    if (f) { if (threadIdx.x == 0) *info = jb + f; return; }
    for (int e = threadIdx.x; e < 256; e += blockDim.x) dinv[(jb >> 4) * 256 + e] = D16[e];
    const int mrem = n - jb - bw;
    if (mrem > 0) {
      double* Pj = M + (jb + bw) + jb * ld;
      wg_mma(mrem, bw, bw, [=](int m, int kk) { return Pj[m + kk * ld]; },
             [=](int kk, int nn_) { return D16[nn_ + kk * 16]; },
             [=](int m, int nn_, double acc) { Pj[m + nn_ * ld] = acc; });
      __syncthreads();
      double* Tr = M + (jb + bw) + (jb + bw) * ld;
      wg_mma(mrem, mrem, bw, [=](int m, int kk) { return Pj[m + kk * ld]; },
             [=](int kk, int nn_) { return Pj[nn_ + kk * ld]; },
             [=](int m, int nn_, double acc) { if (m >= nn_) Tr[m + nn_ * ld] -= acc; }, true);
      __syncthreads();
    }
  }
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    const int i = e % n, j = e / n;
    if (i >= j) A[i + (int64_t)j * lda] = M[i + j * ld];
  }
}
// Triangular solve with one 16 x 16 (bw x bw) diagonal block of the factor by ONE wavefront, by substitution
// (backward stable; multiplying by the explicit block inverse costs the interior-point endgame several digits):
// lane i holds entry i of the right-hand side and row i (trans 0: L y = t) or column i (trans 1: L^T x = t) of
// the block in registers; the solved entries are broadcast with shuffles.  Returns entry `lane` of the solution.
__device__ inline double wave_trsv16(const double* A, int64_t lda, int jb, int bw, double ti, int trans) {
  const int i = threadIdx.x & 63;
  double Lr[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const bool in = i < bw && j < bw && (trans ? j >= i : j <= i);
    Lr[j] = in ? (trans ? A[(jb + j) + (int64_t)(jb + i) * lda] : A[(jb + i) + (int64_t)(jb + j) * lda]) : (i == j ? 1.0 : 0.0);
  }
  double dii = 1.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) if (j == i) dii = Lr[j];
  // one division per lane and call: a division inside each of the sixteen dependent steps was most of the chain
  // (k_dense_potrs_small: 41 us at m = 100 whether the factor came from LDS or from global memory)
  const double rdii = 1.0 / dii;
  double xi = 0.0;
  if (!trans) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double xj = __shfl(ti * rdii, j, 64);
      if (i == j) xi = xj;
      if (i > j) ti -= Lr[j] * xj;
    }
  } else {
#pragma unroll
    for (int j = 15; j >= 0; --j) {
      const double xj = __shfl(ti * rdii, j, 64);
      if (i == j) xi = xj;
      if (i < j) ti -= Lr[j] * xj;
    }
  }
  return xi;
}
// A x = b with the factor of k_dense_potrf_small (one right-hand side, one workgroup, n <= 128).  The factor is copied to
// LDS first (dynamic: n (n | 1) doubles): the 2 ceil(n / 16) block steps each read their diagonal block and the columns
// below / beside it, and from global memory every step was two dependent round trips (41 us at m = 100, twice per solve_
// of the interior-point iteration ... once per solve_); from LDS the steps are the substitution chains alone.
__global__ void __launch_bounds__(256) k_dense_potrs_small(const double* Ag, int n, int64_t ldag, const double* dinv, double* b) {
  extern __shared__ __attribute__((aligned(16))) double sAf[];
  __shared__ double x[128], t[16];
  const int tid = threadIdx.x;
  const int lda = n | 1;
  double* const A = sAf;
  for (int e = tid; e < n * n; e += 256) {
    const int i = e % n, j = e / n;
    if (i >= j) A[i + j * lda] = Ag[i + (int64_t)j * ldag];
  }
  if (tid < n) x[tid] = b[tid];
  __syncthreads();
  for (int jb = 0; jb < n; jb += 16) {            // L y = b
    const int bw = min(16, n - jb);
    if (tid < 64) {
      const double v = wave_trsv16(A, lda, jb, bw, tid < bw ? x[jb + tid] : 0.0, 0);
      if (tid < bw) t[tid] = v;
    }
    __syncthreads();
    if (tid < bw) x[jb + tid] = t[tid];
    const int i = jb + bw + tid;
    if (i < n) { double acc = 0.0; for (int j = 0; j < bw; ++j) acc += A[i + (int64_t)(jb + j) * lda] * t[j]; x[i] -= acc; }
    __syncthreads();
  }
  for (int jb = ((n - 1) >> 4) << 4; jb >= 0; jb -= 16) {   // L^T x = y
    const int bw = min(16, n - jb);
    if (tid < 64) {
      const double v = wave_trsv16(A, lda, jb, bw, tid < bw ? x[jb + tid] : 0.0, 1);
      if (tid < bw) t[tid] = v;
    }
    __syncthreads();
    if (tid < bw) x[jb + tid] = t[tid];
    if (tid < jb) { double acc = 0.0; for (int j = 0; j < bw; ++j) acc += A[(jb + j) + (int64_t)tid * lda] * t[j]; x[tid] -= acc; }
    __syncthreads();
  }
  if (tid < n) b[tid] = x[tid];
}

// One block step of the blocked triangular solves with the Cholesky factor A (lower, n x n).  Every workgroup
// solves the 64-wide diagonal block redundantly (four 16-wide substitutions by wavefront 0, see wave_trsv16);
// workgroup 0 publishes x_blk to xout; then the workgroups update their slice of the remaining rows:
//   trans 0 (L y = b):    b[i] -= sum_j A[i, jb + j] x[j],  i >= jb + w   (one thread per row, coalesced)
//   trans 1 (L^T x = y):  b[i] -= sum_j A[jb + j, i] x[j],  i <  jb       (one wave per row, lanes over j)
__global__ void __launch_bounds__(256) k_dense_trsv_step(const double* A, int n, int64_t lda, const double* dinv, int jb, int w,
                                                         double* b, double* xout, int trans) {
  __shared__ double t[64], x[64];
  const int tid = threadIdx.x;
  if (tid < 64) t[tid] = tid < w ? b[jb + tid] : 0.0;
  __syncthreads();
  if (tid < 64) {
    const int nsb = (w + 15) >> 4;
    for (int q = 0; q < nsb; ++q) {
      const int sb = trans ? nsb - 1 - q : q;          // forward: top sub-block first; transposed: bottom first
      const int s0 = 16 * sb, bw = min(16, w - s0);
      const double v = wave_trsv16(A, lda, jb + s0, bw, tid < bw ? t[s0 + tid] : 0.0, trans);
      if (tid < bw) x[s0 + tid] = v;
      // remaining sub-blocks of this diagonal block (same wavefront: LDS traffic is program-ordered)
      if (!trans) {
        const int i = s0 + bw + tid;
        if (i < w) { double acc = 0.0; for (int j = 0; j < bw; ++j) acc += A[(jb + i) + (int64_t)(jb + s0 + j) * lda] * x[s0 + j]; t[i] -= acc; }
      } else {
        if (tid < s0) { double acc = 0.0; for (int j = 0; j < bw; ++j) acc += A[(jb + s0 + j) + (int64_t)(jb + tid) * lda] * x[s0 + j]; t[tid] -= acc; }
      }
    }
    if (blockIdx.x == 0 && tid < w) xout[jb + tid] = x[tid];
  }
  __syncthreads();
  if (!trans) {
    const int i = jb + w + blockIdx.x * 256 + tid;
    if (i < n) {
      double acc = 0.0;
      const double* Ai = A + i + (int64_t)jb * lda;
      for (int j = 0; j < w; ++j) acc += Ai[(int64_t)j * lda] * x[j];
      b[i] -= acc;
    }
  } else {
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = blockIdx.x * 4 + wave; i < jb; i += gridDim.x * 4) {
      double acc = (lane < w) ? A[(jb + lane) + (int64_t)i * lda] * x[lane] : 0.0;
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
      if (lane == 0) b[i] -= acc;
    }
  }
}

// A x = b with the Cholesky factor A (lower, n x n, 128 < n <= 1024), ONE right-hand side, in ONE launch of one workgroup:
// the step kernel above is 2 ceil(n / 64) dependent launches of ~21 us (m = 1000, config 4: 32 launches, 0.76 ms per solve_ for
// 8 MB of factor).  Here the chain is 2 ceil(n / 16) steps of one 16-wide substitution by wavefront 0 (wave_trsv16 on the
// diagonal blocks, all of them staged in LDS up front: the same arithmetic, so the same backward-stable solve) followed by
// the update of the other rows, one thread per row, whose sixteen factor entries were fetched a step ahead: the factor is
// static, so its stream never waits for the chain.  x lives in LDS; barriers wait for LDS traffic only.
// (Round 4 tried 64-wide steps through the diagonal blocks' cached inverses, x_blk = Dinv r_blk, the other rows updated in
// four sub-steps of sixteen prefetched columns: 0.62 ms against 0.37 here -- a sub-step is 16 multiply-adds, far shorter
// than the ~2.4 us a request to the 8 MB factor takes, and 128 registers per thread hold only one sub-step ahead, so the
// kernel ran 256 exposed round trips where this one hides its 126 behind the one-wave substitutions.)
constexpr int POTRS1_MAXN = 1024;
__host__ __device__ inline size_t potrs_one_lds(int n) { return ((size_t)((n + 15) & ~15) * 17 + 16) * sizeof(double); }
__global__ void __launch_bounds__(1024) k_dense_potrs_one(const double* A, int n, int64_t lda, double* b) {
  extern __shared__ __attribute__((aligned(16))) double sm1[];
  const int tid = threadIdx.x, npad = (n + 15) & ~15, nblk = npad >> 4;
  double* const x = sm1;
  double* const t = sm1 + npad;              // 16
  double* const dg = t + 16;                 // nblk x 256: the diagonal 16 x 16 blocks (ld 16), identity beyond n
  for (int e = tid; e < nblk * 256; e += 1024) {
    const int blk = e >> 8, r = e & 15, cc = (e >> 4) & 15, i = 16 * blk + r, j = 16 * blk + cc;
    dg[e] = (i < n && j < n && i >= j) ? A[i + (int64_t)j * lda] : (r == cc ? 1.0 : 0.0);
  }
  for (int e = tid; e < npad; e += 1024) x[e] = e < n ? b[e] : 0.0;
  const int i = tid;                         // this thread's row (forward) / column (backward)
  double va[16], vb[16];
  // ---- L y = b
  auto fetch_f = [&](int jb, double (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (i < n && i >= jb + 16 && jb + j < n) ? A[i + (int64_t)(jb + j) * lda] : 0.0;
  };
  auto step_f = [&](int blk, const double (&cur)[16]) {
    const int jb = 16 * blk;
    if (tid < 64) {
      const double v = wave_trsv16(dg + 256 * blk, 16, 0, 16, tid < 16 ? x[jb + tid] : 0.0, 0);
      if (tid < 16) t[tid] = v;
    }
    lds_barrier();
    if (tid < 16) x[jb + tid] = t[tid];
    if (i >= jb + 16 && i < n) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += cur[j] * t[j];
      x[i] -= acc;
    }
    lds_barrier();
  };
  fetch_f(0, va);
  __syncthreads();
  for (int blk = 0; blk < nblk; blk += 2) {
    if (blk + 1 < nblk) fetch_f(16 * (blk + 1), vb);
    step_f(blk, va);
    if (blk + 1 < nblk) {
      if (blk + 2 < nblk) fetch_f(16 * (blk + 2), va);
      step_f(blk + 1, vb);
    }
  }
  // ---- L^T x = y
  auto fetch_b = [&](int jb, double (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (i < jb && jb + j < n) ? A[(jb + j) + (int64_t)i * lda] : 0.0;
  };
  auto step_b = [&](int blk, const double (&cur)[16]) {
    const int jb = 16 * blk;
    if (tid < 64) {
      const double v = wave_trsv16(dg + 256 * blk, 16, 0, 16, tid < 16 ? x[jb + tid] : 0.0, 1);
      if (tid < 16) t[tid] = v;
    }
    lds_barrier();
    if (tid < 16) x[jb + tid] = t[tid];
    if (i < jb) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += cur[j] * t[j];
      x[i] -= acc;
    }
    lds_barrier();
  };
  fetch_b(16 * (nblk - 1), va);
  for (int blk = nblk - 1; blk >= 0; blk -= 2) {
    if (blk >= 1) fetch_b(16 * (blk - 1), vb);
    step_b(blk, va);
    if (blk >= 1) {
      if (blk >= 2) fetch_b(16 * (blk - 2), va);
      step_b(blk - 1, vb);
    }
  }
  if (tid < n) b[tid] = x[tid];
}

// y = a*y + x (length m), small
__global__ void k_vec_axpby(int64_t m, double a, const double* x, double b, double* y) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) y[i] = a * x[i] + b * y[i];
}
// y = b*y + a*(x_0 + x_1 + ... in order), x_z = x + z * pstride: the shares of a split k_amap
__global__ void k_vec_axpby_parts(int64_t m, double a, const double* x, int parts, int64_t pstride, double b, double* y) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double s = 0.0;
  for (int z = 0; z < parts; ++z) s += x[i + (int64_t)z * pstride];
  y[i] = a * s + b * y[i];
}

int amap_impl(csp_ctx* c, const double* X, int64_t ldx, int nrhs, double* y, int64_t ldy, hipStream_t st) {
  const DeviceCtx& D = c->D;
  // workgroup size by the average list length (short lists: LP / max-cut constraints have a handful of entries)
  const int64_t avg = D.m ? D.cnnz / D.m : 0;
  const int thr = avg > 4096 ? 1024 : (avg > 512 ? 256 : 64);
  launch(c, KID_amap, k_amap, dim3((unsigned)D.m, nrhs), dim3(thr), st, D.m,
                     D.cptr, D.cidx, D.cwval, X, ldx, y, ldy, (int64_t)0);
  return 0;
}
int aadj_impl(csp_ctx* c, const double* y, double* X, hipStream_t st) {
  const DeviceCtx& D = c->D;
  if (hipMemsetAsync(X, 0, sizeof(double) * c->S.blklen(), st) != hipSuccess) return SMCP_EHIP;
  if (D.rnnz)
    launch(c, KID_aadj, k_aadj, dim3((unsigned)((D.rnnz + 255) / 256)), dim3(256), st, D.rnnz, D.rpos,
                       D.rptr, D.rcon, D.rval, y, X);
  return 0;
}

}  // namespace

extern "C" {

int kkt_set_constraints(csp_ctx* c, int64_t m, const int64_t* cptr, const int64_t* cidx, const double* cval) {
  if (int rc = ready(c)) return rc;
  if (m < 1 || !cptr || !cidx || !cval) return SMCP_EINVAL;
  DeviceCtx& D = c->D;
  const Symbolic& S = c->S;
  HIPCHK(hipSetDevice(D.device));
  // a Schur complement of the OUTGOING constraint set still waiting for its factorisation (deferred status): it is complete
  // and does not depend on what is replaced here -- factor it where it stands, on the stream it was built on
  if (D.h_pending) { if (int rc = flush_pending_potrf(c, D.h_pending_stream, nullptr, false)) return rc; }
  void* old[] = {D.fz_no, D.fz_slot, D.fz_ptr, D.fz_pk, D.fz_s, D.lg_eptr, D.lg_epk, D.lg_ew, D.lg_remap, D.lg_tab, D.cptr, D.cidx, D.cval, D.cwval, D.rpos, D.rptr, D.rcon, D.rval, D.ustack,
                 D.a_r, D.a_c, D.s_rloc, D.s_cloc, D.dlist, D.slist, D.kidx, D.vbuf, D.hd, D.kc_ptr, D.kc_off, D.kc_val, D.kc_ij, D.scm_owner};
  for (void* p : old) if (p) hipFree(p);
  D.cptr = nullptr; D.cidx = nullptr; D.cval = nullptr; D.cwval = nullptr; D.rpos = nullptr;
  D.rptr = nullptr; D.rcon = nullptr; D.rval = nullptr; D.ustack = nullptr;
  D.a_r = D.a_c = D.s_rloc = D.s_cloc = D.dlist = D.slist = D.kidx = nullptr;
  D.scm_owner = nullptr;
  D.vbuf = D.hd = nullptr;
  D.kc_ptr = D.kc_off = nullptr; D.kc_val = nullptr; D.kc_ij = nullptr;
  D.fz_no = D.fz_slot = D.fz_ptr = D.fz_pk = nullptr; D.fz_s = nullptr; D.fz_nfam = 0; D.fz_ok = false;
  D.md = D.ns = D.vcols = 0;
  D.lg_children = 0; D.lg_nochild = false;
  D.lg_eptr = D.lg_epk = D.lg_remap = nullptr; D.lg_ew = D.lg_tab = nullptr;
  c->gsl_key.clear();
  if (D.qr_ws) { hipFree(D.qr_ws); D.bytes -= D.qr_len * 8; D.qr_ws = nullptr; D.qr_len = 0; }
  D.qr_valid = false;
  SetupClock clk("kkt_set_constraints");
  const int64_t nnz = cptr[m];
  // diagonal flags: position -> is it a diagonal entry of its NN block?
  std::vector<double> w(nnz);
  std::vector<int32_t> ar(nnz), ac(nnz);   // entries in (permuted) matrix coordinates
  std::vector<int32_t> ek(nnz), eoff(nnz); // clique of each entry, position inside that clique's panel
  {
    // locate clique by binary search on blkptr (the entry ranges split over host threads)
    const unsigned hw = std::thread::hardware_concurrency();
    const int nth = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hw ? hw : 1), (int64_t)16, nnz / 65536 + 1}));
    std::atomic<int> bad{0};
    auto work = [&](int tix) {
      const int64_t e0 = nnz * tix / nth, e1 = nnz * (tix + 1) / nth;
      int64_t k = 0;
      for (int64_t e = e0; e < e1; ++e) {
        int64_t pos = cidx[e];
        if (pos < 0 || pos >= S.blklen()) { bad = 1; return; }
        if (pos < S.blkptr[k] || pos >= S.blkptr[k + 1])      // (runs of entries share their clique)
          k = (int64_t)(std::upper_bound(S.blkptr.begin(), S.blkptr.end(), pos) - S.blkptr.begin()) - 1;
        int64_t nf = S.nf(k), off = pos - S.blkptr[k];
        int64_t col = off / nf, row = off % nf;
        if (row < col) { bad = 1; return; }  // upper triangle of the NN block is not part of V
        w[e] = (row == col) ? cval[e] : 2.0 * cval[e];
        ar[e] = (int32_t)S.rowidx[S.rowptr[k] + row];
        ac[e] = (int32_t)(S.snptr[k] + col);
        ek[e] = (int32_t)k;
        eoff[e] = (int32_t)off;
      }
    };
    run_threads(nth, work);
    if (bad) return SMCP_EINVAL;
  }
  clk.mark("locate entries");
  // Column-sparse constraints (misc.nzcolumns / misc.matperm, misc.c:682-773, solvers.py:246-268): a constraint
  // whose entries touch at most int(n * tnzcols) distinct rows/columns takes the SCMcolumn2 path (two sparse
  // triangular solves for S^-1[:, K_s], then pairwise contractions) instead of a Hessian sweep.
  std::vector<int32_t> dl, sl, kidx, rloc(nnz, 0), cloc(nnz, 0);
  c->h_kptr.assign(1, 0);
  {
    static int off = -1;
    if (off < 0) { const char* e = sw_str("SMCP_SCM"); off = (e && e[0] == '0') ? 1 : 0; }
    const int64_t tnz = (int64_t)((double)S.n * c->tnzcols);
    // at most this many columns of S^-1 are formed per constraint (n x |K| doubles of workspace)
    const int64_t kcap = std::min<int64_t>(tnz, std::max<int64_t>(1, ((int64_t)256 << 20) / std::max<int64_t>(1, S.n * 8)));
    const int64_t sepsum = std::max<int64_t>(1, S.sepptr[S.nsn]);
    const int64_t trsm_cap = std::max<int64_t>(1, (D.max_rhs * D.tmplen) / sepsum);
    // the distinct rows / columns of every constraint: independent per constraint, host threads take them round-robin
    // (synth50k: 100 sorts of 23 k indices, 65 ms on one thread); the lists are then joined in constraint order
    std::vector<std::vector<int32_t>> kss((size_t)m);
    std::vector<char> is_sparse((size_t)m, 0);
    const bool scm_on = !off && !use_generic(c);
    const int64_t cap = std::min(kcap, trsm_cap);
    {
      const unsigned hw = std::thread::hardware_concurrency();
      const int nth = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hw ? hw : 1), (int64_t)16, m, nnz / 4096 + 1}));
      auto work = [&](int tix) {
        for (int64_t j = tix; j < m; j += nth) {
          std::vector<int32_t>& ks = kss[(size_t)j];
          ks.reserve((size_t)(2 * (cptr[j + 1] - cptr[j])));
          for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) { ks.push_back(ar[e]); ks.push_back(ac[e]); }
          std::sort(ks.begin(), ks.end());
          ks.erase(std::unique(ks.begin(), ks.end()), ks.end());
          const int64_t nz = (int64_t)ks.size();
          const bool sparse = scm_on && nz > 0 && nz <= cap;
          is_sparse[(size_t)j] = sparse ? 1 : 0;
          if (!sparse) { std::vector<int32_t>().swap(ks); continue; }
          for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) {
            rloc[e] = (int32_t)(std::lower_bound(ks.begin(), ks.end(), ar[e]) - ks.begin());
            cloc[e] = (int32_t)(std::lower_bound(ks.begin(), ks.end(), ac[e]) - ks.begin());
          }
        }
      };
      run_threads(nth, work);
    }
    for (int64_t j = 0; j < m; ++j) {
      if (!is_sparse[(size_t)j]) { dl.push_back((int32_t)j); continue; }
      sl.push_back((int32_t)j);
      kidx.insert(kidx.end(), kss[(size_t)j].begin(), kss[(size_t)j].end());
      c->h_kptr.push_back((int64_t)kidx.size());
    }
  }
  c->h_slist = sl;
  clk.mark("classify");
  // CSR by position
  // entries ordered by position, ties in constraint order: a counting sort over the positions of V (a comparison sort
  // of the 1.1 M entries of synth50k took 74 ms)
  // ... split by position range over host threads: every thread walks the entry list for the positions of its range
  // (counts, then places), so the pieces come out in global order and only their offsets are laid out serially
  std::vector<int64_t> rpos, rptr;
  std::vector<int32_t> rcon(nnz);
  std::vector<double> rval(nnz);
  {
    std::vector<int32_t> con(nnz);
    for (int64_t j = 0; j < m; ++j)
      for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) con[e] = (int32_t)j;
    const unsigned hw = std::thread::hardware_concurrency();
    const int nth = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hw ? hw : 1), (int64_t)16, nnz / 65536 + 1}));
    const int64_t P = S.blklen();
    std::vector<std::vector<int64_t>> start((size_t)nth);          // per thread: first slot of every position of its range
    std::vector<int64_t> ecount((size_t)nth + 1, 0), dcount((size_t)nth + 1, 0);
    auto count = [&](int tix) {
      const int64_t p0 = P * tix / nth, p1 = P * (tix + 1) / nth;
      std::vector<int64_t>& st = start[(size_t)tix];
      st.assign((size_t)(p1 - p0) + 1, 0);
      for (int64_t e = 0; e < nnz; ++e) { const int64_t p = cidx[e]; if (p >= p0 && p < p1) ++st[(size_t)(p - p0) + 1]; }
      int64_t distinct = 0;
      for (int64_t p = 0; p < p1 - p0; ++p) { distinct += st[(size_t)p + 1] != 0; st[(size_t)p + 1] += st[(size_t)p]; }
      ecount[(size_t)tix + 1] = st[(size_t)(p1 - p0)];
      dcount[(size_t)tix + 1] = distinct;
    };
    run_threads(nth, count);
    for (int t = 0; t < nth; ++t) { ecount[(size_t)t + 1] += ecount[(size_t)t]; dcount[(size_t)t + 1] += dcount[(size_t)t]; }
    rpos.resize((size_t)dcount[(size_t)nth]);
    rptr.resize((size_t)dcount[(size_t)nth] + 1);
    auto place = [&](int tix) {
      const int64_t p0 = P * tix / nth, p1 = P * (tix + 1) / nth, q0 = ecount[(size_t)tix];
      std::vector<int64_t>& st = start[(size_t)tix];
      int64_t d = dcount[(size_t)tix];
      for (int64_t p = 0; p < p1 - p0; ++p)
        if (st[(size_t)p + 1] != st[(size_t)p]) { rpos[(size_t)d] = p0 + p; rptr[(size_t)d] = q0 + st[(size_t)p]; ++d; }
      for (int64_t e = 0; e < nnz; ++e) {                          // e ascending: ties stay in constraint order
        const int64_t p = cidx[e];
        if (p < p0 || p >= p1) continue;
        const int64_t q = q0 + st[(size_t)(p - p0)]++;
        rcon[(size_t)q] = con[e];
        rval[(size_t)q] = cval[e];
      }
    };
    run_threads(nth, place);
    rptr[(size_t)dcount[(size_t)nth]] = nnz;
  }
  clk.mark("CSR by position");
  std::vector<int64_t> vcptr(cptr, cptr + m + 1), vcidx(cidx, cidx + nnz);
  std::vector<double> vcval(cval, cval + nnz);
  int rc = 0;
  if ((rc = dev_upload(&D.cptr, vcptr, D.bytes))) return rc;
  if ((rc = dev_upload(&D.cidx, vcidx, D.bytes))) return rc;
  if ((rc = dev_upload(&D.cval, vcval, D.bytes))) return rc;
  if ((rc = dev_upload(&D.cwval, w, D.bytes))) return rc;
  if ((rc = dev_upload(&D.rpos, rpos, D.bytes))) return rc;
  if ((rc = dev_upload(&D.rptr, rptr, D.bytes))) return rc;
  if ((rc = dev_upload(&D.rcon, rcon, D.bytes))) return rc;
  if ((rc = dev_upload(&D.rval, rval, D.bytes))) return rc;
  if ((rc = dev_upload(&D.a_r, ar, D.bytes))) return rc;
  if ((rc = dev_upload(&D.a_c, ac, D.bytes))) return rc;
  if ((rc = dev_upload(&D.s_rloc, rloc, D.bytes))) return rc;
  if ((rc = dev_upload(&D.s_cloc, cloc, D.bytes))) return rc;
  if ((rc = dev_upload(&D.dlist, dl, D.bytes))) return rc;
  if ((rc = dev_upload(&D.slist, sl, D.bytes))) return rc;
  if ((rc = dev_upload(&D.kidx, kidx, D.bytes))) return rc;
  clk.mark("uploads");
  D.md = (int64_t)dl.size();
  D.ns = (int64_t)sl.size();
  if (D.ns) {
    int64_t kmax = 1;
    for (int64_t q = 0; q < D.ns; ++q) kmax = std::max(kmax, c->h_kptr[q + 1] - c->h_kptr[q]);
    const int64_t sepsum = std::max<int64_t>(1, S.sepptr[S.nsn]);
    const int64_t trsm_cap = std::max<int64_t>(1, (D.max_rhs * D.tmplen) / sepsum);
    const int64_t want = std::min<int64_t>((int64_t)kidx.size(), ((int64_t)256 << 20) / std::max<int64_t>(1, S.n * 8));
    D.vcols = std::min(trsm_cap, std::max(kmax, want));
    if ((rc = dev_alloc(&D.vbuf, D.vcols * S.n, D.bytes))) return rc;
    if (D.md && (rc = dev_alloc(&D.hd, D.md * D.md, D.bytes))) return rc;
  }
  // entries grouped by (clique, constraint): the sweeps of the Schur complement build their input panels from these
  if (nnz < ((int64_t)1 << 31) && S.nsn * (m + 1) <= ((int64_t)1 << 28)) {
    std::vector<int32_t> kptr((size_t)(S.nsn * (m + 1)) + 1, 0), koff(nnz);
    std::vector<double> kval(nnz);
    // (a slot belongs to one constraint: counting and filling run over the constraints on host threads)
    const unsigned hwc = std::thread::hardware_concurrency();
    const int nthc = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hwc ? hwc : 1), (int64_t)16, m, nnz / 65536 + 1}));
    auto over_constraints = [&](auto body) {
      run_threads(nthc, [&](int tix) { for (int64_t j = tix; j < m; j += nthc) body(j); });
    };
    over_constraints([&](int64_t j) {
      for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) kptr[(size_t)ek[e] * (m + 1) + j + 1]++;
    });
    // exclusive scan over (clique, constraint); slot (k, m) of a clique doubles as the start of clique k + 1
    {
      int64_t run = 0;
      for (int64_t k = 0; k < S.nsn; ++k) {
        for (int64_t j = 0; j <= m; ++j) {
          const size_t idx = (size_t)k * (m + 1) + j;
          const int32_t cnt = (j < m) ? kptr[idx + 1] : 0;
          kptr[idx] = (int32_t)run;
          if (j < m) run += cnt;
        }
      }
    }
    D.kc_maxlist = 0;             // over the cliques that can be members of a family (nn <= 16, na <= 64)
    for (int64_t k = 0; k < S.nsn; ++k)
      if (S.nn(k) <= 16 && S.na(k) <= 64)
        for (int64_t j = 0; j < m; ++j) {
          const size_t q = (size_t)k * (m + 1) + j;
          D.kc_maxlist = std::max<int64_t>(D.kc_maxlist, kptr[q + 1] - kptr[q]);
        }
    // most entries of a (family, constraint) pair: the parent's own + its children's (the entry-driven family sweep of
    // front_famt.hip turns every entry into one term of a rank-T product)
    D.fam_maxterms = 0;
    D.fam_meanterms = 0.0;
    {
      int64_t sum = 0, pairs = 0;
      for (int64_t k = 0; k < S.nsn; ++k)
        if (k < (int64_t)c->fam.size() && c->fam[k] == 2)
          for (int64_t j = 0; j < m; ++j) {
            int64_t tot = kptr[(size_t)k * (m + 1) + j + 1] - kptr[(size_t)k * (m + 1) + j];
            for (int64_t q2 = S.chptr[k]; q2 < S.chptr[k + 1]; ++q2) {
              const size_t q = (size_t)S.chidx[q2] * (m + 1) + j;
              tot += kptr[q + 1] - kptr[q];
            }
            D.fam_maxterms = std::max(D.fam_maxterms, tot);
            sum += tot;
            ++pairs;
          }
      if (pairs) D.fam_meanterms = (double)sum / (double)pairs;
    }
    D.kc_maxlist_large = 0;       // over the childless fronts beyond the small classes (sparse-input sweep of large fronts)
    for (int64_t k = 0; k < S.nsn; ++k)
      if ((S.nn(k) > 16 || S.na(k) > 64) && S.nn(k) <= 64 && S.na(k) <= 128 && S.chptr[k + 1] == S.chptr[k])
        for (int64_t j = 0; j < m; ++j) {
          const size_t q = (size_t)k * (m + 1) + j;
          D.kc_maxlist_large = std::max<int64_t>(D.kc_maxlist_large, kptr[q + 1] - kptr[q]);
        }
    // closed-form Gram blocks of the family children (front_leafgram.hip): per child its entries over all constraints
    // in constraint order (row | column << 8 | constraint << 16, value halved on the diagonal) -- static, so the pair
    // kernel reads them with one coalesced load -- and the record size of the per-step tables
    D.lg_children = 0; D.lg_maxent = 0; D.lg_pairs = 0; D.lg_rows = 0; D.lg_rec = 0;
    c->lg_slot_of.assign((size_t)S.nsn, -1);
    std::vector<int32_t> lg_eptr(1, 0);
    for (int64_t k = 0; k < S.nsn; ++k)
      if (k < (int64_t)c->fam.size() && c->fam[k] == 1) {
        const int64_t E = kptr[(size_t)k * (m + 1) + m] - kptr[(size_t)k * (m + 1)];
        c->lg_slot_of[(size_t)k] = (int32_t)D.lg_children++;
        D.lg_maxent = std::max(D.lg_maxent, E);
        D.lg_pairs += E * (E + 1) / 2;
        D.lg_rows += S.nf(k) * S.nn(k);
        D.lg_rec = std::max<int>(D.lg_rec, (int)(S.nf(k) * S.nf(k) + S.nf(k) * S.nn(k)));
        lg_eptr.push_back(lg_eptr.back() + (int32_t)E);
      }
    // ... and the same positions as (row | column << 16) of the clique's panel, for k_fam_sparse (its members have < 2^16 rows)
    std::vector<int32_t> kij(nnz);
    {
      std::vector<int32_t> fill(kptr.begin(), kptr.end() - 1);
      over_constraints([&](int64_t j) {
        for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) {
          const int64_t nf = S.nf(ek[e]);
          const int32_t q = fill[(size_t)ek[e] * (m + 1) + j]++;
          koff[q] = eoff[e];
          kval[q] = cval[e];
          kij[q] = (int32_t)((eoff[e] % nf) & 0xffff) | (int32_t)((eoff[e] / nf) << 16);
        }
      });
    }
    kptr.pop_back();
    D.kc_sorted = true;           // (CCS columns with ascending rows give ascending panel positions per clique)
    for (size_t q = 0; q + 1 < kptr.size() && D.kc_sorted; ++q)
      for (int32_t e = kptr[q] + 1; e < kptr[q + 1]; ++e)
        if (koff[(size_t)e] <= koff[(size_t)e - 1]) { D.kc_sorted = false; break; }
    clk.mark("entry tables");
    if (D.lg_children > 0 && m < 32768) {
      std::vector<int32_t> epk((size_t)lg_eptr.back()), remap((size_t)m, -1);
      std::vector<double> ewv((size_t)lg_eptr.back());
      for (int64_t k = 0; k < S.nsn; ++k) {
        const int32_t g = c->lg_slot_of[(size_t)k];
        if (g < 0) continue;
        int32_t o = lg_eptr[(size_t)g];
        for (int64_t j = 0; j < m; ++j)
          for (int32_t q = kptr[(size_t)k * (m + 1) + j]; q < kptr[(size_t)k * (m + 1) + j + 1]; ++q, ++o) {
            const int32_t i = kij[q] & 0xffff, jc = kij[q] >> 16;
            epk[(size_t)o] = i | (jc << 8) | ((int32_t)j << 16);
            ewv[(size_t)o] = i == jc ? 0.5 * kval[q] : kval[q];
          }
      }
      for (size_t q = 0; q < dl.size(); ++q) remap[(size_t)dl[q]] = (int32_t)q;
      if ((rc = dev_upload(&D.lg_eptr, lg_eptr, D.bytes))) return rc;
      if ((rc = dev_upload(&D.lg_epk, epk, D.bytes))) return rc;
      if ((rc = dev_upload(&D.lg_ew, ewv, D.bytes))) return rc;
      if ((rc = dev_upload(&D.lg_remap, remap, D.bytes))) return rc;
      if ((rc = dev_alloc(&D.lg_tab, D.lg_children * D.lg_rec, D.bytes))) return rc;
    }
    clk.mark("leaf Gram tables");
    // Static term lists of the family parents (fused extend-add, front_famt.hip lf_add_family): every entry of constraint j inside
    // family f -- the parent's own and its children's -- as (vector ids vx | vy << 16, scale), the mapping k_fam_terms does per
    // launch from the entry lists (front_famt.hip, header): own entry v at (i, j): e_i, e_j, v (v / 2 on the diagonal); child
    // entry at (separator row a, column j): q~_{c,j}, e_{rel_c[a]}, -v; child entry at (i, j) of its supernode block: q~_{c,i},
    // q~_{c,j}, v (v / 2).  Only when every family parent hangs under a large front whose packed triangle fits LDS.
    if (D.fam_maxterms > 0 && famt_terms_ok(D) && !c->fam.empty()) {
      std::vector<int32_t> fno((size_t)S.nsn, -1);
      c->fz_levels.assign((size_t)S.nlev, 0);
      int64_t nfam = 0;
      bool ok = true;
      for (int64_t k = 0; k < S.nsn; ++k)
        if (c->fam[(size_t)k] == 2) {
          fno[(size_t)k] = (int32_t)nfam++;
          const int64_t par = S.snpar[k];
          if (par < 0 || (size_t)par >= c->large_mask.size() || !c->large_mask[(size_t)par] || S.nf(par) > LF_ALDS_MAXNF) ok = false;
          else c->fz_levels[(size_t)S.level[(size_t)par]] = 1;
        }
      if (ok && nfam > 0 && nfam < ((int64_t)1 << 19) && nfam * (m + 1) < ((int64_t)1 << 31)) {
        // sizes first (a (family, constraint) list holds the parent's entries and its children's), then the lists themselves,
        // the families spread over host threads
        std::vector<int32_t> fptr((size_t)(nfam * (m + 1)) + 1, 0);
        std::vector<int64_t> fam_k((size_t)nfam);
        int64_t run = 0;
        for (int64_t k = 0; k < S.nsn; ++k) {
          const int32_t f = fno[(size_t)k];
          if (f < 0) continue;
          fam_k[(size_t)f] = k;
          for (int64_t j = 0; j < m; ++j) {
            fptr[(size_t)f * (m + 1) + j] = (int32_t)run;
            run += kptr[(size_t)k * (m + 1) + j + 1] - kptr[(size_t)k * (m + 1) + j];
            for (int64_t q2 = S.chptr[k]; q2 < S.chptr[k + 1]; ++q2) {
              const size_t q = (size_t)S.chidx[q2] * (m + 1) + j;
              run += kptr[q + 1] - kptr[q];
            }
          }
          fptr[(size_t)f * (m + 1) + m] = (int32_t)run;
        }
        fptr[(size_t)(nfam * (m + 1))] = (int32_t)run;
        std::vector<int32_t> fpk((size_t)run);
        std::vector<double> fsv((size_t)run);
        const int nthf = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hwc ? hwc : 1), (int64_t)16, nfam / 16 + 1}));
        run_threads(nthf, [&](int tix) {
          for (int64_t f = tix; f < nfam; f += nthf) {
            const int64_t k = fam_k[(size_t)f];
            for (int64_t j = 0; j < m; ++j) {
              size_t o = (size_t)fptr[(size_t)f * (m + 1) + j];
              for (int32_t q = kptr[(size_t)k * (m + 1) + j]; q < kptr[(size_t)k * (m + 1) + j + 1]; ++q, ++o) {
                const int32_t i = kij[q] & 0xffff, jc = kij[q] >> 16;
                fpk[o] = i | (jc << 16);
                fsv[o] = i == jc ? 0.5 * kval[q] : kval[q];
              }
              int32_t colbase = 0;
              for (int64_t q2 = S.chptr[k]; q2 < S.chptr[k + 1]; ++q2) {
                const int64_t cc = S.chidx[q2];
                const int32_t nnc = (int32_t)S.nn(cc);
                const int32_t* rel = &S.relidx[S.sepptr[cc]];
                for (int32_t q = kptr[(size_t)cc * (m + 1) + j]; q < kptr[(size_t)cc * (m + 1) + j + 1]; ++q, ++o) {
                  const int32_t i = kij[q] & 0xffff, jc = kij[q] >> 16;
                  if (i >= nnc) { fpk[o] = (FAMT_CHILD + colbase + jc) | (rel[i - nnc] << 16); fsv[o] = -kval[q]; }
                  else { fpk[o] = (FAMT_CHILD + colbase + i) | ((FAMT_CHILD + colbase + jc) << 16); fsv[o] = i == jc ? 0.5 * kval[q] : kval[q]; }
                }
                colbase += nnc;
              }
            }
          }
        });
        if ((rc = dev_upload(&D.fz_no, fno, D.bytes))) return rc;
        if ((rc = dev_upload(&D.fz_ptr, fptr, D.bytes))) return rc;
        if ((rc = dev_upload(&D.fz_pk, fpk, D.bytes))) return rc;
        if ((rc = dev_upload(&D.fz_s, fsv, D.bytes))) return rc;
        if ((rc = dev_alloc(&D.fz_slot, nfam + 8, D.bytes))) return rc;      // (+ the eight task counters of k_lf_assemble_fz)
        HIPCHK(hipMemset(D.fz_slot, 0, sizeof(int32_t) * (nfam + 8)));
        D.fz_nfam = nfam;
        D.fz_ok = true;
      }
    }
    clk.mark("family term lists");
    if ((rc = dev_upload(&D.kc_ij, kij, D.bytes))) return rc;
    if ((rc = dev_upload(&D.kc_ptr, kptr, D.bytes))) return rc;
    if ((rc = dev_upload(&D.kc_off, koff, D.bytes))) return rc;
    if ((rc = dev_upload(&D.kc_val, kval, D.bytes))) return rc;
  }
  clk.mark("table uploads");
  D.ustack_cols = std::max(D.max_rhs, m);
  if ((rc = dev_alloc(&D.ustack, D.ustack_cols * S.blklen(), D.bytes))) return rc;
  // entries the sweeps never write (strict upper triangles of the NN blocks) must stay finite
  HIPCHK(hipMemset(D.ustack, 0, sizeof(double) * D.ustack_cols * S.blklen()));
  if (!D.sw) {
    if ((rc = dev_alloc(&D.sw, S.blklen(), D.bytes))) return rc;
    hipLaunchKernelGGL(k_fill_sqrt_weights, dim3((unsigned)std::min<int64_t>(S.nsn, 4096)), dim3(256), 0, 0, D.cl, (int)S.nsn, D.sw);
    HIPCHK(hipDeviceSynchronize());
  }
  clk.mark("swept stack");
  D.m = m;
  D.cnnz = nnz;
  D.rnnz = (int64_t)rpos.size();
  return 0;
}

int kkt_set_tnzcols(csp_ctx* c, double tnzcols) {
  if (!c || !(tnzcols >= 0.0 && tnzcols <= 1.0)) return SMCP_EINVAL;
  c->tnzcols = tnzcols;
  return 0;
}

int kkt_amap(csp_ctx* c, const double* X, double* y, void* stream) {
  if (int rc = ready(c)) return rc;
  if (!c->D.m) return SMCP_EINVAL;
  amap_impl(c, X, 0, 1, y, 0, (hipStream_t)stream);
  HIPCHK(end_call(c));
  return 0;
}

int kkt_aadj(csp_ctx* c, const double* y, double* X, void* stream) {
  if (int rc = ready(c)) return rc;
  if (!c->D.m) return SMCP_EINVAL;
  if (int rc = aadj_impl(c, y, X, (hipStream_t)stream)) return rc;
  HIPCHK(end_call(c));
  return 0;
}

// launches of the dense Cholesky of A (no status read-back); info: the failure flag the kernels set (the context's flag, or
// a slot of its own when the factorisation runs on a side stream beside kernels that use the context's flag)
static int potrf_launch(csp_ctx* c, double* A, int64_t n, int64_t lda, hipStream_t st, int* info) {
  HIPCHK(hipMemsetAsync(info, 0, sizeof(int), st));
  c->D.hinv_tag = nullptr;
  static int oldp = -1;
  if (oldp < 0) { const char* e = sw_str("SMCP_POTRF_OLD"); oldp = (e && e[0] == '1') ? 1 : 0; }
  if (oldp || use_generic(c)) {
    launch(c, KID_dense_potrf, k_dense_potrf, dim3(1), dim3(1024), st, A, (int)n, lda, info);
    return 0;
  }
  if (n <= 2 * LB) {
    const int64_t need = 8 * 256 + 2 * n;
    if (c->D.hinv_cap < need) {
      if (c->D.hinv) { HIPCHK(hipFree(c->D.hinv)); c->D.bytes -= c->D.hinv_cap * 8; }
      c->D.hinv = nullptr; c->D.hinv_cap = 0;
      if (int rc = dev_alloc(&c->D.hinv, need, c->D.bytes)) return rc;
      c->D.hinv_cap = need;
    }
    const size_t lds = ((size_t)((n | 1) * n) + 256 + 8) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
      HIPCHK(hipFuncSetAttribute((const void*)k_dense_potrf_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
      attr_set = true;
    }
    launch_lds(c, KID_dense_potrf, k_dense_potrf_small, dim3(1), potrf_blk(), lds, st, A, (int)n, lda, info, c->D.hinv);
    return 0;
  }
  // blocked right-looking Cholesky, 64-wide block columns: diagonal block by one workgroup, panel and
  // trailing update as 64 x 64 MFMA tiles over the chip (the kernels of the large fronts, dense view)
  MfmaArgs a = mfma_args(c, nullptr, 0, 1);
  a.t.lev = c->D.lev3idx;
  a.t.info = info;
  a.lfd = c->D.lfd_dense;
  a.dn = (int)n; a.dld = lda;
  dim3 blk(256);
  const int64_t nblocks = (n + LB - 1) / LB, need = nblocks * LB * LB + 2 * n;
  if (c->D.hinv_cap < need) {
    if (c->D.hinv) { HIPCHK(hipFree(c->D.hinv)); c->D.bytes -= c->D.hinv_cap * 8; }
    c->D.hinv = nullptr; c->D.hinv_cap = 0;
    if (int rc = dev_alloc(&c->D.hinv, need, c->D.bytes)) return rc;
    c->D.hinv_cap = need;
  }
  // the whole blocked factorisation in ONE launch (front_flow.hip: tile dataflow inside the launch, the diagonal blocks' inverses
  // straight to their slots); SMCP_FLOW=0 or beyond 4096: three launches per block column
  if (flow_chol(c, st, A, lda, (int)n, c->D.hinv, nullptr, 5, info, 1)) return 0;
  for (int jb = 0; jb < (int)n; jb += LB) {
    a.lfd = c->D.hinv + (int64_t)(jb / LB) * LB * LB;      // the diagonal block's inverse goes straight to its slot (potrs reads it there)
    launch_lds(c, KID_lf_diag, k_lf_diag, dim3(1), dim3(512), LF_DIAG_LDS, st, a, A, (double*)nullptr, 5, jb, 1);
    const int mrem = (int)n - jb - LB;
    if (mrem > 0) {
      const int mt = tiles64(mrem);
      launch(c, KID_lf_chol_panel, k_lf_chol_panel, dim3(mt, 1), blk, st, a, A, (double*)nullptr, 5, jb);
      launch(c, KID_lf_chol_trail, k_lf_chol_trail, dim3(mt * (mt + 1) / 2, 1), blk, st, a, A, (double*)nullptr, 5, jb);
    }
  }
  return 0;
}
// A Schur complement built by kkt_schur_factor under csp_lazy_status is left UNFACTORED until its first use: kkt_solve
// then factors it on a side stream beside its first Hessian sweep (which does not read H), any other reader factors it
// where it stands.  Drops the mark without factoring when the caller is about to factor or rebuild that matrix itself.
static int flush_pending_potrf(csp_ctx* c, hipStream_t st, const void* only, bool drop) {
  DeviceCtx& D = c->D;
  if (!D.h_pending || (only && only != (const void*)D.h_pending)) return 0;
  double* H = D.h_pending;
  D.h_pending = nullptr;
  if (drop) return 0;
  if (int rc = potrf_launch(c, H, D.h_pending_n, D.h_pending_ld, st, c->D.info)) return rc;
  HIPCHK(end_call(c));
  int rc = fetch_info(c, st);
  if (!rc && !use_generic(c)) { D.hinv_tag = H; D.hinv_n = D.h_pending_n; }
  return rc;
}
int dense_potrf(csp_ctx* c, double* A, int64_t n, int64_t lda, void* stream) {
  if (int rc = ready(c)) return rc;
  hipStream_t st = (hipStream_t)stream;
  flush_pending_potrf(c, st, A, true);
  if (int rc = potrf_launch(c, A, n, lda, st, c->D.info)) return rc;
  HIPCHK(end_call(c));
  int rc = fetch_info(c, st);
  if (!rc && !use_generic(c)) { c->D.hinv_tag = A; c->D.hinv_n = n; }
  return rc;
}
// potrs with the factor of dense_potrf.  A single right-hand side of a factor produced by the blocked dense_potrf
// (its diagonal-block inverses are still cached) runs as 2 * ceil(n / 64) block steps over the chip.
static int potrs_impl(csp_ctx* c, const double* A, int64_t n, int64_t lda, double* B, int64_t nrhs, int64_t ldb, hipStream_t st) {
  DeviceCtx& D = c->D;
  static int olds = -1;
  if (olds < 0) { const char* e = sw_str("SMCP_POTRS_OLD"); olds = (e && e[0] == '1') ? 1 : 0; }
  if (olds) {
    launch(c, KID_dense_potrs, k_dense_potrs, dim3(1), dim3(1024), st, A, (int)n, lda, B, (int)nrhs, ldb);
    return 0;
  }
  if (nrhs == 1 && D.hinv_tag == A && D.hinv_n == n && n <= 2 * LB) {
    static bool attr = false;
    if (!attr) attr = hipFuncSetAttribute((const void*)k_dense_potrs_small, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) == hipSuccess;
    launch_lds(c, KID_dense_potrs, k_dense_potrs_small, dim3(1), dim3(256), (size_t)n * (n | 1) * sizeof(double), st, A, (int)n, lda,
               (const double*)D.hinv, B);
    return 0;
  }
  static int steps_only = -1;
  if (steps_only < 0) { const char* e = sw_str("SMCP_POTRS_STEPS"); steps_only = (e && e[0] == '1') ? 1 : 0; }
  if (nrhs == 1 && n > 2 * LB && n <= POTRS1_MAXN && !steps_only && !use_generic(c)) {
    // one launch of one workgroup: the whole substitution chain with the factor streamed a step ahead (k_dense_potrs_one)
    static bool attr1 = false;
    if (!attr1) attr1 = hipFuncSetAttribute((const void*)k_dense_potrs_one, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) == hipSuccess;
    if (attr1 && potrs_one_lds((int)n) <= (size_t)(160 * 1024 - 1024)) {
      launch_lds(c, KID_dense_potrs, k_dense_potrs_one, dim3(1), dim3(1024), potrs_one_lds((int)n), st, A, (int)n, lda, B);
      return 0;
    }
  }
  if (nrhs == 1 && D.hinv_tag == A && D.hinv_n == n && n > 2 * LB) {
    const int64_t nblocks = (n + LB - 1) / LB;
    double* y = D.hinv + nblocks * LB * LB;      // forward solution
    double* z = y + n;                            // backward solution
    for (int jb = 0; jb < (int)n; jb += LB) {
      const int w = (int)std::min<int64_t>(LB, n - jb);
      const int rest = (int)n - jb - w;
      launch(c, KID_dense_potrs, k_dense_trsv_step, dim3((unsigned)std::max(1, (rest + 255) / 256)), dim3(256), st, A, (int)n, lda,
             (const double*)(D.hinv + (int64_t)(jb / LB) * LB * LB), jb, w, B, y, 0);
    }
    for (int jb = (int)((nblocks - 1) * LB); jb >= 0; jb -= LB) {
      const int w = (int)std::min<int64_t>(LB, n - jb);
      launch(c, KID_dense_potrs, k_dense_trsv_step, dim3((unsigned)std::max(1, std::min(256, (jb + 3) / 4))), dim3(256), st, A, (int)n, lda,
             (const double*)(D.hinv + (int64_t)(jb / LB) * LB * LB), jb, w, y, z, 1);
    }
    HIPCHK(hipMemcpyAsync(B, z, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  launch(c, KID_dense_potrs, k_dense_potrs, dim3(1), dim3(1024), st, A, (int)n, lda, B, (int)nrhs, ldb);
  return 0;
}
int dense_potrs(csp_ctx* c, const double* A, int64_t n, int64_t lda, double* B, int64_t nrhs, int64_t ldb,
                void* stream) {
  if (int rc = ready(c)) return rc;
  if (int rc = flush_pending_potrf(c, (hipStream_t)stream, A, false)) return rc;
  if (int rc = potrs_impl(c, A, n, lda, B, nrhs, ldb, (hipStream_t)stream)) return rc;
  HIPCHK(end_call(c));
  return 0;
}

static bool use_gram(const csp_ctx* c) {
  static int g = -1;
  if (g < 0) { const char* e = sw_str("SMCP_GRAM"); g = (e && e[0] == '0') ? 0 : 1; }
  return g == 1 && !use_generic(c);
}

// Gram formulation of the whole Schur complement (what kkt_qr implies, solvers.py:414-420):
// H = G(A)^T G(A) with ONE leaves->root sweep per constraint, then one tall-skinny SYRK.
// ---- Gram formulation, in steps so that the multi-GPU driver can interleave the boundary exchange
static bool leafgram_ok(csp_ctx* c, int64_t mcols);
static int gram_prepare(csp_ctx* c, const double* L, const double* Y, hipStream_t st, bool allow_partial_fac = false) {
  DeviceCtx& D = c->D;
  const int64_t m = D.m, bl = c->S.blklen();
  D.qr_valid = false;      // the stack is about to be rewritten
  D.lg_nochild = false;    // ... by sweeps that decide anew whether the family children's panels are formed
  // the flag read after the sweeps must be theirs: a failed dense_potrf (Schur complement not positive definite) or
  // kkt_qr_factor leaves its own behind
  HIPCHK(zero_flag(c, st));
  prepare_yaa(c, Y, true, st, false, allow_partial_fac);
  if (int rc = prep_lk_cached(c, L, Y, st)) return rc;
  if (!D.kc_ptr) {   // the sweeps read their input from the stack: clear it and scatter the constraints into it
    HIPCHK(hipMemsetAsync(D.ustack, 0, sizeof(double) * m * bl, st));
    for (int64_t jb = 0; jb < m; jb += 65535)
      launch(c, KID_scatter_constraints, k_scatter_constraints, dim3(8, (unsigned)std::min<int64_t>(65535, m - jb)),
             dim3(256), st, jb, D.cptr, D.cidx, D.cval, D.ustack + jb * bl, bl);
  }
  HIPCHK(end_call(c));
  return 0;
}
// sorted blkval ranges with touching ones merged
static std::vector<std::pair<int64_t, int64_t>> gram_merge_ranges(int64_t nranges, const int64_t* ranges) {
  std::vector<std::pair<int64_t, int64_t>> rs;
  for (int64_t q = 0; q < nranges; ++q)
    if (ranges[2 * q + 1] > ranges[2 * q]) rs.push_back({ranges[2 * q], ranges[2 * q + 1]});
  std::sort(rs.begin(), rs.end());
  size_t w = 0;
  for (size_t q = 0; q < rs.size(); ++q) {
    if (w && rs[q].first <= rs[w - 1].second) rs[w - 1].second = std::max(rs[w - 1].second, rs[q].second);
    else rs[w++] = rs[q];
  }
  rs.resize(w);
  return rs;
}
// chunking of a Gram accumulation over `total` rows: ~one resident wave of workgroups (2 per CU) over all ranges
static int64_t gram_chunk_rows(int64_t total) {
  static int64_t minrows = 0;     // SMCP_GRAM_MINCHUNK (timing studies): smallest chunk of rows per workgroup
  if (!minrows) { const char* e = sw_str("SMCP_GRAM_MINCHUNK"); minrows = e ? atoll(e) : 512; if (minrows < GRAM_KS) minrows = GRAM_KS; }   // 512: one rank's share of an 8-rank job 0.202 -> 0.173 ms (with 2048 rows per workgroup two thirds of the chip idle); large problems are chunked by the second term
  return std::max<int64_t>(minrows, ((total / 512 + GRAM_KS - 1) / GRAM_KS) * GRAM_KS);
}
static int gram_count_chunks(const std::vector<std::pair<int64_t, int64_t>>& rs, int64_t chunk) {
  int n = 0;
  for (auto& r : rs) n += (int)((r.second - r.first + chunk - 1) / chunk);
  return n;
}
static int gram_reserve(csp_ctx* c, int64_t m, int nchunk) {
  DeviceCtx& D = c->D;
  const int nb = (int)((m + GRAM_BLK - 1) / GRAM_BLK), nblk = nb * (nb + 1) / 2;
  const int64_t need = (int64_t)nblk * nchunk * 64 * 256;
  if (D.gpart_len < need) {
    if (D.gpart) { HIPCHK(hipFree(D.gpart)); D.bytes -= D.gpart_len * 8; }
    D.gpart = nullptr;
    if (int rc = dev_alloc(&D.gpart, need, D.bytes)) return rc;
    D.gpart_len = need;
  }
  return 0;
}
// ---- slice table of a set of blkval ranges (k_gram_diag128) and the family children inside them (k_leaf_gram) ----------
// Built on the host when the ranges (or the leaf switch) differ from the cached ones: a rank's ranges are the same at
// every step.  leaf: the panels of the family children are not in the stack -- their rows are left out of the slices and
// the cliques are listed for k_leaf_gram.
static int64_t gram_chunk_rows(int64_t total);
// early_first (leaf tables only): the rows of the cliques of levels 0 and 1 first, padded with empty slices to whole chunks, then
// the rest -- D.gsl_spw / D.gsl_early tell the consumers (the early part can be accumulated while the sweep is still above)
static int gram_tables(csp_ctx* c, const std::vector<std::pair<int64_t, int64_t>>& rs, bool leaf, hipStream_t st, bool early_first = false) {
  DeviceCtx& D = c->D;
  const Symbolic& S = c->S;
  std::vector<int64_t> key;
  key.reserve(2 * rs.size() + 1);
  for (auto& r : rs) { key.push_back(r.first); key.push_back(r.second); }
  key.push_back((leaf ? 1 : 0) | (early_first ? 2 : 0));
  if (key == c->gsl_key && D.gsl_start) return 0;
  std::vector<int64_t> start, late_start;
  std::vector<int32_t> len, late_len, list, slot;
  int nf = 0, nn = 0, na = 0;
  auto add_segment = [&](int64_t lo, int64_t hi) {
    // (early_first: a segment is cut at clique boundaries and each piece goes to the part its clique's level says)
    if (!early_first) {
      for (int64_t e = lo; e < hi; e += GRAM_KS) { start.push_back(e); len.push_back((int32_t)std::min<int64_t>(GRAM_KS, hi - e)); }
      return;
    }
    int64_t k = (int64_t)(std::upper_bound(S.blkptr.begin(), S.blkptr.end(), lo) - S.blkptr.begin()) - 1;
    for (int64_t p = lo; p < hi; ++k) {
      const int64_t q = std::min(hi, S.blkptr[k + 1]);
      const bool late = S.level[(size_t)k] >= 2;
      for (int64_t e = p; e < q; e += GRAM_KS) {
        (late ? late_start : start).push_back(e);
        (late ? late_len : len).push_back((int32_t)std::min<int64_t>(GRAM_KS, q - e));
      }
      p = q;
    }
  };
  for (auto& r : rs) {
    if (!leaf) { add_segment(r.first, r.second); continue; }
    int64_t k = (int64_t)(std::upper_bound(S.blkptr.begin(), S.blkptr.end(), r.first) - S.blkptr.begin()) - 1;
    int64_t seg_lo = r.first;
    for (; k < S.nsn && S.blkptr[k] < r.second; ++k) {
      if (c->fam[k] != 1) continue;
      const int64_t b = std::max(S.blkptr[k], r.first), e = std::min(S.blkptr[k + 1], r.second);
      if (b != S.blkptr[k] || e != S.blkptr[k + 1]) return SMCP_EINVAL;     // ranges are unions of whole cliques
      if (b > seg_lo) add_segment(seg_lo, b);
      seg_lo = e;
      list.push_back((int32_t)k);
      slot.push_back(c->lg_slot_of[(size_t)k]);
      nf = std::max(nf, (int)S.nf(k)); nn = std::max(nn, (int)S.nn(k)); na = std::max(na, (int)S.na(k));
    }
    if (r.second > seg_lo) add_segment(seg_lo, r.second);
  }
  D.gsl_spw = 0; D.gsl_early = 0;
  if (early_first) {
    const int64_t chunk = gram_chunk_rows((int64_t)(start.size() + late_start.size()) * GRAM_KS);
    const int spw = (int)(chunk / GRAM_KS);
    while (start.size() % (size_t)spw) { start.push_back(0); len.push_back(0); }       // whole chunks of early rows
    D.gsl_spw = spw;
    D.gsl_early = late_start.empty() ? 0 : (int)(start.size() / (size_t)spw);
    start.insert(start.end(), late_start.begin(), late_start.end());
    len.insert(len.end(), late_len.begin(), late_len.end());
  }
  auto grow = [&](auto** p, int64_t& cap, int64_t need) -> int {
    if (cap >= need && *p) return 0;
    if (*p) { if (hipFree(*p) != hipSuccess) return SMCP_EHIP; *p = nullptr; }
    using T = typename std::remove_pointer<typename std::remove_pointer<decltype(p)>::type>::type;
    T* q = nullptr;
    if (hipMalloc((void**)&q, (size_t)std::max<int64_t>(need, 1) * sizeof(T)) != hipSuccess) return SMCP_ENOMEM;
    *p = q; cap = need;
    return 0;
  };
  HIPCHK(hipStreamSynchronize(st));             // the tables may be in use by launches still queued
  int64_t cap_len = D.gsl_cap;
  if (int rc = grow(&D.gsl_start, D.gsl_cap, (int64_t)start.size())) return rc;
  if (int rc = grow(&D.gsl_len, cap_len, (int64_t)start.size())) return rc;
  int64_t cap_slot = D.lg_cap;
  if (int rc = grow(&D.lg_list, D.lg_cap, (int64_t)list.size())) return rc;
  if (int rc = grow(&D.lg_slot, cap_slot, (int64_t)list.size())) return rc;
  if (!start.empty()) {
    HIPCHK(hipMemcpy(D.gsl_start, start.data(), start.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(D.gsl_len, len.data(), len.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (!list.empty()) {
    HIPCHK(hipMemcpy(D.lg_list, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(D.lg_slot, slot.data(), slot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  D.gsl_n = (int)start.size();
  D.lg_cnt = (int)list.size();
  D.lg_nf = nf; D.lg_nn = nn; D.lg_na = na;
  c->gsl_key = key;
  return 0;
}

// may the Schur sweep of mcols constraints leave the panels of the family children to k_leaf_gram?  (SMCP_LEAFGRAM=0: no)
constexpr int LG_ECAP_MAX = 1024;
static bool leafgram_ok(csp_ctx* c, int64_t mcols) {
  static int on = -1;
  if (on < 0) { const char* e = sw_str("SMCP_LEAFGRAM"); on = (e && e[0] == '0') ? 0 : 1; }
  const DeviceCtx& D = c->D;
  if (!on || !c->leafgram_policy || use_generic(c) || !D.kc_ptr || !D.kc_ij || mcols > GRAM_BLK || mcols < 1 || D.lg_children <= 0) return false;
  if (D.lg_maxent > LG_ECAP_MAX) return false;
  // the pairs of entries cost ~ as much as half as many (row, constraint) pairs moved through HBM twice
  if (c->leafgram_policy < 2 && D.lg_pairs > D.lg_rows * mcols) return false;
  if (!D.lg_eptr || !D.lg_tab) return false;
  const int64_t np = mcols * (mcols + 1) / 2 + 1;
  const int ecap = (int)((std::max<int64_t>(D.lg_maxent, 2) + 1) & ~1);
  return np + leafgram_wave_doubles(D.lg_rec, ecap) <= (int64_t)(LDS_LIMIT / 8);
}

// partial tiles of the listed family children (gram_tables) -> the slots part[0 .. *nl) (at most leafgram_slots())
static int leafgram_slots(csp_ctx* c) {
  return c->D.ncu;
}
static int leafgram_partials(csp_ctx* c, int64_t mcols, const int32_t* ids, hipStream_t st, double* part, int* nl) {
  DeviceCtx& D = c->D;
  *nl = 0;
  if (!D.lg_cnt) return 0;
  const int ncu = leafgram_slots(c);
  const int64_t np = mcols * (mcols + 1) / 2;
  const int ecap = (int)((std::max<int64_t>(D.lg_maxent, 2) + 1) & ~1);
  const int wd = leafgram_wave_doubles(D.lg_rec, ecap);
  const int64_t lim = (int64_t)(LDS_LIMIT / 8);
  const int nw = (int)std::max<int64_t>(1, std::min<int64_t>(8, (lim - ((np + 1) & ~1)) / wd));
  const int nwg = (int)std::max<int64_t>(1, std::min<int64_t>(ncu, (D.lg_cnt + nw - 1) / nw));
  LeafGramArgs a;
  a.cl = D.cl; a.list = D.lg_list; a.slot = D.lg_slot; a.cnt = D.lg_cnt; a.LK = D.lk; a.yaa = D.yaa;
  a.tab = D.lg_tab; a.rec = D.lg_rec;
  a.eptr = D.lg_eptr; a.epk = D.lg_epk; a.ew = D.lg_ew; a.remap = ids ? D.lg_remap : nullptr;
  a.nr = (int)mcols; a.ecap = ecap;
  a.part = part; a.info = D.info;
  { static int sk = -1; if (sk < 0) { const char* e = sw_str("SMCP_LGSKIP"); sk = e ? atoi(e) : 0; } a.skip = sk; }
  launch_lds(c, KID_leaf_tables, k_leaf_tables, dim3(D.lg_cnt), dim3(64), (size_t)leaftab_doubles(D.lg_nf, D.lg_nn, D.lg_na) * sizeof(double), st,
             a, D.lg_nf, D.lg_nn, D.lg_na);
  const size_t lds = (size_t)(((np + 1) & ~1) + (int64_t)nw * wd) * sizeof(double);
  const int rt = (D.lg_rec + 63) / 64, et = (ecap + 63) / 64;
  bool ok = false;
#define SMCP_LG_CASE(RT, ET) \
  if (!ok && rt <= RT && et <= ET) { \
    static bool attr = false; \
    if (!attr) attr = hipFuncSetAttribute((const void*)k_leaf_pairs<RT, ET>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) == hipSuccess; \
    if (!attr) return SMCP_EHIP; \
    launch_lds(c, KID_leaf_gram, k_leaf_pairs<RT, ET>, dim3(nwg), dim3(64 * nw), lds, st, a); \
    ok = true; \
  }
  SMCP_LG_CASE(12, 2) SMCP_LG_CASE(12, 4) SMCP_LG_CASE(24, 2) SMCP_LG_CASE(24, 4) SMCP_LG_CASE(24, 16)
  SMCP_LG_CASE(48, 4) SMCP_LG_CASE(48, 16)
#undef SMCP_LG_CASE
  if (!ok) return SMCP_EINVAL;
  *nl = nwg;
  return 0;
}

// k_gram_diag128 over the chunks c0 .. c1 - 1 of the slice table (spw slices each) into the partial slots of the same numbers
static void gram_launch_chunks(csp_ctx* c, int64_t m, int spw, int c0, int c1, int nchunk_total, hipStream_t st) {
  DeviceCtx& D = c->D;
  const int64_t bl = c->S.blklen();
  static int nw = -1;
  if (nw < 0) { const char* e = sw_str("SMCP_GRAM_NW"); nw = (e && e[0] == '4') ? 4 : ((e && e[0] == '8') ? 8 : 16); }
  const int mti = (int)((m + 15) / 16), npw = (mti * (mti + 1) / 2 + nw - 1) / nw;   // lower tiles per wave
  static int gskip = -1;     // ablation switch for timing studies only (SMCP_GSKIP: 1 = no MFMA phase, 2 = no global loads)
  if (gskip < 0) { const char* e = sw_str("SMCP_GSKIP"); gskip = e ? atoi(e) : 0; }
  const size_t lds = (size_t)GRAM_BLK * GRAM_LDK * sizeof(double);
  const int64_t* ss = D.gsl_start + (int64_t)c0 * spw;
  const int32_t* sl = D.gsl_len + (int64_t)c0 * spw;
  const int nsl = std::min<int>(D.gsl_n - c0 * spw, (c1 - c0) * spw);
  const dim3 grid((unsigned)(c1 - c0), 1);
#define SMCP_GRAM_CASE(N) case N: if (nw == 16) launch_lds(c, KID_gram_diag128, k_gram_diag128<(N <= 3 ? N : 3), 16>, grid, dim3(1024), lds, st, \
               (const double*)D.ustack, bl, (int)m, ss, sl, nsl, spw, (const double*)D.sw, D.gpart, c0, nchunk_total, gskip); \
             else if (nw == 8) launch_lds(c, KID_gram_diag128, k_gram_diag128<(N <= 5 ? N : 5), 8>, grid, dim3(512), lds, st, \
               (const double*)D.ustack, bl, (int)m, ss, sl, nsl, spw, (const double*)D.sw, D.gpart, c0, nchunk_total, gskip); \
             else launch_lds(c, KID_gram_diag128, k_gram_diag128<N, 4>, grid, dim3(256), lds, st, \
               (const double*)D.ustack, bl, (int)m, ss, sl, nsl, spw, (const double*)D.sw, D.gpart, c0, nchunk_total, gskip); break;
  switch (npw) {
    SMCP_GRAM_CASE(1) SMCP_GRAM_CASE(2) SMCP_GRAM_CASE(3) SMCP_GRAM_CASE(4) SMCP_GRAM_CASE(5)
    SMCP_GRAM_CASE(6) SMCP_GRAM_CASE(7) SMCP_GRAM_CASE(8) SMCP_GRAM_CASE(9)
  }
#undef SMCP_GRAM_CASE
}

// H = sum over the given blkval ranges of G^T W G (ranges: host array of nranges (begin, end) pairs); leaf: the stack
// lacks the panels of the family children (D.lg_nochild), whose Gram block k_leaf_gram supplies
static int gram_accumulate(csp_ctx* c, int64_t nranges, const int64_t* ranges, double* H, int64_t ldh, hipStream_t st,
                           int64_t mcols = -1, bool leaf = false, const int32_t* ids = nullptr) {
  DeviceCtx& D = c->D;
  const int64_t m = mcols < 0 ? D.m : mcols, bl = c->S.blklen();
  const std::vector<std::pair<int64_t, int64_t>> rs = gram_merge_ranges(nranges, ranges);
  int64_t total = 0;
  for (auto& r : rs) total += r.second - r.first;
  if (total <= 0) {
    HIPCHK(hipMemset2DAsync(H, ldh * sizeof(double), 0, m * sizeof(double), m, st));
    return 0;
  }
  const int nb = (int)((m + GRAM_BLK - 1) / GRAM_BLK);
  const int nblk = nb * (nb + 1) / 2;
  if (nblk == 1) {
    int spw, nchunk, ngram, nl = 0;
    Fork* side = (Fork*)c->side_fork;
    c->side_fork = nullptr;
    bool early_done = false;
    int early = 0;
    if (c->gpre.valid && leaf) {        // planned, and the leaf partials started, beside the large-front stage of the sweep (schur_gram)
      spw = c->gpre.spw; nchunk = c->gpre.nchunk; ngram = c->gpre.ngram; nl = c->gpre.nl;
      early_done = c->gpre.early_done; early = c->gpre.early;
    } else {
      if (side) { side->join(); delete side; side = nullptr; }
      if (int rc = gram_tables(c, rs, leaf, st)) return rc;
      const int64_t chunk = gram_chunk_rows((int64_t)D.gsl_n * GRAM_KS);
      spw = (int)(chunk / GRAM_KS);
      nchunk = std::max(1, (D.gsl_n + spw - 1) / spw);
      ngram = D.gsl_n > 0 ? nchunk : 0;
      if (int rc = gram_reserve(c, m, nchunk + (leaf ? leafgram_slots(c) : 0) + GRAM_RZ)) return rc;
      if (leaf) { if (int rc = leafgram_partials(c, m, ids, st, D.gpart + (int64_t)ngram * (64 * 256), &nl)) return rc; }
    }
    c->gpre.valid = false; c->gpre.early_done = false;
    // sixteen waves per workgroup (two workgroups per CU: eight waves per SIMD hide the staging and operand latency:
    // 0.96 ms with four waves, 0.75 with eight, 0.72 with sixteen; SMCP_GRAM_NW=4 / 8 select the smaller variants)
    static int nw = -1;
    if (nw < 0) { const char* e = sw_str("SMCP_GRAM_NW"); nw = (e && e[0] == '4') ? 4 : ((e && e[0] == '8') ? 8 : 16); }
    const int mti = (int)((m + 15) / 16), npw = (mti * (mti + 1) / 2 + nw - 1) / nw;   // lower tiles per wave
    static int gskip = -1;     // ablation switch for timing studies only (SMCP_GSKIP: 1 = no MFMA phase, 2 = no global loads)
    if (gskip < 0) { const char* e = sw_str("SMCP_GSKIP"); gskip = e ? atoi(e) : 0; }
    const size_t lds = (size_t)GRAM_BLK * GRAM_LDK * sizeof(double);
    (void)nw; (void)npw; (void)gskip; (void)lds;
    // (the chunks before c0 were accumulated beside the large-front stage of the sweep: gpre.early_done)
    const int c0 = early_done ? early : 0;
    if (D.gsl_n > 0 && nchunk > c0) gram_launch_chunks(c, m, spw, c0, nchunk, nchunk, st);
    if (side) { side->join(); delete side; }      // the leaf partials of the side branch
    if (ngram + nl >= 256 && D.gpart_len >= (int64_t)(ngram + nl + GRAM_RZ) * (64 * 256)) {
      // two stages: GRAM_RZ workgroups per tile, then their GRAM_RZ sums (slots behind the partial tiles: gram_reserve)
      double* const stage = D.gpart + (int64_t)(ngram + nl) * (64 * 256);
      launch(c, KID_gram_reduce, k_gram_reduce, dim3(64, 1, GRAM_RZ), dim3(1024), st, (const double*)D.gpart, ngram + nl, (int)m, H, ldh, stage);
      launch(c, KID_gram_reduce, k_gram_reduce, dim3(64, 1), dim3(256), st, (const double*)stage, GRAM_RZ, (int)m, H, ldh, (double*)nullptr);
    } else
    launch(c, KID_gram_reduce, k_gram_reduce, dim3(64, 1), dim3(ngram + nl >= 64 ? 1024 : 256), st, (const double*)D.gpart, ngram + nl, (int)m, H, ldh, (double*)nullptr);
    HIPCHK(end_call(c));
    return 0;
  }
  if (c->side_fork) { Fork* f = (Fork*)c->side_fork; c->side_fork = nullptr; f->join(); delete f; }
  c->gpre.valid = false;
  if (leaf) return SMCP_EINVAL;                 // leafgram_ok admits one block only
  const int64_t chunk = gram_chunk_rows(total);
  const int nchunk = gram_count_chunks(rs, chunk);
  if (int rc = gram_reserve(c, m, nchunk)) return rc;
  int coff = 0;
  for (auto& r : rs) {
    const int64_t lo = r.first, hi = r.second;
    const int nc = (int)((hi - lo + chunk - 1) / chunk);
    for (int bi = 0; bi < nb; ++bi)
      for (int bj = 0; bj <= bi; ++bj) {
        if (bi == bj)
          launch_lds(c, KID_gram_partial, k_gram_partial<true>, dim3(nc), dim3(256), (size_t)GRAM_KS * GRAM_LD * sizeof(double), st,
                     (const double*)D.ustack, bl, (int)m, lo, hi, (const double*)D.sw, chunk, D.gpart, coff, nchunk, bi, bj);
        else
          launch_lds(c, KID_gram_partial, k_gram_partial<false>, dim3(nc), dim3(256), (size_t)2 * GRAM_KS * GRAM_LD * sizeof(double), st,
                     (const double*)D.ustack, bl, (int)m, lo, hi, (const double*)D.sw, chunk, D.gpart, coff, nchunk, bi, bj);
      }
    coff += nc;
  }
  launch(c, KID_gram_reduce, k_gram_reduce, dim3(64, nblk), dim3(nchunk >= 64 ? 1024 : 256), st, (const double*)D.gpart, nchunk, (int)m, H, ldh, (double*)nullptr);
  HIPCHK(end_call(c));
  return 0;
}

// Gram formulation of the whole Schur complement (what kkt_qr implies, solvers.py:414-420):
// H = G(A)^T G(A) with ONE leaves->root sweep per constraint, then one tall-skinny SYRK.
// part / nparts (kkt_schur_gram_part): this caller's share of the column-sparse constraints -- a contiguous range of the
// sparse list -- and, for part 0, the Gram block of the swept ones; H must have been cleared, the parts are summed
static int schur_gram(csp_ctx* c, const double* L, const double* Y, double* H, int64_t ldh, hipStream_t st, int64_t part = 0,
                      int64_t nparts = 1) {
  DeviceCtx& D = c->D;
  const int64_t m = D.m, bl = c->S.blklen();
  if (!D.ns) {
    if (part) return 0;
    // (sweeps that take the family children's Gram block from k_leaf_pairs do not read those children's chol(Y_AA))
    if (int rc = gram_prepare(c, L, Y, st, leafgram_ok(c, m) && m <= D.max_rhs)) return rc;
    // (Accumulating the Gram tiles of the lower levels on a side stream while the large fronts are still swept was
    // measured in round 2 and dropped: 5.92 against 5.61 ms per step -- the two stages only take CUs from each other.)
    D.lg_request = leafgram_ok(c, m);
    // The closed-form Gram blocks of the family children depend on the factors only, not on the sweep: with one chunk of
    // right-hand sides they start on a side stream as soon as the extend-add of the first large level has been launched and
    // run beside the phase kernels of the top fronts (MFMA at a few workgroups per CU against VALU / LDS work: they share
    // the CUs instead of taking them from each other).  SMCP_LG_SIDE=0: after the sweep, on the caller's stream.
    static int lgside = -1;
    if (lgside < 0) {
      const char* e = sw_str("SMCP_LG_SIDE");
      const char* d0 = sw_str("SMCP_ALDS_DYN");
      const char* d1 = sw_str("SMCP_RHS_SPLIT_DYN");
      // (the two-stream split of the right-hand sides -- the plain extend-add launch, hess_up_fast -- uses the same side streams)
      lgside = ((e && e[0] == '0') || (d0 && d0[0] == '0') || (d1 && d1[0] == '1')) ? 0 : 1;
    }
    c->gpre.valid = false;
    if (lgside && D.lg_request && m <= D.max_rhs && Fork::enabled() && (m + GRAM_BLK - 1) / GRAM_BLK == 1) {
      c->side_work = [c, m, bl](hipStream_t side) {
        DeviceCtx& D = c->D;
        // (started before the family launch has decided -- SMCP_LG_EARLY -- the work is discarded by gram_accumulate when that launch
        // kept the children's panels after all: D.lg_nochild false)
        if (!D.lg_nochild && !D.lg_request) return;  // the family launch kept the children's panels: the Gram kernel takes them
        const int64_t range[2] = {0, bl};
        const std::vector<std::pair<int64_t, int64_t>> rs = gram_merge_ranges(1, range);
        // (round 5, measured and NOT the default: the Gram chunks of the rows of levels 0 / 1 -- 88 % of them on synth50k -- on the side
        // branch beside the phase kernels of the top fronts: k_gram_diag128 0.36 -> 0.66 ms, k_lf_up2 0.24 -> 0.50, the step 3.22 ->
        // 3.37 ms; started earlier still, beside the fused extend-add, that kernel goes 0.45 -> 0.75.  As in round 2: the stages take
        // CUs, LDS and L2 from each other.  SMCP_GRAM_EARLY=1 runs it.)
        static const int gearly = sw_on("SMCP_GRAM_EARLY", 0);
        if (gram_tables(c, rs, true, side, gearly != 0)) return;
        const int64_t chunk = gram_chunk_rows((int64_t)D.gsl_n * GRAM_KS);
        const int spw = D.gsl_spw ? D.gsl_spw : (int)(chunk / GRAM_KS);
        const int nchunk = std::max(1, (D.gsl_n + spw - 1) / spw);
        const int ngram = D.gsl_n > 0 ? nchunk : 0;
        if (gram_reserve(c, m, nchunk + leafgram_slots(c) + GRAM_RZ)) return;
        int nl = 0;
        if (leafgram_partials(c, m, nullptr, side, D.gpart + (int64_t)ngram * (64 * 256), &nl)) return;
        c->gpre.valid = true; c->gpre.ngram = ngram; c->gpre.nchunk = nchunk; c->gpre.spw = spw; c->gpre.nl = nl;
        c->gpre.early = D.gsl_early; c->gpre.early_done = false;
      };
      // behind level 1 of the sweep (hess_up_fast): the panels of the levels 0 and 1 are final -- their chunks of the Gram
      // accumulation go to the side branch (after the leaf partials already queued there), beside the large fronts above
      c->mid_work = [c, m](hipStream_t mainst) {
        Fork* f = (Fork*)c->side_fork;
        if (!f || !f->on || !c->gpre.valid || c->gpre.early <= 0 || !c->D.lg_nochild) return;
        if (hipEventRecord(c->aux_fork, mainst) != hipSuccess || hipStreamWaitEvent(f->s, c->aux_fork, 0) != hipSuccess) return;
        gram_launch_chunks(c, m, c->gpre.spw, 0, c->gpre.early, c->gpre.nchunk, f->s);
        c->gpre.early_done = true;
      };
    }
    // The closed-form leaf blocks start at once (SMCP_LG_EARLY=0: beside the phase kernels of the top fronts, as in round 3), beside the first launches of the sweep (the stray leaves, the
    // families' tables), instead of beside the phase kernels of the top fronts
    static int lgearly = -1;
    if (lgearly < 0) lgearly = sw_on("SMCP_LG_EARLY", 1);
    if (lgearly && c->side_work) {
      std::function<void(hipStream_t)> w = std::move(c->side_work);
      c->side_work = nullptr;
      Fork* f = new Fork(c, st, 1);
      c->side_fork = f;
      w(f->s);
    }
    for (int64_t jb = 0; jb < m; jb += D.max_rhs) {
      int nr = (int)std::min(D.max_rhs, m - jb);
      hess_up_fast(c, D.ustack + jb * bl, nr, bl, D.fac, 2, st, 0, D.kc_ptr ? jb : -1);     // G(A_j) = (G_NN, R^T G_AN)
    }
    c->side_work = nullptr;
    c->mid_work = nullptr;
    D.lg_request = false;
    const int64_t range[2] = {0, bl};
    if (int rc = gram_accumulate(c, 1, range, H, ldh, st, -1, D.lg_nochild)) return rc;
    if (int f = fetch_info(c, st)) return f;   // chol(Y_AA) failure
    return 0;
  }
  // ---- hybrid (solvers.py:479-497): Gram block of the swept constraints + SCMcolumn2 columns of the sparse ones
  const int64_t md = part == 0 ? D.md : 0, n = c->S.n;
  const int64_t qlo = D.ns * part / nparts, qhi = D.ns * (part + 1) / nparts;
  const int32_t* owner = nullptr;
  if (nparts > 1) {
    std::vector<int32_t> ow((size_t)m, -1);
    for (int64_t p = 0; p < nparts; ++p)
      for (int64_t q = D.ns * p / nparts; q < D.ns * (p + 1) / nparts; ++q) ow[(size_t)c->h_slist[(size_t)q]] = (int32_t)p;
    if (!D.scm_owner) { if (int rc = dev_alloc(&D.scm_owner, m, D.bytes)) return rc; }
    HIPCHK(hipMemcpyAsync(D.scm_owner, ow.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));      // ow is a local
    owner = D.scm_owner;
  }
  if (md) {
    prepare_yaa(c, Y, true, st);
    if (int rc = prep_lk_cached(c, L, Y, st)) return rc;
    if (!D.kc_ptr) {
      HIPCHK(hipMemsetAsync(D.ustack, 0, sizeof(double) * md * bl, st));
      for (int64_t jb = 0; jb < md; jb += 65535)
        launch(c, KID_scatter_constraints, k_scatter_constraints_ids, dim3(8, (unsigned)std::min<int64_t>(65535, md - jb)),
               dim3(256), st, (const int32_t*)D.dlist + jb, D.cptr, D.cidx, D.cval, D.ustack + jb * bl, bl);
    }
    D.lg_nochild = false;
    D.lg_request = leafgram_ok(c, md);
    for (int64_t jb = 0; jb < md; jb += D.max_rhs) {
      int nr = (int)std::min(D.max_rhs, md - jb);
      hess_up_fast(c, D.ustack + jb * bl, nr, bl, D.fac, 2, st, 0, D.kc_ptr ? jb : -1, D.kc_ptr ? D.dlist : nullptr);
    }
    D.lg_request = false;
    const int64_t range[2] = {0, bl};
    if (int rc = gram_accumulate(c, 1, range, D.hd, md, st, md, D.lg_nochild, D.kc_ptr ? D.dlist : nullptr)) return rc;
    launch(c, KID_scatter_constraints, k_scatter_hd, dim3((unsigned)std::min<int64_t>(1024, (md * md + 255) / 256)), dim3(256), st,
           (const int32_t*)D.dlist, (int)md, (const double*)D.hd, H, ldh);
  }
  // sparse constraints in chunks of at most vcols columns of S^-1
  std::vector<int64_t> voff;
  for (int64_t q0 = qlo; q0 < qhi;) {
    int64_t q1 = q0, cols = 0;
    voff.clear();
    while (q1 < qhi && cols + (c->h_kptr[q1 + 1] - c->h_kptr[q1]) <= D.vcols && q1 - q0 < 65535) {
      voff.push_back(cols);
      cols += c->h_kptr[q1 + 1] - c->h_kptr[q1];
      ++q1;
    }
    if (q1 == q0) return SMCP_ENOMEM;   // cannot happen: vcols >= max |K_s|
    HIPCHK(hipMemsetAsync(D.vbuf, 0, sizeof(double) * cols * n, st));
    launch(c, KID_scatter_constraints, k_unit_columns, dim3((unsigned)((cols + 255) / 256)), dim3(256), st,
           (const int32_t*)D.kidx + c->h_kptr[q0], cols, D.vbuf, n);
    if (int rc = trsm_impl(c, L, Y, D.vbuf, cols, n, 0, st)) return rc;      // V = L^-T L^-1 E_K  (solvers.py:491-492)
    if (int rc = trsm_impl(c, L, Y, D.vbuf, cols, n, 1, st)) return rc;
    // chunk offsets: a small device array behind the reduction scratch would not fit 65535 entries; use tmp's head
    int64_t* dvoff = reinterpret_cast<int64_t*>(D.tmp);
    HIPCHK(hipMemcpyAsync(dvoff, voff.data(), sizeof(int64_t) * voff.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));    // voff is reused by the next chunk
    launch(c, KID_scatter_constraints, k_scm_columns, dim3((unsigned)m, (unsigned)(q1 - q0)), dim3(64), st, m, D.cptr,
           (const int32_t*)D.a_r, (const int32_t*)D.a_c, (const double*)D.cval, (const int32_t*)D.slist + q0,
           (const int64_t*)dvoff, (const int32_t*)D.s_rloc, (const int32_t*)D.s_cloc, (const double*)D.vbuf, n, H, ldh, owner, (int)part);
    q0 = q1;
  }
  HIPCHK(end_call(c));
  if (md) { if (int f = fetch_info(c, st)) return f; }
  return 0;
}

int kkt_schur_columns(csp_ctx* c, const double* L, const double* Y, double* H, int64_t ldh, int64_t j0,
                      int64_t j1, void* stream) {
  if (int rc = ready(c)) return rc;
  DeviceCtx& D = c->D;
  const int64_t m = D.m, bl = c->S.blklen();
  if (!m || ldh < m || j0 < 0 || j1 > m || j0 > j1) return SMCP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  D.qr_valid = false;
  flush_pending_potrf(c, st, H, true);
  HIPCHK(zero_flag(c, st));
  if (j0 == 0 && j1 == m && use_gram(c)) return schur_gram(c, L, Y, H, ldh, st);
  prepare_yaa(c, Y, false, st);
  if (!use_generic(c)) { if (int rc = prep_lk_cached(c, L, Y, st)) return rc; }
  for (int64_t jb = j0; jb < j1; jb += D.max_rhs) {
    int nr = (int)std::min(D.max_rhs, j1 - jb);
    HIPCHK(hipMemsetAsync(D.ustack, 0, sizeof(double) * nr * bl, st));
    launch(c, KID_scatter_constraints, k_scatter_constraints, dim3(8, nr), dim3(256), st, jb, D.cptr, D.cidx,
           D.cval, D.ustack, bl);
    hessian_impl(c, L, D.ustack, nr, bl, 2, 0, st);
    // H[:, jb+r] = Amap(W(A_{jb+r}))  (full column; H is symmetric)
    amap_impl(c, D.ustack, bl, nr, H + jb * ldh, ldh, st);
  }
  HIPCHK(end_call(c));
  return 0;
}

// The hybrid Schur complement (Gram block of the swept constraints + SCMcolumn2 columns of the column-sparse ones,
// solvers.py:479-497, misc.c:620-663) sharded over `nparts` callers by sparse constraint: this one computes the columns of
// its contiguous share of the sparse list (trsm x 2 + k_scm_columns per chunk) and, as part 0, the Gram block.  H (cleared
// by the caller) receives this part only; the sum of the parts over all callers is the Schur complement (both triangles).
int kkt_schur_gram_part(csp_ctx* c, const double* L, const double* Y, double* H, int64_t ldh, int64_t part, int64_t nparts,
                        void* stream) {
  if (int rc = ready(c)) return rc;
  DeviceCtx& D = c->D;
  if (!D.m || ldh < D.m || nparts < 1 || part < 0 || part >= nparts || !use_gram(c)) return SMCP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  D.qr_valid = false;
  flush_pending_potrf(c, st, H, true);
  HIPCHK(zero_flag(c, st));
  return schur_gram(c, L, Y, H, ldh, st, part, nparts);
}
// how kkt_set_constraints classified the constraints (misc.nzcolumns / matperm, misc.c:682-773): counts[0] = swept through the
// Hessian (dense class), counts[1] = column-sparse (SCMcolumn2 route)
int kkt_constraint_classes(csp_ctx* c, int64_t* counts) {
  if (!c || !counts) return SMCP_EINVAL;
  counts[0] = c->D.ns ? c->D.md : c->D.m;
  counts[1] = c->D.ns;
  return 0;
}
static bool potrf_defer_on() {
  static int on = -1;
  if (on < 0) { const char* e = sw_str("SMCP_POTRF_DEFER"); on = (e && e[0] == '0') ? 0 : 1; }
  return on == 1;
}
int kkt_schur_factor(csp_ctx* c, const double* L, const double* Y, double* H, int64_t ldh, void* stream) {
  // ONE matrix can wait for its factorisation: a pending Schur complement of another system on this context (two KKT systems
  // may share a Symbolic) is factored where it stands before this one takes the slot, so its solve_ never meets a raw H
  if (c && c->D.h_pending && c->D.h_pending != H) { if (int rc = flush_pending_potrf(c, (hipStream_t)stream, nullptr, false)) return rc; }
  if (int rc = kkt_schur_columns(c, L, Y, H, ldh, 0, c ? c->D.m : 0, stream)) return rc;
  if (c->lazy_status && !use_generic(c) && potrf_defer_on() && Fork::enabled()) {
    // deferred status: nobody waits for potrf's verdict here, so the factorisation itself can wait for the first solve_
    // (kkt_solve runs it beside its first Hessian sweep) or for whoever reads H first (flush_pending_potrf)
    c->D.h_pending = H; c->D.h_pending_n = c->D.m; c->D.h_pending_ld = ldh; c->D.h_pending_stream = (hipStream_t)stream;
    c->D.hinv_tag = nullptr;
    return 0;
  }
  return dense_potrf(c, H, c->D.m, ldh, stream);
}
// The caller is about to free (or reuse) the memory of H: whatever the context still remembers about it -- the mark of a
// factorisation that kkt_schur_factor deferred, the cached inverses of its diagonal blocks -- is dropped, nothing is launched.
int kkt_schur_forget(csp_ctx* c, const double* H) {
  if (!c || !H) return SMCP_EINVAL;
  if ((const void*)c->D.h_pending == (const void*)H) c->D.h_pending = nullptr;
  if (c->D.hinv_tag == (const void*)H) c->D.hinv_tag = nullptr;
  return 0;
}

int kkt_solve(csp_ctx* c, const double* L, const double* Y, const double* H, int64_t ldh, double kk,
              double* bx, double* by, void* stream) {
  if (int rc = ready(c)) return rc;
  DeviceCtx& D = c->D;
  const int64_t m = D.m, bl = c->S.blklen();
  if (!m) return SMCP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  D.qr_valid = false;          // r1 below overwrites the first row of the stack
  double* r1 = D.ustack;       // blkval-sized temporaries
  double* ytmp = D.red + 600;  // (m <= 400 fits; larger m uses the tail of ustack)
  if (m > 400) {
    if (D.max_rhs < 2) return SMCP_ENOMEM;
    ytmp = D.ustack + bl;
  }
  HIPCHK(zero_flag(c, st));
  // a Schur complement that kkt_schur_factor left unfactored (deferred status): its Cholesky on a side stream beside the first
  // Hessian sweep, which does not read H -- with a failure flag of its own, latched by the branch itself
  std::unique_ptr<Fork> hf;
  if (D.h_pending && (const void*)D.h_pending == (const void*)H) {
    double* Hw = D.h_pending;
    D.h_pending = nullptr;
    // (only under deferred status: an eager caller -- the status regime may have been switched since kkt_schur_factor -- gets
    // the verdict of this factorisation as the return value, through the in-stream branch)
    hf.reset(c->lazy_status ? new Fork(c, st, 0) : nullptr);
    if (hf && hf->on) {
      int* pinfo = c->D.info + 20;
      if (int rc = potrf_launch(c, Hw, m, ldh, hf->s, pinfo)) return rc;
      hipLaunchKernelGGL(k_latch_status, dim3(1), dim3(64), 0, hf->s, pinfo, 1, c->D.info + 16);
    } else {                                    // no side stream: where it stands, status latched as after dense_potrf
      hf.reset();
      if (int rc = potrf_launch(c, Hw, m, ldh, st, c->D.info)) return rc;
      if (int rc = fetch_info(c, st)) return rc;
      HIPCHK(zero_flag(c, st));
    }
    if (!use_generic(c)) { D.hinv_tag = H; D.hinv_n = m; }
  } else if (int rc = flush_pending_potrf(c, st, nullptr, false)) return rc;    // (another matrix is pending: factor it where it stands)
  // the Y_AA cache must correspond to (L, Y): recompute (cheap, one gather sweep)
  if (!(c->D.yaa_tag == Y && c->D.yaa_tag)) prepare_yaa(c, Y, false, st);
  if (!use_generic(c)) { if (int rc = prep_lk_cached(c, L, Y, st)) return rc; }
  HIPCHK(hipMemcpyAsync(r1, bx, sizeof(double) * bl, hipMemcpyDeviceToDevice, st));
  hessian_impl(c, L, r1, 1, bl, 2, 0, st);                      // r1 = W(bx)
  {
    // Amap(r1), then y = kk*by + Amap(r1).  Few long constraints (synth50k: 100 of 11 k entries each) leave most of the chip idle
    // with one workgroup per constraint, each thread walking three rounds of dependent gathers: the entries are dealt over
    // `parts` workgroups per constraint and the shares added in order by the update of y (deterministic)
    const int64_t avg = D.m ? D.cnnz / D.m : 0;
    int parts = 1;
    if (avg > 4096 && m <= 400) parts = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)8, (2 * (int64_t)D.ncu) / std::max<int64_t>(m, 1), avg / 2048, (int64_t)3072 / std::max<int64_t>(m, 1)}));
    if (parts <= 1) {
      amap_impl(c, r1, 0, 1, ytmp, 0, st);
      launch(c, KID_vec_axpby, k_vec_axpby, dim3((unsigned)((m + 255) / 256)), dim3(256), st, m, 1.0, ytmp, kk, by);
    } else {
      double* const ypart = D.red + 1024;
      launch(c, KID_amap, k_amap, dim3((unsigned)m, 1, (unsigned)parts), dim3(1024), st, m, D.cptr, D.cidx, D.cwval, (const double*)r1, (int64_t)0,
             ypart, (int64_t)0, m);
      launch(c, KID_vec_axpby, k_vec_axpby_parts, dim3((unsigned)((m + 255) / 256)), dim3(256), st, m, 1.0, (const double*)ypart, parts, m, kk, by);
    }
  }
  if (hf) hf->join();                                           // H is factored from here on
  if (int rc = potrs_impl(c, H, m, ldh, by, 1, m, st)) return rc;
  // x = W(Aadj(y) - bx) / kk = -W(bx - Aadj(y)) / kk: the subtraction touches the constraint entries' positions only (no cleared
  // blkval-sized vector, no axpby over the whole of it: 28 us of three launches on synth50k), the sign rides on the final scaling
  if (D.rnnz)
    launch(c, KID_aadj, k_aadj_sub, dim3((unsigned)((D.rnnz + 255) / 256)), dim3(256), st, D.rnnz, D.rpos, D.rptr, D.rcon, D.rval,
           (const double*)by, bx);
  hessian_impl(c, L, bx, 1, bl, 2, 0, st);
  launch(c, KID_axpby, k_axpby, dim3(1024), dim3(256), st, bl, 0.0, (const double*)nullptr, -1.0 / kk, bx);
  HIPCHK(end_call(c));
  return 0;
}


// ---- subtree-sharded Gram path (multi-GPU; orchestrated by smcp_amd/kkt.py ShardedSchur) ----------
int csp_set_partition(csp_ctx* c, const int32_t* owner, int rank) {
  if (int rc = ready(c)) return rc;
  const Symbolic& S = c->S;
  HIPCHK(hipSetDevice(c->D.device));
  for (int set = 1; set <= 2; ++set) {
    LevelSet& LS = c->sets[set];
    if (LS.lev2) { HIPCHK(hipFree(LS.lev2)); LS.lev2 = nullptr; }
    std::vector<int32_t> lev2;
    const int want = (set == 1) ? rank : -1;
    classify_levels(S, [&](int64_t k) { return owner[k] == want; }, LS.lvl, lev2, LS.off);
    if (int rc = dev_upload(&LS.lev2, lev2, c->D.bytes)) return rc;
  }
  // subtree roots (owned clique whose parent lies in the replicated top), rank by rank in ascending clique order
  int world = 0;
  for (int64_t k = 0; k < S.nsn; ++k) world = std::max(world, owner[k] + 1);
  std::vector<int32_t> roots, own;
  std::vector<int64_t> bptr;
  c->xr_size.assign(std::max(world, rank + 1), 0);
  c->xr_npmax = 1;
  for (int r = 0; r < world; ++r)
    for (int64_t k = 0; k < S.nsn; ++k)
      if (owner[k] == r && S.snpar[k] >= 0 && owner[S.snpar[k]] == -1) {
        const int64_t np = S.na(k) * (S.na(k) + 1) / 2;
        roots.push_back((int32_t)k); own.push_back(r); bptr.push_back(c->xr_size[r]);
        c->xr_size[r] += np;
        c->xr_npmax = std::max(c->xr_npmax, np);
      }
  for (void* p : {(void*)c->xr_roots, (void*)c->xr_owner, (void*)c->xr_bptr}) if (p) HIPCHK(hipFree(p));
  c->xr_roots = nullptr; c->xr_owner = nullptr; c->xr_bptr = nullptr;
  c->xr_n = (int64_t)roots.size(); c->xr_me = rank; c->xr_world = world;
  // may the owned sweeps leave the family parents' updates to the extend-add above (fz_on)?  Only when no family parent of this
  // rank is a subtree root: a root's update is what the boundary exchange packs, so it has to exist in the exchange buffer
  c->fz_set1_ok = true;
  for (int64_t k = 0; k < S.nsn; ++k)
    if (k < (int64_t)c->fam.size() && c->fam[(size_t)k] == 2 && owner[k] == rank && (S.snpar[k] < 0 || owner[S.snpar[k]] != rank)) c->fz_set1_ok = false;
  if (int rc = dev_upload(&c->xr_roots, roots, c->D.bytes)) return rc;
  if (int rc = dev_upload(&c->xr_owner, own, c->D.bytes)) return rc;
  if (int rc = dev_upload(&c->xr_bptr, bptr, c->D.bytes)) return rc;
  return 0;
}
int kkt_gram_prepare(csp_ctx* c, const double* L, const double* Y, void* stream) {
  if (int rc = ready(c)) return rc;
  if (!c->D.m || use_generic(c)) return SMCP_EINVAL;
  return gram_prepare(c, L, Y, (hipStream_t)stream);
}
// the same when the factor (L, Y) is valid on the owned cliques and the top only: kkt_prepare_part has been called
// for both sets; this clears the failure flag and rewrites the input stack if the sweeps read their input from it
int kkt_gram_prepare_part(csp_ctx* c, void* stream) {
  if (int rc = ready(c)) return rc;
  DeviceCtx& D = c->D;
  if (!D.m || use_generic(c)) return SMCP_EINVAL;
  if (!D.part_valid) return SMCP_ESTALE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t m = D.m, bl = c->S.blklen();
  D.qr_valid = false;
  D.lg_nochild = false;
  if (!D.kc_ptr) {
    HIPCHK(hipMemsetAsync(D.ustack, 0, sizeof(double) * m * bl, st));
    for (int64_t jb = 0; jb < m; jb += 65535)
      launch(c, KID_scatter_constraints, k_scatter_constraints, dim3(8, (unsigned)std::min<int64_t>(65535, m - jb)),
             dim3(256), st, jb, D.cptr, D.cidx, D.cval, D.ustack + jb * bl, bl);
  }
  HIPCHK(end_call(c));
  return 0;
}
int kkt_gram_sweep(csp_ctx* c, int set, int64_t j0, int64_t j1, void* stream) {
  if (int rc = ready(c)) return rc;
  DeviceCtx& D = c->D;
  if (set < 0 || set > 2 || j0 < 0 || j1 > D.m || j1 <= j0 || j1 - j0 > D.max_rhs) return SMCP_EINVAL;
  if (set && !c->sets[set].lev2) return SMCP_EINVAL;
  const int64_t bl = c->S.blklen();
  D.lg_request = leafgram_ok(c, D.m);
  hess_up_fast(c, D.ustack + j0 * bl, (int)(j1 - j0), bl, D.fac, 2, (hipStream_t)stream, set, D.kc_ptr ? j0 : -1);
  D.lg_request = false;
  HIPCHK(end_call(c));
  return 0;
}
int kkt_gram_accumulate(csp_ctx* c, int64_t nranges, const int64_t* ranges, double* H, int64_t ldh, void* stream) {
  if (int rc = ready(c)) return rc;
  if (!c->D.m || ldh < c->D.m || nranges < 0) return SMCP_EINVAL;
  if (int rc = gram_accumulate(c, nranges, ranges, H, ldh, (hipStream_t)stream, -1, c->D.lg_nochild)) return rc;
  return fetch_info(c, (hipStream_t)stream);
}
// ---- sharded factorisation and solve (every leaves->root / root->leaves sweep shards by subtree) ----
// Y_AA blocks and their Cholesky factors of the cliques of one set (1 = owned, 2 = replicated top; call 2 before 1:
// a subtree root gathers from its parent's panel, which lies in the top); with_lk: also the inverse-form factor of L.
int kkt_prepare_part(csp_ctx* c, const double* L, const double* Y, int set, int with_lk, void* stream) {
  if (int rc = ready(c)) return rc;
  if (set < 1 || set > 2 || !c->sets[set].lev2 || use_generic(c) || !use_large()) return SMCP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  DeviceCtx& D = c->D;
  D.yaa_tag = D.fac_tag = D.faci_tag = nullptr;        // partial content: nothing to claim for the caches
  D.lk_tag_L = D.lk_tag_Y = nullptr;
  D.part_valid = false;
  D.part_Y = nullptr;
  D.fac_gen++;
  if (set == 2) HIPCHK(zero_flag(c, st));
  if (with_lk) prep_lk_set(c, set, L, st);
  gather_set(c, set, Y, 0, 1, D.yaa, st);
  {
    // fac <- yaa on the blocks of THIS set only: the other set's blocks may already hold their factors (a copy of the
    // whole array for set 1 used to put the unfactored Y_AA back over the factors of the top, set 2 -- unnoticed as long
    // as the top was a root without separator)
    TreeArgs t = tree_args(c);
    const LevelSet& LS = c->sets[set];
    for (int64_t l = 0; l < c->S.nlev; ++l) {
      const int cnt = (int)(LS.lvl[l].nI + LS.lvl[l].nII);
      if (!cnt) continue;
      t.lev = LS.lev2 + LS.off[l];
      const int namax = std::max(LS.lvl[l].namaxI, LS.lvl[l].namaxII);
      if (!namax) continue;
      launch(c, KID_axpby, k_copy_upd_blocks, dim3(cnt, umax1(std::min(64, (namax * namax + 1023) / 1024))), dim3(256), st, t,
             (const double*)D.yaa, D.fac);
    }
  }
  MfmaArgs a0 = mfma_args(c, nullptr, 0, 1);
  for (int64_t l = 0; l < c->S.nlev; ++l)
    for_level_classes(c, l, a0, [&](bool lds, MfmaArgs am, int cnt, size_t, int) {
      if (!am.namax) return;
      if (lds) {
        size_t bytes = ((size_t)padld(am.namax) * am.namax + 256 + 8) * sizeof(double);
        launch_lds(c, KID_factor_yaa_lds, k_factor_yaa_lds, dim3(cnt), dim3(fact_threads(am, 256, 2)), bytes, st, am, (const double*)D.yaa, D.fac);
      } else lf_factor_yaa(c, am, cnt, D.fac, st);
    }, set);
  HIPCHK(end_call(c));
  if (set == 2) return 0;
  if (int rc = fetch_info(c, st)) return rc;
  D.part_valid = true;
  D.part_Y = Y;          // (the root blocks of Y serve the fronts without separator in csp_hessian_sweep_part: Z_NN = Y_NN F_NN Y_NN)
  return 0;
}
// one half of the Hessian hessian(L, Y, U, adj=None) over the cliques of a set: dir 0 = leaves->root (with the Y_AA
// scaling of the separator rows), dir 1 = root->leaves.  The caller exchanges the packed updates of the subtree roots
// between the owned and the top pass of dir 0 (csp_exchange_pack / unpack); dir 1 needs no exchange.
int csp_hessian_sweep_part(csp_ctx* c, double* U, int64_t nrhs, int64_t ldu, int set, int dir, void* stream) {
  if (int rc = ready(c)) return rc;
  if (set < 1 || set > 2 || !c->sets[set].lev2 || use_generic(c) || nrhs < 1 || nrhs > c->D.max_rhs) return SMCP_EINVAL;
  if (!c->D.part_valid) return SMCP_ESTALE;
  hipStream_t st = (hipStream_t)stream;
  // (the two halves of a set always come as a pair -- the caller runs dir 1 of every set after dir 0 of every set -- so the
  // fronts without separator take both together, as in hessian_impl)
  if (dir == 0) hess_up_fast(c, U, (int)nrhs, ldu, c->D.yaa, 1, st, set, -1, nullptr, 0, -1, c->D.part_Y);
  else hess_down_fast(c, U, (int)nrhs, ldu, nullptr, 0, st, set, c->D.part_Y);
  HIPCHK(end_call(c));
  return 0;
}
// boundary exchange of the subtree partition: doubles per right-hand side that every rank contributes (host array of
// `world` entries), pack of this rank's subtree roots into buf ([root][rhs][packed block]) and unpack of all OTHER ranks'
// roots from the all-gathered buffers (`width` doubles per rank).  One launch each; nothing is allocated or waited for.
int csp_exchange_sizes(csp_ctx* c, int64_t world, int64_t* sizes) {
  if (!c || !sizes || world < 1) return SMCP_EINVAL;
  for (int64_t r = 0; r < world; ++r) sizes[r] = r < (int64_t)c->xr_size.size() ? c->xr_size[r] : 0;
  return 0;
}
static int exchange_roots(csp_ctx* c, int64_t nrhs, double* buf, int64_t width, int unpack, void* stream, int64_t r0 = 0) {
  if (int rc = ready(c)) return rc;
  if (c->xr_me < 0 || nrhs < 1 || r0 < 0 || r0 + nrhs > c->D.max_rhs || !buf) return SMCP_EINVAL;
  if (!c->xr_n) return 0;
  launch(c, KID_axpby, k_exchange_roots, dim3((unsigned)std::min<int64_t>(32, (c->xr_npmax + 255) / 256), (unsigned)c->xr_n, (unsigned)nrhs),
         dim3(256), (hipStream_t)stream, (const CliqueDesc*)c->D.cl, (const int32_t*)c->xr_roots, (const int32_t*)c->xr_owner,
         (const int64_t*)c->xr_bptr, c->xr_me, (int)nrhs, c->D.updp, c->D.updp_stride ? c->D.updp_stride : c->S.updplen(), buf, width, unpack, (int)r0);
  HIPCHK(end_call(c));
  return 0;
}
int csp_exchange_combine(csp_ctx* c, int64_t nrhs, const double* y, const double* gbuf, int64_t gwidth, double* out, int64_t owidth,
                         int mode, void* stream) {
  if (int rc = ready(c)) return rc;
  if (c->xr_me < 0 || nrhs < 1 || !y || !gbuf || !out || mode < 0 || mode > 1) return SMCP_EINVAL;
  if (!c->xr_n) return 0;
  launch(c, KID_axpby, k_exchange_combine, dim3((unsigned)std::min<int64_t>(32, (c->xr_npmax + 255) / 256), (unsigned)c->xr_n),
         dim3(256), (hipStream_t)stream, (const CliqueDesc*)c->D.cl, (const int32_t*)c->xr_roots, (const int32_t*)c->xr_owner,
         (const int64_t*)c->xr_bptr, c->xr_me, (int)nrhs, y, gbuf, gwidth, out, owidth, mode);
  HIPCHK(end_call(c));
  return 0;
}
int csp_exchange_pack(csp_ctx* c, int64_t nrhs, double* buf, void* stream) { return exchange_roots(c, nrhs, buf, 0, 0, stream); }
// The exchange BY CONSTRAINT SHARE (round 5: the top of the tree sharded by constraint instead of replicated for all of them):
// pack the right-hand sides r0 .. r0 + nrhs - 1 of the sweep that just ran (the share of ONE destination rank) as
// [root][rhs][packed block]; unpack the roots of EVERY rank -- this one's too -- from the all-to-all's receive buffer (`width`
// doubles per source rank) into the slots 0 .. nrhs - 1, where the top sweep of this rank's own share reads them.
int csp_exchange_pack_range(csp_ctx* c, int64_t r0, int64_t nrhs, double* buf, void* stream) { return exchange_roots(c, nrhs, buf, 0, 0, stream, r0); }
int csp_exchange_unpack_all(csp_ctx* c, int64_t nrhs, const double* buf, int64_t width, void* stream) {
  return exchange_roots(c, nrhs, const_cast<double*>(buf), width, 2, stream);
}
// rows [a, b) of the swept stack of the constraints j0 .. j1 - 1 <-> buf[(j - j0) * (b - a) + (row - a)]   (dir 0: stack -> buf,
// 1: buf -> stack): the top's panels of a rank's constraint share travel to the rank that accumulates the top's Gram block
__global__ void k_stack_rows(double* ustack, int64_t bl, int64_t j0, int64_t a, int64_t len, double* buf, int dir) {
  double* row = ustack + (j0 + blockIdx.y) * bl + a;
  double* b = buf + (int64_t)blockIdx.y * len;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += (int64_t)gridDim.x * blockDim.x) {
    if (dir) row[e] = b[e]; else b[e] = row[e];
  }
}
int kkt_stack_rows(csp_ctx* c, int dir, int64_t j0, int64_t j1, int64_t a, int64_t b, double* buf, void* stream) {
  if (int rc = ready(c)) return rc;
  DeviceCtx& D = c->D;
  if (!D.ustack || j0 < 0 || j1 > D.m || j1 < j0 || a < 0 || b > c->S.blklen() || b < a || !buf || dir < 0 || dir > 1) return SMCP_EINVAL;
  if (j1 == j0 || b == a) return 0;
  launch(c, KID_axpby, k_stack_rows, dim3((unsigned)std::min<int64_t>(256, (b - a + 255) / 256), (unsigned)(j1 - j0)), dim3(256), (hipStream_t)stream,
         D.ustack, c->S.blklen(), j0, a, b - a, buf, dir);
  HIPCHK(end_call(c));
  return 0;
}
int csp_exchange_unpack(csp_ctx* c, int64_t nrhs, const double* buf, int64_t width, void* stream) {
  return exchange_roots(c, nrhs, const_cast<double*>(buf), width, 1, stream);
}

}  // extern "C"
