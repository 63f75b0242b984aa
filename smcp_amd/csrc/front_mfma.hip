// LDS-resident / MFMA clique kernels (the fast path).
//
// Formulation: after a factorisation the "inverse-form" factor LK = [Li; K] with
// Li = L_NN^-1 (explicit, lower, zeros above) and K = L_AN Li is prepared once per clique.
// With it every per-right-hand-side step of the Hessian sweeps is a plain
// matrix product -- no triangular solves, no sequential dependency chains:
//   up   : E = F_AN - K F_NN/2 ; Upd = F_AA - K E^T - E K^T ; G_AN = (2E - F_AN) Li^T ;
//          G_NN = Li F_NN Li^T ; (fused scale) Q = Y_AA G_AN
//   down : QL = Q Li ; D = QL - Z_AA K/2 ; Z_AN = 2D - QL ; Z_NN = Li^T G_NN Li - K^T D - D^T K
// All products run on v_mfma_f64_16x16x4_f64 through wg_mma() below, with the working set of a
// (clique, rhs) pair staged in LDS when it fits (template LDS = true) or left in HBM/L2 scratch
// for big fronts (LDS = false; same code through flat pointers).  One workgroup owns one clique
// and loops over a group of right-hand sides so the clique constants are loaded once.
#include <hip/hip_runtime.h>

namespace smcp {

typedef double d4 __attribute__((ext_vector_type(4)));

// C(m,n) = sum_k A(m,k) B(k,n) on 16x16 MFMA tiles; all waves of the workgroup share the tiles.
// A(m,k) / B(k,n) are element functors (any memory), store(m,n,acc) consumes the result.
// MFMA operand map (v_mfma_f64_16x16x4_f64): the instruction's A operand holds [i=lane&15][k=lane>>4],
// its B operand [k=lane>>4][j=lane&15], D register r holds [i=(lane>>4)+4r][j=lane&15].  We feed
// B_my as the instruction's A and A_my as its B so that j (= lane&15) runs along the contiguous
// row index m of our column-major matrices.
template <class FA, class FB, class FC>
__device__ inline void wg_mma(int M, int N, int Kd, FA A, FB B, FC store, bool lower_only = false, int rot = 0) {
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = ((threadIdx.x >> 6) + nw - (rot % nw)) % nw;  // rot shifts which wave takes tile 0
  const int mt = (M + 15) >> 4, nt = (N + 15) >> 4;
  const int l15 = lane & 15, kq = lane >> 4;
  for (int t = wave; t < mt * nt; t += nw) {
    const int tm = t % mt, tn = t / mt;
    if (lower_only && tm < tn) continue;
    const int m = tm * 16 + l15, nb = tn * 16 + l15;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    // operands of KU k-steps are fetched before the KU MFMAs are issued, so the loads of a batch
    // overlap instead of each MFMA waiting for its own pair of loads
    constexpr int KU = 2;
    for (int k0 = 0; k0 < Kd; k0 += 4 * KU) {
      double av[KU], bv[KU];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int k = k0 + 4 * u + kq;
        const bool kin = k < Kd;
        av[u] = (kin && m < M) ? A(m, k) : 0.0;
        bv[u] = (kin && nb < N) ? B(k, nb) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < KU; ++u)
        if (k0 + 4 * u < Kd) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[u], av[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = tn * 16 + kq + 4 * r;
      if (m < M && n < N) store(m, n, acc[r]);
    }
  }
}

// Loop e = begin, begin+step, ... < end in batches of UNR: all loads of a batch are issued before
// any use, so a thread keeps UNR independent memory requests in flight instead of one.
template <int UNR, class LoadF, class UseF>
__device__ inline void batched_loop(int begin, int end, int step, LoadF load, UseF use) {
  for (int e0 = begin; e0 < end; e0 += step * UNR) {
    decltype(load(0)) v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int e = e0 + u * step;
      if (e < end) v[u] = load(e);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int e = e0 + u * step;
      if (e < end) use(e, v[u]);
    }
  }
}

struct MfmaArgs {
  TreeArgs t;
  const double* LK;   // inverse-form factor, blkval layout: rows 0..nn-1 = Li (zeros above diag), rows nn.. = K
  const double* ysc;  // scaling operand in update-matrix layout (yaa or its Cholesky factor) or null
  int ymode;          // 0 none, 1 symmetric (Y_AA), 2 R^T, 3 R
  int nnmax, namax;   // LDS layout sizing (max over the cliques of this launch)
  int nrhs;
  int nchmax, panmax, pkmax, plansum;  // LDS sizing of the index tables of the padded kernels
  int skip;  // debug-only phase mask (SMCP_SKIP env), 0 in production
  unsigned long long* dbg;  // diagnostic builds: cycle-stamp accumulator (null otherwise)
  double* lfd;              // 64 x 64 scratch per large front (inverse of the current diagonal block)
  int dn; int64_t dld;      // dense-matrix view (blocked Cholesky of the Schur complement): order, leading dimension
  // Sparse input of the Schur-complement sweeps: right-hand side r is the constraint matrix A_j, j = kc_ids ?
  // kc_ids[kc_j0 + r] : kc_j0 + r, given per clique k as the entries kc_ptr[k * kc_stride + j] .. [+1) of
  // (kc_off: position inside the clique's panel, kc_val).  The sweep then never reads its input from u (u is output
  // only), so the m x blklen stack needs neither clearing nor scattering.  null = dense input in u.
  const int32_t* kc_ptr; const int32_t* kc_off; const double* kc_val; const int32_t* kc_ids;
  int kc_stride, kc_j0;
  // sparse-input sweep of childless large fronts (front_lfsp.hip): R^T (update layout) and R^T K (blkval layout, K rows)
  const double* sp_rt; const double* sp_mk;
  // host-side only (level loop of hess_up_fast): level index and the family tail of the LDS class (LevelClass::nS ...)
  int level, nS, famna, fampan, fampk, famcna, famnn, famcnn;
  int nnmin;     // narrowest supernode of a large-front class
  // sibling groups of the sparse-input sweep of childless large fronts (front_lfsp.hip): group g = the cliques
  // grp_list[grp_ptr[g] .. grp_ptr[g + 1]); chskip[k] != 0: the packed update slot of child k is not written in this sweep
  const int32_t* grp_ptr; const int32_t* grp_list; const uint8_t* chskip;
  int famt_ngrp;   // host-side: sibling groups of the family parents of this launch (k_fam_terms_grp), 0 = none
  // Fused extend-add of the family parents' updates (front_large.hip, lf_add_family): when fz_on the family sweep of this
  // call (k_fam_terms) does NOT form the parents' update matrices; the extend-add of their parent front computes them from
  // the families' tables (fz_tab: the records of k_famt_prep, fz_recl doubles each, record of family f at slot fz_slot[f]) and
  // the static term lists of kkt_set_constraints (terms fz_ptr[f * fz_stride + j] .. [+1) of (fz_pk: vector ids vx | vy << 16,
  // fz_s: scale); fz_no: clique -> family number or -1) straight into the front it holds in LDS -- the packed updates never
  // reach HBM
  int fz_on, fz_nat, fz_cnn, fz_stride;
  int64_t fz_recl;
  const double* fz_tab; const int32_t* fz_no; const int32_t* fz_slot; const int32_t* fz_ptr; const int32_t* fz_pk; const double* fz_s;
};

__device__ __host__ inline int padld(int x) { return x | 1; }

// doubles of LDS needed by the largest kernel (hess_up with scaling) for a front (nn, na)
__host__ __device__ inline int64_t mfma_lds_doubles(int nn, int na) {
  int64_t lk = padld(na), ll = padld(nn), lf = padld(nn + na);
  int64_t nnc = nn;
  //        K        Li        Y              F         Fnn       E         G         T        Upd
  return lk * nnc + ll * nnc + lk * (int64_t)na + lf * nnc + ll * nnc + lk * nnc + lk * nnc + ll * nnc + lk * (int64_t)na + 16 * 16 + 8 + (na + 2) / 2;
}

// the LDS buffers a kernel uses (make_work / mfma_lds_doubles_for): kernels that need only some of them get a compact
// layout, i.e. more resident workgroups per CU (the factorisation kernels on the small fronts are rounds x latency)
enum : int { WK_K = 1, WK_LI = 2, WK_Y = 4, WK_F = 8, WK_FNN = 16, WK_E = 32, WK_G = 64, WK_T = 128, WK_U = 256, WK_ALL = 511 };
constexpr int WK_CHOL = WK_F | WK_U;                       // k_chol_mfma: front, update block (+ the 16 x 16 scratch)
constexpr int WK_PINV = WK_K | WK_LI | WK_E | WK_U;        // k_pinv_mfma: inverse-form factor, Y_AA, T = Y_AA K
constexpr int WK_DOWN0 = WK_ALL & ~(WK_Y | WK_FNN);        // k_hess_down_mfma without a scaling operand (ymode 0)
constexpr int WK_COMPL = WK_Y | WK_F | WK_FNN | WK_E | WK_G | WK_T;   // k_completion_mfma (no inverse-form factor, no update block)
__host__ __device__ inline int64_t mfma_lds_doubles_for(int mask, int nn, int na) {
  const int64_t lk = padld(na), ll = padld(nn), lf = padld(nn + na), nnc = nn;
  int64_t t = 0;
  if (mask & WK_K) t += lk * nnc;
  if (mask & WK_LI) t += ll * nnc;
  if (mask & WK_Y) t += lk * (int64_t)na;
  if (mask & WK_F) t += lf * nnc;
  if (mask & WK_FNN) t += ll * nnc;
  if (mask & WK_E) t += lk * nnc;
  if (mask & WK_G) t += lk * nnc;
  if (mask & WK_T) t += ll * nnc;
  if (mask & WK_U) t += lk * (int64_t)na;
  return t + 16 * 16 + 8 + (na + 2) / 2;
}
__host__ __device__ inline int64_t completion_lds_doubles(int nn, int na) {
  return mfma_lds_doubles_for(WK_COMPL, nn, na) + (nn > 16 ? 16 * 128 : 0);
}

struct Work {  // working-set pointers of one (clique, rhs) pair
  const double* K; int ldk;
  const double* Li; int ldl;
  const double* Y; int ldy;
  double* F; int ldf;     // panel nf x nn
  double* Fnn; int ldn;   // nn x nn full symmetric copy
  double* E; int lde;     // na x nn
  double* G; int ldg;     // na x nn
  double* T; int ldt;     // nn x nn
  double* U; int ldu;     // na x na update / separator matrix (lower)
  double* D16;            // 16 x 16 scratch
};

template <bool LDS, int MASK = WK_ALL>
__device__ inline Work make_work(const MfmaArgs& a, const CliqueDesc& d, double* smem, int k, int r) {
  Work w;
  const int nn = d.nn, na = d.na, nf = nn + na;
  if (LDS) {
    const int lk = padld(a.namax), ll = padld(a.nnmax), lf = padld(a.nnmax + a.namax);
    double* p = smem;
    // a buffer outside MASK gets no room (its pointer aliases the next one and must not be used)
    w.K = p; w.ldk = lk; if (MASK & WK_K) p += (int64_t)lk * a.nnmax;
    w.Li = p; w.ldl = ll; if (MASK & WK_LI) p += (int64_t)ll * a.nnmax;
    w.Y = p; w.ldy = lk; if (MASK & WK_Y) p += (int64_t)lk * a.namax;
    w.F = p; w.ldf = lf; if (MASK & WK_F) p += (int64_t)lf * a.nnmax;
    w.Fnn = p; w.ldn = ll; if (MASK & WK_FNN) p += (int64_t)ll * a.nnmax;
    w.E = p; w.lde = lk; if (MASK & WK_E) p += (int64_t)lk * a.nnmax;
    w.G = p; w.ldg = lk; if (MASK & WK_G) p += (int64_t)lk * a.nnmax;
    w.T = p; w.ldt = ll; if (MASK & WK_T) p += (int64_t)ll * a.nnmax;
    w.U = p; w.ldu = lk; if (MASK & WK_U) p += (int64_t)lk * a.namax;
    w.D16 = p;
  } else {
    w.Li = a.LK ? a.LK + d.blk : nullptr; w.ldl = nf;
    w.K = a.LK ? a.LK + d.blk + nn : nullptr; w.ldk = nf;
    w.Y = a.ysc ? a.ysc + d.upd : nullptr; w.ldy = na;
    double* s = a.t.tmp + (int64_t)r * a.t.tmplen + a.t.tmpptr[k];  // 2*nf*nn doubles
    w.Fnn = s; w.ldn = nn; s += (int64_t)nn * nn;
    w.T = s; w.ldt = nn; s += (int64_t)nn * nn;
    w.E = s; w.lde = na; s += (int64_t)na * nn;
    w.G = s; w.ldg = na; s += (int64_t)na * nn;
    w.D16 = s;                   // 256 doubles reserved per clique behind the 2*nf*nn scratch
    w.F = nullptr; w.ldf = nf;   // set per rhs (in place)
    w.U = nullptr; w.ldu = na;   // set per rhs (in place)
  }
  return w;
}

// load clique constants into LDS (LDS mode only)
__device__ inline void load_consts(const MfmaArgs& a, const CliqueDesc& d, const Work& w, bool needY) {
  const int nn = d.nn, na = d.na, nf = nn + na;
  if (a.LK) {
    const double* src = a.LK + d.blk;
    double* Li = const_cast<double*>(w.Li);
    double* K = const_cast<double*>(w.K);
    const int ldl = w.ldl, ldk = w.ldk;
    batched_loop<8>(threadIdx.x, nf * nn, blockDim.x, [=](int e) { return src[e]; },
                    [=](int e, double v) {
                      int i = e % nf, j = e / nf;
                      if (i < nn) Li[i + j * ldl] = v; else K[(i - nn) + j * ldk] = v;
                    });
  }
  if (needY && a.ysc) {
    const double* src = a.ysc + d.upd;
    double* Y = const_cast<double*>(w.Y);
    const int ldy = w.ldy;
    batched_loop<8>(threadIdx.x, na * na, blockDim.x,
                    [=](int e) { return (e % na) >= (e / na) ? src[e] : 0.0; },
                    [=](int e, double v) { if ((e % na) >= (e / na)) Y[(e % na) + (e / na) * ldy] = v; });
  }
}

// UNC: one unconditional load and a mask (the mirrored position of a triangular factor is allocated memory: zeros or
// scratch) -- what the sixteen-wave tile products of front_large.hip need (no branch around a load: see gemm_tile64)
template <int UNC = 1>
__device__ inline double yacc(const double* Y, int ld, int mode, int m, int k) {
  if constexpr (UNC == 4) {
    const bool tr = mode == 1 ? m < k : mode == 2;           // read (k, m) instead of (m, k)
    const int i = tr ? k : m, j = tr ? m : k;
    const double v = Y[i + j * ld];
    if (mode == 1) return v;
    return i >= j ? v : 0.0;                                  // R^T (mode 2) / R
  } else {
    if (mode == 1) return m >= k ? Y[m + k * ld] : Y[k + m * ld];
    if (mode == 2) return k >= m ? Y[k + m * ld] : 0.0;   // R^T
    return m >= k ? Y[m + k * ld] : 0.0;                    // R
  }
}

// children's update matrices (global, lower) scatter-added into the front [F | U].
// One wave per child, all children in flight at once; collisions between children are resolved
// by hardware fp64 atomic adds (ds_add_f64 / global_atomic_add_f64), so the summation order --
// and with it the last bit -- may differ between runs.  One barrier at the end.

// Atomic-free, deterministic extend-add: every front position that receives contributions is owned
// by one thread, which sums the children's entries in a fixed order (plan built at device_init).
// add(code, value) applies the sum to the front (code: bit 30 = update block, i | j << 15).
template <class AddF>
__device__ inline void gather_children_plan(const TreeArgs& t, int k, const double* updbase, AddF add) {
  const int64_t t0 = t.gp_tptr[k], t1 = t.gp_tptr[k + 1];
  for (int64_t tt = t0 + threadIdx.x; tt < t1; tt += blockDim.x) {
    const int32_t code = t.gp_tgt[tt];
    const int64_t c0 = t.gp_cptr[tt], c1 = t.gp_cptr[tt + 1];
    double acc = 0.0;
    int64_t cc = c0;
    for (; cc + 4 <= c1; cc += 4) {
      const int32_t s0 = t.gp_src[cc], s1 = t.gp_src[cc + 1], s2 = t.gp_src[cc + 2], s3 = t.gp_src[cc + 3];
      const double v0 = updbase[s0], v1 = updbase[s1], v2 = updbase[s2], v3 = updbase[s3];
      acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; cc < c1; ++cc) acc += updbase[t.gp_src[cc]];
    add(code, acc);
  }
}


// packed lower-triangular (column-major) storage of the exchanged update matrices
__device__ __host__ inline int pk_col(int j, int n) { return j * n - (j * (j - 1)) / 2; }
__device__ __host__ inline int pk_idx(int i, int j, int n) { return pk_col(j, n) + (i - j); }
// inverse map: packed position e -> (i, j), i >= j
__device__ inline void pk_unpack(int e, int n, int& i, int& j) {
  const float t = (float)(2 * n + 1);
  j = (int)((t - sqrtf(t * t - 8.0f * (float)e)) * 0.5f);
  if (j < 0) j = 0;
  if (j > n - 1) j = n - 1;
  while (j + 1 < n && pk_col(j + 1, n) <= e) ++j;
  while (j > 0 && pk_col(j, n) > e) --j;
  i = j + (e - pk_col(j, n));
}

struct ChildEntry { double v; int ri, rj; };
__device__ inline void add_children_front(const TreeArgs& t, const CliqueDesc& d, const double* updbase,
                                          double* F, int ldf, double* U, int ldu, double sp, double su,
                                          int dbg = 0) {
  const int nn = d.nn;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (t.gp_tptr && d.chend > d.chbeg) {
    const int kk = t.cl[t.chidx[d.chbeg]].parent;   // this clique's own id
    gather_children_plan(t, kk, updbase, [=](int32_t code, double v) {
      const int i = code & 0x7fff, j = (code >> 15) & 0x7fff;
      if (code & (1 << 30)) U[i + j * ldu] += su * v; else F[i + j * ldf] += sp * v;
    });
    __syncthreads();
    return;
  }
  for (int q = d.chbeg + wave; q < d.chend; q += nw) {
    const CliqueDesc c = t.cl[t.chidx[q]];
    const int nac = c.na;
    const int32_t* rel = t.relidx + c.rel;
    const double* Uc = updbase + c.updp;
    batched_loop<16>(lane, nac * (nac + 1) / 2, 64,
      [=](int e) {
        ChildEntry x;
        int i, j;
        pk_unpack(e, nac, i, j);
        x.ri = rel[i]; x.rj = rel[j]; x.v = (dbg & 16) ? 1.0 : Uc[e];
        return x;
      },
      [=](int e, const ChildEntry& x) {
        if (x.ri < 0) return;
        if (dbg & 8) { if (x.v == 123.456) F[0] = x.v; return; }
        if (x.rj < nn) unsafeAtomicAdd(&F[x.ri + x.rj * ldf], sp * x.v);
        else unsafeAtomicAdd(&U[(x.ri - nn) + (x.rj - nn) * ldu], su * x.v);
      });
  }
  __syncthreads();
}
// separator block of the parent's front (global) gathered into U (lower); optionally mirrored to
// global.  `p` (parent descriptor) and `rel` (relative indices, LDS or global) are loop invariants
// hoisted by the caller so that the per-right-hand-side loads are all independent.
__device__ inline void gather_front(const CliqueDesc& d, const CliqueDesc& p, const int32_t* rel,
                                    const double* xbase, const double* updbase, double* U, int ldu,
                                    double* mirror) {
  if (d.parent < 0 || d.na == 0) return;
  const int na = d.na, nnp = p.nn, nfp = p.nn + p.na, nap = p.na;
  const double* Pp = xbase + p.blk;
  const double* Up = updbase + p.upd;
  batched_loop<8>(threadIdx.x, na * na, blockDim.x,
    [=](int e) {
      int i = e % na, j = e / na;
      if (i < j) return 0.0;
      int ri = rel[i], rj = rel[j];
      return (rj < nnp) ? Pp[ri + (int64_t)rj * nfp] : Up[(ri - nnp) + (int64_t)(rj - nnp) * nap];
    },
    [=](int e, double v) {
      int i = e % na, j = e / na;
      if (i < j) return;
      U[i + j * ldu] = v;
      if (mirror) mirror[e] = v;
    });
}
// hoisted invariants for gather_front
template <bool LDS>
__device__ inline const int32_t* hoist_rel(const TreeArgs& t, const CliqueDesc& d, double* lds_tail) {
  const int32_t* g = t.relidx + d.rel;
  if (!LDS) return g;
  int32_t* r = reinterpret_cast<int32_t*>(lds_tail);
  for (int i = threadIdx.x; i < d.na; i += blockDim.x) r[i] = g[i];
  return r;
}

// 16 x 16 diagonal blocks: factorisation and inversion by ONE wavefront, matrix rows in registers (lane i
// holds row i), pivots and multipliers broadcast with v_readlane -- two workgroup barriers per call
// instead of three per column.
__device__ inline double readlane_f64(double v, int srclane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}
// 1/x from the hardware estimate plus two Newton steps (the pivots of a sequential elimination sit on the
// critical path: a full IEEE division there costs more than the rest of the column)
__device__ inline double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}
// sqrt(x) and 1/sqrt(x) from the hardware rsq estimate plus Newton steps
__device__ inline void fast_sqrt_rsqrt(double x, double& sq, double& rs) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x * r, r, 1.5);
  r = r * fma(-0.5 * x * r, r, 1.5);
  double g = x * r;
  g = fma(0.5 * r, fma(-g, g, x), g);
  sq = g;
  rs = fma(r, fma(-g, r, 1.0), r);
}
// rows of the inverse of the lower-triangular 16 x 16 matrix held as a[j] = L[lane][j]; lane c gets column c
__device__ inline void wave_tri_inv16(const double (&a)[16], double (&x)[16], int c) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) s -= readlane_f64(a[k], i) * x[k];
    x[i] = s * fast_rcp(readlane_f64(a[i], i));
  }
}
// in-place Cholesky of a w x w (w <= 16) block + its inverse (Dinv 16x16, ld 16, zeros elsewhere).
// Uniform control flow, every thread calls it.  Returns 0 or j+1.
__device__ inline int potrf_inv16_readlane(double* D, int ld, int w, double* Dinv) {
  __shared__ int fail_flag;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int i = threadIdx.x;
    double a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = (i < w && j <= i) ? D[i + j * ld] : ((i == j && i < 16) ? 1.0 : 0.0);
    int fail = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double djj = readlane_f64(a[j], j);
      if (!(djj > 0.0) && !fail) fail = j + 1;
      double sd, rsd;
      fast_sqrt_rsqrt(fail ? 1.0 : djj, sd, rsd);
      a[j] = (i == j) ? sd : a[j] * rsd;
#pragma unroll
      for (int c = j + 1; c < 16; ++c) a[c] -= a[j] * readlane_f64(a[j], c);
    }
    if (!fail) {
      double x[16];
      wave_tri_inv16(a, x, i);
      if (i < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (i < w && j <= i) D[i + j * ld] = a[j];
          Dinv[j + i * 16] = (i < w && j < w) ? x[j] : 0.0;   // lane i holds column i of the inverse
        }
      }
    }
    if (i == 0) fail_flag = fail;
  }
  __syncthreads();
  return fail_flag;
}
// The same on the matrix pipe (round 2: potrf_inv16_readlane above costs 6.8 us per call -- 256 dependent v_readlane pairs
// -- and sits on the critical path of every factorisation kernel; tools/micro/potrf16_bench.hip).  All 64 lanes hold the
// block in the accumulator layout of v_mfma_f64_16x16x4: lane (j = lane & 15, q = lane >> 4), register r <-> entry
// (row j, column q + 4 r).  In that layout a register r of a matrix W is at once the instruction's A operand
// W[:, 4r:4r+4] and its B operand W[:, 4r:4r+4]^T, and mfma(a from U, b from W) leaves W U^T in the same layout: no
// data moves between lanes.  Four block steps of four columns: the 4 x 4 diagonal block (ten v_readlane pairs) is
// factored and inverted redundantly by every lane in closed form, the rows below are scaled by one MFMA
// (P = M[:, blk] X^T), the trailing block updated by another (M -= P P^T), and the transpose T of the inverse grows
// by T[0:4b, blk] = -(T11 L21^T) X^T (b + 1 MFMAs).  Same interface and results as potrf_inv16_readlane.
// Round 5: the routine is issue-bound (a wave issues ~4.6 cycles per instruction here), so the count is what matters --
// 1415 instructions inlined as written in round 2, 1110 now: the pivot guards are ONE compare + scalar AND each (a failed
// pivot lets NaN / Inf run through arithmetic that feeds no address; the caller gets the verdict and stores nothing), the
// rows of X and L at a lane's column quad come from 0 / 1 lane masks by FMA instead of nested selects, and the routine is
// FORCED inline: left to the inliner it was an out-of-line function (36 call sites) reaching its LDS operands through flat
// pointers.  3.90 -> 2.48 us for a full block, 2.23 -> 1.42 us for w = 5 (the leaves of synth50k).  Returns 0 or 1.
// (the body: called by ONE full wave -- any wave of the workgroup -- with no barrier inside; returns the verdict, wave-uniform)
__device__ __forceinline__ bool wave_potrf_inv16(double* D, int ld, int w, double* Dinv) {
  {
    const int l = threadIdx.x & 63, j = l & 15, q = l >> 4;
    double m[4], t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = q + 4 * r;
      m[r] = (j < w && c <= j) ? D[j + c * ld] : (j == c ? 1.0 : 0.0);
      t[r] = 0.0;
    }
    bool ok = true;
    const int nb = (w + 3) >> 2;
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    const double e0 = q == 0 ? 1.0 : 0.0, e1 = q == 1 ? 1.0 : 0.0, e2 = q == 2 ? 1.0 : 0.0, e3 = q == 3 ? 1.0 : 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b < nb) {
        const int r0 = 4 * b;
        const double d00 = readlane_f64(m[b], r0), d10 = readlane_f64(m[b], r0 + 1), d20 = readlane_f64(m[b], r0 + 2),
                     d30 = readlane_f64(m[b], r0 + 3), d11 = readlane_f64(m[b], r0 + 17), d21 = readlane_f64(m[b], r0 + 18),
                     d31 = readlane_f64(m[b], r0 + 19), d22 = readlane_f64(m[b], r0 + 34), d32 = readlane_f64(m[b], r0 + 35),
                     d33 = readlane_f64(m[b], r0 + 51);
        double l00, l11, l22, l33, i0, i1, i2, i3;
        ok = ok && (d00 > 0.0);
        fast_sqrt_rsqrt(d00, l00, i0);
        const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
        const double p1 = fma(-l10, l10, d11);
        ok = ok && (p1 > 0.0);
        fast_sqrt_rsqrt(p1, l11, i1);
        const double l21 = fma(-l20, l10, d21) * i1, l31 = fma(-l30, l10, d31) * i1;
        const double p2 = fma(-l21, l21, fma(-l20, l20, d22));
        ok = ok && (p2 > 0.0);
        fast_sqrt_rsqrt(p2, l22, i2);
        const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * i2;
        const double p3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33)));
        ok = ok && (p3 > 0.0);
        fast_sqrt_rsqrt(p3, l33, i3);
        // X = inverse of the 4 x 4 factor
        const double x10 = -l10 * i0 * i1, x21 = -l21 * i1 * i2, x32 = -l32 * i2 * i3;
        const double x20 = -fma(l21, x10, l20 * i0) * i2, x31 = -fma(l32, x21, l31 * i1) * i3;
        const double x30 = -fma(l32, x20, fma(l31, x10, l30 * i0)) * i3;
        // lane (row 4b + u, q): X[u][q] (operand "X at the rows of the block") and the factor's L[u][q]: rows u = 0 .. 3 of both at
        // this lane's q from the 0 / 1 masks, then the row this lane's j asks for
        const int u = j - r0;
        const double xr0 = e0 * i0, xr1 = fma(e0, x10, e1 * i1), xr2 = fma(e0, x20, fma(e1, x21, e2 * i2)),
                     xr3 = fma(e0, x30, fma(e1, x31, fma(e2, x32, e3 * i3)));
        const double lr0 = e0 * l00, lr1 = fma(e0, l10, e1 * l11), lr2 = fma(e0, l20, fma(e1, l21, e2 * l22)),
                     lr3 = fma(e0, l30, fma(e1, l31, fma(e2, l32, e3 * l33)));
        const bool u0 = u == 0, u1 = u == 1, u2 = u == 2, u3 = u == 3;
        const double xa = u0 ? xr0 : (u1 ? xr1 : (u2 ? xr2 : (u3 ? xr3 : 0.0)));
        const double la = u0 ? lr0 : (u1 ? lr1 : (u2 ? lr2 : lr3));
        // rows below the block: P[j][q] = sum_k M[j][4b + k] X[q][k]  (register b of the result)
        const d4 pan = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, m[b], zero4, 0, 0, 0);
        const bool below = j >= r0 + 4;
        m[b] = below ? pan[b] : la;
        if (b + 1 < nb) {            // trailing block -= P P^T
          const double pb = below ? pan[b] : 0.0;
          const d4 upd = __builtin_amdgcn_mfma_f64_16x16x4f64(pb, pb, zero4, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) if (r > b) m[r] -= upd[r];
        }
        // transpose of the inverse: rows < 4b of the block's columns = -(T11 L21^T) X^T, the block itself X^T
        const bool inblk = u >= 0 && u < 4;
        d4 g = zero4;
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (s < b) g = __builtin_amdgcn_mfma_f64_16x16x4f64(inblk ? m[s] : 0.0, t[s], g, 0, 0, 0);
        const double wb = j < r0 ? g[b] : ((inblk && u == q) ? -1.0 : 0.0);
        const d4 tn = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, wb, zero4, 0, 0, 0);
        t[b] = j < r0 + 4 ? -tn[b] : 0.0;
      }
    }
    if (ok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = q + 4 * r;
        if (j < w && c <= j) D[j + c * ld] = m[r];
        Dinv[c + j * 16] = (c < w && j < w) ? t[r] : 0.0;      // t[r] = X(c, j)
      }
    }
    return ok;
  }
}
__device__ __forceinline__ int potrf_inv16(double* D, int ld, int w, double* Dinv) {
  __shared__ int fail_flag2;
  __syncthreads();
  if (threadIdx.x < 64) {
    const bool ok = wave_potrf_inv16(D, ld, w, Dinv);
    if (threadIdx.x == 0) fail_flag2 = ok ? 0 : 1;
  }
  __syncthreads();
  return fail_flag2;
}
// inverse only (block already a Cholesky factor / lower triangular)
__device__ __forceinline__ void tri_inv16(const double* D, int ld, int w, double* Dinv) {
  __syncthreads();
  if (threadIdx.x < 64) {
    const int i = threadIdx.x;
    double a[16], x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = (i < w && j <= i) ? D[i + j * ld] : ((i == j && i < 16) ? 1.0 : 0.0);
    wave_tri_inv16(a, x, i);
    if (i < 16) {
#pragma unroll
      for (int j = 0; j < 16; ++j) Dinv[j + i * 16] = (i < w && j < w) ? x[j] : 0.0;
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------ Hessian, leaves -> root
template <bool LDS>
__global__ void k_hess_up_mfma(MfmaArgs a, double* u, int64_t ldu) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  Work w = make_work<LDS>(a, d, smem, k, blockIdx.y);
  if (LDS) { load_consts(a, d, w, a.ymode != 0); }
  const int ymode = a.ymode;
  for (int r = blockIdx.y; r < a.nrhs; r += gridDim.y) {
    double* P = u + (int64_t)r * ldu + d.blk;
    const double* ub = a.t.updp + (int64_t)r * a.t.updplen;          // children: packed exchange buffer
    double* UkG = a.t.upd + (int64_t)r * a.t.updlen + d.upd;
    double* UkP = a.t.updp + (int64_t)r * a.t.updplen + d.updp;
    if (LDS) {
      {
        double* Fl = w.F; const int ldfl = w.ldf;
        batched_loop<8>(threadIdx.x, nf * nn, blockDim.x, [=](int e) { return P[e]; },
                        [=](int e, double v) { Fl[(e % nf) + (e / nf) * ldfl] = v; });
      }
      for (int e = threadIdx.x; e < na * na; e += blockDim.x) w.U[(e % na) + (e / na) * w.ldu] = 0.0;
    } else {
      w.F = P;
      w.U = UkG;
      for (int e = threadIdx.x; e < na * na; e += blockDim.x) UkG[e] = 0.0;
    }
    __syncthreads();
    if (!(a.skip & 1)) add_children_front(a.t, d, ub, w.F, w.ldf, w.U, w.ldu, 1.0, 1.0, a.skip);
    const Work v = w;
    if (!(a.skip & 2)) {
    // phase 1: E = F_AN - K Fnn / 2 ; T = Li Fnn
    wg_mma(na, nn, nn, [=](int m, int kk) { return v.K[m + kk * v.ldk]; },
           [=](int kk, int n) { return kk >= n ? v.F[kk + n * v.ldf] : v.F[n + kk * v.ldf]; },
           [=](int m, int n, double acc) { v.E[m + n * v.lde] = v.F[nn + m + n * v.ldf] - 0.5 * acc; });
    wg_mma(nn, nn, nn, [=](int m, int kk) { return v.Li[m + kk * v.ldl]; },
           [=](int kk, int n) { return kk >= n ? v.F[kk + n * v.ldf] : v.F[n + kk * v.ldf]; },
           [=](int m, int n, double acc) { v.T[m + n * v.ldt] = acc; }, false, (na + 15) >> 4);
    __syncthreads();
    // phase 2: U -= K E^T + E K^T (lower) ; G = (2E - F_AN) Li^T ; G_NN = T Li^T (lower, into the panel)
    wg_mma(na, na, 2 * nn,
           [=](int m, int kk) { return kk < nn ? v.K[m + kk * v.ldk] : v.E[m + (kk - nn) * v.lde]; },
           [=](int kk, int n) { return kk < nn ? v.E[n + kk * v.lde] : v.K[n + (kk - nn) * v.ldk]; },
           [=](int m, int n, double acc) { if (m >= n) v.U[m + n * v.ldu] -= acc; }, true);
    wg_mma(na, nn, nn, [=](int m, int kk) { return 2.0 * v.E[m + kk * v.lde] - v.F[nn + m + kk * v.ldf]; },
           [=](int kk, int n) { return v.Li[n + kk * v.ldl]; },
           [=](int m, int n, double acc) { v.G[m + n * v.ldg] = acc; }, false, 3);
    __syncthreads();  // all reads of F_AN by the G product are done before the panel is overwritten below
    wg_mma(nn, nn, nn, [=](int m, int kk) { return v.T[m + kk * v.ldt]; },
           [=](int kk, int n) { return v.Li[n + kk * v.ldl]; },
           [=](int m, int n, double acc) { if (m >= n) v.F[m + n * v.ldf] = acc; }, false, (na + 15) >> 4);
    // phase 3: Q = Ysc G (or G) into the AN rows of the panel
    if (ymode) {
      wg_mma(na, nn, na, [=](int m, int kk) { return yacc(v.Y, v.ldy, ymode, m, kk); },
             [=](int kk, int n) { return v.G[kk + n * v.ldg]; },
             [=](int m, int n, double acc) { v.F[nn + m + n * v.ldf] = acc; });
    } else {
      for (int e = threadIdx.x; e < na * nn; e += blockDim.x) {
        int i = e % na, j = e / na;
        v.F[nn + i + j * v.ldf] = v.G[i + j * v.ldg];
      }
    }
    }
    __syncthreads();
    if (LDS && !(a.skip & 4)) {
      for (int e = threadIdx.x; e < nf * nn; e += blockDim.x) {
        int i = e % nf, j = e / nf;
        if (i >= j) P[e] = v.F[i + j * v.ldf];
      }
      for (int e = threadIdx.x; e < na * na; e += blockDim.x) {
        int i = e % na, j = e / na;
        if (i >= j) UkP[pk_idx(i, j, na)] = v.U[i + j * v.ldu];
      }
      __syncthreads();
    } else {
      for (int e = threadIdx.x; e < na * na; e += blockDim.x) {
        int i = e % na, j = e / na;
        if (i >= j) UkP[pk_idx(i, j, na)] = UkG[e];
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------ Hessian, root -> leaves
template <bool LDS, int MASK = WK_ALL>
__global__ void k_hess_down_mfma(MfmaArgs a, double* u, int64_t ldu) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  Work w = make_work<LDS, MASK>(a, d, smem, k, blockIdx.y);
  if (LDS) { load_consts(a, d, w, a.ymode != 0); }
  const int ymode = a.ymode;
  const CliqueDesc par = a.t.cl[d.parent < 0 ? k : d.parent];
  const int32_t* rel = hoist_rel<LDS>(a.t, d, w.D16 + 256);
  __syncthreads();
  for (int r = blockIdx.y; r < a.nrhs; r += gridDim.y) {
    double* ur = u + (int64_t)r * ldu;
    double* P = ur + d.blk;
    double* ub = a.t.upd + (int64_t)r * a.t.updlen;
    double* UkG = ub + d.upd;
    if (LDS) {
      {
        double* Fl = w.F; const int ldfl = w.ldf;
        batched_loop<8>(threadIdx.x, nf * nn, blockDim.x, [=](int e) { return P[e]; },
                        [=](int e, double v) { Fl[(e % nf) + (e / nf) * ldfl] = v; });
      }
      // (the copy of Z_AA in global memory is what the CHILDREN gather from: a childless clique -- 7168 of the 8073 of
      // synth50k, 27 MB of stores per right-hand side -- keeps it in LDS only)
      gather_front(d, par, rel, ur, ub, w.U, w.ldu, d.chend > d.chbeg ? UkG : nullptr);
    } else {
      w.F = P;
      w.U = UkG;
      gather_front(d, par, rel, ur, ub, UkG, na, nullptr);
    }
    __syncthreads();
    const Work v = w;
    // phase 0: Q (into G): either R * Ghat_AN or a plain copy
    if (ymode) {
      wg_mma(na, nn, na, [=](int m, int kk) { return yacc(v.Y, v.ldy, ymode, m, kk); },
             [=](int kk, int n) { return v.F[nn + kk + n * v.ldf]; },
             [=](int m, int n, double acc) { v.G[m + n * v.ldg] = acc; });
    } else {
      for (int e = threadIdx.x; e < na * nn; e += blockDim.x) {
        int i = e % na, j = e / na;
        v.G[i + j * v.ldg] = v.F[nn + i + j * v.ldf];
      }
    }
    __syncthreads();
    // phase 1: QL = Q Li (into E) ; T = G_NN Li
    wg_mma(na, nn, nn, [=](int m, int kk) { return v.G[m + kk * v.ldg]; },
           [=](int kk, int n) { return v.Li[kk + n * v.ldl]; },
           [=](int m, int n, double acc) { v.E[m + n * v.lde] = acc; });
    wg_mma(nn, nn, nn, [=](int m, int kk) { return m >= kk ? v.F[m + kk * v.ldf] : v.F[kk + m * v.ldf]; },
           [=](int kk, int n) { return v.Li[kk + n * v.ldl]; },
           [=](int m, int n, double acc) { v.T[m + n * v.ldt] = acc; }, false, (na + 15) >> 4);
    __syncthreads();
    // phase 2: D = QL - Z_AA K / 2 (into G; Q is dead)
    wg_mma(na, nn, na, [=](int m, int kk) { return m >= kk ? v.U[m + kk * v.ldu] : v.U[kk + m * v.ldu]; },
           [=](int kk, int n) { return v.K[kk + n * v.ldk]; },
           [=](int m, int n, double acc) { v.G[m + n * v.ldg] = v.E[m + n * v.lde] - 0.5 * acc; });
    __syncthreads();
    // phase 3: Z_AN = 2D - QL ; Z_NN = Li^T T - K^T D - D^T K (lower)
    for (int e = threadIdx.x; e < na * nn; e += blockDim.x) {
      int i = e % na, j = e / na;
      v.F[nn + i + j * v.ldf] = 2.0 * v.G[i + j * v.ldg] - v.E[i + j * v.lde];
    }
    wg_mma(nn, nn, nn + 2 * na,
           [=](int m, int kk) {
             return kk < nn ? v.Li[kk + m * v.ldl]
                            : (kk < nn + na ? -v.K[(kk - nn) + m * v.ldk] : -v.G[(kk - nn - na) + m * v.ldg]);
           },
           [=](int kk, int n) {
             return kk < nn ? v.T[kk + n * v.ldt]
                            : (kk < nn + na ? v.G[(kk - nn) + n * v.ldg] : v.K[(kk - nn - na) + n * v.ldk]);
           },
           [=](int m, int n, double acc) { if (m >= n) v.F[m + n * v.ldf] = acc; });
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

// ------------------------------------------------------------------ Cholesky
// PREP (supernodes of at most 16 columns, LDS class): the inverse-form factor [Li; K] of the clique goes to lkout as well -- Li is
// the inverse of the single diagonal block, which potrf_inv16 has just left in D16, and K = L_AN Li is one small product on
// the front still in LDS: what k_prep_lk would read back from HBM in a launch of its own
template <bool LDS, bool PREP = false>
__global__ void k_chol_mfma(MfmaArgs a, double* x, double* lkout) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int k = a.t.lev[blockIdx.x];
  if (*info_of(a.t, k)) return;
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  Work w = make_work<LDS, WK_CHOL>(a, d, smem, k, 0);
  double* P = x + d.blk;
  double* UkG = a.t.upd + d.upd;
  double* UkP = a.t.updp + d.updp;
  if (LDS) {
    {
      double* Fl = w.F; const int ldfl = w.ldf;
      batched_loop<8>(threadIdx.x, nf * nn, blockDim.x, [=](int e) { return P[e]; },
                      [=](int e, double v) { Fl[(e % nf) + (e / nf) * ldfl] = v; });
    }
    for (int e = threadIdx.x; e < na * na; e += blockDim.x) w.U[(e % na) + (e / na) * w.ldu] = 0.0;
  } else {
    w.F = P;
    w.U = UkG;
    for (int e = threadIdx.x; e < na * na; e += blockDim.x) UkG[e] = 0.0;
  }
  __syncthreads();
  add_children_front(a.t, d, a.t.updp, w.F, w.ldf, w.U, w.ldu, 1.0, 1.0);
  const Work v = w;
  for (int jb = 0; jb < nn; jb += 16) {
    const int bw = min(16, nn - jb);
    int f = potrf_inv16(v.F + jb + jb * v.ldf, v.ldf, bw, v.D16);
    if (f) { if (threadIdx.x == 0) atomicCAS(info_of(a.t, k), 0, info_val(a.t, k)); return; }
    const int mrem = nf - jb - bw, ncr = nn - jb - bw;
    const double* Pj = v.F + (jb + bw) + jb * v.ldf;  // rows below the diagonal block, block column jb
    double* Pw = v.F + (jb + bw) + jb * v.ldf;
    // rows below <- rows below * Dinv^T  (single column tile: in place is safe)
    wg_mma(mrem, bw, bw, [=](int m, int kk) { return Pj[m + kk * v.ldf]; },
           [=](int kk, int n) { return v.D16[n + kk * 16]; },
           [=](int m, int n, double acc) { Pw[m + n * v.ldf] = acc; });
    __syncthreads();
    if (ncr > 0) {
      double* Tr = v.F + (jb + bw) + (jb + bw) * v.ldf;
      wg_mma(mrem, ncr, bw, [=](int m, int kk) { return Pj[m + kk * v.ldf]; },
             [=](int kk, int n) { return Pj[n + kk * v.ldf]; },
             [=](int m, int n, double acc) { if (m >= n) Tr[m + n * v.ldf] -= acc; }, true);
      __syncthreads();
    }
  }
  if (na) {
    const double* La = v.F + nn;
    wg_mma(na, na, nn, [=](int m, int kk) { return La[m + kk * v.ldf]; },
           [=](int kk, int n) { return La[n + kk * v.ldf]; },
           [=](int m, int n, double acc) { if (m >= n) v.U[m + n * v.ldu] -= acc; }, true);
  }
  __syncthreads();
  if (LDS && PREP) {
    double* Li = lkout + d.blk;
    const double* Dv = v.D16;
    for (int e = threadIdx.x; e < nn * nn; e += blockDim.x) {
      int i = e % nn, j = e / nn;
      Li[i + (int64_t)j * nf] = (i >= j) ? Dv[i + j * 16] : 0.0;
    }
    if (na) {
      const double* La = v.F + nn;
      double* Kk = Li + nn;
      wg_mma(na, nn, nn, [=](int m, int kk) { return La[m + kk * v.ldf]; },
             [=](int kk, int n) { return kk >= n ? Dv[kk + n * 16] : 0.0; },
             [=](int m, int n, double acc) { Kk[m + (int64_t)n * nf] = acc; });
    }
  }
  if (LDS) {
    for (int e = threadIdx.x; e < nf * nn; e += blockDim.x) {
      int i = e % nf, j = e / nf;
      if (i >= j) P[e] = v.F[i + j * v.ldf];
    }
    for (int e = threadIdx.x; e < na * na; e += blockDim.x) {
      int i = e % na, j = e / na;
      if (i >= j) UkP[pk_idx(i, j, na)] = v.U[i + j * v.ldu];
    }
  } else {
    for (int e = threadIdx.x; e < na * na; e += blockDim.x) {
      int i = e % na, j = e / na;
      if (i >= j) UkP[pk_idx(i, j, na)] = UkG[e];
    }
  }
}

// ------------------------------------------------------------------ inverse-form factor LK = [L_NN^-1 ; L_AN L_NN^-1]
// One workgroup per clique, operands in HBM/L2 (runs once per factorisation).
__global__ void k_prep_lk(TreeArgs t, const double* L, double* LK) {
  const int k = t.lev ? t.lev[blockIdx.x] : blockIdx.x;
  const CliqueDesc d = t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const double* Lk = L + d.blk;
  double* Li = LK + d.blk;
  double* Kk = Li + nn;
  __shared__ double Dinv[256];
  // 16 x (up to 256) row-block scratch, needed by supernodes of more than 16 columns only: dynamic, sized by the launch
  // (prep_lk_lds_bytes) -- as a static 32 KB array it capped the launch at four workgroups per CU on the 8064 small
  // cliques of synth50k, whose supernodes have 5 and 15 columns
  extern __shared__ __attribute__((aligned(16))) double S[];
  // zero the upper triangle of Li
  for (int e = threadIdx.x; e < nn * nn; e += blockDim.x) {
    int i = e % nn, j = e / nn;
    if (i < j) Li[i + (int64_t)j * nf] = 0.0;
  }
  for (int ib = 0; ib < nn; ib += 16) {
    const int bw = min(16, nn - ib);
    tri_inv16(Lk + ib + (int64_t)ib * nf, nf, bw, Dinv);
    // diagonal block of Li
    for (int e = threadIdx.x; e < bw * bw; e += blockDim.x) {
      int i = e % bw, j = e / bw;
      if (i >= j) Li[(ib + i) + (int64_t)(ib + j) * nf] = Dinv[i + j * 16];
    }
    // off-diagonal blocks: Li[ib, 0:ib] = -Dinv * (L[ib, 0:ib] * Li[0:ib, 0:ib]), in column chunks of 256
    for (int c0 = 0; c0 < ib; c0 += 256) {
      const int cw = min(256, ib - c0);
      __syncthreads();
      wg_mma(bw, cw, ib - c0, [=](int m, int kk) { return Lk[(ib + m) + (int64_t)(c0 + kk) * nf]; },
             [=](int kk, int n) { return (c0 + kk >= c0 + n) ? Li[(c0 + kk) + (int64_t)(c0 + n) * nf] : 0.0; },
             [=](int m, int n, double acc) { S[m + n * 16] = acc; });
      __syncthreads();
      wg_mma(bw, cw, bw, [=](int m, int kk) { return Dinv[m + kk * 16]; },
             [=](int kk, int n) { return S[kk + n * 16]; },
             [=](int m, int n, double acc) { Li[(ib + m) + (int64_t)(c0 + n) * nf] = -acc; });
    }
    __syncthreads();
  }
  // K = L_AN Li
  if (na) {
    wg_mma(na, nn, nn, [=](int m, int kk) { return Lk[(nn + m) + (int64_t)kk * nf]; },
           [=](int kk, int n) { return kk >= n ? Li[kk + (int64_t)n * nf] : 0.0; },
           [=](int m, int n, double acc) { Kk[m + (int64_t)n * nf] = acc; });
  }
}

// ------------------------------------------------------------------ projected inverse (needs LK of the factor)
template <bool LDS>
__global__ void k_pinv_mfma(MfmaArgs a, double* x) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  Work w = make_work<LDS, WK_PINV>(a, d, smem, k, 0);
  double* P = x + d.blk;
  double* UkG = a.t.upd + d.upd;
  const CliqueDesc par = a.t.cl[d.parent < 0 ? k : d.parent];
  const int32_t* rel = a.t.relidx + d.rel;
  if (LDS) {
    load_consts(a, d, w, false);
    gather_front(d, par, rel, x, a.t.upd, w.U, w.ldu, UkG);
  } else {
    w.U = UkG;
    gather_front(d, par, rel, x, a.t.upd, UkG, na, nullptr);
  }
  __syncthreads();
  const Work v = w;
  // E = Y_AA K
  wg_mma(na, nn, na, [=](int m, int kk) { return m >= kk ? v.U[m + kk * v.ldu] : v.U[kk + m * v.ldu]; },
         [=](int kk, int n) { return v.K[kk + n * v.ldk]; },
         [=](int m, int n, double acc) { v.E[m + n * v.lde] = acc; });
  __syncthreads();
  // Y_NN = Li^T Li + K^T E (lower) ; Y_AN = -E   (written straight to the global panel)
  wg_mma(nn, nn, nn + na,
         [=](int m, int kk) { return kk < nn ? v.Li[kk + m * v.ldl] : v.K[(kk - nn) + m * v.ldk]; },
         [=](int kk, int n) { return kk < nn ? v.Li[kk + n * v.ldl] : v.E[(kk - nn) + n * v.lde]; },
         [=](int m, int n, double acc) { if (m >= n) P[m + (int64_t)n * nf] = acc; });
  for (int e = threadIdx.x; e < na * nn; e += blockDim.x) {
    int i = e % na, j = e / na;
    P[nn + i + (int64_t)j * nf] = -v.E[i + j * v.lde];
  }
}




// fac[k] <- chol(yaa[k]) for cliques whose separator block fits LDS: blocked (16-wide) in LDS on MFMA
__global__ void __launch_bounds__(256) k_factor_yaa_lds(MfmaArgs a, const double* yaa, double* fac) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int na = d.na;
  if (!na) return;
  const int ld = padld(a.namax);
  double* const M = smem;
  double* const D16 = smem + (int64_t)ld * a.namax;
  const double* src = yaa + d.upd;
  double* dst = fac + d.upd;
  batched_loop<8>(threadIdx.x, na * na, blockDim.x, [=](int e) { return (e % na) >= (e / na) ? src[e] : 0.0; },
                  [=](int e, double v) { M[(e % na) + (e / na) * ld] = v; });
  for (int jb = 0; jb < na; jb += 16) {
    const int bw = min(16, na - jb);
    int f = potrf_inv16(M + jb + jb * ld, ld, bw, D16);
    if (f) { if (threadIdx.x == 0) atomicCAS(info_of(a.t, k), 0, info_val(a.t, k)); return; }
    const int mrem = na - jb - bw;
    if (mrem > 0) {
      double* Pj = M + (jb + bw) + jb * ld;
      wg_mma(mrem, bw, bw, [=](int m, int kk) { return Pj[m + kk * ld]; },
             [=](int kk, int n) { return D16[n + kk * 16]; },
             [=](int m, int n, double acc) { Pj[m + n * ld] = acc; });
      __syncthreads();
      double* Tr = M + (jb + bw) + (jb + bw) * ld;
      wg_mma(mrem, mrem, bw, [=](int m, int kk) { return Pj[m + kk * ld]; },
             [=](int kk, int n) { return Pj[n + kk * ld]; },
             [=](int m, int n, double acc) { if (m >= n) Tr[m + n * ld] -= acc; }, true);
      __syncthreads();
    }
  }
  for (int e = threadIdx.x; e < na * na; e += blockDim.x) {
    int i = e % na, j = e / na;
    if (i >= j) dst[e] = M[i + j * ld];
  }
}

// ------------------------------------------------------------------ padded-LDS fast kernels
// Every LDS matrix is padded to multiples of 16 rows / columns with zero fill, so the MFMA loops
// need no bounds checks or branches: each k-step is two ds_read_b64, two pointer bumps, one MFMA.
struct PadL {
  int NN, NA, ldn, lda;
  int oK, oBD, oY, oFnn, oFan, oE, oG, oQ, oT, oU, oInt, total;
  int iCh, iPan, iOut, iTgt;   // int-table offsets (in ints, relative to the int region)
};
// nchmax: max #children, panmax: max nf*nn, pkmax: max na(na+1)/2, plansum: max sum of the children's packed sizes
__host__ __device__ inline PadL pad_layout(int nnmax, int namax, int nchmax = 0, int panmax = 0, int pkmax = 0,
                                           int plansum = 0) {
  PadL L;
  L.NN = (nnmax + 15) & ~15;
  L.NA = (namax + 15) & ~15;
  L.ldn = L.NN + 1;
  L.lda = L.NA + 1;
  int o = 0;
  L.oK = o; o += L.lda * L.NN;
  L.oBD = o; o += L.ldn * L.NN;   // kron(I_rb, Li^T): applies Li^T to every stacked right-hand side at once
  L.oY = o; o += L.lda * L.NA;
  L.oFnn = o; o += L.ldn * L.NN;
  L.oFan = o; o += L.lda * L.NN;
  L.oE = o; o += L.lda * L.NN;
  // one column tile (NN = 16): G = X BD is computed tile-in-place over X, and Q goes to the (dead) E buffer
  if (L.NN == 16) { L.oG = L.oFan; L.oQ = L.oE; } else { L.oG = o; o += L.lda * L.NN; L.oQ = L.oFan; }
  L.oT = o; o += L.ldn * L.NN;
  L.oU = o; if (nchmax > 0) o += L.lda * L.NA;   // childless classes store their update matrices straight from the accumulators
  L.oInt = o;
  // RHS-invariant index tables, built once per workgroup, so the per-rhs loop has no index arithmetic:
  //   child metadata (4 ints per child), panel position -> LDS offset, packed own-update position ->
  //   LDS offset, packed child-update position -> LDS offset of its target in the front
  L.iCh = 0;
  L.iPan = 4 * nchmax;
  L.iOut = L.iPan + panmax;
  L.iTgt = L.iOut + pkmax;
  o += (L.iTgt + plansum + 3) / 2;
  L.total = o + 2;
  return L;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global
// load and store (s_waitcnt vmcnt(0)), which would serialise the register prefetch of the next right-hand
// side and the write-out of the previous one with the compute phases; registers filled by global loads are
// still guarded by the compiler's own vmcnt bookkeeping at their first use.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// acc += sum over ks k-steps; pa / pb: this lane's operand addresses for k-step 0, sa / sb: bump per k-step
__device__ inline void mma_run(d4& acc, const double* pa, int sa, const double* pb, int sb, int ks) {
#pragma unroll 4
  for (int s = 0; s < ks; ++s) {
    const double av = *pa, bv = *pb;
    pa += sa;
    pb += sb;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc, 0, 0, 0);
  }
}

__global__ void __launch_bounds__(512) k_hess_up_pad(MfmaArgs a, double* u, int64_t ldu) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const PadL L = pad_layout(a.nnmax, a.namax, a.nchmax, a.panmax, a.pkmax, a.plansum);
  int* const sInt = reinterpret_cast<int*>(smem + L.oInt);
  int* const sCh = sInt + L.iCh;
  int* const sPan = sInt + L.iPan;
  int* const sOut = sInt + L.iOut;
  int* const sTgt = sInt + L.iTgt;
  double* const sK = smem + L.oK;
  double* const sBD = smem + L.oBD;
  double* const sY = smem + L.oY;
  double* const sFnn = smem + L.oFnn;
  double* const sFan = smem + L.oFan;
  double* const sE = smem + L.oE;
  double* const sG = smem + L.oG;
  double* const sQ = smem + L.oQ;
  double* const sT = smem + L.oT;
  double* const sU = smem + L.oU;
  const int ldn = L.ldn, lda = L.lda;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int ymode = a.ymode;
  const int npan = nf * nn, npk = na * (na + 1) / 2;
  const int nch = d.chend - d.chbeg;
  // Childless cliques with a narrow supernode stack several right-hand sides side by side in the
  // 16-column tiles (column = q*nn + n): one tile pass serves rb right-hand sides, and their update
  // matrices go from the accumulators straight to HBM.
  const int rb = (nch == 0) ? max(1, min(4, L.NN / max(nn, 1))) : 1;
  const int ncs = rb * nn;
  // zero everything once (pads must stay zero)
  for (int e = tid; e < L.oInt; e += nthr) smem[e] = 0.0;
  // ---- RHS-invariant index tables
  for (int q = tid; q < nch; q += nthr) {
    const CliqueDesc c = a.t.cl[a.t.chidx[d.chbeg + q]];
    sCh[4 * q] = (int)(c.updp & 0xffffffff);
    sCh[4 * q + 1] = (int)(c.updp >> 32);
    sCh[4 * q + 2] = c.na;
  }
  for (int e = tid; e < npan; e += nthr) {       // panel entry -> LDS offset; bit 30: F_NN block; -1: unused
    const int i = e % nf, j = e / nf;
    sPan[e] = (i >= nn) ? (i - nn) + j * lda : (i >= j ? ((1 << 30) | (i + j * ldn)) : -1);
  }
  for (int e = tid; e < npk; e += nthr) {        // packed own update entry -> LDS offset
    int i, j;
    pk_unpack(e, na, i, j);
    sOut[e] = L.oU + i + j * lda;
  }
  __syncthreads();
  if (tid == 0) {
    int off = 0;
    for (int q = 0; q < nch; ++q) { sCh[4 * q + 3] = off; off += sCh[4 * q + 2] * (sCh[4 * q + 2] + 1) / 2; }
  }
  __syncthreads();
  for (int q = wave; q < nch; q += nw) {         // packed child entry -> LDS offset of its target
    const int32_t* rel = a.t.relidx + a.t.cl[a.t.chidx[d.chbeg + q]].rel;
    const int nac = sCh[4 * q + 2], tb = sCh[4 * q + 3];
    for (int e = lane; e < nac * (nac + 1) / 2; e += 64) {
      int i, j;
      pk_unpack(e, nac, i, j);
      const int ri = rel[i], rj = rel[j];
      sTgt[tb + e] = (rj >= nn) ? L.oU + (ri - nn) + (rj - nn) * lda
                                : (ri >= nn ? L.oFan + (ri - nn) + rj * lda : L.oFnn + ri + rj * ldn);
    }
  }
  {
    const double* src = a.LK + d.blk;
    batched_loop<8>(tid, npan, nthr, [=](int e) { return src[e]; },
                    [=](int e, double v) {
                      int i = e % nf, j = e / nf;
                      if (i < nn) {
                        if (i >= j)                               // Li^T on the diagonal blocks of BD
                          for (int q = 0; q < rb; ++q) sBD[(q * nn + j) + (q * nn + i) * ldn] = v;
                      } else sK[(i - nn) + j * lda] = v;
                    });
    if (ymode) {
      const double* ys = a.ysc + d.upd;
      batched_loop<8>(tid, na * na, nthr, [=](int e) { return (e % na) >= (e / na) ? ys[e] : 0.0; },
                      [=](int e, double v) {
                        int i = e % na, j = e / na;
                        if (i < j) return;
                        if (ymode == 1) { sY[i + j * lda] = v; sY[j + i * lda] = v; }
                        else if (ymode == 2) sY[j + i * lda] = v;     // R^T
                        else sY[i + j * lda] = v;                       // R
                      });
    }
  }
  const int NAt = L.NA >> 4, NNt = L.NN >> 4;
  const int NCt = (ncs + 15) >> 4;                  // column tiles actually used
  const int ksn = (nn + 3) >> 2, ksa = (na + 3) >> 2, ksc = (ncs + 3) >> 2;
#ifdef SMCP_STAMPS   // diagnostic build only (SMCP_STAMPS=1 python -m smcp_amd.build --force): cycle stamps of thread 0
  const bool stamp = tid == 0 && a.dbg;
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamp ? clock64() : 0;
#define STAMP(i) do { if (stamp) { unsigned long long tn_ = clock64(); tph[i] += tn_ - tlast; tlast = tn_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
  // Software pipeline (cliques with children, one rhs per pass): the panel and the children's values of
  // the NEXT right-hand side are fetched into registers while the current one is being computed.
  constexpr int PP = 4, PC = 8;
  const bool pipe = nch > 0 && nch <= nw && npan <= PP * nthr && a.plansum > 0;
  bool pipe_ok = pipe;
  if (pipe) {
    bool fits = true;
    for (int q = 0; q < nch; ++q) fits = fits && (sCh[4 * q + 2] * (sCh[4 * q + 2] + 1) / 2 <= PC * 64);
    pipe_ok = fits;
  }
  double pre_p[PP], pre_c[PC];
  auto prefetch = [&](int rr) {
    const double* P = u + (int64_t)rr * ldu + d.blk;
#pragma unroll
    for (int x = 0; x < PP; ++x) { const int e = tid + x * nthr; pre_p[x] = e < npan ? P[e] : 0.0; }
    if (wave < nch) {
      const int nac = sCh[4 * wave + 2], np_ = nac * (nac + 1) / 2;
      const double* Uc = a.t.updp + (int64_t)rr * a.t.updplen + (((int64_t)sCh[4 * wave + 1] << 32) | (uint32_t)sCh[4 * wave]);
#pragma unroll
      for (int x = 0; x < PC; ++x) { const int e = lane + 64 * x; pre_c[x] = e < np_ ? Uc[e] : 0.0; }
    }
  };
  // Childless cliques (stacked right-hand sides): same idea, one panel entry per thread and stacked rhs
  const bool lpipe = nch == 0 && npan <= nthr;
  double pre_l[4] = {0.0, 0.0, 0.0, 0.0}, pre_l2[4] = {0.0, 0.0, 0.0, 0.0};   // two passes ahead
  auto prefetch_leaf = [&](int r0n) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = r0n + q * (int)gridDim.y;
      pre_l[q] = pre_l2[q];
      pre_l2[q] = (q < rb && rr < a.nrhs && tid < npan) ? u[(int64_t)rr * ldu + d.blk + tid] : 0.0;
    }
  };
  if (pipe_ok && (int)blockIdx.y < a.nrhs) prefetch(blockIdx.y);
  if (lpipe) { prefetch_leaf(blockIdx.y); prefetch_leaf(blockIdx.y + (int)gridDim.y * rb); }
  for (int r0 = blockIdx.y; r0 < a.nrhs; r0 += gridDim.y * rb) {
    // this pass: right-hand sides r0, r0 + gridDim.y, ... (rbc of them)
    const int rbc = min(rb, (a.nrhs - r0 + (int)gridDim.y - 1) / (int)gridDim.y);
    lds_barrier();
    STAMP(0);
    // ---- assemble the front(s): panel + children (lower triangles), then mirror F_NN
    if (pipe_ok) {
#pragma unroll
      for (int x = 0; x < PP; ++x) {
        const int e = tid + x * nthr;
        if (e < npan) { const int o = sPan[e]; if (o >= 0) smem[(o & (1 << 30)) ? L.oFnn + (o & 0x3fffffff) : L.oFan + o] = pre_p[x]; }
      }
    } else if (lpipe) {
      const int o = tid < npan ? sPan[tid] : -1;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < rbc && o >= 0)
          smem[(o & (1 << 30)) ? L.oFnn + q * nn * ldn + (o & 0x3fffffff) : L.oFan + q * nn * lda + o] = pre_l[q];
      prefetch_leaf(r0 + 2 * (int)gridDim.y * rb);  // in flight during this pass and the next
    } else
    for (int q = 0; q < rbc; ++q) {
      const double* P = u + (int64_t)(r0 + q * gridDim.y) * ldu + d.blk;
      const int oan = L.oFan + q * nn * lda, onn = L.oFnn + q * nn * ldn;
      batched_loop<8>(tid, npan, nthr, [=](int e) { return P[e]; },
                      [=](int e, double v) {
                        const int o = sPan[e];
                        if (o >= 0) smem[(o & (1 << 30)) ? onn + (o & 0x3fffffff) : oan + o] = v;
                      });
    }
    if (nch) for (int e = tid; e < npk; e += nthr) smem[sOut[e]] = 0.0;
    lds_barrier();
    STAMP(1);
    if (pipe_ok) {
      if (wave < nch) {
        const int nac = sCh[4 * wave + 2], np_ = nac * (nac + 1) / 2;
        const int* tg = sTgt + sCh[4 * wave + 3];
#pragma unroll
        for (int x = 0; x < PC; ++x) { const int e = lane + 64 * x; if (e < np_ && !(a.skip & 128)) unsafeAtomicAdd(&smem[tg[e]], pre_c[x]); }
      }
      lds_barrier();
      const int rn = r0 + (int)gridDim.y;
      if (rn < a.nrhs) prefetch(rn);          // in flight during the three compute phases below
    } else if (nch) {
      const double* ub = a.t.updp + (int64_t)r0 * a.t.updplen;      // children: packed exchange buffer
      for (int q = wave; q < nch; q += nw) {
        const int nac = sCh[4 * q + 2];
        const int* tg = sTgt + sCh[4 * q + 3];
        const double* Uc = ub + (((int64_t)sCh[4 * q + 1] << 32) | (uint32_t)sCh[4 * q]);
        batched_loop<8>(lane, nac * (nac + 1) / 2, 64, [=](int e) { return Uc[e]; },
                        [=](int e, double vv) { unsafeAtomicAdd(&smem[tg[e]], vv); });
      }
      lds_barrier();
    }
    for (int q = 0; q < rbc; ++q) {
      double* Fq = sFnn + q * nn * ldn;
      for (int j = wave; j < nn; j += nw)
        for (int i = j + 1 + lane; i < nn; i += 64) Fq[j + i * ldn] = Fq[i + j * ldn];
    }
    lds_barrier();
    STAMP(2);
    // ---- phase 1: E = F_AN - K F_NN / 2, X = F_AN - K F_NN (in place of F_AN) ; T = Li F_NN   (all stacked columns)
    {
      const int nE = NAt * NCt, nT = NNt * NCt;
      for (int t = wave; t < nE + nT; t += nw) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        if (t < nE) {
          const int tm = t % NAt, tn = t / NAt;
          mma_run(acc, sK + tm * 16 + l15 + kq * lda, 4 * lda, sFnn + kq + (tn * 16 + l15) * ldn, 4, ksn);
          const int m = tm * 16 + l15;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int n = tn * 16 + kq + 4 * rr;
            const double f = sFan[m + n * lda];
            sE[m + n * lda] = f - 0.5 * acc[rr];
            sFan[m + n * lda] = f - acc[rr];
          }
        } else {
          const int tt = t - nE, tm = tt % NNt, tn = tt / NNt;
          // Li[m][k] = BD[k][m] (first diagonal block of BD)
          mma_run(acc, sBD + kq + (tm * 16 + l15) * ldn, 4, sFnn + kq + (tn * 16 + l15) * ldn, 4, ksn);
          const int m = tm * 16 + l15;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) sT[m + (tn * 16 + kq + 4 * rr) * ldn] = acc[rr];
        }
      }
    }
    lds_barrier();
    STAMP(3);
    // ---- phase 2: U_q -= K E_q^T + E_q K^T (lower tiles, per stacked rhs) ; G = X BD ; G_NN = T BD (into sFnn)
    {
      const int nUq = NAt * (NAt + 1) / 2, nU = nUq * rbc, nG = NAt * NCt, nN = NNt * NCt;
      for (int t = wave; t < nU + nG + nN; t += nw) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        if (t < nU) {
          const int q = t / nUq;
          int tm = 0, rem = t - q * nUq;
          while (rem > tm) { rem -= tm + 1; ++tm; }
          const int tn = rem;
          const double* Eq = sE + q * nn * lda;
          mma_run(acc, sK + tm * 16 + l15 + kq * lda, 4 * lda, Eq + tn * 16 + l15 + kq * lda, 4 * lda, ksn);
          mma_run(acc, Eq + tm * 16 + l15 + kq * lda, 4 * lda, sK + tn * 16 + l15 + kq * lda, 4 * lda, ksn);
          const int m = tm * 16 + l15;
          if (nch) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int n = tn * 16 + kq + 4 * rr;
              if (m >= n) sU[m + n * lda] -= acc[rr];
            }
          } else {   // no children: the update matrix is exactly -acc, stored packed
            double* UkP = a.t.updp + (int64_t)(r0 + q * gridDim.y) * a.t.updplen + d.updp;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int n = tn * 16 + kq + 4 * rr;
              if (m >= n && m < na) UkP[n * na - (n * (n - 1)) / 2 + (m - n)] = -acc[rr];
            }
          }
        } else if (t < nU + nG) {
          const int tt = t - nU, tm = tt % NAt, tn = tt / NAt;
          mma_run(acc, sFan + tm * 16 + l15 + kq * lda, 4 * lda, sBD + kq + (tn * 16 + l15) * ldn, 4, ksc);
          const int m = tm * 16 + l15;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) sG[m + (tn * 16 + kq + 4 * rr) * lda] = acc[rr];
        } else {
          const int tt = t - nU - nG, tm = tt % NNt, tn = tt / NNt;
          mma_run(acc, sT + tm * 16 + l15 + kq * ldn, 4 * ldn, sBD + kq + (tn * 16 + l15) * ldn, 4, ksc);
          const int m = tm * 16 + l15;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) sFnn[m + (tn * 16 + kq + 4 * rr) * ldn] = acc[rr];
        }
      }
    }
    lds_barrier();
    STAMP(4);
    // ---- phase 3: Q = Ysc G into the F_AN buffer (X is dead), or plain G
    {
      const int nQ = NAt * NCt;
      for (int t = wave; t < nQ; t += nw) {
        const int tm = t % NAt, tn = t / NAt;
        const int m = tm * 16 + l15;
        if (ymode) {
          d4 acc = {0.0, 0.0, 0.0, 0.0};
          mma_run(acc, sY + m + kq * lda, 4 * lda, sG + kq + (tn * 16 + l15) * lda, 4, ksa);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) sQ[m + (tn * 16 + kq + 4 * rr) * lda] = acc[rr];
        } else {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int n = tn * 16 + kq + 4 * rr;
            sQ[m + n * lda] = sG[m + n * lda];
          }
        }
      }
    }
    lds_barrier();
    STAMP(5);
    // ---- write out: panel(s) (lower of NN + AN) and, with children, the update matrix (lower, packed)
    for (int q = 0; q < rbc; ++q) {
      double* P = u + (int64_t)(r0 + q * gridDim.y) * ldu + d.blk;
      const int oan = L.oQ + q * nn * lda, onn = L.oFnn + q * nn * ldn;
      for (int e = tid; e < npan; e += nthr) {
        const int o = sPan[e];
        if (o >= 0) P[e] = smem[(o & (1 << 30)) ? onn + (o & 0x3fffffff) : oan + o];
      }
    }
    if (nch) {
      double* UkP = a.t.updp + (int64_t)r0 * a.t.updplen + d.updp;
      for (int e = tid; e < npk; e += nthr) UkP[e] = smem[sOut[e]];
    }
    STAMP(6);
  }
#ifdef SMCP_STAMPS
  if (stamp)
    for (int i = 0; i < 7; ++i) atomicAdd(a.dbg + i + 8 * (d.na > 40), tph[i]);
#endif
#undef STAMP
}

// ------------------------------------------------------------------ Gram matrix  H = G^T W G
// G: the m swept constraint matrices (column r at G + r*ldg, length len), W = diag(w) with the
// cspmatrix inner-product weights (1 diagonal, 2 off-diagonal lower, 0 unused upper entries),
// passed as sw = sqrt(w) so that both MFMA operands are the same LDS image sqrt(w) * G.
// Grid: (chunks of the long dimension, lower 128x128 blocks of H).  Each workgroup streams its
// chunk through LDS in slices of 64 entries (coalesced along the contiguous dimension) and keeps
// its 16x16 output tiles in registers; partial sums go to `partial` and are reduced in a fixed
// order by k_gram_reduce (deterministic).
constexpr int GRAM_BLK = 128;   // columns of H per block
constexpr int GRAM_KS = 64;     // slice of the long dimension staged per step
constexpr int GRAM_LD = GRAM_BLK + 1;
constexpr int GRAM_LDK = GRAM_KS + 2;   // k_gram_diag128: [column][k] image
constexpr int GRAM_RZ = 8;              // workgroups per tile of the first stage of a two-stage k_gram_reduce

template <bool DIAG>
__global__ void __launch_bounds__(256) k_gram_partial(const double* G, int64_t ldg, int m, int64_t e_lo, int64_t e_hi,
                                                      const double* sw, int64_t chunk, double* partial, int coff,
                                                      int nchunk_total, int bi, int bj) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const sA = smem;
  double* const sB = DIAG ? smem : smem + GRAM_KS * GRAM_LD;
  // block (bi, bj), bi >= bj (DIAG: bi == bj), one launch per block; its index in the partial buffer is that of the
  // lower-triangular enumeration
  const int blk = bi * (bi + 1) / 2 + bj;
  const int ci0 = bi * GRAM_BLK, cj0 = bj * GRAM_BLK;
  const int ni = min(GRAM_BLK, m - ci0), nj = min(GRAM_BLK, m - cj0);
  const int mti = (ni + 15) >> 4, mtj = (nj + 15) >> 4;
  const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // tiles of this block owned by this wave, decoded once (compile-time accumulator indices keep them in registers)
  constexpr int MAXT = DIAG ? 9 : 16;
  int tms[MAXT], tns[MAXT];
  const int ntile = DIAG ? mti * (mti + 1) / 2 : mti * mtj;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    int p = wave + 4 * i;
    if (p >= ntile) { tms[i] = -1; tns[i] = 0; continue; }
    if (DIAG) { int tm = 0; while (p > tm) { p -= tm + 1; ++tm; } tms[i] = tm; tns[i] = p; }
    else { tms[i] = p % mti; tns[i] = p / mti; }
  }
  d4 acc[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  for (int e = threadIdx.x; e < (DIAG ? 1 : 2) * GRAM_KS * GRAM_LD; e += 256) smem[e] = 0.0;   // columns beyond the block stay zero
  const int64_t e_begin = e_lo + (int64_t)blockIdx.x * chunk, e_end = min(e_hi, e_begin + chunk);
  const double* const la = sA + kq * GRAM_LD + l15;
  const double* const lb = sB + kq * GRAM_LD + l15;
  for (int64_t e0 = e_begin; e0 < e_end; e0 += GRAM_KS) {
    __syncthreads();
    // stage sqrt(w) * G[e0 .. e0+64) for the columns of both blocks: one wave instruction = one column
    {
      const int64_t e = e0 + lane;
      const bool ein = e < e_end;
      const double swe = ein ? sw[e] : 0.0;
      batched_loop<16>(wave, ni, 4, [=](int c) { return ein ? G[(int64_t)(ci0 + c) * ldg + e] : 0.0; },
                       [=](int c, double v) { sA[lane * GRAM_LD + c] = swe != 0.0 ? v * swe : 0.0; });
      if (!DIAG)
        batched_loop<16>(wave, nj, 4, [=](int c) { return ein ? G[(int64_t)(cj0 + c) * ldg + e] : 0.0; },
                         [=](int c, double v) { sB[lane * GRAM_LD + c] = swe != 0.0 ? v * swe : 0.0; });
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      if (tms[i] < 0) continue;
      const double* const pa = la + 16 * tms[i];
      const double* const pb = lb + 16 * tns[i];
      d4 a = acc[i];
#pragma unroll
      for (int s2 = 0; s2 < GRAM_KS / 4; ++s2)
        a = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[s2 * 4 * GRAM_LD], pa[s2 * 4 * GRAM_LD], a, 0, 0, 0);
      acc[i] = a;
    }
  }
  // write partial tiles: layout [block][chunk][tile][256], tile elements column-major
  double* out = partial + ((int64_t)blk * nchunk_total + coff + blockIdx.x) * (int64_t)(64 * 256);
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    if (tms[i] < 0) continue;
    const int t = tms[i] + tns[i] * mti;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(int64_t)t * 256 + (kq + 4 * r) * 16 + l15] = acc[i][r];
  }
}

// Single-block variant (m <= 128: the whole lower triangle of H is one 128 x 128 block).  The accumulators of
// the (at most 9) lower tiles a wave owns stay in registers under compile-time indices, the slice of the next
// step is fetched into registers while the MFMAs of the current one run (barriers order LDS traffic only), and
// the k-steps of a tile use immediate LDS offsets.  Same partial-tile output layout as k_gram_partial.
// NT = lower tiles per wave (ceil(ntl / 4), a template parameter: the tile loop is straight-line code, tiles are
// processed in PAIRS with independent accumulator chains so that an MFMA never waits for the operands or the result
// of the one before it -- with one dependent chain per tile the MFMA pipe was busy 45 % of the time, PMC
// SQ_VALU_MFMA_BUSY_CYCLES).  Slots past the last tile of a wave recompute tile 0 and are not stored.
// The rows to accumulate are given as a table of SLICES of at most GRAM_KS consecutive blkval positions (sl_start,
// sl_len): the host cuts the blkval ranges of a call -- a rank's subtrees, minus the cliques whose Gram block comes from
// k_leaf_gram -- into slices once per set of ranges; workgroup w takes the slices w * spw .. (w + 1) * spw - 1.  A wave
// keeps the descriptors of 64 slices in one register each (lane i: slice i of the window) and reads them with
// v_readlane, so that walking the table costs no memory round trip.
template <int NT, int NW>
__global__ void __launch_bounds__(64 * NW, (NW == 16 ? 2 : 1)) k_gram_diag128(const double* G, int64_t ldg, int m, const int64_t* sl_start,
                                                      const int32_t* sl_len, int nsl, int spw,
                                                      const double* sw, double* partial, int coff,
                                                      int nchunk_total, int skip) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // LDS image [column][k = 64], ld GRAM_LDK = 66 doubles: the staging store is contiguous along k (no transpose)
  // and the operand reads (row = column l15, k = kq + 4 s) hit 32 distinct 8-byte bank pairs per half wave
  // (bank = 4 l15 + 2 kq): conflict-free, where the [k][column] image with ld 129 had two-way conflicts on every read
  double* const sA = smem;
  const int ni = m;
  const int mti = (ni + 15) >> 4;
  const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NC = GRAM_BLK / NW;
  // this wave's lower tiles p = wave, wave + 4, ... of the (tm >= tn) enumeration
  int tms[NT], tns[NT];
  bool tv[NT];
  const int ntl = mti * (mti + 1) / 2;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    int p = wave + NW * i, tm = 0;
    tv[i] = p < ntl;
    if (!tv[i]) p = 0;
    while (p > tm) { p -= tm + 1; ++tm; }
    tms[i] = tm; tns[i] = p;
  }
  d4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  for (int e = threadIdx.x; e < GRAM_BLK * GRAM_LDK; e += 64 * NW) sA[e] = 0.0;   // columns >= m stay zero
  const int s_begin = (int)blockIdx.x * spw, s_end = min(nsl, s_begin + spw);
  double pre[NC], pre_sw = 0.0;
  int64_t dstart = 0;          // descriptors of the window of 64 slices that holds the running one
  int dlen = 0;
  auto window = [&](int s0) {
    const int s = s0 + lane;
    dstart = s < s_end ? sl_start[s] : 0;
    dlen = s < s_end ? sl_len[s] : 0;
  };
  auto fetch = [&](int s) {    // slice s (inside the loaded window)
    const int i = (s - s_begin) & 63;
    const int64_t e0 = ((int64_t)__builtin_amdgcn_readlane((int)(dstart >> 32), i) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)dstart, i);
    const int len = __builtin_amdgcn_readlane(dlen, i);
    const int64_t e = e0 + lane;
    const bool ein = lane < len;
    pre_sw = ein ? sw[e] : 0.0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int cc = wave + NW * j;
      pre[j] = (ein && cc < ni) ? ((skip & 2) ? 1.0 : G[(int64_t)cc * ldg + e]) : 0.0;
    }
  };
  if (s_begin < s_end) { window(s_begin); fetch(s_begin); }
  const double* const lbase = sA + l15 * GRAM_LDK + kq;
  const double* pa[NT];
  const double* pb[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) { pa[i] = lbase + 16 * GRAM_LDK * tms[i]; pb[i] = lbase + 16 * GRAM_LDK * tns[i]; }
  for (int sl = s_begin; sl < s_end; ++sl) {
    lds_barrier();                                 // the tiles of the previous slice have been consumed
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int cc = wave + NW * j;
      if (cc < ni) sA[cc * GRAM_LDK + lane] = pre_sw != 0.0 ? pre[j] * pre_sw : 0.0;   // weight 0: never-written entries
    }
    lds_barrier();
    if (sl + 1 < s_end) {                          // in flight while the MFMAs below run
      if (((sl + 1 - s_begin) & 63) == 0) window(sl + 1);
      fetch(sl + 1);
    }
    if (!(skip & 1))
#pragma unroll
    for (int i = 0; i < NT; i += 2) {
      if (!tv[i]) continue;                              // slots are filled in order: nothing beyond an empty one
      const bool two = (i + 1 < NT) && tv[(i + 1 < NT) ? i + 1 : i];
      d4 a0 = acc[i], a1 = acc[(i + 1 < NT) ? i + 1 : i];
      if (two) {
#pragma unroll
        for (int s2 = 0; s2 < GRAM_KS / 4; ++s2) {
          a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[i][s2 * 4], pa[i][s2 * 4], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[(i + 1 < NT) ? i + 1 : i][s2 * 4], pa[(i + 1 < NT) ? i + 1 : i][s2 * 4], a1, 0, 0, 0);
        }
        acc[(i + 1 < NT) ? i + 1 : i] = a1;
      } else {
#pragma unroll
        for (int s2 = 0; s2 < GRAM_KS / 4; ++s2)
          a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[i][s2 * 4], pa[i][s2 * 4], a0, 0, 0, 0);
      }
      acc[i] = a0;
    }
  }
  double* out = partial + ((int64_t)coff + blockIdx.x) * (int64_t)(64 * 256);
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (!tv[i]) continue;
    const int t = tms[i] + tns[i] * mti;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(int64_t)t * 256 + (kq + 4 * r) * 16 + l15] = acc[i][r];
  }
}

// H (m x m, ld ldh, both triangles) <- sum over chunks of the partial tiles (those of k_gram_diag128 and those of
// k_leaf_pairs, front_leafgram.hip), in a fixed order
// part_out != null (first of two stages; several hundred chunks on 28 tile workgroups were 43 us of every Schur complement of
// synth50k): gridDim.z workgroups per tile, workgroup z sums the chunks of its range into slot z of part_out -- the layout of
// `partial` with gridDim.z chunks, which the second stage then reduces into H.  The order of the additions stays fixed.
__global__ void k_gram_reduce(const double* partial, int nchunk, int m, double* H, int64_t ldh, double* part_out) {
  int bi = 0, rem = blockIdx.y;
  while (rem > bi) { rem -= bi + 1; ++bi; }
  const int bj = rem;
  const bool diag = bi == bj;
  const int ci0 = bi * GRAM_BLK, cj0 = bj * GRAM_BLK;
  const int ni = min(GRAM_BLK, m - ci0), nj = min(GRAM_BLK, m - cj0);
  const int mti = (ni + 15) >> 4;
  const int t = blockIdx.x;             // tile within the block
  const int tm = t % mti, tn = t / mti;
  if (tn * 16 >= nj || (diag && tm < tn)) return;
  // blockDim.x = 256 x nz: the chunks in nz consecutive ranges, one per group of 256 threads (a tile has few workgroups and
  // several hundred chunks: 763 on synth50k, 54 us with one range), their sums added in range order
  __shared__ double zs[4][256];
  const int idx = threadIdx.x & 255;    // element within the tile (column-major 16 x 16)
  const int z = threadIdx.x >> 8, nz = blockDim.x >> 8;
  const int Z = part_out ? (int)gridDim.z : 1, zz = part_out ? (int)blockIdx.z : 0;
  const int r0 = (int)(((int64_t)nchunk * zz) / Z), r1 = (int)(((int64_t)nchunk * (zz + 1)) / Z);       // this workgroup's chunks
  const int c0 = r0 + (int)(((int64_t)(r1 - r0) * z) / nz), c1 = r0 + (int)(((int64_t)(r1 - r0) * (z + 1)) / nz);
  const double* p = partial + (int64_t)blockIdx.y * nchunk * (int64_t)(64 * 256) + (int64_t)t * 256 + idx;
  // eight independent partial sums (eight loads in flight; the order of the additions is fixed: deterministic)
  double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int c = c0;
  for (; c + 8 <= c1; c += 8) {
#pragma unroll
    for (int q = 0; q < 8; ++q) s8[q] += p[(int64_t)(c + q) * (64 * 256)];
  }
  for (; c < c1; ++c) s8[0] += p[(int64_t)c * (64 * 256)];
  double s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  if (nz > 1) {
    zs[z & 3][idx] = s;
    __syncthreads();
    if (z) return;
    for (int q = 1; q < nz; ++q) s += zs[q][idx];
  }
  if (part_out) {
    part_out[((int64_t)blockIdx.y * Z + zz) * (int64_t)(64 * 256) + (int64_t)t * 256 + idx] = s;
    return;
  }
  const int i = ci0 + tm * 16 + (idx & 15), j = cj0 + tn * 16 + (idx >> 4);
  if (i < ci0 + ni && j < cj0 + nj) {
    H[i + (int64_t)j * ldh] = s;
    H[j + (int64_t)i * ldh] = s;
  }
}

// sw[e] = sqrt(weight of blkval position e) : 1 on the diagonal, sqrt(2) below it, 0 above it
__global__ void k_fill_sqrt_weights(const CliqueDesc* cl, int nsn, double* sw) {
  for (int k = blockIdx.x; k < nsn; k += gridDim.x) {
    const CliqueDesc d = cl[k];
    const int nn = d.nn, nf = d.nn + d.na;
    for (int e = threadIdx.x; e < nf * nn; e += blockDim.x) {
      int i = e % nf, j = e / nf;
      sw[d.blk + e] = (i == j) ? 1.0 : (i > j ? 1.4142135623730951 : 0.0);
    }
  }
}

// self-test of the MFMA operand map: C (M x N) = A (M x Kd) * B (Kd x N), column-major, one workgroup
__global__ void k_selftest_mma(int M, int N, int Kd, const double* A, const double* B, double* C) {
  wg_mma(M, N, Kd, [=](int m, int kk) { return A[m + kk * M]; }, [=](int kk, int n) { return B[kk + n * Kd]; },
         [=](int m, int n, double acc) { C[m + n * M] = acc; });
}

}  // namespace smcp
