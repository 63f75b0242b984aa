// Large-front path: fronts whose working set does not fit LDS are processed by PHASE kernels whose
// grid covers (output tile, clique, right-hand side).  Each workgroup computes one 64x64 output tile
// with operands staged through LDS in 16-deep slices (coalesced global -> LDS, MFMA from LDS) -- the
// classic tiled GEMM -- so a single big front (config 2: one 4096 front) spreads over the whole
// chip and the 192-fronts of config 3 run as batched GEMMs.  Intermediates (E, G, T) live in the
// per-(clique, rhs) scratch, which stays L2 / Infinity-Cache resident between phases.
#include <hip/hip_runtime.h>
#include <type_traits>

namespace smcp {

constexpr int LT = 64;    // output tile edge
constexpr int LKC = 16;   // k slice
constexpr int LSA = LT + 1, LSB = LKC + 1;

// acc (2x2 MFMA tiles per wave, waves arranged 2x2 over the 64x64 tile) += A[m0.., :] * B[:, n0..]
// la(m, k) / lb(k, n): element loaders (global memory, any layout / symmetry); out-of-range -> 0.
// The products run over k in [kbeg, Kd): a caller whose operand is triangular passes the range where it is nonzero
// (kbeg is rounded down to a slice boundary; Li is stored with explicit zeros above its diagonal, so the bounds only
// skip slices of zeros -- half of T = Li F_NN, G = X Li^T ... on a big front).
// masked operand element: the sixteen-wave shape loads unconditionally and masks (see gemm_tile64), the four-wave shape
// keeps the load under its condition (the masked half of a triangular operand is not fetched at all)
template <int PD>
__device__ inline double ldm(bool c, const double* p) {
  if constexpr (PD == 4) { const double v = *p; return c ? v : 0.0; }
  else return c ? *p : 0.0;
}
// Two shapes of the same tile product.  PD = 1 (256 threads): four waves, 2 x 2 MFMA tiles each -- the batched sweeps, where
// several workgroups share a CU.  Every kernel of this shape is compiled for FOUR waves per SIMD (__launch_bounds__(256, 4):
// 128 registers, accumulators in VGPRs; round 5): left to itself the compiler took 126 - 134 VGPRs + 32 - 64 AGPRs, i.e. two or
// three workgroups per CU, and a tile's life is mostly the round trips before and after its MFMAs -- k_lf_up2 on the top fronts of
// synth50k 139 -> 107 us, on the 4096 front of config 2 15.5 -> 14.9 ms, k_lf_down3 there 1.89 -> 1.62 ms (no or <= 6 spills).  PD = 4 (1024 threads, "W16"): sixteen waves, ONE 16 x 16 MFMA tile each, for launches of
// a few tiles (one right-hand side on the top fronts: the solves of the interior-point iteration).  In-kernel stamps
// (scratch/stamps_lf.py) showed a slice of the 256-thread shape to cost ~1.3 us on an otherwise idle CU -- 16 MFMAs (64 cycles
// each), the address arithmetic of 8 loads and two barriers, all on ONE wave per SIMD, instruction bound, not latency
// bound (deeper prefetch and batched loads changed nothing); with sixteen waves a slice is 4 MFMAs and 2 loads per wave.
// Every load is issued unconditionally at an index clamped into the operand and masked afterwards: a load under a
// branch makes the number of loads in flight unknown to the compiler and every wait becomes vmcnt(0).
template <int PD = 1, class LA, class LB>
__device__ inline void gemm_tile64(d4 (&acc)[2][2], int M, int N, int Kd, int m0, int n0, LA la, LB lb,
                                   double* sA, double* sB, int kbeg = 0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int kfirst = (kbeg / LKC) * LKC;
  if (kfirst >= Kd) return;
  if constexpr (PD == 4) {
    const int wm = wave & 3, wn = wave >> 2;
    const int ia = tid & 63, ka = tid >> 6, kb = tid & 15, jb = tid >> 4;
    const int mi = max(0, min(m0 + ia, M - 1)), nj = max(0, min(n0 + jb, N - 1));
    const bool inA = m0 + ia < M, inB = n0 + jb < N;
    // Two slices in flight.  The operands were written by the previous launch, on some other XCD: every first touch is a
    // trip to the memory side of the fabric.  Requests past the last slice would be waited for at the end of the product
    // (one more such trip), and a request under a run-time condition makes every wait a wait for all loads -- so the main
    // loop requests unconditionally and the last (up to three) slices are peeled without requests.
    double va[2], vb[2];
    auto fetch = [&](int p, int k0) {
      const double av = la(mi, min(k0 + ka, Kd - 1)), bv = lb(min(k0 + kb, Kd - 1), nj);
      va[p] = (inA && k0 + ka < Kd) ? av : 0.0;
      vb[p] = (inB && k0 + kb < Kd) ? bv : 0.0;
    };
    auto slice = [&](int p, int k0, auto ahead) {
      __syncthreads();
      sA[ia + ka * LSA] = va[p];
      sB[kb + jb * LSB] = vb[p];
      __syncthreads();
      if constexpr (decltype(ahead)::value) fetch(p, k0 + 2 * LKC);
#pragma unroll
      for (int ks = 0; ks < LKC / 4; ++ks) {
        const int kk = 4 * ks + kq;
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sB[kk + (16 * wn + l15) * LSB], sA[(16 * wm + l15) + kk * LSA], acc[0][0], 0, 0, 0);
      }
    };
    constexpr std::integral_constant<bool, true> yes{};
    constexpr std::integral_constant<bool, false> no{};
    const int nsl = (Kd - kfirst + LKC - 1) / LKC;
    fetch(0, kfirst);
    fetch(1, kfirst + LKC);            // (a product of one slice: clamped, masked, waited for at the end)
    int sl = 0, k0 = kfirst;
    for (; sl + 3 < nsl; sl += 2, k0 += 2 * LKC) { slice(0, k0, yes); slice(1, k0 + LKC, yes); }
    const int rest = nsl - sl;         // 1 .. 3
    if (rest == 3) { slice(0, k0, yes); slice(1, k0 + LKC, no); slice(0, k0 + 2 * LKC, no); }
    else if (rest == 2) { slice(0, k0, no); slice(1, k0 + LKC, no); }
    else slice(0, k0, no);
    return;
  } else {
  const int wm = wave & 1, wn = wave >> 1;
  // the operands of slice k0 + LKC are fetched into registers while the MFMAs of slice k0 run
  double va[4], vb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      const int i = e & 63, kk = e >> 6;
      // (guarded loads here: with clamped, unconditional ones -- what the sixteen-wave shape needs -- the batched sweeps
      // lost 10 - 45 %: k_lf_up1 on the root of synth50k, 100 right-hand sides, 76 -> 110 us)
      va[u] = (m0 + i < M && k0 + kk < Kd) ? la(m0 + i, k0 + kk) : 0.0;
      const int kb = e & 15, j = e >> 4;
      vb[u] = (n0 + j < N && k0 + kb < Kd) ? lb(k0 + kb, n0 + j) : 0.0;
    }
  };
  fetch(kfirst);
  for (int k0 = kfirst; k0 < Kd; k0 += LKC) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      sA[(e & 63) + (e >> 6) * LSA] = va[u];
      sB[(e & 15) + (e >> 4) * LSB] = vb[u];
    }
    __syncthreads();
    if (k0 + LKC < Kd) fetch(k0 + LKC);
#pragma unroll
    for (int ks = 0; ks < LKC / 4; ++ks) {
      const int kk = 4 * ks + kq;
      const double a0 = sA[(32 * wm + l15) + kk * LSA], a1 = sA[(32 * wm + 16 + l15) + kk * LSA];
      const double b0 = sB[kk + (32 * wn + l15) * LSB], b1 = sB[kk + (32 * wn + 16 + l15) * LSB];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
    }
  }
  }
}
// The same product for the common case of the batched shape (PD = 1, 256 threads): both operands plain matrices whose FIRST
// index is the contiguous one -- A(m, k) = A[m + k lda], B(k, n) = B[n + k ldb], pointers already at the tile's first row --
// the whole 64 x 64 tile inside both, the inner range [0, Kd) a multiple of the slice, 16-byte aligned columns.  Each thread
// moves two row pairs per operand and slice with 16-byte loads at pointers that only advance (no index arithmetic, no guards),
// both operands are staged row-contiguous (ld 80: k and k + 1 land 32 banks apart, the MFMA operand reads are conflict free)
// and the staging area is double buffered: one barrier per slice.  On the 4096 front of config 2 (tools/micro/tile_gemm.hip:
// lower tiles of Z Li^T + Li Z^T, six right-hand sides) 46.9 -> 57.7 TFLOP/s, 62.9 with the widest tiles dispatched first.
constexpr int LRC = 80;                         // leading dimension of a staged slice whose rows are contiguous in memory
constexpr int LCC = 18;                         // ... of one whose inner index is (k contiguous: k + row * 18, operand reads conflict free)
constexpr int LRC_DOUBLES = 4 * LKC * LRC;      // two operands, two buffers (a slice: 16 x 80 or 64 x 18 doubles)
__device__ inline bool rc_aligned(const double* p, int64_t ld) { return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ((ld & 1) == 0); }
// one operand of the fast product: XC = false: X(row, k) = X[row + k ld], XC = true: X(row, k) = X[k + row ld]; p at (first row, first k)
template <bool XC>
struct RcOperand {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const double* p; int64_t step, second;
  int st0, st1;                                  // staging offsets of the two pairs
  d2 r[2];
  __device__ inline void init(const double* X, int64_t ld) {
    const int tid = threadIdx.x;
    if constexpr (XC) {
      const int k2 = tid & 7, row = tid >> 3;    // pairs (k, k + 1) of rows `row` and `row + 32`
      p = X + 2 * k2 + (int64_t)row * ld; step = LKC; second = 32 * ld;
      st0 = 2 * k2 + row * LCC; st1 = st0 + 32 * LCC;
    } else {
      const int r2 = tid & 31, kk0 = tid >> 5;   // row pairs (2 r2, 2 r2 + 1) at k = kk0 and kk0 + 8
      p = X + 2 * r2 + (int64_t)kk0 * ld; step = (int64_t)LKC * ld; second = 8 * ld;
      st0 = 2 * r2 + kk0 * LRC; st1 = st0 + 8 * LRC;
    }
  }
  __device__ inline void fetch() { r[0] = *reinterpret_cast<const d2*>(p); r[1] = *reinterpret_cast<const d2*>(p + second); p += step; }
  __device__ inline void stage(double* sX) const { *reinterpret_cast<d2*>(sX + st0) = r[0]; *reinterpret_cast<d2*>(sX + st1) = r[1]; }
  static __device__ inline double frag(const double* sX, int row, int kk) { return XC ? sX[kk + row * LCC] : sX[row + kk * LRC]; }
};
// acc += A[rows of the tile, k0 .. k1) * B[k0 .. k1), columns of the tile]; A at (m0, k0), B at (n0, k0); k1 - k0 a multiple of 16
template <bool AC, bool BC>
__device__ inline void gemm_tile64_rc(d4 (&acc)[2][2], const double* A, int64_t lda, const double* B, int64_t ldb, int nk, double* smem) {
  constexpr int SL = LKC * LRC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  const int ns = nk / LKC;
  if (ns <= 0) return;
  RcOperand<AC> oa;
  RcOperand<BC> ob;
  oa.init(A, lda);
  ob.init(B, ldb);
  oa.fetch(); ob.fetch();
  __syncthreads();                               // (a previous product of this workgroup may still read the buffers)
  oa.stage(smem); ob.stage(smem + 2 * SL);
  __syncthreads();
  for (int s = 0; s < ns; ++s) {
    const double* sA = smem + (s & 1) * SL;
    const double* sB = smem + 2 * SL + (s & 1) * SL;
    if (s + 1 < ns) { oa.fetch(); ob.fetch(); }
#pragma unroll
    for (int ks = 0; ks < LKC / 4; ++ks) {
      const int kk = 4 * ks + kq;
      const double a0 = RcOperand<AC>::frag(sA, 32 * wm + l15, kk), a1 = RcOperand<AC>::frag(sA, 32 * wm + 16 + l15, kk);
      const double b0 = RcOperand<BC>::frag(sB, 32 * wn + l15, kk), b1 = RcOperand<BC>::frag(sB, 32 * wn + 16 + l15, kk);
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
    }
    if (s + 1 < ns) { oa.stage(smem + ((s + 1) & 1) * SL); ob.stage(smem + 2 * SL + ((s + 1) & 1) * SL); }
    __syncthreads();
  }
}
// Plain operands over the inner range [k0, k1), k0 a multiple of the slice: A(m, k) = A[m + k lda] (AC: A[k + m lda]), M rows;
// B(k, n) = B[n + k ldb] (BC: B[k + n ldb]), N columns.  The fast product where it applies, else the general one.
template <int PD, bool AC = false, bool BC = false>
__device__ inline void gemm_tile64_plain(d4 (&acc)[2][2], const double* A, int64_t lda, int M, const double* B, int64_t ldb, int N, int k0, int k1,
                                         int m0, int n0, double* smem) {
  if (k1 <= k0) return;
  if constexpr (PD == 1) {
    const double* A0 = AC ? A + k0 + (int64_t)m0 * lda : A + m0 + (int64_t)k0 * lda;
    const double* B0 = BC ? B + k0 + (int64_t)n0 * ldb : B + n0 + (int64_t)k0 * ldb;
    if (m0 + LT <= M && n0 + LT <= N && ((k1 - k0) % LKC) == 0 && (k0 % LKC) == 0 && rc_aligned(A0, lda) && rc_aligned(B0, ldb)) {
      gemm_tile64_rc<AC, BC>(acc, A0, lda, B0, ldb, k1 - k0, smem);
      return;
    }
  }
  gemm_tile64<PD>(acc, M, N, k1, m0, n0, [=](int m, int kk) { return AC ? A[kk + (int64_t)m * lda] : A[m + (int64_t)kk * lda]; },
                  [=](int kk, int n) { return BC ? B[kk + (int64_t)n * ldb] : B[n + (int64_t)kk * ldb]; }, smem, smem + LKC * LSA, k0);
}
// accumulator element (a, b, r) of this lane -> (m, n) inside the 64 x 64 tile; false: the lane has no such element
// (the sixteen-wave shape keeps one MFMA tile per wave in acc[0][0])
__device__ inline bool tile64_pos(int a, int b, int r, int& m, int& n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  if (blockDim.x == 1024) {
    m = 16 * (wave & 3) + l15; n = 16 * (wave >> 2) + kq + 4 * r;
    return a == 0 && b == 0;
  }
  m = 32 * (wave & 1) + 16 * a + l15; n = 32 * (wave >> 1) + 16 * b + kq + 4 * r;
  return true;
}
// visit the accumulator elements of this lane: f(m, n, value)
template <class F>
__device__ inline void tile64_foreach(const d4 (&acc)[2][2], int m0, int n0, int M, int N, F f) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m, n;
        const bool has = tile64_pos(a, b, r, m, n);
        m += m0; n += n0;
        if (has && m < M && n < N) f(m, n, acc[a][b][r]);
      }
}
// The same for results that are combined with values already in memory: ld(m, n) is called for ALL sixteen elements of the
// lane first (unconditionally, at indices clamped into the M x N operand), then st(m, n, value, loaded) for those inside.
// A load inside the guarded visit is a load under a branch: the compiler waits for each before it issues the next, and
// sixteen dependent memory round trips were most of the time of a launch of a few tiles (k_lf_up1 / up2 / down2 for
// one right-hand side on the top fronts of synth50k: 13 - 17 us each).
template <class LD, class ST>
__device__ inline void tile64_rmw(const d4 (&acc)[2][2], int m0, int n0, int M, int N, LD ld, ST st) {
  double old[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m, n;
        tile64_pos(a, b, r, m, n);
        old[a][b][r] = ld(max(0, min(m0 + m, M - 1)), max(0, min(n0 + n, N - 1)));
      }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m, n;
        const bool has = tile64_pos(a, b, r, m, n);
        m += m0; n += n0;
        if (has && m < M && n < N) st(m, n, acc[a][b][r], old[a][b][r]);
      }
}
__device__ inline void tile64_zero(d4 (&acc)[2][2]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
}
__host__ __device__ inline int tiles64(int x) { return (x + LT - 1) / LT; }
// linear index -> (tm >= tn) lower tile pair
__device__ inline void lower_pair(int t, int& tm, int& tn) {
  tm = 0;
  while (t > tm) { t -= tm + 1; ++tm; }
  tn = t;
}
// the same tiles, the last tile column first: products whose inner range ends at the tile's last column (triangular operands)
// cost in proportion to tn + 1, and workgroups are dispatched in index order -- with the cheap tiles last the launch ends on
// them instead of on a few workgroups still walking the longest ranges (4096 front, one right-hand side, three workgroups
// per CU: a list-scheduling model gives 0.58 of the balanced time for the row-major order, 0.93 for this one)
__device__ inline void lower_pair_wide_first(int t, int nt, int& tm, int& tn) {
  int q = 0;
  while (t > q) { t -= q + 1; ++q; }
  tn = nt - 1 - q;
  tm = tn + t;
}

struct LfCtx {   // per-workgroup view of one (clique, rhs) pair
  int k, nn, na, nf;
  bool hasch;                          // the front has children (its assembled update block is not identically zero)
  const double* Li; const double* K;   // ld nf
  const double* Ys;                    // ld na (lower stored) or null
  double* P;                           // panel of this rhs (ld nf)
  double* U;                           // update / separator matrix of this rhs (ld na)
  double* UP;                          // its packed lower triangle in the exchange buffer
  double* T; double* E; double* G;     // scratch: nn x nn, na x nn, na x nn
};
__device__ inline LfCtx lf_ctx(const MfmaArgs& a, double* u, int64_t ldu) {
  LfCtx c;
  c.k = a.t.lev[blockIdx.y];
  const CliqueDesc d = a.t.cl[c.k];
  const int r = blockIdx.z;
  c.nn = d.nn; c.na = d.na; c.nf = d.nn + d.na;
  c.hasch = d.chend > d.chbeg;
  c.Li = a.LK ? a.LK + d.blk : nullptr;
  c.K = a.LK ? a.LK + d.blk + d.nn : nullptr;
  c.Ys = a.ysc ? a.ysc + d.upd : nullptr;
  c.P = u + (int64_t)r * ldu + d.blk;
  c.U = a.t.upd + (int64_t)r * a.t.updlen + d.upd;
  c.UP = a.t.updp + (int64_t)r * a.t.updplen + d.updp;
  double* s = a.t.tmp + (int64_t)r * a.t.tmplen + a.t.tmpptr[c.k];
  c.T = s; s += (int64_t)d.nn * d.nn;
  c.E = s; s += (int64_t)d.na * d.nn;
  c.G = s;
  return c;
}

// ---- phase 0: zero the update block and add the children (gather plan: one owner per position)
// sgn 0: U = children, panel += children (U was cleared);  sgn 1: U += children, panel -= children;
// sgn 2: U += children, panel += children
// CH: contributions requested per round (16; 8 for fronts of at most eight children: a position of the root of synth50k sums
// at most eight and the clamped requests beyond the list are wasted loads -- 63 -> see DESIGN)
template <int CH>
__global__ void k_lf_assemble(MfmaArgs a, double* u, int64_t ldu, int sgn) {
  const int k = a.t.lev[blockIdx.y];
  const CliqueDesc d = a.t.cl[k];
  const int r = blockIdx.z;
  const int nn = d.nn, na = d.na, nf = nn + na;
  double* P = u + (int64_t)r * ldu + d.blk;
  const double* ubase = a.t.updp + (int64_t)r * a.t.updplen;   // children: packed exchange buffer
  double* U = a.t.upd + (int64_t)r * a.t.updlen + d.upd;
  const int64_t t0 = a.t.gp_tptr[k], t1 = a.t.gp_tptr[k + 1];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // positions of U that receive no contribution must read as zero: clear the lower triangle first
  // (phase-0 launches for one level are preceded by this kernel's own clear pass: see host code)
  for (int64_t tt = t0 + gid; tt < t1; tt += stride) {
    const int32_t code = a.t.gp_tgt[tt];
    const int64_t c0 = a.t.gp_cptr[tt], c1 = a.t.gp_cptr[tt + 1];
    // sixteen contributions per round of two dependent loads (source index, value), issued unconditionally at clamped
    // positions: a position of the (64,128) fronts of synth50k sums 12.6 children on average, and with four per round
    // and a tail loop the chain was ~9 memory round trips (24 us for one right-hand side); the sum keeps its fixed order
    double acc = 0.0;
    for (int64_t cc = c0; cc < c1; cc += CH) {
      int32_t sx[CH];
      double vx[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) sx[q] = a.t.gp_src[min(cc + q, c1 - 1)];
#pragma unroll
      for (int q = 0; q < CH; ++q) vx[q] = ubase[sx[q]];
#pragma unroll
      for (int q = 0; q < CH; ++q) acc += (cc + q < c1) ? vx[q] : 0.0;
    }
    const int i = code & 0x7fff, j = (code >> 15) & 0x7fff;
    if (sgn) {
      if (code & (1 << 30)) U[i + (int64_t)j * na] += acc; else P[i + (int64_t)j * nf] += (sgn == 2 ? acc : -acc);
    } else {
      if (code & (1 << 30)) U[i + (int64_t)j * na] = acc;     // U was cleared: plain store
      else P[i + (int64_t)j * nf] += acc;
    }
  }
}
// Extend-add with the whole front in LDS: one workgroup owns the packed lower triangle of the front of (clique,
// rhs) -- nf (nf + 1) / 2 doubles, i.e. fronts of up to 198 rows -- and STREAMS the children's packed update matrices
// through it child-major: every child column is one contiguous, fully used segment (the gather plan above reads
// 8-byte words scattered over the children, ~1.2 TB/s), the scatter into the front is an LDS atomic add
// (ds_add_f64) at (rel[i], rel[j]).  Waves take children round-robin and keep four columns in flight.
// Same sgn convention as k_lf_assemble.
__device__ inline double col_at(const double* Uc, int nac, int i, int j) { return Uc[pk_col(j, nac) - j + i]; }
constexpr int LF_ALDS_MAXNF = 198;
__host__ __device__ inline int lf_alds_doubles(int nf) { return nf * (nf + 1) / 2; }
// The children of one wave as ONE stream of column batches.  A batch is sixteen wave loads -- sixteen columns of a child with at most 64 rows, eight columns x two row halves
// of a taller one -- plus the two loads of the child's relative indices.  Every load is unconditional (column and row
// clamped into the child's packed matrix; masked at the add), so the compiler counts them (s_waitcnt vmcnt(N)) instead of
// draining the queue before the first add as it does for loads under branches: with two batch buffers used in turn the
// NEXT batch -- of this child or of the wave's next child -- is in flight while the current one goes into the front.
// (Rows and columns beyond 128 are left to lf_add_child_tail.)
struct AldsBatch { double v[16]; int rA, rB; };
struct AldsItem { int q, nac, j0, b; };
__device__ inline int alds_cols(int nac) { return nac <= 64 ? 16 : 8; }
__device__ inline int alds_batches(int nac) { return nac > 0 ? (min(nac, 128) + alds_cols(nac) - 1) / alds_cols(nac) : 0; }
// The batches of ALL children of the task, in child order, are dealt round-robin to the nw waves (batch g to wave g mod nw):
// the children of a front lie one after the other in the exchange buffer, so the workgroup reads ONE sequential stream with
// its waves a few KB apart, instead of sixteen streams 16 KB apart that each stop and start (what the DRAM pages see: 256
// streams on the chip instead of 4096) -- and a front with fewer children than waves keeps every wave busy without a
// special case.
__device__ __forceinline__ void lf_add_children_stream(double* T, int nf, const double* ubase, const int32_t* relidx, const int64_t* sCu,
                                              const int64_t* sCr, const int* sCn, int nmine, int wave, int nw, int lane) {
  auto cb = [nf](int j) { return j * nf - ((j * (j - 1)) >> 1) - j; };
  auto norm = [&](int q, int b) {                       // (child q, batch b of it or of a later child) -> item
    AldsItem it{nmine, 0, 0, 0};
    for (; q < nmine; ++q) {
      const int nac = sCn[q], nb = alds_batches(nac);
      if (b < nb) { it.q = q; it.nac = nac; it.j0 = b * alds_cols(nac); it.b = b; break; }
      b -= nb;
    }
    it.q = __builtin_amdgcn_readfirstlane(it.q); it.nac = __builtin_amdgcn_readfirstlane(it.nac);
    it.j0 = __builtin_amdgcn_readfirstlane(it.j0); it.b = __builtin_amdgcn_readfirstlane(it.b);
    return it;
  };
  // (children of at most 64 rows -- four batches each -- are dealt whole instead, child q to wave q mod nw, when there are
  // at least nw of them: measured on synth50k, 112 children of 64 rows per front, 0.58 against 0.62 ms per sweep; on config 3,
  // 250 summed updates of 128 rows under the root, the sequential deal wins, 0.81 against 1.13 ms)
  auto normc = [&](int q, int b) {
    AldsItem it{nmine, 0, 0, 0};
    for (; q < nmine; q += nw, b = 0) {
      const int nac = sCn[q], nb = alds_batches(nac);
      if (b < nb) { it.q = q; it.nac = nac; it.j0 = b * alds_cols(nac); it.b = b; break; }
    }
    it.q = __builtin_amdgcn_readfirstlane(it.q); it.nac = __builtin_amdgcn_readfirstlane(it.nac);
    it.j0 = __builtin_amdgcn_readfirstlane(it.j0); it.b = __builtin_amdgcn_readfirstlane(it.b);
    return it;
  };
  bool tall = false;                                    // any child of more than 64 rows (skipped group members count as empty)
  for (int q = lane; q < nmine; q += 64) tall = tall || sCn[q] > 64;
  const bool seq = nmine < nw || __builtin_amdgcn_ballot_w64(tall) != 0;
  auto next = [&](const AldsItem& it) { return seq ? norm(it.q, it.b + nw) : normc(it.q, it.b + 1); };
  auto issue = [&](AldsBatch& b, const AldsItem& it) {
    const double* Uc = ubase + sCu[it.q];
    const int32_t* rel = relidx + sCr[it.q];
    const int nac = it.nac, last = nac - 1;
    const bool two = nac > 64;
    b.rA = rel[min(lane, last)];
    b.rB = rel[min(lane + 64, last)];
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int j = min(it.j0 + (two ? x >> 1 : x), last);
      const int i = min(max(lane + ((two && (x & 1)) ? 64 : 0), j), last);
      b.v[x] = __builtin_nontemporal_load(&Uc[pk_col(j, nac) - j + i]);
    }
  };
  auto process = [&](const AldsBatch& b, const AldsItem& it) {
    const int nac = it.nac, nh = min(nac, 128);
    const bool two = nac > 64;
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int j = it.j0 + (two ? x >> 1 : x);
      const bool hi = two && (x & 1);
      const int i = lane + (hi ? 64 : 0);
      const int jc = min(j, nh - 1);
      const int cj = jc < 64 ? __builtin_amdgcn_readlane(b.rA, jc & 63) : __builtin_amdgcn_readlane(b.rB, jc & 63);
      if (j < nh && i >= j && i < nac) unsafeAtomicAdd(&T[cb(cj) + (hi ? b.rB : b.rA)], b.v[x]);
    }
  };
  AldsItem it = seq ? norm(0, wave) : normc(wave, 0);
  if (it.q >= nmine) return;
  AldsBatch A, B;
  issue(A, it);
  for (;;) {
    const AldsItem n1 = next(it);
    const bool v1 = n1.q < nmine;
    issue(B, v1 ? n1 : it);
    process(A, it);
    if (!v1) break;
    const AldsItem n2 = next(n1);
    const bool v2 = n2.q < nmine;
    issue(A, v2 ? n2 : n1);
    process(B, n1);
    if (!v2) break;
    it = n2;
  }
}
// rows (and with them columns) beyond 128 of a child: plain loop, one wave per child
__device__ inline void lf_add_child_tail(double* T, int nf, const double* Uc, const int32_t* rel, int nac, int lane) {
  auto cb = [nf](int j) { return j * nf - ((j * (j - 1)) >> 1) - j; };
  for (int j = 0; j < nac; ++j) {
    const int cbj = cb(rel[j]);
    for (int i = 128 + lane; i < nac; i += 64)
      if (i >= j) unsafeAtomicAdd(&T[cbj + rel[i]], col_at(Uc, nac, i, j));
  }
}
// (body of one (front, right-hand side, share of the children) task; bx / by / bz and nz: block index and z-extent of the grid
// in the plain launch)
// sgn 3 (nz = 1 only): sgn 0 for a sparse right-hand side whose dense input panel has NOT been built -- the constraint's
// entries of the front (MfmaArgs::kc_*) are added to the front in LDS and the panel is stored whole (no k_panel_fill pass
// over the panels before, no read of them here: 2 x 79 MB per Schur sweep on synth50k)
// fam: hook of the fused extend-add (front_famt.hip: k_lf_assemble_fz) -- fam(T, nf, r, sFz, nmine, wave, nw, lane) adds the
// update matrices of the children listed in sFz (family parents whose updates were not formed by the family sweep, a.fz_on);
// the stream skips those children
struct AldsNoFam { __device__ void operator()(double*, int, int, const int*, int, int, int, int) const {} };
template <class Fam = AldsNoFam>
__device__ inline void lf_alds_task(const MfmaArgs& a, double* u, int64_t ldu, int sgn, int bx, int by, int bz, int nz, double* T, Fam fam = Fam()) {
  const bool fill = sgn == 3;
  if (fill) sgn = 0;
  const int k = a.t.lev[bx];
  const CliqueDesc d = a.t.cl[k];
  if (d.chend == d.chbeg && !fill) {
    if (!sgn) {      // a childless front among fronts with children: its update block is assigned too (zero)
      double* U0 = a.t.upd + (int64_t)by * a.t.updlen + d.upd;
      for (int e = threadIdx.x; e < d.na * d.na; e += blockDim.x) U0[e] = 0.0;
    }
    return;
  }
  const int r = by;
  const int nn = d.nn, na = d.na, nf = nn + na;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int ntot = lf_alds_doubles(nf);
  // child table in LDS (packed-update offset, separator size, relative-index offset): a wave starting on a child
  // then pays one memory latency (its relative indices and first columns together), not a chain of four
  const int nch = d.chend - d.chbeg;
  // (gridDim.z workgroups share the children of the pair: this one takes the children wz, wz + nz, ... -- its table
  // holds only those)
  const int wz = bz;
  const int nmine = (nch - wz + nz - 1) / nz;
  const int tcap = (a.nchmax + nz - 1) / nz;
  int64_t* const sCu = reinterpret_cast<int64_t*>(T + lf_alds_doubles(a.nnmax + a.namax));
  int64_t* const sCr = sCu + tcap;
  int* const sCn = reinterpret_cast<int*>(sCr + tcap);
  int* const sFz = sCn + tcap;                 // (a.fz_on only: the launch sizes the table for it)
  for (int qi = tid; qi < nmine; qi += nthr) {
    const int ck = a.t.chidx[d.chbeg + wz + qi * nz];
    const CliqueDesc c = a.t.cl[ck];
    const bool isfz = a.fz_on && a.fz_no[ck] >= 0;
    // (a member of a sibling group that did not write its slot in this sweep -- its sum is in the leader's -- counts as empty;
    // so does a family parent whose update the hook computes)
    sCu[qi] = c.updp; sCr[qi] = c.rel; sCn[qi] = ((a.chskip && a.chskip[ck]) || isfz) ? 0 : c.na;
    if (a.fz_on) {
      // a family child: everything its wave needs to start -- the slot of the family's tables | nn << 19 | na << 24, and (in the
      // unused packed-update entry) the first term of (family, this right-hand side's constraint) | the term count << 32
      int z = -1;
      if (isfz) {
        const int f = a.fz_no[ck];
        const int jz = a.kc_ids ? a.kc_ids[a.kc_j0 + by] : a.kc_j0 + by;
        const int32_t* pp = a.fz_ptr + (int64_t)f * a.fz_stride + jz;
        const int p0 = pp[0], tn = pp[1] - p0;
        sCu[qi] = (int64_t)(uint32_t)p0 | ((int64_t)tn << 32);
        z = a.fz_slot[f] | (c.nn << 19) | (c.na << 24);      // (nn <= 16, na <= 64, fewer than 2^19 families: host-checked)
      }
      sFz[qi] = z;
    }
  }
  for (int e = tid; e < ntot; e += nthr) T[e] = 0.0;
  __syncthreads();
  const double* ubase = a.t.updp + (int64_t)r * a.t.updplen;
  // packed column start minus the column index: T[cb(j) + i] = front(i, j), i >= j
  auto cb = [nf](int j) { return j * nf - ((j * (j - 1)) >> 1) - j; };
  // fewer children than waves: `parts` waves share a child and take its column batches round-robin (eight children
  // on sixteen waves left half of the workgroup idle)
  // gridDim.z workgroups share the children of a (front, right-hand side) pair (child q belongs to workgroup
  // q mod gridDim.z) and add their partial fronts into the panel / the cleared update block with global atomics: a
  // front with very many children (config 3: 1999 under the root, i.e. 100 workgroups for 13 GB of child blocks) or a
  // launch whose workgroup count leaves a poor last round is spread finer this way
  lf_add_children_stream(T, nf, ubase, a.t.relidx, sCu, sCr, sCn, nmine, wave, nw, lane);
  for (int qi = wave; qi < nmine; qi += nw)
    if (sCn[qi] > 128) lf_add_child_tail(T, nf, ubase + sCu[qi], a.t.relidx + sCr[qi], sCn[qi], lane);
  if (a.fz_on) fam(T, nf, r, sFz, nmine, wave, nw, lane);
  __syncthreads();
  double* P = u + (int64_t)r * ldu + d.blk;
  double* U = a.t.upd + (int64_t)r * a.t.updlen + d.upd;
  if (fill) {
    if (wz == 0) {       // (shared pairs: the constraint's own entries once)
      const int j = a.kc_ids ? a.kc_ids[a.kc_j0 + r] : a.kc_j0 + r;
      const int32_t* kp = a.kc_ptr + (int64_t)k * a.kc_stride;
      for (int p = kp[j] + tid; p < kp[j + 1]; p += nthr) {
        const int off = a.kc_off[p], i = off % nf, jc = off / nf;
        if (i >= jc) unsafeAtomicAdd(&T[cb(jc) + i], a.kc_val[p]);
      }
    }
    __syncthreads();
    if (nz > 1) {        // a share of the pair's children: what it gathered goes into the panel and the update block the launch cleared
      for (int e = tid; e < nf * nn; e += nthr) {
        const int i = e % nf, jc = e / nf;
        if (i >= jc) { const double v = T[cb(jc) + i]; if (v != 0.0) unsafeAtomicAdd(&P[e], v); }
      }
      for (int e = tid; e < na * na; e += nthr) {
        const int i = e % na, jc = e / na;
        if (i >= jc) { const double v = T[cb(nn + jc) + nn + i]; if (v != 0.0) unsafeAtomicAdd(&U[e], v); }
      }
      return;
    }
    for (int e = tid; e < nf * nn; e += nthr) {
      const int i = e % nf, jc = e / nf;
      if (i >= jc) P[e] = T[cb(jc) + i];
    }
    for (int e = tid; e < na * na; e += nthr) {
      const int i = e % na, jc = e / na;
      if (i >= jc) U[e] = T[cb(nn + jc) + nn + i];
    }
    return;
  }
  if (nz > 1) {        // partial front: atomics into the panel and the (pre-cleared or accumulating) update block
    const double spz = (sgn == 1) ? -1.0 : 1.0;
    for (int e = tid; e < nf * nn; e += nthr) {
      const int i = e % nf, j = e / nf;
      if (i >= j) { const double v = T[cb(j) + i]; if (v != 0.0) unsafeAtomicAdd(&P[e], spz * v); }
    }
    for (int e = tid; e < na * na; e += nthr) {
      const int i = e % na, j = e / na;
      if (i >= j) { const double v = T[cb(nn + j) + nn + i]; if (v != 0.0) unsafeAtomicAdd(&U[e], v); }
    }
    return;
  }
  // write-out: flat loops over the rectangular panel / the square update block with eight read-modify-writes in
  // flight per thread (a column-by-column loop would pay one memory latency per column)
  const double sp = (sgn == 1) ? -1.0 : 1.0;
  batched_loop<8>(tid, nf * nn, nthr, [=](int e) { return P[e]; },
                  [=](int e, double pv) {
                    const int i = e % nf, j = e / nf;
                    if (i >= j) { const double v = T[cb(j) + i]; if (v != 0.0) P[e] = pv + sp * v; }
                  });
  if (sgn) {
    batched_loop<8>(tid, na * na, nthr, [=](int e) { return U[e]; },
                    [=](int e, double uv) {
                      const int i = e % na, j = e / na;
                      if (i >= j) { const double v = T[cb(nn + j) + nn + i]; if (v != 0.0) U[e] = uv + v; }
                    });
  } else {
    for (int e = tid; e < na * na; e += nthr) {
      const int i = e % na, j = e / na;
      if (i >= j) U[e] = T[cb(nn + j) + nn + i];                       // full assignment of the lower triangle
    }
  }
}
// panel and update block of the (front, right-hand side) pairs whose children are shared among several workgroups of
// k_lf_assemble_fz (the pairs tail_first .. of every queue): cleared here, added to there.  grid (chunks, pairs per queue, 8)
__global__ void k_lf_zero_pairs(MfmaArgs a, double* u, int64_t ldu, int cnt, int nrhs, int tail_first) {
  const int q = blockIdx.z, tt = tail_first + (int)blockIdx.y;
  const int f = q + 8 * (tt / nrhs), r = tt % nrhs;
  if (f >= cnt) return;
  const CliqueDesc d = a.t.cl[a.t.lev[f]];
  const int nn = d.nn, na = d.na, nf = nn + na;
  double* P = u + (int64_t)r * ldu + d.blk;
  double* U = a.t.upd + (int64_t)r * a.t.updlen + d.upd;
  const int stride = gridDim.x * blockDim.x, g = blockIdx.x * blockDim.x + threadIdx.x;
  for (int e = g; e < nf * nn; e += stride) P[e] = 0.0;
  for (int e = g; e < na * na; e += stride) U[e] = 0.0;
}
__global__ void __launch_bounds__(1024) k_lf_assemble_lds(MfmaArgs a, double* u, int64_t ldu, int sgn) {
  extern __shared__ __attribute__((aligned(16))) double T[];
  lf_alds_task(a, u, ldu, sgn, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.z, T);
}
// The same with the tasks dealt out dynamically: a grid of one workgroup per CU draws (front, right-hand side, share)
// triples from a device counter until they are used up.  A plain launch of 800 workgroups on 256 CUs (synth50k: 8
// fronts x 100 right-hand sides, one workgroup per CU because of the 148 KB front) runs 3.125 rounds, i.e. four, the
// last one on an eighth of the chip; here every CU stays busy until the last task is taken.  Every workgroup leaves the
// loop as soon as the counter passes the number of tasks: no workgroup waits for another.
__global__ void __launch_bounds__(1024) k_lf_assemble_lds_dyn(MfmaArgs a, double* u, int64_t ldu, int sgn, int cnt, int nrhs, int nz, int* counter) {
  extern __shared__ __attribute__((aligned(16))) double T[];
  __shared__ int stask;
  const int total = cnt * nrhs * nz;
  for (;;) {
    __syncthreads();                                  // the previous task's write-out has read the front
    if (threadIdx.x == 0) stask = atomicAdd(counter, 1);
    __syncthreads();
    const int t = stask;
    if (t >= total) break;
    lf_alds_task(a, u, ldu, sgn, t % cnt, (t / cnt) % nrhs, t / (cnt * nrhs), nz, T);
  }
}
// Tiled extend-add: one workgroup owns an LF_TR x LF_TW tile of the front of (clique, rhs) in LDS and streams the
// children's packed update matrices through it CHILD-MAJOR -- every child column is a contiguous segment, so the
// reads are coalesced and each fetched line is used entirely (the gather plan above reads one 8-byte word per
// line).  The relative indices of a child are ascending, hence the part of the child that lands in the tile is a
// contiguous range of its rows and of its columns, found with two ballots per 64 indices.  Waves take children
// round-robin; collisions between children inside the tile are LDS atomic adds.  Same sgn convention as above.
constexpr int LF_TR = 256, LF_TW = 16;
__global__ void __launch_bounds__(256) k_lf_assemble_tiled(MfmaArgs a, double* u, int64_t ldu, int sgn) {
  __shared__ double T[LF_TR * LF_TW];
  const int k = a.t.lev[blockIdx.y];
  const CliqueDesc d = a.t.cl[k];
  if (d.chend == d.chbeg) return;
  const int r = blockIdx.z;
  const int nn = d.nn, na = d.na, nf = nn + na;
  const int ncb = (a.nnmax + a.namax + LF_TW - 1) / LF_TW;
  const int cb = blockIdx.x % ncb, rbk = blockIdx.x / ncb;
  const int c0 = cb * LF_TW, r0 = rbk * LF_TR;
  if (c0 >= nf || r0 >= nf || r0 + LF_TR <= c0) return;           // outside the front or strictly above the diagonal
  const int c1 = min(c0 + LF_TW, nf), r1 = min(r0 + LF_TR, nf);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int e = threadIdx.x; e < LF_TR * LF_TW; e += blockDim.x) T[e] = 0.0;
  __syncthreads();
  const double* ubase = a.t.updp + (int64_t)r * a.t.updplen;
  for (int q = d.chbeg + wave; q < d.chend; q += nw) {
    const CliqueDesc c = a.t.cl[a.t.chidx[q]];
    const int nac = c.na;
    const int32_t* rel = a.t.relidx + c.rel;
    const double* Uc = ubase + c.updp;
    // [j0, j1): child columns landing in [c0, c1);  [i0, i1): child rows landing in [r0, r1)
    int j0 = 0, j1 = 0, i0 = 0, i1 = 0;
    for (int b = 0; b < nac; b += 64) {
      const int v = (b + lane < nac) ? rel[b + lane] : 0x7fffffff;
      j0 += __popcll(__ballot(v < c0));
      j1 += __popcll(__ballot(v < c1));
      i0 += __popcll(__ballot(v < r0));
      i1 += __popcll(__ballot(v < r1));
    }
    for (int j = j0; j < j1; ++j) {
      const int cj = rel[j] - c0;
      const double* col = Uc + pk_col(j, nac) - j;               // col[i] = U_c(i, j), i >= j
      for (int i = max(i0, j) + lane; i < i1; i += 64) unsafeAtomicAdd(&T[(rel[i] - r0) + cj * LF_TR], col[i]);
    }
  }
  __syncthreads();
  double* P = u + (int64_t)r * ldu + d.blk;
  double* U = a.t.upd + (int64_t)r * a.t.updlen + d.upd;
  const int nr = r1 - r0;
  for (int e = threadIdx.x; e < nr * (c1 - c0); e += blockDim.x) {
    const int i = r0 + e % nr, j = c0 + e / nr;
    if (i < j) continue;
    const double v = T[(i - r0) + (j - c0) * LF_TR];
    if (j < nn) {
      if (v != 0.0) P[i + (int64_t)j * nf] += (sgn == 1) ? -v : v;
    } else if (sgn) {
      if (v != 0.0) U[(i - nn) + (int64_t)(j - nn) * na] += v;
    } else {
      U[(i - nn) + (int64_t)(j - nn) * na] = v;                    // full assignment: no prior clear needed
    }
  }
}
// packed update slots of the children marked in MfmaArgs::chskip <- 0 (for the extend-add routes that read every slot)
__global__ void k_lf_zero_skipped(MfmaArgs a) {
  const CliqueDesc d = a.t.cl[a.t.lev[blockIdx.x]];
  const int r = blockIdx.y;
  for (int q = d.chbeg; q < d.chend; ++q) {
    const int ck = a.t.chidx[q];
    if (!a.chskip[ck]) continue;
    const CliqueDesc c = a.t.cl[ck];
    double* UP = a.t.updp + (int64_t)r * a.t.updplen + c.updp;
    for (int e = threadIdx.x; e < c.na * (c.na + 1) / 2; e += blockDim.x) UP[e] = 0.0;
  }
}
__global__ void k_lf_clear_upd(MfmaArgs a) {
  const int k = a.t.lev[blockIdx.y];
  const CliqueDesc d = a.t.cl[k];
  double* U = a.t.upd + (int64_t)blockIdx.z * a.t.updlen + d.upd;
  const int64_t len = (int64_t)d.na * d.na;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += (int64_t)gridDim.x * blockDim.x) U[e] = 0.0;
}

// G_NN = Li F_NN Li^T through Z = Li Fl (F_NN = Fl + Fl^T) for wide fronts, through T = Li F_NN otherwise: in tile products
// of 64 the two routes cost sum_t (t+1) nt + sum (nt-t)(t+1) against 3 sum (nt-t)(t+1) -- equal at four column tiles, 12 % less
// at eight, 25 % less in the limit
__host__ __device__ inline bool lf_sym_split(int nn) { return nn > 6 * LT; }
// ---- up-sweep phase 1: E = F_AN - K F_NN / 2, X = F_AN - K F_NN (in place) ; Z = Li Fl (into T)
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_up1(MfmaArgs a, double* u, int64_t ldu) {
  __shared__ __attribute__((aligned(16))) double smem[PD == 1 ? LRC_DOUBLES : LKC * LSA + LT * LSB];
  double* const sA = smem; double* const sB = smem + LKC * LSA;
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int nE = mtA * ntN, nT = ntN * ntN;
  const int t = blockIdx.x;
  if (t >= nE + nT) return;
  const double* P = c.P;
  auto fsym = [=](int kk, int n) { return P[max(kk, n) + (int64_t)min(kk, n) * nf]; };
  d4 acc[2][2];
  tile64_zero(acc);
  if (t < nE) {
    const int m0 = (t % mtA) * LT, n0 = (t / mtA) * LT;
    const double* K = c.K;
    gemm_tile64<PD>(acc, na, nn, nn, m0, n0, [=](int m, int kk) { return K[m + (int64_t)kk * nf]; }, fsym, sA, sB);
    double* E = c.E; double* Pw = c.P;
    tile64_rmw(acc, m0, n0, na, nn, [=](int m, int n) { return Pw[nn + m + (int64_t)n * nf]; },
               [=](int m, int n, double v, double f) {
                 E[m + (int64_t)n * na] = f - 0.5 * v;
                 Pw[nn + m + (int64_t)n * nf] = f - v;
               });
  } else {
    // Z = Li Fl with Fl = the lower triangle of F_NN, its diagonal halved (F_NN = Fl + Fl^T): a product of two lower
    // triangular matrices -- only the lower tiles, k from the tile's first column to its last row (nn^3 / 3 flops where
    // T = Li F_NN took nn^3); phase 2 forms G_NN = Z Li^T + Li Z^T.  Tiles above the diagonal are never read.
    const int tt = t - nE, m0 = (tt % ntN) * LT, n0 = (tt / ntN) * LT;
    const double* Li = c.Li;
    // (tiles above the diagonal are never read in either form: phase 2 forms the lower tiles (tm, tn) of G_NN from T(tm, k <= tn) --
    // the root of synth50k, four tile rows: ten products instead of sixteen)
    if (n0 > m0) return;
    if (lf_sym_split(nn)) {
      // (the mask only matters over the tile's own columns; the rows of Fl below them are whole: the plain product)
      gemm_tile64<PD>(acc, nn, nn, min(nn, n0 + LT), m0, n0, [=](int m, int kk) { return Li[m + (int64_t)kk * nf]; },
                  [=](int kk, int n) { const double v_ = ldm<PD>(kk >= n, &P[kk + (int64_t)n * nf]); return kk == n ? 0.5 * v_ : v_; },
                  sA, sB, n0);
      gemm_tile64_plain<PD, false, true>(acc, Li, nf, nn, P, nf, nn, n0 + LT, min(nn, m0 + LT), m0, n0, smem);
    } else {
      gemm_tile64<PD>(acc, nn, nn, min(nn, m0 + LT), m0, n0, [=](int m, int kk) { return Li[m + (int64_t)kk * nf]; }, fsym, sA, sB);   // Li(m, k) = 0 for k > m
    }
    double* T = c.T;
    tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { T[m + (int64_t)n * nn] = v; });
  }
}
// ---- up-sweep phase 1 of a wide childless front WITHOUT separator whose right-hand side is a sparse constraint A_j (entry
// list of the clique, ascending in panel position: the entries of a column are one run):
//   Z[:, c] = sum over the entries (r, c), r >= c, of column c:  w Li[:, r]      (w halved on the diagonal; rows >= r)
// i.e. 2 nnz(A_j) n / 3 multiply-adds instead of the n^3 / 3 of the product of two dense triangles -- 3.4e8 against 2.3e10
// per constraint on the 4096 clique of config 2 at 0.5 % density -- and no dense input panel.  One workgroup per column of Z:
// the column is accumulated in LDS (row i belongs to thread (i - r0) mod 256 throughout: no synchronisation between entries),
// every read of Li is a contiguous run of one of its columns.  Rows r0 .. c - 1 of the column (r0 = the first row of its
// diagonal tile) are written as zeros: phase 2 reads whole lower tiles of Z.
constexpr int LF_ZSP_MAXNN = 7168;        // the column buffer: 56 KB of LDS
__global__ void __launch_bounds__(256) k_lf_zsp(MfmaArgs a, double* u, int64_t ldu) {
  extern __shared__ __attribute__((aligned(16))) double zc[];
  __shared__ int s_row[64];
  __shared__ double s_w[64];
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, nf = c.nf;
  const int col = blockIdx.x;
  if (col >= nn) return;
  const int tid = threadIdx.x;
  const int r0 = (col / LT) * LT;
  const int r = blockIdx.z;
  const int j = a.kc_ids ? a.kc_ids[a.kc_j0 + r] : a.kc_j0 + r;
  const int32_t* kp = a.kc_ptr + (int64_t)c.k * a.kc_stride;
  const int e0 = kp[j], e1 = kp[j + 1];
  // the run of column `col`: first entry at or after col * nf, first at or after (col + 1) * nf   (uniform binary searches)
  auto lower = [&](int64_t key) {
    int lo = e0, hi = e1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int64_t)a.kc_off[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
  };
  const int b0 = lower((int64_t)col * nf), b1 = lower((int64_t)(col + 1) * nf);
  for (int i = r0 + tid; i < nn; i += 256) zc[i] = 0.0;
  const double* Li = c.Li;
  for (int base = b0; base < b1; base += 64) {
    __syncthreads();
    if (tid < 64 && base + tid < b1) {
      const int off = a.kc_off[base + tid];
      const int row = off - col * nf;                   // < nn: the front has no separator rows
      s_row[tid] = row;
      s_w[tid] = row == col ? 0.5 * a.kc_val[base + tid] : a.kc_val[base + tid];
    }
    __syncthreads();
    const int ne = min(64, b1 - base);
    for (int q = 0; q < ne; ++q) {
      const int row = s_row[q];
      const double w = s_w[q];
      const double* Lc = Li + (int64_t)row * nf;
      // first row i >= row owned by this thread: i = r0 + tid (mod 256)
      int i = r0 + tid;
      if (i < row) i += ((row - i + 255) >> 8) << 8;
      for (; i + 768 < nn; i += 1024) {
        const double v0 = Lc[i], v1 = Lc[i + 256], v2 = Lc[i + 512], v3 = Lc[i + 768];
        zc[i] += w * v0; zc[i + 256] += w * v1; zc[i + 512] += w * v2; zc[i + 768] += w * v3;
      }
      for (; i < nn; i += 256) zc[i] += w * Lc[i];
    }
  }
  double* T = c.T + (int64_t)col * nn;
  for (int i = r0 + tid; i < nn; i += 256) T[i] = zc[i];
}
// ---- up-sweep phase 2: U -= K E^T + E K^T (lower tiles) ; G = X Li^T ; G_NN = Z Li^T + Li Z^T (lower, in place)
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_up2(MfmaArgs a, double* u, int64_t ldu) {
  __shared__ __attribute__((aligned(16))) double smem[PD == 1 ? LRC_DOUBLES : LKC * LSA + LT * LSB];
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int nU = mtA * (mtA + 1) / 2, nG = mtA * ntN, nN = ntN * (ntN + 1) / 2;
  const int t = blockIdx.x;
  if (t >= nU + nG + nN) return;
  const double* K = c.K; const double* E = c.E; const double* Li = c.Li; const double* T = c.T; const double* P = c.P;
  d4 acc[2][2];
  tile64_zero(acc);
  if (t < nU) {
    int tm, tn;
    lower_pair(t, tm, tn);
    const int m0 = tm * LT, n0 = tn * LT;
    gemm_tile64_plain<PD>(acc, K, nf, na, E, na, na, 0, nn, m0, n0, smem);     // K E^T
    gemm_tile64_plain<PD>(acc, E, na, na, K, nf, na, 0, nn, m0, n0, smem);     // + E K^T
    // The parent takes the update from the packed exchange buffer only; the square block is read (what the extend-add
    // assembled) but never written back, and a childless front does not even read it: its assembled block is zero
    // and the host skips clearing / assembling it (lf_up).  For config 3 (1999 childless (64,128) fronts x 100
    // right-hand sides) that is 26 GB of reads + 26 GB of writes + 26 GB of clears per sweep less.
    const double* U = c.U; double* UP = c.UP;
    if (c.hasch)
      tile64_rmw(acc, m0, n0, na, na, [=](int m, int n) { return U[m + (int64_t)n * na]; },
                 [=](int m, int n, double v, double uo) { if (m >= n) UP[pk_idx(m, n, na)] = uo - v; });
    else
      tile64_foreach(acc, m0, n0, na, na, [=](int m, int n, double v) {
        if (m >= n) UP[pk_idx(m, n, na)] = -v;
      });
  } else if (t < nU + nG) {
    const int tt = nG - 1 - (t - nU), m0 = (tt % mtA) * LT, n0 = (tt / mtA) * LT;      // (widest inner ranges first)
    gemm_tile64_plain<PD>(acc, P + nn, nf, na, Li, nf, nn, 0, min(nn, n0 + LT), m0, n0, smem);      // X Li^T; Li(n, k) = 0 for k > n
    double* G = c.G;
    tile64_foreach(acc, m0, n0, na, nn, [=](int m, int n, double v) { G[m + (int64_t)n * na] = v; });
  } else {
    int tm, tn;
    lower_pair_wide_first(t - nU - nG, ntN, tm, tn);
    const int m0 = tm * LT, n0 = tn * LT;
    gemm_tile64_plain<PD>(acc, T, nn, nn, Li, nf, nn, 0, min(nn, n0 + LT), m0, n0, smem);           // Z Li^T  (T holds Z, phase 1)
    if (lf_sym_split(nn))
      gemm_tile64_plain<PD>(acc, Li, nf, nn, T, nn, nn, 0, min(nn, n0 + LT), m0, n0, smem);         // + Li Z^T
    double* Pw = c.P;
    tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { if (m >= n) Pw[m + (int64_t)n * nf] = v; });
  }
}
// ---- up-sweep phase 3: Q = Ysc G into the AN rows of the panel (X is dead)
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_up3(MfmaArgs a, double* u, int64_t ldu) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
#ifdef SMCP_STAMPS
  const bool stamp = a.dbg && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && a.nrhs == 1;
  unsigned long long ts0 = stamp ? wall_clock64() : 0, ts1 = 0, ts2 = 0;
#endif
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int t = blockIdx.x;
  if (t >= mtA * ntN) return;
  const int m0 = (t % mtA) * LT, n0 = (t / mtA) * LT;
  const double* G = c.G; const double* Y = c.Ys;
  double* Pw = c.P;
  const int ymode = a.ymode;
  d4 acc[2][2];
  tile64_zero(acc);
  if (ymode) {
    // ymode 2: R^T(m, k) = 0 for k < m; any other mode but 1 (symmetric Y_AA): R(m, k) = 0 for k > m
#ifdef SMCP_STAMPS
    if (stamp) ts1 = wall_clock64();
#endif
    gemm_tile64<PD>(acc, na, nn, (ymode == 1 || ymode == 2) ? na : min(na, m0 + LT), m0, n0,
                [=](int m, int kk) { return yacc<PD>(Y, na, ymode, m, kk); },
                [=](int kk, int n) { return G[kk + (int64_t)n * na]; }, sA, sB, ymode == 2 ? m0 : 0);
#ifdef SMCP_STAMPS
    if (stamp) ts2 = wall_clock64();
#endif
    tile64_foreach(acc, m0, n0, na, nn, [=](int m, int n, double v) { Pw[nn + m + (int64_t)n * nf] = v; });
#ifdef SMCP_STAMPS
    if (stamp) {
      __builtin_amdgcn_s_waitcnt(0);
      const unsigned long long ts3 = wall_clock64();
      atomicAdd(a.dbg + 26, ts1 - ts0); atomicAdd(a.dbg + 27, ts2 - ts1); atomicAdd(a.dbg + 28, ts3 - ts2); atomicAdd(a.dbg + 29, 1ull);
    }
#endif
  } else {
    tile64_rmw(acc, m0, n0, na, nn, [=](int m, int n) { return G[m + (int64_t)n * na]; },
               [=](int m, int n, double, double g) { Pw[nn + m + (int64_t)n * nf] = g; });
  }
}

// ---- down-sweep phase 1: QL = Q Li (into E) ; T = G_NN Li        (Q = AN rows of the panel)
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_down1(MfmaArgs a, double* u, int64_t ldu) {
  __shared__ __attribute__((aligned(16))) double smem[PD == 1 ? LRC_DOUBLES : LKC * LSA + LT * LSB];
  double* const sA = smem; double* const sB = smem + LKC * LSA;
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int nE = mtA * ntN, nT = ntN * ntN;
  const int t = blockIdx.x;
  if (t >= nE + nT) return;
  const double* P = c.P; const double* Li = c.Li;
  auto li = [=](int kk, int n) { return Li[kk + (int64_t)n * nf]; };
  d4 acc[2][2];
  tile64_zero(acc);
  if (t < nE) {
    const int m0 = (t % mtA) * LT, n0 = (t / mtA) * LT;
    gemm_tile64_plain<PD, false, true>(acc, P + nn, nf, na, Li, nf, nn, n0, nn, m0, n0, smem);   // Li(k, n) = 0 for k < n
    double* E = c.E;
    tile64_foreach(acc, m0, n0, na, nn, [=](int m, int n, double v) { E[m + (int64_t)n * na] = v; });
  } else {
    const int tt = t - nE, m0 = (tt % ntN) * LT, n0 = (tt / ntN) * LT;
    if (lf_sym_split(nn)) {
      // wide fronts: Z' = Gl Li with Gl = the lower triangle of G_NN, diagonal halved (lower tiles only, k from the tile's
      // first column to its last row); phase 3 forms Li^T G_NN Li = Li^T Z' + Z'^T Li  (see k_lf_up1)
      if (n0 > m0) return;
      // (columns before the tile's first row are whole columns of Gl: the plain product; the mask only matters from m0 on)
      gemm_tile64_plain<PD, false, true>(acc, P, nf, nn, Li, nf, nn, n0, m0, m0, n0, smem);
      gemm_tile64<PD>(acc, nn, nn, min(nn, m0 + LT), m0, n0,
                  [=](int m, int kk) { const double v_ = ldm<PD>(m >= kk, &P[m + (int64_t)kk * nf]); return m == kk ? 0.5 * v_ : v_; }, li, sA, sB, m0);
    } else {
      gemm_tile64<PD>(acc, nn, nn, nn, m0, n0,
                  [=](int m, int kk) { return P[max(m, kk) + (int64_t)min(m, kk) * nf]; }, li, sA, sB, n0);
    }
    double* T = c.T;
    tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { T[m + (int64_t)n * nn] = v; });
  }
}
// ---- down-sweep phase 2: D = QL - Z_AA K / 2 (into G) ; Z_AN = 2D - QL (into the panel)
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_down2(MfmaArgs a, double* u, int64_t ldu) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int t = blockIdx.x;
  if (t >= mtA * ntN) return;
  const int m0 = (t % mtA) * LT, n0 = (t / mtA) * LT;
  const double* Z = c.U; const double* K = c.K; const double* E = c.E;
  d4 acc[2][2];
  tile64_zero(acc);
  gemm_tile64<PD>(acc, na, nn, na, m0, n0,
              [=](int m, int kk) { return Z[max(m, kk) + (int64_t)min(m, kk) * na]; },
              [=](int kk, int n) { return K[kk + (int64_t)n * nf]; }, sA, sB);
  double* G = c.G; double* Pw = c.P;
  tile64_rmw(acc, m0, n0, na, nn, [=](int m, int n) { return E[m + (int64_t)n * na]; },
             [=](int m, int n, double v, double ql) {
               const double dd = ql - 0.5 * v;
               G[m + (int64_t)n * na] = dd;
               Pw[nn + m + (int64_t)n * nf] = 2.0 * dd - ql;
             });
}
// ---- down-sweep phase 3: Z_NN = Li^T T - K^T D - D^T K (lower tiles, into the panel)
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_down3(MfmaArgs a, double* u, int64_t ldu) {
  __shared__ __attribute__((aligned(16))) double smem[PD == 1 ? LRC_DOUBLES : LKC * LSA + LT * LSB];
  double* const sA = smem; double* const sB = smem + LKC * LSA;
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int ntN = tiles64(nn);
  const int t = blockIdx.x;
  if (t >= ntN * (ntN + 1) / 2) return;
  int tm, tn;
  lower_pair(t, tm, tn);
  const int m0 = tm * LT, n0 = tn * LT;
  const double* Li = c.Li; const double* K = c.K; const double* T = c.T; const double* D = c.G;
  d4 acc[2][2];
  tile64_zero(acc);
  gemm_tile64_plain<PD, true, true>(acc, Li, nf, nn, T, nn, nn, m0, nn, m0, n0, smem);            // Li^T T; Li(k, m) = 0 for k < m
  if (lf_sym_split(nn))
    gemm_tile64_plain<PD, true, true>(acc, T, nn, nn, Li, nf, nn, m0, nn, m0, n0, smem);          // + Z'^T Li (T holds Z')
  gemm_tile64<PD>(acc, nn, nn, na, m0, n0, [=](int m, int kk) { return -K[kk + (int64_t)m * nf]; },
              [=](int kk, int n) { return D[kk + (int64_t)n * na]; }, sA, sB);
  gemm_tile64<PD>(acc, nn, nn, na, m0, n0, [=](int m, int kk) { return -D[kk + (int64_t)m * na]; },
              [=](int kk, int n) { return K[kk + (int64_t)n * nf]; }, sA, sB);
  double* Pw = c.P;
  tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { if (m >= n) Pw[m + (int64_t)n * nf] = v; });
}

// ---- fronts WITHOUT separator (the roots of the forest) in hessian(adj = None): both sweeps together are
//     Z_NN = Li^T (Li F_NN Li^T) Li = Y_NN F_NN Y_NN,   Y_NN = Li^T Li = the root block of the projected inverse Y itself,
// i.e. two products with a matrix the caller already holds, in place of the four phase launches up1 / up2 / down1 / down3
// (the root of synth50k, one right-hand side: 58 us of the 345 us of a Hessian).  Phase 1 (end of the up sweep, after the
// extend-add): T = F_NN Y_NN; phase 2 (start of the down sweep): Z_NN = Y_NN T, lower, into the panel.
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_root1(MfmaArgs a, double* u, int64_t ldu, const double* Yb) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, nf = c.nf, ntN = tiles64(nn);
  const int t = blockIdx.x;
  if (t >= ntN * ntN) return;
  const int m0 = (t % ntN) * LT, n0 = (t / ntN) * LT;
  const double* P = c.P;
  const double* Y = Yb + a.t.cl[c.k].blk;
  d4 acc[2][2];
  tile64_zero(acc);
  gemm_tile64<PD>(acc, nn, nn, nn, m0, n0, [=](int m, int kk) { return P[max(m, kk) + (int64_t)min(m, kk) * nf]; },
                  [=](int kk, int n) { return Y[max(kk, n) + (int64_t)min(kk, n) * nf]; }, sA, sB);
  double* T = c.T;
  tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { T[m + (int64_t)n * nn] = v; });
}
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_root2(MfmaArgs a, double* u, int64_t ldu, const double* Yb) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, nf = c.nf, ntN = tiles64(nn);
  if ((int)blockIdx.x >= ntN * (ntN + 1) / 2) return;
  int tm, tn;
  lower_pair(blockIdx.x, tm, tn);
  const int m0 = tm * LT, n0 = tn * LT;
  const double* T = c.T;
  const double* Y = Yb + a.t.cl[c.k].blk;
  d4 acc[2][2];
  tile64_zero(acc);
  gemm_tile64<PD>(acc, nn, nn, nn, m0, n0, [=](int m, int kk) { return Y[max(m, kk) + (int64_t)min(m, kk) * nf]; },
                  [=](int kk, int n) { return T[kk + (int64_t)n * nn]; }, sA, sB);
  double* Pw = c.P;
  tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { if (m >= n) Pw[m + (int64_t)n * nf] = v; });
}

// ---- supernodal triangular solves with a dense n x nrhs right-hand side (chompack.trsm; the S^-1[:, K_j] columns of the
// SCMcolumn2 route, solvers.py:490-492) as tile products with the inverse-form factor LK = [Li; K]:
//   forward  (B <- L^-1 B), per clique:  x_N = Li b_N,  B[rows_A] -= K b_N        (L_AN x_N = K b_N)
//   backward (B <- L^-T B), per clique:  x_N = Li^T b_N - K^T x_A                (L_NN^-T L_AN^T = K^T)
// One workgroup per (64 front rows, clique, 64 columns of B); the cliques of a level are independent, the separator
// rows of different cliques of a level may coincide (global atomics in the forward sweep).  x_N goes to a scratch image
// X of B first: other tiles of the same clique still read b_N.
__global__ void __launch_bounds__(256, 4) k_trsm_mm_fwd(MfmaArgs a, double* B, int nrhs, int64_t ldb, const int32_t* rowidx, double* X) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const int k = a.t.lev[blockIdx.y];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const int m0 = blockIdx.x * LT, n0 = blockIdx.z * LT;
  if (m0 >= nf) return;
  const double* LK = a.LK + d.blk;
  const double* Bn = B + d.first;
  const int kend = (m0 + LT <= nn) ? min(nn, m0 + LT) : nn;     // Li(m, k) = 0 for k > m
  d4 acc[2][2];
  tile64_zero(acc);
  gemm_tile64(acc, nf, nrhs, kend, m0, n0, [=](int m, int kk) { return LK[m + (int64_t)kk * nf]; },
              [=](int kk, int n) { return Bn[kk + (int64_t)n * ldb]; }, sA, sB);
  const int32_t* rows = rowidx + d.rows;
  double* Xn = X + d.first;
  tile64_foreach(acc, m0, n0, nf, nrhs, [=](int m, int n, double v) {
    if (m < nn) Xn[m + (int64_t)n * ldb] = v;
    else if (v != 0.0) unsafeAtomicAdd(&B[rows[m] + (int64_t)n * ldb], -v);
  });
}
__global__ void k_trsm_mm_copy(MfmaArgs a, double* B, int nrhs, int64_t ldb, const double* X) {
  const CliqueDesc d = a.t.cl[a.t.lev[blockIdx.y]];
  const int nn = d.nn;
  const int64_t tot = (int64_t)nn * nrhs;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % nn, col = e / nn;
    B[d.first + i + col * ldb] = X[d.first + i + col * ldb];
  }
}
__global__ void __launch_bounds__(256, 4) k_trsm_mm_bwd(MfmaArgs a, double* B, int nrhs, int64_t ldb, const int32_t* rowidx, double* X) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const int k = a.t.lev[blockIdx.y];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const int m0 = blockIdx.x * LT, n0 = blockIdx.z * LT;
  if (m0 >= nn) return;
  const double* LK = a.LK + d.blk;
  const int32_t* rows = rowidx + d.rows;
  const double* Bc = B;
  const int first = d.first;
  d4 acc[2][2];
  tile64_zero(acc);
  // [Li^T | -K^T](m, kk) = +-LK[kk + m nf] (zero for kk < m: Li is stored with zeros above its diagonal)
  gemm_tile64(acc, nn, nrhs, nf, m0, n0,
              [=](int m, int kk) { const double v = LK[kk + (int64_t)m * nf]; return kk < nn ? v : -v; },
              [=](int kk, int n) { return Bc[(kk < nn ? first + kk : rows[kk]) + (int64_t)n * ldb]; }, sA, sB, m0);
  double* Xn = X + d.first;
  tile64_foreach(acc, m0, n0, nn, nrhs, [=](int m, int n, double v) { Xn[m + (int64_t)n * ldb] = v; });
}

// ---- projected inverse, large fronts: E = Y_AA K ; Y_NN = Li^T Li + K^T E (lower) ; Y_AN = -E
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_pinv1(MfmaArgs a, double* x) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const LfCtx c = lf_ctx(a, x, 0);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int t = blockIdx.x;
  if (t >= mtA * ntN) return;
  const int m0 = (t % mtA) * LT, n0 = (t / mtA) * LT;
  const double* Y = c.U; const double* K = c.K;
  d4 acc[2][2];
  tile64_zero(acc);
  gemm_tile64<PD>(acc, na, nn, na, m0, n0,
              [=](int m, int kk) { return Y[max(m, kk) + (int64_t)min(m, kk) * na]; },
              [=](int kk, int n) { return K[kk + (int64_t)n * nf]; }, sA, sB);
  double* E = c.E;
  tile64_foreach(acc, m0, n0, na, nn, [=](int m, int n, double v) { E[m + (int64_t)n * na] = v; });
}
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_pinv2(MfmaArgs a, double* x) {
  __shared__ __attribute__((aligned(16))) double smem[PD == 1 ? LRC_DOUBLES : LKC * LSA + LT * LSB];
  double* const sA = smem; double* const sB = smem + LKC * LSA;
  const LfCtx c = lf_ctx(a, x, 0);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int nN = ntN * (ntN + 1) / 2, nA = mtA * ntN;
  const int t = blockIdx.x;
  if (t >= nN + nA) return;
  double* Pw = c.P;
  const double* E = c.E;
  if (t < nN) {
    int tm, tn;
    lower_pair(t, tm, tn);
    const int m0 = tm * LT, n0 = tn * LT;
    const double* Li = c.Li; const double* K = c.K;
    d4 acc[2][2];
    tile64_zero(acc);
    gemm_tile64_plain<PD, true, true>(acc, Li, nf, nn, Li, nf, nn, max(m0, n0), nn, m0, n0, smem);     // Li^T Li (zeros above the diagonal are stored)
    gemm_tile64_plain<PD, true, true>(acc, K, nf, nn, E, na, nn, 0, na, m0, n0, smem);                 // + K^T E
    tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { if (m >= n) Pw[m + (int64_t)n * nf] = v; });
  } else {
    const int tt = t - nN, m0 = (tt % mtA) * LT, n0 = (tt / mtA) * LT;
    for (int e = threadIdx.x; e < LT * LT; e += blockDim.x) {
      const int m = m0 + (e & 63), n = n0 + (e >> 6);
      if (m < na && n < nn) Pw[nn + m + (int64_t)n * nf] = -E[m + (int64_t)n * na];
    }
  }
}

}  // namespace smcp

namespace smcp {

// ---------------------------------------------------------------------------------------------
// Blocked Cholesky / triangular inversion of large fronts: 64-wide block columns, one launch
// per step and phase (diag block in LDS by one workgroup per clique; panel and trailing updates
// as 64x64 MFMA tiles over the whole chip).
// ---------------------------------------------------------------------------------------------
constexpr int LB = 64;          // block-column width
constexpr int LBD = LB + 1;     // LDS leading dimension of the diagonal block

// one 16 x 16 output tile by the CALLING WAVE: store(m, n, sum_k A(m, k) B(k, n)), m, n < 16, k < Kd (a multiple of 4)
template <class FA, class FB, class FC>
__device__ inline void wave_tile16(int Kd, FA A, FB B, FC store) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < Kd; k0 += 8) {
    const double a0 = A(l15, k0 + kq), b0 = B(k0 + kq, l15);
    const bool two = k0 + 4 < Kd;
    const double a1 = two ? A(l15, k0 + 4 + kq) : 0.0, b1 = two ? B(k0 + 4 + kq, l15) : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc, 0, 0, 0);
    if (two) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) store(l15, kq + 4 * r, acc[r]);
}
// Inverse of a w x w (w <= 64) lower triangular block D in LDS (ld LBD; entries above the diagonal zero) -> Di (ld LBD, zeroed
// by the caller).  The four 16 x 16 diagonal blocks are inverted side by side, one wave each; the rest by recursive doubling,
// inv([A 0; B C]) = [Ai 0; -Ci B Ai, Ci]: at block size 16 the two pairs on two waves (the intermediate B Ai stays inside the
// wave), at block size 32 the four tiles of each product on four waves.  Four workgroup barriers in all, where the
// block-row scheme of potrf_inv64 runs four one-wave inversions and six tile products one after the other (k_lf_diag_inv on
// the root of synth50k: 36 us on the critical path of every factorisation).  s16: 1024 doubles of scratch.  Needs >= 4 waves.
// the off-diagonal part: Di holds the inverses of the 16 x 16 diagonal blocks (zeros elsewhere), D the triangle itself with
// zeros beyond w; two doubling levels, three workgroup barriers
template <int LBD = LB + 1>
__device__ inline void tri_inv64_offdiag(const double* D, int w, double* Di, double* s16) {
  const int wave = threadIdx.x >> 6;
  if (wave < 2 && 32 * wave + 16 < w) {                      // pairs (0, 1) and (2, 3) at block size 16
    const int r0 = 32 * wave;
    const double* Bm = D + (r0 + 16) + r0 * LBD;
    const double* Ai = Di + r0 + r0 * LBD;
    const double* Ci = Di + (r0 + 16) + (r0 + 16) * LBD;
    double* W = s16 + 256 * wave;
    wave_tile16(16, [=](int m, int k) { return Bm[m + k * LBD]; }, [=](int k, int n) { return Ai[k + n * LBD]; },
                [=](int m, int n, double v) { W[m + n * 16] = v; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double* X = Di + (r0 + 16) + r0 * LBD;
    wave_tile16(16, [=](int m, int k) { return Ci[m + k * LBD]; }, [=](int k, int n) { return W[k + n * 16]; },
                [=](int m, int n, double v) { X[m + n * LBD] = -v; });
  }
  __syncthreads();
  if (w > 32) {                                              // the pair of 32 x 32 blocks
    const double* Bm = D + 32;
    double* W = s16;                                         // 32 x 32, ld 32
    if (wave < 4) {
      const int tm = 16 * (wave & 1), tn = 16 * (wave >> 1);
      wave_tile16(32, [=](int m, int k) { return Bm[(tm + m) + k * LBD]; }, [=](int k, int n) { return Di[k + (tn + n) * LBD]; },
                  [=](int m, int n, double v) { W[(tm + m) + (tn + n) * 32] = v; });
    }
    __syncthreads();
    if (wave < 4) {
      const int tm = 16 * (wave & 1), tn = 16 * (wave >> 1);
      const double* Ci = Di + 32 + 32 * LBD;
      double* X = Di + 32;
      wave_tile16(32, [=](int m, int k) { return Ci[(tm + m) + k * LBD]; }, [=](int k, int n) { return W[k + (tn + n) * 32]; },
                  [=](int m, int n, double v) { X[(tm + m) + (tn + n) * LBD] = -v; });
    }
  }
  __syncthreads();
}
template <int LBD = LB + 1>
__device__ inline void tri_inv64_pad(double* D, int w) {     // rows / columns beyond w read as zero
  constexpr int NB = LBD - 1;
  for (int e = threadIdx.x; e < NB * NB; e += blockDim.x) {
    const int i = e % NB, j = e / NB;
    if (i >= w || j >= w) D[i + j * LBD] = 0.0;
  }
}
// (LBD: leading dimension of D and Di in LDS; LBD - 1 = 64 or, for blocks of at most 32 rows, 32)
template <int LBD = LB + 1>
__device__ __forceinline__ void tri_inv64_rd(double* D, int w, double* Di, double* s16) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  tri_inv64_pad<LBD>(D, w);
  __syncthreads();
  if (wave < 4 && 16 * wave < w) {                           // diagonal block `wave`
    const int b0 = 16 * wave, bw = min(16, w - b0);
    const double* Db = D + b0 + b0 * LBD;
    double a[16], x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = (lane < bw && j <= lane) ? Db[lane + j * LBD] : ((lane == j && lane < 16) ? 1.0 : 0.0);
    wave_tri_inv16(a, x, lane);
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < 16; ++j) if (lane < bw && j < bw) Di[(b0 + j) + (b0 + lane) * LBD] = x[j];
    }
  }
  __syncthreads();
  tri_inv64_offdiag<LBD>(D, w, Di, s16);
}

// Cholesky + inverse of a w x w (w <= 64) lower block D in LDS (ld LBD; EVERYTHING outside the lower triangle of the leading
// w x w block zero, up to 64 x 64) -> D = L, Di = L^-1 (ld LBD, zeros above the diagonal).  xs: 4 x 256 doubles (the inverses
// of the four 16 x 16 diagonal blocks), s16: 1024 doubles.  At least four waves.  Returns 0 or 1 (uniform).
// Round 5: the sixteen-column steps of potrf_inv64 below each ran potrf_inv16 on one wave with the others idle, then two
// generic wg_mma calls (rows below x the block's inverse; trailing update) and three barriers -- 26 us for a full block on
// four waves, 2/3 of a 64-column step of the blocked Cholesky whichever way its launches are organised.  Here a step has
// two barriers and its critical path is wave 0's alone: update of the next diagonal 16 x 16 tile (4 MFMAs), its Cholesky
// and inverse (wave_potrf_inv16, no barrier inside); the other waves meanwhile do the REST of the previous step's trailing
// update; after the barrier one wave per 16 x 16 tile scales the rows below (4 MFMAs each).
__device__ inline int potrf_inv64_fast(double* D, int w, double* Di, double* xs, double* s16) {
  __shared__ int fail64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  for (int e = tid; e < LB * LBD; e += blockDim.x) Di[e] = 0.0;
  if (tid == 0) fail64 = 0;
  const int nb = (w + 15) >> 4;
  // one 16 x 16 tile product on the calling wave: acc(m, n) = sum_k A[m + k lda] B[n + k ldb], k < 16
  auto tile_abt = [&](const double* A, int lda, const double* B, int ldb) {
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(B[l15 + (k0 + kq) * ldb], A[l15 + (k0 + kq) * lda], acc, 0, 0, 0);
    return acc;           // register r <-> (m = l15, n = kq + 4 r)
  };
  __syncthreads();
  for (int b = 0; b < nb; ++b) {
    const int bw = min(16, w - 16 * b);
    if (wave == 0) {
      double* Dbb = D + 16 * b + 16 * b * LBD;
      if (b > 0) {                      // the diagonal tile's share of the previous step's trailing update
        const double* P = D + 16 * b + 16 * (b - 1) * LBD;
        const d4 u = tile_abt(P, LBD, P, LBD);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int n = kq + 4 * r; if (l15 >= n) Dbb[l15 + n * LBD] -= u[r]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      const bool ok = wave_potrf_inv16(Dbb, LBD, bw, xs + 256 * b);
      if (!ok && lane == 0) fail64 = 1;
    } else if (b > 0 && wave < 4) {     // the other tiles (r, c), b <= c <= r < nb, (r, c) != (b, b), dealt over waves 1 .. 3
      int t = 0;
      for (int c = b; c < nb; ++c)
        for (int r = c; r < nb; ++r) {
          if (r == b && c == b) continue;
          if (t++ % 3 != wave - 1) continue;
          const double* Pr = D + 16 * r + 16 * (b - 1) * LBD;
          const double* Pc = D + 16 * c + 16 * (b - 1) * LBD;
          const d4 u = tile_abt(Pr, LBD, Pc, LBD);
          double* T = D + 16 * r + 16 * c * LBD;
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int n = kq + 4 * q; if (r > c || l15 >= n) T[l15 + n * LBD] -= u[q]; }
        }
    }
    __syncthreads();
    if (fail64) return 1;
    // rows below the block x its inverse (transposed), one wave per 16 x 16 tile; the block's inverse to its place in Di
    if (wave < 4) {
      const double* X = xs + 256 * b;
      for (int r = b + 1 + wave; r < nb; r += 4) {
        double* P = D + 16 * r + 16 * b * LBD;
        const d4 u = tile_abt(P, LBD, X, 16);          // (P X^T)(m, n) = sum_k P(m, k) X(n, k)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) P[l15 + (kq + 4 * q) * LBD] = u[q];
      }
      if (wave == min(3, nb - b - 1)) {                // the first wave without a tile (or the last one) copies the inverse block
        for (int e = lane; e < 256; e += 64) { const int i = e & 15, j = e >> 4; if (i >= j) Di[(16 * b + i) + (16 * b + j) * LBD] = X[i + j * 16]; }
      }
    }
    __syncthreads();
  }
  tri_inv64_offdiag(D, w, Di, s16);
  return 0;
}

// In-LDS Cholesky (do_potrf) and inverse of a w x w (w <= 64) lower block D (ld LBD); the inverse goes
// to Di (ld LBD, zeros above the diagonal).  d16: 256 doubles, s16: 16 x 64 doubles of scratch.
// Returns 0 or the 1-based failing pivot (uniform).
__device__ inline int potrf_inv64(double* D, int w, double* Di, double* d16, double* s16, bool do_potrf) {
  if (do_potrf && blockDim.x >= 256) {        // the four-wave routine above (s16 serves as its block-inverse buffer as well: disjoint uses)
    tri_inv64_pad(D, w);
    __syncthreads();
    return potrf_inv64_fast(D, w, Di, s16, s16);
  }
  for (int e = threadIdx.x; e < LB * LBD; e += blockDim.x) Di[e] = 0.0;
  __syncthreads();
  const bool doubling = blockDim.x >= 256;
  if (!do_potrf && doubling) { tri_inv64_rd(D, w, Di, s16); return 0; }
  if (doubling) tri_inv64_pad(D, w);          // (the potrf steps below begin with a barrier)
  for (int jb = 0; jb < w; jb += 16) {
    const int bw = min(16, w - jb);
    if (do_potrf) {
      int f = potrf_inv16(D + jb + jb * LBD, LBD, bw, d16);
      if (f) return jb + f;
      const int mrem = w - jb - bw;
      if (mrem > 0) {
        double* Pj = D + (jb + bw) + jb * LBD;
        wg_mma(mrem, bw, bw, [=](int m, int kk) { return Pj[m + kk * LBD]; },
               [=](int kk, int n) { return d16[n + kk * 16]; },
               [=](int m, int n, double acc) { Pj[m + n * LBD] = acc; });
        __syncthreads();
        double* Tr = D + (jb + bw) + (jb + bw) * LBD;
        wg_mma(mrem, mrem, bw, [=](int m, int kk) { return Pj[m + kk * LBD]; },
               [=](int kk, int n) { return Pj[n + kk * LBD]; },
               [=](int m, int n, double acc) { if (m >= n) Tr[m + n * LBD] -= acc; }, true);
        __syncthreads();
      }
    } else {
      tri_inv16(D + jb + jb * LBD, LBD, bw, d16);
    }
    // block row jb of the inverse: diagonal block, then -Dinv16 * (D[jb, 0:jb] * Di[0:jb, 0:jb])
    for (int e = threadIdx.x; e < bw * bw; e += blockDim.x) {
      int i = e % bw, j = e / bw;
      if (i >= j) Di[(jb + i) + (jb + j) * LBD] = d16[i + j * 16];
    }
    __syncthreads();
    if (jb > 0 && !doubling) {
      wg_mma(bw, jb, jb, [=](int m, int kk) { return D[(jb + m) + kk * LBD]; },
             [=](int kk, int n) { return Di[kk + n * LBD]; },
             [=](int m, int n, double acc) { s16[m + n * 16] = acc; });
      __syncthreads();
      wg_mma(bw, jb, bw, [=](int m, int kk) { return d16[m + kk * 16]; },
             [=](int kk, int n) { return s16[kk + n * 16]; },
             [=](int m, int n, double acc) { Di[(jb + m) + n * LBD] = -acc; });
      __syncthreads();
    }
  }
  // the rest of the inverse by recursive doubling from the diagonal blocks' inverses (three barriers instead of nine)
  if (doubling) tri_inv64_offdiag(D, w, Di, s16);
  return 0;
}

// A square or panel matrix view processed by the blocked kernels: M (rows) x N (cols) at ptr (ld), optional
// trailing block `upd` (na x na, ld na) that receives -P_A P_A^T (Cholesky of a front), scratch for Dinv.
struct LfMat {
  double* A; int64_t ld; int nrow, ncol;   // factor the leading ncol x ncol block, nrow >= ncol
  double* upd; int na;                     // rows ncol..nrow-1 update this (na = nrow - ncol) or null
  double* dinv;                            // w x w inverse of the current diagonal block (ld = w)
};
// mode 0: panel of a front in x (Cholesky), 1: L -> LK preparation (triangular inverse), 2: Y_AA -> its Cholesky factor,
// 3: the nn x nn matrix at the head of the clique's scratch (completion), 4: update-layout matrix in aux, inverse only,
// 5: one plain dense matrix x (order a.dn, leading dimension a.dld; a.lfd = its 64 x 64 slot; the clique is ignored)
__device__ inline LfMat lf_mat(const MfmaArgs& a, int k, int mode, double* x, double* aux) {
  const CliqueDesc d = a.t.cl[k];
  LfMat m;
  double* scratch = a.lfd + (int64_t)d.pad * (LB * LB);   // d.pad = slot of this clique among the large fronts
  if (mode == 5) {
    m.A = x; m.ld = a.dld; m.nrow = a.dn; m.ncol = a.dn; m.upd = nullptr; m.na = 0;
    scratch = a.lfd;
  } else if (mode == 3) {
    m.A = a.t.tmp + a.t.tmpptr[k]; m.ld = d.nn; m.nrow = d.nn; m.ncol = d.nn; m.upd = nullptr; m.na = 0;
  } else if (mode == 2 || mode == 4) {
    m.A = aux + d.upd; m.ld = d.na; m.nrow = d.na; m.ncol = d.na; m.upd = nullptr; m.na = 0;
  } else {
    m.A = x + d.blk; m.ld = d.nn + d.na; m.nrow = d.nn + d.na; m.ncol = d.nn;
    m.upd = (mode == 0) ? a.t.upd + d.upd : nullptr; m.na = d.na;
  }
  m.dinv = scratch;
  return m;
}

// diag step: factor (or only invert) the diagonal block at column jb; L block written back, inverse to scratch
__global__ void __launch_bounds__(1024) k_lf_diag(MfmaArgs a, double* x, double* aux, int mode, int jb, int do_potrf) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const D = smem;
  double* const Di = D + LB * LBD;
  double* const d16 = Di + LB * LBD;
  double* const s16 = d16 + 256;
  const int k = a.t.lev[blockIdx.x];
  if (*info_of(a.t, k)) return;
  const LfMat M = lf_mat(a, k, mode, x, aux);
  if (jb >= M.ncol) return;
  const int w = min(LB, M.ncol - jb);
  double* Ab = M.A + jb + (int64_t)jb * M.ld;
  for (int e = threadIdx.x; e < w * w; e += blockDim.x) {
    int i = e % w, j = e / w;
    D[i + j * LBD] = (i >= j) ? Ab[i + (int64_t)j * M.ld] : 0.0;
  }
  __syncthreads();
  int f = potrf_inv64(D, w, Di, d16, s16, do_potrf != 0);
  if (f) { if (threadIdx.x == 0) atomicCAS(info_of(a.t, k), 0, info_val(a.t, k)); return; }
  for (int e = threadIdx.x; e < w * w; e += blockDim.x) {
    int i = e % w, j = e / w;
    if (do_potrf && i >= j) Ab[i + (int64_t)j * M.ld] = D[i + j * LBD];
    M.dinv[i + j * w] = Di[i + j * LBD];
  }
}
// panel step: rows below the diagonal block <- rows * Dinv^T   (one 64x64 tile per workgroup, in place)
__global__ void __launch_bounds__(256) k_lf_chol_panel(MfmaArgs a, double* x, double* aux, int mode, int jb) {
  __shared__ __attribute__((aligned(16))) double smem[LRC_DOUBLES];
  const int k = a.t.lev[blockIdx.y];
  if (*info_of(a.t, k)) return;
  const LfMat M = lf_mat(a, k, mode, x, aux);
  if (jb >= M.ncol) return;
  const int w = min(LB, M.ncol - jb);
  const int mrem = M.nrow - jb - w;
  const int m0 = blockIdx.x * LT;
  if (m0 >= mrem) return;
  double* Pj = M.A + (jb + w) + (int64_t)jb * M.ld;
  const double* Di = M.dinv;
  const int64_t ld = M.ld;
  d4 acc[2][2];
  tile64_zero(acc);
  gemm_tile64_plain<1>(acc, Pj, ld, mrem, Di, w, w, 0, w, m0, 0, smem);
  __syncthreads();   // every thread has consumed its rows before they are overwritten
  tile64_foreach(acc, m0, 0, mrem, w, [=](int m, int n, double v) { Pj[m + n * ld] = v; });
}
// trailing step: remaining columns of the panel and the update block -= P P^T (lower tiles)
__global__ void __launch_bounds__(256) k_lf_chol_trail(MfmaArgs a, double* x, double* aux, int mode, int jb) {
  __shared__ __attribute__((aligned(16))) double smem[LRC_DOUBLES];
  const int k = a.t.lev[blockIdx.y];
  if (*info_of(a.t, k)) return;
  const LfMat M = lf_mat(a, k, mode, x, aux);
  if (jb >= M.ncol) return;
  const int w = min(LB, M.ncol - jb);
  const int mrem = M.nrow - jb - w;       // rows below the diagonal block
  const int ncr = M.ncol - jb - w;        // remaining columns of the factored part
  const int mt = tiles64(mrem), nt = tiles64(ncr);
  // tasks: (a) tiles (tm, tn) with tn < nt, tm >= tn of the in-panel trailing block; (b) lower tiles of upd
  const int mtA = tiles64(M.na);
  const int nP = (ncr > 0) ? (nt * (nt + 1) / 2 + (mt - nt) * nt) : 0;
  const int nU = M.upd ? mtA * (mtA + 1) / 2 : 0;
  const int t = blockIdx.x;
  if (t >= nP + nU) return;
  const double* Pj = M.A + (jb + w) + (int64_t)jb * M.ld;
  const int64_t ld = M.ld;
  d4 acc[2][2];
  tile64_zero(acc);
  if (t < nP) {
    int tm, tn;
    const int ntri = nt * (nt + 1) / 2;
    if (t < ntri) lower_pair(t, tm, tn);
    else { const int tt = t - ntri; tm = nt + tt / nt; tn = tt % nt; }
    const int m0 = tm * LT, n0 = tn * LT;
    gemm_tile64_plain<1>(acc, Pj, ld, mrem, Pj, ld, ncr, 0, w, m0, n0, smem);
    double* Tr = M.A + (jb + w) + (int64_t)(jb + w) * M.ld;
    tile64_rmw(acc, m0, n0, mrem, ncr, [=](int m, int n) { return Tr[m + n * ld]; },
               [=](int m, int n, double v, double o) { if (m >= n) Tr[m + n * ld] = o - v; });
  } else {
    int tm, tn;
    lower_pair(t - nP, tm, tn);
    const int m0 = tm * LT, n0 = tn * LT;
    const double* Pa = M.A + M.ncol + (int64_t)jb * M.ld;   // separator rows of block column jb
    const int na = M.na;
    gemm_tile64_plain<1>(acc, Pa, ld, na, Pa, ld, na, 0, w, m0, n0, smem);
    double* U = M.upd;
    tile64_rmw(acc, m0, n0, na, na, [=](int m, int n) { return U[m + (int64_t)n * na]; },
               [=](int m, int n, double v, double o) { if (m >= n) U[m + (int64_t)n * na] = o - v; });
  }
}
// after the last step of a front's Cholesky: publish the update block as packed lower triangle
__global__ void k_lf_pack_upd(MfmaArgs a) {
  const int k = a.t.lev[blockIdx.y];
  const CliqueDesc d = a.t.cl[k];
  const int na = d.na;
  const double* U = a.t.upd + (int64_t)blockIdx.z * a.t.updlen + d.upd;
  double* UP = a.t.updp + (int64_t)blockIdx.z * a.t.updplen + d.updp;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < na * na; e += gridDim.x * blockDim.x) {
    int i = e % na, j = e / na;
    if (i >= j) UP[pk_idx(i, j, na)] = U[e];
  }
}

// ---------------------------------------------------------------------------------------------
// Fronts of at most MID_MAXROWS rows: the whole blocked Cholesky of one front in ONE workgroup (sixteen waves), one
// 64-wide block column at a time in LDS.  The per-step kernels above spend ~37 us per diagonal block and ~7-10 us per
// panel / trailing launch on fronts this small (the (64,128) mid fronts and the (208,0) root of synth50k: 12 + 4
// launches for the two levels); here a block column is loaded once, factored in place by 16-column steps
// (potrf_inv16 on the diagonal 16 x 16, the rows below scaled by its inverse, the rest of the block column updated --
// all operands in LDS), written back, and the columns to its right and the update block are updated from the LDS copy.
// mode 0: front of x with its update block (assembled beforehand; published as packed lower triangle after the last
// block column); mode 2: the Y_AA block in aux (fac), in place.
// ---------------------------------------------------------------------------------------------
constexpr int MID_MAXROWS = 272;
__host__ __device__ inline int mid_ld(int rows) { return ((rows + 15) / 32) * 32 + 16; }   // >= rows, = 16 mod 32 (bank spread)
__host__ __device__ inline size_t mid_chol_lds(int rowsmax) { return ((size_t)mid_ld(rowsmax) * LB + 256 + 8) * sizeof(double); }

__global__ void __launch_bounds__(1024) k_mid_chol(MfmaArgs a, double* x, double* aux, int mode) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int k = a.t.lev[blockIdx.x];
  if (*info_of(a.t, k)) return;
  const LfMat M = lf_mat(a, k, mode, x, aux);
  const CliqueDesc d = a.t.cl[k];
  const int tid = threadIdx.x;
  const int ldp = mid_ld(M.nrow);
  double* const Pb = smem;
  double* const d16 = Pb + (size_t)ldp * LB;
  const int64_t ld = M.ld;
  __shared__ int mid_fail;
  if (tid == 0) mid_fail = 0;
  const int lane = tid & 63, wave = tid >> 6, nwv = (int)(blockDim.x >> 6), l15 = lane & 15, kq = lane >> 4;
  // one 16 x 16 tile product on the calling wave: (m, n) = sum_{k < 16} A[m + k lda] B[n + k ldb]; register r <-> (l15, kq + 4 r)
  auto tile_abt = [&](const double* A, int lda, const double* B, int ldb) {
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(B[l15 + (k0 + kq) * ldb], A[l15 + (k0 + kq) * lda], acc, 0, 0, 0);
    return acc;
  };
  for (int jb = 0; jb < M.ncol; jb += LB) {
    const int w = min(LB, M.ncol - jb), rows = M.nrow - jb;
    const double* Ab = M.A + jb + (int64_t)jb * ld;
    // the block column, zero-padded to whole 16 x 16 tiles (rows up to a multiple of 16 lie inside ldp; all 64 columns are allocated):
    // the tile products below read whole tiles, and a stray NaN times a zero of the block inverse would still be a NaN
    const int rows16 = (rows + 15) & ~15;
    for (int e = tid; e < rows16 * LB; e += 1024) {
      const int i = e % rows16, j = e / rows16;
      Pb[i + j * ldp] = (i < rows && j < w && i >= j) ? Ab[i + (int64_t)j * ld] : 0.0;
    }
    __syncthreads();
    // Sixteen-column steps with TWO barriers each (round 5; was: potrf_inv16 with fifteen waves idle, then two generic products and
    // three barriers).  Wave 0 alone carries the critical path: its share of the previous step's trailing update -- the next
    // diagonal 16 x 16 tile --, then that tile's Cholesky and inverse (wave_potrf_inv16: no barrier inside); the other waves do
    // the rest of that trailing update meanwhile.  After the barrier one wave per tile scales the rows below.
    const int nbc = (w + 15) >> 4, nbr = rows16 >> 4;
    for (int b = 0; b < nbc; ++b) {
      const int bw = min(16, w - 16 * b);
      if (wave == 0) {
        double* Dbb = Pb + 16 * b + 16 * b * ldp;
        if (b > 0) {
          const double* P = Pb + 16 * b + 16 * (b - 1) * ldp;
          const d4 u4 = tile_abt(P, ldp, P, ldp);
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int n = kq + 4 * r; if (l15 >= n) Dbb[l15 + n * ldp] -= u4[r]; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (!wave_potrf_inv16(Dbb, ldp, bw, d16) && lane == 0) mid_fail = 1;
      } else if (b > 0) {
        // tiles (r, c) of the trailing part, b <= c < nbc, c <= r < nbr, without (b, b): dealt over the waves 1 .. nwv - 1
        int t = 0;
        for (int c = b; c < nbc; ++c)
          for (int r = c; r < nbr; ++r) {
            if (r == b && c == b) continue;
            if (t++ % (nwv - 1) != wave - 1) continue;
            const d4 u4 = tile_abt(Pb + 16 * r + 16 * (b - 1) * ldp, ldp, Pb + 16 * c + 16 * (b - 1) * ldp, ldp);
            double* T = Pb + 16 * r + 16 * c * ldp;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int n = kq + 4 * q; if (r > c || l15 >= n) T[l15 + n * ldp] -= u4[q]; }
          }
      }
      __syncthreads();
      if (mid_fail) { if (tid == 0) atomicCAS(info_of(a.t, k), 0, info_val(a.t, k)); return; }
      for (int r = b + 1 + wave; r < nbr; r += nwv) {            // rows below x the block's inverse (transposed)
        double* P = Pb + 16 * r + 16 * b * ldp;
        const d4 u4 = tile_abt(P, ldp, d16, 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) P[l15 + (kq + 4 * q) * ldp] = u4[q];
      }
      if (bw < 16 && wave == nwv - 1) {
        // a partial last column tile: the rows of ITS row tile beyond the bw x bw diagonal block are rows below as well
        // (wave_potrf_inv16 left them as they were; the block inverse is zero beyond bw)
        double* P = Pb + 16 * b + 16 * b * ldp;
        const d4 u4 = tile_abt(P, ldp, d16, 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) if (l15 >= bw) P[l15 + (kq + 4 * q) * ldp] = u4[q];
      }
      __syncthreads();
    }
    // (the last step's trailing update inside the block column is empty: its columns end with the block)
    // the factored block column goes back; the columns to its right and the update block take -P P^T from the LDS copy
    double* Aw = M.A + jb + (int64_t)jb * ld;
    for (int e = tid; e < rows * w; e += 1024) {
      const int i = e % rows, j = e / rows;
      if (i >= j) Aw[i + (int64_t)j * ld] = Pb[i + j * ldp];
    }
    const int ncr = M.ncol - jb - w, mrem = rows - w;
    if (ncr > 0) {
      double* Tr = M.A + (jb + w) + (int64_t)(jb + w) * ld;
      const double* Pr = Pb + w;
      wg_mma(mrem, ncr, w, [=](int m, int kk) { return Pr[m + kk * ldp]; }, [=](int kk, int n) { return Pr[n + kk * ldp]; },
             [=](int m, int n, double acc) { if (m >= n) Tr[m + (int64_t)n * ld] -= acc; }, true);
    }
    if (M.upd) {
      const int na = M.na;
      const double* Pa = Pb + (M.ncol - jb);            // separator rows of this block column
      double* U = M.upd;
      if (jb + w >= M.ncol) {
        double* UP = a.t.updp + d.updp;
        wg_mma(na, na, w, [=](int m, int kk) { return Pa[m + kk * ldp]; }, [=](int kk, int n) { return Pa[n + kk * ldp]; },
               [=](int m, int n, double acc) { if (m >= n) UP[pk_idx(m, n, na)] = U[m + (int64_t)n * na] - acc; }, true);
      } else {
        wg_mma(na, na, w, [=](int m, int kk) { return Pa[m + kk * ldp]; }, [=](int kk, int n) { return Pa[n + kk * ldp]; },
               [=](int m, int n, double acc) { if (m >= n) U[m + (int64_t)n * na] -= acc; }, true);
      }
    }
    __syncthreads();
  }
}

// ---- inverse-form factor of large fronts: Li = L_NN^-1 by block rows, K = L_AN Li
// step ib: (after k_lf_diag wrote Dinv of block ib) S = L[ib, 0:ib] Li[0:ib, 0:ib] ; Li[ib, 0:ib] = -Dinv S
// view of one triangular inversion: src (lower, ld) -> dst (ld), order n, row-block scratch S
// mode 0: L_NN -> Li of LK;  3: the scratch-head matrix of the completion, in place;  4: fac -> faci (update layout)
struct InvView { const double* src; double* dst; int64_t ld; int n; double* S; };
__device__ inline InvView inv_view(const MfmaArgs& a, int k, const CliqueDesc& d, int mode, const double* L, double* LK) {
  InvView v;
  double* tmp = a.t.tmp + a.t.tmpptr[k];
  if (mode == 3) {
    v.src = tmp; v.dst = tmp; v.ld = d.nn; v.n = d.nn; v.S = tmp + (int64_t)d.nn * d.nn + 2 * (int64_t)d.na * d.nn;
  } else if (mode == 4) {
    // row-block scratch = the (otherwise unused) transposed position above the diagonal of dst
    v.src = L + d.upd; v.dst = LK + d.upd; v.ld = d.na; v.n = d.na; v.S = LK + d.upd;
  } else {
    v.src = L + d.blk; v.dst = LK + d.blk; v.ld = d.nn + d.na; v.n = d.nn; v.S = tmp;
  }
  return v;
}
__global__ void __launch_bounds__(256, 4) k_lf_prep_s(MfmaArgs a, const double* L, double* LK, int ib, int mode) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const int k = a.t.lev[blockIdx.y];
  const CliqueDesc d = a.t.cl[k];
  const InvView V = inv_view(a, k, d, mode, L, LK);
  const int nn = V.n;
  const int64_t nf = V.ld;
  if (ib >= nn || ib == 0) return;
  const int w = min(LB, nn - ib);
  const int n0 = blockIdx.x * LT;
  if (n0 >= ib) return;
  const double* Lk = V.src;
  const double* Li = V.dst;
  double* S = V.S;   // w x ib (ld w; transposed into dst's upper triangle in mode 4)
  const int64_t sm = mode == 4 ? nf : 1, sn = mode == 4 ? 1 : w, s0 = mode == 4 ? (int64_t)ib * nf : 0;
  d4 acc[2][2];
  tile64_zero(acc);
  gemm_tile64(acc, w, ib, ib, 0, n0, [=](int m, int kk) { return Lk[(ib + m) + (int64_t)kk * nf]; },
              [=](int kk, int n) { return ldm<1>(kk >= n, &Li[kk + (int64_t)n * nf]); }, sA, sB, n0);
  tile64_foreach(acc, 0, n0, w, ib, [=](int m, int n, double v) { S[s0 + m * sm + n * sn] = v; });
}
// hoisted != 0: the inverses of ALL diagonal blocks are already in place in dst (k_lf_diag_inv)
__global__ void __launch_bounds__(256, 4) k_lf_prep_row(MfmaArgs a, const double* L, double* LK, int ib, int mode, int hoisted) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const int k = a.t.lev[blockIdx.y];
  const CliqueDesc d = a.t.cl[k];
  const InvView V = inv_view(a, k, d, mode, L, LK);
  const int nn = V.n;
  const int64_t nf = V.ld;
  if (ib >= nn) return;
  const int w = min(LB, nn - ib);
  const double* Di = hoisted ? V.dst + ib + (int64_t)ib * nf : a.lfd + (int64_t)d.pad * (LB * LB);       // w x w
  const int64_t ldd = hoisted ? nf : w;
  double* Li = V.dst;
  const int t = blockIdx.x;
  const int ntS = tiles64(ib);
  if (t == ntS) {   // diagonal block of Li (and zeros above it)
    if (hoisted) return;
    for (int e = threadIdx.x; e < w * w; e += blockDim.x) {
      int i = e % w, j = e / w;
      Li[(ib + i) + (int64_t)(ib + j) * nf] = (i >= j) ? Di[i + j * w] : 0.0;
    }
    if (mode != 4)
      for (int e = threadIdx.x; e < ib * w; e += blockDim.x) {   // upper part: rows < ib of these columns
        int i = e % ib, j = e / ib;
        Li[i + (int64_t)(ib + j) * nf] = 0.0;
      }
    return;
  }
  if (t > ntS) return;
  const int n0 = t * LT;
  const double* S = V.S;
  const int64_t sm = mode == 4 ? nf : 1, sn = mode == 4 ? 1 : w, s0 = mode == 4 ? (int64_t)ib * nf : 0;
  d4 acc[2][2];
  tile64_zero(acc);
  gemm_tile64(acc, w, ib, w, 0, n0, [=](int m, int kk) { return Di[m + kk * ldd]; },
              [=](int kk, int n) { return S[s0 + kk * sm + n * sn]; }, sA, sB);
  tile64_foreach(acc, 0, n0, w, ib, [=](int m, int n, double v) { Li[(ib + m) + (int64_t)n * nf] = -v; });
}
// inverses of ALL 64 x 64 diagonal blocks of L_NN at once (they do not depend on each other), straight into the
// diagonal blocks of Li, with the zeros above the diagonal and in the rows above each block (mode 0 of inv_view only)
__global__ void __launch_bounds__(256) k_lf_diag_inv(MfmaArgs a, const double* L, double* LK) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const D = smem;
  double* const Di = D + LB * LBD;
  double* const d16 = Di + LB * LBD;
  double* const s16 = d16 + 256;
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, ib = blockIdx.y * LB;
  const int64_t nf = d.nn + d.na;
  if (ib >= nn) return;
  const int w = min(LB, nn - ib);
  const double* Ab = L + d.blk + ib + (int64_t)ib * nf;
  for (int e = threadIdx.x; e < w * w; e += blockDim.x) {
    int i = e % w, j = e / w;
    D[i + j * LBD] = (i >= j) ? Ab[i + (int64_t)j * nf] : 0.0;
  }
  __syncthreads();
  potrf_inv64(D, w, Di, d16, s16, false);
  double* Li = LK + d.blk;
  for (int e = threadIdx.x; e < w * w; e += blockDim.x) {
    int i = e % w, j = e / w;
    Li[(ib + i) + (int64_t)(ib + j) * nf] = (i >= j) ? Di[i + j * LBD] : 0.0;
  }
  for (int e = threadIdx.x; e < ib * w; e += blockDim.x) {   // rows above the block in these columns
    int i = e % ib, j = e / ib;
    Li[i + (int64_t)(ib + j) * nf] = 0.0;
  }
}
// Recursive blocked inversion of L_NN, one doubling level per pair of launches (after k_lf_diag_inv has inverted the
// 64 x 64 diagonal blocks): at block size b (64, 128, 256, ...) the pair p covers rows / columns [2pb, 2pb + 2b) with
// A = the first b of them, C = the rest (b2 <= b), B = L[C, A], and inv([A 0; B C]) = [Ai 0; -Ci B Ai, Ci].
// step 0: W = B * Ai into the clique's scratch;  step 1: X[C, A] = -Ci * W.  Every level is two launches of
// (pairs x (b/64)^2) tiles -- log2(nn/64) levels with growing parallelism -- where the row-by-row scheme (k_lf_prep_s /
// k_lf_prep_row per 64-row block) runs nn/64 dependent steps of at most nn/64 workgroups: 4096 front, 18.4 ms -> see DESIGN.
__global__ void __launch_bounds__(256, 4) k_lf_trtri(MfmaArgs a, const double* L, double* LK, int b, int step) {
  __shared__ __attribute__((aligned(16))) double smem[LRC_DOUBLES];
  double* const sA = smem; double* const sB = smem + LKC * LSA;
  const int k = a.t.lev[blockIdx.y];
  const CliqueDesc d = a.t.cl[k];
  const InvView V = inv_view(a, k, d, 0, L, LK);
  const int nn = V.n;
  const int64_t nf = V.ld;
  const int tpb = b / LT, tpp = tpb * tpb;
  const int p = blockIdx.x / tpp, t = blockIdx.x % tpp;
  const int r0 = 2 * p * b;
  if (r0 + b >= nn) return;                       // no C block: nothing to do for this pair at this level
  const int b2 = min(b, nn - r0 - b);
  const int m0 = (t % tpb) * LT, n0 = (t / tpb) * LT;
  if (m0 >= b2) return;
  double* W = V.S + (int64_t)p * b * b;           // b2 x b, ld b
  const double* Lk = V.src;
  double* Li = V.dst;
  d4 acc[2][2];
  tile64_zero(acc);
  if (step == 0) {
    const double* Bm = Lk + (r0 + b) + (int64_t)r0 * nf;
    const double* Ai = Li + r0 + (int64_t)r0 * nf;
    // Ai(k, n) = 0 for k < n: the mask over the tile's own columns, whole columns of Ai below them
    gemm_tile64(acc, b2, b, min(b, n0 + LT), m0, n0, [=](int m, int kk) { return Bm[m + (int64_t)kk * nf]; },
                [=](int kk, int n) { return ldm<1>(kk >= n, &Ai[kk + (int64_t)n * nf]); }, sA, sB, n0);
    gemm_tile64_plain<1, false, true>(acc, Bm, nf, b2, Ai, nf, b, n0 + LT, b, m0, n0, smem);
    tile64_foreach(acc, m0, n0, b2, b, [=](int m, int n, double v) { W[m + (int64_t)n * b] = v; });
  } else {
    const double* Ci = Li + (r0 + b) + (int64_t)(r0 + b) * nf;
    // Ci(m, k) = 0 for k > m: whole rows of Ci before the tile's first row, the mask from there on
    gemm_tile64_plain<1, false, true>(acc, Ci, nf, b2, W, b, b, 0, m0, m0, n0, smem);
    gemm_tile64(acc, b2, b, min(b2, m0 + LT), m0, n0, [=](int m, int kk) { return ldm<1>(m >= kk, &Ci[m + (int64_t)kk * nf]); },
                [=](int kk, int n) { return W[kk + (int64_t)n * b]; }, sA, sB, m0);
    double* X = Li + (r0 + b) + (int64_t)r0 * nf;
    tile64_foreach(acc, m0, n0, b2, b, [=](int m, int n, double v) { X[m + (int64_t)n * nf] = -v; });
  }
}
__global__ void __launch_bounds__(256) k_lf_prep_k(MfmaArgs a, const double* L, double* LK) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const int k = a.t.lev[blockIdx.y];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int t = blockIdx.x;
  if (t >= mtA * ntN) return;
  const int m0 = (t % mtA) * LT, n0 = (t / mtA) * LT;
  const double* Lk = L + d.blk;
  const double* Li = LK + d.blk;
  double* Kk = LK + d.blk + nn;
  d4 acc[2][2];
  tile64_zero(acc);
  gemm_tile64(acc, na, nn, nn, m0, n0, [=](int m, int kk) { return Lk[(nn + m) + (int64_t)kk * nf]; },
              [=](int kk, int n) { return ldm<1>(kk >= n, &Li[kk + (int64_t)n * nf]); }, sA, sB, n0);
  tile64_foreach(acc, m0, n0, na, nn, [=](int m, int n, double v) { Kk[m + (int64_t)n * nf] = v; });
}


// ---------------------------------------------------------------------------------------------
// Inverse Hessian factors and completion of large fronts (formulas: front_inv.hip).  a.LK = the factor L
// itself, a.ysc = faci (R^-1, lower; positions above the diagonal are scratch and must be masked).
// ---------------------------------------------------------------------------------------------
template <class F>
__device__ inline void tile64_foreach2(const d4 (&a1)[2][2], const d4 (&a2)[2][2], int m0, int n0, int M, int N, F f) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m, n;
        const bool has = tile64_pos(a, b, r, m, n);
        m += m0; n += n0;
        if (has && m < M && n < N) f(m, n, a1[a][b][r], a2[a][b][r]);
      }
}
// AN rows of the panel <-> G scratch.  dir 0: panel = G, 1: G = panel
__global__ void k_lf_copy_an(MfmaArgs a, double* u, int64_t ldu, int dir) {
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < na * nn; e += gridDim.x * blockDim.x) {
    const int i = e % na, j = e / na;
    if (dir) c.G[e] = c.P[nn + i + (int64_t)j * nf]; else c.P[nn + i + (int64_t)j * nf] = c.G[e];
  }
}
// G = Ri^T (AN rows of the panel)  (tr = 1)   or   G = Ri (AN rows)  (tr = 0)
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_ri_an(MfmaArgs a, double* u, int64_t ldu, int tr) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int t = blockIdx.x;
  if (t >= mtA * ntN) return;
  const int m0 = (t % mtA) * LT, n0 = (t / mtA) * LT;
  const double* Y = c.Ys; const double* P = c.P;
  d4 acc[2][2];
  tile64_zero(acc);
  if (tr)
    gemm_tile64<PD>(acc, na, nn, na, m0, n0, [=](int m, int kk) { return ldm<PD>(kk >= m, &Y[kk + (int64_t)m * na]); },
                [=](int kk, int n) { return P[nn + kk + (int64_t)n * nf]; }, sA, sB);
  else
    gemm_tile64<PD>(acc, na, nn, na, m0, n0, [=](int m, int kk) { return ldm<PD>(m >= kk, &Y[m + (int64_t)kk * na]); },
                [=](int kk, int n) { return P[nn + kk + (int64_t)n * nf]; }, sA, sB);
  double* G = c.G;
  tile64_foreach(acc, m0, n0, na, nn, [=](int m, int n, double v) { G[m + (int64_t)n * na] = v; });
}

// ---- G^-adj phase 1: Q = Z_AN L_NN + Z_AA L_AN (into G) ; Q'' = Z_AN L_NN + Z_AA L_AN / 2 (into E) ; T = Z_NN L_NN
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_dinv1(MfmaArgs a, double* u, int64_t ldu) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int nE = mtA * ntN, nT = ntN * ntN;
  const int t = blockIdx.x;
  if (t >= nE + nT) return;
  const double* P = c.P; const double* Lnn = c.Li; const double* Lan = c.K; const double* Z = c.U;
  auto lnn = [=](int kk, int n) { return Lnn[kk + (int64_t)n * nf]; };
  d4 acc[2][2];
  tile64_zero(acc);
  if (t < nE) {
    const int m0 = (t % mtA) * LT, n0 = (t / mtA) * LT;
    d4 acc2[2][2];
    tile64_zero(acc2);
    gemm_tile64<PD>(acc, na, nn, nn, m0, n0, [=](int m, int kk) { return P[nn + m + (int64_t)kk * nf]; }, lnn, sA, sB);
    gemm_tile64<PD>(acc2, na, nn, na, m0, n0,
                [=](int m, int kk) { return Z[max(m, kk) + (int64_t)min(m, kk) * na]; },
                [=](int kk, int n) { return Lan[kk + (int64_t)n * nf]; }, sA, sB);
    double* E = c.E; double* G = c.G;
    tile64_foreach2(acc, acc2, m0, n0, na, nn, [=](int m, int n, double v1, double v2) {
      G[m + (int64_t)n * na] = v1 + v2;
      E[m + (int64_t)n * na] = v1 + 0.5 * v2;
    });
  } else {
    const int tt = t - nE, m0 = (tt % ntN) * LT, n0 = (tt / ntN) * LT;
    gemm_tile64<PD>(acc, nn, nn, nn, m0, n0,
                [=](int m, int kk) { return P[max(m, kk) + (int64_t)min(m, kk) * nf]; }, lnn, sA, sB);
    double* T = c.T;
    tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { T[m + (int64_t)n * nn] = v; });
  }
}
// ---- G^-adj phase 2: G_NN = L_NN^T T + L_AN^T Q'' + Q''^T L_AN (lower, into the panel) ; AN rows = Q or Ri Q
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_dinv2(MfmaArgs a, double* u, int64_t ldu) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int nN = ntN * (ntN + 1) / 2, nA = mtA * ntN;
  const int t = blockIdx.x;
  if (t >= nN + nA) return;
  const double* Lnn = c.Li; const double* Lan = c.K; const double* T = c.T; const double* E = c.E; const double* G = c.G;
  double* Pw = c.P;
  d4 acc[2][2];
  tile64_zero(acc);
  if (t < nN) {
    int tm, tn;
    lower_pair(t, tm, tn);
    const int m0 = tm * LT, n0 = tn * LT;
    gemm_tile64<PD>(acc, nn, nn, nn, m0, n0, [=](int m, int kk) { return Lnn[kk + (int64_t)m * nf]; },
                [=](int kk, int n) { return T[kk + (int64_t)n * nn]; }, sA, sB);
    gemm_tile64<PD>(acc, nn, nn, na, m0, n0, [=](int m, int kk) { return Lan[kk + (int64_t)m * nf]; },
                [=](int kk, int n) { return E[kk + (int64_t)n * na]; }, sA, sB);
    gemm_tile64<PD>(acc, nn, nn, na, m0, n0, [=](int m, int kk) { return E[kk + (int64_t)m * na]; },
                [=](int kk, int n) { return Lan[kk + (int64_t)n * nf]; }, sA, sB);
    tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { if (m >= n) Pw[m + (int64_t)n * nf] = v; });
  } else {
    const int tt = t - nN, m0 = (tt % mtA) * LT, n0 = (tt / mtA) * LT;
    if (a.ymode) {
      const double* Y = c.Ys;
      gemm_tile64<PD>(acc, na, nn, na, m0, n0, [=](int m, int kk) { return ldm<PD>(m >= kk, &Y[m + (int64_t)kk * na]); },
                  [=](int kk, int n) { return G[kk + (int64_t)n * na]; }, sA, sB);
      tile64_foreach(acc, m0, n0, na, nn, [=](int m, int n, double v) { Pw[nn + m + (int64_t)n * nf] = v; });
    } else {
      tile64_rmw(acc, m0, n0, na, nn, [=](int m, int n) { return G[m + (int64_t)n * na]; },
               [=](int m, int n, double, double g) { Pw[nn + m + (int64_t)n * nf] = g; });
    }
  }
}
// ---- G^-1 phase 1: V = G_AN + L_AN G_NN / 2 (into E) ; T = G_NN L_NN^T
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_uinv1(MfmaArgs a, double* u, int64_t ldu) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int nE = mtA * ntN, nT = ntN * ntN;
  const int t = blockIdx.x;
  if (t >= nE + nT) return;
  const double* P = c.P; const double* Lnn = c.Li; const double* Lan = c.K;
  auto gsym = [=](int i, int j) { return P[max(i, j) + (int64_t)min(i, j) * nf]; };
  d4 acc[2][2];
  tile64_zero(acc);
  if (t < nE) {
    const int m0 = (t % mtA) * LT, n0 = (t / mtA) * LT;
    gemm_tile64<PD>(acc, na, nn, nn, m0, n0, [=](int m, int kk) { return Lan[m + (int64_t)kk * nf]; }, gsym, sA, sB);
    double* E = c.E;
    tile64_rmw(acc, m0, n0, na, nn, [=](int m, int n) { return P[nn + m + (int64_t)n * nf]; },
               [=](int m, int n, double v, double f) { E[m + (int64_t)n * na] = f + 0.5 * v; });
  } else {
    const int tt = t - nE, m0 = (tt % ntN) * LT, n0 = (tt / ntN) * LT;
    gemm_tile64<PD>(acc, nn, nn, nn, m0, n0, gsym, [=](int kk, int n) { return Lnn[n + (int64_t)kk * nf]; }, sA, sB);
    double* T = c.T;
    tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { T[m + (int64_t)n * nn] = v; });
  }
}
// ---- G^-1 phase 2: U = -(V L_AN^T + L_AN V^T) (lower) ; G = (2V - G_AN) L_NN^T ; F_NN = L_NN T (lower, into the panel)
template <int PD>
__global__ void __launch_bounds__(PD == 4 ? 1024 : 256, 4) k_lf_uinv2(MfmaArgs a, double* u, int64_t ldu) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const LfCtx c = lf_ctx(a, u, ldu);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int nU = mtA * (mtA + 1) / 2, nG = mtA * ntN, nN = ntN * (ntN + 1) / 2;
  const int t = blockIdx.x;
  if (t >= nU + nG + nN) return;
  const double* Lan = c.K; const double* E = c.E; const double* Lnn = c.Li; const double* T = c.T; const double* P = c.P;
  d4 acc[2][2];
  tile64_zero(acc);
  if (t < nU) {
    int tm, tn;
    lower_pair(t, tm, tn);
    const int m0 = tm * LT, n0 = tn * LT;
    gemm_tile64<PD>(acc, na, na, nn, m0, n0, [=](int m, int kk) { return E[m + (int64_t)kk * na]; },
                [=](int kk, int n) { return Lan[n + (int64_t)kk * nf]; }, sA, sB);
    gemm_tile64<PD>(acc, na, na, nn, m0, n0, [=](int m, int kk) { return Lan[m + (int64_t)kk * nf]; },
                [=](int kk, int n) { return E[n + (int64_t)kk * na]; }, sA, sB);
    double* U = c.U;
    tile64_foreach(acc, m0, n0, na, na, [=](int m, int n, double v) { if (m >= n) U[m + (int64_t)n * na] = -v; });
  } else if (t < nU + nG) {
    const int tt = t - nU, m0 = (tt % mtA) * LT, n0 = (tt / mtA) * LT;
    gemm_tile64<PD>(acc, na, nn, nn, m0, n0,
                [=](int m, int kk) { return 2.0 * E[m + (int64_t)kk * na] - P[nn + m + (int64_t)kk * nf]; },
                [=](int kk, int n) { return Lnn[n + (int64_t)kk * nf]; }, sA, sB);
    double* G = c.G;
    tile64_foreach(acc, m0, n0, na, nn, [=](int m, int n, double v) { G[m + (int64_t)n * na] = v; });
  } else {
    int tm, tn;
    lower_pair(t - nU - nG, tm, tn);
    const int m0 = tm * LT, n0 = tn * LT;
    gemm_tile64<PD>(acc, nn, nn, nn, m0, n0, [=](int m, int kk) { return Lnn[m + (int64_t)kk * nf]; },
                [=](int kk, int n) { return T[kk + (int64_t)n * nn]; }, sA, sB);
    __syncthreads();
    double* Pw = c.P;
    tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { if (m >= n) Pw[m + (int64_t)n * nf] = v; });
  }
}

// ---- completion, large fronts (scratch: T nn x nn | E na x nn | G na x nn | row-block scratch of the inversion)
// step 0: E = Ri X_AN ; step 1: G = Ri^T E ; step 2: T = reversed(X_NN - X_AN^T G) ; step 3: L_NN = reversed(T^-1)^T, L_AN = -G L_NN
__global__ void __launch_bounds__(256, 4) k_lf_completion(MfmaArgs a, double* x, int step) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const LfCtx c = lf_ctx(a, x, 0);
  if (*info_of(a.t, c.k)) return;
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int t = blockIdx.x;
  const double* Y = c.Ys; const double* P = c.P;
  double* E = c.E; double* G = c.G; double* T = c.T;
  d4 acc[2][2];
  tile64_zero(acc);
  if (step <= 1) {
    if (t >= mtA * ntN) return;
    const int m0 = (t % mtA) * LT, n0 = (t / mtA) * LT;
    if (step == 0)
      gemm_tile64(acc, na, nn, na, m0, n0, [=](int m, int kk) { return ldm<1>(m >= kk, &Y[m + (int64_t)kk * na]); },
                  [=](int kk, int n) { return P[nn + kk + (int64_t)n * nf]; }, sA, sB);
    else
      gemm_tile64(acc, na, nn, na, m0, n0, [=](int m, int kk) { return ldm<1>(kk >= m, &Y[kk + (int64_t)m * na]); },
                  [=](int kk, int n) { return E[kk + (int64_t)n * na]; }, sA, sB);
    double* O = step == 0 ? E : G;
    tile64_foreach(acc, m0, n0, na, nn, [=](int m, int n, double v) { O[m + (int64_t)n * na] = v; });
  } else if (step == 2) {
    if (t >= ntN * ntN) return;
    const int m0 = (t % ntN) * LT, n0 = (t / ntN) * LT;
    gemm_tile64(acc, nn, nn, na, m0, n0, [=](int m, int kk) { return P[nn + kk + (int64_t)m * nf]; },
                [=](int kk, int n) { return G[kk + (int64_t)n * na]; }, sA, sB);
    tile64_rmw(acc, m0, n0, nn, nn, [=](int m, int n) { return P[max(m, n) + (int64_t)min(m, n) * nf]; },
               [=](int m, int n, double v, double f) { T[(nn - 1 - m) + (int64_t)(nn - 1 - n) * nn] = f - v; });
  } else {
    const int nN = ntN * ntN, nA = mtA * ntN;
    if (t >= nN + nA) return;
    double* Pw = c.P;
    if (t < nN) {
      const int m0 = (t % ntN) * LT, n0 = (t / ntN) * LT;
      for (int e = threadIdx.x; e < LT * LT; e += blockDim.x) {
        const int m = m0 + (e & 63), n = n0 + (e >> 6);
        if (m < nn && n < nn) Pw[m + (int64_t)n * nf] = m >= n ? T[(nn - 1 - n) + (int64_t)(nn - 1 - m) * nn] : 0.0;
      }
    } else {
      const int tt = t - nN, m0 = (tt % mtA) * LT, n0 = (tt / mtA) * LT;
      gemm_tile64(acc, na, nn, nn, m0, n0, [=](int m, int kk) { return G[m + (int64_t)kk * na]; },
                  [=](int kk, int n) { return ldm<1>(kk >= n, &T[(nn - 1 - n) + (int64_t)(nn - 1 - kk) * nn]); }, sA, sB);
      tile64_foreach(acc, m0, n0, na, nn, [=](int m, int n, double v) { Pw[nn + m + (int64_t)n * nf] = -v; });
    }
  }
}

// ---- llt, large fronts.  step 0: T = L_NN L_NN^T (lower), G = L_AN L_NN^T, U = L_AN L_AN^T (lower, assigned);
// step 1: panel <- (T lower, G).  The children are added afterwards by the assemble kernel (sgn 2) and the update
// block is published packed by k_lf_pack_upd.
__global__ void __launch_bounds__(256, 4) k_lf_llt(MfmaArgs a, double* x, int step) {
  __shared__ double sA[LKC * LSA], sB[LT * LSB];
  const LfCtx c = lf_ctx(a, x, 0);
  const int nn = c.nn, na = c.na, nf = c.nf;
  const int mtA = tiles64(na), ntN = tiles64(nn);
  const int nT = ntN * (ntN + 1) / 2, nG = mtA * ntN, nU = mtA * (mtA + 1) / 2;
  const int t = blockIdx.x;
  if (t >= nT + nG + nU) return;
  double* Pw = c.P;
  if (step == 1) {
    if (t >= nT + nG) return;
    int tm, tn;
    if (t < nT) lower_pair(t, tm, tn); else { tm = (t - nT) % mtA; tn = (t - nT) / mtA; }
    for (int e = threadIdx.x; e < LT * LT; e += blockDim.x) {
      const int m = tm * LT + (e & 63), n = tn * LT + (e >> 6);
      if (t < nT) { if (m < nn && n < nn && m >= n) Pw[m + (int64_t)n * nf] = c.T[m + (int64_t)n * nn]; }
      else if (m < na && n < nn) Pw[nn + m + (int64_t)n * nf] = c.G[m + (int64_t)n * na];
    }
    return;
  }
  const double* P = c.P;
  auto lnnT = [=](int kk, int n) { return ldm<1>(n >= kk, &P[n + (int64_t)kk * nf]); };
  d4 acc[2][2];
  tile64_zero(acc);
  if (t < nT) {
    int tm, tn;
    lower_pair(t, tm, tn);
    const int m0 = tm * LT, n0 = tn * LT;
    gemm_tile64(acc, nn, nn, nn, m0, n0, [=](int m, int kk) { return ldm<1>(m >= kk, &P[m + (int64_t)kk * nf]); }, lnnT, sA, sB);
    double* T = c.T;
    tile64_foreach(acc, m0, n0, nn, nn, [=](int m, int n, double v) { if (m >= n) T[m + (int64_t)n * nn] = v; });
  } else if (t < nT + nG) {
    const int tt = t - nT, m0 = (tt % mtA) * LT, n0 = (tt / mtA) * LT;
    gemm_tile64(acc, na, nn, nn, m0, n0, [=](int m, int kk) { return P[nn + m + (int64_t)kk * nf]; }, lnnT, sA, sB);
    double* G = c.G;
    tile64_foreach(acc, m0, n0, na, nn, [=](int m, int n, double v) { G[m + (int64_t)n * na] = v; });
  } else {
    int tm, tn;
    lower_pair(t - nT - nG, tm, tn);
    const int m0 = tm * LT, n0 = tn * LT;
    gemm_tile64(acc, na, na, nn, m0, n0, [=](int m, int kk) { return P[nn + m + (int64_t)kk * nf]; },
                [=](int kk, int n) { return P[nn + n + (int64_t)kk * nf]; }, sA, sB);
    double* U = c.U;
    tile64_foreach(acc, m0, n0, na, na, [=](int m, int n, double v) { if (m >= n) U[m + (int64_t)n * na] = v; });
  }
}

// ---- multi-GPU boundary exchange: packed update blocks of the listed cliques <-> a contiguous buffer
// buffer layout: [clique in list][rhs][packed entries];  bptr[c] = start of clique c's slab (in doubles per rhs)
__global__ void k_exchange_copy(const CliqueDesc* cl, const int64_t* list, const int64_t* bptr, int nrhs,
                                double* updp, int64_t updplen, double* buf, int unpack) {
  const int64_t k = list[blockIdx.y];
  const CliqueDesc d = cl[k];
  const int np = d.na * (d.na + 1) / 2;
  const int r = blockIdx.z;
  double* slab = buf + bptr[blockIdx.y] * nrhs + (int64_t)r * np;
  double* src = updp + (int64_t)r * updplen + d.updp;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < np; e += gridDim.x * blockDim.x) {
    if (unpack) src[e] = slab[e]; else slab[e] = src[e];
  }
}

// the same for the subtree roots of a partition (csp_set_partition): root q belongs to rank owner[q] and starts bptr[q]
// doubles (per right-hand side) into its rank's region.  pack: this rank's roots -> buf; unpack: every OTHER rank's roots
// <- buf + owner * width (the all-gathered regions one after the other).  One launch each, nothing on the host.
// unpack: 0 = pack this rank's roots, 1 = unpack the other ranks' roots, 2 = unpack EVERY rank's roots (its own too: the
// exchange by constraint share, where the sender's slots are numbered by the chunk and the receiver's by its share);
// r0: first slot of the exchange buffer's right-hand sides on the packing side
__global__ void k_exchange_roots(const CliqueDesc* cl, const int32_t* roots, const int32_t* owner, const int64_t* bptr, int me,
                                 int nrhs, double* updp, int64_t updplen, double* buf, int64_t width, int unpack, int r0 = 0) {
  const int q = blockIdx.y;
  if (unpack == 1 ? owner[q] == me : (unpack == 0 && owner[q] != me)) return;
  const CliqueDesc d = cl[roots[q]];
  const int np = d.na * (d.na + 1) / 2;
  const int r = blockIdx.z;
  double* slab = buf + (unpack ? owner[q] * width : 0) + bptr[q] * nrhs + (int64_t)r * np;
  double* src = updp + (int64_t)(r + (unpack ? 0 : r0)) * updplen + d.updp;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < np; e += gridDim.x * blockDim.x) {
    if (unpack) src[e] = slab[e]; else slab[e] = src[e];
  }
}

// Boundary blocks of a sweep whose input is a linear combination of inputs swept before (the second Hessian of solve_:
// Aadj(y) - bx, solvers.py:528-531): for every subtree root of ANOTHER rank
//   out[root] = sum_i y[i] g[root][i] - out[root]   (mode 0: out holds the blocks received for bx)
//   out[root] += sum_i y[i] g[root][i]               (mode 1: a further chunk of constraints)
// with g the buffer gathered during the Schur sweeps of the constraints (region of rank o at o * gwidth, root q at
// bptr[q] * nrhs, right-hand side i at + i * np) -- no collective for this sweep.
__global__ void k_exchange_combine(const CliqueDesc* cl, const int32_t* roots, const int32_t* owner, const int64_t* bptr, int me,
                                   int nrhs, const double* y, const double* g, int64_t gwidth, double* out, int64_t owidth, int mode) {
  const int q = blockIdx.y;
  if (owner[q] == me) return;
  const CliqueDesc d = cl[roots[q]];
  const int np = d.na * (d.na + 1) / 2;
  const double* gq = g + owner[q] * gwidth + bptr[q] * nrhs;
  double* oq = out + owner[q] * owidth + bptr[q];
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < np; e += gridDim.x * blockDim.x) {
    // twenty loads in flight per thread, summed in the order of the constraints (one at a time was a chain of nrhs memory
    // round trips: 84 us for the 100 constraints of synth50k on rank 0 of 8)
    double s = 0.0;
    const double o0 = oq[e];
    for (int i0 = 0; i0 < nrhs; i0 += 20) {
      double v[20];
#pragma unroll
      for (int u = 0; u < 20; ++u) v[u] = gq[(int64_t)min(i0 + u, nrhs - 1) * np + e];
#pragma unroll
      for (int u = 0; u < 20; ++u) s += (i0 + u < nrhs) ? y[min(i0 + u, nrhs - 1)] * v[u] : 0.0;
    }
    oq[e] = mode ? o0 + s : s - o0;
  }
}

// dst <- src on the update-layout blocks (na x na at d.upd) of the cliques of a launch list
__global__ void k_copy_upd_blocks(TreeArgs a, const double* src, double* dst) {
  const CliqueDesc d = a.cl[a.lev[blockIdx.x]];
  const int64_t len = (int64_t)d.na * d.na;
  for (int64_t e = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; e < len; e += (int64_t)gridDim.y * blockDim.x)
    dst[d.upd + e] = src[d.upd + e];
}

}  // namespace smcp
