// Sparse-input family kernel of the Schur-complement sweeps (leaves->root half-Hessian with the scaling R^T, ymode 2):
// one workgroup owns a small parent front (nn <= 16, na <= 64) TOGETHER with its <= 8 childless children, as
// k_hess_up_fam does, but built around three facts measured on that kernel:
//
//  1. A constraint touches a childless clique in a handful of entries (synth50k: 0.85 on average, 43 % of the
//     (child, constraint) pairs have none), and the sweep of ONE entry is closed form -- with a = Li[:, i],
//     b = Li[:, j], p = K[:, i], q = K[:, j], R R^T = Y_AA, M = R^T:
//       entry v at (separator row i, column j):   G_NN = 0,  G_AN = v e_i b^T,  Q = M G_AN = v M[:, i] b^T,
//                                                update = -v (q e_i^T + e_i q^T)       (one row / column)
//       entry v at (i, j) of the supernode block: G_NN = v (a b^T + b a^T),  Q = -v ((M p) b^T + (M q) a^T),
//                                                update = v (p q^T + q p^T)            (halved for i == j)
//     i.e. O(na nn) multiply-adds per entry instead of the 36 padded 16 x 16 x 4 MFMAs of the dense child sweep,
//     on the vector pipe, which runs beside the matrix pipe.  The children's constants (K, M K, Li, R by rows,
//     relative indices) are laid out once per sweep call by k_fam2_prep and copied into LDS; the children never
//     touch the matrix pipe.
//  2. vmcnt is ONE in-order counter for loads and stores: a wave that streams its results out and then waits for
//     a load (entry lists, spilled registers) waits for all its older stores to reach L2.  The steady-state loop
//     issues NO vector load: the entry lists of all the right-hand sides of the workgroup are staged in LDS during
//     the set-up (overflow: read through the scalar cache, s_load, counted by lgkmcnt), everything else comes from
//     LDS or registers.
//  3. The parent's sweep is a chain of small dependent products; one rhs per workgroup leaves the matrix pipe idle
//     while LDS operands travel.  Two independent groups of four waves each sweep their own right-hand side on
//     their own front (no workgroup barrier after the set-up; a group synchronises on an LDS counter), so each SIMD
//     always has a second wave to issue from.  The front's update part is held packed (lower triangle, the layout
//     of the exchange buffer), E is computed once per row tile and shared through LDS, K lives in registers.
//
// Per right-hand side a group runs:  children (vector pipe, LDS atomics into the front) | barrier |
//   A: E_t = F_AN - K F_NN / 2, G_t = (F_AN - K F_NN) Li^T by the wave that owns row tile t (+ T, G_NN by role 0) |
//   barrier | B: update tiles (t, tn <= t) -> packed, straight to HBM; Q_t = R^T G -> panel | barrier.
// Mathematics as in front_mfma.hip (reference call site solvers.py:483 through the Gram
// formulation of solvers.py:414-420).
#include <hip/hip_runtime.h>

#include <type_traits>

namespace smcp {

// result stores of the family kernels: streamed once, read again only by the Gram kernel a millisecond later
#define FAM2_ST(p, v) (*(p) = (v))

typedef const int32_t __attribute__((address_space(4))) cs_i32;
typedef const double __attribute__((address_space(4))) cs_f64;
__device__ inline cs_i32* as_scalar(const int32_t* p) { return (cs_i32*)(unsigned long long)p; }
__device__ inline cs_f64* as_scalar(const double* p) { return (cs_f64*)(unsigned long long)p; }

// ---- per-child constants in their LDS layout (doubles):  K | M K | Li | R by rows (packed) | rel (ints)
struct Fam2C { int cK, cMK, cLi, cR, cRel, cstride; };
__host__ __device__ inline Fam2C fam2_child_layout(int cnn, int csa) {
  Fam2C C{};
  int c = 0;
  C.cK = c; c += cnn * csa;
  C.cMK = c; c += cnn * csa;
  C.cLi = c; c += cnn * 16;
  C.cR = c; c += ((csa * (csa + 1) / 2 + 1) & ~1);
  C.cRel = c; c += csa / 2;
  C.cstride = c;
  return C;
}
// Header of a family record (ints): [0] parent clique, [1] nn, [2] na, [3] children, [4,5] panel offset, [6,7] offset of
// R in the factor buffer, [8,9] offset of the packed update; child c at 16 + 6 c: clique (-1: none), nn, na, -, panel
// offset (two ints).  The main kernel then needs no dependent descriptor loads.
constexpr int FAM2_HDR = 32;      // doubles
__host__ __device__ inline int64_t fam2_const_doubles(int cnn, int csa) { return FAM2_HDR + 8 * (int64_t)fam2_child_layout(cnn, csa).cstride; }

struct Fam2L {   // LDS layout in doubles
  int oGrp, gstride;                       // two groups: F_NN | F_AN | U (packed lower) | E | G
  int gFnn, gFan, gU, gE, gG;
  int oCh;                                 // eight children (Fam2C)
  int oCnt;                                // two barrier counters (ints)
  int oTab;                                // entry table: per (pass, member) two ints, then the staged entries
};
template <int NAT>
__host__ __device__ inline Fam2L fam2_layout(int cnn, int csa) {
  constexpr int NA = 16 * NAT, LDA = NA + 1, LDN = 17;
  Fam2L L{};
  int o = 0;
  L.oGrp = o;
  int g = 0;
  L.gFnn = g; g += LDN * 16;
  L.gFan = g; g += LDA * 16;
  L.gU = g; g += ((NA * (NA + 1) / 2 + 1) & ~1);
  L.gE = g; g += LDA * 16;
  L.gG = g; g += LDA * 16;
  L.gstride = g;
  o += 2 * g;
  L.oCh = o;
  o += 8 * fam2_child_layout(cnn, csa).cstride;
  L.oCnt = o; o += 6;                      // ints: two counters, the epoch length, -, the children's separator sizes
  L.oTab = o;
  return L;
}

// barrier among the four waves of a group (monotone LDS counter; bounded spin: a logic error shows as a failed solve)
__device__ inline void fam2_barrier(int* cnt, int& target, int lane, int* info) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  target += 4;
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  int guard = 0;
  // no s_sleep between the polls: the wait is short (the four waves arrive within a few hundred cycles of each other)
  // and a sleeping wave adds its 64-cycle granule to every one of the three barriers of a right-hand side
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    if (++guard > (1 << 24)) { if (lane == 0) atomicCAS(info, 0, -7); break; }
  }
  asm volatile("" ::: "memory");
}

// One workgroup per family parent, wave w = child w: the constants of the children in the layout k_fam_sparse
// copies into LDS.  Runs once per sweep call (the factor may have changed since the last one).
__global__ void __launch_bounds__(512) k_fam2_prep(MfmaArgs a, double* famc, int cnn, int csa) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Fam2C C = fam2_child_layout(cnn, csa);
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nch = d.chend - d.chbeg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* const out = famc + (int64_t)blockIdx.x * fam2_const_doubles(cnn, csa);
  for (int e = tid; e < 8 * C.cstride; e += 512) smem[e] = 0.0;
  __syncthreads();
  if (wave < nch) {
    const int ck = a.t.chidx[d.chbeg + wave];
    const CliqueDesc cd = a.t.cl[ck];
    const int nnc = cd.nn, nac = cd.na, nfc = nnc + nac;
    double* const cb = smem + wave * C.cstride;
    const double* lk = a.LK + cd.blk;
    for (int e = lane; e < nfc * nnc; e += 64) {
      const int i = e % nfc, j = e / nfc;
      if (i >= nnc) cb[C.cK + j * csa + (i - nnc)] = lk[e];
      else if (i >= j) cb[C.cLi + j * 16 + i] = lk[e];
    }
    const double* ys = a.ysc + cd.upd;           // R (lower, column-major na x na); row i packed at i (i + 1) / 2
    for (int e = lane; e < nac * nac; e += 64) {
      const int i = e % nac, m = e / nac;
      if (i >= m) cb[C.cR + i * (i + 1) / 2 + m] = ys[e];
    }
    int* const crel = reinterpret_cast<int*>(cb + C.cRel);
    for (int e = lane; e < nac; e += 64) crel[e] = a.t.relidx[cd.rel + e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // M K with M = R^T:  (M K)[m][c] = sum_{q >= m} R[q][m] K[q][c]
    for (int e = lane; e < nac * nnc; e += 64) {
      const int m = e % nac, c = e / nac;
      double s = 0.0;
      for (int q = m; q < nac; ++q) s += cb[C.cR + q * (q + 1) / 2 + m] * cb[C.cK + c * csa + q];
      cb[C.cMK + c * csa + m] = s;
    }
    if (lane == 0) {
      int* const hdr = reinterpret_cast<int*>(out) + 16 + 6 * wave;
      hdr[0] = ck; hdr[1] = nnc; hdr[2] = nac; hdr[3] = 0;
      hdr[4] = (int)(cd.blk & 0xffffffffll); hdr[5] = (int)(cd.blk >> 32);
    }
  } else if (wave < 8 && lane == 0) {
    int* const hdr = reinterpret_cast<int*>(out) + 16 + 6 * wave;
    hdr[0] = -1; hdr[1] = 1; hdr[2] = 0; hdr[3] = 0; hdr[4] = 0; hdr[5] = 0;
  }
  if (tid == 0) {
    int* const hdr = reinterpret_cast<int*>(out);
    hdr[0] = k; hdr[1] = d.nn; hdr[2] = d.na; hdr[3] = nch;
    hdr[4] = (int)(d.blk & 0xffffffffll); hdr[5] = (int)(d.blk >> 32);
    hdr[6] = (int)(d.upd & 0xffffffffll); hdr[7] = (int)(d.upd >> 32);
    hdr[8] = (int)(d.updp & 0xffffffffll); hdr[9] = (int)(d.updp >> 32);
  }
  __syncthreads();
  for (int e = tid; e < 8 * C.cstride; e += 512) out[FAM2_HDR + e] = smem[e];
}


// A staged entry: packed word (row | column << 8 | front row of its separator row << 16 | FAM2_COOP) and its value.
// Entries in the supernode block of a child (rank-2 update of the whole separator block, na (na + 1) / 2 LDS atomics)
// are also listed per pass (FAM2_NNCAP of them; more: the owning half-wave does the update itself) and their
// updates are shared by the four waves of the group.
constexpr int FAM2_COOP = 1 << 24;
constexpr int FAM2_NNCAP = 3;
// doubles of LDS behind Fam2L::oTab for a table of T passes (nine members at most) and room for E staged entries
__host__ __device__ inline int fam2_tail_doubles(int T, int nmem) {
  return ((T * nmem + 1) & ~1) + (((T + 1) / 2 + 1) & ~1) + (((FAM2_NNCAP * T + 1) / 2 + 1) & ~1) + FAM2_NNCAP * T;
}

// CHP = false: the children's panels are NOT formed (their block of the Gram matrix comes from k_leaf_gram in closed
// form, front_leafgram.hip); the children then only send their updates to the parent's front.
template <int NAT, int KSN, bool CHP>
__global__ void __launch_bounds__(512) k_fam_sparse(MfmaArgs a, double* u, int64_t ldu, const double* famc, int cnn, int csa,
                                                    const int32_t* kc_ij, int tabpasses, int ecap) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NA = 16 * NAT, LDA = NA + 1, LDN = 17;
  const Fam2L L = fam2_layout<NAT>(cnn, csa);
  const Fam2C C = fam2_child_layout(cnn, csa);
  const double* const fc = famc + (int64_t)blockIdx.x * fam2_const_doubles(cnn, csa);
  const int32_t* const hdr = reinterpret_cast<const int32_t*>(fc);
  const int k = hdr[0], nn = hdr[1], na = hdr[2], nch = hdr[3], nf = nn + na;
  const int64_t pblk = (int64_t)(uint32_t)hdr[4] | ((int64_t)hdr[5] << 32);
  const int64_t pupd = (int64_t)(uint32_t)hdr[6] | ((int64_t)hdr[7] << 32);
  const int64_t pupdp = (int64_t)(uint32_t)hdr[8] | ((int64_t)hdr[9] << 32);
  const int tid = threadIdx.x;
  const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, gw = wave & 3;
  const int rt = grp ? 3 - gw : gw;             // row tile owned in the parent's sweep (mirrored in group 1: the
                                                // two waves of a SIMD then carry 40 + 44 / 36 + 40 MFMAs per pair)
  const int nmem = nch + 1;                     // members with entry lists: 0 = the parent, 1 + c = child c
  const int gy = (int)gridDim.y;
  const int npass = ((int)a.nrhs - (int)blockIdx.y + gy - 1) / gy;
  int* const cnt = reinterpret_cast<int*>(smem + L.oCnt) + grp;
  int* const tab = reinterpret_cast<int*>(smem + L.oTab);                    // [pass in epoch][member] -> (count, where)
  int* const nncnt = reinterpret_cast<int*>(smem + L.oTab + ((tabpasses * nmem + 1) & ~1));          // per pass
  int* const nnpk = nncnt + 2 * (((tabpasses + 1) / 2 + 1) & ~1);                                     // child | i << 8 | j << 16
  double* const nnv = reinterpret_cast<double*>(nnpk + 2 * (((FAM2_NNCAP * tabpasses + 1) / 2 + 1) & ~1));
  double* const lval = nnv + FAM2_NNCAP * tabpasses;                         // staged entries: values ...
  int* const lpk = reinterpret_cast<int*>(lval + ecap);                      // ... and packed words
  int* const cnac = reinterpret_cast<int*>(smem + L.oCnt) + 4;               // separator sizes of the eight children

#ifdef SMCP_STAMPS   // diagnostic build only (SMCP_STAMPS=1 python -m smcp_amd.build --force; scratch/stamps_fam2.py)
  const bool stamp = a.dbg && grp == 0 && lane == 0;
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamp ? clock64() : 0;
#define STAMP(i) do { if (stamp) { unsigned long long tn_ = clock64(); tph[i] += tn_ - tlast; tlast = tn_; } } while (0)
  unsigned long long tset[8] = {0, 0, 0, 0, 0, 0, 0, 0}, slast = tlast;
#define SETUP(i) do { if (stamp && rt == 0) { unsigned long long tn_ = clock64(); tset[i] += tn_ - slast; slast = tn_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#define SETUP(i) do { } while (0)
#endif
  // ---- set-up: fronts cleared, children's constants copied
  for (int e = tid; e < L.oCh; e += 512) smem[e] = 0.0;
  for (int e = tid; e < 8 * C.cstride; e += 512) smem[L.oCh + e] = fc[FAM2_HDR + e];
  if (tid < 12) reinterpret_cast<int*>(smem + L.oCnt)[tid] = (tid >= 4 && tid - 4 < nch) ? hdr[16 + 6 * (tid - 4) + 2] : 0;
  SETUP(0);
  // parent operands in registers: K of row tile rt (left operand), Li (right operand of X Li^T, left of Li F_NN) and
  // sixteen role-dependent values: the K tiles tn < rt (right operands of the update tiles left of the diagonal) in
  // creg[4 tn + s], this wave's rows of M = R^T from its diagonal block on in creg[4 tb + s], tb >= rt
  double kPm[4], bdP[4], creg[4 * NAT];
  {
    const double* lk = a.LK + pblk;
    const double* ys = a.ysc + pupd;
    const int m = 16 * rt + l15;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kk = kq + 4 * s;
      kPm[s] = (m < na && kk < nn) ? lk[(nn + m) + (int64_t)kk * nf] : 0.0;
      bdP[s] = (l15 < nn && kk <= l15) ? lk[l15 + (int64_t)kk * nf] : 0.0;
    }
#pragma unroll
    for (int x = 0; x < 4 * NAT; ++x) {
      const int t = x >> 2, kk = kq + 4 * (x & 3), mt = 16 * t + l15, kc = kq + 4 * x;
      creg[x] = t < rt ? ((mt < na && kk < nn) ? lk[(nn + mt) + (int64_t)kk * nf] : 0.0)
                       : ((m < na && kc < na && kc >= m) ? ys[kc + (int64_t)m * na] : 0.0);
    }
  }
  SETUP(1);
  // Children: each HALF of a wave owns one child (lanes 0..31: slot gw, lanes 32..63: slot gw + 4), so the two sweeps
  // run in the same instructions and every step of their LDS chains overlaps.
  const int hl = lane & 31, half = lane >> 5;
  const int cs = gw + 4 * half;
  const bool haschild = cs < nch;
  const int nnc = haschild ? hdr[16 + 6 * cs + 1] : 1, nac = haschild ? hdr[16 + 6 * cs + 2] : 0;
  const int nfc = nnc + nac, npan = haschild ? nfc * nnc : 0;
  const int64_t cblk = haschild ? ((int64_t)(uint32_t)hdr[16 + 6 * cs + 4] | ((int64_t)hdr[16 + 6 * cs + 5] << 32)) : 0;
  const int cbo = L.oCh + cs * C.cstride;                              // this child's constants in LDS
  int pdec[6];                                                         // first 192 panel positions: row | column << 16
#pragma unroll
  for (int x = 0; x < 6; ++x) { const int e = hl + 32 * x; pdec[x] = (e % nfc) | ((e / nfc) << 16); }
  const int npanmax = max(__builtin_amdgcn_readlane(npan, 0), __builtin_amdgcn_readlane(npan, 32));
  const int nacmax = max(__builtin_amdgcn_readlane(nac, 0), __builtin_amdgcn_readlane(nac, 32));

  double* const gb = smem + L.oGrp + grp * L.gstride;
  double* const sFnn = gb + L.gFnn;
  double* const sFan = gb + L.gFan;
  double* const sU = gb + L.gU;
  double* const sE = gb + L.gE;
  double* const sG = gb + L.gG;
  // position of (hi, lo), hi >= lo (rows of the parent's front), inside the group's front
  auto fpos = [&](int hi, int lo) -> int {
    if (lo >= nn) { const int m = hi - nn, n = lo - nn; return L.gU + n * na - ((n * (n - 1)) >> 1) + (m - n); }
    return hi >= nn ? L.gFan + (hi - nn) + lo * LDA : L.gFnn + hi + lo * LDN;
  };
  int target = 0;
  int relr = 0;

  int* const epfit = reinterpret_cast<int*>(smem + L.oCnt) + 2;   // passes of the running epoch whose entries are staged
  for (int q0 = 0; q0 < npass;) {
    // ===================================================================================================
    // epoch set-up (whole workgroup): the entry lists of the passes q0 .. q0 + ep - 1 -> LDS.  ep = as many of the next
    // tabpasses passes as fit the staging area (the host guarantees room for one pass at least), so the loop below
    // never reads an entry from global memory.
    // ===================================================================================================
    const int epmax = min(tabpasses, npass - q0), npairs = epmax * nmem;
    SETUP(2);
    __syncthreads();                                           // both groups are done with the previous epoch
    SETUP(3);
    if (q0 == 0) relr = (haschild && hl < nac) ? reinterpret_cast<const int*>(smem + cbo + C.cRel)[hl] : 0;
    if (tid == 0) *epfit = epmax;
    for (int e = tid; e < epmax; e += 512) nncnt[e] = 0;
    int myp0[2] = {0, 0};                                      // npairs <= 1024 (host): at most two pairs per thread
#pragma unroll
    for (int h = 0; h < 2; ++h) {                              // counts and global positions
      const int idx = tid + 512 * h;
      if (idx < npairs) {
        const int qq = idx / nmem, mem = idx - qq * nmem;
        const int r = (int)blockIdx.y + (q0 + qq) * gy;
        const int j = a.kc_ids ? a.kc_ids[a.kc_j0 + r] : a.kc_j0 + r;
        const int ck = mem ? hdr[16 + 6 * (mem - 1)] : k;
        const int32_t* kp = a.kc_ptr + (int64_t)ck * a.kc_stride;
        myp0[h] = kp[j];
        tab[2 * idx] = kp[j + 1] - myp0[h];
      }
    }
    SETUP(4);
    __syncthreads();
    if (wave == 0) {                                           // exclusive scan of the counts (one wave)
      const int per = (npairs + 63) / 64, b = lane * per;
      int sum = 0;
      for (int x = 0; x < per; ++x) sum += (b + x < npairs) ? tab[2 * (b + x)] : 0;
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
      int run = incl - sum;
      for (int x = 0; x < per; ++x)
        if (b + x < npairs) {
          const int c = tab[2 * (b + x)];
          tab[2 * (b + x) + 1] = run;
          run += c;
          if (run > ecap) atomicMin(epfit, (b + x) / nmem);    // this pass does not fit any more
        }
    }
    SETUP(5);
    __syncthreads();
    const int ep = max(1, *epfit);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int idx = tid + 512 * h;
      if (idx < ep * nmem) {
        const int c = tab[2 * idx], p0 = myp0[h], off = tab[2 * idx + 1];
        const int qq = idx / nmem, mem = idx - qq * nmem;
        const int mnn = mem ? hdr[16 + 6 * (mem - 1) + 1] : 0;
        const int* const mrel = reinterpret_cast<const int*>(smem + L.oCh + (mem - 1) * C.cstride + C.cRel);
        for (int t = 0; t < c && off + t < ecap; ++t) {
          const int ij = kc_ij[p0 + t];
          const int i = ij & 0xffff, jc = ij >> 16;
          const double v = a.kc_val[p0 + t];
          int pk = i | (jc << 8);
          if (mem && i >= mnn) pk |= mrel[i - mnn] << 16;
          else if (mem) {                                      // supernode-block entry of a child: shared update
            const int slot = atomicAdd(&nncnt[qq], 1);
            if (slot < FAM2_NNCAP) {
              nnpk[qq * FAM2_NNCAP + slot] = (mem - 1) | (i << 8) | (jc << 16);
              nnv[qq * FAM2_NNCAP + slot] = v;
              pk |= FAM2_COOP;
            }
          }
          lpk[off + t] = pk;
          lval[off + t] = v;
        }
      }
    }
    // vmcnt(0) through the BUILTIN: the compiler's wait-count pass then knows that every load of the set-up has
    // landed.  With an opaque asm statement it still counts them as pending at the head of the loop below and puts an
    // s_waitcnt vmcnt(0) there -- which, the counter being in order, waits for the panel stores of the previous pass.
    SETUP(6);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    SETUP(7);
    STAMP(0);

    if (grp == 1 && q0 == 0 && !(a.skip & 16))
      for (int z = 0; z < a.dn; ++z) __builtin_amdgcn_s_sleep(16);    // stagger: a.dn x ~1000 cycles
    for (int qq = grp; qq < ep; qq += 2) {
      const int r = (int)blockIdx.y + (q0 + qq) * gy;
      const int* const trow = tab + 2 * qq * nmem;
      // =================================================================================================
      // children (half-waves).  Branch-free per entry: every LDS operand of an entry (K column, twelve factors of the
      // six panel positions of the lane) is requested in one batch at clamped addresses, then selected.
      // =================================================================================================
      if (!(a.skip & 4)) {
        const int ne = haschild ? trow[2 * (1 + cs)] : 0;
        const int where = haschild ? trow[2 * (1 + cs) + 1] : 0;
        const int nemax = max(__builtin_amdgcn_readlane(ne, 0), __builtin_amdgcn_readlane(ne, 32));
        if constexpr (!CHP) {
          // updates only: an entry v at (separator row iA, column jc) sends -v (K[:, jc] e_iA^T + e_iA K[:, jc]^T) to the
          // parent's front (one LDS atomic per separator row, lane hl = row), an entry in the supernode block beyond the
          // shared list the rank-2 update of the whole separator block
          for (int t = 0; t < nemax; ++t) {
            const bool valid = t < ne;
            const int pk = lpk[valid ? where + t : 0];
            const double v = lval[valid ? where + t : 0];
            const int i = pk & 0xff, jc = valid ? ((pk >> 8) & 0xff) : 0, relA = (pk >> 16) & 0xff;
            const bool isAN = valid && i >= nnc;
            const int iA = isAN ? i - nnc : 0;
            const double kv = smem[cbo + C.cK + jc * csa + hl];
            if (isAN && hl < nac) {
              const int hi = max(relr, relA), lo = min(relr, relA);
              const int m2 = hi - nn, n2 = lo - nn;
              const int pU = L.gU + n2 * na - ((n2 * (n2 - 1)) >> 1) + (m2 - n2);
              const int pL = hi >= nn ? L.gFan + m2 + lo * LDA : L.gFnn + hi + lo * LDN;
              unsafeAtomicAdd(&gb[lo >= nn ? pU : pL], -(hl == iA ? 2.0 : 1.0) * v * kv);
            }
            const bool isNN = valid && i < nnc;
            if (__builtin_amdgcn_ballot_w64(isNN && !(pk & FAM2_COOP))) {
              const double w = i == jc ? 0.5 * v : v;
              const int ic = isNN ? i : 0;
              const double* pK = smem + cbo + C.cK + ic * csa;
              const double* qK = smem + cbo + C.cK + jc * csa;
              const int* const crel = reinterpret_cast<const int*>(smem + cbo + C.cRel);
              for (int c = 0; c < nacmax; ++c) {
                const int rr2 = c + hl;
                if (isNN && !(pk & FAM2_COOP) && rr2 < nac)
                  unsafeAtomicAdd(&gb[fpos(crel[rr2], crel[c])], w * (pK[rr2] * qK[c] + qK[rr2] * pK[c]));
              }
            }
          }
        } else {
        double* const Pc = u + (int64_t)r * ldu + cblk;
        for (int e0 = 0; e0 < npanmax; e0 += 192) {
          double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
          int dec[6];
#pragma unroll
          for (int x = 0; x < 6; ++x) dec[x] = pdec[x];
          if (e0)
#pragma unroll
            for (int x = 0; x < 6; ++x) { const int e = e0 + hl + 32 * x; dec[x] = (e % nfc) | ((e / nfc) << 16); }
          for (int t = 0; t < nemax; ++t) {
            const bool valid = t < ne;
            const int pk = lpk[valid ? where + t : 0];
            const double v = lval[valid ? where + t : 0];
            const int i = pk & 0xff, jc = valid ? ((pk >> 8) & 0xff) : 0, relA = (pk >> 16) & 0xff;
            const bool isAN = valid && i >= nnc;
            const int iA = isAN ? i - nnc : 0;
            const double kv = smem[cbo + C.cK + jc * csa + hl];
            double rr[6], bb[6];
            const double* rrow = smem + cbo + C.cR + iA * (iA + 1) / 2;
            const double* bcol = smem + cbo + C.cLi + jc * 16;
#pragma unroll
            for (int x = 0; x < 6; ++x) {
              const int m = (dec[x] & 0xffff) - nnc, n = dec[x] >> 16;
              rr[x] = rrow[min(max(m, 0), iA)];
              bb[x] = bcol[min(n, 15)];
            }
            if (isAN && e0 == 0 && hl < nac) {                   // update: row / column iA of the child's separator block
              const int hi = max(relr, relA), lo = min(relr, relA);
              const int m2 = hi - nn, n2 = lo - nn;
              const int pU = L.gU + n2 * na - ((n2 * (n2 - 1)) >> 1) + (m2 - n2);
              const int pL = hi >= nn ? L.gFan + m2 + lo * LDA : L.gFnn + hi + lo * LDN;
              unsafeAtomicAdd(&gb[lo >= nn ? pU : pL], -(hl == iA ? 2.0 : 1.0) * v * kv);
            }
#pragma unroll
            for (int x = 0; x < 6; ++x) {
              const int m = (dec[x] & 0xffff) - nnc, n = dec[x] >> 16;
              acc[x] += (isAN && m >= 0 && m <= iA && n < nnc) ? v * rr[x] * bb[x] : 0.0;
            }
            const bool isNN = valid && i < nnc;
            if (__builtin_amdgcn_ballot_w64(isNN)) {             // entries in the supernode block: rare
              const double w = i == jc ? 0.5 * v : v;
              const int ic = isNN ? i : 0;
              const double* pK = smem + cbo + C.cK + ic * csa;
              const double* qK = smem + cbo + C.cK + jc * csa;
              if (e0 == 0 && __builtin_amdgcn_ballot_w64(isNN && !(pk & FAM2_COOP))) {      // beyond the shared list
                const int* const crel = reinterpret_cast<const int*>(smem + cbo + C.cRel);
                for (int c = 0; c < nacmax; ++c) {
                  const int rr2 = c + hl;
                  if (isNN && !(pk & FAM2_COOP) && rr2 < nac)
                    unsafeAtomicAdd(&gb[fpos(crel[rr2], crel[c])], w * (pK[rr2] * qK[c] + qK[rr2] * pK[c]));
                }
              }
              if (isNN) {
                const double* acol = smem + cbo + C.cLi + ic * 16;
                const double* mp = smem + cbo + C.cMK + ic * csa;
                const double* mq = smem + cbo + C.cMK + jc * csa;
#pragma unroll
                for (int x = 0; x < 6; ++x) {
                  const int ie = dec[x] & 0xffff, n = dec[x] >> 16;
                  if (n < nnc) {
                    if (ie < nnc) { if (ie >= n) acc[x] += w * (acol[ie] * bcol[n] + bcol[ie] * acol[n]); }
                    else { const int m = ie - nnc; acc[x] -= w * (mp[m] * bcol[n] + mq[m] * acol[n]); }
                  }
                }
              }
            }
          }
#pragma unroll
          for (int x = 0; x < 6; ++x) { const int e = e0 + hl + 32 * x; if (e < npan && !(a.skip & 1)) FAM2_ST(&Pc[e], acc[x]); }
        }
      }
        }
      STAMP(7);
      // supernode-block entries of the children, shared: wave gw takes a quarter of the na (na + 1) / 2 positions of
      // the update v (p q^T + q p^T) (halved for i == j), position p = (row r, column c <= r) in row-major order
      if (!(a.skip & 4)) {
        const int nnn = min(__builtin_amdgcn_readfirstlane(nncnt[qq]), FAM2_NNCAP);
        for (int it = 0; it < nnn; ++it) {
          const int pkk = __builtin_amdgcn_readfirstlane(nnpk[qq * FAM2_NNCAP + it]);
          const double v = nnv[qq * FAM2_NNCAP + it];
          const int c2 = pkk & 0xff, i = (pkk >> 8) & 0xff, jc = pkk >> 16;
          const int nac2 = __builtin_amdgcn_readfirstlane(cnac[c2]);
          const double* const cb2 = smem + L.oCh + c2 * C.cstride;
          const int* const crel = reinterpret_cast<const int*>(cb2 + C.cRel);
          const double* pK = cb2 + C.cK + i * csa;
          const double* qK = cb2 + C.cK + jc * csa;
          const double w = i == jc ? 0.5 * v : v;
          const int np = nac2 * (nac2 + 1) / 2, per = (np + 3) / 4, pend = min(np, (gw + 1) * per);
          for (int p = gw * per + lane; p < pend; p += 64) {
            int r2 = (int)((__fsqrt_rn(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
            if (r2 * (r2 + 1) / 2 > p) --r2;
            else if ((r2 + 1) * (r2 + 2) / 2 <= p) ++r2;
            const int c = p - r2 * (r2 + 1) / 2;
            unsafeAtomicAdd(&gb[fpos(crel[r2], crel[c])], w * (pK[r2] * qK[c] + qK[r2] * pK[c]));
          }
        }
      }
      // the parent's own entries (role 1: the lightest sweep load; role 0 when there is no second row tile)
      if (rt == (na > 16 ? 1 : 0)) {
        const int ne = __builtin_amdgcn_readfirstlane(trow[0]);
        const int where = __builtin_amdgcn_readfirstlane(trow[1]);
        for (int t = lane; t < ne; t += 64) {
          const int pk = lpk[where + t];
          const double v = lval[where + t];
          const int i = pk & 0xff, n = (pk >> 8) & 0xff;
          if (i >= nn) unsafeAtomicAdd(&sFan[(i - nn) + n * LDA], v);
          else if (i >= n) unsafeAtomicAdd(&sFnn[i + n * LDN], v);
        }
      }
      STAMP(1);
      fam2_barrier(cnt, target, lane, a.t.info);
      STAMP(2);
      // =================================================================================================
      // A: E, G of row tile rt (T, G_NN by role 0)
      // =================================================================================================
      double* const P = u + (int64_t)r * ldu + pblk;
      const int m = 16 * rt + l15;
      const bool active = (16 * rt < na || rt == 0) && !(a.skip & 8);        // role 0 also forms T and G_NN (a root parent has na = 0)
      double evm[4] = {0.0, 0.0, 0.0, 0.0};
      if (active) {
        double fnn[4], fan[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int kr = kq + 4 * s;
          fnn[s] = sFnn[kr >= l15 ? kr + l15 * LDN : l15 + kr * LDN];
          fan[s] = sFan[m + kr * LDA];
        }
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KSN; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fnn[s], kPm[s], acc, 0, 0, 0);
        double xv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          evm[s] = fan[s] - 0.5 * acc[s];
          xv[s] = fan[s] - acc[s];
          sE[m + (kq + 4 * s) * LDA] = evm[s];
          sFan[m + (kq + 4 * s) * LDA] = 0.0;
        }
        d4 g = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KSN; ++s) g = __builtin_amdgcn_mfma_f64_16x16x4f64(bdP[s], xv[s], g, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) sG[m + (kq + 4 * s) * LDA] = g[s];
        if (rt == 0) {
          d4 tt = {0.0, 0.0, 0.0, 0.0}, gn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < KSN; ++s) tt = __builtin_amdgcn_mfma_f64_16x16x4f64(fnn[s], bdP[s], tt, 0, 0, 0);
#pragma unroll
          for (int s = 0; s < KSN; ++s) gn = __builtin_amdgcn_mfma_f64_16x16x4f64(bdP[s], tt[s], gn, 0, 0, 0);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int jn = kq + 4 * s;
            if (l15 < nn && jn <= l15 && !(a.skip & 2)) FAM2_ST(&P[l15 + (int64_t)jn * nf], gn[s]);
          }
        }
      }
      STAMP(3);
      fam2_barrier(cnt, target, lane, a.t.info);
      STAMP(4);
      // =================================================================================================
      // B: update tiles (rt, tn <= rt) and Q of row tile rt -- straight-line code per role: every LDS operand of the
      // phase is requested before the first product
      // =================================================================================================
      if (rt == 0)
        for (int e = lane; e < LDN * 16; e += 64) sFnn[e] = 0.0;
      double* const UkP = a.t.updp + (int64_t)r * a.t.updplen + pupdp;
      auto phaseB = [&](auto RTc) {
        constexpr int RT = decltype(RTc)::value;
#pragma unroll
        for (int tn = 0; tn <= RT; ++tn) {
          double eB[4], uv[4];
          int po[4];
          if (tn < RT)
#pragma unroll
            for (int s = 0; s < KSN; ++s) eB[s] = sE[(16 * tn + l15) + (kq + 4 * s) * LDA];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int n = 16 * tn + kq + 4 * s;
            const bool ok = m >= n && m < na;
            po[s] = ok ? n * na - ((n * (n - 1)) >> 1) + (m - n) : -1;
            uv[s] = ok ? sU[po[s]] : 0.0;
          }
          d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < KSN; ++s) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tn < RT ? eB[s] : evm[s], kPm[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tn < RT ? creg[4 * tn + s] : kPm[s], evm[s], acc, 0, 0, 0);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s)
            if (po[s] >= 0) {
              sU[po[s]] = 0.0;
              if (!(a.skip & 2)) FAM2_ST(&UkP[po[s]], uv[s] - acc[s]);
            }
        }
        // Q = R^T G: row tile RT; R^T is zero left of its diagonal block
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int tb = RT; tb < NAT; ++tb) {
          double gv[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) gv[s] = sG[(kq + 4 * (4 * tb + s)) + l15 * LDA];
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(gv[s], creg[4 * tb + s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int n = kq + 4 * s;
          if (m < na && n < nn && !(a.skip & 2)) FAM2_ST(&P[(nn + m) + (int64_t)n * nf], acc[s]);
        }
      };
      if (active) {
        if (rt == 0) phaseB(std::integral_constant<int, 0>{});
        if constexpr (NAT > 1) { if (rt == 1) phaseB(std::integral_constant<int, 1>{}); }
        if constexpr (NAT > 2) { if (rt == 2) phaseB(std::integral_constant<int, 2>{}); }
        if constexpr (NAT > 3) { if (rt == 3) phaseB(std::integral_constant<int, 3>{}); }
      }
      STAMP(5);
      fam2_barrier(cnt, target, lane, a.t.info);
      STAMP(6);
    }
    q0 += ep;
  }
#ifdef SMCP_STAMPS
  if (stamp && rt != 3) for (int i = 0; i < 8; ++i) atomicAdd(a.dbg + 8 * rt + i, tph[i]);
  if (stamp && rt == 0) for (int i = 0; i < 8; ++i) atomicAdd(a.dbg + 24 + i, tset[i]);
#endif
#undef STAMP
#undef SETUP
}


}  // namespace smcp
