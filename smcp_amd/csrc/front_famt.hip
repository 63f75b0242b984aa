// Entry-driven family sweep (round 3): the Schur-complement sweep of a small parent front (nn <= 16, na <= 64) with
// its <= 8 childless children, for the right-hand sides whose children's panels are NOT formed (their Gram block comes
// from front_leafgram.hip).  Replaces the LDS front, the atomics and the three barrier-separated phases of
// k_fam_sparse (front_fam2.hip) by ONE set of independent rank-T products per (parent, right-hand side):
//
//   The front of a right-hand side is a short sum of symmetric dyads,  F = sum_t s_t (x_t y_t^T + y_t x_t^T):
//     an own entry v at (i, j) of the parent's panel:        x = e_i, y = e_j,                 s = v  (v / 2 if i == j)
//     an entry v at (separator row a, column j) of child c:  x = q~_{c,j}, y = e_{rel_c[a]},   s = -v
//     an entry v at (i, j) of the supernode block of c:      x = q~_{c,i}, y = q~_{c,j},       s = v  (v / 2 if i == j)
//   with q~_{c,j} = column j of K_c scattered to the parent's front rows (the closed forms of front_fam2.hip, fact 1).
//   The sweep is linear in F:  with R_N = [Li 0], R_A = [-K I] (reference call site solvers.py:483
//   through the Gram formulation of solvers.py:414-420)
//     G_NN = R_N F R_N^T,   Q = R^T (R_A F R_N^T),   Upd = R_A F R_A^T,
//   so with the images n(x) = R_N x, a(x) = R_A x, m(x) = R^T a(x) of the few distinct vectors x -- columns of Li, -K,
//   -R^T K, R^T and unit vectors for the e_i, and per (child, column) constants for the q~ -- every output is
//     G_NN = sum_t' s n(x') n(y')^T,   Q = sum_t' s m(x') n(y')^T,   Upd = sum_t' s a(x') a(y')^T
//   over the 2 T ordered pairs t' = (x, y), (y, x): products with inner dimension 2 T ~ 26 whose operands are GATHERED
//   from tables in LDS.  One wave owns a (parent, right-hand side) pair: per step of four t' it reads 14 operand values
//   and issues 15 independent v_mfma_f64_16x16x4 (ten update tiles, four Q tiles, G_NN); the accumulators go straight
//   to HBM.  No front in LDS, no atomics, no dependent product chain, no barrier after the set-up.
//
// k_famt_prep lays the tables of every family out once per sweep call (the factor may have changed); k_fam_terms copies
// them to LDS, stages the entry lists of its right-hand sides there (as k_fam_sparse does: the steady-state loop issues
// no vector load, which would wait behind the streaming stores -- vmcnt is one in-order counter) and sweeps.
#include <hip/hip_runtime.h>

namespace smcp {

// tables of a family in LDS / in its global record (doubles): Li (16 x 16) | -K (NA x 16) | -R^T K (NA x 16) |
// R^T (NA x NA, upper triangular, column-major) | NA zeros | per child column: n (16) | a (NA) | m (NA)
struct FamtL { int oLi, onK, onMK, oRt, oZero, oCN, oCA, oCM, total; };
template <int NAT>
__host__ __device__ inline FamtL famt_layout(int ncol) {
  constexpr int NA = 16 * NAT;
  FamtL L{};
  int o = 0;
  L.oLi = o; o += 256;
  L.onK = o; o += NA * 16;
  L.onMK = o; o += NA * 16;
  L.oRt = o; o += NA * NA;
  L.oZero = o; o += NA;
  L.oCN = o; o += ncol * 16;
  L.oCA = o; o += ncol * NA;
  L.oCM = o; o += ncol * NA;
  L.total = o;
  return L;
}
// Results leave through raw buffer stores: an element that is not to be written (above the diagonal of a diagonal
// tile, beyond na / nn) gets the byte offset 0xffffffff, which the hardware range check of the buffer drops -- one
// v_cndmask per store instead of an exec-mask region with its branch (the per-element if () form compiled to ~8
// instructions per store, masks spilled to VGPR lanes, and made the kernel's run time depend on where its code
// landed in memory: 0.86 to 1.05 ms for the same instructions at four 256-byte offsets, scratch/famt_place.sh).
typedef unsigned int famt_u2 __attribute__((ext_vector_type(2)));
__device__ inline __amdgpu_buffer_rsrc_t famt_rsrc(double* base, int doubles) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, doubles * 8, 0x00020000);      // raw buffer, 32-bit data format (gfx9)
}
#ifndef SMCP_FAMT_AUX
#define SMCP_FAMT_AUX 0          // cache-policy bits of the result stores (2 = nt: streamed once, read ~1 ms later by other kernels)
#endif
__device__ inline void famt_store(__amdgpu_buffer_rsrc_t r, bool ok, int pos, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(famt_u2, v), r, ok ? pos * 8 : -1, 0, SMCP_FAMT_AUX);
}
#ifndef SMCP_FAMT_NW
#define SMCP_FAMT_NW 12
#endif
// waves of a k_fam_terms workgroup (one workgroup per CU: the tables fill LDS).  Round 3, when the kernel formed the update tiles
// as well (15 accumulator tiles): 8 = two per SIMD; 12 (168 registers, three passes of at most six tiles) 0.82 against 0.80 ms.
// Since the fused extend-add took the update tiles (round 4) the headline variant holds five tiles in 115 registers and is a
// chain of LDS gathers: 12 waves 0.345 against 0.39 ms per Schur sweep of synth50k, 16 waves the same as 12 (round 5)
constexpr int FAMT_NW = SMCP_FAMT_NW;
constexpr int FAMT_HDR = 32;        // doubles: the header of a record (ints, as FAM2: [0] clique, [1] nn, [2] na, [3] children,
                                    // [4,5] panel offset, [8,9] packed-update offset; child c at 16 + 6 c: clique, nn, na,
                                    // first column in the child tables, -, -)
constexpr int FAMT_TCAP = 96;       // ordered pairs t' per (parent, right-hand side): 48 entries (host-checked)
constexpr int FAMT_TMAX = 384;      // longest (family, constraint) list the entry-driven sweeps take (in chunks of FAMT_TCAP / 2; host-checked)
constexpr int FAMT_CHILD = 128;     // vector ids: < 128 unit vector e_id of the parent's front, >= 128 child column id - 128

// LDS of k_famt_prep (doubles): R^T | -K | Li | per wave q~ (16 + NA) and the a images of its child's columns (cnn x NA)
template <int NAT>
__host__ __device__ inline int famt_prep_doubles(int cnn) { constexpr int NA = 16 * NAT; return NA * NA + NA * 16 + 256 + 8 * (NA + 16 + cnn * NA); }

// One workgroup per family, wave w = child w.  Only what later steps of the kernel read again is kept in LDS (R^T, -K, Li:
// 42 KB for NA = 64 + 9 KB per wave), the tables themselves go straight to the record -- with the whole record staged in
// LDS (136 KB, one workgroup per CU) the chain of dependent global loads of every workgroup was exposed and the launch
// took 0.16 ms on synth50k.
template <int NAT>
__global__ void __launch_bounds__(512) k_famt_prep(MfmaArgs a, double* famt, int cnn, int32_t* fz_slot) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NA = 16 * NAT;
  const int ncol = 8 * cnn;
  const FamtL L = famt_layout<NAT>(ncol);
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na, nch = d.chend - d.chbeg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* const sRt = smem;                    // R^T[p][q] at p + q NA (upper triangular)
  double* const snK = sRt + NA * NA;           // -K[p][c] at p + c NA
  double* const sLi = snK + NA * 16;           // Li[i][j] at i + j 16 (lower)
  double* const qt = sLi + 256 + wave * (NA + 16 + cnn * NA);     // q~: rows of the parent's front (16 then NA)
  double* const sCA = qt + NA + 16;                                // a images of this wave's columns: [j][p]
  double* const out = famt + (int64_t)blockIdx.x * (FAMT_HDR + L.total);
  double* const T = out + FAMT_HDR;
  // (the children's descriptors in one round trip: lane c loads child c, the column bases come from a wave scan)
  const int myck = lane < nch ? a.t.chidx[d.chbeg + lane] : 0;
  const int mynn = lane < nch ? a.t.cl[myck].nn : 0;
  int inc = mynn;
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
  const int colbase = __shfl(inc - mynn, min(wave, 7));
  const int ck = __shfl(myck, min(wave, 7));
  for (int e = tid; e < NA * NA + NA * 16 + 256; e += 512) smem[e] = 0.0;
  __syncthreads();
  {
    const double* lk = a.LK + d.blk;
    const double* ys = a.ysc + d.upd;                  // R: lower, column-major na x na
    for (int e = tid; e < nf * nn; e += 512) {
      const int i = e % nf, j = e / nf;
      if (i >= nn) snK[(i - nn) + j * NA] = -lk[e];
      else if (i >= j) sLi[i + j * 16] = lk[e];
    }
    batched_loop<8>(tid, na * na, 512, [=](int e) { return ys[e]; },
                    [=](int e, double v) {
                      const int i = e % na, p = e / na;   // R[i][p], i >= p  ->  R^T[p][i]
                      if (i >= p) sRt[p + i * NA] = v;
                    });
  }
  __syncthreads();
  // the parent's own tables -> record (zero-padded to the fixed strides)
  for (int e = tid; e < 256; e += 512) T[L.oLi + e] = sLi[e];
  for (int e = tid; e < NA * 16; e += 512) T[L.onK + e] = snK[e];
  for (int e = tid; e < NA * NA; e += 512) T[L.oRt + e] = sRt[e];
  for (int e = tid; e < NA; e += 512) T[L.oZero + e] = 0.0;
  for (int e = tid; e < NA * 16; e += 512) {           // -(R^T K)[p][c] = sum_{q >= p} R^T[p][q] (-K)[q][c]
    const int p = e % NA, c = e / NA;
    double s = 0.0;
    if (p < na && c < nn)
      for (int q = p; q < na; ++q) s += sRt[p + q * NA] * snK[q + c * NA];
    T[L.onMK + e] = s;
  }
  // children: wave w = child w; columns of absent children stay zero
  if (wave < nch) {
    const CliqueDesc cd = a.t.cl[ck];
    const int nnc = cd.nn, nac = cd.na, nfc = nnc + nac;
    const double* lkc = a.LK + cd.blk;
    const int rel = lane < nac ? a.t.relidx[cd.rel + lane] : 0;
    for (int j = 0; j < nnc; ++j) {
      const double kv = lane < nac ? lkc[(nnc + lane) + (int64_t)j * nfc] : 0.0;     // K_c[lane][j]
      for (int e = lane; e < NA + 16; e += 64) qt[e] = 0.0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < nac) qt[rel < nn ? rel : 16 + (rel - nn)] = kv;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int g = colbase + j;
      // n = Li q~_N ; a = q~_A + (-K) q~_N
      if (lane < 16) {
        double s = 0.0;
        if (lane < nn)
          for (int q = 0; q <= lane; ++q) s += sLi[lane + q * 16] * qt[q];
        T[L.oCN + g * 16 + lane] = s;
      }
      for (int p = lane; p < NA; p += 64) {
        double s = 0.0;
        if (p < na) {
          s = qt[16 + p];
          for (int q = 0; q < nn; ++q) s += snK[p + q * NA] * qt[q];
        }
        sCA[j * NA + p] = s;
        T[L.oCA + g * NA + p] = s;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // m = R^T a for the columns of this child: lane = row p, R^T[p][q] read once per q, the a values by broadcast
    for (int j0 = 0; j0 < nnc; j0 += 8)
      for (int p = lane; p < NA; p += 64) {
        double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (p < na)
          for (int q = p; q < na; ++q) {
            const double rt = sRt[p + q * NA];
#pragma unroll
            for (int x = 0; x < 8; ++x) s8[x] += rt * sCA[min(j0 + x, cnn - 1) * NA + q];
          }
#pragma unroll
        for (int x = 0; x < 8; ++x) if (j0 + x < nnc) T[L.oCM + (colbase + j0 + x) * NA + p] = s8[x];
      }
    if (lane == 0) {
      int* const hdr = reinterpret_cast<int*>(out) + 16 + 6 * wave;
      hdr[0] = ck; hdr[1] = nnc; hdr[2] = nac; hdr[3] = colbase; hdr[4] = 0; hdr[5] = 0;
    }
  } else if (wave < 8 && lane == 0) {
    int* const hdr = reinterpret_cast<int*>(out) + 16 + 6 * wave;
    hdr[0] = -1; hdr[1] = 1; hdr[2] = 0; hdr[3] = 0; hdr[4] = 0; hdr[5] = 0;
  }
  // table columns no child owns (fewer than 8 cnn columns in all): zero
  {
    const int used = __shfl(inc, 7);                    // columns of all children
    for (int e = tid; e < (ncol - used) * 16; e += 512) T[L.oCN + used * 16 + e] = 0.0;
    for (int e = tid; e < (ncol - used) * NA; e += 512) { T[L.oCA + used * NA + e] = 0.0; T[L.oCM + used * NA + e] = 0.0; }
  }
  if (tid == 0) {
    if (fz_slot && a.fz_no && a.fz_no[k] >= 0) fz_slot[a.fz_no[k]] = (int32_t)blockIdx.x;      // where the fused extend-add finds this family's tables
    int* const hdr = reinterpret_cast<int*>(out);
    hdr[0] = k; hdr[1] = nn; hdr[2] = na; hdr[3] = nch;
    hdr[4] = (int)(d.blk & 0xffffffffll); hdr[5] = (int)(d.blk >> 32);
    hdr[6] = 0; hdr[7] = 0;
    hdr[8] = (int)(d.updp & 0xffffffffll); hdr[9] = (int)(d.updp >> 32);
  }
}

// The record of a family -> LDS: sixteen-byte loads, eight in flight per thread (the plain copy loop compiled to one load,
// s_waitcnt vmcnt(0), one LDS store per trip: 24 dependent round trips per workgroup, ~20 us of every workgroup's life)
template <int NTH>
__device__ inline void famt_copy_tables(double* smem, const double* src, int total, int tid) {
  const double2* const s2 = reinterpret_cast<const double2*>(src);
  double2* const d2 = reinterpret_cast<double2*>(smem);
  const int n2 = total >> 1;                                  // every table has an even number of doubles
  for (int e0 = tid; e0 < n2; e0 += NTH * 8) {
    double2 v[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) v[x] = s2[min(e0 + x * NTH, n2 - 1)];
#pragma unroll
    for (int x = 0; x < 8; ++x) if (e0 + x * NTH < n2) d2[e0 + x * NTH] = v[x];
  }
}

// LDS of k_fam_terms behind the tables (doubles): per wave the descriptors of FAMT_TCAP ordered pairs (scale: 1 double,
// offsets: 4 ints), then the entry table of k_fam_sparse: per (pass, member) two ints, then the staged entries
__host__ __device__ inline int famt_desc_doubles() { return FAMT_TCAP * 3; }

// UPD = false: the parents' update matrices are not formed here -- the extend-add of their parent front computes them into the
// front it holds in LDS (lf_add_family, front_large.hip); one pass with the Q and G_NN tiles only
template <int NAT, bool UPD = true>
__global__ void __launch_bounds__(64 * FAMT_NW) k_fam_terms(MfmaArgs a, double* u, int64_t ldu, const double* famt, int cnn,
                                                   const int32_t* kc_ij, int tabpasses, int ecap) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NA = 16 * NAT, NW = FAMT_NW, NTH = 64 * NW;
  const int ncol = 8 * cnn;
  const FamtL L = famt_layout<NAT>(ncol);
  const double* const fc = famt + (int64_t)blockIdx.x * (FAMT_HDR + L.total);
  const int32_t* const hdr = reinterpret_cast<const int32_t*>(fc);
  const int k = hdr[0], nn = hdr[1], na = hdr[2], nch = hdr[3], nf = nn + na;
  const int64_t pblk = (int64_t)(uint32_t)hdr[4] | ((int64_t)hdr[5] << 32);
  const int64_t pupdp = (int64_t)(uint32_t)hdr[8] | ((int64_t)hdr[9] << 32);
  const int tid = threadIdx.x;
  const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nmem = nch + 1;                     // members with entry lists: 0 = the parent, 1 + c = child c
  const int gy = (int)gridDim.y;
  const int npass = ((int)a.nrhs - (int)blockIdx.y + gy - 1) / gy;
  double* const dS = smem + L.total + wave * famt_desc_doubles();              // scale of pair t'
  int* const dI = reinterpret_cast<int*>(dS + FAMT_TCAP);                       // four words per pair
  int* const misc = reinterpret_cast<int*>(smem + L.total + NW * famt_desc_doubles());
  int* const epfit = misc;                                                     // passes of the running epoch whose entries are staged
  int* const cdim = misc + 2;                                                  // per child: nn | colbase << 8
  int* const tab = misc + 12;                                                  // [pass in epoch][member] -> (count, where)
  double* const lval = reinterpret_cast<double*>(tab + 2 * ((tabpasses * nmem + 1) & ~1));
  int* const lpk = reinterpret_cast<int*>(lval + ecap);

  famt_copy_tables<NTH>(smem, fc + FAMT_HDR, L.total, tid);
  if (tid < 8) cdim[tid] = tid < nch ? (hdr[16 + 6 * tid + 1] | (hdr[16 + 6 * tid + 3] << 8)) : 1;

  for (int q0 = 0; q0 < npass;) {
    // ===================================================================================================
    // epoch set-up (whole workgroup): the entry lists of the passes q0 .. q0 + ep - 1 -> LDS, as in k_fam_sparse
    // (word = row | column << 8 | parent front row of a child's separator row << 16)
    // ===================================================================================================
    const int epmax = min(tabpasses, npass - q0), npairs = epmax * nmem;
    __syncthreads();
    if (tid == 0) *epfit = epmax;
    int myp0[2] = {0, 0};                                      // npairs <= 1024 (host): at most two pairs per thread
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int idx = tid + NTH * h;
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
    __syncthreads();
    const int ep = max(1, *epfit);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int idx = tid + NTH * h;
      if (idx < ep * nmem) {
        const int c = tab[2 * idx], p0 = myp0[h], off = tab[2 * idx + 1];
        const int qq = idx / nmem, mem = idx - qq * nmem;
        (void)qq;
        const int mnn = mem ? hdr[16 + 6 * (mem - 1) + 1] : 0;
        const int mck = mem ? hdr[16 + 6 * (mem - 1)] : 0;
        const int32_t* const mrel = a.t.relidx + a.t.cl[mem ? mck : k].rel;
        for (int t = 0; t < c && off + t < ecap; ++t) {
          const int ij = kc_ij[p0 + t];
          const int i = ij & 0xffff, jc = ij >> 16;
          int pk = i | (jc << 8);
          if (mem && i >= mnn) pk |= mrel[i - mnn] << 16;
          lpk[off + t] = pk;
          lval[off + t] = a.kc_val[p0 + t];
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) the compiler's wait-count pass knows about (see k_fam_sparse)
    __syncthreads();

    for (int qq = wave; qq < ep; qq += NW) {
      const int r = (int)blockIdx.y + (q0 + qq) * gy;
      const int* const trow = tab + 2 * qq * nmem;
      // =================================================================================================
      // the ordered pairs of this right-hand side: lane e <-> entry e of the concatenated member lists
      // =================================================================================================
      int cntm = lane < nmem ? trow[2 * lane] : 0;
      int incl = cntm;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
      // (lists of more than FAMT_TCAP / 2 entries -- a constraint with a long list among short ones, e.g. a multiple of the
      // identity: 15 + 8 x 5 diagonal entries per family -- go through the descriptor area in chunks of that many; the
      // accumulators of a pass persist across the chunks)
      const int Ttot = min(__builtin_amdgcn_readlane(incl, 15), FAMT_TMAX);
      const int ndp = max(1, (Ttot + FAMT_TCAP / 2 - 1) / (FAMT_TCAP / 2));
      auto build = [&](int dp) {
        const int e = dp * (FAMT_TCAP / 2) + lane;              // this lane's entry of the concatenated member lists
        const int T = min(FAMT_TCAP / 2, Ttot - dp * (FAMT_TCAP / 2));
        int mem = 0, base = 0;
#pragma unroll
        for (int mm = 0; mm < 8; ++mm) {
          const int up = __builtin_amdgcn_readlane(incl, mm);
          if (e >= up) { mem = mm + 1; base = up; }
        }
        if (lane < T) {
          const int where = trow[2 * mem + 1];
          const int pk = lpk[where + (e - base)];
          const double v = lval[where + (e - base)];
          const int i = pk & 0xff, jc = (pk >> 8) & 0xff, rl = (pk >> 16) & 0xff;
          int vx, vy;
          double s;
          if (mem == 0) { vx = i; vy = jc; s = i == jc ? 0.5 * v : v; }
          else {
            const int cd = cdim[mem - 1], nnc = cd & 0xff, cb = cd >> 8;
            if (i >= nnc) { vx = FAMT_CHILD + cb + jc; vy = rl; s = -v; }
            else { vx = FAMT_CHILD + cb + i; vy = FAMT_CHILD + cb + jc; s = i == jc ? 0.5 * v : v; }
          }
          // images of a vector id: offsets of n, a, m in LDS and the index of the unit part of a (255: none)
          auto img = [&](int vid, int& on, int& oa, int& om, int& ui) {
            if (vid >= FAMT_CHILD) { const int g = vid - FAMT_CHILD; on = L.oCN + 16 * g; oa = L.oCA + NA * g; om = L.oCM + NA * g; ui = 255; }
            else if (vid < nn) { on = L.oLi + 16 * vid; oa = L.onK + NA * vid; om = L.onMK + NA * vid; ui = 255; }
            else { on = L.oZero; oa = L.oZero; om = L.oRt + NA * (vid - nn); ui = vid - nn; }
          };
          int nx, ax, mx, ux, ny, ay, my, uy;
          img(vx, nx, ax, mx, ux);
          img(vy, ny, ay, my, uy);
          dS[2 * lane] = s; dS[2 * lane + 1] = s;
          int4 w0 = {nx | (ax << 16), mx | (ux << 16), ny | (ay << 16), uy};
          int4 w1 = {ny | (ay << 16), my | (uy << 16), nx | (ax << 16), ux};
          reinterpret_cast<int4*>(dI)[2 * lane] = w0;
          reinterpret_cast<int4*>(dI)[2 * lane + 1] = w1;
        }
        // pad to a multiple of four pairs
        if (lane < 4 && 2 * T + lane < ((2 * T + 3) & ~3)) {
          dS[2 * T + lane] = 0.0;
          int4 z = {L.oZero | (L.oZero << 16), L.oZero | (255 << 16), L.oZero | (L.oZero << 16), 255};
          reinterpret_cast<int4*>(dI)[2 * T + lane] = z;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      };
      build(0);
      // =================================================================================================
      // products: step s takes the pairs t' = kq + 4 s.  Accumulator register x of lane (l15, kq) of a tile holds
      // (row l15 of the b operand's tile, column kq + 4 x of the a operand's tile)
      // =================================================================================================
      // Row tiles in two passes when there are four of them (rows 0..2, then row 3 + G_NN): 15 live accumulator tiles
      // plus the operands of a step exceed the 256 registers of a wave at two waves per SIMD (measured: 35 spilled
      // registers, i.e. scratch loads behind the streaming stores); the operands are re-read from LDS, which is cheap.
      const __amdgpu_buffer_rsrc_t rP = famt_rsrc(u + (int64_t)r * ldu + pblk, nf * nn);
      const __amdgpu_buffer_rsrc_t rU = famt_rsrc(a.t.updp + (int64_t)r * a.t.updplen + pupdp, (na * (na + 1)) >> 1);
      bool fresh = true;                     // the descriptor area holds chunk 0
      auto pass = [&](auto LOc, auto HIc, auto Gc) {
        constexpr int LO = decltype(LOc)::value, HI = decltype(HIc)::value;      // row tiles LO .. HI - 1
        constexpr bool WITHG = decltype(Gc)::value;
        constexpr int NTU = UPD ? HI * (HI + 1) / 2 - LO * (LO + 1) / 2 : 1;
        d4 accU[NTU], accQ[HI - LO], accG = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int x = 0; x < NTU; ++x) accU[x] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int x = 0; x < HI - LO; ++x) accQ[x] = d4{0.0, 0.0, 0.0, 0.0};
        for (int dp = 0; dp < ndp; ++dp) {
        if (!(fresh && dp == 0)) {           // (every lane has read the previous chunk's descriptors)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          build(dp);
        }
        fresh = ndp == 1;
        const int ks = (2 * min(FAMT_TCAP / 2, Ttot - dp * (FAMT_TCAP / 2)) + 3) >> 2;
        for (int s = 0; s < ks; ++s) {
          const int tp = kq + 4 * s;
          const double sc = dS[tp];
          const int4 w = reinterpret_cast<const int4*>(dI)[tp];
          const int nx = w.x & 0xffff, ax = w.x >> 16, mx = w.y & 0xffff, ux = w.y >> 16;
          const int ny = w.z & 0xffff, ay = w.z >> 16, uy = w.w;
          double bA[HI - LO], bM[HI - LO], aA[HI];
#pragma unroll
          for (int t = 0; t < HI; ++t) {
            const int row = 16 * t + l15;
            if constexpr (UPD) aA[t] = sc * (smem[ay + row] + (row == uy ? 1.0 : 0.0));
            if (t >= LO) {
              if constexpr (UPD) bA[t - LO] = smem[ax + row] + (row == ux ? 1.0 : 0.0);
              bM[t - LO] = smem[mx + row];
            }
          }
          const double aN = sc * smem[ny + l15];
#pragma unroll
          for (int rt = LO; rt < HI; ++rt) {
            if constexpr (UPD) {
#pragma unroll
              for (int ct = 0; ct <= rt; ++ct) {
                constexpr int base = LO * (LO + 1) / 2;
                const int x = rt * (rt + 1) / 2 + ct - base;
                accU[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(aA[ct], bA[rt - LO], accU[x], 0, 0, 0);
              }
            }
            accQ[rt - LO] = __builtin_amdgcn_mfma_f64_16x16x4f64(aN, bM[rt - LO], accQ[rt - LO], 0, 0, 0);
          }
          if constexpr (WITHG) {
            const double bN = smem[nx + l15];
            accG = __builtin_amdgcn_mfma_f64_16x16x4f64(aN, bN, accG, 0, 0, 0);
          }
        }
        }
        // results straight from the accumulators: packed update (column-major lower), Q and G_NN into the panel
#pragma unroll
        for (int rt = LO; rt < HI; ++rt) {
          const int m = 16 * rt + l15;
          const bool mok = m < na;
          if constexpr (UPD) {
#pragma unroll
            for (int ct = 0; ct <= rt; ++ct)
#pragma unroll
              for (int x = 0; x < 4; ++x) {
                const int n = 16 * ct + kq + 4 * x;
                const int cb = (n * (2 * na - 1 - n)) >> 1;                      // packed column start minus the column index
                famt_store(rU, ct < rt ? mok : (mok && m >= n), cb + m, accU[rt * (rt + 1) / 2 + ct - LO * (LO + 1) / 2][x]);
              }
          }
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int n = kq + 4 * x;
            famt_store(rP, mok && n < nn, (nn + m) + n * nf, accQ[rt - LO][x]);
          }
        }
        if constexpr (WITHG) {
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int jn = kq + 4 * x;
            famt_store(rP, l15 < nn && jn <= l15, l15 + jn * nf, accG[x]);
          }
        }
      };
      if constexpr (!UPD) {                           // Q and G_NN only: five accumulator tiles at most, one pass
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, NAT>{}, std::true_type{});
      } else if constexpr (NAT == 4 && FAMT_NW > 8) {       // three passes: at most six accumulator tiles live (168 registers, three waves per SIMD)
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, std::true_type{});
        pass(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, std::false_type{});
        pass(std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, std::false_type{});
      } else if constexpr (NAT == 3 && FAMT_NW > 8) {
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, std::true_type{});
        pass(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, std::false_type{});
      } else if constexpr (NAT == 4) {
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, std::false_type{});
        pass(std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, std::true_type{});
      } else {
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, NAT>{}, std::true_type{});
      }
    }
    q0 += ep;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused extend-add (round 4): the update matrix of a family parent, Upd = sum_t' s a(x') a(y')^T, is a function of the
// family's tables and of the constraint's entries alone, and its only reader is the extend-add of the parent front above,
// which holds that front in LDS.  k_fam_terms<NAT, false> therefore leaves it out, and the extend-add's workgroup for
// (front, right-hand side) forms it itself: one wave per family child, the term list of (family, constraint) -- static,
// built by kkt_set_constraints: vector ids and scale of every entry -- one term per lane, the operands a(x)[row] gathered
// straight from the family's record in global memory (L2 / MALL resident: 88 MB on synth50k, read by the 100 right-hand
// sides of the front), ten accumulator tiles on MFMA, ds_add_f64 at (rel[i], rel[j]) of the front.  1.49 GB of packed updates
// written by one kernel and read back by the other per Schur complement never exist.
// ---------------------------------------------------------------------------------------------------------------------
// PFG: steps whose operand gathers are requested together, ahead of their products (1: a step's gather, then its products).
// A step is one gather round trip (the tables live in L2 / MALL) followed by ten MFMAs: a latency chain per family.  With
// sixteen waves per workgroup (128 registers each, 80 of them accumulators) nothing more fits; eight waves (256) hold the
// operands of four steps, so a family of the usual 13 entries (seven steps) costs two round trips instead of seven.
template <int NAT, int PFG = 1>
__device__ inline void lf_add_family(double* T, int nf, const MfmaArgs& a, int z, int64_t pt, const int32_t* rel, int lane) {
  constexpr int NA = 16 * NAT, NTU = NAT * (NAT + 1) / 2;
  const int l15 = lane & 15, kq = lane >> 4;
  const int slot = z & 0x7ffff, nnp = (z >> 19) & 31, nap = (z >> 24) & 127;
  const FamtL L = famt_layout<NAT>(8 * a.fz_cnn);
  const double* const tab = a.fz_tab + (int64_t)slot * a.fz_recl + FAMT_HDR;
  // (a list of more than 64 terms -- one per lane -- is taken in chunks of 64, the accumulators persist)
  const int pbase = (int)(uint32_t)(pt & 0xffffffffll), Ttot = min((int)(pt >> 32), FAMT_TMAX);
  int relv[NAT];
#pragma unroll
  for (int t = 0; t < NAT; ++t) relv[t] = rel[min(16 * t + l15, max(nap - 1, 0))];
  d4 acc[NTU];
#pragma unroll
  for (int x = 0; x < NTU; ++x) acc[x] = d4{0.0, 0.0, 0.0, 0.0};
  auto chunk = [&](const int p0, const int Tn) {
  int pk = 0;
  double sv = 0.0;
  if (lane < Tn) { pk = a.fz_pk[p0 + lane]; sv = a.fz_s[p0 + lane]; }
  const int ks = (2 * Tn + 3) >> 2;
  // operands of step s: the a images of the two vectors of ordered pair tp = kq + 4 s (entry tp >> 1 taken as (x, y) or (y, x)).
  // The two ordered pairs of an entry sit sixteen lanes apart (kq even / odd) and want the same two images with the roles
  // swapped: every lane gathers the image of ITS y only and takes the image of its x from the partner lane (lane ^ 16: a
  // cross-lane move instead of a second 128-byte gather: half the gather traffic, the same time -- the step is a latency chain).
  auto fetch = [&](int s, double (&aA)[NAT], double& se, int& ux, int& uy) {
    const int tp = kq + 4 * s, e = tp >> 1;
    const int pke = __shfl(pk, e, 64);
    se = __shfl(sv, e, 64);                                  // (lanes beyond the list hold a zero scale)
    const int v0 = pke & 0xffff, v1 = (pke >> 16) & 0xffff;
    const int vx = (tp & 1) ? v1 : v0, vy = (tp & 1) ? v0 : v1;
    // a image of a vector id: offset in the record and the index of its unit part (255: none)
    ux = (vx < FAMT_CHILD && vx >= nnp) ? vx - nnp : 255;
    const int ay = vy >= FAMT_CHILD ? L.oCA + NA * (vy - FAMT_CHILD) : (vy < nnp ? L.onK + NA * vy : L.oZero);
    uy = (vy < FAMT_CHILD && vy >= nnp) ? vy - nnp : 255;
#pragma unroll
    for (int t = 0; t < NAT; ++t) aA[t] = tab[ay + 16 * t + l15];
  };
  auto mma = [&](double (&aA)[NAT], double se, int ux, int uy) {
    double bA[NAT];
#pragma unroll
    for (int t = 0; t < NAT; ++t) bA[t] = __shfl_xor(aA[t], 16, 64);
#pragma unroll
    for (int t = 0; t < NAT; ++t) {
      const int row = 16 * t + l15;
      aA[t] = se * (aA[t] + (row == uy ? 1.0 : 0.0));
      bA[t] = bA[t] + (row == ux ? 1.0 : 0.0);
    }
#pragma unroll
    for (int rt = 0; rt < NAT; ++rt)
#pragma unroll
      for (int ct = 0; ct <= rt; ++ct)
        acc[rt * (rt + 1) / 2 + ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(aA[ct], bA[rt], acc[rt * (rt + 1) / 2 + ct], 0, 0, 0);
  };
  // (requesting step s + 1 before the products of step s was measured again with the halved gathers: 34 spilled registers at
  // sixteen waves, 0.60 ms against 0.53)
  if constexpr (PFG == 1) {
    for (int s = 0; s < ks; ++s) {
      double aA[NAT], se;
      int ux, uy;
      fetch(s, aA, se, ux, uy);
      mma(aA, se, ux, uy);
    }
  } else {
    for (int s0 = 0; s0 < ks; s0 += PFG) {
      double aA[PFG][NAT], se[PFG];
      int ux[PFG], uy[PFG];
      // (unconditional requests at a clamped step -- a load under a branch turns every wait into a wait for all loads; a step
      // beyond the list gets a zero scale)
#pragma unroll
      for (int g = 0; g < PFG; ++g) {
        fetch(min(s0 + g, ks - 1), aA[g], se[g], ux[g], uy[g]);
        if (s0 + g >= ks) se[g] = 0.0;
      }
#pragma unroll
      for (int g = 0; g < PFG; ++g) mma(aA[g], se[g], ux[g], uy[g]);
    }
  }
  };
  // (the usual list is one chunk: straight-line, as before the chunks existed -- the kernel runs at its register limit)
  if (Ttot <= 64) chunk(pbase, Ttot);
  else for (int c0 = 0; c0 < Ttot; c0 += 64) chunk(pbase + c0, min(64, Ttot - c0));
  // element (m, n) of the update -> front position (rel[m], rel[n]); packed column start minus the column index as in lf_alds_task
#pragma unroll
  for (int rt = 0; rt < NAT; ++rt) {
    const int m = 16 * rt + l15;
#pragma unroll
    for (int ct = 0; ct <= rt; ++ct)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int n = 16 * ct + kq + 4 * x;
        const int cj = __shfl(relv[ct], kq + 4 * x, 64);
        const bool ok = m < nap && n < nap && (ct < rt || m >= n);
        if (ok) unsafeAtomicAdd(&T[cj * nf - ((cj * (cj - 1)) >> 1) - cj + relv[rt]], acc[rt * (rt + 1) / 2 + ct][x]);
      }
  }
}
template <int NAT, int PFG = 1>
struct AldsFam {
  const MfmaArgs* a;
  const int64_t* sCu; const int64_t* sCr;
  __device__ void operator()(double* T, int nf, int r, const int* sFz, int nmine, int wave, int nw, int lane) const {
    (void)r;
    for (int qi = wave; qi < nmine; qi += nw) {
      const int z = sFz[qi];
      if (z >= 0) lf_add_family<NAT, PFG>(T, nf, *a, z, sCu[qi], a->t.relidx + sCr[qi], lane);
    }
  }
};
// the task-drawing extend-add (k_lf_assemble_lds_dyn) with the hook.  The draw is XCD-aware: workgroups are dealt to the
// eight XCDs round-robin by block index, and queue q = blockIdx.x & 7 holds the (front, right-hand side) tasks of the fronts
// q, q + 8, ... -- the hundred right-hand sides of one front gather from the SAME families' tables (11 MB per front on
// synth50k, 88 MB in all), so a front worked on by one XCD keeps its tables in that XCD's L2 instead of every L2 seeing all
// of them; a workgroup whose queue is empty takes from the others (counters[0 .. 7], zeroed by the launch).
// The last round: 800 pairs on 256 workgroups are 3.125 rounds, and a pair is one workgroup's task because its front fills
// LDS.  The pairs from `tail_first` on in every queue (the host passes the ones that would make up a last round less than
// half full) are therefore dealt in `nzt` shares of their children each: every share gathers its children into a front of
// its own and adds what it gathered to the panel and update block with global atomics (cleared by k_lf_zero_pairs before
// the launch; share 0 brings the constraint's own entries).  tail_first < 0: every pair whole.
template <int NAT, int NTH>
__global__ void __launch_bounds__(NTH) k_lf_assemble_fz(MfmaArgs a, double* u, int64_t ldu, int sgn, int cnt, int nrhs, int* counters,
                                                        int tail_first, int nzt) {
  extern __shared__ __attribute__((aligned(16))) double T[];
  __shared__ int stask;
  // (the child table of lf_alds_task behind the front: the hook reads the entries the table phase prepared for it)
  const int64_t* const sCu = reinterpret_cast<const int64_t*>(T + lf_alds_doubles(a.nnmax + a.namax));
  const int64_t* const sCr = sCu + a.nchmax;
  const int q0 = (int)(blockIdx.x & 7);
  for (int dq = 0; dq < 8; ++dq) {
    const int q = (q0 + dq) & 7;
    const int nfq = (cnt - q + 7) >> 3;                 // fronts q, q + 8, ... below cnt
    const int total = nfq * nrhs;
    if (total <= 0) continue;
    const int full = (tail_first >= 0 && tail_first < total) ? tail_first : total;
    const int units = full + (total - full) * nzt;
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) stask = atomicAdd(counters + q, 1);
      __syncthreads();
      const int t = stask;
      if (t >= units) break;
      if (t < full) {
        lf_alds_task(a, u, ldu, sgn, q + 8 * (t / nrhs), t % nrhs, 0, 1, T, AldsFam<NAT, (NTH == 512 ? 4 : 1)>{&a, sCu, sCr});
      } else {
        const int s2 = t - full, tt = full + s2 / nzt, share = s2 % nzt;
        const int tc = (a.nchmax + nzt - 1) / nzt;              // (lf_alds_task lays the child table out for its share)
        lf_alds_task(a, u, ldu, sgn, q + 8 * (tt / nrhs), tt % nrhs, share, nzt, T, AldsFam<NAT, (NTH == 512 ? 4 : 1)>{&a, sCu, sCu + tc});
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same sweep by SIBLING GROUPS: family parents under one large front whose separators are the same rows of that front
// (identical relative indices, hence equal na) hand it update matrices that the extend-add only ever uses as their SUM.
// One workgroup takes a group of up to FAMT_GMAX such families and eight right-hand sides (one per wave): the members'
// tables pass through LDS one after the other, every wave adds the update tiles of (member, its right-hand side) to the
// SAME accumulators, the members' panels (Q, G_NN) are stored as before and the summed update is stored ONCE, into the slot
// of the group's first member -- the other members' slots of the exchange buffer are not written, and every extend-add
// above is told so (MfmaArgs::chskip).  synth50k: 112 parents under each of the 8 top fronts, 14 groups of 8 per front: the
// exchange shrinks from 1.49 GB written here and read again by k_lf_assemble_lds to 0.19 GB each way per Schur sweep.
// The entry lists of all members for the round's right-hand sides are staged in one set-up (three dependent global round
// trips per round instead of per member); what a member switch costs is the copy of its tables (L.total doubles).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int FAMT_GMAX = 8;
constexpr int FAMT_GTAB = 2 * 8 * FAMT_GMAX * 9;      // ints of the (pass, member, list) table
__host__ __device__ inline int famt_grp_misc_doubles() { return (12 + FAMT_GTAB) / 2; }

template <int NAT>
__global__ void __launch_bounds__(512) k_fam_terms_grp(MfmaArgs a, double* u, int64_t ldu, const double* famt, int cnn,
                                                       const int32_t* kc_ij, int ecap) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NA = 16 * NAT, NW = 8, NMEM = 9, NTU = NAT * (NAT + 1) / 2;
  const int ncol = 8 * cnn;
  const FamtL L = famt_layout<NAT>(ncol);
  const int64_t recl = FAMT_HDR + L.total;
  const int g0 = a.grp_ptr[blockIdx.x];
  const int M = min(a.grp_ptr[blockIdx.x + 1] - g0, FAMT_GMAX);
  const int32_t* const glist = a.grp_list + g0;              // record indices of the members
  const int tid = threadIdx.x;
  const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gy = (int)gridDim.y;
  const int npass = ((int)a.nrhs - (int)blockIdx.y + gy - 1) / gy;
  double* const dS = smem + L.total + wave * famt_desc_doubles();
  int* const dI = reinterpret_cast<int*>(dS + FAMT_TCAP);
  int* const misc = reinterpret_cast<int*>(smem + L.total + NW * famt_desc_doubles());
  int* const epfit = misc;
  int* const cdim = misc + 2;
  int* const tab = misc + 12;                                // [pass][member][list] -> (count, where)
  double* const lval = reinterpret_cast<double*>(tab + FAMT_GTAB);
  int* const lpk = reinterpret_cast<int*>(lval + ecap);
  const int32_t* const hlead = reinterpret_cast<const int32_t*>(famt + (int64_t)glist[0] * recl);
  const int na = hlead[2];                                   // the same for every member
  const int64_t lupdp = (int64_t)(uint32_t)hlead[8] | ((int64_t)hlead[9] << 32);
  const int stride = M * NMEM;

  for (int q0 = 0; q0 < npass;) {
    // ------------------------------------------------------------------------------------------------- set-up of a round
    const int epmax = min(NW, npass - q0), npairs = epmax * stride;
    __syncthreads();
    if (tid == 0) *epfit = epmax;
    int myp0[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int idx = tid + 512 * h;
      if (idx < npairs) {
        const int qq = idx / stride, rem = idx - qq * stride, mf = rem / NMEM, mem = rem - mf * NMEM;
        const int r = (int)blockIdx.y + (q0 + qq) * gy;
        const int j = a.kc_ids ? a.kc_ids[a.kc_j0 + r] : a.kc_j0 + r;
        const int32_t* const hm = reinterpret_cast<const int32_t*>(famt + (int64_t)glist[mf] * recl);
        const int ck = mem ? hm[16 + 6 * (mem - 1)] : hm[0];
        int cnt = 0;
        if (ck >= 0) {
          const int32_t* kp = a.kc_ptr + (int64_t)ck * a.kc_stride;
          myp0[h] = kp[j];
          cnt = kp[j + 1] - myp0[h];
        }
        tab[2 * idx] = cnt;
      }
    }
    __syncthreads();
    if (wave == 0) {
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
          if (run > ecap) atomicMin(epfit, (b + x) / stride);
        }
    }
    __syncthreads();
    const int ep = max(1, *epfit);                           // (one pass always fits: host-checked)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int idx = tid + 512 * h;
      if (idx < ep * stride) {
        const int c = tab[2 * idx], p0 = myp0[h], off = tab[2 * idx + 1];
        const int qq = idx / stride, rem = idx - qq * stride, mf = rem / NMEM, mem = rem - mf * NMEM;
        if (c > 0) {
          const int32_t* const hm = reinterpret_cast<const int32_t*>(famt + (int64_t)glist[mf] * recl);
          const int mnn = mem ? hm[16 + 6 * (mem - 1) + 1] : 0;
          const int mck = mem ? hm[16 + 6 * (mem - 1)] : hm[0];
          const int32_t* const mrel = a.t.relidx + a.t.cl[mck].rel;
          for (int t = 0; t < c && off + t < ecap; ++t) {
            const int ij = kc_ij[p0 + t];
            const int i = ij & 0xffff, jc = ij >> 16;
            int pk = i | (jc << 8);
            if (mem && i >= mnn) pk |= mrel[i - mnn] << 16;
            lpk[off + t] = pk;
            lval[off + t] = a.kc_val[p0 + t];
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // ------------------------------------------------------------------------------------------------- the members in turn
    const int qq = wave;
    const bool active = qq < ep;
    const int r = (int)blockIdx.y + (q0 + min(qq, ep - 1)) * gy;
    d4 accU[NTU];
#pragma unroll
    for (int x = 0; x < NTU; ++x) accU[x] = d4{0.0, 0.0, 0.0, 0.0};
    for (int mf = 0; mf < M; ++mf) {
      const double* const fcm = famt + (int64_t)glist[mf] * recl;
      const int32_t* const hm = reinterpret_cast<const int32_t*>(fcm);
      const int nn = hm[1], nch = hm[3], nf = nn + na;
      const int64_t pblk = (int64_t)(uint32_t)hm[4] | ((int64_t)hm[5] << 32);
      __syncthreads();                                       // the staged entries are in place / the previous tables are consumed
      famt_copy_tables<512>(smem, fcm + FAMT_HDR, L.total, tid);
      if (tid < 8) cdim[tid] = tid < nch ? (hm[16 + 6 * tid + 1] | (hm[16 + 6 * tid + 3] << 8)) : 1;
      __syncthreads();
      if (!active) continue;
      const int* const trow = tab + 2 * ((qq * M + mf) * NMEM);
      int cntm = lane < NMEM ? trow[2 * lane] : 0;
      int incl = cntm;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
      const int T = min(__builtin_amdgcn_readlane(incl, 15), FAMT_TCAP / 2);
      {
        int mem = 0, base = 0;
#pragma unroll
        for (int mm = 0; mm < 8; ++mm) {
          const int up = __builtin_amdgcn_readlane(incl, mm);
          if (lane >= up) { mem = mm + 1; base = up; }
        }
        if (lane < T) {
          const int where = trow[2 * mem + 1];
          const int pk = lpk[where + (lane - base)];
          const double v = lval[where + (lane - base)];
          const int i = pk & 0xff, jc = (pk >> 8) & 0xff, rl = (pk >> 16) & 0xff;
          int vx, vy;
          double s;
          if (mem == 0) { vx = i; vy = jc; s = i == jc ? 0.5 * v : v; }
          else {
            const int cd = cdim[mem - 1], nnc = cd & 0xff, cb = cd >> 8;
            if (i >= nnc) { vx = FAMT_CHILD + cb + jc; vy = rl; s = -v; }
            else { vx = FAMT_CHILD + cb + i; vy = FAMT_CHILD + cb + jc; s = i == jc ? 0.5 * v : v; }
          }
          auto img = [&](int vid, int& on, int& oa, int& om, int& ui) {
            if (vid >= FAMT_CHILD) { const int g = vid - FAMT_CHILD; on = L.oCN + 16 * g; oa = L.oCA + NA * g; om = L.oCM + NA * g; ui = 255; }
            else if (vid < nn) { on = L.oLi + 16 * vid; oa = L.onK + NA * vid; om = L.onMK + NA * vid; ui = 255; }
            else { on = L.oZero; oa = L.oZero; om = L.oRt + NA * (vid - nn); ui = vid - nn; }
          };
          int nx, ax, mx, ux, ny, ay, my, uy;
          img(vx, nx, ax, mx, ux);
          img(vy, ny, ay, my, uy);
          dS[2 * lane] = s; dS[2 * lane + 1] = s;
          int4 w0 = {nx | (ax << 16), mx | (ux << 16), ny | (ay << 16), uy};
          int4 w1 = {ny | (ay << 16), my | (uy << 16), nx | (ax << 16), ux};
          reinterpret_cast<int4*>(dI)[2 * lane] = w0;
          reinterpret_cast<int4*>(dI)[2 * lane + 1] = w1;
        }
        if (lane < 4 && 2 * T + lane < ((2 * T + 3) & ~3)) {
          dS[2 * T + lane] = 0.0;
          int4 z = {L.oZero | (L.oZero << 16), L.oZero | (255 << 16), L.oZero | (L.oZero << 16), 255};
          reinterpret_cast<int4*>(dI)[2 * T + lane] = z;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int ks = (2 * T + 3) >> 2;
      const __amdgpu_buffer_rsrc_t rP = famt_rsrc(u + (int64_t)r * ldu + pblk, nf * nn);
      auto pass = [&](auto LOc, auto HIc, auto Gc) {
        constexpr int LO = decltype(LOc)::value, HI = decltype(HIc)::value;
        constexpr bool WITHG = decltype(Gc)::value;
        d4 accQ[HI - LO], accG = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int x = 0; x < HI - LO; ++x) accQ[x] = d4{0.0, 0.0, 0.0, 0.0};
        for (int s = 0; s < ks; ++s) {
          const int tp = kq + 4 * s;
          const double sc = dS[tp];
          const int4 w = reinterpret_cast<const int4*>(dI)[tp];
          const int nx = w.x & 0xffff, ax = w.x >> 16, mx = w.y & 0xffff, ux = w.y >> 16;
          const int ny = w.z & 0xffff, ay = w.z >> 16, uy = w.w;
          double bA[HI - LO], bM[HI - LO], aA[HI];
#pragma unroll
          for (int t = 0; t < HI; ++t) {
            const int row = 16 * t + l15;
            aA[t] = sc * (smem[ay + row] + (row == uy ? 1.0 : 0.0));
            if (t >= LO) {
              bA[t - LO] = smem[ax + row] + (row == ux ? 1.0 : 0.0);
              bM[t - LO] = smem[mx + row];
            }
          }
          const double aN = sc * smem[ny + l15];
#pragma unroll
          for (int rt = LO; rt < HI; ++rt) {
#pragma unroll
            for (int ct = 0; ct <= rt; ++ct) {
              const int x = rt * (rt + 1) / 2 + ct;
              accU[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(aA[ct], bA[rt - LO], accU[x], 0, 0, 0);
            }
            accQ[rt - LO] = __builtin_amdgcn_mfma_f64_16x16x4f64(aN, bM[rt - LO], accQ[rt - LO], 0, 0, 0);
          }
          if constexpr (WITHG) {
            const double bN = smem[nx + l15];
            accG = __builtin_amdgcn_mfma_f64_16x16x4f64(aN, bN, accG, 0, 0, 0);
          }
        }
#pragma unroll
        for (int rt = LO; rt < HI; ++rt) {
          const int m = 16 * rt + l15;
          const bool mok = m < na;
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int n = kq + 4 * x;
            famt_store(rP, mok && n < nn, (nn + m) + n * nf, accQ[rt - LO][x]);
          }
        }
        if constexpr (WITHG) {
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int jn = kq + 4 * x;
            famt_store(rP, l15 < nn && jn <= l15, l15 + jn * nf, accG[x]);
          }
        }
      };
      if constexpr (NAT == 4) {
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, std::false_type{});
        pass(std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, std::true_type{});
      } else {
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, NAT>{}, std::true_type{});
      }
    }
    // the summed update of the group -> the first member's slot (packed, column-major lower)
    if (active) {
      const __amdgpu_buffer_rsrc_t rU = famt_rsrc(a.t.updp + (int64_t)r * a.t.updplen + lupdp, (na * (na + 1)) >> 1);
#pragma unroll
      for (int rt = 0; rt < NAT; ++rt) {
        const int m = 16 * rt + l15;
        const bool mok = m < na;
#pragma unroll
        for (int ct = 0; ct <= rt; ++ct)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int n = 16 * ct + kq + 4 * x;
            const int cb = (n * (2 * na - 1 - n)) >> 1;
            famt_store(rU, ct < rt ? mok : (mok && m >= n), cb + m, accU[rt * (rt + 1) / 2 + ct][x]);
          }
      }
    }
    q0 += ep;
  }
}

}  // namespace smcp
