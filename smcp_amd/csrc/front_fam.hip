// Family kernel of the leaves->root Hessian sweep: one workgroup owns a small parent front (nn <= 16, na <= 64)
// TOGETHER with its childless children (nn <= 16, na <= 16 NATC <= 32, at most eight), for the sparse-input sweeps
// of the Schur complement.  The per-level kernels (front_n16.hip) hand every child's update matrix to the parent
// through HBM: na(na+1)/2 doubles written by the child and read back by the parent per right-hand side -- on
// synth50k that is 5.7 of the 9 GB a Gram sweep moves.  Here the update matrices never exist in HBM:
//   * waves 4..7 (the child group) each own up to two children.  A child's constants K, Li, R^T never change and
//     live in VGPRs as MFMA operands; its sweep runs out of a wave-private LDS scratch, the output panel is written
//     straight from the accumulators and the update -K E^T - E K^T is added to the parent's front in LDS (ds_add_f64);
//   * waves 0..3 (the parent group) run the phases of k_hess_up_n16<NAT, true> on the assembled front;
//   * the parent's front is double buffered, so the child group assembles right-hand side i + 1 while the parent
//     group sweeps right-hand side i: one workgroup-wide barrier per right-hand side; the phase boundaries inside
//     the parent group are a counter barrier among its four waves only.
// Mathematics per clique as in front_mfma.hip (reference call site solvers.py:483).
#include <hip/hip_runtime.h>

namespace smcp {

struct FamL {   // LDS layout (doubles)
  int oK, oBD, oE, oT, oB0, bw, oC, cw, oInt;
};
// one front buffer: F_NN | F_AN (adjacent, as in k_hess_up_n16) | U
template <int NAT, int NATC>
__host__ __device__ constexpr FamL fam_layout() {
  constexpr int NA = 16 * NAT, LDA = NA + 1, LDN = 17, LDC = 16 * NATC + 1;
  FamL L{};
  int o = 0;
  L.oK = o; o += LDA * 16;
  L.oBD = o; o += LDN * 16;
  L.oE = o; o += LDA * 16;
  L.oT = o; o += LDN * 16;
  L.oB0 = o;
  L.bw = LDN * 16 + LDA * 16 + LDA * NA;
  o += 2 * L.bw;
  L.oC = o;                                   // per child wave: F_NN (later T) | F_AN (later X, G) | E
  L.cw = LDN * 16 + 2 * LDC * 16;
  o += 4 * L.cw;
  L.oInt = o;
  return L;
}
template <int NAT, int NATC>
__host__ inline size_t fam_lds_bytes(int panmax, int pkmax) {
  return (size_t)(fam_layout<NAT, NATC>().oInt + 2 + (panmax + pkmax + 3) / 4 + 2) * sizeof(double);
}

// acc += Left * Right for one k-step: left = Left[row l15][k = kq + 4 s], right = Right[k = kq + 4 s][col l15];
// the result register rr of a lane is element (row l15, col kq + 4 rr)   (operand map: front_mfma.hip, wg_mma)
__device__ inline void fmma(d4& acc, double left, double right) {
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(right, left, acc, 0, 0, 0);
}
// orders the LDS traffic of ONE wave (its lanes exchange data through the wave-private scratch): LDS operations
// of a wave execute in order, the fence keeps the compiler from moving them across
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Barrier among the four waves of the parent group: a monotone LDS counter (every wave adds one per barrier and
// waits until 4 * barriers-so-far have arrived).  Every wave of the group executes the same barrier sequence; the
// spin is bounded so that a logic error shows up as a failed solve (info flag), never as a hung GPU.
__device__ inline void group_barrier(int* cnt, int& target, int lane, int* info) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  target += 4;
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  int guard = 0;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++guard > (1 << 24)) { if (lane == 0) atomicCAS(info, 0, -7); break; }
  }
  asm volatile("" ::: "memory");
}

// acc += sum over ks (<= 4) k-steps.  All operands are fetched before the first MFMA (the 16-column LDS buffers are
// zero padded, so the loads need no guard): one LDS round trip per tile instead of one per k-step.
__device__ inline void mma_pre(d4& acc, const double* pa, int sa, const double* pb, int sb, int ks) {
  double av[4], bv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) { av[s] = pa[s * sa]; bv[s] = pb[s * sb]; }
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[s], av[s], acc, 0, 0, 0);
}

template <int NAT, int NATC>
__global__ void __launch_bounds__(512) k_hess_up_fam(MfmaArgs a, double* u, int64_t ldu) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NA = 16 * NAT, LDA = NA + 1, LDN = 17, LDC = 16 * NATC + 1;
  constexpr FamL L = fam_layout<NAT, NATC>();
  constexpr int NU = NAT * (NAT + 1) / 2;
  constexpr int bFnn = 0, bFan = LDN * 16, bU = LDN * 16 + LDA * 16;     // offsets inside a front buffer
  typedef unsigned short u16;
  constexpr u16 NONE = 0xffff;
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  int* const gcnt = reinterpret_cast<int*>(smem + L.oInt);          // parent-group barrier counter
  u16* const sPan = reinterpret_cast<u16*>(smem + L.oInt + 2);      // panel entry -> offset inside a front buffer
  u16* const sOut = sPan + a.panmax;                                 // packed own update entry -> offset inside a buffer
  const int tid = threadIdx.x;
  const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool isP = wave < 4;                    // parent group: waves 0..3, child group: waves 4..7
  const int gw = wave & 3, gtid = tid & 255;    // wave / thread index inside the group
  const int ymode = a.ymode;
  const int npan = nf * nn, npk = na * (na + 1) / 2;
  const int nch = d.chend - d.chbeg;            // <= 8 (host guarantee)
  const int gy = (int)gridDim.y;
  const int ksn = (nn + 3) >> 2, ksa = (na + 3) >> 2;
  const int npass = ((int)a.nrhs - (int)blockIdx.y + gy - 1) / gy;

  for (int e = tid; e < L.oInt + 2; e += 512) smem[e] = 0.0;    // pads must be (and stay) zero; counter = 0
  for (int e = tid; e < npan; e += 512) {
    const int i = e % nf, j = e / nf;
    sPan[e] = (u16)((i >= nn) ? bFan + (i - nn) + j * LDA : (i >= j ? bFnn + i + j * LDN : NONE));
  }
  for (int e = tid; e < npk; e += 512) {
    int i, j;
    pk_unpack(e, na, i, j);
    sOut[e] = (u16)(bU + i + j * LDA);
  }
  __syncthreads();
  {
    const double* src = a.LK + d.blk;
    double* const sK = smem + L.oK;
    double* const sBD = smem + L.oBD;
    batched_loop<8>(tid, npan, 512, [=](int e) { return src[e]; },
                    [=](int e, double v) {
                      const int i = e % nf, j = e / nf;
                      if (i < nn) { if (i >= j) sBD[j + i * LDN] = v; }      // BD = Li^T
                      else sK[(i - nn) + j * LDA] = v;
                    });
  }
  __syncthreads();

#ifdef SMCP_STAMPS   // diagnostic build only (SMCP_STAMPS=1 python -m smcp_amd.build --force; scratch/stamps_fam.py)
  const bool stamp = gtid == 0 && a.dbg;
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamp ? clock64() : 0;
#define STAMP(i) do { if (stamp) { unsigned long long tn_ = clock64(); tph[i] += tn_ - tlast; tlast = tn_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
  if (isP) {
    // =====================================================================================================
    // parent group
    // =====================================================================================================
    double yreg[4 * NAT];       // this wave's 16-row slice of the scaling operand (as k_hess_up_n16)
#pragma unroll
    for (int s2 = 0; s2 < 4 * NAT; ++s2) yreg[s2] = 0.0;
    if (ymode && gw < NAT) {
      const double* ys = a.ysc + d.upd;
      const int m = 16 * gw + l15;
#pragma unroll
      for (int s2 = 0; s2 < 4 * NAT; ++s2) {
        const int kk = kq + 4 * s2;
        double v = 0.0;
        if (m < na && kk < na) {
          if (ymode == 1) v = m >= kk ? ys[m + (int64_t)kk * na] : ys[kk + (int64_t)m * na];
          else if (ymode == 2) v = kk >= m ? ys[kk + (int64_t)m * na] : 0.0;     // R^T
          else v = m >= kk ? ys[m + (int64_t)kk * na] : 0.0;                      // R
        }
        yreg[s2] = v;
      }
    }
    const double* const aRowA = smem + l15 + kq * LDA;
    const double* const bColA = smem + kq + l15 * LDA;
    const double* const bColN = smem + kq + l15 * LDN;
    double* const cA = smem + l15 + kq * LDA;
    double* const cN = smem + l15 + kq * LDN;
    // this thread's panel / packed-update entries e = gtid + 256 i: buffer offsets kept in registers, so the
    // write-out and the clearing pass read no index table
    constexpr int NPO = (NA + 16) * 16 / 256, NUO = (NA * (NA + 1) / 2 + 255) / 256;
    u16 po[NPO], uo[NUO];
#pragma unroll
    for (int i = 0; i < NPO; ++i) { const int e = gtid + 256 * i; po[i] = e < npan ? sPan[e] : NONE; }
#pragma unroll
    for (int i = 0; i < NUO; ++i) { const int e = gtid + 256 * i; uo[i] = e < npk ? sOut[e] : NONE; }
    int gtarget = 0;
    for (int st = 0; st <= npass; ++st) {
      if (st > 0) {
        const int r = (int)blockIdx.y + (st - 1) * gy;
        const int oB = L.oB0 + ((st - 1) & 1) * L.bw;        // the front of this right-hand side
        const int oFnn = oB + bFnn, oFan = oB + bFan, oU = oB + bU;
        for (int e = gtid; e < nn * nn; e += 256) {          // mirror the strict lower triangle of F_NN
          const int i = e % nn, j = e / nn;
          if (i > j) smem[oFnn + j + i * LDN] = smem[oFnn + i + j * LDN];
        }
        group_barrier(gcnt, gtarget, lane, a.t.info);
        STAMP(1);
        // phase 1: E = F_AN - K F_NN / 2, X = F_AN - K F_NN (in place of F_AN) ; T = Li F_NN
        for (int t = gw; t < NAT + 1; t += 4) {
          d4 acc = {0.0, 0.0, 0.0, 0.0};
          if (t < NAT) {
            mma_pre(acc, aRowA + L.oK + 16 * t, 4 * LDA, bColN + oFnn, 4, ksn);
            double* const f = cA + oFan + 16 * t;
            double* const e = cA + L.oE + 16 * t;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const double fv = f[rr * 4 * LDA];
              e[rr * 4 * LDA] = fv - 0.5 * acc[rr];
              f[rr * 4 * LDA] = fv - acc[rr];
            }
          } else {
            mma_pre(acc, bColN + L.oBD, 4, bColN + oFnn, 4, ksn);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) (cN + L.oT)[rr * 4 * LDN] = acc[rr];
          }
        }
        group_barrier(gcnt, gtarget, lane, a.t.info);
        STAMP(2);
        // phase 2: U -= K E^T + E K^T (lower tiles) ; G = X BD (in place) ; G_NN = T BD (into F_NN)
        for (int t = gw; t < NU + NAT + 1; t += 4) {
          d4 acc = {0.0, 0.0, 0.0, 0.0};
          if (t < NU) {
            int tm = 0, rem = t;
            while (rem > tm) { rem -= tm + 1; ++tm; }
            const int tn = rem;
            mma_pre(acc, aRowA + L.oK + 16 * tm, 4 * LDA, aRowA + L.oE + 16 * tn, 4 * LDA, ksn);
            mma_pre(acc, aRowA + L.oE + 16 * tm, 4 * LDA, aRowA + L.oK + 16 * tn, 4 * LDA, ksn);
            const int m = 16 * tm + l15;
            double* const up = cA + oU + 16 * tm + 16 * tn * LDA;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
              if (m >= 16 * tn + kq + 4 * rr) up[rr * 4 * LDA] -= acc[rr];
          } else if (t < NU + NAT) {
            const int tm = t - NU;
            mma_pre(acc, aRowA + oFan + 16 * tm, 4 * LDA, bColN + L.oBD, 4, ksn);
            double* const g = cA + oFan + 16 * tm;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) g[rr * 4 * LDA] = acc[rr];
          } else {
            mma_pre(acc, smem + L.oT + l15 + kq * LDN, 4 * LDN, bColN + L.oBD, 4, ksn);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) (cN + oFnn)[rr * 4 * LDN] = acc[rr];
          }
        }
        group_barrier(gcnt, gtarget, lane, a.t.info);
        STAMP(3);
        // phase 3: Q = Ysc G into the (dead) E buffer, or plain G
        for (int t = gw; t < NAT; t += 4) {
          double* const qo = cA + L.oE + 16 * t;
          if (ymode) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            const double* const pg = bColA + oFan;
            double gv[4 * NAT];
#pragma unroll
            for (int s2 = 0; s2 < 4 * NAT; ++s2) gv[s2] = pg[4 * s2];
#pragma unroll
            for (int s2 = 0; s2 < 4 * NAT; ++s2)      // R^T (ymode 2) is zero left of the diagonal block, R (3) right of it
              if (s2 < ksa && !(ymode == 2 && s2 < 4 * t) && !(ymode == 3 && s2 >= 4 * (t + 1)))
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(gv[s2], yreg[s2], acc, 0, 0, 0);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) qo[rr * 4 * LDA] = acc[rr];
          } else {
            const double* const g = cA + oFan + 16 * t;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) qo[rr * 4 * LDA] = g[rr * 4 * LDA];
          }
        }
        group_barrier(gcnt, gtarget, lane, a.t.info);
        STAMP(4);
        // write out: the panel (lower of NN from F_NN, AN rows from the E buffer) and the packed update
        {
          double* P = u + (int64_t)r * ldu + d.blk;
          double pv[NPO], uv[NUO];
#pragma unroll
          for (int i = 0; i < NPO; ++i) pv[i] = po[i] != NONE ? (po[i] < bFan ? smem[oB + po[i]] : smem[L.oE + (po[i] - bFan)]) : 0.0;
#pragma unroll
          for (int i = 0; i < NUO; ++i) uv[i] = uo[i] != NONE ? smem[oB + uo[i]] : 0.0;
#pragma unroll
          for (int i = 0; i < NPO; ++i) if (po[i] != NONE) P[gtid + 256 * i] = pv[i];
          double* UkP = a.t.updp + (int64_t)r * a.t.updplen + d.updp;
#pragma unroll
          for (int i = 0; i < NUO; ++i) if (uo[i] != NONE) UkP[gtid + 256 * i] = uv[i];
        }
        group_barrier(gcnt, gtarget, lane, a.t.info);
        STAMP(5);
        // clear the buffer for the right-hand side after next
        for (int e = gtid; e < (LDN + LDA) * 16; e += 256) smem[oFnn + e] = 0.0;
#pragma unroll
        for (int i = 0; i < NUO; ++i) if (uo[i] != NONE) smem[oB + uo[i]] = 0.0;
        STAMP(6);
      }
      lds_barrier();        // stage boundary (whole workgroup)
      STAMP(0);
    }
#ifdef SMCP_STAMPS
    if (stamp) for (int i = 0; i < 7; ++i) atomicAdd(a.dbg + i, tph[i]);
#endif
  } else {
    // =====================================================================================================
    // child group: wave gw owns children gw and gw + 4
    // =====================================================================================================
    const int32_t* const kpp = a.kc_ptr + (int64_t)k * a.kc_stride;
    const int32_t* kpc[2];
    CliqueDesc cd[2];
    bool hasc[2];
    int nnc[2], nac[2], nfc[2];
    double kreg[2][NATC][4], bdreg[2][4], ycreg[2][NATC][4 * NATC];
    int rm[2][NATC], rn[2][NATC][4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      hasc[c] = gw + 4 * c < nch;
      const int ck = hasc[c] ? a.t.chidx[d.chbeg + gw + 4 * c] : 0;
      cd[c] = a.t.cl[ck];
      kpc[c] = a.kc_ptr + (int64_t)ck * a.kc_stride;
      nnc[c] = hasc[c] ? cd[c].nn : 0;
      nac[c] = hasc[c] ? cd[c].na : 0;
      nfc[c] = nnc[c] + nac[c];
#pragma unroll
      for (int t = 0; t < NATC; ++t) {
#pragma unroll
        for (int s = 0; s < 4; ++s) kreg[c][t][s] = 0.0;
#pragma unroll
        for (int s2 = 0; s2 < 4 * NATC; ++s2) ycreg[c][t][s2] = 0.0;
        rm[c][t] = -1;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) rn[c][t][rr] = -1;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) bdreg[c][s] = 0.0;
      if (hasc[c]) {
        const double* lk = a.LK + cd[c].blk;
        const double* ys = a.ysc + cd[c].upd;
        const int32_t* rel = a.t.relidx + cd[c].rel;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int kk = kq + 4 * s;
          if (l15 < nnc[c] && kk <= l15) bdreg[c][s] = lk[l15 + (int64_t)kk * nfc[c]];        // Li[l15][kk]
#pragma unroll
          for (int t = 0; t < NATC; ++t) {
            const int m = 16 * t + l15;
            if (m < nac[c] && kk < nnc[c]) kreg[c][t][s] = lk[(nnc[c] + m) + (int64_t)kk * nfc[c]];  // K[m][kk]
          }
        }
#pragma unroll
        for (int t = 0; t < NATC; ++t) {
          const int m = 16 * t + l15;
          if (m < nac[c]) rm[c][t] = rel[m];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int n = 16 * t + kq + 4 * rr;
            if (n < nac[c]) rn[c][t][rr] = rel[n];
          }
          if (ymode)
#pragma unroll
            for (int s2 = 0; s2 < 4 * NATC; ++s2) {
              const int kk = kq + 4 * s2;
              double v = 0.0;
              if (m < nac[c] && kk < nac[c]) {
                if (ymode == 1) v = m >= kk ? ys[m + (int64_t)kk * nac[c]] : ys[kk + (int64_t)m * nac[c]];
                else if (ymode == 2) v = kk >= m ? ys[kk + (int64_t)m * nac[c]] : 0.0;
                else v = m >= kk ? ys[m + (int64_t)kk * nac[c]] : 0.0;
              }
              ycreg[c][t][s2] = v;
            }
        }
      }
    }
    double* const cFnn = smem + L.oC + gw * L.cw;       // F_NN (full symmetric), then T
    double* const cFan = cFnn + LDN * 16;               // F_AN, then X, then G
    double* const cE = cFan + LDC * 16;
    // Entry lists of the sparse input.  Their two dependent global loads (range, then entries) would cost two
    // memory latencies per right-hand side, so: lane l of a wave keeps the entry ranges of pass 64 b + l (refreshed
    // every 64 passes) and the entries of the NEXT pass are fetched into registers (one per lane per child, one per
    // thread for the parent; longer lists finish with direct loads) once those of the current pass are consumed.
    int cp0[2] = {0, 0}, cp1[2] = {0, 0}, pp0 = 0, pp1 = 0;
    int e_off[2] = {0, 0}, q_off = 0;
    double e_val[2] = {0.0, 0.0}, q_val = 0.0;
    bool pre_ok = false;
    for (int st = 0; st <= npass; ++st) {
      if (st < npass) {
        const int r = (int)blockIdx.y + st * gy;
        const int oB = L.oB0 + (st & 1) * L.bw;
        const int sl = st & 63;
        if (sl == 0) {
          const int rl = r + lane * gy;
          cp0[0] = cp1[0] = cp0[1] = cp1[1] = pp0 = pp1 = 0;
          if (rl < a.nrhs) {
            const int j = a.kc_ids ? a.kc_ids[a.kc_j0 + rl] : a.kc_j0 + rl;
            pp0 = kpp[j]; pp1 = kpp[j + 1];
#pragma unroll
            for (int c = 0; c < 2; ++c)
              if (hasc[c]) { cp0[c] = kpc[c][j]; cp1[c] = kpc[c][j + 1]; }
          }
          pre_ok = false;
        }
        const int q0 = __builtin_amdgcn_readlane(pp0, sl), q1 = __builtin_amdgcn_readlane(pp1, sl);
        int p0[2], p1[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) { p0[c] = __builtin_amdgcn_readlane(cp0[c], sl); p1[c] = __builtin_amdgcn_readlane(cp1[c], sl); }
        if (!pre_ok) {
          if (gtid < q1 - q0) { q_off = a.kc_off[q0 + gtid]; q_val = a.kc_val[q0 + gtid]; }
#pragma unroll
          for (int c = 0; c < 2; ++c)
            if (lane < p1[c] - p0[c]) { e_off[c] = a.kc_off[p0[c] + lane]; e_val[c] = a.kc_val[p0[c] + lane]; }
        }
        const bool more = sl != 63 && st + 1 < npass;      // the next pass exists and its ranges are in the lanes
        const int sn = (sl + 1) & 63;
        // the parent's own constraint entries (the buffer was cleared by the parent group two stages ago)
        if (gtid < q1 - q0) {
          const int o = sPan[q_off];
          if (o != NONE) unsafeAtomicAdd(&smem[oB + o], q_val);
        }
        for (int p = q0 + 256 + gtid; p < q1; p += 256) {
          const int o = sPan[a.kc_off[p]];
          if (o != NONE) unsafeAtomicAdd(&smem[oB + o], a.kc_val[p]);
        }
        if (more) {
          const int q0n = __builtin_amdgcn_readlane(pp0, sn), q1n = __builtin_amdgcn_readlane(pp1, sn);
          if (gtid < q1n - q0n) { q_off = a.kc_off[q0n + gtid]; q_val = a.kc_val[q0n + gtid]; }
        }
        STAMP(1);
        // LDS offset of the parent-front position (ri, rj), ri >= rj
        auto ptgt = [&](int ri, int rj) -> int {
          return oB + (rj >= nn ? bU + (ri - nn) + (rj - nn) * LDA : (ri >= nn ? bFan + (ri - nn) + rj * LDA : bFnn + ri + rj * LDN));
        };
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (!hasc[c]) continue;
          double* const Pc = u + (int64_t)r * ldu + cd[c].blk;
          const int nnc_ = nnc[c], nac_ = nac[c], nfc_ = nfc[c];
          const int ksnc = (nnc_ + 3) >> 2, ksac = (nac_ + 3) >> 2;
          if (p0[c] == p1[c]) {
            // the constraint does not touch this child: zero panel, zero update
            for (int e = lane; e < nfc_ * nnc_; e += 64) Pc[e] = 0.0;
          } else {
            for (int e = lane; e < LDN * 16 + LDC * 16; e += 64) cFnn[e] = 0.0;        // F_NN and F_AN (adjacent)
            wave_sync();
            {
              auto put = [&](int e, double v) {
                const int i = e % nfc_, j = e / nfc_;
                if (i >= nnc_) cFan[(i - nnc_) + j * LDC] = v;
                else if (i >= j) { cFnn[i + j * LDN] = v; cFnn[j + i * LDN] = v; }
              };
              if (lane < p1[c] - p0[c]) put(e_off[c], e_val[c]);
              for (int p = p0[c] + 64 + lane; p < p1[c]; p += 64) put(a.kc_off[p], a.kc_val[p]);
            }
            wave_sync();
            // phase 1: E = F_AN - K F_NN / 2, X = F_AN - K F_NN (in place), T = Li F_NN (over F_NN)
            {
              d4 accE[NATC], accT = {0.0, 0.0, 0.0, 0.0};
              double fnn[4];
#pragma unroll
              for (int s = 0; s < 4; ++s) fnn[s] = cFnn[(kq + 4 * s) + l15 * LDN];
#pragma unroll
              for (int t = 0; t < NATC; ++t) {
                accE[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s)
                  if (s < ksnc) fmma(accE[t], kreg[c][t][s], fnn[s]);
              }
#pragma unroll
              for (int s = 0; s < 4; ++s)
                if (s < ksnc) fmma(accT, bdreg[c][s], fnn[s]);
              wave_sync();                               // every lane has its F_NN operands before T overwrites them
#pragma unroll
              for (int t = 0; t < NATC; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  const int idx = (16 * t + l15) + (kq + 4 * rr) * LDC;
                  const double fv = cFan[idx];
                  cE[idx] = fv - 0.5 * accE[t][rr];
                  cFan[idx] = fv - accE[t][rr];
                }
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) cFnn[l15 + (kq + 4 * rr) * LDN] = accT[rr];
            }
            wave_sync();
            // phase 2: update -(K E^T + E K^T) -> parent front (LDS atomics); G = X Li^T; G_NN = T Li^T
            {
#pragma unroll
              for (int tm = 0; tm < NATC; ++tm)
#pragma unroll
                for (int tn = 0; tn <= tm; ++tn) {
                  if (16 * tm >= nac_) continue;
                  d4 acc = {0.0, 0.0, 0.0, 0.0};
                  double en[4], em[4];
#pragma unroll
                  for (int s = 0; s < 4; ++s) { en[s] = cE[(16 * tn + l15) + (kq + 4 * s) * LDC]; em[s] = cE[(16 * tm + l15) + (kq + 4 * s) * LDC]; }
#pragma unroll
                  for (int s = 0; s < 4; ++s)
                    if (s < ksnc) {
                      fmma(acc, kreg[c][tm][s], en[s]);
                      fmma(acc, em[s], kreg[c][tn][s]);
                    }
                  const int ri = rm[c][tm];
#pragma unroll
                  for (int rr = 0; rr < 4; ++rr) {
                    const int rj = rn[c][tn][rr];
                    if (ri >= 0 && rj >= 0 && 16 * tm + l15 >= 16 * tn + kq + 4 * rr) unsafeAtomicAdd(&smem[ptgt(ri, rj)], -acc[rr]);
                  }
                }
              d4 accG[NATC], accN = {0.0, 0.0, 0.0, 0.0};
              double xv[NATC][4], tv[4];
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                tv[s] = cFnn[l15 + (kq + 4 * s) * LDN];
#pragma unroll
                for (int t = 0; t < NATC; ++t) xv[t][s] = cFan[(16 * t + l15) + (kq + 4 * s) * LDC];
              }
#pragma unroll
              for (int t = 0; t < NATC; ++t) {
                accG[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s)
                  if (s < ksnc) fmma(accG[t], xv[t][s], bdreg[c][s]);
              }
#pragma unroll
              for (int s = 0; s < 4; ++s)
                if (s < ksnc) fmma(accN, tv[s], bdreg[c][s]);
              wave_sync();                               // all X operands read before G overwrites them
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) {
                const int jn = kq + 4 * rr;
                if (l15 < nnc_ && jn <= l15) Pc[l15 + (int64_t)jn * nfc_] = accN[rr];     // G_NN (lower)
              }
              if (ymode) {
#pragma unroll
                for (int t = 0; t < NATC; ++t)
#pragma unroll
                  for (int rr = 0; rr < 4; ++rr) cFan[(16 * t + l15) + (kq + 4 * rr) * LDC] = accG[t][rr];
              } else {
#pragma unroll
                for (int t = 0; t < NATC; ++t)
#pragma unroll
                  for (int rr = 0; rr < 4; ++rr) {
                    const int m = 16 * t + l15, n = kq + 4 * rr;
                    if (m < nac_ && n < nnc_) Pc[(nnc_ + m) + (int64_t)n * nfc_] = accG[t][rr];
                  }
              }
            }
            // phase 3: Q = M G straight to the output panel
            if (ymode) {
              wave_sync();
              double gv[4 * NATC];
#pragma unroll
              for (int s2 = 0; s2 < 4 * NATC; ++s2) gv[s2] = cFan[(kq + 4 * s2) + l15 * LDC];
#pragma unroll
              for (int t = 0; t < NATC; ++t) {
                if (16 * t >= nac_) continue;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s2 = 0; s2 < 4 * NATC; ++s2)     // R^T (ymode 2) is zero left of the diagonal block, R (3) right of it
                  if (s2 < ksac && !(ymode == 2 && s2 < 4 * t) && !(ymode == 3 && s2 >= 4 * (t + 1)))
                    fmma(acc, ycreg[c][t][s2], gv[s2]);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  const int m = 16 * t + l15, n = kq + 4 * rr;
                  if (m < nac_ && n < nnc_) Pc[(nnc_ + m) + (int64_t)n * nfc_] = acc[rr];
                }
              }
              wave_sync();                               // G is consumed before the next child clears the scratch
            }
          }
          if (more) {
            const int p0n = __builtin_amdgcn_readlane(cp0[c], sn), p1n = __builtin_amdgcn_readlane(cp1[c], sn);
            if (lane < p1n - p0n) { e_off[c] = a.kc_off[p0n + lane]; e_val[c] = a.kc_val[p0n + lane]; }
          }
          STAMP(2 + c);
        }
        pre_ok = more;
      }
      lds_barrier();        // stage boundary (whole workgroup)
      STAMP(0);
    }
#ifdef SMCP_STAMPS
    if (stamp) for (int i = 0; i < 4; ++i) atomicAdd(a.dbg + 16 + i, tph[i]);
#endif
  }
#undef STAMP
}

}  // namespace smcp
