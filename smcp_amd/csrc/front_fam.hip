// Family kernel of the leaves->root Hessian sweep: one workgroup owns a small parent front (nn <= 16, na <= 64)
// TOGETHER with its childless children (nn <= 16, na <= 16 NATC <= 32, at most one child per wave), for the
// sparse-input sweeps of the Schur complement.  The per-level kernels (front_n16.hip) hand every child's update
// matrix to the parent through HBM: na(na+1)/2 doubles written by the child and read back by the parent per
// right-hand side -- on synth50k that is 5.7 of the 9 GB a Gram sweep moves.  Here wave w computes child w's sweep
// out of registers (its constants K, Li, R^T never change and live in VGPRs as MFMA operands) and a private LDS
// scratch, writes the child's output panel straight from the accumulators and adds the update matrix
// -K E^T - E K^T to the parent's front in LDS (ds_add_f64) -- it never exists in HBM.  The parent then runs
// exactly the phases of k_hess_up_n16<NAT, true>.
// Mathematics per clique as in front_mfma.hip (reference call site solvers.py:483).
#include <hip/hip_runtime.h>

namespace smcp {

struct FamL {   // LDS layout (doubles)
  int oK, oBD, oFnn, oFan, oE, oT, oU, oC, cw, oInt;
};
template <int NAT, int NATC>
__host__ __device__ constexpr FamL fam_layout() {
  constexpr int NA = 16 * NAT, LDA = NA + 1, LDN = 17, LDC = 16 * NATC + 1;
  FamL L{};
  int o = 0;
  L.oK = o; o += LDA * 16;
  L.oBD = o; o += LDN * 16;
  L.oFnn = o; o += LDN * 16;
  L.oFan = o; o += LDA * 16;
  L.oE = o; o += LDA * 16;
  L.oT = o; o += LDN * 16;
  L.oU = o; o += LDA * NA;
  L.oC = o;                                   // per-wave child scratch: F_NN (later T) | F_AN (later X, G) | E
  L.cw = LDN * 16 + 2 * LDC * 16;
  o += 8 * L.cw;
  L.oInt = o;
  return L;
}
template <int NAT, int NATC>
__host__ inline size_t fam_lds_bytes(int panmax, int pkmax) {
  return (size_t)(fam_layout<NAT, NATC>().oInt + (panmax + pkmax + 3) / 4 + 2) * sizeof(double);
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

template <int NAT, int NATC>
__global__ void __launch_bounds__(512) k_hess_up_fam(MfmaArgs a, double* u, int64_t ldu) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NA = 16 * NAT, LDA = NA + 1, LDN = 17, LDC = 16 * NATC + 1;
  constexpr FamL L = fam_layout<NAT, NATC>();
  constexpr int NU = NAT * (NAT + 1) / 2;
  typedef unsigned short u16;
  constexpr u16 NONE = 0xffff;
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  u16* const sPan = reinterpret_cast<u16*>(smem + L.oInt);
  u16* const sOut = sPan + a.panmax;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int ymode = a.ymode;
  const int npan = nf * nn, npk = na * (na + 1) / 2;
  const int nch = d.chend - d.chbeg;            // <= nw (host guarantee)
  const int gy = (int)gridDim.y;
  const int ksn = (nn + 3) >> 2, ksa = (na + 3) >> 2;

  for (int e = tid; e < L.oInt; e += nthr) smem[e] = 0.0;      // pads must be (and stay) zero
  for (int e = tid; e < npan; e += nthr) {       // panel entry -> LDS offset (F_NN lower / F_AN); NONE: unused
    const int i = e % nf, j = e / nf;
    sPan[e] = (u16)((i >= nn) ? L.oFan + (i - nn) + j * LDA : (i >= j ? L.oFnn + i + j * LDN : NONE));
  }
  for (int e = tid; e < npk; e += nthr) {        // packed own update entry -> LDS offset
    int i, j;
    pk_unpack(e, na, i, j);
    sOut[e] = (u16)(L.oU + i + j * LDA);
  }
  __syncthreads();
  {
    const double* src = a.LK + d.blk;
    double* const sK = smem + L.oK;
    double* const sBD = smem + L.oBD;
    batched_loop<8>(tid, npan, nthr, [=](int e) { return src[e]; },
                    [=](int e, double v) {
                      const int i = e % nf, j = e / nf;
                      if (i < nn) { if (i >= j) sBD[j + i * LDN] = v; }      // BD = Li^T
                      else sK[(i - nn) + j * LDA] = v;
                    });
  }
  // parent: this wave's 16-row slice of the scaling operand in registers (as k_hess_up_n16)
  double yreg[4 * NAT];
#pragma unroll
  for (int s2 = 0; s2 < 4 * NAT; ++s2) yreg[s2] = 0.0;
  if (ymode && wave < NAT) {
    const double* ys = a.ysc + d.upd;
    const int m = 16 * wave + l15;
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
  // ---- child of this wave: constants as MFMA operands in registers
  const bool hasc = wave < nch;
  const int ck = hasc ? a.t.chidx[d.chbeg + wave] : 0;
  const CliqueDesc cd = a.t.cl[ck];
  const int nnc = hasc ? cd.nn : 0, nac = hasc ? cd.na : 0, nfc = nnc + nac;
  const int ksnc = (nnc + 3) >> 2, ksac = (nac + 3) >> 2;
  double kreg[NATC][4], bdreg[4], ycreg[NATC][4 * NATC];
  int rm[NATC], rn[NATC][4];
#pragma unroll
  for (int t = 0; t < NATC; ++t) {
#pragma unroll
    for (int s = 0; s < 4; ++s) kreg[t][s] = 0.0;
#pragma unroll
    for (int s2 = 0; s2 < 4 * NATC; ++s2) ycreg[t][s2] = 0.0;
    rm[t] = -1;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) rn[t][rr] = -1;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) bdreg[s] = 0.0;
  if (hasc) {
    const double* lk = a.LK + cd.blk;
    const double* ys = a.ysc + cd.upd;
    const int32_t* rel = a.t.relidx + cd.rel;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kk = kq + 4 * s;
      if (l15 < nnc && kk <= l15) bdreg[s] = lk[l15 + (int64_t)kk * nfc];        // Li[l15][kk]
#pragma unroll
      for (int t = 0; t < NATC; ++t) {
        const int m = 16 * t + l15;
        if (m < nac && kk < nnc) kreg[t][s] = lk[(nnc + m) + (int64_t)kk * nfc];  // K[m][kk]
      }
    }
#pragma unroll
    for (int t = 0; t < NATC; ++t) {
      const int m = 16 * t + l15;
      if (m < nac) rm[t] = rel[m];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int n = 16 * t + kq + 4 * rr;
        if (n < nac) rn[t][rr] = rel[n];
      }
      if (ymode)
#pragma unroll
        for (int s2 = 0; s2 < 4 * NATC; ++s2) {
          const int kk = kq + 4 * s2;
          double v = 0.0;
          if (m < nac && kk < nac) {
            if (ymode == 1) v = m >= kk ? ys[m + (int64_t)kk * nac] : ys[kk + (int64_t)m * nac];
            else if (ymode == 2) v = kk >= m ? ys[kk + (int64_t)m * nac] : 0.0;
            else v = m >= kk ? ys[m + (int64_t)kk * nac] : 0.0;
          }
          ycreg[t][s2] = v;
        }
    }
  }
  double* const cFnn = smem + L.oC + wave * L.cw;     // F_NN (full symmetric), then T
  double* const cFan = cFnn + LDN * 16;               // F_AN, then X, then G
  double* const cE = cFan + LDC * 16;
  const int32_t* const kpc = a.kc_ptr + (int64_t)ck * a.kc_stride;
  const int32_t* const kpp = a.kc_ptr + (int64_t)k * a.kc_stride;
  // LDS offset of the parent-front position (ri, rj), ri >= rj
  auto ptgt = [&](int ri, int rj) -> int {
    return rj >= nn ? L.oU + (ri - nn) + (rj - nn) * LDA : (ri >= nn ? L.oFan + (ri - nn) + rj * LDA : L.oFnn + ri + rj * LDN);
  };
  // ---- per-lane invariant operand positions of the parent phases (as k_hess_up_n16)
  const double* const aRowA = smem + l15 + kq * LDA;
  const double* const bColA = smem + kq + l15 * LDA;
  const double* const bColN = smem + kq + l15 * LDN;
  double* const cA = smem + l15 + kq * LDA;
  double* const cN = smem + l15 + kq * LDN;
  __syncthreads();

  // Entry lists of the sparse input.  Their two dependent global loads (range, then entries) would cost two memory
  // latencies per pass with nothing to hide them behind (one workgroup per CU), so: lane l of every wave keeps the
  // entry ranges of pass 64 b + l (refreshed every 64 passes), and the entries of the NEXT pass are fetched into
  // registers (one per lane for the child, one per thread for the parent; longer lists finish with direct loads)
  // as soon as those of the current pass have been consumed.
#ifdef SMCP_STAMPS   // diagnostic build only (SMCP_STAMPS=1 python -m smcp_amd.build --force; scratch/stamps_fam.py)
  const bool stamp = tid == 0 && a.dbg;
  unsigned long long tph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamp ? clock64() : 0;
#define STAMP(i) do { if (stamp) { unsigned long long tn_ = clock64(); tph[i] += tn_ - tlast; tlast = tn_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
  int cp0 = 0, cp1 = 0, pp0 = 0, pp1 = 0;
  int e_off = 0, q_off = 0;
  double e_val = 0.0, q_val = 0.0;
  bool pre_ok = false;
  int it = 0;
  for (int r0 = blockIdx.y; r0 < a.nrhs; r0 += gy, ++it) {
    const int sl = it & 63;
    if (sl == 0) {
      const int r = r0 + lane * gy;
      cp0 = cp1 = pp0 = pp1 = 0;
      if (r < a.nrhs) {
        const int j = a.kc_ids ? a.kc_ids[a.kc_j0 + r] : a.kc_j0 + r;
        pp0 = kpp[j]; pp1 = kpp[j + 1];
        if (hasc) { cp0 = kpc[j]; cp1 = kpc[j + 1]; }
      }
      pre_ok = false;
    }
    const int p0 = __builtin_amdgcn_readlane(cp0, sl), p1 = __builtin_amdgcn_readlane(cp1, sl);
    const int q0 = __builtin_amdgcn_readlane(pp0, sl), q1 = __builtin_amdgcn_readlane(pp1, sl);
    if (!pre_ok) {
      if (lane < p1 - p0) { e_off = a.kc_off[p0 + lane]; e_val = a.kc_val[p0 + lane]; }
      if (tid < q1 - q0) { q_off = a.kc_off[q0 + tid]; q_val = a.kc_val[q0 + tid]; }
    }
    const bool more = sl != 63 && r0 + gy < a.nrhs;       // the next pass exists and its ranges are in the lanes
    const int p0n = more ? __builtin_amdgcn_readlane(cp0, (sl + 1) & 63) : 0, p1n = more ? __builtin_amdgcn_readlane(cp1, (sl + 1) & 63) : 0;
    const int q0n = more ? __builtin_amdgcn_readlane(pp0, (sl + 1) & 63) : 0, q1n = more ? __builtin_amdgcn_readlane(pp1, (sl + 1) & 63) : 0;
    pre_ok = more;
    lds_barrier();
    STAMP(0);
    for (int e = tid; e < (LDN + LDA) * 16; e += nthr) smem[L.oFnn + e] = 0.0;     // parent F_NN and F_AN (adjacent)
    for (int e = tid; e < npk; e += nthr) smem[sOut[e]] = 0.0;
    lds_barrier();
    STAMP(1);
    // parent's own constraint entries (concurrent with the children's atomics below)
    if (tid < q1 - q0) {
      const int o = sPan[q_off];
      if (o != NONE) unsafeAtomicAdd(&smem[o], q_val);
    }
    for (int p = q0 + nthr + tid; p < q1; p += nthr) {
      const int o = sPan[a.kc_off[p]];
      if (o != NONE) unsafeAtomicAdd(&smem[o], a.kc_val[p]);
    }
    if (tid < q1n - q0n) { q_off = a.kc_off[q0n + tid]; q_val = a.kc_val[q0n + tid]; }
    STAMP(2);
    // ================= child sweep (wave-private) =================
    if (hasc) {
      double* const Pc = u + (int64_t)r0 * ldu + cd.blk;
      if (p0 == p1) {
        // the constraint does not touch this child: zero panel, zero update
        for (int e = lane; e < nfc * nnc; e += 64) Pc[e] = 0.0;
      } else {
        for (int e = lane; e < LDN * 16 + LDC * 16; e += 64) cFnn[e] = 0.0;        // F_NN and F_AN (adjacent)
        wave_sync();
        {
          auto put = [&](int e, double v) {
            const int i = e % nfc, j = e / nfc;
            if (i >= nnc) cFan[(i - nnc) + j * LDC] = v;
            else if (i >= j) { cFnn[i + j * LDN] = v; cFnn[j + i * LDN] = v; }
          };
          if (lane < p1 - p0) put(e_off, e_val);
          for (int p = p0 + 64 + lane; p < p1; p += 64) put(a.kc_off[p], a.kc_val[p]);
        }
        wave_sync();
        STAMP(3);
        // phase 1: E = F_AN - K F_NN / 2, X = F_AN - K F_NN (in place), T = Li F_NN (over F_NN)
        {
          d4 accE[NATC], accT = {0.0, 0.0, 0.0, 0.0};
          double fnn[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) fnn[s] = s < ksnc ? cFnn[(kq + 4 * s) + l15 * LDN] : 0.0;
#pragma unroll
          for (int t = 0; t < NATC; ++t) {
            accE[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s)
              if (s < ksnc) fmma(accE[t], kreg[t][s], fnn[s]);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s)
            if (s < ksnc) fmma(accT, bdreg[s], fnn[s]);
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
        STAMP(4);
        // phase 2: update -(K E^T + E K^T) -> parent front (LDS atomics); G = X Li^T; G_NN = T Li^T
        {
#pragma unroll
          for (int tm = 0; tm < NATC; ++tm)
#pragma unroll
            for (int tn = 0; tn <= tm; ++tn) {
              if (16 * tm >= nac) continue;
              d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int s = 0; s < 4; ++s)
                if (s < ksnc) {
                  fmma(acc, kreg[tm][s], cE[(16 * tn + l15) + (kq + 4 * s) * LDC]);
                  fmma(acc, cE[(16 * tm + l15) + (kq + 4 * s) * LDC], kreg[tn][s]);
                }
              const int ri = rm[tm];
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) {
                const int rj = rn[tn][rr];
                if (ri >= 0 && rj >= 0 && 16 * tm + l15 >= 16 * tn + kq + 4 * rr) unsafeAtomicAdd(&smem[ptgt(ri, rj)], -acc[rr]);
              }
            }
          d4 accG[NATC], accN = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int t = 0; t < NATC; ++t) {
            accG[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s)
              if (s < ksnc) fmma(accG[t], cFan[(16 * t + l15) + (kq + 4 * s) * LDC], bdreg[s]);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s)
            if (s < ksnc) fmma(accN, cFnn[l15 + (kq + 4 * s) * LDN], bdreg[s]);
          wave_sync();                               // all X operands read before G overwrites them
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int jn = kq + 4 * rr;
            if (l15 < nnc && jn <= l15) Pc[l15 + (int64_t)jn * nfc] = accN[rr];     // G_NN (lower)
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
                if (m < nac && n < nnc) Pc[(nnc + m) + (int64_t)n * nfc] = accG[t][rr];
              }
          }
        }
        STAMP(5);
        // phase 3: Q = M G straight to the output panel
        if (ymode) {
          wave_sync();
#pragma unroll
          for (int t = 0; t < NATC; ++t) {
            if (16 * t >= nac) continue;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s2 = 0; s2 < 4 * NATC; ++s2)     // R^T (ymode 2) is zero left of the diagonal block, R (3) right of it
              if (s2 < ksac && !(ymode == 2 && s2 < 4 * t) && !(ymode == 3 && s2 >= 4 * (t + 1)))
                fmma(acc, ycreg[t][s2], cFan[(kq + 4 * s2) + l15 * LDC]);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int m = 16 * t + l15, n = kq + 4 * rr;
              if (m < nac && n < nnc) Pc[(nnc + m) + (int64_t)n * nfc] = acc[rr];
            }
          }
        }
      }
    }
    if (hasc && lane < p1n - p0n) { e_off = a.kc_off[p0n + lane]; e_val = a.kc_val[p0n + lane]; }
    STAMP(6);
    lds_barrier();
    STAMP(7);
    // ================= parent: the phases of k_hess_up_n16<NAT, true> =================
    for (int e = tid; e < nn * nn; e += nthr) {      // mirror the strict lower triangle of F_NN
      const int i = e % nn, j = e / nn;
      if (i > j) smem[L.oFnn + j + i * LDN] = smem[L.oFnn + i + j * LDN];
    }
    lds_barrier();
    STAMP(8);
    // phase 1: E = F_AN - K F_NN / 2, X = F_AN - K F_NN (in place of F_AN) ; T = Li F_NN
    for (int t = wave; t < NAT + 1; t += nw) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      if (t < NAT) {
        mma_n<4, 4 * LDA, 4>(acc, aRowA + L.oK + 16 * t, bColN + L.oFnn, ksn);
        double* const f = cA + L.oFan + 16 * t;
        double* const e = cA + L.oE + 16 * t;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const double fv = f[rr * 4 * LDA];
          e[rr * 4 * LDA] = fv - 0.5 * acc[rr];
          f[rr * 4 * LDA] = fv - acc[rr];
        }
      } else {
        mma_n<4, 4, 4>(acc, bColN + L.oBD, bColN + L.oFnn, ksn);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) (cN + L.oT)[rr * 4 * LDN] = acc[rr];
      }
    }
    lds_barrier();
    STAMP(9);
    // phase 2: U -= K E^T + E K^T (lower tiles) ; G = X BD (in place) ; G_NN = T BD (into F_NN)
    for (int t = wave; t < NU + NAT + 1; t += nw) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      if (t < NU) {
        int tm = 0, rem = t;
        while (rem > tm) { rem -= tm + 1; ++tm; }
        const int tn = rem;
        mma_n<4, 4 * LDA, 4 * LDA>(acc, aRowA + L.oK + 16 * tm, aRowA + L.oE + 16 * tn, ksn);
        mma_n<4, 4 * LDA, 4 * LDA>(acc, aRowA + L.oE + 16 * tm, aRowA + L.oK + 16 * tn, ksn);
        const int m = 16 * tm + l15;
        double* const up = cA + L.oU + 16 * tm + 16 * tn * LDA;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (m >= 16 * tn + kq + 4 * rr) up[rr * 4 * LDA] -= acc[rr];
      } else if (t < NU + NAT) {
        const int tm = t - NU;
        mma_n<4, 4 * LDA, 4>(acc, aRowA + L.oFan + 16 * tm, bColN + L.oBD, ksn);
        double* const g = cA + L.oFan + 16 * tm;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) g[rr * 4 * LDA] = acc[rr];
      } else {
        mma_n<4, 4 * LDN, 4>(acc, smem + L.oT + l15 + kq * LDN, bColN + L.oBD, ksn);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) (cN + L.oFnn)[rr * 4 * LDN] = acc[rr];
      }
    }
    lds_barrier();
    STAMP(10);
    // phase 3: Q = Ysc G into the (dead) E buffer, or plain G
    for (int t = wave; t < NAT; t += nw) {
      double* const qo = cA + L.oE + 16 * t;
      if (ymode) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const double* const pg = bColA + L.oFan;
#pragma unroll
        for (int s2 = 0; s2 < 4 * NAT; ++s2)
          if (s2 < ksa && !(ymode == 2 && s2 < 4 * t) && !(ymode == 3 && s2 >= 4 * (t + 1)))
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pg[4 * s2], yreg[s2], acc, 0, 0, 0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) qo[rr * 4 * LDA] = acc[rr];
      } else {
        const double* const g = cA + L.oFan + 16 * t;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) qo[rr * 4 * LDA] = g[rr * 4 * LDA];
      }
    }
    lds_barrier();
    STAMP(11);
    // write out: the parent's panel (lower of NN + AN rows from the E buffer) and its packed update
    {
      double* P = u + (int64_t)r0 * ldu + d.blk;
      for (int e = tid; e < npan; e += nthr) {
        const u16 o = sPan[e];
        if (o != NONE) P[e] = smem[o + (o < L.oFan ? 0 : (L.oE - L.oFan))];
      }
      double* UkP = a.t.updp + (int64_t)r0 * a.t.updplen + d.updp;
      for (int e = tid; e < npk; e += nthr) UkP[e] = smem[sOut[e]];
    }
    STAMP(12);
  }
#ifdef SMCP_STAMPS
  if (stamp) for (int i = 0; i < 13; ++i) atomicAdd(a.dbg + i, tph[i]);
#endif
#undef STAMP
}

}  // namespace smcp
