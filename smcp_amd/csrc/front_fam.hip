// Family kernel of the leaves->root Hessian sweep: one workgroup owns a small parent front (nn <= 16, na <= 64)
// TOGETHER with its childless children (nn <= 16, na <= 16 NATC <= 32, at most eight), for the sparse-input sweeps
// of the Schur complement.  The per-level kernels (front_n16.hip) hand every child's update matrix to the parent
// through HBM: na(na+1)/2 doubles written by the child and read back by the parent per right-hand side -- on
// synth50k that is 5.7 of the 9 GB a Gram sweep moves.  Here the update matrices never exist in HBM:
//   * waves 4..11 (the child group) each own one child.  A child's constants K, Li, R^T never change and
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
  int oK, oG, oB0, bw, oC, cw, oInt;
};
// one front buffer: F_NN | F_AN (adjacent) | U
template <int NAT, int NATC>
__host__ __device__ constexpr FamL fam_layout() {
  constexpr int NA = 16 * NAT, LDA = NA + 1, LDN = 17, LDC = 16 * NATC + 1;
  FamL L{};
  int o = 0;
  L.oK = o; o += LDA * 16;                    // K of the parent (operand of the E and update products)
  L.oG = o; o += LDA * 16;                    // G of the parent, transposed through LDS for the scaling product
  L.oB0 = o;
  L.bw = LDN * 16 + LDA * 16 + LDA * NA;
  o += 2 * L.bw;
  L.oC = o;                                   // per child wave: F_NN | F_AN (later G)
  L.cw = LDN * 16 + LDC * 16;
  o += 8 * L.cw;
  L.oInt = o;
  return L;
}
template <int NAT, int NATC>
__host__ inline size_t fam_lds_bytes(int panmax, int /*pkmax*/) {
  return (size_t)(fam_layout<NAT, NATC>().oInt + 2 + (panmax + 3) / 4 + 2) * sizeof(double);
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

// SP: sparse input (the right-hand sides are constraints given by their per-clique entry lists, MfmaArgs::kc_*) or
// dense input panels read from u (a template parameter: the sparse instantiation carries none of the dense path's
// registers)
template <int NAT, int NATC, bool SP>
__global__ void __launch_bounds__(768) k_hess_up_fam(MfmaArgs a, double* u, int64_t ldu) {
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
  const int tid = threadIdx.x;
  const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool isP = wave < 4;                    // parent group: waves 0..3, child group: waves 4..11
  const int gw = isP ? wave : wave - 4;         // wave index inside the group
  const int gtid = isP ? tid : tid - 256;       // thread index inside the group
  const int ymode = a.ymode;
  const int npan = nf * nn, npk = na * (na + 1) / 2;
  const int nch = d.chend - d.chbeg;            // <= 8 (host guarantee)
  const int gy = (int)gridDim.y;
  const int ksn = (nn + 3) >> 2, ksa = (na + 3) >> 2;
  const int npass = ((int)a.nrhs - (int)blockIdx.y + gy - 1) / gy;
  constexpr bool sp = SP;

  for (int e = tid; e < L.oInt + 2; e += 768) smem[e] = 0.0;    // pads must be (and stay) zero; counter = 0
  for (int e = tid; e < npan; e += 768) {
    const int i = e % nf, j = e / nf;
    sPan[e] = (u16)((i >= nn) ? bFan + (i - nn) + j * LDA : (i >= j ? bFnn + i + j * LDN : NONE));
  }
  for (int e = tid; e < npan; e += 768) {
    const int i = e % nf, j = e / nf;
    if (i >= nn) smem[L.oK + (i - nn) + j * LDA] = a.LK[d.blk + e];
  }
  __syncthreads();

#ifdef SMCP_STAMPS   // diagnostic build only (SMCP_STAMPS=1 python -m smcp_amd.build --force; scratch/stamps_fam.py)
  const bool stamp = (isP ? gtid == 192 : gtid == 0) && a.dbg;   // parent wave 3 (the heaviest row), child wave 0
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamp ? clock64() : 0;
#define STAMP(i) do { if (stamp) { unsigned long long tn_ = clock64(); tph[i] += tn_ - tlast; tlast = tn_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
  if (isP) {
    // =====================================================================================================
    // parent group
    // =====================================================================================================
    // Constants as MFMA operands in registers: K (all row tiles), Li, and this wave's 16-row slice of the scaling
    // operand.  Every wave forms ALL tiles of E / X / T itself (20 MFMAs instead of 5): the result register rr of
    // a tile is the operand of k-step rr of the next product, so phases 1 and 2 need no LDS exchange and no barrier.
    double yreg[4 * NAT], bdP[4];
    {
      const double* lk = a.LK + d.blk;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kk = kq + 4 * s;
        bdP[s] = (l15 < nn && kk <= l15) ? lk[l15 + (int64_t)kk * nf] : 0.0;              // Li[l15][kk]
      }
    }
    double kPm[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kk = kq + 4 * s, m = 16 * gw + l15;
      kPm[s] = (m < na && kk < nn) ? a.LK[d.blk + (nn + m) + (int64_t)kk * nf] : 0.0;
    }
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
    // The front's own constraint entries of right-hand side st go into the buffer the child group is assembling
    // (atomics, concurrent with the children's).  Their two dependent global loads (range, then entries) would cost
    // two memory latencies per right-hand side, so: lane l keeps the entry range of pass 64 b + l (refreshed every
    // 64 passes) and the entries of the NEXT pass are fetched into registers (one per thread; longer lists finish
    // with direct loads) once those of the current pass are consumed.
    const int32_t* const kpp = sp ? a.kc_ptr + (int64_t)k * a.kc_stride : nullptr;
    int pp0 = 0, pp1 = 0, q_off = 0;
    double q_val = 0.0;
    bool pre_ok = false;
    int gtarget = 0;
    for (int st = 0; st <= npass; ++st) {
      if (st < npass && !sp) {
        // dense input: the front's own panel (lower of NN + AN rows) is added to the buffer being assembled
        const double* Pin = u + (int64_t)((int)blockIdx.y + st * gy) * ldu + d.blk;
        const int oBn = L.oB0 + (st & 1) * L.bw;
        batched_loop<8>(gtid, npan, 256, [=](int e) { return Pin[e]; },
                        [=](int e, double v) {
                          const int o = sPan[e];
                          if (o != NONE) unsafeAtomicAdd(&smem[oBn + o], v);
                        });
      }
      if (st < npass && sp) {
        const int sl = st & 63;
        if (sl == 0) {
          const int rl = (int)blockIdx.y + (st + lane) * gy;
          pp0 = pp1 = 0;
          if (rl < a.nrhs) {
            const int j = a.kc_ids ? a.kc_ids[a.kc_j0 + rl] : a.kc_j0 + rl;
            pp0 = kpp[j]; pp1 = kpp[j + 1];
          }
          pre_ok = false;
        }
        const int q0 = __builtin_amdgcn_readlane(pp0, sl), q1 = __builtin_amdgcn_readlane(pp1, sl);
        if (!pre_ok && gtid < q1 - q0) { q_off = a.kc_off[q0 + gtid]; q_val = a.kc_val[q0 + gtid]; }
        const int oBn = L.oB0 + (st & 1) * L.bw;
        if (gtid < q1 - q0) {
          const int o = sPan[q_off];
          if (o != NONE) unsafeAtomicAdd(&smem[oBn + o], q_val);
        }
        for (int p = q0 + 256 + gtid; p < q1; p += 256) {
          const int o = sPan[a.kc_off[p]];
          if (o != NONE) unsafeAtomicAdd(&smem[oBn + o], a.kc_val[p]);
        }
        pre_ok = sl != 63 && st + 1 < npass;
        if (pre_ok) {
          const int sn = (sl + 1) & 63;
          const int q0n = __builtin_amdgcn_readlane(pp0, sn), q1n = __builtin_amdgcn_readlane(pp1, sn);
          if (gtid < q1n - q0n) { q_off = a.kc_off[q0n + gtid]; q_val = a.kc_val[q0n + gtid]; }
        }
      }
      if (st > 0) {
        const int r = (int)blockIdx.y + (st - 1) * gy;
        const int oB = L.oB0 + ((st - 1) & 1) * L.bw;        // the front of this right-hand side
        const int oFnn = oB + bFnn, oFan = oB + bFan, oU = oB + bU;
        double* const P = u + (int64_t)r * ldu + d.blk;
        double* const UkP = a.t.updp + (int64_t)r * a.t.updplen + d.updp;
        // Wave gw owns row tile gw of the update matrix, of G and of Q.  It forms the E tiles 0..gw itself (the
        // result register rr of a tile is the operand of k-step rr of the next product: no LDS exchange, no barrier
        // before the update products); wave 0 also forms T and G_NN.
        double ev[NAT][4], evm[4] = {0.0, 0.0, 0.0, 0.0}, xvm[4] = {0.0, 0.0, 0.0, 0.0};
        d4 accT = {0.0, 0.0, 0.0, 0.0};
        {
          double fnn[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int kr = kq + 4 * s;                       // F_NN[kr][l15] from the lower triangle (symmetric)
            fnn[s] = smem[oFnn + (kr >= l15 ? kr + l15 * LDN : l15 + kr * LDN)];
          }
#pragma unroll
          for (int t = 0; t < NAT; ++t) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) ev[t][rr] = 0.0;
            if (t <= gw && 16 * t < na) {
              double fan[4];
#pragma unroll
              for (int s = 0; s < 4; ++s) fan[s] = smem[oFan + (16 * t + l15) + (kq + 4 * s) * LDA];
              d4 acc = {0.0, 0.0, 0.0, 0.0};
              double kt[4];
#pragma unroll
              for (int s = 0; s < 4; ++s) kt[s] = smem[L.oK + (16 * t + l15) + (kq + 4 * s) * LDA];
#pragma unroll
              for (int s = 0; s < 4; ++s)
                if (s < ksn) fmma(acc, kt[s], fnn[s]);
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) ev[t][rr] = fan[rr] - 0.5 * acc[rr];       // E = F_AN - K F_NN / 2
              if (t == gw) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) { evm[rr] = ev[t][rr]; xvm[rr] = fan[rr] - acc[rr]; }   // X = F_AN - K F_NN
              }
            }
          }
          if (gw == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
              if (s < ksn) fmma(accT, bdP[s], fnn[s]);
          }
        }
        STAMP(1);
        // update tiles (gw, tn), tn <= gw: U_out = U_assembled - K E^T - E K^T, packed, straight to HBM; the LDS
        // entries are cleared on the way for the right-hand side after next
        if (gw < NAT && 16 * gw < na) {
          const int m = 16 * gw + l15;
#pragma unroll
          for (int tn = 0; tn < NAT; ++tn) {
            if (tn > gw) continue;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            double kt[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) kt[s] = smem[L.oK + (16 * tn + l15) + (kq + 4 * s) * LDA];
#pragma unroll
            for (int s = 0; s < 4; ++s)
              if (s < ksn) {
                fmma(acc, kPm[s], ev[tn][s]);
                fmma(acc, evm[s], kt[s]);
              }
            double* const up = smem + oU + m + (16 * tn + kq) * LDA;
            double uv[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) uv[rr] = up[rr * 4 * LDA];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int n = 16 * tn + kq + 4 * rr;
              if (m >= n) {
                up[rr * 4 * LDA] = 0.0;
                if (m < na) UkP[n * na - ((n * (n - 1)) >> 1) + (m - n)] = uv[rr] - acc[rr];
              }
            }
          }
          // G = X Li^T: row tile gw
          {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s)
              if (s < ksn) fmma(acc, xvm[s], bdP[s]);
            if (ymode) {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) smem[L.oG + m + (kq + 4 * rr) * LDA] = acc[rr];
            } else {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) {
                const int n = kq + 4 * rr;
                if (m < na && n < nn) P[(nn + m) + (int64_t)n * nf] = acc[rr];
              }
            }
          }
        }
        // G_NN = T Li^T (lower), straight to HBM
        if (gw == 0) {
          d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 4; ++s)
            if (s < ksn) fmma(acc, accT[s], bdP[s]);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int jn = kq + 4 * rr;
            if (l15 < nn && jn <= l15) P[l15 + (int64_t)jn * nf] = acc[rr];
          }
        }
        STAMP(2);
        group_barrier(gcnt, gtarget, lane, a.t.info);      // G complete; every wave is done with F_NN / F_AN
        STAMP(3);
        // Q = Ysc G (row tile gw), straight to HBM
        if (ymode && gw < NAT && 16 * gw < na) {
          double gv[4 * NAT];
#pragma unroll
          for (int s2 = 0; s2 < 4 * NAT; ++s2) gv[s2] = smem[L.oG + (kq + 4 * s2) + l15 * LDA];
          d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s2 = 0; s2 < 4 * NAT; ++s2)      // R^T (ymode 2) is zero left of the diagonal block, R (3) right of it
            if (s2 < ksa && !(ymode == 2 && s2 < 4 * gw) && !(ymode == 3 && s2 >= 4 * (gw + 1)))
              fmma(acc, yreg[s2], gv[s2]);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int m = 16 * gw + l15, n = kq + 4 * rr;
            if (m < na && n < nn) P[(nn + m) + (int64_t)n * nf] = acc[rr];
          }
        }
        STAMP(4);
        // clear F_NN / F_AN of this buffer for the right-hand side after next (U was cleared tile by tile above)
        for (int e = gtid; e < (LDN + LDA) * 16; e += 256) smem[oFnn + e] = 0.0;
        STAMP(5);
      }
      lds_barrier();        // stage boundary (whole workgroup)
      STAMP(0);
    }
#ifdef SMCP_STAMPS
    if (stamp) for (int i = 0; i < 7; ++i) atomicAdd(a.dbg + i, tph[i]);
#endif
  } else {
    // =====================================================================================================
    // child group: wave gw owns child gw
    // =====================================================================================================
    const int32_t* kpc[2];
    CliqueDesc cd[2];
    bool hasc[2];
    int nnc[2], nac[2], nfc[2];
    double kreg[2][NATC][4], bdreg[2][4], ycreg[2][NATC][4 * NATC];
    int rm[2][NATC], rn[2][NATC][4];
#pragma unroll
    for (int c = 0; c < 1; ++c) {
      hasc[c] = gw + 8 * c < nch;
      const int ck = hasc[c] ? a.t.chidx[d.chbeg + gw + 8 * c] : 0;
      cd[c] = a.t.cl[ck];
      kpc[c] = sp ? a.kc_ptr + (int64_t)ck * a.kc_stride : nullptr;
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
    double* const cFnn = smem + L.oC + gw * L.cw;       // F_NN (full symmetric)
    double* const cFan = cFnn + LDN * 16;               // F_AN, later G (transposed for the scaling product)
    // entry lists of the child: ranges in the lanes, entries of the next pass prefetched (as in the parent group)
    int cp0[2] = {0, 0}, cp1[2] = {0, 0};
    int e_off[2] = {0, 0};
    double e_val[2] = {0.0, 0.0};
    bool pre_ok = false;
    for (int st = 0; st <= npass; ++st) {
      if (st < npass) {
        const int r = (int)blockIdx.y + st * gy;
        const int oB = L.oB0 + (st & 1) * L.bw;
        const int sl = st & 63;
        if (sl == 0) {
          const int rl = r + lane * gy;
          cp0[0] = cp1[0] = cp0[1] = cp1[1] = 0;
          if (sp && rl < a.nrhs) {
            const int j = a.kc_ids ? a.kc_ids[a.kc_j0 + rl] : a.kc_j0 + rl;
#pragma unroll
            for (int c = 0; c < 1; ++c)
              if (hasc[c]) { cp0[c] = kpc[c][j]; cp1[c] = kpc[c][j + 1]; }
          }
          pre_ok = false;
        }
        int p0[2], p1[2];
#pragma unroll
        for (int c = 0; c < 1; ++c) { p0[c] = __builtin_amdgcn_readlane(cp0[c], sl); p1[c] = __builtin_amdgcn_readlane(cp1[c], sl); }
        if (!pre_ok && sp) {
#pragma unroll
          for (int c = 0; c < 1; ++c)
            if (lane < p1[c] - p0[c]) { e_off[c] = a.kc_off[p0[c] + lane]; e_val[c] = a.kc_val[p0[c] + lane]; }
        }
        const bool more = sp && sl != 63 && st + 1 < npass;      // the next pass exists and its ranges are in the lanes
        const int sn = (sl + 1) & 63;
        STAMP(1);
        // LDS offset of the parent-front position (ri, rj), ri >= rj
        auto ptgt = [&](int ri, int rj) -> int {
          return oB + (rj >= nn ? bU + (ri - nn) + (rj - nn) * LDA : (ri >= nn ? bFan + (ri - nn) + rj * LDA : bFnn + ri + rj * LDN));
        };
#pragma unroll
        for (int c = 0; c < 1; ++c) {
          if (!hasc[c]) continue;
          double* const Pc = u + (int64_t)r * ldu + cd[c].blk;
          const int nnc_ = nnc[c], nac_ = nac[c], nfc_ = nfc[c];
          const int ksnc = (nnc_ + 3) >> 2, ksac = (nac_ + 3) >> 2;
          if (sp && p0[c] == p1[c]) {
            // the constraint does not touch this child: zero panel, zero update
            for (int e = lane; e < nfc_ * nnc_; e += 64) Pc[e] = 0.0;
          } else {
            if (!sp) {
              // dense input: the child's panel overwrites every valid entry of the scratch (the pads stay zero)
              auto putd = [&](int e, double v) {
                const int i = e % nfc_, j = e / nfc_;
                if (i >= nnc_) cFan[(i - nnc_) + j * LDC] = v;
                else if (i >= j) { cFnn[i + j * LDN] = v; cFnn[j + i * LDN] = v; }
              };
              batched_loop<8>(lane, nfc_ * nnc_, 64, [=](int e) { return Pc[e]; }, putd);
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
            }
            wave_sync();
            // phases 1 and 2 in registers: the result register rr of a 16 x 16 tile (row l15, column kq + 4 rr) is
            // exactly what the next product needs as its operand of k-step rr (row l15, k = kq + 4 rr), so E, X and T
            // go from accumulator to operand without passing through LDS.
            //   E = F_AN - K F_NN / 2, X = F_AN - K F_NN, T = Li F_NN;
            //   update -(K E^T + E K^T) -> parent front (LDS atomics); G = X Li^T; G_NN = T Li^T
            d4 accG[NATC], accN = {0.0, 0.0, 0.0, 0.0};
            {
              d4 accE[NATC], accT = {0.0, 0.0, 0.0, 0.0};
              double fnn[4], fan[NATC][4];
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                fnn[s] = cFnn[(kq + 4 * s) + l15 * LDN];
#pragma unroll
                for (int t = 0; t < NATC; ++t) fan[t][s] = cFan[(16 * t + l15) + (kq + 4 * s) * LDC];
              }
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
              double ev[NATC][4], xv[NATC][4];
#pragma unroll
              for (int t = 0; t < NATC; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  ev[t][rr] = fan[t][rr] - 0.5 * accE[t][rr];
                  xv[t][rr] = fan[t][rr] - accE[t][rr];
                }
#pragma unroll
              for (int tm = 0; tm < NATC; ++tm)
#pragma unroll
                for (int tn = 0; tn <= tm; ++tn) {
                  if (16 * tm >= nac_) continue;
                  d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                  for (int s = 0; s < 4; ++s)
                    if (s < ksnc) {
                      fmma(acc, kreg[c][tm][s], ev[tn][s]);
                      fmma(acc, ev[tm][s], kreg[c][tn][s]);
                    }
                  const int ri = rm[c][tm];
#pragma unroll
                  for (int rr = 0; rr < 4; ++rr) {
                    const int rj = rn[c][tn][rr];
                    if (ri >= 0 && rj >= 0 && 16 * tm + l15 >= 16 * tn + kq + 4 * rr) unsafeAtomicAdd(&smem[ptgt(ri, rj)], -acc[rr]);
                  }
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
                if (s < ksnc) fmma(accN, accT[s], bdreg[c][s]);
              wave_sync();                               // every lane has read its F_AN values before G overwrites them
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
