// Shape-specialised leaves->root Hessian sweep for the fronts that carry almost all right-hand-side traffic of
// the block-arrow workloads: supernodes of at most 16 columns (one MFMA column tile) and separators of at
// most 64 rows.  Same mathematics and LDS working set as k_hess_up_pad (front_mfma.hip); what changes is
// the instruction stream.  k_hess_up_pad is VALU-issue bound (about 35 vector-ALU and 23 scalar instructions
// per MFMA: tile decoding, operand pointers and loop control are all computed per lane at run time).  Here
//   * the separator tile count NAT and hence every LDS leading dimension and buffer offset are template
//     constants: k-steps are unrolled and use the immediate-offset field of ds_read_b64 (no pointer bumps);
//   * the wave index is read into a scalar register, so tile decoding and loop control run on the scalar unit;
//   * per-lane operand/result offsets are computed once, outside the right-hand-side loop.
#include <hip/hip_runtime.h>

namespace smcp {

struct N16L {   // LDS layout (doubles) of k_hess_up_n16<NAT, CH>
  int oK, oBD, oFnn, oFan, oE, oT, oU, oInt;
};
template <int NAT, bool CH>
__host__ __device__ constexpr N16L n16_layout() {
  constexpr int NA = 16 * NAT, LDA = NA + 1, LDN = 17;
  N16L L{};
  int o = 0;
  L.oK = o; o += LDA * 16;
  L.oBD = o; o += LDN * 16;
  L.oFnn = o; o += LDN * 16;
  L.oFan = o; o += LDA * 16;
  L.oE = o; o += LDA * 16;
  L.oT = o; o += LDN * 16;
  L.oU = o; o += CH ? LDA * NA : 0;
  L.oInt = o;
  return L;
}
// bytes of dynamic LDS: the double buffers + the index tables (child metadata: 4 ints per child; then 16-bit LDS
// offsets: panel map, own-update map, child targets)
template <int NAT, bool CH>
__host__ inline size_t n16_lds_bytes(int nchmax, int panmax, int pkmax, int plansum) {
  const int shorts = panmax + (CH ? pkmax + plansum : 0);
  return (size_t)(n16_layout<NAT, CH>().oInt + (4 * nchmax + 1) / 2 + (shorts + 3) / 4 + 2) * sizeof(double);
}

// acc += A B over ks (<= MAXS) k-steps, both operands in LDS at compile-time strides
template <int MAXS, int SA, int SB>
__device__ inline void mma_n(d4& acc, const double* pa, const double* pb, int ks) {
#pragma unroll
  for (int s = 0; s < MAXS; ++s)
    if (s < ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[s * SB], pa[s * SA], acc, 0, 0, 0);
}

template <int NAT, bool CH>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) k_hess_up_n16(MfmaArgs a, double* u, int64_t ldu) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NA = 16 * NAT, LDA = NA + 1, LDN = 17;
  constexpr N16L L = n16_layout<NAT, CH>();
  constexpr int NU = NAT * (NAT + 1) / 2;
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  typedef unsigned short u16;
  constexpr u16 NONE = 0xffff;
  int* const sCh = reinterpret_cast<int*>(smem + L.oInt);
  u16* const sPan = reinterpret_cast<u16*>(sCh + 2 * ((4 * a.nchmax + 1) / 2));   // panel entry -> LDS offset (stacked rhs 0)
  u16* const sOut = sPan + a.panmax;
  u16* const sTgt = sOut + (CH ? a.pkmax : 0);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int ymode = a.ymode;
  const int npan = nf * nn, npk = na * (na + 1) / 2;
  const int nch = CH ? d.chend - d.chbeg : 0;
  const int gy = (int)gridDim.y;
  const int rb = CH ? 1 : max(1, min(4, 16 / max(nn, 1)));   // stacked right-hand sides (column = q*nn + n)
  const int ksn = (nn + 3) >> 2, ksa = (na + 3) >> 2, ksc = (rb * nn + 3) >> 2;

  for (int e = tid; e < L.oInt; e += nthr) smem[e] = 0.0;      // pads must be (and stay) zero
  // ---- right-hand-side invariant index tables
  for (int q = tid; q < nch; q += nthr) {
    const CliqueDesc c = a.t.cl[a.t.chidx[d.chbeg + q]];
    sCh[4 * q] = (int)(c.updp & 0xffffffff);
    sCh[4 * q + 1] = (int)(c.updp >> 32);
    sCh[4 * q + 2] = c.na;
  }
  for (int e = tid; e < npan; e += nthr) {       // panel entry -> LDS offset (F_NN lower / F_AN); NONE: unused
    const int i = e % nf, j = e / nf;
    sPan[e] = (u16)((i >= nn) ? L.oFan + (i - nn) + j * LDA : (i >= j ? L.oFnn + i + j * LDN : NONE));
  }
  if (CH)
    for (int e = tid; e < npk; e += nthr) {      // packed own update entry -> LDS offset
      int i, j;
      pk_unpack(e, na, i, j);
      sOut[e] = (u16)(L.oU + i + j * LDA);
    }
  __syncthreads();
  if (CH) {
    if (tid == 0) {
      int off = 0;
      for (int q = 0; q < nch; ++q) { sCh[4 * q + 3] = off; off += sCh[4 * q + 2] * (sCh[4 * q + 2] + 1) / 2; }
    }
    __syncthreads();
    for (int q = wave; q < nch; q += nw) {       // packed child entry -> LDS offset of its target
      const int32_t* rel = a.t.relidx + a.t.cl[a.t.chidx[d.chbeg + q]].rel;
      const int nac = sCh[4 * q + 2], tb = sCh[4 * q + 3];
      for (int e = lane; e < nac * (nac + 1) / 2; e += 64) {
        int i, j;
        pk_unpack(e, nac, i, j);
        const int ri = rel[i], rj = rel[j];
        sTgt[tb + e] = (u16)((rj >= nn) ? L.oU + (ri - nn) + (rj - nn) * LDA
                                        : (ri >= nn ? L.oFan + (ri - nn) + rj * LDA : L.oFnn + ri + rj * LDN));
      }
    }
  }
  {
    const double* src = a.LK + d.blk;
    double* const sK = smem + L.oK;
    double* const sBD = smem + L.oBD;
    batched_loop<8>(tid, npan, nthr, [=](int e) { return src[e]; },
                    [=](int e, double v) {
                      const int i = e % nf, j = e / nf;
                      if (i < nn) {
                        if (i >= j)                               // Li^T on the diagonal blocks of BD
                          for (int q = 0; q < rb; ++q) sBD[(q * nn + j) + (q * nn + i) * LDN] = v;
                      } else sK[(i - nn) + j * LDA] = v;
                    });
  }
  // The scaling operand of phase 3 never changes and wave t only ever multiplies its own 16-row slice of it:
  // keep that slice in registers (A operand: row m = 16 t + l15, k = kq + 4 s) instead of LDS.
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
  // ---- per-lane invariant operand positions
  const double* const aRowA = smem + l15 + kq * LDA;        // + buffer offset + 16*tile: A operand [m = l15][k = kq], ld LDA
  const double* const bColA = smem + kq + l15 * LDA;        // B operand [k = kq][n = l15], ld LDA
  const double* const bColN = smem + kq + l15 * LDN;        // B operand, ld LDN
  double* const cA = smem + l15 + kq * LDA;                 // results: [m = l15][n = kq + 4 rr], ld LDA
  double* const cN = smem + l15 + kq * LDN;

  // ---- software pipelines (registers filled for later passes while the current one computes)
  constexpr int PP = 4, PC = 8;
  bool pipe_ok = CH && nch > 0 && nch <= nw && npan <= PP * nthr && a.plansum > 0;
  if (pipe_ok)
    for (int q = 0; q < nch; ++q) pipe_ok = pipe_ok && (sCh[4 * q + 2] * (sCh[4 * q + 2] + 1) / 2 <= PC * 64);
  const bool sp = a.kc_ptr != nullptr;         // sparse input: the panels are built from the constraint entries
  double pre_p[PP], pre_c[PC];
  auto prefetch = [&](int rr) {
    const double* P = u + (int64_t)rr * ldu + d.blk;
#pragma unroll
    for (int x = 0; x < PP; ++x) { const int e = tid + x * nthr; pre_p[x] = (!sp && e < npan) ? P[e] : 0.0; }
    if (wave < nch) {
      const int nac = sCh[4 * wave + 2], np_ = nac * (nac + 1) / 2;
      const double* Uc = a.t.updp + (int64_t)rr * a.t.updplen + (((int64_t)sCh[4 * wave + 1] << 32) | (uint32_t)sCh[4 * wave]);
#pragma unroll
      for (int x = 0; x < PC; ++x) { const int e = lane + 64 * x; pre_c[x] = e < np_ ? Uc[e] : 0.0; }
    }
  };
  const bool lpipe = !CH && npan <= nthr;
  const int myPan = (tid < npan && sPan[tid] != NONE) ? sPan[tid] : -1;   // this thread's panel entry (lpipe)
  const bool myNN = myPan >= 0 && myPan < L.oFan;                          // ... lies in the F_NN block
  double pre_l[4] = {0.0, 0.0, 0.0, 0.0}, pre_l2[4] = {0.0, 0.0, 0.0, 0.0};
  auto prefetch_leaf = [&](int r0n) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = r0n + q * gy;
      pre_l[q] = pre_l2[q];
      pre_l2[q] = (!sp && q < rb && rr < a.nrhs && tid < npan) ? u[(int64_t)rr * ldu + d.blk + tid] : 0.0;
    }
  };
  if (pipe_ok && (int)blockIdx.y < a.nrhs) prefetch(blockIdx.y);
  if (lpipe) { prefetch_leaf(blockIdx.y); prefetch_leaf(blockIdx.y + gy * rb); }

#ifdef SMCP_STAMPS   // diagnostic build only (SMCP_STAMPS=1 python -m smcp_amd.build --force; scratch/stamps2.py): cycle stamps of thread 0
  const bool stamp = tid == 0 && a.dbg;
  unsigned long long tph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamp ? clock64() : 0;
#define STAMP(i) do { if (stamp) { unsigned long long tn_ = clock64(); tph[i] += tn_ - tlast; tlast = tn_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
  for (int r0 = blockIdx.y; r0 < a.nrhs; r0 += gy * rb) {
    const int rbc = min(rb, (a.nrhs - r0 + gy - 1) / gy);   // right-hand sides r0, r0 + gy, ... of this pass
    lds_barrier();
    STAMP(0);
    // ---- assemble the front(s): panel + children (lower triangles), then mirror F_NN
    if (sp) {
      for (int e = tid; e < (LDN + LDA) * 16; e += nthr) smem[L.oFnn + e] = 0.0;     // F_NN and F_AN are adjacent
    } else if (pipe_ok) {
#pragma unroll
      for (int x = 0; x < PP; ++x) {
        const int e = tid + x * nthr;
        if (e < npan) { const u16 o = sPan[e]; if (o != NONE) smem[o] = pre_p[x]; }
      }
    } else if (lpipe) {
      if (myPan >= 0) {
        const int qs = myNN ? nn * LDN : nn * LDA;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < rbc) smem[myPan + q * qs] = pre_l[q];
      }
      prefetch_leaf(r0 + 2 * gy * rb);
    } else {
      for (int q = 0; q < rbc; ++q) {
        const double* P = u + (int64_t)(r0 + q * gy) * ldu + d.blk;
        const int qan = q * nn * LDA, qnn = q * nn * LDN;
        batched_loop<8>(tid, npan, nthr, [=](int e) { return P[e]; },
                        [=](int e, double v) {
                          const u16 o = sPan[e];
                          if (o != NONE) smem[o + (o < L.oFan ? qnn : qan)] = v;
                        });
      }
    }
    if (CH) for (int e = tid; e < npk; e += nthr) smem[sOut[e]] = 0.0;
    lds_barrier();
    STAMP(1);
    if (sp) {
      // the constraint's entries of this clique (a handful): with children they are added concurrently with the
      // children's atomics; without, both triangles of F_NN are written here and the mirror pass is skipped
      const int32_t* kp = a.kc_ptr + (int64_t)k * a.kc_stride;
      for (int q = 0; q < rbc; ++q) {
        const int r = r0 + q * gy;
        const int j = a.kc_ids ? a.kc_ids[a.kc_j0 + r] : a.kc_j0 + r;
        const int qan = q * nn * LDA, qnn = q * nn * LDN;
        for (int p = kp[j] + tid; p < kp[j + 1]; p += nthr) {
          const int o = sPan[a.kc_off[p]];
          if (o == NONE) continue;
          const double v = a.kc_val[p];
          if (CH) unsafeAtomicAdd(&smem[o], v);
          else if (o >= L.oFan) smem[o + qan] = v;
          else {
            const int rel = o - L.oFnn, i = rel % LDN, jj = rel / LDN;
            smem[o + qnn] = v;
            smem[L.oFnn + jj + i * LDN + qnn] = v;
          }
        }
      }
    }
    if (CH) {
      if (pipe_ok) {
        if (wave < nch) {
          const int nac = sCh[4 * wave + 2], np_ = nac * (nac + 1) / 2;
          const u16* tg = sTgt + sCh[4 * wave + 3];
#pragma unroll
          for (int x = 0; x < PC; ++x) { const int e = lane + 64 * x; if (e < np_) unsafeAtomicAdd(&smem[tg[e]], pre_c[x]); }
        }
        lds_barrier();
        const int rn = r0 + gy;
        if (rn < a.nrhs) prefetch(rn);          // in flight during the three compute phases below
      } else if (nch) {
        const double* ub = a.t.updp + (int64_t)r0 * a.t.updplen;      // children: packed exchange buffer
        for (int q = wave; q < nch; q += nw) {
          const int nac = sCh[4 * q + 2];
          const u16* tg = sTgt + sCh[4 * q + 3];
          const double* Uc = ub + (((int64_t)sCh[4 * q + 1] << 32) | (uint32_t)sCh[4 * q]);
          batched_loop<8>(lane, nac * (nac + 1) / 2, 64, [=](int e) { return Uc[e]; },
                          [=](int e, double vv) { unsafeAtomicAdd(&smem[tg[e]], vv); });
        }
        lds_barrier();
      } else if (sp) lds_barrier();
    }
    STAMP(2);
    if (CH || !sp)
    for (int q = 0; q < rbc; ++q) {              // mirror the strict lower triangle of each F_NN
      double* Fq = smem + L.oFnn + q * nn * LDN;
      for (int e = tid; e < nn * nn; e += nthr) {
        const int i = e % nn, j = e / nn;
        if (i > j) Fq[j + i * LDN] = Fq[i + j * LDN];
      }
    }
    lds_barrier();
    STAMP(3);
    // ---- phase 1: E = F_AN - K F_NN / 2, X = F_AN - K F_NN (in place of F_AN) ; T = Li F_NN   (all stacked columns)
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
      } else {                                   // Li[m][k] = BD[k][m] (first diagonal block of BD)
        mma_n<4, 4, 4>(acc, bColN + L.oBD, bColN + L.oFnn, ksn);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) (cN + L.oT)[rr * 4 * LDN] = acc[rr];
      }
    }
    lds_barrier();
    STAMP(4);
    // ---- phase 2: U_q -= K E_q^T + E_q K^T (lower tiles, per stacked rhs) ; G = X BD (tile in place) ; G_NN = T BD (into F_NN)
    {
      const int nU = NU * rbc;
      for (int t = wave; t < nU + NAT + 1; t += nw) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        if (t < nU) {
          const int q = t / NU;
          int tm = 0, rem = t - q * NU;
          while (rem > tm) { rem -= tm + 1; ++tm; }
          const int tn = rem;
          const int oEq = L.oE + q * nn * LDA;
          mma_n<4, 4 * LDA, 4 * LDA>(acc, aRowA + L.oK + 16 * tm, aRowA + oEq + 16 * tn, ksn);
          mma_n<4, 4 * LDA, 4 * LDA>(acc, aRowA + oEq + 16 * tm, aRowA + L.oK + 16 * tn, ksn);
          const int m = 16 * tm + l15;
          if (CH) {
            double* const up = cA + L.oU + 16 * tm + 16 * tn * LDA;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
              if (m >= 16 * tn + kq + 4 * rr) up[rr * 4 * LDA] -= acc[rr];
          } else {   // no children: the update matrix is exactly -acc, stored packed straight from the accumulators
            double* const UkP = a.t.updp + (int64_t)(r0 + q * gy) * a.t.updplen + d.updp;
            int n = 16 * tn + kq;
            int off = n * na - ((n * (n - 1)) >> 1) + (m - n);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              if (m >= n && m < na) UkP[off] = -acc[rr];
              off += 4 * na - 4 * n - 10;      // packed offset of (m, n + 4) minus that of (m, n)
              n += 4;
            }
          }
        } else if (t < nU + NAT) {
          const int tm = t - nU;
          mma_n<4, 4 * LDA, 4>(acc, aRowA + L.oFan + 16 * tm, bColN + L.oBD, ksc);
          double* const g = cA + L.oFan + 16 * tm;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) g[rr * 4 * LDA] = acc[rr];
        } else {
          mma_n<4, 4 * LDN, 4>(acc, smem + L.oT + l15 + kq * LDN, bColN + L.oBD, ksc);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) (cN + L.oFnn)[rr * 4 * LDN] = acc[rr];
        }
      }
    }
    lds_barrier();
    STAMP(5);
    // ---- phase 3: Q = Ysc G into the (dead) E buffer, or plain G
    for (int t = wave; t < NAT; t += nw) {
      double* const qo = cA + L.oE + 16 * t;
      if (ymode) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const double* const pg = bColA + L.oFan;
#pragma unroll
        for (int s2 = 0; s2 < 4 * NAT; ++s2)
          if (s2 < ksa) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pg[4 * s2], yreg[s2], acc, 0, 0, 0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) qo[rr * 4 * LDA] = acc[rr];
      } else {
        const double* const g = cA + L.oFan + 16 * t;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) qo[rr * 4 * LDA] = g[rr * 4 * LDA];
      }
    }
    lds_barrier();
    STAMP(6);
    // ---- write out: panel(s) (lower of NN + AN) and, with children, the update matrix (lower, packed)
    if (lpipe) {
      if (myPan >= 0) {
        const int base = myNN ? myPan : myPan + (L.oE - L.oFan);     // the AN rows now live in the E buffer
        const int qs = myNN ? nn * LDN : nn * LDA;
        double* P = u + (int64_t)r0 * ldu + d.blk + tid;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < rbc) P[(int64_t)q * gy * ldu] = smem[base + q * qs];
      }
    } else {
      for (int q = 0; q < rbc; ++q) {
        double* P = u + (int64_t)(r0 + q * gy) * ldu + d.blk;
        const int qan = q * nn * LDA + (L.oE - L.oFan), qnn = q * nn * LDN;
        for (int e = tid; e < npan; e += nthr) {
          const u16 o = sPan[e];
          if (o != NONE) P[e] = smem[o + (o < L.oFan ? qnn : qan)];
        }
      }
    }
    STAMP(7);
    if (CH) {
      double* UkP = a.t.updp + (int64_t)r0 * a.t.updplen + d.updp;
      for (int e = tid; e < npk; e += nthr) UkP[e] = smem[sOut[e]];
    }
    STAMP(8);
  }
#ifdef SMCP_STAMPS
  if (stamp) for (int i = 0; i < 9; ++i) atomicAdd(a.dbg + i + 16 * (CH ? 1 : 0), tph[i]);
#endif
#undef STAMP
}

// Root -> leaves Hessian sweep (no scaling operand: the second half of hessian(adj=None), solvers.py:524, 531) of fronts with
// nn <= 16 and na <= 16 NAT <= 64 by ONE WAVE per (clique, right-hand side), four cliques per workgroup, no LDS and no
// barrier: every operand is loaded straight into the lane that needs it as an MFMA operand, and the products are arranged so
// that each loaded register serves as a left operand in one product and as a right operand in another and every
// intermediate is consumed in the accumulator layout it was produced in (the transposed D^T = Li^T Q^T - K^T Z_AA / 2 is
// formed instead of D; register rr of a result tile is the operand of k-step rr of the next product):
//     Tt = Li^T G_NN,  Z1 = Tt Li,  D^T = Li^T Q^T - K^T Z_AA / 2,  Z_AN^T = Li^T Q^T - K^T Z_AA,
//     Z_NN = Z1 - K^T D - D^T K      (= Li^T G_NN Li - K^T D - D^T K, front_mfma.hip k_hess_down_mfma)
// k_hess_down_mfma stages the same products through LDS with five workgroup barriers per clique: 65 us for the 7168 leaves
// of synth50k and 47 us for their 896 parents per right-hand side, against the ~30 us their traffic takes.
// fmma(acc, left, right): left = Left[row l15][k = kq + 4 s], right = Right[k = kq + 4 s][col l15], acc[rr] = (row l15, col kq + 4 rr).
template <int NAT>
__global__ void __launch_bounds__(256, NAT <= 2 ? 3 : 1) k_hess_down_w(MfmaArgs a, double* u, int64_t ldu, int cnt) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const int idx = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (idx >= cnt) return;
  const int k = a.t.lev[idx];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const bool haspar = d.parent >= 0 && na > 0;
  const CliqueDesc par = a.t.cl[haspar ? d.parent : k];
  const int nnp = par.nn, nfp = par.nn + par.na, nap = par.na;
  const bool mirror = d.chend > d.chbeg;               // the children gather Z_AA from the copy in global memory
  const double* LK = a.LK + d.blk;
  double li[4], kk[4 * NAT];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int r = kq + 4 * s;
    const double v = LK[min(r, nn - 1) + (int64_t)min(l15, nn - 1) * nf];
    li[s] = (r < nn && l15 < nn && r >= l15) ? v : 0.0;                       // Li[r][l15], lower triangular
  }
#pragma unroll
  for (int s = 0; s < 4 * NAT; ++s) {
    const int j = kq + 4 * s;
    // (a front without separator has no K rows: the clamped row index falls back into the supernode block -- element 0 of
    // the column -- instead of one row past the panel, which for the last clique is one element past the array)
    const double v = LK[(na > 0 ? nn + min(j, na - 1) : 0) + (int64_t)min(l15, nn - 1) * nf];
    kk[s] = (j < na && l15 < nn) ? v : 0.0;                                    // K[j][l15]
  }
  int ri[NAT], rj[4 * NAT];
  const int32_t* rel = a.t.relidx + d.rel;
#pragma unroll
  for (int t = 0; t < NAT; ++t) ri[t] = haspar ? rel[min(16 * t + l15, na - 1)] : 0;
#pragma unroll
  for (int s = 0; s < 4 * NAT; ++s) rj[s] = haspar ? rel[min(kq + 4 * s, na - 1)] : 0;
  const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
  for (int r = blockIdx.y; r < a.nrhs; r += gridDim.y) {
    double* P = u + (int64_t)r * ldu + d.blk;
    const double* Pp = u + (int64_t)r * ldu + par.blk;
    const double* Up = a.t.upd + (int64_t)r * a.t.updlen + par.upd;
    double g[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = kq + 4 * s;
      const int hi = min(max(l15, c), nn - 1), lo = min(min(l15, c), nn - 1);
      const double v = P[hi + lo * nf];
      g[s] = (l15 < nn && c < nn) ? v : 0.0;                                  // G_NN[c][l15] (symmetric, lower stored)
    }
    // Tt = Li^T G_NN, Z1 = Tt Li
    d4 tt = zero4, z1 = zero4;
#pragma unroll
    for (int s = 0; s < 4; ++s) tt = __builtin_amdgcn_mfma_f64_16x16x4f64(g[s], li[s], tt, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) z1 = __builtin_amdgcn_mfma_f64_16x16x4f64(li[s], tt[s], z1, 0, 0, 0);
    d4 w = zero4;
    double* UkG = a.t.upd + (int64_t)r * a.t.updlen + d.upd;
#pragma unroll
    for (int t = 0; t < NAT; ++t) {
      if (16 * t < na) {
        // the operands of ONE row tile at a time (a separator of 64 rows would otherwise keep 100 loads and their
        // addresses live: 512 registers and spills); the scheduling barrier keeps the next tile's loads behind this tile
        double q[4], z[4 * NAT];
        const int m = 16 * t + l15;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int c = kq + 4 * s;
          const double v = P[nn + min(m, na - 1) + min(c, nn - 1) * nf];
          q[s] = (m < na && c < nn) ? v : 0.0;                                  // Q[m][c]
        }
#pragma unroll
        for (int s = 0; s < 4 * NAT; ++s) {
          const int j = kq + 4 * s;
          int i1 = ri[t], j1 = rj[s];
          asm volatile("" : "+v"(i1), "+v"(j1));       // (keeps the 16 NAT^2 right-hand-side-invariant offsets from being hoisted out of the loop: registers)
          const int hi = max(i1, j1), lo = min(i1, j1);
          // parent's Z: columns of its supernode in its panel, the rest in its separator block (both lower)
          const double* src = lo < nnp ? Pp + (hi + lo * nfp) : Up + ((hi - nnp) + (lo - nnp) * nap);
          const double v = haspar ? *src : 0.0;
          z[s] = (m < na && j < na) ? v : 0.0;                                  // Z_AA[m][j]
        }
        d4 qlt = zero4, zkt = zero4;
#pragma unroll
        for (int s = 0; s < 4; ++s) qlt = __builtin_amdgcn_mfma_f64_16x16x4f64(q[s], li[s], qlt, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4 * NAT; ++s) zkt = __builtin_amdgcn_mfma_f64_16x16x4f64(z[s], kk[s], zkt, 0, 0, 0);
        d4 dt;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          dt[rr] = qlt[rr] - 0.5 * zkt[rr];                                     // D^T[l15][16 t + kq + 4 rr]
          const int mm = 16 * t + kq + 4 * rr;
          if (mm < na && l15 < nn) P[nn + mm + l15 * nf] = qlt[rr] - zkt[rr];              // Z_AN[mm][l15]
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          w = __builtin_amdgcn_mfma_f64_16x16x4f64(dt[s], kk[4 * t + s], w, 0, 0, 0);       // K^T D
          w = __builtin_amdgcn_mfma_f64_16x16x4f64(kk[4 * t + s], dt[s], w, 0, 0, 0);       // D^T K
        }
        if (mirror) {
#pragma unroll
          for (int s = 0; s < 4 * NAT; ++s) {
            const int j = kq + 4 * s;
            if (m < na && j <= m) UkG[m + j * na] = z[s];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int c = kq + 4 * rr;
      if (l15 < nn && c <= l15) P[l15 + (int64_t)c * nf] = z1[rr] - w[rr];               // Z_NN, lower
    }
  }
}

// Dense input panels for the cliques of one launch whose sweep kernel reads its input from u: panel of (clique,
// rhs r) <- A_j restricted to the clique (zeros + the constraint's entries), j as in MfmaArgs::kc_*.
__global__ void k_panel_fill(MfmaArgs a, double* u, int64_t ldu) {
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int r = blockIdx.y;
  double* P = u + (int64_t)r * ldu + d.blk;
  const int nf = d.nn + d.na;
  // gridDim.z workgroups share a panel by column ranges; the sweeps read the NN block through its lower triangle only,
  // so a column is cleared from its diagonal down (half of the panel of a front without separator: config 2)
  const int c0 = (int)(((int64_t)d.nn * blockIdx.z) / gridDim.z), c1 = (int)(((int64_t)d.nn * (blockIdx.z + 1)) / gridDim.z);
  for (int j = c0; j < c1; ++j)
    for (int i = j + threadIdx.x; i < nf; i += blockDim.x) P[i + (int64_t)j * nf] = 0.0;
  __syncthreads();
  const int j = a.kc_ids ? a.kc_ids[a.kc_j0 + r] : a.kc_j0 + r;
  const int32_t* kp = a.kc_ptr + (int64_t)k * a.kc_stride;
  for (int p = kp[j] + threadIdx.x; p < kp[j + 1]; p += blockDim.x) {
    const int off = a.kc_off[p], col = off / nf;
    if (col >= c0 && col < c1) P[off] = a.kc_val[p];
  }
}

}  // namespace smcp
