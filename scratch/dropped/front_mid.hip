// Fused leaves->root sweep of a MID-SIZE front with children (round 3): fronts beyond the LDS class whose packed lower
// triangle still fits LDS (nf <= 198, nn <= 64, na <= 128: the eight (64,128) fronts of synth50k with 112 children each).
// One workgroup of sixteen waves owns a (front, right-hand side) pair from the extend-add to the results:
//
//   0  the packed lower triangle of the front F = [F_NN .; F_AN F_AA] in LDS: the front's own input (panel) plus the
//      children's packed update blocks streamed through ds_add_f64 (the body of k_lf_assemble_lds), or -- one
//      right-hand side, where a single workgroup per front would read its 1.9 MB of children alone -- the front
//      that k_lf_assemble has already put into the panel and the update block;
//   1  waves 0..7, one 16-row tile of the separator rows each:  P = K F_NN,  E = F_AN - P / 2 (in place of F_AN in
//      LDS),  X = F_AN - P kept in the accumulators and fed, as it is, to  G = X Li^T  (the accumulator layout of
//      v_mfma_f64_16x16x4 is its own operand layout);  waves 8..11, one 16-row tile of the supernode rows each:
//      T = Li F_NN and G_NN = T Li^T the same way;
//   2  all waves: the 36 lower update tiles  Upd = F_AA - K E^T - E K^T  (E and F_AA from LDS) -> packed exchange
//      buffer, and the 32 tiles of  Q = M G  (M = R^T, Y_AA or R; G through a 64 KB scratch that stays in L2).
//
// Replaces, per level of such fronts, k_lf_clear_upd + k_lf_assemble(_lds) + k_lf_up1 + k_lf_up2 + k_lf_up3 and the
// panel-sized HBM round trips between them (the assembled front was written and read back, E, X, T, G likewise).
// Mathematics as in front_large.hip (reference call site solvers.py:483 / 524).
#include <hip/hip_runtime.h>

namespace smcp {

constexpr int MIDU_MAXNN = 64, MIDU_MAXNA = 128;

__global__ void __launch_bounds__(1024) k_mid_up(MfmaArgs a, double* u, int64_t ldu, int prefilled) {
  extern __shared__ __attribute__((aligned(16))) double T[];
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int r = blockIdx.y;
  const int nn = d.nn, na = d.na, nf = nn + na;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntot = lf_alds_doubles(nf);
  const int nch = d.chend - d.chbeg;
  double* const P = u + (int64_t)r * ldu + d.blk;
  double* const U = a.t.upd + (int64_t)r * a.t.updlen + d.upd;
  double* const UP = a.t.updp + (int64_t)r * a.t.updplen + d.updp;
  double* const Gt = a.t.tmp + (int64_t)r * a.t.tmplen + a.t.tmpptr[k] + (int64_t)nn * nn + (int64_t)na * nn;   // G^T: [j + m nn]
  const double* const Li = a.LK + d.blk;          // Li(i, kk) = Li[i + kk nf], zeros above the diagonal
  const double* const Kp = a.LK + d.blk + nn;     // K(m, kk) = Kp[m + kk nf]
  const double* const Ys = a.ysc ? a.ysc + d.upd : nullptr;
  // packed column start minus the column index: T[cb(j) + i] = front(i, j), i >= j
  auto cb = [nf](int j) { return j * nf - ((j * (j - 1)) >> 1) - j; };

  // ---------------------------------------------------------------------------------------------------------------
  // 0: the front
  // ---------------------------------------------------------------------------------------------------------------
  int64_t* const sCu = reinterpret_cast<int64_t*>(T + lf_alds_doubles(a.nnmax + a.namax));
  int64_t* const sCr = sCu + a.nchmax;
  int* const sCn = reinterpret_cast<int*>(sCr + a.nchmax);
  if (!prefilled)
    for (int q = tid; q < nch; q += 1024) {
      const CliqueDesc c = a.t.cl[a.t.chidx[d.chbeg + q]];
      sCu[q] = c.updp; sCr[q] = c.rel; sCn[q] = c.na;
    }
  for (int e = tid; e < ntot; e += 1024) T[e] = 0.0;
  __syncthreads();
  batched_loop<8>(tid, nf * nn, 1024, [=](int e) { return P[e]; },
                  [=](int e, double pv) { const int i = e % nf, j = e / nf; if (i >= j) T[cb(j) + i] = pv; });
  if (prefilled) {
    if (nch > 0)
      batched_loop<8>(tid, na * na, 1024, [=](int e) { return U[e]; },
                      [=](int e, double uv) { const int i = e % na, j = e / na; if (i >= j) T[cb(nn + j) + nn + i] = uv; });
  } else {
    __syncthreads();                                        // the children add into positions the panel may have set
    const double* ubase = a.t.updp + (int64_t)r * a.t.updplen;
    const int parts = nch < 16 ? max(1, 16 / max(nch, 1)) : 1;
    const int part = parts > 1 ? wave / max(nch, 1) : 0;
    for (int q = parts > 1 ? wave % max(nch, 1) : wave; q < nch && part < parts; q += 16) {
      const int nac = sCn[q];
      const int32_t* rel = a.t.relidx + sCr[q];
      const double* Uc = ubase + sCu[q];
      if (nac <= 64) lf_add_child<16, false>(T, nf, Uc, rel, nac, lane, part, parts);
      else lf_add_child<8, true>(T, nf, Uc, rel, nac, lane, part, parts);
    }
  }
  __syncthreads();
  // F_NN(kk, n), symmetric, from the packed lower triangle
  auto fnn = [&](int kk, int n) -> double {
    const int hi = max(kk, n), lo = min(kk, n);
    return (hi < nn) ? T[cb(lo) + hi] : 0.0;
  };
  const int ktn = (nn + 3) >> 2;                  // k-steps over the supernode columns
  const int ntN = (nn + 15) >> 4, mtA = (na + 15) >> 4;

  // ---------------------------------------------------------------------------------------------------------------
  // 1: E, G (waves 0..7: separator row tile rt) | G_NN (waves 8..11: supernode row tile it)
  // ---------------------------------------------------------------------------------------------------------------
  if (wave < 8 && 16 * wave < na) {
    const int rt = wave, m = 16 * rt + l15;
    double bK[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { const int kk = kq + 4 * s; bK[s] = (m < na && kk < nn) ? Kp[m + (int64_t)kk * nf] : 0.0; }
    d4 X[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      if (ct < ntN) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
          if (s < ktn) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fnn(kq + 4 * s, 16 * ct + l15), bK[s], acc, 0, 0, 0);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int n = 16 * ct + kq + 4 * x;
          double fan = 0.0;
          if (m < na && n < nn) {
            const int pos = cb(n) + nn + m;
            fan = T[pos];
            T[pos] = fan - 0.5 * acc[x];                    // E, in place
          }
          acc[x] = fan - acc[x];                            // X
        }
      }
      X[ct] = acc;
    }
    // G^T[j][m] = sum_n X[m][n] Li[j][n]  (Li lower: n <= j): a = X from the accumulators, b = Li rows
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      if (jt >= ntN) break;
      const int j = 16 * jt + l15;
      double bL[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) { const int kk = kq + 4 * s; bL[s] = (s < 4 * (jt + 1) && j < nn && kk < nn) ? Li[j + (int64_t)kk * nf] : 0.0; }
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        if (ct <= jt)
#pragma unroll
          for (int x = 0; x < 4; ++x) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[ct][x], bL[4 * ct + x], acc, 0, 0, 0);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int mm = 16 * rt + kq + 4 * x;
        if (mm < na && j < nn) Gt[j + (int64_t)mm * nn] = acc[x];
      }
    }
  } else if (wave >= 8 && wave < 12 && 16 * (wave - 8) < nn) {
    const int it = wave - 8, m = 16 * it + l15;
    double bLi[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { const int kk = kq + 4 * s; bLi[s] = (s < 4 * (it + 1) && m < nn && kk < nn) ? Li[m + (int64_t)kk * nf] : 0.0; }
    d4 Tt[4];                                     // T = Li F_NN, row tile it
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      if (ct < ntN)
#pragma unroll
        for (int s = 0; s < 16; ++s)
          if (s < 4 * (it + 1) && s < ktn) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fnn(kq + 4 * s, 16 * ct + l15), bLi[s], acc, 0, 0, 0);
      Tt[ct] = acc;
    }
    // G_NN[m][j] = sum_n T[m][n] Li[j][n], j <= m: a = Li rows of tile jt, b = T from the accumulators
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      if (jt > it) break;
      const int j = 16 * jt + l15;
      double aL[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) { const int kk = kq + 4 * s; aL[s] = (s < 4 * (jt + 1) && j < nn && kk < nn) ? Li[j + (int64_t)kk * nf] : 0.0; }
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        if (ct <= jt)
#pragma unroll
          for (int x = 0; x < 4; ++x) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aL[4 * ct + x], Tt[ct][x], acc, 0, 0, 0);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int jj = 16 * jt + kq + 4 * x;
        if (m < nn && jj <= m) P[m + (int64_t)jj * nf] = acc[x];
      }
    }
  }
  __threadfence_block();
  __syncthreads();

  // ---------------------------------------------------------------------------------------------------------------
  // 2: update tiles and Q tiles, dealt over the sixteen waves
  // ---------------------------------------------------------------------------------------------------------------
  const int nU = mtA * (mtA + 1) / 2, nQ = mtA * ntN;
  const int ymode = a.ymode;
  for (int item = wave; item < nU + nQ; item += 16) {
    if (item < nU) {
      int tm = 0, t = item;
      while (t > tm) { t -= tm + 1; ++tm; }
      const int tn = t;
      const int m = 16 * tm + l15, nrow = 16 * tn + l15;
      double kM[16], kN[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int kk = kq + 4 * s;
        kM[s] = (m < na && kk < nn) ? Kp[m + (int64_t)kk * nf] : 0.0;
        kN[s] = (nrow < na && kk < nn) ? Kp[nrow + (int64_t)kk * nf] : 0.0;
      }
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 16; ++s)
        if (s < ktn) {
          const int kk = kq + 4 * s;
          const double eN = (nrow < na && kk < nn) ? T[cb(kk) + nn + nrow] : 0.0;
          const double eM = (m < na && kk < nn) ? T[cb(kk) + nn + m] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(eN, kM[s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kN[s], eM, acc, 0, 0, 0);
        }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int n = 16 * tn + kq + 4 * x;
        if (m >= n && m < na) UP[pk_idx(m, n, na)] = T[cb(nn + n) + nn + m] - acc[x];
      }
    } else {
      const int q = item - nU, rt = q % mtA, jt = q / mtA;
      const int m = 16 * rt + l15, j = 16 * jt + l15;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      if (ymode) {
        // k range where M(m, k) can be nonzero for this row tile: R^T (2): k >= 16 rt; R (3): k < 16 (rt + 1); Y_AA (1): all
        const int s0 = ymode == 2 ? 4 * rt : 0, s1 = ymode == 3 ? min(4 * (rt + 1), (na + 3) >> 2) : (na + 3) >> 2;
        for (int sb = s0; sb < s1; sb += 8) {
          double gv[8], mv[8];
#pragma unroll
          for (int z = 0; z < 8; ++z) {
            const int kk = kq + 4 * (sb + z);
            const bool on = sb + z < s1 && kk < na;
            gv[z] = (on && j < nn) ? Gt[j + (int64_t)kk * nn] : 0.0;
            mv[z] = (on && m < na) ? yacc(Ys, na, ymode, m, kk) : 0.0;
          }
#pragma unroll
          for (int z = 0; z < 8; ++z) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(gv[z], mv[z], acc, 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int jj = 16 * jt + kq + 4 * x;
          if (m < na && jj < nn) P[nn + m + (int64_t)jj * nf] = acc[x];
        }
      } else {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int jj = 16 * jt + kq + 4 * x;
          if (m < na && jj < nn) P[nn + m + (int64_t)jj * nf] = Gt[jj + (int64_t)m * nn];
        }
      }
    }
  }
}

}  // namespace smcp
