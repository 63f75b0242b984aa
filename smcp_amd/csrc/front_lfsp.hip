// Sparse-input up-sweep of CHILDLESS fronts beyond the LDS class (config 3: 1999 fronts of (64,128), m = 100 constraints of
// ~60 entries per front): the sweep of a constraint A_j through such a front is a sum of outer products of COLUMNS of the
// front's constants, one or two per entry, so it is formed directly from the entry list instead of from a dense panel:
//
//   F_NN = sum_NN v (e_i e_j^T + e_j e_i^T),  F_AN = sum_AN v e_r e_j^T        (entries of A_j in this front's panel)
//   G_NN = Li F_NN Li^T                    = sum_NN v (l_i l_j^T + l_j l_i^T)                       l_c = Li[:, c]
//   Q    = R^T (F_AN - K F_NN) Li^T        = sum_AN v rt_r l_j^T - sum_NN v (mk_i l_j^T + mk_j l_i^T)   rt_r = R^T e_r, mk_c = (R^T K)[:, c]
//   Upd  = -(K E^T + E K^T), E = F_AN - K F_NN / 2
//        = sum_NN v (k_i k_j^T + k_j k_i^T) - sum_AN v (k_j e_r^T + e_r k_j^T)                          k_c = K[:, c]
//
// (the dense formulation: k_lf_up1/2/3, front_large.hip).  The rank-one sums run on v_mfma_f64_16x16x4 with the four terms of
// a k-step gathered straight from global memory into the operand registers (lane (l15, kq) holds row l15 of the column that
// term kq names: sixteen contiguous doubles per term), the e_r parts of Upd are added to the accumulators of the tiles they
// touch, and the results leave from the accumulators: no dense panel is built, read or rewritten, and a front costs
// ~0.9 k MFMAs per constraint instead of ~4 k.  R^T and R^T K are formed once per factorisation (k_lfsp_prep).
#include <hip/hip_runtime.h>

namespace smcp {

constexpr int LFSP_ECAP = 384;   // entries of one (front, constraint) list that the term tables in LDS hold

// rt <- R^T (full, zeros below its diagonal), mk <- R^T K for the fronts of the launch; R = chol(Y_AA) in fac (lower, ld na)
__global__ void __launch_bounds__(256) k_lfsp_prep(MfmaArgs a, const double* fac, double* rt, double* mk) {
  const int k = a.t.lev[blockIdx.x];
  const CliqueDesc d = a.t.cl[k];
  const int nn = d.nn, na = d.na, nf = nn + na;
  const double* R = fac + d.upd;
  double* Rt = rt + d.upd;
  for (int e = threadIdx.x; e < na * na; e += blockDim.x) {
    const int m = e % na, r = e / na;
    Rt[e] = (r >= m) ? R[r + (int64_t)m * na] : 0.0;
  }
  const double* K = a.LK + d.blk + nn;
  double* MK = mk + d.blk + nn;
  wg_mma(na, nn, na, [=](int m, int kk) { return kk >= m ? R[kk + (int64_t)m * na] : 0.0; },
         [=](int kk, int n) { return K[kk + (int64_t)n * nf]; },
         [=](int m, int n, double acc) { MK[m + (int64_t)n * nf] = acc; });
}

__device__ inline int lfsp_below(unsigned long long m) {       // set bits of m in the lanes below this one
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

// acc[x][y] += sum over the terms t0 .. t1 - 1 (a multiple of four) of left_x(term) right_y(term)^T in rounds of KS k-steps,
// software pipelined over two register sets: the operands of round i + 1 are requested before the MFMAs of round i, so a
// round costs max(memory round trip, its MFMAs) instead of their sum.
// fl(x, t) / fr(y, t): element of this lane (row l15 of tile x / y) of the left / right column of term t; use(x, y): tile wanted
// sc(t): factor of term t, applied to the left operands after ALL loads of the round have been issued (a product right
// behind its load makes the compiler wait for that load before it issues the next one)
template <int KS, int NL, int NR, bool PIPE = true, class FL, class FR, class SC, class US>
__device__ inline void lfsp_rank(d4 (&acc)[NL][NR], int t0, int t1, int kq, FL fl, FR fr, SC sc, US use, bool nold = false) {
  if (t0 >= t1) return;
  double a0[KS][NL], b0[KS][NR], a1[KS][NL], b1[KS][NR];
  auto load = [&](double (&aa)[KS][NL], double (&bb)[KS][NR], int k0) {
    double scl[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bool in = k0 + 4 * s < t1;
      const int t = (in ? k0 + 4 * s : t0) + kq;
      if (nold) {                      // timing ablation: no operand loads at all
#pragma unroll
        for (int x = 0; x < NL; ++x) aa[s][x] = 1.0;
#pragma unroll
        for (int y = 0; y < NR; ++y) bb[s][y] = 1.0;
        continue;
      }
#pragma unroll
      for (int x = 0; x < NL; ++x) aa[s][x] = fl(x, t);
#pragma unroll
      for (int y = 0; y < NR; ++y) bb[s][y] = fr(y, t);          // t is a valid term either way: no branch around the load
      scl[s] = in ? sc(t) : 0.0;
    }
    if (!nold)
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int x = 0; x < NL; ++x) aa[s][x] *= scl[s];
  };
  auto mma = [&](const double (&aa)[KS][NL], const double (&bb)[KS][NR]) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int x = 0; x < NL; ++x)
#pragma unroll
        for (int y = 0; y < NR; ++y)
          if (use(x, y)) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(bb[s][y], aa[s][x], acc[x][y], 0, 0, 0);
  };
  if (!PIPE) {
    for (int k0 = t0; k0 < t1; k0 += 4 * KS) { load(a0, b0, k0); mma(a0, b0); }
    return;
  }
  load(a0, b0, t0);
  for (int k0 = t0; k0 < t1; k0 += 8 * KS) {
    const bool more = k0 + 4 * KS < t1;
    if (more) load(a1, b1, k0 + 4 * KS);
    mma(a0, b0);
    if (!more) break;
    if (k0 + 8 * KS < t1) load(a0, b0, k0 + 8 * KS);
    mma(a1, b1);
  }
}

// na <= 16 NTA, nn <= 16 NTN; grid (right-hand-side groups, fronts), 256 threads
// GRP (the update phase only): blockIdx.y names a GROUP of sibling fronts whose separators are the same rows of their
// parent (config 3: all 1999 leaves hang off the root's 128 columns) -- their updates add up element by element, so the
// workgroup sweeps the members one after the other into the same accumulators and stores ONE packed update, into the
// slot of the group's leader; the parent's extend-add skips the other members (MfmaArgs::chskip).  On config 3 the
// per-front updates were 13 GB written and 13 GB read back per Schur complement.
template <int NTA, int NTN, int PH, int OCC, int KSQ, bool PIPE, bool EXACT, bool GRP = false>
__global__ void __launch_bounds__(256, OCC) k_lfsp_up(MfmaArgs a, double* u, int64_t ldu) {
  static_assert(!GRP || PH == 2, "groups exist for the update phase only");
  constexpr int EC = LFSP_ECAP;
  __shared__ int s_ta[2 * EC + 4], s_tb[2 * EC + 4];     // NN terms: columns (ta, tb) and value tv
  __shared__ double s_tv[2 * EC + 4];
  __shared__ int s_ar[EC + 4 * NTA + 4], s_aj[EC + 4 * NTA + 4], s_ao[EC + 4 * NTA + 4];   // s_aj: column offset j * nf; s_ao: ra * na             // AN entries (row inside A, column, value), grouped by row tile
  __shared__ double s_av[EC + 4 * NTA + 4];
  __shared__ int s_bkt[NTA + 2];                         // start of the entries of row tile b in s_a*
  __shared__ int s_n[2];                                 // NN terms (padded to a multiple of 4), AN entries
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int g0 = GRP ? a.grp_ptr[blockIdx.y] : 0, nmem = GRP ? a.grp_ptr[blockIdx.y + 1] - g0 : 1;
  // (the leader rotates with the group index: the parent's extend-add deals its children round-robin over several
  // workgroups, and leaders at every eighth position would all land on the same few)
  const int kleader = GRP ? a.grp_list[g0 + (int)(blockIdx.y % (unsigned)nmem)] : 0;
  d4 uA[1][NTA], uB[1][NTA];                    // GRP: the two tile rows of this wave, summed over the members
  for (int r = blockIdx.x; r < a.nrhs; r += gridDim.x)
  for (int mi = 0; mi < nmem; ++mi) {
  const int k = GRP ? a.grp_list[g0 + mi] : a.t.lev[blockIdx.y];   // right-hand-side groups of one front are neighbours in the launch
  const CliqueDesc d = a.t.cl[k];              // order: the fronts in flight at any time are few and their constants stay in L2
  const int nn = d.nn, na = d.na;
  const int nf = nn + na;                       // panels of these fronts have < 2^31 / 8 elements: 32-bit element offsets
  const double* Li = a.LK + d.blk;
  const double* Kk = Li + nn;
  const double* MK = a.sp_mk + d.blk + nn;
  const double* Rt = a.sp_rt + d.upd;
  const int32_t* kp = a.kc_ptr + (int64_t)k * a.kc_stride;
  {
    const int j = a.kc_ids ? a.kc_ids[a.kc_j0 + r] : a.kc_j0 + r;
    const int e0 = kp[j], e1 = kp[j + 1];
    __syncthreads();                                     // the tables of the previous right-hand side have been consumed
    if (wave == 0 && !((a.skip & 256) && r != (int)blockIdx.x)) {
      // ---- term tables, in the order of the entry list (deterministic): wave 0, 64 entries per pass
      int cntb[NTA];
#pragma unroll
      for (int b = 0; b < NTA; ++b) cntb[b] = 0;
      for (int base = e0; base < e1; base += 64) {
        const int e = base + lane;
        const bool valid = e < e1;
        const int off = a.kc_off[min(e, e1 - 1)];
        const int row = off % nf;
        const bool isAN = valid && row >= nn;
        const int b = (row - nn) >> 4;
#pragma unroll
        for (int bb = 0; bb < NTA; ++bb) cntb[bb] += __popcll(__ballot(isAN && b == bb));
      }
      int st[NTA + 1];                                   // every row tile's entries padded to a multiple of four (zero values)
      st[0] = 0;
#pragma unroll
      for (int b = 0; b < NTA; ++b) st[b + 1] = st[b] + ((cntb[b] + 3) & ~3);
      if (lane <= NTA) {
        int v = 0;
#pragma unroll
        for (int b = 0; b <= NTA; ++b) if (lane == b) v = st[b];
        s_bkt[lane] = v;
      }
      int run[NTA];
#pragma unroll
      for (int b = 0; b < NTA; ++b) run[b] = 0;
      int nnbase = 0;
      for (int base = e0; base < e1; base += 64) {
        const int e = base + lane;
        const bool valid = e < e1;
        const int off = a.kc_off[min(e, e1 - 1)];
        const double val = a.kc_val[min(e, e1 - 1)];
        const int row = off % nf, col = off / nf;
        const bool isAN = valid && row >= nn, isNN = valid && row < nn, isOff = isNN && row != col;
        const int b = (row - nn) >> 4;
#pragma unroll
        for (int bb = 0; bb < NTA; ++bb) {
          const unsigned long long m = __ballot(isAN && b == bb);
          if (isAN && b == bb) {
            const int pos = st[bb] + run[bb] + lfsp_below(m);
            s_ar[pos] = row - nn; s_aj[pos] = col * nf; s_ao[pos] = (row - nn) * na; s_av[pos] = val;
          }
          run[bb] += __popcll(m);
        }
        const unsigned long long mNN = __ballot(isNN), mOff = __ballot(isOff);
        const int pos = nnbase + lfsp_below(mNN) + lfsp_below(mOff);
        if (isNN) { s_ta[pos] = row * nf; s_tb[pos] = col * nf; s_tv[pos] = val; }       // element offsets of the two columns
        if (isOff) { s_ta[pos + 1] = col * nf; s_tb[pos + 1] = row * nf; s_tv[pos + 1] = val; }
        nnbase += __popcll(mNN) + __popcll(mOff);
      }
      const int nnpad = (nnbase + 3) & ~3;
      if (lane < nnpad - nnbase) { s_ta[nnbase + lane] = 0; s_tb[nnbase + lane] = 0; s_tv[nnbase + lane] = 0.0; }
#pragma unroll
      for (int b = 0; b < NTA; ++b)
        if (lane < st[b + 1] - st[b] - cntb[b]) {
          const int pos = st[b] + cntb[b] + lane;
          s_ar[pos] = 16 * b; s_aj[pos] = 0; s_ao[pos] = 0; s_av[pos] = 0.0;
        }
      if (lane == 0) { s_n[0] = nnpad; s_n[1] = st[NTA]; }
    }
    __syncthreads();
    const int TN = s_n[0], ANp = s_n[1];
    double* P = u + (int64_t)r * ldu + d.blk;
    // Every phase keeps its results in the accumulators until ALL operand loads of the workgroup have been issued and
    // consumed: vmcnt is one in-order counter, so a load that follows a streaming store is not visible to its consumer
    // before that store has been acknowledged by HBM (microseconds).  With one right-hand side per workgroup (the
    // default) the stores are the last thing a workgroup does.  PH selects the phases of this launch (1 Q, 2 Upd, 4 G_NN).
    const bool nold = (a.skip & 1024) != 0;
    constexpr int RPW = (NTA + 3) / 4;
    d4 q[RPW][NTN], u1[1][NTA], g[1][NTN];
    const int tm0 = wave, tm1 = NTA - 1 - wave;
    const bool has0 = 16 * tm0 < na, has1 = tm1 > tm0 && 16 * tm1 < na, hasg = wave < NTN && 16 * wave < nn;
    // ---- Q = sum_AN v rt_r l_j^T - sum_NN v mk_a l_b^T : wave w owns row tiles w * RPW .. of the na x nn block
    if (PH & 1) {
#pragma unroll
      for (int x = 0; x < RPW; ++x)
#pragma unroll
        for (int y = 0; y < NTN; ++y) q[x][y] = d4{0.0, 0.0, 0.0, 0.0};
      auto all = [](int, int) { return true; };
      // (the scale v of a term goes to the LEFT operand: RPW products per k-step instead of NTN)
      lfsp_rank<KSQ, RPW, NTN, PIPE>(q, 0, TN, kq,
          [&](int x, int t) { const int m = 16 * (RPW * wave + x) + l15; const double v = MK[(EXACT ? m : min(m, na - 1)) + s_ta[t]]; return (EXACT || m < na) ? v : 0.0; },
          [&](int y, int t) { const int n = 16 * y + l15; const double v = Li[(EXACT ? n : min(n, nn - 1)) + s_tb[t]]; return (EXACT || n < nn) ? v : 0.0; },
          [&](int t) { return -s_tv[t]; }, all, nold);
      lfsp_rank<KSQ, RPW, NTN, PIPE>(q, 0, ANp, kq,
          [&](int x, int t) { const int m = 16 * (RPW * wave + x) + l15; const double v = Rt[(EXACT ? m : min(m, na - 1)) + s_ao[t]]; return (EXACT || m < na) ? v : 0.0; },
          [&](int y, int t) { const int n = 16 * y + l15; const double v = Li[(EXACT ? n : min(n, nn - 1)) + s_aj[t]]; return (EXACT || n < nn) ? v : 0.0; },
          [&](int t) { return s_av[t]; }, all, nold);
    }
    // ---- Upd = sum_NN v k_a k_b^T - sum_AN v (k_j e_r^T + e_r k_j^T), lower tiles; wave w owns the tile rows w and NTA-1-w
    auto upd_row = [&](auto& ua, int tm, bool fresh) {
      constexpr int NR = sizeof(ua[0]) / sizeof(d4);
      if (fresh)
#pragma unroll
      for (int tn = 0; tn < NR; ++tn) ua[0][tn] = d4{0.0, 0.0, 0.0, 0.0};
      const int m = 16 * tm + l15, mc = EXACT ? m : min(m, na - 1);
      auto low = [&](int, int tn) { return tn <= tm; };
      lfsp_rank<1, 1, NR, PIPE>(ua, 0, TN, kq,
          [&](int, int t) { const double v = Kk[mc + s_ta[t]]; return (EXACT || m < na) ? v : 0.0; },
          [&](int tn, int t) { const int n = 16 * tn + l15; const double v = Kk[(EXACT ? n : min(n, na - 1)) + s_tb[t]]; return (tn <= tm && (EXACT || n < na)) ? v : 0.0; },
          [&](int t) { return s_tv[t]; }, low, nold);
      // - e_r k_j^T for the entries whose row lies in tile row tm: left = -v times the unit vector (no load), right = k_j
      lfsp_rank<1, 1, NR, PIPE>(ua, s_bkt[tm], s_bkt[tm + 1], kq,
          [&](int, int t) { return l15 == (s_ar[t] & 15) ? 1.0 : 0.0; },
          [&](int tn, int t) { const int n = 16 * tn + l15; const double v = Kk[(EXACT ? n : min(n, na - 1)) + s_aj[t]]; return (tn <= tm && (EXACT || n < na)) ? v : 0.0; },
          [&](int t) { return -s_av[t]; }, low, nold);
      // - k_j e_r^T for the entries whose row is a column of tile (tm, tn): left = k_j, right = -v times the unit vector.
      // Tile (tm, tn) takes the entries of row tile tn (a handful: one or two k-steps); the left operands of the first
      // two k-steps of four tiles at a time are requested together, longer lists (rare) finish in a plain loop
#pragma unroll
      for (int h = 0; h < NR; h += 4) {
        if (h > tm) break;
        double kc[2][4], bc[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int tn = h + i;
          const int b0 = tn < NR ? s_bkt[tn] : 0, n4 = (tn < NR && tn <= tm) ? s_bkt[tn + 1] - b0 : 0;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bool in = 4 * s2 < n4;
            const int t = in ? b0 + 4 * s2 + kq : 0;
            kc[s2][i] = Kk[mc + (in ? s_aj[t] : 0)];            // (a pair without AN entries leaves the tables unwritten)
            bc[s2][i] = (in && l15 == (s_ar[t] & 15)) ? -s_av[t] : 0.0;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (h + i < NR && h + i <= tm) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
              ua[0][h + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bc[s2][i], (EXACT || m < na) ? kc[s2][i] : 0.0, ua[0][h + i], 0, 0, 0);
          }
      }
#pragma unroll
      for (int tn = 0; tn < NR; ++tn)
        if (tn <= tm) {
          for (int t0 = s_bkt[tn] + 8; t0 < s_bkt[tn + 1]; t0 += 4) {
            const int t = t0 + kq;
            const double kv = Kk[mc + s_aj[t]];
            const double bv = l15 == (s_ar[t] & 15) ? -s_av[t] : 0.0;
            ua[0][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, (EXACT || m < na) ? kv : 0.0, ua[0][tn], 0, 0, 0);
          }
        }
    };
    double* UP = a.t.updp + (int64_t)r * a.t.updplen + (GRP ? a.t.cl[kleader].updp : d.updp);
    auto put = [&](auto& ua, int tm) {
      constexpr int NR = sizeof(ua[0]) / sizeof(d4);
      const int m = 16 * tm + l15;
      if (a.skip & 512) return;
#pragma unroll
      for (int tn = 0; tn < NR; ++tn)
        if (tn <= tm) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int n = 16 * tn + kq + 4 * rr;
            if ((EXACT || m < na) && n <= m) UP[pk_idx(m, n, na)] = ua[0][tn][rr];
          }
        }
    };
    if (PH & 2) {
      if constexpr (GRP) {
        if (has0) upd_row(uA, tm0, mi == 0);
        if (has1) upd_row(uB, tm1, mi == 0);
        if (mi == nmem - 1) {
          if (has0) put(uA, tm0);
          if (has1) put(uB, tm1);
        }
      } else {                                       // the shorter row first; its stores overlap the longer row's products
        if (has0) { upd_row(u1, tm0, true); put(u1, tm0); }
        if (has1) { upd_row(u1, tm1, true); put(u1, tm1); }
      }
    }
    // ---- G_NN = sum_NN v l_a l_b^T, lower tiles: wave w owns tile row w
    if ((PH & 4) && hasg) {
      const int tm = wave, m = 16 * tm + l15;
#pragma unroll
      for (int tn = 0; tn < NTN; ++tn) g[0][tn] = d4{0.0, 0.0, 0.0, 0.0};
      lfsp_rank<KSQ, 1, NTN, PIPE>(g, 0, TN, kq,
          [&](int, int t) { const double v = Li[(EXACT ? m : min(m, nn - 1)) + s_ta[t]]; return (EXACT || m < nn) ? v : 0.0; },
          [&](int tn, int t) { const int n = 16 * tn + l15; const double v = Li[(EXACT ? n : min(n, nn - 1)) + s_tb[t]]; return (tn <= tm && (EXACT || n < nn)) ? v : 0.0; },
          [&](int t) { return s_tv[t]; }, [&](int, int tn) { return tn <= tm; }, nold);
    }
    // ---- the results leave
    if (a.skip & 512) continue;
    if (PH & 1) {
#pragma unroll
      for (int x = 0; x < RPW; ++x)
#pragma unroll
        for (int y = 0; y < NTN; ++y)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int m = 16 * (RPW * wave + x) + l15, n = 16 * y + kq + 4 * rr;
            if (m < na && n < nn) P[nn + m + n * nf] = q[x][y][rr];
          }
    }
    if ((PH & 4) && hasg) {
      const int tm = wave, m = 16 * tm + l15;
#pragma unroll
      for (int tn = 0; tn < NTN; ++tn)
        if (tn <= tm) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int n = 16 * tn + kq + 4 * rr;
            if (m < nn && n <= m) P[m + n * nf] = g[0][tn][rr];
          }
        }
    }
  }
  }
}

}  // namespace smcp
