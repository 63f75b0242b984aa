// Dropped in round 3 (measured: no gain, the extend-add is bound by the ~2.8 TB/s the memory system delivers for this
// read stream, not by the bytes in flight): flat walk over 64-row children for k_lf_assemble_lds.
// Children with exactly 64 separator rows (the 112 children of a (64,128) front of synth50k), read FLAT: the packed update
// of a child is 2080 contiguous doubles, lane l takes the elements l, l + 64, ... -- every lane busy in every load where the
// column-by-column walk above leaves half of them masked (columns of a lower triangle), so twice the bytes are in flight
// for the same registers: one workgroup per CU (the 148 KB front) and ~7 us of memory latency under load make the bytes
// in flight the bound (Little's law: 65 KB per CU gave 9 GB/s per CU = 2.3 TB/s on the chip).  The (row, column) of an
// element depends only on its position: a per-lane table, 16 bits per element, built once per workgroup; the front rows
// rel[row], rel[column] come from the lane that holds them (ds_bpermute: no LDS space).  Two halves of 17 / 16 loads per
// child alternate between two register buffers: the next half is in flight while the current one is added.
constexpr int LF_FLAT_NA = 64, LF_FLAT_NP = LF_FLAT_NA * (LF_FLAT_NA + 1) / 2;      // 2080
__device__ inline void lf_flat_table(int lane, int (&ij)[17]) {
#pragma unroll
  for (int t = 0; t < 17; ++t) ij[t] = 0;
#pragma unroll
  for (int t = 0; t < 33; ++t) {
    const int e = lane + 64 * t;
    int i = 0, j = 0;
    if (e < LF_FLAT_NP) pk_unpack(e, LF_FLAT_NA, i, j);
    ij[t >> 1] |= (i | (j << 8)) << (16 * (t & 1));
  }
}
__device__ inline void lf_add_children_flat64(double* T, int nf, const double* ubase, const int64_t* sCu, const int64_t* sCr,
                                              const int32_t* relidx, int nmine, int wave, int nw, int lane, const int (&ij)[17]) {
  auto cb = [nf](int j) { return j * nf - ((j * (j - 1)) >> 1) - j; };
  double va[17], vb[16];
  int qi = wave;
  if (qi >= nmine) return;
  const double* Uc = ubase + sCu[qi];
  int rel_cur = relidx[sCr[qi] + lane];
#pragma unroll
  for (int t = 0; t < 17; ++t) va[t] = Uc[lane + 64 * t];
  for (; qi < nmine; qi += nw) {
#pragma unroll
    for (int t = 0; t < 16; ++t) vb[t] = Uc[min(lane + 64 * (17 + t), LF_FLAT_NP - 1)];     // second half in flight (no branch around a
                                                                                            // load: the wait counts must stay countable)
#pragma unroll
    for (int t = 0; t < 17; ++t) {
      const int w = (ij[t >> 1] >> (16 * (t & 1))) & 0xffff;
      const int ri = __builtin_amdgcn_ds_bpermute(4 * (w & 0xff), rel_cur), rj = __builtin_amdgcn_ds_bpermute(4 * (w >> 8), rel_cur);
      unsafeAtomicAdd(&T[cb(rj) + ri], va[t]);
    }
    // first half of the next child in flight while the second half of this one is added
    const int qn = min(qi + nw, nmine - 1);            // (past the last child: its blocks once more, unused -- no branch)
    Uc = ubase + sCu[qn];
    const int rel_nxt = relidx[sCr[qn] + lane];
#pragma unroll
    for (int t = 0; t < 17; ++t) va[t] = Uc[lane + 64 * t];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int tt = 17 + t;
      const int w = (ij[tt >> 1] >> (16 * (tt & 1))) & 0xffff;
      const int ri = __builtin_amdgcn_ds_bpermute(4 * (w & 0xff), rel_cur), rj = __builtin_amdgcn_ds_bpermute(4 * (w >> 8), rel_cur);
      if (lane + 64 * tt < LF_FLAT_NP) unsafeAtomicAdd(&T[cb(rj) + ri], vb[t]);
    }
    rel_cur = rel_nxt;
  }
}
