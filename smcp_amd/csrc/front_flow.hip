// One-launch blocked Cholesky of ONE dense symmetric positive definite matrix (order 129 .. 4096) with the dependencies
// between its 64 x 64 tiles resolved INSIDE the launch (round 5; VERDICT r4 "what's weak" #3).
//
// The per-step route (k_lf_diag, k_lf_chol_panel, k_lf_chol_trail: three launches per 64 columns) leaves the chip idle
// during every diagonal-block step and pays a launch boundary three times per step: 49 us per step, 16 / 9 / 12 steps for
// potrf(H), the root front and chol(Y_AA) of the max-cut configuration, 64 steps for the 4096 front of config 2.  Here the
// matrix is cut into tiles (i, k), k <= i, each OWNED by one workgroup of a persistent grid (static, balanced on the host:
// flow_plan): only the owner ever reads or writes the tile in the matrix itself, so those accesses need no coherence
// protocol at all.  What crosses workgroups is published once, write-through, in buffers nobody has read before:
//   * P(i, j) -- the final panel tile L(i, j) -- as a packed 64 x 64 copy in a workspace, flag pflag[tile],
//   * Dinv(j) -- the inverse of the diagonal block L(j, j) (what dense_potrs reads afterwards as well) --, flag dflag[j].
// (MI355X_MICROARCH.md, inter-workgroup visibility: payload stored sc1 by every lane, every storing wave drained, workgroup
// barrier, ONE lane's sc1 flag store; the consumer polls the flag with ONE sc1 load per poll, a workgroup barrier, then
// every load of the payload is an sc1 load, which bypasses the CU's L1 -- so two workgroups may share a CU.)
// A tile (i, k) goes through k updates  A(i, k) -= P(i, j) P(k, j)^T, j = 0 .. k - 1  (needs pflag of both panel tiles) and
// one final step: the diagonal tile is factored and inverted in LDS (potrf_inv64), a panel tile is multiplied by
// Dinv(k)^T (needs dflag[k]).  The owner of the diagonal tile (j, j) also owns (j, j - 1): the critical path of a step --
// Dinv(j - 1) arrives, P(j, j - 1) = A(j, j - 1) Dinv^T, A(j, j) -= P P^T, potrf of (j, j) -- stays inside one workgroup.
// Scheduling is ready-driven, not step-synchronous: lane l of a workgroup's first wave tracks own tile l (updates applied,
// final or not), every round each lane polls the flags its tile's next action needs, and the first ready tile in
// (column, row) order is processed by the whole workgroup.  A workgroup never blocks on a particular flag, so the scheme
// cannot deadlock as long as every workgroup of the grid is resident (the grid is at most FLOW_MAXWG workgroups of 77 KB
// LDS each: four such launches fit the chip side by side -- ranks sharing a GPU in the tests); every wait is bounded all
// the same: after FLOW_TIMEOUT_TICKS without progress a workgroup raises the abort flag and the call fails with SMCP_ETIMEOUT.
#include <hip/hip_runtime.h>

namespace smcp {

constexpr int FLOW_LD = 65;                       // LDS leading dimension of a staged tile
constexpr int FLOW_MAXOWN = 64;                   // own tiles per workgroup: one lane of the scheduling wave each
constexpr int FLOW_MAXWG = 112;                   // workgroups per launch (two fit a CU: 4 x 112 <= 2 x 256)
constexpr int FLOW_MAXN = 4096;
constexpr long long FLOW_TIMEOUT_TICKS = 300000000ll;   // 3 s of the 100 MHz counter
constexpr size_t FLOW_LDS_BYTES = (size_t)(2 * 64 * FLOW_LD + 1024 + 8) * sizeof(double);   // 75 KB: D / Di (or two operand tiles), s16

struct FlowArgs {
  double* A; int64_t ld; int n;      // the matrix (lower triangle read and overwritten by its factor)
  double* dinv;                      // block j at dinv + j * 4096: w_j x w_j inverse of L(j, j), leading dimension w_j
  double* P;                         // ntiles packed 64 x 64 panel tiles
  unsigned* flags;                   // pflag[ntiles], dflag[nt], abort[1]
  unsigned epoch;                    // value of a set flag in this launch (flags are never cleared between launches)
  const int32_t* own_ptr;            // per workgroup: first own tile
  const int32_t* own_tile;           // own tiles as (i << 16 | k), sorted by (k, i)
  int* info; int info_val;           // failure flag of the caller (not positive definite: info_val)
  // the matrix of a front instead (lev != null: clique lev[0]): mode 0 = its panel in A (a front WITHOUT separator: nn x nn,
  // leading dimension nn), mode 2 = its Y_AA block in A (update layout: na x na); info / info_val then follow from the clique
  const CliqueDesc* cl; const int32_t* lev; int mode; int nsn1;
  long long* dbg;                    // timing studies (SMCP_FLOW_STAMPS=1): per workgroup 8 words -- ticks idle / updating / factoring / panel, counts of each
  // several fronts of one level side by side (gridDim.y of them, front f = clique lev[f]): every front has its own slice of the
  // workspace (P, dinv, flags: strides in elements) laid out for the LARGEST order n; a front's own order comes from its clique
  int64_t p_stride, dinv_stride; int flag_stride;
};

__host__ __device__ inline int flow_tile_id(int i, int k) { return i * (i + 1) / 2 + k; }
__device__ inline unsigned flow_ld_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void flow_st_flag(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double flow_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void flow_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every storing wave has drained its stores; the workgroup's barrier; ONE lane signals
__device__ inline void flow_publish(unsigned* flag, unsigned epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) flow_st_flag(flag, epoch);
}
// a packed 64 x 64 tile published by another workgroup -> LDS (leading dimension FLOW_LD); 16-byte sc1 loads (aux 16), eight in
// flight per lane (8-byte sc1 accesses run at 0.54 - 0.70 of the 16-byte rate, MI355X_MICROARCH.md)
typedef unsigned int flow_u4 __attribute__((ext_vector_type(4)));
__device__ inline void flow_fetch_tile(const double* src, double* T) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(src), 0, 4096 * 8, 0x00020000);
  flow_u4 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * (threadIdx.x + 256 * u), 0, 16);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = 2 * (threadIdx.x + 256 * u);
    double2 d2 = __builtin_bit_cast(double2, v[u]);
    double* q = T + (e & 63) + (e >> 6) * FLOW_LD;
    q[0] = d2.x; q[1] = d2.y;
  }
}
// acc (four waves, 2 x 2 MFMA tiles each) = Ta Tb^T over the 64 staged columns: out(m, n) = sum_k Ta[m + k ld] Tb[n + k ld]
__device__ inline void flow_mma(d4 (&acc)[2][2], const double* Ta, const double* Tb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kq = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  tile64_zero(acc);
#pragma unroll 4
  for (int k0 = 0; k0 < 64; k0 += 4) {
    const int kk = k0 + kq;
    const double a0 = Ta[(32 * wm + l15) + kk * FLOW_LD], a1 = Ta[(32 * wm + 16 + l15) + kk * FLOW_LD];
    const double b0 = Tb[(32 * wn + l15) + kk * FLOW_LD], b1 = Tb[(32 * wn + 16 + l15) + kk * FLOW_LD];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
  }
}

__global__ void __launch_bounds__(256) k_chol_flow(FlowArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const T0 = smem;                       // operand tile / the diagonal block D
  double* const T1 = T0 + 64 * FLOW_LD;          // operand tile / its inverse Di
  double* const s16 = T1 + 64 * FLOW_LD;         // 1024 doubles: the 16 x 16 block inverses, then the scratch of the inverse's assembly
  __shared__ int s_pick, s_ti, s_tk, s_pg, s_fuse;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntL = (a.n + 63) >> 6, ntilesL = ntL * (ntL + 1) / 2;       // the layout of the flags (and the plan): the largest order
  int n = a.n;
  if (a.lev) {
    const int kc = a.lev[blockIdx.y];
    const CliqueDesc d = a.cl[kc];
    if (a.mode == 0) { a.A += d.blk; a.ld = d.nn + d.na; n = d.nn; } else { a.A += d.upd; a.ld = d.na; n = d.na; }
    a.info += kc / a.nsn1; a.info_val = kc % a.nsn1 + 1;
    a.P += (int64_t)blockIdx.y * a.p_stride; a.dinv += (int64_t)blockIdx.y * a.dinv_stride; a.flags += (int64_t)blockIdx.y * a.flag_stride;
  }
  const int nt = (n + 63) >> 6;
  if (*a.info) return;                             // (an earlier level of this factorisation has failed: as the per-step kernels)
  const int64_t ld = a.ld;
  unsigned* const pflag = a.flags;
  unsigned* const dflag = a.flags + ntilesL;
  unsigned* const abortf = dflag + ntL;
  const unsigned epoch = a.epoch;
  const int t0 = a.own_ptr[blockIdx.x], cnt = a.own_ptr[blockIdx.x + 1] - t0;
  // scheduling state of the first wave: lane l <-> own tile l
  int ti = 0, tk = 0, pg = 0;
  bool fin = true;
  if (wave == 0 && lane < cnt) {
    const int packed = a.own_tile[t0 + lane];
    ti = packed >> 16; tk = packed & 0xffff; fin = ti >= nt;      // (a tile beyond this front's own order: nothing to do, nobody waits for it)
  }
  long long tlast = wall_clock64();
  long long tmark = tlast, acct[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sub[4] = {0, 0, 0, 0}, tsub = 0;
  auto stamp = [&](int what) {
    if (a.dbg && tid == 0) { const long long t = wall_clock64(); acct[what] += t - tmark; acct[4 + what] += 1; tmark = t; }
  };
  // the diagonal tile kd (order wd, at Ad in the matrix) stands in T0 (lower triangle, zeros elsewhere): factor, invert, store, publish
  auto factor_diag = [&](int kd, int wd, double* Ad) -> bool {
    if (a.dbg && tid == 0) { tsub = wall_clock64(); sub[0] += tsub - tmark; }
    const int f = potrf_inv64_fast(T0, wd, T1, s16, s16);
    if (a.dbg && tid == 0) { const long long t = wall_clock64(); sub[1] += t - tsub; tsub = t; }
    if (f) {
      if (tid == 0) { atomicCAS(a.info, 0, a.info_val); flow_st_flag(abortf, epoch); }
      return true;
    }
    double* const Dg = a.dinv + (int64_t)kd * 4096;
    for (int e = tid; e < wd * wd; e += 256) {
      const int r = e % wd, c = e / wd;
      if (r >= c) Ad[r + (int64_t)c * ld] = T0[r + c * FLOW_LD];
      flow_st(Dg + e, T1[r + c * FLOW_LD]);
    }
    flow_publish(dflag + kd, epoch);
    if (a.dbg && tid == 0) { const long long t = wall_clock64(); sub[2] += t - tsub; }
    return false;
  };
  for (;;) {
    if (wave == 0) {
      bool ready = false;
      if (!fin) {
        if (pg == tk) ready = (ti == tk) || flow_ld_flag(dflag + tk) == epoch;
        else {
          const unsigned f1 = flow_ld_flag(pflag + flow_tile_id(ti, pg));
          const unsigned f2 = (ti == tk) ? f1 : flow_ld_flag(pflag + flow_tile_id(tk, pg));
          ready = f1 == epoch && f2 == epoch;
        }
      }
      const unsigned long long rb = __ballot(ready), open = __ballot(!fin);
      int pick = -2;                              // -2: every own tile is final
      if (open) pick = rb ? __ffsll((long long)rb) - 1 : -1;
      if (pick == -1) {                           // nothing to do yet: has somebody given up?  have we waited too long?
        int stop = 0;
        if (lane == 0) {
          if (flow_ld_flag(abortf) == epoch) stop = 1;
          else if (wall_clock64() - tlast > FLOW_TIMEOUT_TICKS) {
            atomicCAS(a.info, 0, SMCP_ETIMEOUT);
            flow_st_flag(abortf, epoch);
            stop = 1;
          }
        }
        if (__builtin_amdgcn_readfirstlane(stop)) pick = -3;
        else __builtin_amdgcn_s_sleep(16);
      }
      // a sub-diagonal panel tile (j, j - 1) about to be finished whose diagonal neighbour (j, j) -- same owner by the plan -- lacks
      // exactly this tile's update: the three steps of the critical path (panel product, A(j, j) -= P P^T, Cholesky + inverse of the
      // diagonal tile) run as ONE action, the intermediate results never leaving the workgroup
      int fuse = -1;
      if (pick >= 0) {
        const int pi = __shfl(ti, pick), pk = __shfl(tk, pick), pp = __shfl(pg, pick);
        const unsigned long long fm = __ballot(!fin && ti == pi && tk == pi && pg == pk && pp == pk && pi == pk + 1);
        if (fm) fuse = __ffsll((long long)fm) - 1;
      }
      if (lane == 0) { s_pick = pick; s_fuse = fuse; }
      if (pick >= 0 && lane == pick) { s_ti = ti; s_tk = tk; s_pg = pg; }
    }
    __syncthreads();
    const int pick = s_pick;
    if (pick <= -2) break;
    if (pick == -1) { __syncthreads(); continue; }
    stamp(0);
    const int i = s_ti, k = s_tk, g = s_pg;
    const int hi = min(64, n - 64 * i), wk = min(64, n - 64 * k);     // rows of tile row i, columns of tile column k
    double* const At = a.A + 64 * i + (int64_t)(64 * k) * ld;        // the tile in the matrix: owner-only, plain accesses
    bool failed = false;
    if (g < k) {
      // ---- update: A(i, k) -= P(i, g) P(k, g)^T   (the tile's own sixteen values per lane are requested FIRST: they travel while
      // the panel tiles are fetched, staged and multiplied)
      double old[2][2][4];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int m, nn_;
            tile64_pos(x, y, r, m, nn_);
            old[x][y][r] = At[min(m, hi - 1) + (int64_t)min(nn_, wk - 1) * ld];
          }
      flow_fetch_tile(a.P + (int64_t)flow_tile_id(i, g) * 4096, T0);
      if (i != k) flow_fetch_tile(a.P + (int64_t)flow_tile_id(k, g) * 4096, T1);
      __syncthreads();
      d4 acc[2][2];
      flow_mma(acc, T0, i != k ? T1 : T0);
      const bool diag = i == k;
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int m, nn_;
            tile64_pos(x, y, r, m, nn_);
            if (m < hi && nn_ < wk && (!diag || m >= nn_)) At[m + (int64_t)nn_ * ld] = old[x][y][r] - acc[x][y][r];
          }
    } else if (i == k) {
      // ---- diagonal tile: factor and invert in LDS
      for (int e = tid; e < 64 * 64; e += 256) {
        const int r = e & 63, c = e >> 6;
        T0[r + c * FLOW_LD] = (r < wk && c < wk && r >= c) ? At[r + (int64_t)c * ld] : 0.0;
      }
      __syncthreads();
      failed = factor_diag(k, wk, At);
    } else {
      // ---- panel tile: P(i, k) = A(i, k) Dinv(k)^T   (k < nt - 1 here: the column is 64 wide)
      flow_fetch_tile(a.dinv + (int64_t)k * 4096, T1);          // Dinv(n, kk) at T1[n + kk ld]
      for (int e = tid; e < 64 * 64; e += 256) {
        const int r = e & 63, c = e >> 6;
        T0[r + c * FLOW_LD] = r < hi ? At[r + (int64_t)c * ld] : 0.0;
      }
      __syncthreads();
      d4 acc[2][2];
      flow_mma(acc, T0, T1);
      double* const Pg = a.P + (int64_t)flow_tile_id(i, k) * 4096;
      tile64_foreach(acc, 0, 0, 64, 64, [=](int m, int nn_, double v) {
        if (m < hi) At[m + (int64_t)nn_ * ld] = v;
        flow_st(Pg + m + 64 * nn_, v);
      });
      flow_publish(pflag + flow_tile_id(i, k), epoch);
      if (s_fuse >= 0) {
        // the fused continuation: A(i, i) -= P P^T with P still in registers, then the diagonal tile's own final step
        tile64_foreach(acc, 0, 0, 64, 64, [=](int m, int nn_, double v) { T0[m + nn_ * FLOW_LD] = v; });      // (the barrier of flow_publish lies behind every read of T0)
        __syncthreads();
        d4 acc2[2][2];
        flow_mma(acc2, T0, T0);
        __syncthreads();
        double* const Ad = a.A + 64 * i + (int64_t)(64 * i) * ld;
        tile64_rmw(acc2, 0, 0, 64, 64, [=](int m, int nn_) { return (m < hi && nn_ < hi) ? Ad[min(m, hi - 1) + (int64_t)min(nn_, hi - 1) * ld] : 0.0; },
                   [=](int m, int nn_, double v, double o) { T0[m + nn_ * FLOW_LD] = (m >= nn_ && m < hi && nn_ < hi) ? o - v : 0.0; });
        __syncthreads();
        stamp(3);
        failed = factor_diag(i, hi, Ad);
      }
    }
    stamp(g < k ? 1 : (i == k || s_fuse >= 0 ? 2 : 3));
    if (wave == 0 && lane == pick) { if (g < k) ++pg; else fin = true; }
    if (wave == 0 && s_fuse >= 0 && lane == s_fuse) { ++pg; fin = true; }
    if (wave == 0) tlast = wall_clock64();
    __syncthreads();            // LDS (tiles, s_pick) is reused by the next round
    if (failed) break;
  }
  if (a.dbg && tid == 0) { stamp(0); for (int q = 0; q < 8; ++q) a.dbg[8 * blockIdx.x + q] = acct[q]; if (blockIdx.x == 1) for (int q = 0; q < 3; ++q) a.dbg[8 * blockIdx.x + 5 + q] = sub[q]; }
}

// ---- host: ownership plan ------------------------------------------------------------------------------------------------
struct FlowPlan {
  int n = 0, nt = 0, nwg = 0;
  int32_t* own_ptr = nullptr; int32_t* own_tile = nullptr;     // device
};
// greedy balance (longest processing time first) of the tiles over the workgroups; the diagonal tile (j, j) and its left
// neighbour (j, j - 1) always go together (the critical path of a step stays inside one workgroup), and consecutive
// diagonal tiles go to DIFFERENT workgroups (while one factors, the next prepares).  Work of a tile ~ its k updates + one
// final step; a diagonal tile's final step (potrf + inverse in LDS) counts as several tile products.
inline void flow_make_plan(int n, int maxwg, std::vector<int32_t>& own_ptr, std::vector<int32_t>& own_tile) {
  const int nt = (n + 63) / 64, ntiles = nt * (nt + 1) / 2;
  const int nwg = std::max(1, std::min(maxwg, ntiles - (nt - 1)));          // (diagonal pairs count once)
  struct Item { double w; std::vector<int32_t> tiles; };
  std::vector<Item> items;
  for (int j = 0; j < nt; ++j) {
    Item it; it.w = j + 8.0; it.tiles.push_back(j << 16 | j);
    if (j > 0) { it.w += (j - 1) + 1.0; it.tiles.push_back(j << 16 | (j - 1)); }
    items.push_back(it);
  }
  for (int i = 0; i < nt; ++i)
    for (int k = 0; k + 1 < i; ++k) { Item it; it.w = k + 1.0; it.tiles.push_back(i << 16 | k); items.push_back(it); }
  std::vector<double> load((size_t)nwg, 0.0);
  std::vector<std::vector<int32_t>> own((size_t)nwg);
  // the diagonal pairs first, round-robin (consecutive steps on different workgroups)
  for (int j = 0; j < nt; ++j) {
    const int w = j % nwg;
    for (int32_t t : items[(size_t)j].tiles) own[(size_t)w].push_back(t);
    load[(size_t)w] += items[(size_t)j].w;
  }
  std::vector<int> order;
  for (int q = nt; q < (int)items.size(); ++q) order.push_back(q);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return items[(size_t)x].w > items[(size_t)y].w; });
  for (int q : order) {
    int best = 0;
    for (int w = 1; w < nwg; ++w)
      if (load[(size_t)w] < load[(size_t)best] || (load[(size_t)w] == load[(size_t)best] && own[(size_t)w].size() < own[(size_t)best].size())) best = w;
    if ((int)own[(size_t)best].size() >= FLOW_MAXOWN) {           // (cannot happen for n <= 4096 and >= 33 workgroups; keep the table sound anyway)
      for (int w = 0; w < nwg; ++w) if ((int)own[(size_t)w].size() < (int)own[(size_t)best].size()) best = w;
    }
    own[(size_t)best].push_back(items[(size_t)q].tiles[0]);
    load[(size_t)best] += items[(size_t)q].w;
  }
  own_ptr.assign(1, 0);
  own_tile.clear();
  for (int w = 0; w < nwg; ++w) {
    std::sort(own[(size_t)w].begin(), own[(size_t)w].end(), [](int32_t x, int32_t y) {
      const int xi = x >> 16, xk = x & 0xffff, yi = y >> 16, yk = y & 0xffff;
      return xk != yk ? xk < yk : xi < yi;
    });
    for (int32_t t : own[(size_t)w]) own_tile.push_back(t);
    own_ptr.push_back((int32_t)own_tile.size());
  }
}

}  // namespace smcp
