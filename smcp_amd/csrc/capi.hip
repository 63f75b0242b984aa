// C-ABI implementation (include/smcp_amd.h): host-side drivers that walk the clique tree
// level by level and launch the HIP kernels.  No CPU compute fallback exists here.
#include <hip/hip_runtime.h>
#include <memory>
#include <unordered_map>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <atomic>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/smcp_amd.h"
#include "context.hpp"
#include "switches.hpp"
#include "front_generic.hip"
#include "front_mfma.hip"
#include "front_large.hip"
#include "front_inv.hip"
#include "front_n16.hip"
#include "front_fam.hip"
#include "front_fam2.hip"
#include "front_lfsp.hip"
#include "front_leafgram.hip"
#include "front_famt.hip"
#include "front_flow.hip"

using namespace smcp;

#define HIPCHK(x)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "smcp_amd: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return SMCP_EHIP;                                                                  \
    }                                                                                    \
  } while (0)

namespace {

// ---- kernel ids + optional per-kernel HIP-event timing (csp_profile_*) ------------------
enum {
  KID_chol_level = 0, KID_llt_level, KID_pinv_level, KID_gather_level, KID_completion_all,
  KID_hess_up_level, KID_hess_down_level, KID_hess_down_inv_all, KID_hess_up_inv_level, KID_scale_an,
  KID_factor_yaa, KID_trsm_fwd_level, KID_trsm_bwd_level, KID_amap, KID_aadj, KID_scatter_constraints,
  KID_dense_potrf, KID_dense_potrs, KID_vec_axpby, KID_axpby, KID_reduce_cliques, KID_reduce_final,
  KID_hess_up_mfma, KID_hess_down_mfma, KID_chol_mfma, KID_pinv_mfma, KID_prep_lk,
  KID_hess_up_mfma_hbm, KID_hess_down_mfma_hbm, KID_chol_mfma_hbm, KID_pinv_mfma_hbm,
  KID_gram_partial, KID_gram_reduce, KID_hess_up_pad,
  KID_lf_assemble, KID_lf_clear_upd, KID_lf_up1, KID_lf_up2, KID_lf_up3, KID_lf_down1, KID_lf_down2, KID_lf_down3,
  KID_lf_pinv1, KID_lf_pinv2, KID_lf_diag, KID_lf_chol_panel, KID_lf_chol_trail, KID_lf_pack_upd,
  KID_lf_prep_s, KID_lf_prep_row, KID_lf_prep_k, KID_factor_yaa_lds,
  KID_factor_inverse, KID_hess_down_inv_mfma, KID_hess_down_inv_mfma_hbm, KID_hess_up_inv_mfma, KID_hess_up_inv_mfma_hbm,
  KID_completion_mfma, KID_completion_mfma_hbm, KID_lf_copy_an, KID_lf_ri_an, KID_lf_dinv1, KID_lf_dinv2,
  KID_lf_uinv1, KID_lf_uinv2, KID_lf_completion, KID_hess_up_n16, KID_llt_mfma, KID_llt_mfma_hbm, KID_lf_llt,
  KID_hess_up_fam, KID_qr_rmul, KID_qr_dots, KID_qr_comb, KID_qr_small, KID_fam2_prep, KID_mid_chol, KID_lf_diag_inv, KID_lfsp_up, KID_lfsp_prep, KID_leaf_gram, KID_leaf_tables, KID_fam_sparse, KID_gram_diag128, KID_lf_assemble_lds, KID_fam_terms, KID_famt_prep, KID_lf_assemble_lds_dyn, KID_lf_zsp, KID_fam_terms_grp, KID_lf_assemble_fz, KID_factor_inverse_lds, KID_chol_flow,
  KID_COUNT
};
const char* const KID_NAMES[KID_COUNT] = {
  "k_chol_level", "k_llt_level", "k_pinv_level", "k_gather_level", "k_completion_all",
  "k_hess_up_level", "k_hess_down_level", "k_hess_down_inv_all", "k_hess_up_inv_level", "k_scale_an",
  "k_factor_yaa", "k_trsm_fwd_level", "k_trsm_bwd_level", "k_amap", "k_aadj", "k_scatter_constraints",
  "k_dense_potrf", "k_dense_potrs", "k_vec_axpby", "k_axpby", "k_reduce_cliques", "k_reduce_final",
  "k_hess_up_mfma<true>", "k_hess_down_mfma<true>", "k_chol_mfma<true>", "k_pinv_mfma<true>", "k_prep_lk",
  "k_hess_up_mfma<false>", "k_hess_down_mfma<false>", "k_chol_mfma<false>", "k_pinv_mfma<false>",
  "k_gram_partial", "k_gram_reduce", "k_hess_up_pad",
  "k_lf_assemble", "k_lf_clear_upd", "k_lf_up1", "k_lf_up2", "k_lf_up3", "k_lf_down1", "k_lf_down2", "k_lf_down3",
  "k_lf_pinv1", "k_lf_pinv2", "k_lf_diag", "k_lf_chol_panel", "k_lf_chol_trail", "k_lf_pack_upd",
  "k_lf_prep_s", "k_lf_prep_row", "k_lf_prep_k", "k_factor_yaa_lds",
  "k_factor_inverse", "k_hess_down_inv_mfma<true>", "k_hess_down_inv_mfma<false>", "k_hess_up_inv_mfma<true>",
  "k_hess_up_inv_mfma<false>", "k_completion_mfma<true>", "k_completion_mfma<false>", "k_lf_copy_an", "k_lf_ri_an",
  "k_lf_dinv1", "k_lf_dinv2", "k_lf_uinv1", "k_lf_uinv2", "k_lf_completion", "k_hess_up_n16",
  "k_llt_mfma<true>", "k_llt_mfma<false>", "k_lf_llt", "k_hess_up_fam",
  "k_stack_trsm", "k_stack_dots", "k_stack_comb", "k_qr_small", "k_fam2_prep", "k_mid_chol", "k_lf_diag_inv", "k_lfsp_up", "k_lfsp_prep", "k_leaf_pairs", "k_leaf_tables", "k_fam_sparse", "k_gram_diag128", "k_lf_assemble_lds", "k_fam_terms", "k_famt_prep", "k_lf_assemble_lds_dyn", "k_lf_zsp", "k_fam_terms_grp", "k_lf_assemble_fz", "k_factor_inverse_lds", "k_chol_flow"};

// A launch that the runtime refuses (bad configuration, LDS over the limit, ...) must reach the caller: the helpers
// record the first failure in the context and every entry point ends with end_call(), which returns it.
inline void note_launch(csp_ctx* c, int kid) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess && !c->launch_err) {
    c->launch_err = (int)e;
    fprintf(stderr, "smcp_amd: launch of %s failed: %s\n", (kid >= 0 && kid < KID_COUNT) ? KID_NAMES[kid] : "?", hipGetErrorString(e));
  }
}
inline hipError_t end_call(csp_ctx* c) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && c->launch_err) e = c->launch_err > 0 ? (hipError_t)c->launch_err : hipErrorLaunchFailure;
  c->launch_err = 0;
  return e;
}

// Independent clique-local stages (the inverse-form factor of the small cliques beside that of the large fronts; the
// Cholesky factors of the separator blocks of the leaves, of the mid fronts and of the large fronts) are launches of a
// few hundred workgroups or of eight: one after the other they leave most of the chip idle.  Fork puts a branch on a
// side stream that starts after everything already queued on the caller's stream and is joined back before anything
// that follows (events only, no host synchronisation).  SMCP_FORK=0: everything on the caller's stream.
// Events that only order streams of ONE device against each other: a device-scope release when they are recorded (the
// default, a system-scope release with its cache write-back, showed as ~12 us of idle stream behind every record in the
// rocprofv3 timeline of csp_cholesky_projected_inverse).  SMCP_EVENT_SYSFENCE=1: the default flags.
// ---- delay injection (tools/race_hunt.sh, tests/test_gpu_distributed.py::test_sharded_step_under_delay_injection) ------
// A missing stream edge shows only when the unordered side happens to run late.  With SMCP_RACE=<seed> (or
// csp_tune(ctx, CSP_TUNE_RACE, seed); process-wide, 0 = off) a one-wave spin kernel of a seeded random 5 .. 200 us is put
//   * EITHER at the head of a side branch (the branch starts late: a consumer that does not wait for its join reads old
//     data) OR on the caller's stream right behind the fork (the caller runs late: a branch that needs something the caller
//     launches AFTER the fork reads old data) -- one of the two per fork, by the draw,
//   * at the tail of every other side branch, before its join event is recorded,
//   * before one launch in four of the launch helpers, on whatever stream the launch goes to (host-side readers, copies
//     and collectives issued by the caller between two library calls meet a device that is still busy).
// The spin reads the constant 100 MHz counter and gives up after a bounded number of polls: it cannot hang a stream.
__global__ void k_race_spin(long long ticks) {
  const long long t0 = wall_clock64();
  for (int it = 0; it < (1 << 22) && wall_clock64() - t0 < ticks; ++it) __builtin_amdgcn_s_sleep(8);
}
struct RaceInject {
  std::atomic<uint64_t> state{0};
  std::atomic<int64_t> injected{0};
  std::atomic<int> max_us{200};
  std::atomic<int> drop_joins{0};      // CSP_TUNE_RACE_DROP_JOINS: the harness's own sensitivity test (results are WRONG by design)
  bool on() const { return state.load(std::memory_order_relaxed) != 0; }
  void seed(uint64_t s) { state.store(s ? (s * 0x9E3779B97F4A7C15ull) | 1ull : 0ull); }
  uint64_t next() {
    uint64_t x = state.load(std::memory_order_relaxed);
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    state.store(x | 1ull, std::memory_order_relaxed);
    return x;
  }
};
inline RaceInject& race_inject() {
  static RaceInject r;
  static const bool init = [] { const char* e = sw_str("SMCP_RACE"); if (e && atoll(e) > 0) r.seed((uint64_t)atoll(e)); return true; }();
  (void)init;
  return r;
}
// always = false: one call in four injects (the launch helpers)
inline void race_delay(hipStream_t s, bool always = true) {
  RaceInject& r = race_inject();
  if (!r.on()) return;
  const uint64_t x = r.next();
  if (!always && (x >> 40) % 4 != 0) return;
  const long long us = 5 + (long long)((x >> 16) % (uint64_t)std::max(1, r.max_us.load(std::memory_order_relaxed) - 4));
  hipLaunchKernelGGL(k_race_spin, dim3(1), dim3(64), 0, s, us * 100);
  r.injected.fetch_add(1, std::memory_order_relaxed);
}

inline unsigned sync_event_flags() {
  static int sys = -1;
  if (sys < 0) { const char* e = sw_str("SMCP_EVENT_SYSFENCE"); sys = (e && e[0] == '1') ? 1 : 0; }
  return sys ? hipEventDisableTiming : (hipEventDisableTiming | hipEventReleaseToDevice);
}
struct Fork {
  csp_ctx* c; hipStream_t main; hipStream_t s; int which; bool on;
  static bool enabled() {
    static int e = -1;
    if (e < 0) { const char* v = sw_str("SMCP_FORK"); e = (v && v[0] == '0') ? 0 : 1; }
    return e == 1 && !trace_on_early();
  }
  static bool trace_on_early() { static const bool t = [] { const char* e = sw_str("SMCP_TRACE"); return e && e[0] == '1'; }(); return t; }
  Fork(csp_ctx* c_, hipStream_t st, int which_) : c(c_), main(st), s(st), which(which_), on(false) {
    if (!enabled()) return;
    if (!c->aux_fork && hipEventCreateWithFlags(&c->aux_fork, sync_event_flags()) != hipSuccess) { c->aux_fork = nullptr; return; }
    if (!c->aux_stream[which]) {
      // side stream 0 carries the LONG POLE of every forked stage -- a handful of workgroups factoring or inverting the large
      // fronts beside launches of thousands of small-clique workgroups (k_mid_chol on the Y_AA blocks of the eight mid fronts
      // of synth50k: 55 us alone, 140 us beside k_factor_yaa_lds) -- so it is created with the highest priority the device
      // offers: its few workgroups get their CUs first.  SMCP_AUX_PRIO=0: default priority.
      static int prio = -2;
      if (prio == -2) { const char* e = sw_str("SMCP_AUX_PRIO"); prio = (e && e[0] == '0') ? 0 : 1; }
      int lo = 0, hi = 0;
      hipError_t rc = hipErrorUnknown;
      // side stream 1 carries FILLER work (thousands of small-clique workgroups beside a chain on the caller's stream): lowest
      // priority, so that it does not take the CUs a chain's few workgroups are waiting for (SMCP_AUX_PRIO1=0: default)
      static int prio1 = -2;
      if (prio1 == -2) { const char* e = sw_str("SMCP_AUX_PRIO1"); prio1 = (e && e[0] == '0') ? 0 : 1; }
      if (which == 0 && prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo)
        rc = hipStreamCreateWithPriority(&c->aux_stream[which], hipStreamNonBlocking, hi);
      if (which == 1 && prio1 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo)
        rc = hipStreamCreateWithPriority(&c->aux_stream[which], hipStreamNonBlocking, lo);
      if (rc != hipSuccess) rc = hipStreamCreateWithFlags(&c->aux_stream[which], hipStreamNonBlocking);
      if (rc != hipSuccess) { c->aux_stream[which] = nullptr; return; }
    }
    if (!c->aux_join[which] && hipEventCreateWithFlags(&c->aux_join[which], sync_event_flags()) != hipSuccess) { c->aux_join[which] = nullptr; return; }
    if (hipEventRecord(c->aux_fork, main) != hipSuccess) return;
    if (hipStreamWaitEvent(c->aux_stream[which], c->aux_fork, 0) != hipSuccess) return;
    s = c->aux_stream[which];
    on = true;
    // (delay injection: EITHER the branch starts late OR the caller's stream goes on late -- both at once would cancel)
    // (the harness's self-test, drop_joins: the branch is late by the longest delay, every time -- the removed edge MUST show)
    if (race_inject().on()) {
      if (race_inject().drop_joins.load(std::memory_order_relaxed)) {
        hipLaunchKernelGGL(k_race_spin, dim3(1), dim3(64), 0, s, (long long)race_inject().max_us.load() * 100);
      } else race_delay((race_inject().next() >> 33) & 1 ? s : main);
    }
  }
  void join() {
    if (!on) return;
    on = false;
    if (race_inject().on() && ((race_inject().next() >> 33) & 1)) race_delay(s);        // (delay injection: the branch ends late)
    if (race_inject().drop_joins.load(std::memory_order_relaxed)) { (void)hipEventRecord(c->aux_join[which], s); return; }   // (harness self-test: the edge is removed)
    if (hipEventRecord(c->aux_join[which], s) != hipSuccess || hipStreamWaitEvent(main, c->aux_join[which], 0) != hipSuccess)
      (void)hipStreamSynchronize(s);      // fall back to a host wait: the branch must be complete before the caller goes on
  }
  ~Fork() { join(); }
};

// SMCP_TRACE=1: name every launch on stderr and wait for it (a device fault aborts the process: the last name printed
// is the kernel that faulted)
inline bool trace_on() {
  static int t = -1;
  if (t < 0) { const char* e = sw_str("SMCP_TRACE"); t = (e && e[0] == '1') ? 1 : 0; }
  return t == 1;
}
inline void trace_launch(int kid, dim3 grid, dim3 block, size_t lds, hipStream_t st, bool before) {
  if (before) fprintf(stderr, "launch %s grid (%u,%u,%u) block %u lds %zu\n", KID_NAMES[kid], grid.x, grid.y, grid.z, block.x, lds);
  else (void)hipStreamSynchronize(st);
}
template <class K, class... A>
inline void launch_lds(csp_ctx* c, int kid, K kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... args) {
  if (trace_on()) trace_launch(kid, grid, block, lds, st, true);
  Profiler& P = c->prof;
  const bool timed = P.want(kid);
  if (timed) (void)hipEventRecord(P.next(), st);
  race_delay(st, false);
  hipLaunchKernelGGL(kern, grid, block, lds, st, args...);
  note_launch(c, kid);
  if (trace_on()) trace_launch(kid, grid, block, lds, st, false);
  if (timed) {
    (void)hipEventRecord(P.next(), st);
    P.kids.push_back(kid);
  }
}
template <class K, class... A>
inline void launch(csp_ctx* c, int kid, K kern, dim3 grid, dim3 block, hipStream_t st, A... args) {
  if (trace_on()) trace_launch(kid, grid, block, 0, st, true);
  Profiler& P = c->prof;
  const bool timed = P.want(kid);
  if (timed) (void)hipEventRecord(P.next(), st);
  race_delay(st, false);
  hipLaunchKernelGGL(kern, grid, block, 0, st, args...);
  note_launch(c, kid);
  if (trace_on()) trace_launch(kid, grid, block, 0, st, false);
  if (timed) {
    (void)hipEventRecord(P.next(), st);
    P.kids.push_back(kid);
  }
}

// host worker threads for the set-up phases: work(tix) for tix = 0 .. nth - 1.  Thread creation can fail
// (std::system_error); nothing may propagate across the extern "C" boundary, so whatever did not start runs inline.
template <class F>
void run_threads(int nth, F work) {
  if (nth <= 1) { work(0); return; }
  std::vector<std::thread> pool;
  int started = 0;
  try {
    for (; started < nth; ++started) pool.emplace_back(work, started);
  } catch (...) {
  }
  for (int tix = started; tix < nth; ++tix) work(tix);
  for (auto& th : pool) th.join();
}

template <class T>
int dev_upload(T** dst, const std::vector<T>& src, int64_t& bytes) {
  size_t n = std::max<size_t>(src.size(), 1) * sizeof(T);
  if (hipMalloc((void**)dst, n) != hipSuccess) return SMCP_ENOMEM;
  if (!src.empty() && hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
    return SMCP_EHIP;
  bytes += (int64_t)n;
  return 0;
}
template <class T>
int dev_alloc(T** dst, int64_t count, int64_t& bytes) {
  size_t n = (size_t)std::max<int64_t>(count, 1) * sizeof(T);
  // SMCP_CONTIG=1 (placement studies): large buffers from physically contiguous memory (hipDeviceMallocContiguous)
  static int contig = -1;
  if (contig < 0) { const char* e = sw_str("SMCP_CONTIG"); contig = (e && e[0] == '1') ? 1 : 0; }
  hipError_t arc = hipErrorUnknown;
  if (contig && n >= ((size_t)1 << 24)) arc = hipExtMallocWithFlags((void**)dst, n, hipDeviceMallocContiguous);
  if (arc != hipSuccess) { (void)hipGetLastError(); arc = hipMalloc((void**)dst, n); }
  if (arc != hipSuccess) return SMCP_ENOMEM;
  bytes += (int64_t)n;
  // SMCP_POISON=1 (hunting reads of never-written workspace): every fp64 buffer starts as 4.5e150 in every entry instead of
  // whatever the previous owner of the memory left there -- which, in a re-run of the same test, is the same data at the same
  // addresses and hides the read.  Index arrays are left alone (a poisoned index would fault, not mis-compute).
  if (std::is_same<T, double>::value) {
    static int poison = -1;
    if (poison < 0) poison = sw_on("SMCP_POISON", 0);
    if (poison && hipMemset((void*)*dst, 0x5F, n) != hipSuccess) return SMCP_EHIP;
  }
  { static int dbg = -1; if (dbg < 0) { const char* e = sw_str("SMCP_DEBUG_ADDR"); dbg = (e && e[0] == '1') ? 1 : 0; }      // placement studies
    if (dbg && n >= ((size_t)1 << 24)) fprintf(stderr, "smcp_amd: alloc %zu MB at %p\n", n >> 20, (void*)*dst); }
  return 0;
}

TreeArgs tree_args(csp_ctx* c) {
  TreeArgs a;
  a.cl = c->D.cl;
  a.relidx = c->D.relidx;
  a.chidx = c->D.chidx;
  a.lev = c->D.levidx;
  a.updlen = c->S.updlen();
  a.tmplen = c->D.tmplen;
  a.tmpptr = c->D.tmpptr;
  a.upd = c->D.upd;
  a.updp = c->D.updp;
  a.updplen = c->D.updp_stride ? c->D.updp_stride : c->S.updplen();
  a.tmp = c->D.tmp;
  a.info = c->D.info;
  a.nsn1 = (int)(c->S.nsn / c->ntrial);
  a.gp_tptr = c->D.gp_tptr;
  a.gp_tgt = c->D.gp_tgt;
  a.gp_cptr = c->D.gp_cptr;
  a.gp_src = c->D.gp_src;
  return a;
}

// SMCP_TIMING=1: wall-clock marks of the host-side set-up phases on stderr (csp_device_init, kkt_set_constraints)
struct SetupClock {
  bool on; const char* what; std::chrono::steady_clock::time_point t0;
  explicit SetupClock(const char* w) : what(w), t0(std::chrono::steady_clock::now()) { const char* e = sw_str("SMCP_TIMING"); on = e && e[0] == '1'; }
  void mark(const char* step) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[setup] %s: %-24s %7.1f ms\n", what, step, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

int ready(csp_ctx* c) {
  if (!c) return SMCP_EINVAL;
  if (c->D.device < 0) return SMCP_ENODEV;
  return 0;
}

// Deferred status (csp_lazy_status): instead of reading the failure flag back after every factorisation, a one-thread
// kernel moves it into a latch word (first failure wins) and clears it, exactly as the read-back does; csp_status
// reads the latch.  The host then synchronises once per KKT solve instead of four times (the gaps were ~0.15 ms of a
// 5.9 ms step on synth50k: the launch latency of whatever follows each read-back).
__global__ void k_latch_status(int* info, int ntrial, int* latch) {
  if (threadIdx.x == 0) {
    int v = 0;
    for (int t = 0; t < ntrial; ++t) if (info[t] && !v) v = info[t];
    if (v) {
      if (!*latch) *latch = v;
      for (int t = 0; t < ntrial; ++t) info[t] = 0;
    }
  }
}
// read back the device failure flag (synchronises the stream)
int fetch_info(csp_ctx* c, hipStream_t st) {
  if (c->launch_err) { c->launch_err = 0; return SMCP_EHIP; }
  if (c->lazy_status) {
    hipLaunchKernelGGL(k_latch_status, dim3(1), dim3(64), 0, st, c->D.info, (int)c->ntrial, c->D.info + 16);
    if (hipGetLastError() != hipSuccess) return SMCP_EHIP;
    c->flags_clean = true;
    return 0;
  }
  HIPCHK(hipMemcpyAsync(c->D.info_host, c->D.info, sizeof(int) * c->ntrial, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int64_t t = 0; t < c->ntrial; ++t)      // a replicated context: the first copy that failed (csp_trial_flags has them all)
    if (c->D.info_host[t]) {
      // the flag must not outlive the call that reports it: kernels of later calls that do not clear it themselves
      // (the factor of Y_AA inside kkt_qr_solve, say) return early on a set flag and would leave their output stale
      HIPCHK(hipMemsetAsync(c->D.info, 0, sizeof(int) * c->ntrial, st));
      return c->D.info_host[t];
    }
  return 0;
}

constexpr int NT = 256;

template <class F>
void for_levels_up(csp_ctx* c, F f) {
  for (int64_t l = 0; l < c->S.nlev; ++l) {
    int64_t b = c->S.levptr[l], e = c->S.levptr[l + 1];
    f(c->D.levidx + b, (int)(e - b));
  }
}
template <class F>
void for_levels_down(csp_ctx* c, F f) {
  for (int64_t l = c->S.nlev - 1; l >= 0; --l) {
    int64_t b = c->S.levptr[l], e = c->S.levptr[l + 1];
    f(c->D.levidx + b, (int)(e - b));
  }
}

// workgroups per clique of a gather launch whose largest separator is namax
// (a launch over a few cliques -- `pairs` (clique, right-hand side) pairs on a chip of ncu CUs -- is a chain of dependent
// index loads per thread: more, shorter workgroups then; k_gather_level on the eight mid fronts of synth50k, one
// right-hand side: 21 us with 2 parts)
inline unsigned gather_parts(int namax, int64_t pairs = (int64_t)1 << 40, int ncu = 0) {
  const int base = std::max(1, std::min(64, (namax * namax) / (NT * 32)));
  if (pairs * base >= 2 * (int64_t)ncu) return (unsigned)base;
  const int cap = std::max(1, std::min(64, (namax * namax) / (NT * 2)));
  return (unsigned)std::max<int64_t>(base, std::min<int64_t>(cap, (2 * (int64_t)ncu + pairs - 1) / std::max<int64_t>(1, pairs)));
}
inline int namax_of(csp_ctx* c, const int32_t* lev) {   // lev points into D.levidx at the start of a level
  const int64_t off = lev - c->D.levidx;
  const auto& lp = c->S.levptr;
  const int64_t l = std::upper_bound(lp.begin(), lp.end(), off) - lp.begin() - 1;
  return (l >= 0 && l < (int64_t)c->lev_namax.size()) ? c->lev_namax[l] : 0;
}

// upd[r][k] <- X_r[A_k, A_k] for all cliques, top-down
void gather_all(csp_ctx* c, const double* x, int64_t ldx, int nrhs, double* updbase, hipStream_t st) {
  TreeArgs a = tree_args(c);
  for_levels_down(c, [&](const int32_t* lev, int cnt) {
    a.lev = lev;
    launch(c, KID_gather_level, k_gather_level, dim3(cnt, nrhs, gather_parts(namax_of(c, lev), (int64_t)cnt * nrhs, c->D.ncu)), dim3(NT), st, a, x, ldx, updbase);
  });
}

int prepare_yaa(csp_ctx* c, const double* Y, bool need_fac, hipStream_t st, bool need_inv = false, bool allow_partial = false);
void complete_fac(csp_ctx* c, hipStream_t st);


// ---- fast path (front_mfma.hip): per level, LDS-class cliques then HBM-class cliques ----------
constexpr size_t LDS_LIMIT = 160 * 1024 - 2048;   // dynamic working set; the rest is left to small static buffers

bool cache_off() {
  static int nocache = -1;
  if (nocache < 0) { const char* e = sw_str("SMCP_NOCACHE"); nocache = (e && e[0] == '1') ? 1 : 0; }
  return nocache == 1;
}

// the any-size, fixed-order kernels of front_generic.hip for every tree operation: SMCP_GENERIC=1 (process-wide) or
// csp_tune(ctx, CSP_TUNE_DETERMINISTIC, 1) -- no floating-point atomics anywhere, results bit-identical from run to run
bool use_generic(const csp_ctx* c) {
  static int g = -1;
  if (g < 0) { const char* e = sw_str("SMCP_GENERIC"); g = (e && e[0] == '1') ? 1 : 0; }
  return g == 1 || (c && c->deterministic);
}

MfmaArgs mfma_args(csp_ctx* c, const double* ysc, int ymode, int nrhs) {
  MfmaArgs a;
  a.t = tree_args(c);
  a.t.lev = c->D.lev2idx;
  a.LK = c->D.lk;
  a.ysc = ysc;
  a.ymode = ymode;
  a.nnmax = a.namax = 0;
  a.nchmax = a.panmax = a.pkmax = a.plansum = 0;
  a.nrhs = nrhs;
  { static int sk = -1; if (sk < 0) { const char* e = sw_str("SMCP_SKIP"); sk = e ? atoi(e) : 0; } a.skip = sk; }
  a.dbg = (a.skip & 64) ? (unsigned long long*)(c->D.red + 768) : nullptr;
  a.lfd = c->D.lfd;
  a.dn = 0; a.dld = 0;
  a.kc_ptr = nullptr; a.kc_off = nullptr; a.kc_val = nullptr; a.kc_ids = nullptr;
  a.sp_rt = c->D.sp_rt; a.sp_mk = c->D.sp_mk;
  a.kc_stride = 0; a.kc_j0 = 0;
  a.nnmin = 0;
  a.grp_ptr = nullptr; a.grp_list = nullptr; a.chskip = nullptr; a.famt_ngrp = 0;
  a.fz_on = 0; a.fz_nat = 0; a.fz_cnn = 0; a.fz_stride = 0; a.fz_recl = 0;
  a.fz_tab = nullptr; a.fz_no = nullptr; a.fz_slot = nullptr; a.fz_ptr = nullptr; a.fz_pk = nullptr; a.fz_s = nullptr;
  a.level = 0; a.nS = 0; a.famna = a.fampan = a.fampk = a.famcna = a.famnn = a.famcnn = 0;
  return a;
}

int rhs_groups(int ncl, int nrhs, int target) {
  int g = (target + ncl - 1) / ncl;
  return std::max(1, std::min(g, nrhs));
}

// f(lds_mode, MfmaArgs-with-lev-set, count, lds_bytes, threads) for the two classes of level l
template <class F>
void for_level_classes(csp_ctx* c, int64_t l, MfmaArgs a, F f, int set = 0) {
  const LevelClass& L = set ? c->sets[set].lvl[l] : c->lvl[l];
  const int32_t* base = set ? c->sets[set].lev2 + c->sets[set].off[l] : c->D.lev2idx + c->S.levptr[l];
  if (L.nI) {
    a.t.lev = base;
    a.nnmax = L.nnmaxI;
    a.namax = L.namaxI;
    a.nchmax = L.nchmaxI;
    a.panmax = L.panmaxI;
    a.pkmax = L.pkmaxI;
    a.plansum = L.plansumI;
    a.level = (int)l; a.nS = (int)L.nS; a.famna = L.famna; a.fampan = L.fampan; a.fampk = L.fampk; a.famcna = L.famcna; a.famnn = L.famnn; a.famcnn = L.famcnn;
    size_t lds = (size_t)mfma_lds_doubles(L.nnmaxI, L.namaxI) * sizeof(double);
    f(true, a, (int)L.nI, lds, lds > 48 * 1024 ? 512 : 256);
  }
  if (L.nII) {
    a.t.lev = base + L.nI;
    a.level = (int)l;
    a.nnmax = L.nnmaxII;
    a.nnmin = L.nnminII;
    a.namax = L.namaxII;
    a.nchmax = L.nchmaxII;
    f(false, a, (int)L.nII, (size_t)0, 1024);
  }
}


// f(MfmaArgs over ALL large fronts, count): for clique-local operations (no level order needed)
template <class F>
void for_all_large(csp_ctx* c, MfmaArgs a, F f) {
  if (!c->D.nII_total) return;
  a.t.lev = c->D.lev3idx + c->D.nI_total;
  a.nnmax = c->D.nnmaxII_all;
  a.namax = c->D.namaxII_all;
  f(a, (int)c->D.nII_total);
}

bool use_large() {
  static int g = -1;
  if (g < 0) { const char* e = sw_str("SMCP_LARGE"); g = (e && e[0] == '0') ? 0 : 1; }
  return g == 1;
}
inline unsigned umax1(int x) { return (unsigned)std::max(1, x); }
// threads of the clique-local factorisation kernels on a class of small fronts (nn <= 16, na <= 32: the leaves of
// synth50k).  Measured (bench.py kernel times, SMCP_FTHR_* sweeps): k_factor_yaa_lds 0.149 -> 0.110 ms per step with 64 threads (a chain of
// one-wave 16 x 16 factorisations: more, smaller workgroups in flight), k_chol_mfma 0.179 -> 0.171 with 128,
// k_pinv_mfma is fastest with 256; the larger classes do not care.  SMCP_FTHR_CHOL / _PINV / _YAA override.
static int fact_threads(const MfmaArgs& am, int thr, int kind) {
  static int t[3] = {0, 0, 0};
  if (!t[0]) {
    // (k_factor_yaa_lds is compiled for at most 256 threads: a larger value was a launch failure, SMCP_EHIP, not a tuning)
    auto rd = [](const char* n, int d, int hi) { const char* e = sw_str(n); int v = e ? atoi(e) : d; return (v >= 64 && v <= hi && !(v & 63)) ? v : d; };
    t[0] = rd("SMCP_FTHR_CHOL", 128, 1024); t[1] = rd("SMCP_FTHR_PINV", 256, 1024); t[2] = rd("SMCP_FTHR_YAA", 64, 256);
  }
  if (kind == 2 && am.nnmax <= 16 && am.namax > 32 && am.namax <= 64) {      // Y_AA blocks of 33 .. 64 rows (the mid fronts of synth50k)
    static int tm = 0;
    if (!tm) { const char* e = sw_str("SMCP_FTHR_YAA_MID"); tm = e ? atoi(e) : 256; if (tm < 64 || tm > 256 || (tm & 63)) tm = 256; }
    return tm;
  }
  return (am.nnmax <= 16 && am.namax <= 32) ? t[kind] : thr;
}

// children -> front extend-add of the large fronts of one level: gather plan (one owner thread per front position;
// measured 1.27 ms on the synth50k top level) or, without a plan / with SMCP_ASM=tiled, the child-major tiled kernel
// (coalesced reads but a latency-bound scan of every child by every tile: 2.40 ms on the same level)
// clear_first (sgn 0 callers): the update blocks must read as zero where no child contributes -- the gather plan and the
// tiled kernel store only what they receive, so the blocks are cleared before them; k_lf_assemble_lds assigns the whole
// lower triangle (zeros included) and needs no clear pass (26 us per Schur sweep on synth50k: 105 MB of stores)
// does the streaming extend-add (k_lf_assemble_lds) take this launch, and with how many workgroups per (front, rhs) pair
static bool alds_route(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, int& nz, size_t& bytes) {
  // fronts whose packed lower triangle fits LDS: stream the children through it (k_lf_assemble_lds)
  static int alds = -1;
  if (alds < 0) { const char* e = sw_str("SMCP_ALDS"); alds = (e && e[0] == '0') ? 0 : 1; }
  const int nfmax = a.nnmax + a.namax;
  // workgroups per (front, right-hand side): one, unless a front has so many children that one CU would stream them
  // for long (config 3: 1999 children of the root, 100 pairs) -- then the children are dealt over nz workgroups whose
  // partial fronts meet in global memory (atomics; the update blocks are cleared first).  SMCP_ALDS_Z overrides.
  static int zenv = -1;
  if (zenv < 0) { const char* e = sw_str("SMCP_ALDS_Z"); zenv = e ? atoi(e) : 0; }
  const int64_t pairs = std::max<int64_t>(1, (int64_t)cnt * nrhs);
  nz = 1;
  if (zenv > 0) nz = zenv;
  else if (a.nchmax >= 256) nz = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)a.nchmax / 16, (int64_t)64, (4 * (int64_t)c->D.ncu + pairs - 1) / pairs}));
  // fewer (front, right-hand side) pairs than half the CUs, each with many children (one rank's share of an eight-rank job
  // on synth50k: ONE front x 100 constraints, 112 children each -- 150 us with a workgroup per pair): deal the children
  else if (nrhs >= 16 && 2 * pairs <= c->D.ncu && a.nchmax >= 32)
    nz = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)a.nchmax / 16, (int64_t)8, (2 * (int64_t)c->D.ncu + pairs - 1) / pairs}));
  nz = std::max(1, std::min(nz, std::max(1, a.nchmax)));
  bytes = (size_t)(lf_alds_doubles(nfmax) + 4 * ((a.nchmax + nz - 1) / nz) + 2) * sizeof(double);   // front + child table (+ the family column of k_lf_assemble_fz)
  if (!(alds && a.nchmax > 0 && pairs * nz >= 32 && nfmax <= LF_ALDS_MAXNF && bytes <= LDS_LIMIT)) return false;   // enough workgroups to fill the chip
  static bool attr = false;
  if (!attr) { attr = hipFuncSetAttribute((const void*)k_lf_assemble_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) == hipSuccess; }
  return attr;
}
// (sparse right-hand sides) would lf_assemble build the input panels itself -- sgn 3 of lf_alds_task -- so that the caller
// can skip k_panel_fill?  SMCP_ALDS_FILL=0: never.
// (shares: the caller is the fused extend-add, whose workgroups may share the children of a pair -- k_lf_assemble_fz clears
// and adds; the plain streaming kernels build panels only with one workgroup per pair)
static bool lf_assemble_fills(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, bool shares = false) {
  static int on = -1;
  if (on < 0) { const char* e = sw_str("SMCP_ALDS_FILL"); on = (e && e[0] == '0') ? 0 : 1; }
  int nz; size_t bytes;
  return on && a.kc_ptr && alds_route(c, a, cnt, nrhs, nz, bytes) && (nz == 1 || shares);
}
template <int NAT>
static bool launch_assemble_fz(csp_ctx* c, const MfmaArgs& az, int cnt, int nrhs, double* U, int64_t ldu, int sgn, size_t bytes, int* counter, hipStream_t st, int shares) {
  static bool attr = false;
  if (!attr) attr = hipFuncSetAttribute((const void*)k_lf_assemble_fz<NAT, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) == hipSuccess &&
                    hipFuncSetAttribute((const void*)k_lf_assemble_fz<NAT, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) == hipSuccess;
  if (!attr) return false;
  // sixteen waves (128 registers each) or eight (256): SMCP_FZ_THREADS=512 selects the latter (twelve waves -- 166 registers, no
  // spills either -- measured in round 5: 0.48 against 0.465 ms)
  static int thr = 0;
  if (!thr) { const char* e = sw_str("SMCP_FZ_THREADS"); thr = (e && atoi(e) == 512) ? 512 : 1024; }
  const dim3 grid((unsigned)std::min<int64_t>(c->D.ncu, (int64_t)cnt * nrhs * std::max(1, shares)));
  // a last round less than half full (synth50k: 100 pairs per queue of 32 workgroups: four in the fourth round) is dealt in
  // shares of the pairs' children (k_lf_assemble_fz, header); SMCP_FZ_TAIL=0: every pair whole
  int tail_first = -1, nzt = 1;
  static const bool tsplit = sw_on("SMCP_FZ_TAIL", true);
  if (tsplit && sgn == 3 && cnt % 8 == 0 && grid.x % 8 == 0 && grid.x >= 8) {
    const int wq = (int)grid.x / 8, total_q = (cnt / 8) * nrhs, rem = total_q % wq;
    if (total_q > wq && rem > 0 && 2 * rem <= wq) {
      nzt = std::min({8, wq / rem, std::max(1, az.nchmax / 8)});
      if (nzt >= 2) tail_first = total_q - rem;
      else nzt = 1;
    }
  }
  // fewer pairs than half the CUs (one rank's share of a sharded sweep: alds_route asked for `shares` workgroups per pair): every pair shared
  if (shares > 1) {
    if (sgn != 3) return false;
    tail_first = 0; nzt = shares;
  }
  if (tail_first >= 0)
    launch(c, KID_lf_clear_upd, k_lf_zero_pairs, dim3(16, (unsigned)(((cnt + 7) / 8) * nrhs - tail_first), 8), dim3(256), st, az, U, ldu, cnt, nrhs, tail_first);
  if (thr == 512) launch_lds(c, KID_lf_assemble_fz, k_lf_assemble_fz<NAT, 512>, grid, dim3(512), bytes, st, az, U, ldu, sgn, cnt, nrhs, counter, tail_first, nzt);
  else launch_lds(c, KID_lf_assemble_fz, k_lf_assemble_fz<NAT, 1024>, grid, dim3(1024), bytes, st, az, U, ldu, sgn, cnt, nrhs, counter, tail_first, nzt);
  return true;
}
void lf_assemble(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, double* U, int64_t ldu, int sgn, hipStream_t st, bool clear_first = false) {
  static int plan = -1;
  if (plan < 0) { const char* e = sw_str("SMCP_ASM"); plan = (e && e[0] == 't') ? 0 : 1; }
  if (c->fz_live && (size_t)a.level < c->fz_levels.size() && c->fz_levels[(size_t)a.level]) {
    // the family launch of this sweep left its parents' updates to this extend-add: the streaming kernel with the hook that
    // forms them (k_lf_assemble_fz) -- hess_up_fast has checked that this launch qualifies
    int nz; size_t bytes;
    bool ok = alds_route(c, a, cnt, nrhs, nz, bytes) && (nz == 1 || sgn == 3) && c->D.info;
    if (ok) {
      MfmaArgs az = a;
      az.fz_on = 1; az.fz_nat = c->fz_nat; az.fz_cnn = c->fz_cnn; az.fz_recl = c->fz_recl; az.fz_stride = (int)(c->D.m + 1);
      az.fz_tab = c->D.famc; az.fz_no = c->D.fz_no; az.fz_slot = c->D.fz_slot; az.fz_ptr = c->D.fz_ptr; az.fz_pk = c->D.fz_pk; az.fz_s = c->D.fz_s;
      int* counter = c->D.fz_slot + c->D.fz_nfam;       // eight task counters behind the slot table (kkt_set_constraints)
      (void)hipMemsetAsync(counter, 0, 8 * sizeof(int), st);
      switch (c->fz_nat) {
        case 1: ok = launch_assemble_fz<1>(c, az, cnt, nrhs, U, ldu, sgn, bytes, counter, st, nz); break;
        case 2: ok = launch_assemble_fz<2>(c, az, cnt, nrhs, U, ldu, sgn, bytes, counter, st, nz); break;
        case 3: ok = launch_assemble_fz<3>(c, az, cnt, nrhs, U, ldu, sgn, bytes, counter, st, nz); break;
        case 4: ok = launch_assemble_fz<4>(c, az, cnt, nrhs, U, ldu, sgn, bytes, counter, st, nz); break;
        default: ok = false;
      }
    }
    if (!ok) {
      fprintf(stderr, "smcp_amd: fused extend-add not launchable for a level it was promised to\n");
      if (!c->launch_err) c->launch_err = -1;
    }
    return;
  }
  {
    int nz; size_t bytes;
    if (alds_route(c, a, cnt, nrhs, nz, bytes)) {
      const int64_t pairs = std::max<int64_t>(1, (int64_t)cnt * nrhs);
      {
        if (nz > 1 && sgn == 0)
          launch(c, KID_lf_clear_upd, k_lf_clear_upd, dim3(umax1(std::min(64, (a.namax * a.namax + 2047) / 2048)), cnt, nrhs), dim3(256), st, a);
        // more tasks than CUs (one workgroup per CU: the front fills LDS): a persistent grid draws them from a counter, so
        // that the last round does not leave most of the chip idle.  SMCP_ALDS_DYN=0: one workgroup per task.
        static int dyn = -1;
        if (dyn < 0) { const char* e = sw_str("SMCP_ALDS_DYN"); dyn = (e && e[0] == '0') ? 0 : 1; }
        const int64_t tasks = pairs * nz;
        if (dyn && tasks > c->D.ncu && tasks < ((int64_t)1 << 30) && c->D.info) {
          static bool attr2 = false;
          if (!attr2) attr2 = hipFuncSetAttribute((const void*)k_lf_assemble_lds_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) == hipSuccess;
          if (attr2) {
            int* counter = c->D.info + 24 + (st == c->aux_stream[0] ? 1 : (st == c->aux_stream[1] ? 2 : 0));      // one counter per stream in use
            (void)hipMemsetAsync(counter, 0, sizeof(int), st);
            launch_lds(c, KID_lf_assemble_lds_dyn, k_lf_assemble_lds_dyn, dim3(c->D.ncu), dim3(1024), bytes, st, a, U, ldu, sgn, cnt, nrhs, nz, counter);
            return;
          }
        }
        launch_lds(c, KID_lf_assemble_lds, k_lf_assemble_lds, dim3(cnt, nrhs, nz), dim3(1024), bytes, st, a, U, ldu, sgn);
        return;
      }
    }
  }
  // (the gather plan and the tiled kernel read every child slot: members of sibling groups that did not write theirs in
  // this sweep -- MfmaArgs::chskip -- get them cleared first; the streaming kernel above skips them instead)
  if (a.chskip) launch(c, KID_lf_clear_upd, k_lf_zero_skipped, dim3(cnt, nrhs), dim3(256), st, a);
  // (the gather plan lists every position of a large front's update block -- csp_device_init -- and assigns them all)
  if (clear_first && !(plan && a.t.gp_tptr && c->plan_full_upd))
    launch(c, KID_lf_clear_upd, k_lf_clear_upd, dim3(umax1(std::min(64, (a.namax * a.namax + 2047) / 2048)), cnt, nrhs), dim3(256), st, a);
  if (plan && a.t.gp_tptr) {
    // a thread per front position when the launch is small (few fronts, one right-hand side): each position is a chain of
    // dependent loads, and 32 workgroups per front leave two or three positions per thread
    const int nfm = a.nnmax + a.namax;
    const int gx = (int64_t)cnt * nrhs * 32 < 2 * (int64_t)c->D.ncu ? std::max(32, std::min(128, (nfm * (nfm + 1) / 2 + 255) / 256)) : 32;
    if (a.nchmax <= 8) launch(c, KID_lf_assemble, k_lf_assemble<8>, dim3(gx, cnt, nrhs), dim3(256), st, a, U, ldu, sgn);
    else launch(c, KID_lf_assemble, k_lf_assemble<16>, dim3(gx, cnt, nrhs), dim3(256), st, a, U, ldu, sgn);
    return;
  }
  const int nfmax = a.nnmax + a.namax;
  const int ncb = (nfmax + LF_TW - 1) / LF_TW, nrb = (nfmax + LF_TR - 1) / LF_TR;
  launch(c, KID_lf_assemble, k_lf_assemble_tiled, dim3(ncb * nrb, cnt, nrhs), dim3(256), st, a, U, ldu, sgn);
}

// Phase kernels of the large fronts come in two shapes (gemm_tile64<PD>, front_large.hip): sixteen waves per 64 x 64 tile
// for launches of at most one workgroup per CU (a few tiles: one right-hand side on the top fronts), four waves otherwise
// (the batched sweeps, where the workgroups sharing a CU keep its four SIMDs busy).  SMCP_PD=1: four waves everywhere.
static bool pd_deep() {
  static int on = -1;
  if (on < 0) { const char* e = sw_str("SMCP_PD"); on = (e && e[0] == '1') ? 0 : 1; }
  return on == 1;
}
// (SMCP_PD_WGS: the largest launch, in workgroups per CU, that still takes the sixteen-wave shape; default 1)
static int64_t pd_wgs() {
  static int64_t v = -1;
  if (v < 0) { const char* e = sw_str("SMCP_PD_WGS"); v = e ? std::max(1, atoi(e)) : 1; }
  return v;
}
#define LAUNCH_PD(c, kid, kern, grid, blk, ...)                                                        \
  do {                                                                                                 \
    const dim3 g_ = (grid);                                                                            \
    if (pd_deep() && (int64_t)g_.x * g_.y * g_.z <= pd_wgs() * (int64_t)(c)->D.ncu) launch(c, kid, kern<4>, g_, dim3(1024), __VA_ARGS__); \
    else launch(c, kid, kern<1>, g_, blk, __VA_ARGS__);                                                \
  } while (0)

// (two products with the dense Y_NN cost 3 nn^3 flops against 2 nn^3 of the four phases that exploit the triangular Li:
// the fusion is for roots whose sweeps are launch-bound, not for config 2's 4096 clique)
constexpr int ROOT_FUSED_MAXNN = 512;
// OFF unless SMCP_ROOT_FUSED=1.  The explicit Y_NN = Li^T Li carries the SQUARE of the factor's condition number into every
// entry it touches, the nested congruences Li^T (Li F Li^T) Li do not, and solve_ must apply the operator the Schur
// complement was built from (H = <G(A_i), G(A_j)>) to rounding x cond(Li), not x cond(Li)^2: over 288 interior-point runs
// of scratch/fuzz_ipm.py (48 random problems x 6 driver / solver combinations) the fused route ended three runs with
// status "unknown" in the last, ill-conditioned iterations where the nested route (and round 2) end optimal; it saves
// 27 us per Hessian on synth50k (solve_ 0.79 -> 0.74 ms) and is kept for studies only.
static bool root_fused() {
  static int on = -1;
  if (on < 0) { const char* e = sw_str("SMCP_ROOT_FUSED"); on = (e && e[0] == '1') ? 1 : 0; }
  return on == 1;
}
// fill: the input panels have not been built (sparse right-hand sides, lf_assemble_fills): the extend-add builds them
void lf_up(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, double* U, int64_t ldu, hipStream_t st, bool fill = false) {
  const int mtA = tiles64(a.namax), ntN = tiles64(a.nnmax);
  dim3 blk(256);
  if (a.nchmax > 0) {     // a level of childless large fronts has nothing to assemble (k_lf_up2 does not read their update blocks)
    lf_assemble(c, a, cnt, nrhs, U, ldu, fill ? 3 : 0, st, true);
    if (c->mid_work && a.level >= 2) {      // the levels 0 and 1 are complete and the first extend-add above them is queued: work of the
      std::function<void(hipStream_t)> w = std::move(c->mid_work);     // caller that only needs their panels, beside the phase kernels
      c->mid_work = nullptr;
      w(st);
    }
    if (c->side_work) {   // independent work of the caller: beside the phase kernels from here on (csp_ctx::side_work)
      std::function<void(hipStream_t)> w = std::move(c->side_work);
      c->side_work = nullptr;
      Fork* f = new Fork(c, st, 1);           // (normal priority: the phase kernels on the caller's stream are the long pole here)
      c->side_fork = f;
      w(f->s);
    }
  }
  LAUNCH_PD(c, KID_lf_up1, k_lf_up1, dim3(umax1(mtA * ntN + ntN * ntN), cnt, nrhs), blk, st, a, U, ldu);
  LAUNCH_PD(c, KID_lf_up2, k_lf_up2, dim3(umax1(mtA * (mtA + 1) / 2 + mtA * ntN + ntN * (ntN + 1) / 2), cnt, nrhs), blk, st, a, U, ldu);
  if (a.namax) LAUNCH_PD(c, KID_lf_up3, k_lf_up3, dim3(umax1(mtA * ntN), cnt, nrhs), blk, st, a, U, ldu);
}
void lf_down(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, double* U, int64_t ldu, hipStream_t st) {
  const int mtA = tiles64(a.namax), ntN = tiles64(a.nnmax);
  dim3 blk(256);
  if (a.namax) launch(c, KID_gather_level, k_gather_level, dim3(cnt, nrhs, gather_parts(a.namax, (int64_t)cnt * nrhs, c->D.ncu)), dim3(NT), st, a.t, (const double*)U, ldu, a.t.upd);
  LAUNCH_PD(c, KID_lf_down1, k_lf_down1, dim3(umax1(mtA * ntN + ntN * ntN), cnt, nrhs), blk, st, a, U, ldu);
  if (a.namax) LAUNCH_PD(c, KID_lf_down2, k_lf_down2, dim3(umax1(mtA * ntN), cnt, nrhs), blk, st, a, U, ldu);
  LAUNCH_PD(c, KID_lf_down3, k_lf_down3, dim3(umax1(ntN * (ntN + 1) / 2), cnt, nrhs), blk, st, a, U, ldu);
}
void lf_pinv(csp_ctx* c, const MfmaArgs& a, int cnt, double* x, hipStream_t st) {
  const int mtA = tiles64(a.namax), ntN = tiles64(a.nnmax);
  dim3 blk(256);
  if (a.namax) {
    launch(c, KID_gather_level, k_gather_level, dim3(cnt, 1, gather_parts(a.namax, cnt, c->D.ncu)), dim3(NT), st, a.t, (const double*)x, (int64_t)0, a.t.upd);
    LAUNCH_PD(c, KID_lf_pinv1, k_lf_pinv1, dim3(umax1(mtA * ntN), cnt, 1), blk, st, a, x);
  }
  LAUNCH_PD(c, KID_lf_pinv2, k_lf_pinv2, dim3(umax1(ntN * (ntN + 1) / 2 + mtA * ntN), cnt, 1), blk, st, a, x);
}

constexpr size_t LF_DIAG_LDS = (size_t)(2 * LB * LBD + 256 + 16 * LB) * sizeof(double);
// threads of the one-workgroup diagonal-block step (SMCP_DIAG_THREADS, timing studies; multiple of 64, at most 1024)
static dim3 diag_blk() {
  static int t = 0;
  if (!t) { const char* e = sw_str("SMCP_DIAG_THREADS"); t = e ? atoi(e) : 512; if (t < 64 || t > 1024 || (t & 63)) t = 512; }
  return dim3(t);
}

// blocked Cholesky of the large fronts of one level (children already factored): clear + assemble + steps
// fronts of at most MID_MAXROWS rows: the blocked Cholesky of a front in one workgroup (k_mid_chol); SMCP_MID=0: per-step kernels
bool use_mid(int rowsmax) {
  static int g = -1;
  if (g < 0) {
    const char* e = sw_str("SMCP_MID");
    g = (e && e[0] == '0') ? 0 : 1;
    if (g && hipFuncSetAttribute((const void*)k_mid_chol, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_chol_lds(MID_MAXROWS)) != hipSuccess) g = 0;
  }
  return g == 1 && rowsmax <= MID_MAXROWS;
}
// ---- one-launch blocked Cholesky with in-launch tile dataflow (front_flow.hip) ---------------------------------------------
// Factors ONE matrix of order n (129 .. 4096): either the plain dense matrix A (leading dimension ld; fa == nullptr: info /
// info_val as given) or the matrix of the single front of the launch list of *fa (mode 0: a front without separator, base =
// the blkval array; mode 2: its Y_AA block, base = fac).  dinv: where the inverses of the diagonal blocks go (64 x 64 slots;
// nullptr: the workspace's own).  Returns false when the route does not apply or its set-up fails (the caller takes the
// per-step kernels); SMCP_FLOW=0: never.
bool flow_chol(csp_ctx* c, hipStream_t st, double* A, int64_t ld, int n, double* dinv, const MfmaArgs* fa, int mode, int* info, int info_val, int cnt = 1) {
  static int on = -1, wgs = 0;
  if (on < 0) { on = sw_on("SMCP_FLOW", 1); wgs = std::max(1, std::min(FLOW_MAXWG, sw_int("SMCP_FLOW_WG", FLOW_MAXWG))); }
  if (!on || use_generic(c) || n <= 2 * LB || n > FLOW_MAXN || !info) return false;
  // several fronts of a level in one launch (cnt > 1: their cliques through fa->t.lev, n = the largest order): the fronts share
  // the budget of workgroups of ONE launch (wgs: what may stand beside other launches on the chip), at least eight each
  if (cnt < 1 || (cnt > 1 && (!fa || dinv || cnt > wgs / 8))) return false;
  static int attr = -1;
  if (attr < 0) attr = hipFuncSetAttribute((const void*)k_chol_flow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_LDS_BYTES) == hipSuccess ? 1 : 0;
  if (!attr) return false;
  const int nt = (n + 63) / 64, ntiles = nt * (nt + 1) / 2;
  csp_ctx::FlowWs& W = c->flow_ws[st == c->aux_stream[0] && st ? 1 : (st == c->aux_stream[1] && st ? 2 : 0)];
  if (W.cap_nt < nt || W.cap_fronts < cnt) {
    // (the buffers may be in use by a launch still queued on this stream: hipFree waits for the device)
    const int ntc = std::max(nt, W.cap_nt), ntilesc = ntc * (ntc + 1) / 2, frc = std::max(cnt, W.cap_fronts);
    for (void* q : {(void*)W.P, (void*)W.dinv, (void*)W.flags}) if (q) (void)hipFree(q);
    W = csp_ctx::FlowWs();
    int64_t junk = 0;
    if (dev_alloc(&W.P, (int64_t)frc * ntilesc * 4096, junk) || dev_alloc(&W.dinv, (int64_t)frc * ntc * 4096, junk) ||
        hipMalloc((void**)&W.flags, sizeof(unsigned) * (size_t)frc * (size_t)(ntilesc + ntc + 4)) != hipSuccess ||
        hipMemset(W.flags, 0, sizeof(unsigned) * (size_t)frc * (size_t)(ntilesc + ntc + 4)) != hipSuccess) {
      (void)hipGetLastError();
      for (void* q : {(void*)W.P, (void*)W.dinv, (void*)W.flags}) if (q) (void)hipFree(q);
      W = csp_ctx::FlowWs();
      return false;
    }
    c->D.bytes += junk + (int64_t)sizeof(unsigned) * frc * (ntilesc + ntc + 4);
    W.cap_nt = ntc; W.cap_fronts = frc;
  }
  const int wgs_front = cnt > 1 ? wgs / cnt : (n > 2048 ? (wgs * 10) / 7 : wgs);
  if ((int64_t)ntiles > (int64_t)FLOW_MAXOWN * wgs_front) return false;      // (a workgroup tracks at most FLOW_MAXOWN tiles)
  const int plan_key = n * 1024 + std::min(1023, wgs_front);
  auto it = c->flow_plans.find(plan_key);
  if (it == c->flow_plans.end()) {
    std::vector<int32_t> optr, otile;
    // (beyond order 2048 -- 528 tiles -- more workgroups: the trailing updates, not the chain of diagonal tiles, set the time there;
    // 160: three such launches of three processes sharing the GPU still fit the chip side by side, two workgroups per CU)
    flow_make_plan(n, wgs_front, optr, otile);
    csp_ctx::FlowPlanDev P;
    P.nwg = (int)optr.size() - 1;
    int64_t junk = 0;
    if (dev_upload(&P.own_ptr, optr, junk) || dev_upload(&P.own_tile, otile, junk)) { (void)hipGetLastError(); return false; }
    c->D.bytes += junk;
    it = c->flow_plans.emplace(plan_key, P).first;
  }
  FlowArgs f;
  f.A = A; f.ld = ld; f.n = n;
  f.dinv = dinv ? dinv : W.dinv;
  f.P = W.P; f.flags = W.flags;
  // flags are never cleared: a set flag of this launch carries its epoch.  The words of a SMALLER matrix's launch lie elsewhere
  // in the array (the offsets depend on nt), but any word ever written holds an older epoch, which never equals a newer one
  f.epoch = ++W.epoch;
  f.own_ptr = it->second.own_ptr; f.own_tile = it->second.own_tile;
  f.info = info; f.info_val = info_val;
  f.cl = fa ? fa->t.cl : nullptr; f.lev = fa ? fa->t.lev : nullptr; f.mode = mode; f.nsn1 = fa ? fa->t.nsn1 : 1;
  f.dbg = nullptr;
  // (per-front slices of the workspace: the layout of the largest order n this launch was given)
  f.p_stride = (int64_t)ntiles * 4096; f.dinv_stride = (int64_t)nt * 4096; f.flag_stride = ntiles + nt + 4;
  static int stamps = -1;
  if (stamps < 0) stamps = sw_on("SMCP_FLOW_STAMPS", 0);
  static long long* dbg = nullptr;
  if (stamps) { if (!dbg) (void)hipMalloc((void**)&dbg, sizeof(long long) * 16 * FLOW_MAXWG); f.dbg = dbg; }
  launch_lds(c, KID_chol_flow, k_chol_flow, dim3(it->second.nwg, cnt), dim3(256), FLOW_LDS_BYTES, st, f);
  if (stamps && dbg) {           // timing studies: what the workgroups spent their time on, on stderr
    std::vector<long long> h((size_t)8 * it->second.nwg);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    for (int w = 0; w < std::min(it->second.nwg, 20); ++w)
      fprintf(stderr, "flow n %d wg %3d: idle %7.1f us (%lld)  update %7.1f us (%lld)  potrf %7.1f us (%lld)  panel %7.1f us (%lld)\n", n, w,
              h[8 * w] / 100.0, h[8 * w + 4], h[8 * w + 1] / 100.0, h[8 * w + 5], h[8 * w + 2] / 100.0, h[8 * w + 6], h[8 * w + 3] / 100.0, h[8 * w + 7]);
  }
  return true;
}

void lf_chol(csp_ctx* c, const MfmaArgs& a, int cnt, double* x, hipStream_t st) {
  dim3 blk(256);
  const int nfmax = a.nnmax + a.namax;
  lf_assemble(c, a, cnt, 1, x, 0, 0, st, true);      // (clears the update blocks first where its route does not assign them whole)
  if (use_mid(nfmax)) {
    launch_lds(c, KID_mid_chol, k_mid_chol, dim3(cnt), dim3(1024), mid_chol_lds(nfmax), st, a, x, (double*)nullptr, 0);
    return;
  }
  // ONE front without separator (a root): the whole blocked factorisation in one launch
  if (cnt == 1 && a.namax == 0 && flow_chol(c, st, x, 0, a.nnmax, nullptr, &a, 0, a.t.info, 0)) return;
  const int mtA = tiles64(a.namax);
  for (int jb = 0; jb < a.nnmax; jb += LB) {
    launch_lds(c, KID_lf_diag, k_lf_diag, dim3(cnt), diag_blk(), LF_DIAG_LDS, st, a, x, (double*)nullptr, 0, jb, 1);
    const int mrem = nfmax - jb - 1, ncr = std::max(0, a.nnmax - jb - 1);
    if (mrem > 0) launch(c, KID_lf_chol_panel, k_lf_chol_panel, dim3(umax1(tiles64(mrem)), cnt), blk, st, a, x, (double*)nullptr, 0, jb);
    const int mt = tiles64(mrem), nt = tiles64(ncr);
    const int ntask = nt * (nt + 1) / 2 + std::max(0, mt - nt) * nt + mtA * (mtA + 1) / 2;
    if (ntask > 0) launch(c, KID_lf_chol_trail, k_lf_chol_trail, dim3(umax1(ntask), cnt), blk, st, a, x, (double*)nullptr, 0, jb);
  }
  if (a.namax) launch(c, KID_lf_pack_upd, k_lf_pack_upd, dim3(umax1(std::min(64, (a.namax * a.namax + 255) / 256)), cnt), blk, st, a);
}
// chol(Y_AA) of the large fronts of one level, in place in fac (already a copy of yaa)
void lf_factor_yaa(csp_ctx* c, const MfmaArgs& a, int cnt, double* fac, hipStream_t st) {
  dim3 blk(256);
  if (use_mid(a.namax)) {
    launch_lds(c, KID_mid_chol, k_mid_chol, dim3(cnt), dim3(1024), mid_chol_lds(a.namax), st, a, (double*)nullptr, fac, 2);
    return;
  }
  // (one front, or the fronts of the level side by side in one launch -- config 4: levels of three and five fronts with
  // separators of 300 - 500 rows were 8 + 5 steps of three launches each)
  if (flow_chol(c, st, fac, 0, a.namax, nullptr, &a, 2, a.t.info, 0, cnt)) return;
  for (int jb = 0; jb < a.namax; jb += LB) {
    launch_lds(c, KID_lf_diag, k_lf_diag, dim3(cnt), diag_blk(), LF_DIAG_LDS, st, a, (double*)nullptr, fac, 2, jb, 1);
    const int mrem = a.namax - jb - 1;
    if (mrem > 0) {
      launch(c, KID_lf_chol_panel, k_lf_chol_panel, dim3(umax1(tiles64(mrem)), cnt), blk, st, a, (double*)nullptr, fac, 2, jb);
      const int mt = tiles64(mrem);
      launch(c, KID_lf_chol_trail, k_lf_chol_trail, dim3(umax1(mt * (mt + 1) / 2), cnt), blk, st, a, (double*)nullptr, fac, 2, jb);
    }
  }
}
// inverse-form factor of the large fronts of one level
void lf_prep(csp_ctx* c, const MfmaArgs& a, int cnt, const double* L, hipStream_t st) {
  dim3 blk(256);
  static int hoist = -1;
  if (hoist < 0) { const char* e = sw_str("SMCP_MID"); hoist = (e && e[0] == '0') ? 0 : 1; }
  if (hoist) {
    // the diagonal blocks' inverses do not depend on each other: one launch for all of them, then the block rows
    launch_lds(c, KID_lf_diag_inv, k_lf_diag_inv, dim3(cnt, tiles64(a.nnmax)), blk, LF_DIAG_LDS, st, a, L, c->D.lk);
    static int rec = -1;
    if (rec < 0) { const char* e = sw_str("SMCP_TRTRI"); rec = (e && e[0] == '0') ? 0 : 1; }
    if (rec) {
      // recursive doubling: log2(nn / 64) levels of two tile-product launches each (k_lf_trtri)
      for (int b = LB; b < a.nnmax; b *= 2) {
        const int pairs = (a.nnmax + 2 * b - 1) / (2 * b), tpb = b / LB;
        for (int step = 0; step < 2; ++step)
          launch(c, KID_lf_prep_s, k_lf_trtri, dim3(umax1(pairs * tpb * tpb), cnt), blk, st, a, L, c->D.lk, b, step);
      }
    } else
    for (int ib = LB; ib < a.nnmax; ib += LB) {
      launch(c, KID_lf_prep_s, k_lf_prep_s, dim3(umax1(tiles64(ib)), cnt), blk, st, a, L, c->D.lk, ib, 0);
      launch(c, KID_lf_prep_row, k_lf_prep_row, dim3(umax1(tiles64(ib)), cnt), blk, st, a, L, c->D.lk, ib, 0, 1);
    }
  } else
  for (int ib = 0; ib < a.nnmax; ib += LB) {
    launch_lds(c, KID_lf_diag, k_lf_diag, dim3(cnt), diag_blk(), LF_DIAG_LDS, st, a, const_cast<double*>(L), (double*)nullptr, 1, ib, 0);
    if (ib > 0) launch(c, KID_lf_prep_s, k_lf_prep_s, dim3(umax1(tiles64(ib)), cnt), blk, st, a, L, c->D.lk, ib, 0);
    launch(c, KID_lf_prep_row, k_lf_prep_row, dim3(umax1(tiles64(ib) + 1), cnt), blk, st, a, L, c->D.lk, ib, 0, 0);
  }
  if (a.namax) launch(c, KID_lf_prep_k, k_lf_prep_k, dim3(umax1(tiles64(a.namax) * tiles64(a.nnmax)), cnt), blk, st, a, L, c->D.lk);
}

// dynamic LDS of k_prep_lk: the row-block scratch of supernodes wider than 16 columns (none otherwise)
static size_t prep_lk_lds_bytes(int nnmax) { return nnmax > 16 ? (size_t)16 * 16 * 16 * sizeof(double) : 0; }
static int prep_lk_threads(int nnmax) { return nnmax > 16 ? NT : 128; }
// ---- cache verification (csp_tune(ctx, CSP_TUNE_VERIFY_CACHE, 1); debug aid) ------------------------------------
// The derived-quantity caches are keyed by the ADDRESS of the matrix they came from; a caller that rewrites the
// matrix in place without telling the library (csp_touch) would be served stale factors.  With verification on, a
// fingerprint of the matrix -- the wrap-around sum of the bit patterns of ALL its panel entries, each multiplied by an
// odd number that depends on its position, so exact and independent of the summation order -- is latched on the device
// when a cache entry is created and compared when the entry is reused; a mismatch makes the call return SMCP_ESTALE.
__global__ void k_diag_fingerprint(const CliqueDesc* cl, int nsn, const double* x, unsigned long long* slot, int* bad, int check) {
  // every entry of every panel (one pass over blkval: 20 MB on synth50k), weighted by an odd number that depends on its position
  __shared__ unsigned long long part[256];
  unsigned long long s = 0;
  for (int k = blockIdx.x; k < nsn; k += gridDim.x) {
    const CliqueDesc d = cl[k];
    const int64_t len = (int64_t)(d.nn + d.na) * d.nn;
    for (int64_t e = threadIdx.x; e < len; e += blockDim.x)
      s += (unsigned long long)__double_as_longlong(x[d.blk + e]) * (2ull * (unsigned long long)(d.blk + e) + 1ull);
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(slot + (check ? 4 : 0), part[0]);
  (void)bad;
}
__global__ void k_fingerprint_compare(unsigned long long* slots, int which, int* bad) {
  if (threadIdx.x == 0) { if (slots[which] != slots[which + 4]) *bad = 1 + which; slots[which + 4] = 0; }
}
// which: 0 = LK derived from L at this address, 1 = LK valid for the pair (L, Y) with Y at this address, 2 = Y_AA blocks of Y
static void fp_record(csp_ctx* c, int which, const double* x, hipStream_t st) {
  if (!c->verify_cache || !c->D.fp) return;
  (void)hipMemsetAsync(c->D.fp + which, 0, sizeof(unsigned long long), st);
  hipLaunchKernelGGL(k_diag_fingerprint, dim3(1024), dim3(256), 0, st, c->D.cl, (int)c->S.nsn, x, c->D.fp + which, c->D.fp_bad, 0);
}
static int fp_check(csp_ctx* c, int which, const double* x, hipStream_t st) {
  if (!c->verify_cache || !c->D.fp) return 0;
  hipLaunchKernelGGL(k_diag_fingerprint, dim3(1024), dim3(256), 0, st, c->D.cl, (int)c->S.nsn, x, c->D.fp + which, c->D.fp_bad, 1);
  hipLaunchKernelGGL(k_fingerprint_compare, dim3(1), dim3(64), 0, st, c->D.fp, which, c->D.fp_bad);
  int bad = 0;
  if (hipMemcpyAsync(&bad, c->D.fp_bad, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return SMCP_EHIP;
  if (bad) {
    (void)hipMemsetAsync(c->D.fp_bad, 0, sizeof(int), st);
    fprintf(stderr, "smcp_amd: stale cache: the matrix at %p changed since the quantities cached for it (kind %d) were derived; "
                    "call csp_touch after writing to a matrix outside the library\n", (const void*)x, which);
    return SMCP_ESTALE;
  }
  return 0;
}

void prep_lk(csp_ctx* c, const double* L, hipStream_t st) {
  TreeArgs t = tree_args(c);
  if (use_large()) {
    t.lev = c->D.lev3idx;
    MfmaArgs a0 = mfma_args(c, nullptr, 0, 1);
    {
      Fork f(c, st, 0);          // the blocked inversions of the large fronts run beside the small cliques' launch
      for_all_large(c, a0, [&](MfmaArgs am, int cnt) { lf_prep(c, am, cnt, L, f.s); });
      if (c->D.nI_total) {
        int nnI = 0;
        for (const LevelClass& Lc : c->lvl) if (Lc.nI) nnI = std::max(nnI, (int)Lc.nnmaxI);
        launch_lds(c, KID_prep_lk, k_prep_lk, dim3((int)c->D.nI_total), dim3(prep_lk_threads(nnI)), prep_lk_lds_bytes(nnI), st, t, L, c->D.lk);
      }
    }
  } else {
    t.lev = nullptr;
    launch_lds(c, KID_prep_lk, k_prep_lk, dim3((int)c->S.nsn), dim3(NT), prep_lk_lds_bytes(1 << 20), st, t, L, c->D.lk);
  }
  c->D.lk_tag_L = L;
  c->D.lk_tag_Y = nullptr;
  c->D.part_valid = false;
  c->D.lk_gen++;
  fp_record(c, 0, L, st);
}
// the same for the cliques of one set of the partition (clique-local: no order among the levels needed)
void prep_lk_set(csp_ctx* c, int set, const double* L, hipStream_t st) {
  TreeArgs t = tree_args(c);
  MfmaArgs a0 = mfma_args(c, nullptr, 0, 1);
  for (int64_t l = 0; l < c->S.nlev; ++l)
    for_level_classes(c, l, a0, [&](bool lds, MfmaArgs am, int cnt, size_t, int) {
      if (lds) { t.lev = am.t.lev; launch_lds(c, KID_prep_lk, k_prep_lk, dim3(cnt), dim3(prep_lk_threads(am.nnmax)), prep_lk_lds_bytes(am.nnmax), st, t, L, c->D.lk); }
      else lf_prep(c, am, cnt, L, st);
    }, set);
  c->D.lk_gen++;
}
// The KKT entry points are called with (L, Y) where either LK was just prepared from this very L,
// or Y = projected_inverse(L) was produced by csp_projected_inverse (which prepares LK from L
// before overwriting it).  In both cases the cached LK is still the inverse form of L.
int prep_lk_cached(csp_ctx* c, const double* L, const double* Y, hipStream_t st) {
  const bool nocache = cache_off();
  if (!nocache && c->D.lk_tag_L && c->D.lk_tag_L == L) return fp_check(c, 0, L, st);
  if (!nocache && c->D.lk_tag_Y && c->D.lk_tag_Y == Y) return fp_check(c, 1, Y, st);
  prep_lk(c, L, st);
  return 0;
}
// an in-place operation is about to change the matrix stored at p: forget what was derived from it
void invalidate_tags(csp_ctx* c, const void* p) {
  if (c->D.lk_tag_L == p) c->D.lk_tag_L = nullptr;
  if (c->D.lk_tag_Y == p) c->D.lk_tag_Y = nullptr;
  if (c->D.yaa_tag == p) c->D.yaa_tag = c->D.fac_tag = c->D.faci_tag = nullptr;
}


// shape-specialised sweep kernel (front_n16.hip) for a class with nn <= 16, na <= 64; false if it does not apply
template <int NAT, bool CH>
bool launch_n16(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, double* U, int64_t ldu, hipStream_t st) {
  const size_t bytes = n16_lds_bytes<NAT, CH>(a.nchmax, a.panmax, a.pkmax, a.plansum);
  if (bytes > LDS_LIMIT) return false;
  int thr = bytes > 48 * 1024 ? 512 : 256;
  if (!CH && NAT <= 2 && nrhs <= 2) {
    // one or two right-hand sides on childless small fronts (the Hessians of solve_): the launch is rounds x per-workgroup
    // set-up latency, and two waves carry the three tile tasks of a phase as well as four: 128 threads put twice as many
    // fronts on a CU (k_hess_up_n16 0.297 -> 0.274 ms per step; 64 threads: 0.286; SMCP_N16_THR_LEAF overrides)
    static int tl = 0;
    if (!tl) { const char* e = sw_str("SMCP_N16_THR_LEAF"); tl = e ? atoi(e) : 128; if (tl < 64 || tl > 512 || (tl & 63)) tl = 128; }
    thr = tl;
  }
  // Split the right-hand sides over g workgroups per clique so that the grid fills a whole number of rounds of
  // the resident-workgroup slots: cost(g) = rounds x (passes per workgroup + set-up, counted as 4 passes).
  static size_t nb_bytes = 0;    // occupancy of this instantiation, cached per LDS size
  static int nb = 1;
  if (bytes != nb_bytes) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_hess_up_n16<NAT, CH>, thr, bytes) != hipSuccess || nb < 1) nb = 1;
    nb_bytes = bytes;
  }
  const int ncu = c->D.ncu;
  const int64_t slots = (int64_t)ncu * nb;
  const int rb = CH ? 1 : std::max(1, std::min(4, 16 / std::max(a.nnmax, 1)));
  int g = 1;
  int64_t best = -1;
  for (int gc = 1; gc <= std::min(nrhs, 32); ++gc) {
    const int64_t rounds = ((int64_t)cnt * gc + slots - 1) / slots;
    const int64_t passes = (nrhs + gc * rb - 1) / (gc * rb);
    const int64_t cost = rounds * (passes + 4);
    if (best < 0 || cost < best) { best = cost; g = gc; }
  }
  static int dbg = -1;
  if (dbg < 0) { const char* e = sw_str("SMCP_OCC"); dbg = (e && e[0] == '1') ? 1 : 0; }
  if (dbg) fprintf(stderr, "n16<%d,%d>: cnt %d nrhs %d g %d threads %d lds %zu -> %d workgroups/CU\n", NAT, (int)CH, cnt, nrhs, g, thr, bytes, nb);
  launch_lds(c, KID_hess_up_n16, k_hess_up_n16<NAT, CH>, dim3(cnt, g), dim3(thr), bytes, st, a, U, ldu);
  return true;
}
bool try_n16(csp_ctx* c, const MfmaArgs& a, int cnt, int g, double* U, int64_t ldu, hipStream_t st) {
  static int off = -1;
  if (off < 0) { const char* e = sw_str("SMCP_N16"); off = (e && e[0] == '0') ? 1 : 0; }
  if (off || a.nnmax > 16 || a.namax > 64) return false;
  const int nat = std::max(1, (a.namax + 15) / 16);
  const bool ch = a.nchmax > 0;
  switch (nat * 2 + (ch ? 1 : 0)) {
    case 2: return launch_n16<1, false>(c, a, cnt, g, U, ldu, st);
    case 3: return launch_n16<1, true>(c, a, cnt, g, U, ldu, st);
    case 4: return launch_n16<2, false>(c, a, cnt, g, U, ldu, st);
    case 5: return launch_n16<2, true>(c, a, cnt, g, U, ldu, st);
    case 6: return launch_n16<3, false>(c, a, cnt, g, U, ldu, st);
    case 7: return launch_n16<3, true>(c, a, cnt, g, U, ldu, st);
    case 8: return launch_n16<4, false>(c, a, cnt, g, U, ldu, st);
    case 9: return launch_n16<4, true>(c, a, cnt, g, U, ldu, st);
  }
  return false;
}

// family kernel (front_fam.hip) for the nS family parents at the tail of a level's LDS class
template <int NAT, int NATC, bool SP>
bool launch_fam(csp_ctx* c, MfmaArgs a, int cnt, int nrhs, double* U, int64_t ldu, hipStream_t st) {
  a.panmax = a.fampan;
  a.pkmax = a.fampk;
  const size_t bytes = fam_lds_bytes<NAT, NATC>(a.fampan, a.fampk);
  if (bytes > LDS_LIMIT) return false;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k_hess_up_fam<NAT, NATC, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return false;
    attr = true;
  }
  const int ncu = c->D.ncu;
  // one workgroup per CU (LDS): split the right-hand sides so that the grid fills whole rounds; set-up ~ 3 passes
  int g = 1;
  int64_t best = -1;
  for (int gc = 1; gc <= std::min(nrhs, 32); ++gc) {
    const int64_t rounds = ((int64_t)cnt * gc + ncu - 1) / ncu;
    const int64_t passes = (nrhs + gc - 1) / gc;
    const int64_t cost = rounds * (passes + 3);
    if (best < 0 || cost < best) { best = cost; g = gc; }
  }
  launch_lds(c, KID_hess_up_fam, k_hess_up_fam<NAT, NATC, SP>, dim3(cnt, g), dim3(768), bytes, st, a, U, ldu);
  return true;
}
template <bool SP>
bool try_fam_sp(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, double* U, int64_t ldu, hipStream_t st) {
  const int nat = std::max(1, (a.famna + 15) / 16), natc = std::max(1, (a.famcna + 15) / 16);
  switch (nat * 2 + natc - 1) {
    case 2: return launch_fam<1, 1, SP>(c, a, cnt, nrhs, U, ldu, st);
    case 3: return launch_fam<1, 2, SP>(c, a, cnt, nrhs, U, ldu, st);
    case 4: return launch_fam<2, 1, SP>(c, a, cnt, nrhs, U, ldu, st);
    case 5: return launch_fam<2, 2, SP>(c, a, cnt, nrhs, U, ldu, st);
    case 6: return launch_fam<3, 1, SP>(c, a, cnt, nrhs, U, ldu, st);
    case 7: return launch_fam<3, 2, SP>(c, a, cnt, nrhs, U, ldu, st);
    case 8: return launch_fam<4, 1, SP>(c, a, cnt, nrhs, U, ldu, st);
    case 9: return launch_fam<4, 2, SP>(c, a, cnt, nrhs, U, ldu, st);
  }
  return false;
}
// sparse-input family kernel (front_fam2.hip): R^T scaling (the Gram sweeps), entry lists short on average, the
// children's constants within the LDS budget; false = not applicable (the caller falls back to k_hess_up_fam)
template <int NAT, int KSN>
bool launch_fam2(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, double* U, int64_t ldu, hipStream_t st) {
  DeviceCtx& D = c->D;
  const int cnn = std::max(1, a.famcnn), csa = 16 * std::max(1, (a.famcna + 15) / 16);
  const int64_t lim = (160 * 1024 - 1024) / 8;                          // doubles of LDS a workgroup may use
  const int64_t fixed = fam2_layout<NAT>(cnn, csa).oTab;
  if (fixed + fam2_tail_doubles(4, 9) + 64 > lim) return false;          // room for a table of four passes at least
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k_fam_sparse<NAT, KSN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return false;
    if (hipFuncSetAttribute((const void*)k_fam_sparse<NAT, KSN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return false;
    if (hipFuncSetAttribute((const void*)k_fam2_prep, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return false;
    attr = true;
  }
  const int64_t need = (int64_t)cnt * fam2_const_doubles(cnn, csa);
  if (D.famc_len < need) {
    if (D.famc) { if (hipFree(D.famc) != hipSuccess) return false; D.bytes -= D.famc_len * 8; }
    D.famc = nullptr; D.famc_len = 0;
    if (dev_alloc(&D.famc, need, D.bytes)) return false;
    D.famc_len = need;
  }
  const int ncu = D.ncu;
  // one workgroup per CU (LDS), two right-hand sides in flight per workgroup: split the right-hand sides so that the
  // grid fills whole rounds; set-up ~ 4 passes
  int g = 1;
  int64_t best = -1;
  for (int gc = 1; gc <= std::min(nrhs, 32); ++gc) {
    const int64_t rounds = ((int64_t)cnt * gc + ncu - 1) / ncu;
    const int64_t passes = (nrhs + gc - 1) / gc;
    const int64_t cost = rounds * ((passes + 1) / 2 + 4);
    if (best < 0 || cost < best) { best = cost; g = gc; }
  }
  // entry table: all the passes of a workgroup if the LDS left over allows (one double per (pass, member) pair, 1.5 per
  // staged entry, sized for twice the average list length), otherwise epochs of fewer passes
  // entry table: all the passes of a workgroup if the LDS left over allows (one double per (pass, member) pair, two per
  // staged entry, sized for 1.5 x the average list length), otherwise epochs of fewer passes
  const int passes = (nrhs + g - 1) / g;
  const double avg = 9.0 * (double)D.cnnz / ((double)c->S.nsn * (double)std::max<int64_t>(1, D.m));   // entries per (family, rhs)
  int tabpasses = std::min(passes, 113);
  while (tabpasses > 4 && fixed + fam2_tail_doubles(tabpasses, 9) + (int64_t)(1.5 * avg * tabpasses) + 8 > lim) tabpasses = (tabpasses + 1) / 2;
  const int64_t left = lim - fixed - fam2_tail_doubles(tabpasses, 9) - 4;
  const int ecap = (int)std::max<int64_t>(0, (left * 2) / 3 - 2);
  if (9 * D.kc_maxlist > ecap) return false;      // the entries of one pass must fit the staging area (the kernel sizes its epochs itself)
  launch_lds(c, KID_fam2_prep, k_fam2_prep, dim3(cnt), dim3(512), (size_t)8 * fam2_child_layout(cnn, csa).cstride * sizeof(double), st,
             a, D.famc, cnn, csa);
  MfmaArgs a2 = a;
  { static int stag = -1; if (stag < 0) { const char* e = sw_str("SMCP_FAM2_STAG"); stag = e ? atoi(e) : 0; } a2.dn = stag; }
  // the children's panels are left out when the caller takes their Gram block from k_leaf_gram (D.lg_request)
  if (D.lg_request) {
    launch_lds(c, KID_fam_sparse, k_fam_sparse<NAT, KSN, false>, dim3(cnt, g), dim3(512), (size_t)lim * 8, st, a2, U, ldu,
               (const double*)D.famc, cnn, csa, (const int32_t*)D.kc_ij, tabpasses, ecap);
    D.lg_nochild = true;
  } else {
    launch_lds(c, KID_fam_sparse, k_fam_sparse<NAT, KSN, true>, dim3(cnt, g), dim3(512), (size_t)lim * 8, st, a2, U, ldu,
               (const double*)D.famc, cnn, csa, (const int32_t*)D.kc_ij, tabpasses, ecap);
  }
  return true;
}
// Entry-driven family sweep (front_famt.hip) for sweeps that leave the children's panels to k_leaf_gram: false = not
// applicable (the caller falls back to k_fam_sparse).  SMCP_FAMT=0 disables.
inline bool famt_disabled() {
  static int off = -1;
  if (off < 0) { const char* e = sw_str("SMCP_FAMT"); off = (e && e[0] == '0') ? 1 : 0; }
  return off != 0;
}
// The entry-driven sweeps (k_fam_terms, the fused extend-add) cost in proportion to the entries of the (family, constraint)
// lists -- 0.87 ms per Schur sweep on synth50k at 13 entries per list -- the dense family sweep (k_hess_up_fam) 2.84 ms
// whatever the lists hold: break-even near 42 entries per list ON AVERAGE.  Rounds 3 - 4 gated on the LONGEST list (48: one
// chunk of the descriptor area), so ONE long list -- a multiple of the identity among sparse constraints: 15 + 8 x 5 diagonal
// entries per family -- sent the whole set to the dense route (1.86 x the step).  Now the kernels take long lists in chunks
// (FAMT_TMAX) and the gate is the mean.  SMCP_FAMT_MEAN moves it.
inline bool famt_terms_ok(const DeviceCtx& D) {
  static const int gate = sw_int("SMCP_FAMT_MEAN", 36);
  return D.fam_maxterms <= FAMT_TMAX && D.fam_meanterms <= (double)gate;
}
// LDS of the grouped kernel: entries it can stage (< 0: the fixed part does not fit); one pass of a group needs at most
// FAMT_GMAX x FAMT_TCAP / 2 of them
template <int NAT>
int famt_grp_ecap(int cnn) {
  const int64_t lim = (160 * 1024 - 1024) / 8;
  const int64_t fixed = famt_layout<NAT>(8 * cnn).total + 8 * famt_desc_doubles() + famt_grp_misc_doubles();
  const int64_t left = lim - fixed - 4;
  if (left <= 0 || famt_prep_doubles<NAT>(cnn) > lim) return -1;
  return (int)std::max<int64_t>(0, (left * 2) / 3 - 2);
}
inline bool famt_grp_fits(int nat, int cnn) {
  int ecap = -1;
  switch (nat) {
    case 1: ecap = famt_grp_ecap<1>(cnn); break;
    case 2: ecap = famt_grp_ecap<2>(cnn); break;
    case 3: ecap = famt_grp_ecap<3>(cnn); break;
    case 4: ecap = famt_grp_ecap<4>(cnn); break;
  }
  return ecap >= FAMT_GMAX * (FAMT_TCAP / 2);
}
template <int NAT>
bool launch_famt(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, double* U, int64_t ldu, hipStream_t st) {
  DeviceCtx& D = c->D;
  if (a.famt_ngrp > 0) {
    // sibling groups: the levels above have been told that the non-leaders' slots stay unwritten -- no other kernel may
    // take this launch, so everything that could refuse it fails the call instead (end_call returns SMCP_EHIP)
    const int cnn = std::max(1, a.famcnn);
    const FamtL L = famt_layout<NAT>(8 * cnn);
    const int ecap = famt_grp_ecap<NAT>(cnn);
    static bool attrg = false;
    bool ok = !famt_disabled() && D.lg_request && D.fam_maxterms <= FAMT_TCAP / 2 && ecap >= FAMT_GMAX * (FAMT_TCAP / 2);
    if (ok && !attrg) {
      ok = hipFuncSetAttribute((const void*)k_fam_terms_grp<NAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) == hipSuccess &&
           hipFuncSetAttribute((const void*)k_famt_prep<NAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) == hipSuccess;
      attrg = ok;
    }
    const int64_t need = (int64_t)cnt * (FAMT_HDR + L.total);
    if (ok && D.famc_len < need) {
      if (D.famc) { if (hipFree(D.famc) != hipSuccess) ok = false; D.bytes -= D.famc_len * 8; }
      D.famc = nullptr; D.famc_len = 0;
      if (ok && dev_alloc(&D.famc, need, D.bytes)) ok = false;
      if (ok) D.famc_len = need;
    }
    if (!ok) {
      fprintf(stderr, "smcp_amd: grouped family sweep not launchable\n");
      if (!c->launch_err) c->launch_err = -1;
      return true;
    }
    const int gy = std::min(65535, (nrhs + 7) / 8);
    launch_lds(c, KID_famt_prep, k_famt_prep<NAT>, dim3(cnt), dim3(512), (size_t)famt_prep_doubles<NAT>(cnn) * sizeof(double), st, a, D.famc, cnn, (int32_t*)nullptr);
    launch_lds(c, KID_fam_terms_grp, k_fam_terms_grp<NAT>, dim3(a.famt_ngrp, gy), dim3(512), (size_t)(160 * 1024 - 1024), st, a, U, ldu,
               (const double*)D.famc, cnn, (const int32_t*)D.kc_ij, ecap);
    D.lg_nochild = true;
    return true;
  }
  if (famt_disabled() || !D.lg_request) return false;
  const int cnn = std::max(1, a.famcnn);
  const FamtL L = famt_layout<NAT>(8 * cnn);
  const int64_t lim = (160 * 1024 - 1024) / 8;                          // doubles of LDS a workgroup may use
  const int64_t fixed = L.total + FAMT_NW * famt_desc_doubles() + 6;
  if (!famt_terms_ok(D)) return false;                                   // (long lists: in chunks of the descriptor area)
  if (fixed + 9 * 4 + 64 > lim) return false;
  const int64_t prep_doubles = famt_prep_doubles<NAT>(cnn);
  if (prep_doubles > lim) return false;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k_fam_terms<NAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return false;
    if (hipFuncSetAttribute((const void*)k_famt_prep<NAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return false;
    attr = true;
  }
  const int64_t need = (int64_t)cnt * (FAMT_HDR + L.total);
  if (D.famc_len < need) {
    if (D.famc) { if (hipFree(D.famc) != hipSuccess) return false; D.bytes -= D.famc_len * 8; }
    D.famc = nullptr; D.famc_len = 0;
    if (dev_alloc(&D.famc, need, D.bytes)) return false;
    D.famc_len = need;
  }
  const int ncu = D.ncu;
  // one workgroup per CU (LDS), eight right-hand sides in flight per workgroup: split the right-hand sides so that the
  // grid fills whole rounds; the set-up (tables, entry lists) costs about as much as four passes (half a round of eight: since
  // the tables reach LDS with eight loads in flight per thread; sixteen before.  synth50k: two slices, seven full rounds)
  int g = 1;
  int64_t best = -1;
  for (int gc = 1; gc <= std::min(nrhs, 32); ++gc) {
    const int64_t rounds = ((int64_t)cnt * gc + ncu - 1) / ncu;
    const int64_t passes = (nrhs + gc - 1) / gc;
    const int64_t cost = rounds * (2 * ((passes + FAMT_NW - 1) / FAMT_NW) + 1);
    if (best < 0 || cost < best) { best = cost; g = gc; }
  }
  static int genv = -1;
  if (genv < 0) { const char* e = sw_str("SMCP_FAMT_G"); genv = e ? atoi(e) : 0; }
  if (genv > 0) g = std::min(genv, nrhs);
  const int passes = (nrhs + g - 1) / g;
  const double avg = 9.0 * (double)D.cnnz / ((double)c->S.nsn * (double)std::max<int64_t>(1, D.m));   // entries per (family, rhs)
  int tabpasses = std::min(passes, 113);
  auto tail = [&](int tp) { return (int64_t)2 * ((tp * 9 + 1) & ~1) / 2 + 2; };      // doubles of the (pass, member) table
  while (tabpasses > 4 && fixed + tail(tabpasses) + (int64_t)(1.5 * 1.5 * avg * tabpasses) + 8 > lim) tabpasses = (tabpasses + 1) / 2;
  const int64_t left = lim - fixed - tail(tabpasses) - 4;
  const int ecap = (int)std::max<int64_t>(0, (left * 2) / 3 - 2);
  if (9 * D.kc_maxlist > ecap) return false;
  if (a.fz_on) {
    // requested by hess_up_fast (every front above takes the LDS extend-add): the parents' updates are left to that extend-add
    // (k_lf_assemble_fz), which finds the tables through the context (fz_live: set here, read by lf_assemble)
    static bool attrz = false;
    if (!attrz) attrz = hipFuncSetAttribute((const void*)k_fam_terms<NAT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) == hipSuccess;
    if (attrz && D.fz_slot) {
      launch_lds(c, KID_famt_prep, k_famt_prep<NAT>, dim3(cnt), dim3(512), (size_t)prep_doubles * sizeof(double), st, a, D.famc, cnn, D.fz_slot);
      launch_lds(c, KID_fam_terms, k_fam_terms<NAT, false>, dim3(cnt, g), dim3(64 * FAMT_NW), (size_t)lim * 8, st, a, U, ldu,
                 (const double*)D.famc, cnn, (const int32_t*)D.kc_ij, tabpasses, ecap);
      D.lg_nochild = true;
      c->fz_live = true; c->fz_nat = NAT; c->fz_cnn = cnn; c->fz_recl = (int64_t)(FAMT_HDR + L.total);
      return true;
    }
  }
  launch_lds(c, KID_famt_prep, k_famt_prep<NAT>, dim3(cnt), dim3(512), (size_t)prep_doubles * sizeof(double), st, a, D.famc, cnn, (int32_t*)nullptr);
  launch_lds(c, KID_fam_terms, k_fam_terms<NAT>, dim3(cnt, g), dim3(64 * FAMT_NW), (size_t)lim * 8, st, a, U, ldu,
             (const double*)D.famc, cnn, (const int32_t*)D.kc_ij, tabpasses, ecap);
  D.lg_nochild = true;
  return true;
}
inline bool fam2_disabled() {
  static int off = -1;
  if (off < 0) { const char* e = sw_str("SMCP_FAM2"); off = (e && e[0] == '0') ? 1 : 0; }
  return off != 0;
}
bool try_fam2(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, double* U, int64_t ldu, hipStream_t st) {
  if (fam2_disabled() || !a.kc_ptr || !c->D.kc_ij || a.ymode != 2 || !a.ysc) return false;
  // short lists only: the children's sweep costs O(entries) here, the dense MFMA sweep of k_hess_up_fam does not
  if (c->D.cnnz > (int64_t)4 * c->S.nsn * std::max<int64_t>(1, c->D.m)) return false;
  const int nat = std::max(1, (a.famna + 15) / 16);
  const bool k2 = a.famnn <= 8;
  switch (nat) {      // sweeps whose children's Gram block comes from k_leaf_gram: the entry-driven kernel
    case 1: if (launch_famt<1>(c, a, cnt, nrhs, U, ldu, st)) return true; break;
    case 2: if (launch_famt<2>(c, a, cnt, nrhs, U, ldu, st)) return true; break;
    case 3: if (launch_famt<3>(c, a, cnt, nrhs, U, ldu, st)) return true; break;
    case 4: if (launch_famt<4>(c, a, cnt, nrhs, U, ldu, st)) return true; break;
  }
  switch (nat) {
    case 1: return k2 ? launch_fam2<1, 2>(c, a, cnt, nrhs, U, ldu, st) : launch_fam2<1, 4>(c, a, cnt, nrhs, U, ldu, st);
    case 2: return k2 ? launch_fam2<2, 2>(c, a, cnt, nrhs, U, ldu, st) : launch_fam2<2, 4>(c, a, cnt, nrhs, U, ldu, st);
    case 3: return k2 ? launch_fam2<3, 2>(c, a, cnt, nrhs, U, ldu, st) : launch_fam2<3, 4>(c, a, cnt, nrhs, U, ldu, st);
    case 4: return k2 ? launch_fam2<4, 2>(c, a, cnt, nrhs, U, ldu, st) : launch_fam2<4, 4>(c, a, cnt, nrhs, U, ldu, st);
  }
  return false;
}
bool try_fam(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, double* U, int64_t ldu, hipStream_t st) {
  if (try_fam2(c, a, cnt, nrhs, U, ldu, st)) return true;
  if (a.famt_ngrp > 0) {
    // sibling groups were promised to the levels above (MfmaArgs::chskip: the non-leaders' slots are skipped there): only the
    // grouped kernel writes the summed updates, so a per-parent fallback would silently drop them -- fail the call instead
    fprintf(stderr, "smcp_amd: grouped family sweep refused by try_fam2 (conditions of hess_up_fast and try_fam2 disagree)\n");
    if (!c->launch_err) c->launch_err = -1;
    return true;
  }
  if (a.ysc == c->D.fac) complete_fac(c, st);      // k_hess_up_fam scales the children's panels by their own factors
  return a.kc_ptr ? try_fam_sp<true>(c, a, cnt, nrhs, U, ldu, st) : try_fam_sp<false>(c, a, cnt, nrhs, U, ldu, st);
}

// childless fronts beyond the LDS class with nn <= 64, na <= 128, sparse input, R^T scaling: swept from their entry lists
// (front_lfsp.hip).  R^T and R^T K are formed when fac or lk have changed since they were formed last.  SMCP_LFSP=0 disables.
bool try_lfsp(csp_ctx* c, const MfmaArgs& a, int cnt, int nrhs, double* U, int64_t ldu, hipStream_t st, const csp_ctx::LfspGroups* grp = nullptr) {
  static int on = -1;
  if (on < 0) { const char* e = sw_str("SMCP_LFSP"); on = (e && e[0] == '0') ? 0 : 1; }
  DeviceCtx& D = c->D;
  if (!on || !a.kc_ptr || a.nchmax != 0 || a.ymode != 2 || !a.ysc || a.nnmax > 64 || a.namax > 128) return false;
  if (D.kc_maxlist_large <= 0 || D.kc_maxlist_large > LFSP_ECAP || !D.lfsp_cnt) return false;
  const bool exact = D.lfsp_exact;      // every such front is exactly (64, 128): no bounds checks in the kernel
  if (!D.sp_rt) {
    if (dev_alloc(&D.sp_rt, std::max<int64_t>(c->S.updlen(), 1), D.bytes)) return false;
    if (dev_alloc(&D.sp_mk, std::max<int64_t>(c->S.blklen(), 1), D.bytes)) return false;
  }
  MfmaArgs b = a;
  b.sp_rt = D.sp_rt; b.sp_mk = D.sp_mk;
  if (D.sp_fac_gen != D.fac_gen || D.sp_lk_gen != D.lk_gen) {      // for ALL such fronts of the tree at once
    MfmaArgs p = b;
    p.t.lev = D.lfsp_list;
    launch(c, KID_lfsp_prep, k_lfsp_prep, dim3((unsigned)D.lfsp_cnt), dim3(256), st, p, a.ysc, D.sp_rt, D.sp_mk);
    D.sp_fac_gen = D.fac_gen; D.sp_lk_gen = D.lk_gen;
  }
  static int gdiv = -1;
  if (gdiv < 0) { const char* e = sw_str("SMCP_LFSP_G"); gdiv = e ? std::max(1, atoi(e)) : 1; }
  const int g = std::max(1, (nrhs + gdiv - 1) / gdiv);            // right-hand sides per workgroup: gdiv
  // one launch per phase (Q, Upd, G_NN): fewer live accumulators per wave, three to four waves per SIMD
  const dim3 grid(g, cnt), blk(256);
  // the update phase by sibling groups (one summed update per group; the caller hands MfmaArgs::chskip to the levels above)
  const bool grouped = grp && grp->ngroups > 0;
  MfmaArgs bg = b;
  if (grouped) { bg.grp_ptr = grp->ptr; bg.grp_list = grp->list; }
  const dim3 ggrid(g, grouped ? grp->ngroups : cnt);
  // (SMCP_LFSP_OCC=1, study: the grouped update phase with ONE wave per SIMD -- 512 registers -- instead of two: at two it keeps
  // sixteen accumulator tiles in 256 registers and spills ~100 of them to scratch)
  static const int gocc = sw_int("SMCP_LFSP_OCC", 2);
  if (grouped && gocc == 1) {
    if (exact) {
      launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 1, 3, 1, false, true>, grid, blk, st, b, U, ldu);
      launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 2, 1, 1, false, true, true>, ggrid, blk, st, bg, U, ldu);
      launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 4, 4, 1, false, true>, grid, blk, st, b, U, ldu);
    } else {
      launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 1, 3, 1, false, false>, grid, blk, st, b, U, ldu);
      launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 2, 1, 1, false, false, true>, ggrid, blk, st, bg, U, ldu);
      launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 4, 4, 1, false, false>, grid, blk, st, b, U, ldu);
    }
    return true;
  }
  if (exact) {
    launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 1, 3, 1, false, true>, grid, blk, st, b, U, ldu);
    if (grouped) launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 2, 2, 1, false, true, true>, ggrid, blk, st, bg, U, ldu);
    else launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 2, 2, 1, false, true>, grid, blk, st, b, U, ldu);
    launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 4, 4, 1, false, true>, grid, blk, st, b, U, ldu);
  } else {
    launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 1, 3, 1, false, false>, grid, blk, st, b, U, ldu);
    if (grouped) launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 2, 2, 1, false, false, true>, ggrid, blk, st, bg, U, ldu);
    else launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 2, 2, 1, false, false>, grid, blk, st, b, U, ldu);
    launch(c, KID_lfsp_up, k_lfsp_up<8, 4, 4, 4, 1, false, false>, grid, blk, st, b, U, ldu);
  }
  return true;
}
// would try_lfsp take a class whose static shape qualifies (the conditions that do not depend on the class)
bool lfsp_dynamic_ok(csp_ctx* c, const MfmaArgs& a) {
  static int on = -1;
  if (on < 0) { const char* e = sw_str("SMCP_LFSP"); on = (e && e[0] == '0') ? 0 : 1; }
  const DeviceCtx& D = c->D;
  return on && a.kc_ptr && a.ymode == 2 && a.ysc && D.kc_maxlist_large > 0 && D.kc_maxlist_large <= LFSP_ECAP && D.lfsp_cnt;
}

// sparse_j0 >= 0: the right-hand sides are the constraints sparse_j0 .. (through `ids` if given) and are taken from
// their per-clique entry lists (MfmaArgs::kc_*) -- U is output only and need not be cleared or scattered into
// yroot: the projected inverse Y (blkval) when the caller runs the down sweep right after this one (hessian(adj = None)):
// fronts without separator then take Z_NN = Y_NN F_NN Y_NN in two products (k_lf_root1 here, k_lf_root2 in hess_down_fast)
void hess_up_fast(csp_ctx* c, double* U, int nrhs, int64_t ldu, const double* ysc, int ymode, hipStream_t st, int set = 0,
                  int64_t sparse_j0 = -1, const int32_t* ids = nullptr, int64_t lev_lo = 0, int64_t lev_hi = -1,
                  const double* yroot = nullptr) {
  MfmaArgs a0 = mfma_args(c, ysc, ymode, nrhs);
  const bool sparse = sparse_j0 >= 0 && c->D.kc_ptr;
  if (sparse) {
    a0.kc_ptr = c->D.kc_ptr; a0.kc_off = c->D.kc_off; a0.kc_val = c->D.kc_val; a0.kc_ids = ids;
    a0.kc_stride = (int)(c->D.m + 1); a0.kc_j0 = (int)sparse_j0;
  }
  // sibling groups of the sparse-input sweep of childless large fronts: when they are in use the non-leaders' slots of the
  // exchange buffer stay unwritten, and every extend-add above must know (MfmaArgs::chskip)
  const bool groups_on = sparse && set == 0 && c->lfsp_any_groups && c->D.lfsp_skip && lfsp_dynamic_ok(c, a0);
  // the same for the sibling groups of family parents (k_fam_terms_grp): on exactly when try_fam2 / launch_famt will take the
  // family launches of this sweep (their conditions, repeated here: the decision must hold for every level of the sweep)
  const bool fgroups_on = sparse && set == 0 && c->famt_any_groups && c->D.famt_skip && !famt_disabled() && !fam2_disabled() && c->D.lg_request &&
                          c->D.kc_ij && a0.ymode == 2 && a0.ysc && c->D.fam_maxterms <= FAMT_TCAP / 2 &&
                          c->D.cnnz <= (int64_t)4 * c->S.nsn * std::max<int64_t>(1, c->D.m);
  if (groups_on || fgroups_on) a0.chskip = groups_on && fgroups_on ? c->D.both_skip : (groups_on ? c->D.lfsp_skip : c->D.famt_skip);
  // Fused extend-add of the family parents' updates (front_famt.hip, lf_add_family): requested when the whole tree (or, sharded,
  // the whole of this rank's subtrees: set 1, which holds the families together with the fronts above them) is swept in
  // one call with sparse input, the entry-driven family kernel is the route (its conditions, as for the groups above), there
  // is ONE level of family parents, and every large level with children takes the LDS extend-add that also builds the
  // panels (lf_assemble_fills); the family launch then decides (csp_ctx::fz_live).  SMCP_FZ=0: never.
  c->fz_live = false;
  {
    static int fzenv = -1;
    if (fzenv < 0) { const char* e = sw_str("SMCP_FZ"); fzenv = (e && e[0] == '0') ? 0 : 1; }
    bool want = fzenv && sparse && (set == 0 || (set == 1 && c->fz_set1_ok)) && lev_lo == 0 && lev_hi < 0 && !fgroups_on && c->D.fz_ok && !famt_disabled() && !fam2_disabled() &&
                c->D.lg_request && c->D.kc_ij && a0.ymode == 2 && a0.ysc && famt_terms_ok(c->D) &&
                c->D.cnnz <= (int64_t)4 * c->S.nsn * std::max<int64_t>(1, c->D.m) && !use_generic(c) && use_large() && c->D.gp_tptr;
    int famlevels = 0, nofill = 0;
    if (want) {
      for (int64_t l = 0; l < c->S.nlev; ++l)
        for_level_classes(c, l, a0, [&](bool lds, MfmaArgs a, int cnt, size_t, int) {
          if (lds && a.nS > 0 && a.level > 0) ++famlevels;
          if (!lds && (size_t)a.level < c->fz_levels.size() && c->fz_levels[(size_t)a.level] && !(a.nchmax > 0 && lf_assemble_fills(c, a, cnt, nrhs, true))) { want = false; ++nofill; }
        }, set);
      if (famlevels != 1) want = false;
    }
    if (want) { a0.fz_on = 1; a0.fz_no = c->D.fz_no; }
    static int occ = -1;
    if (occ < 0) occ = sw_on("SMCP_OCC", 0);
    if (occ && sparse)
      fprintf(stderr, "fz: want %d famlevels %d nofill %d nlev %d (env %d set %d lev %d..%d fgroups %d fz_ok %d famt_off %d fam2_off %d lg_request %d kc_ij %d ymode %d ysc %d maxterms %d cnnz %lld nsn %lld m %lld generic %d nrhs %d)\n",
              (int)want, famlevels, nofill, (int)c->S.nlev, fzenv, set, (int)lev_lo, (int)lev_hi, (int)fgroups_on, (int)c->D.fz_ok, (int)famt_disabled(), (int)fam2_disabled(), (int)c->D.lg_request,
              c->D.kc_ij != nullptr, a0.ymode, a0.ysc != nullptr, (int)c->D.fam_maxterms, (long long)c->D.cnnz, (long long)c->S.nsn, (long long)c->D.m, (int)use_generic(c), nrhs);
  }
  // kernels that read their input from U get dense panels built first (zeros + the constraint's entries)
  auto dense_input_on = [&](MfmaArgs& a, int cnt, double* Ub, int nr, hipStream_t s) {
    if (!sparse) return;
    for (int r0 = 0; r0 < nr; r0 += 65535) {
      MfmaArgs af = a;
      af.kc_j0 = a.kc_j0 + r0;
      // big panels are shared by several workgroups (column ranges): a single 4096 front would otherwise be filled by
      // one workgroup per right-hand side
      const int64_t pan = (int64_t)(a.nnmax + a.namax) * a.nnmax;
      const int nz = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(a.nnmax, 64), pan / 16384));
      launch(c, KID_scatter_constraints, k_panel_fill, dim3(cnt, std::min(65535, nr - r0), nz), dim3(256), s, af, Ub + (int64_t)r0 * ldu, ldu);
    }
    a.kc_ptr = nullptr;
  };
  auto dense_input = [&](MfmaArgs& a, int cnt) { dense_input_on(a, cnt, U, nrhs, st); };
  // (A handful of level-0 fronts outside the families -- synth50k: ONE leaf hangs off a mid front directly -- is a launch of
  // its own in the chain, 15 us for one workgroup.  On a side stream next to the family launch of level 1 it costs as
  // much in event waits: 6 us before and after the family launch, measured.)
  for (int64_t l = lev_lo; l < (lev_hi < 0 ? c->S.nlev : lev_hi); ++l) {    // [lev_lo, lev_hi): the caller may sweep in two parts
    for_level_classes(c, l, a0, [&](bool lds, MfmaArgs a, int cnt, size_t bytes, int thr) {
      static int fam_minrhs = -1;
      if (fam_minrhs < 0) { const char* e = sw_str("SMCP_FAM_MINRHS"); fam_minrhs = e ? atoi(e) : 1; }
      if (lds && a.nS > 0 && (sparse || nrhs >= fam_minrhs)) {
        // families: the childless members (level 0) are swept inside their parents' workgroups (k_hess_up_fam);
        // for one or two dense right-hand sides the per-workgroup set-up outweighs the saved exchange (measured)
        const int nS = a.nS;
        a.nS = 0;
        if (a.level > 0) {
          MfmaArgs af = a;
          af.t.lev = a.t.lev + (cnt - nS);
          if (fgroups_on && (size_t)l < c->famt_grp.size() && c->famt_grp[(size_t)l].ngroups > 0) {
            af.grp_ptr = c->famt_grp[(size_t)l].ptr; af.grp_list = c->famt_grp[(size_t)l].list; af.famt_ngrp = c->famt_grp[(size_t)l].ngroups;
          }
          if (!try_fam(c, af, nS, nrhs, U, ldu, st)) {
            // the level-0 members were skipped for this kernel: without it their panels and the parents' updates
            // would be missing, so the call must fail (end_call returns SMCP_EHIP)
            fprintf(stderr, "smcp_amd: family kernel not launchable (LDS budget or attribute)\n");
            if (!c->launch_err) c->launch_err = -1;
          }
        }
        cnt -= nS;
        if (cnt == 0) return;
      }
      int g = rhs_groups(cnt, nrhs, lds ? 2048 : 1024);
      if (lds) {
        hipStream_t ls = st;
        static int oldk = -1;
        if (oldk < 0) { const char* e = sw_str("SMCP_OLDLDS"); oldk = (e && e[0] == '1') ? 1 : 0; }
        size_t pbytes = (size_t)pad_layout(a.nnmax, a.namax, a.nchmax, a.panmax, a.pkmax, a.plansum).total * sizeof(double);
        if (!oldk && try_n16(c, a, cnt, nrhs, U, ldu, ls)) {
        } else if (!oldk && pbytes <= LDS_LIMIT) {
          dense_input_on(a, cnt, U, nrhs, ls);
          launch_lds(c, KID_hess_up_pad, k_hess_up_pad, dim3(cnt, g), dim3(pbytes > 48 * 1024 ? 512 : 256), pbytes, ls, a, U, ldu);
        } else {
          dense_input_on(a, cnt, U, nrhs, ls);
          launch_lds(c, KID_hess_up_mfma, k_hess_up_mfma<true>, dim3(cnt, g), dim3(thr), bytes, ls, a, U, ldu);
        }
      } else if (use_large() && c->D.gp_tptr) {
        {
          const csp_ctx::LfspGroups* grp = groups_on && (size_t)l < c->lfsp_grp.size() && c->lfsp_grp[(size_t)l].ngroups > 0 ? &c->lfsp_grp[(size_t)l] : nullptr;
          if (sparse && try_lfsp(c, a, cnt, nrhs, U, ldu, st, grp)) return;
          if (grp) {      // the parents would skip slots that the dense route is about to leave ... written, but the sums would be missing
            fprintf(stderr, "smcp_amd: sparse-input sweep of a grouped level not launchable\n");
            if (!c->launch_err) c->launch_err = -1;
          }
        }
        // Many right-hand sides on a few large fronts with children (the Schur sweeps over the top of synth50k: 8 fronts
        // x 100): the extend-add runs one workgroup per CU (LDS front) and leaves a partial last round, the three phase
        // kernels after it are short.  The right-hand sides are independent: two halves on two streams fill each
        // other's tails (every per-right-hand-side buffer is addressed through bases shifted to the half's first one).
        static int split_min = -1;
        if (split_min < 0) { const char* e = sw_str("SMCP_RHS_SPLIT"); split_min = e ? atoi(e) : 16; }
        static int parts = -1;
        if (parts < 0) { const char* e = sw_str("SMCP_RHS_PARTS"); parts = (e && e[0] == '3') ? 3 : 2; }
        const int ncu_split = c->D.ncu;
        // only when the extend-add needs more than one round of the chip: with fewer (front, right-hand side) pairs than
        // CUs (one rank's share of an 8-rank job: 100 pairs) there is no tail to fill and the halves only add launches
        // (measured on the partition of rank 0 of 8: 2.83 ms per step with the split, 2.67 without)
        // (with the task-drawing extend-add -- lf_assemble, SMCP_ALDS_DYN -- one launch keeps every CU busy to the last task
        // and a second one on another stream could not start before it ends, both holding the whole LDS of a CU: the split
        // is for the plain launch only.  Measured with both: 4.50 against 4.55 ms per step -- and the second launch's
        // duration would count its wait for CUs.)
        static int dynsplit = -1;
        if (dynsplit < 0) { const char* e = sw_str("SMCP_ALDS_DYN"); const char* f = sw_str("SMCP_RHS_SPLIT_DYN"); dynsplit = ((e && e[0] == '0') || (f && f[0] == '1')) ? 1 : 0; }
        if (dynsplit && split_min > 0 && a.nchmax > 0 && nrhs >= split_min && (int64_t)cnt * nrhs > ncu_split && Fork::enabled()) {
          auto part = [&](int r0, int nr, hipStream_t s) {       // right-hand sides r0 .. r0 + nr - 1 on stream s
            MfmaArgs ap = a;
            ap.nrhs = nr;
            ap.t.upd += (int64_t)r0 * a.t.updlen;
            ap.t.updp += (int64_t)r0 * a.t.updplen;
            ap.t.tmp += (int64_t)r0 * a.t.tmplen;
            ap.kc_j0 = a.kc_j0 + r0;
            dense_input_on(ap, cnt, U + (int64_t)r0 * ldu, nr, s);
            lf_up(c, ap, cnt, nr, U + (int64_t)r0 * ldu, ldu, s);
          };
          if (parts == 3 && nrhs >= 3) {
            const int h1 = nrhs / 3, h2 = (2 * nrhs) / 3;
            Fork f0(c, st, 0);
            part(h2, nrhs - h2, f0.s);
            Fork f1(c, st, 1);
            part(h1, h2 - h1, f1.s);
            part(0, h1, st);
            return;
          }
          const int h = nrhs / 2;
          Fork f(c, st, 0);
          part(h, nrhs - h, f.s);
          part(0, h, st);
          return;
        }
        // Wide childless fronts without separator (config 2: ONE dense 4096 clique) whose right-hand sides are sparse
        // constraints: the first phase, Z = Li Fl, is a sparse combination of columns of Li per column of Z (k_lf_zsp)
        // instead of a product of two dense triangles, the second phase never reads the input panel -- no dense panel is built
        // at all.  SMCP_ZSP=0: the dense route.
        static int zsp = -1;
        if (zsp < 0) { const char* e = sw_str("SMCP_ZSP"); zsp = (e && e[0] == '0') ? 0 : 1; }
        if (zsp && sparse && c->D.kc_sorted && a.nchmax == 0 && a.namax == 0 && lf_sym_split(a.nnmin) && a.nnmax <= LF_ZSP_MAXNN) {
          const int ntN = tiles64(a.nnmax);
          launch_lds(c, KID_lf_zsp, k_lf_zsp, dim3(a.nnmax, cnt, nrhs), dim3(256), (size_t)a.nnmax * sizeof(double), st, a, U, ldu);
          LAUNCH_PD(c, KID_lf_up2, k_lf_up2, dim3(umax1(ntN * (ntN + 1) / 2), cnt, nrhs), dim3(256), st, a, U, ldu);
          return;
        }
        const bool fzlev = c->fz_live && (size_t)a.level < c->fz_levels.size() && c->fz_levels[(size_t)a.level];
        if (sparse && a.nchmax > 0 && lf_assemble_fills(c, a, cnt, nrhs, fzlev)) {     // no k_panel_fill: the extend-add builds the panels
          lf_up(c, a, cnt, nrhs, U, ldu, st, true);
          return;
        }
        dense_input(a, cnt);
        if (yroot && a.namax == 0 && a.nnmax <= ROOT_FUSED_MAXNN && root_fused()) {
          if (a.nchmax > 0) lf_assemble(c, a, cnt, nrhs, U, ldu, 0, st, true);
          const int ntN = tiles64(a.nnmax);
          LAUNCH_PD(c, KID_lf_up1, k_lf_root1, dim3(umax1(ntN * ntN), cnt, nrhs), dim3(256), st, a, U, ldu, yroot);
          return;
        }
        lf_up(c, a, cnt, nrhs, U, ldu, st);
      }
      else { dense_input(a, cnt); launch_lds(c, KID_hess_up_mfma_hbm, k_hess_up_mfma<false>, dim3(cnt, nrhs), dim3(thr), 0, st, a, U, ldu); }
    }, set);
  }
}
// the same for the cliques of a set of the partition (1 = owned, 2 = replicated top), root -> leaves
void gather_set(csp_ctx* c, int set, const double* x, int64_t ldx, int nrhs, double* updbase, hipStream_t st) {
  TreeArgs a = tree_args(c);
  const LevelSet& LS = c->sets[set];
  for (int64_t l = c->S.nlev - 1; l >= 0; --l) {
    const LevelClass& L = LS.lvl[l];
    const int cnt = (int)(L.nI + L.nII);
    if (!cnt) continue;
    a.lev = LS.lev2 + LS.off[l];
    launch(c, KID_gather_level, k_gather_level, dim3(cnt, nrhs, gather_parts(std::max(L.namaxI, L.namaxII), (int64_t)cnt * nrhs, c->D.ncu)), dim3(NT), st, a, x, ldx, updbase);
  }
}
void hess_down_fast(csp_ctx* c, double* U, int nrhs, int64_t ldu, const double* ysc, int ymode, hipStream_t st, int set = 0,
                    const double* yroot = nullptr) {
  MfmaArgs a0 = mfma_args(c, ysc, ymode, nrhs);
  for (int64_t l = c->S.nlev - 1; l >= 0; --l)
    for_level_classes(c, l, a0, [&](bool lds, MfmaArgs a, int cnt, size_t bytes, int thr) {
      int g = rhs_groups(cnt, nrhs, lds ? 2048 : 1024);
      if (lds) {
        // (fewer threads per workgroup do NOT help this kernel on the leaves: 0.302 ms per step with 256, 0.330 with 128)
        // small fronts without scaling operand: one wave per (clique, right-hand side), operands in registers (front_n16.hip)
        static int dw = -1;
        if (dw < 0) { const char* e = sw_str("SMCP_DOWN_W"); dw = (e && e[0] == '0') ? 0 : 1; }
        if (dw && ymode == 0 && a.nnmax <= 16 && a.namax <= 64 && a.nnmax >= 1) {
          const int gw = rhs_groups((cnt + 3) / 4, nrhs, 4096);
          const dim3 grid((cnt + 3) / 4, gw), blk(256);
          switch ((std::max(a.namax, 1) + 15) / 16) {
            case 1: launch(c, KID_hess_down_mfma, k_hess_down_w<1>, grid, blk, st, a, U, ldu, cnt); break;
            case 2: launch(c, KID_hess_down_mfma, k_hess_down_w<2>, grid, blk, st, a, U, ldu, cnt); break;
            case 3: launch(c, KID_hess_down_mfma, k_hess_down_w<3>, grid, blk, st, a, U, ldu, cnt); break;
            default: launch(c, KID_hess_down_mfma, k_hess_down_w<4>, grid, blk, st, a, U, ldu, cnt); break;
          }
        } else
        if (ymode == 0) {     // no scaling operand: compact layout (the Y block is the largest buffer of the general one)
          const size_t b0 = (size_t)mfma_lds_doubles_for(WK_DOWN0, a.nnmax, a.namax) * sizeof(double);
          launch_lds(c, KID_hess_down_mfma, k_hess_down_mfma<true, WK_DOWN0>, dim3(cnt, g), dim3(b0 > 48 * 1024 ? 512 : 256), b0, st, a, U, ldu);
        } else
        launch_lds(c, KID_hess_down_mfma, k_hess_down_mfma<true>, dim3(cnt, g), dim3(thr), bytes, st, a, U, ldu);
      }
      else if (use_large() && yroot && a.namax == 0 && a.nnmax <= ROOT_FUSED_MAXNN && root_fused() && c->D.gp_tptr) {
        const int ntN = tiles64(a.nnmax);
        LAUNCH_PD(c, KID_lf_down3, k_lf_root2, dim3(umax1(ntN * (ntN + 1) / 2), cnt, nrhs), dim3(256), st, a, U, ldu, yroot);
      }
      else if (use_large() && (ymode == 0 || ymode == 3 || ymode == 2)) {
        if (ymode && a.namax) {   // Q = R Ghat_AN (ymode 3) / R^T Ghat_AN (ymode 2) first, then the unscaled sweep
          const int mtA = tiles64(a.namax), ntN = tiles64(a.nnmax);
          LAUNCH_PD(c, KID_lf_ri_an, k_lf_ri_an, dim3(umax1(mtA * ntN), cnt, nrhs), dim3(256), st, a, U, ldu, ymode == 2 ? 1 : 0);
          launch(c, KID_lf_copy_an, k_lf_copy_an, dim3(umax1(std::min(64, (a.namax * a.nnmax + 255) / 256)), cnt, nrhs), dim3(256), st, a, U, ldu, 0);
        }
        MfmaArgs a2 = a;
        a2.ymode = 0; a2.ysc = nullptr;
        lf_down(c, a2, cnt, nrhs, U, ldu, st);
      }
      else launch_lds(c, KID_hess_down_mfma_hbm, k_hess_down_mfma<false>, dim3(cnt, nrhs), dim3(thr), 0, st, a, U, ldu);
    }, set);
}

// inverse of the triangular factors in fac (large fronts: blocked over the chip; the rest: one workgroup each)
void lf_factor_inverse(csp_ctx* c, const MfmaArgs& a, int cnt, hipStream_t st) {
  dim3 blk(256);
  for (int ib = 0; ib < a.namax; ib += LB) {
    launch_lds(c, KID_lf_diag, k_lf_diag, dim3(cnt), diag_blk(), LF_DIAG_LDS, st, a, (double*)nullptr, c->D.fac, 4, ib, 0);
    if (ib > 0) launch(c, KID_lf_prep_s, k_lf_prep_s, dim3(umax1(tiles64(ib)), cnt), blk, st, a, (const double*)c->D.fac, c->D.faci, ib, 4);
    launch(c, KID_lf_prep_row, k_lf_prep_row, dim3(umax1(tiles64(ib) + 1), cnt), blk, st, a, (const double*)c->D.fac, c->D.faci, ib, 4, 0);
  }
}

// (with deferred status reports every call ends with k_latch_status, which moves a set flag to the latch and leaves the
// flags zero: the clear at the next entry point -- a 5 us launch in the chain, five or six per KKT solve -- is then skipped)
inline hipError_t zero_flag(csp_ctx* c, hipStream_t st) {
  const bool skip = c->lazy_status && c->flags_clean;
  c->flags_clean = false;
  return skip ? hipSuccess : hipMemsetAsync(c->D.info, 0, sizeof(int) * c->ntrial, st);
}
// yaa <- separator blocks of Y; fac <- their Cholesky factors (need_fac); faci <- inverses of those (need_inv).
// Each stage is skipped when the cache already holds it for the matrix at this address (see invalidate_tags).
// chol(Y_AA) of the family children that csp_cholesky_projected_inverse left out (DeviceCtx::fac_partial)
void complete_fac(csp_ctx* c, hipStream_t st) {
  if (!c->D.fac_partial) return;
  c->D.fac_partial = false;
  MfmaArgs a0 = mfma_args(c, nullptr, 0, 1);
  for_level_classes(c, 0, a0, [&](bool lds, MfmaArgs am, int cnt, size_t, int) {
    if (!lds || !am.namax || am.nS <= 0) return;
    am.t.lev = am.t.lev + (cnt - am.nS);
    const size_t bytes = ((size_t)padld(am.namax) * am.namax + 256 + 8) * sizeof(double);
    launch_lds(c, KID_factor_yaa_lds, k_factor_yaa_lds, dim3(am.nS), dim3(fact_threads(am, 256, 2)), bytes, st, am, (const double*)c->D.yaa, c->D.fac);
  });
}
int prepare_yaa(csp_ctx* c, const double* Y, bool need_fac, hipStream_t st, bool need_inv, bool allow_partial) {
  TreeArgs a = tree_args(c);
  const bool nocache = cache_off();
  if (need_inv) need_fac = true;
  if (need_fac && !allow_partial && c->D.fac_partial && c->D.fac_tag == Y && Y && !nocache) complete_fac(c, st);
  if (nocache || c->D.yaa_tag != Y || !Y) {
    gather_all(c, Y, 0, 1, c->D.yaa, st);
    c->D.yaa_tag = Y;
    c->D.fac_tag = c->D.faci_tag = nullptr;
    c->D.part_valid = false;
    if (Y) fp_record(c, 2, Y, st);
  } else if (int rc = fp_check(c, 2, Y, st)) return rc;
  const bool fast = !use_generic(c) && use_large();
  if (need_fac && c->D.fac_tag != Y) {
    if (fast) {
      // k_factor_yaa_lds reads yaa and writes the lower triangle of fac (what every reader of fac uses; the strict upper
      // triangles of fac stay zero from csp_device_init); only the large fronts factor in place and need their blocks
      // copied first -- not all 85 MB of synth50k (37 us per step)
      MfmaArgs a0 = mfma_args(c, nullptr, 0, 1);
      for_all_large(c, a0, [&](MfmaArgs am, int cnt) {
        if (am.namax) launch(c, KID_axpby, k_copy_upd_blocks, dim3(cnt, umax1(std::min(64, (am.namax * am.namax + 1023) / 1024))), dim3(256), st,
                             am.t, (const double*)c->D.yaa, c->D.fac);
      });
      {
        // clique-local: the large fronts' factorisations on one side stream, the LDS classes alternating between the
        // caller's stream and a second one (leaves and mid fronts of synth50k run side by side)
        Fork fl(c, st, 0);
        for_all_large(c, a0, [&](MfmaArgs am, int cnt) { if (am.namax) lf_factor_yaa(c, am, cnt, c->D.fac, fl.s); });
        Fork fs(c, st, 1);
        int turn = 0;
        for (int64_t l = 0; l < c->S.nlev; ++l)
          for_level_classes(c, l, a0, [&](bool lds, MfmaArgs am, int cnt, size_t, int) {
            if (!lds) return;
            size_t bytes = ((size_t)padld(am.namax) * am.namax + 256 + 8) * sizeof(double);
            if (am.namax) launch_lds(c, KID_factor_yaa_lds, k_factor_yaa_lds, dim3(cnt), dim3(fact_threads(am, 256, 2)), bytes, (turn++ & 1) ? fs.s : st, am, (const double*)c->D.yaa, c->D.fac);
          });
      }
    } else {
      launch(c, KID_factor_yaa, k_factor_yaa, dim3((int)c->S.nsn), dim3(NT), st, a, c->D.yaa, c->D.fac);
    }
    c->D.fac_tag = Y;
    c->D.fac_partial = false;
    c->D.faci_tag = nullptr;
    c->D.fac_gen++;
  }
  if (need_inv && c->D.faci_tag != Y) {
    if (fast) {
      a.lev = c->D.lev3idx;
      MfmaArgs a0 = mfma_args(c, nullptr, 0, 1);
      static const bool rd = sw_on("SMCP_FACI_LDS", true);
      bool all_lds = rd;
      if (rd)
        for (int64_t l = 0; l < c->S.nlev; ++l)
          for_level_classes(c, l, a0, [&](bool lds, MfmaArgs am, int, size_t, int) { if (lds && am.namax > 64) all_lds = false; });
      if (!c->D.nI_total) {
      } else if (!all_lds) {
        launch(c, KID_factor_inverse, k_factor_inverse, dim3((int)c->D.nI_total), dim3(256), st, a, (const double*)c->D.fac, c->D.faci);
      } else {
        for (int64_t l = 0; l < c->S.nlev; ++l)
          for_level_classes(c, l, a0, [&](bool lds, MfmaArgs am, int cnt, size_t, int) {
            if (!lds || !am.namax) return;
            if (am.namax <= 32)
              launch_lds(c, KID_factor_inverse_lds, k_factor_inverse_lds<33>, dim3(cnt), dim3(128), (2 * 33 * 32 + 256) * sizeof(double), st, am, (const double*)c->D.fac, c->D.faci);
            else
              launch_lds(c, KID_factor_inverse_lds, k_factor_inverse_lds<65>, dim3(cnt), dim3(256), (2 * 65 * 64 + 1024) * sizeof(double), st, am, (const double*)c->D.fac, c->D.faci);
          });
      }
      for_all_large(c, a0, [&](MfmaArgs am, int cnt) { if (am.namax) lf_factor_inverse(c, am, cnt, st); });
    } else {
      a.lev = c->D.levidx;
      launch(c, KID_factor_inverse, k_factor_inverse, dim3((int)c->S.nsn), dim3(256), st, a, (const double*)c->D.fac, c->D.faci);
    }
    c->D.faci_tag = Y;
  }
  return 0;
}

// The inverse-factor kernels keep S (16 KB, completion only) in static LDS next to the dynamic working set
constexpr size_t INV_STATIC_LDS = 16 * 128 * sizeof(double);
// fronts with at least this many columns are spread over the chip by the tiled phase kernels; narrower
// large fronts (long chains of thin cliques with big separators) take one workgroup and one launch each
constexpr int LF_INV_MIN_NN = 24;

// G^-adj (clique-local once Z_AA of every clique has been gathered from the input), then the scaling `ymode`
void hess_down_inv_fast(csp_ctx* c, const double* L, double* U, int nrhs, int64_t ldu, int ymode, hipStream_t st) {
  gather_all(c, U, ldu, nrhs, c->D.upd, st);
  MfmaArgs a0 = mfma_args(c, c->D.faci, ymode, nrhs);
  a0.LK = L;
  dim3 blk(256);
  for (int64_t l = 0; l < c->S.nlev; ++l)
    for_level_classes(c, l, a0, [&](bool lds, MfmaArgs a, int cnt, size_t bytes, int thr) {
      if (lds) {
        int g = rhs_groups(cnt, nrhs, 2048);
        launch_lds(c, KID_hess_down_inv_mfma, k_hess_down_inv_mfma<true>, dim3(cnt, g), dim3(thr), bytes, st, a, U, ldu);
      } else if (!use_large()) {
        launch_lds(c, KID_hess_down_inv_mfma_hbm, k_hess_down_inv_mfma<false>, dim3(cnt, nrhs), dim3(thr), 0, st, a, U, ldu);
      }
    });
  if (use_large())
    for_all_large(c, a0, [&](MfmaArgs a, int cnt) {   // clique-local: all large fronts of the tree in one set of launches
      const int mtA = tiles64(a.namax), ntN = tiles64(a.nnmax);
      LAUNCH_PD(c, KID_lf_dinv1, k_lf_dinv1, dim3(umax1(mtA * ntN + ntN * ntN), cnt, nrhs), blk, st, a, U, ldu);
      LAUNCH_PD(c, KID_lf_dinv2, k_lf_dinv2, dim3(umax1(ntN * (ntN + 1) / 2 + mtA * ntN), cnt, nrhs), blk, st, a, U, ldu);
      if (ymode == 1 && a.namax) {
        LAUNCH_PD(c, KID_lf_ri_an, k_lf_ri_an, dim3(umax1(mtA * ntN), cnt, nrhs), blk, st, a, U, ldu, 1);
        launch(c, KID_lf_copy_an, k_lf_copy_an, dim3(umax1(std::min(64, (a.namax * a.nnmax + 255) / 256)), cnt, nrhs), blk, st, a, U, ldu, 0);
      }
    });
}
// G^-1, leaves -> root; ymode 3: G_AN = Ri^T Ghat_AN first
void hess_up_inv_fast(csp_ctx* c, const double* L, double* U, int nrhs, int64_t ldu, int ymode, hipStream_t st) {
  MfmaArgs a0 = mfma_args(c, c->D.faci, ymode, nrhs);
  a0.LK = L;
  dim3 blk(256);
  for (int64_t l = 0; l < c->S.nlev; ++l)
    for_level_classes(c, l, a0, [&](bool lds, MfmaArgs a, int cnt, size_t bytes, int thr) {
      if (lds) {
        int g = rhs_groups(cnt, nrhs, 2048);
        launch_lds(c, KID_hess_up_inv_mfma, k_hess_up_inv_mfma<true>, dim3(cnt, g), dim3(thr), bytes, st, a, U, ldu);
      } else if (use_large() && c->D.gp_tptr && a.nnmax >= LF_INV_MIN_NN) {
        const int mtA = tiles64(a.namax), ntN = tiles64(a.nnmax);
        const dim3 gcopy(umax1(std::min(64, (a.namax * a.nnmax + 255) / 256)), cnt, nrhs);
        if (ymode == 3 && a.namax) {
          LAUNCH_PD(c, KID_lf_ri_an, k_lf_ri_an, dim3(umax1(mtA * ntN), cnt, nrhs), blk, st, a, U, ldu, 1);
          launch(c, KID_lf_copy_an, k_lf_copy_an, gcopy, blk, st, a, U, ldu, 0);
        }
        LAUNCH_PD(c, KID_lf_uinv1, k_lf_uinv1, dim3(umax1(mtA * ntN + ntN * ntN), cnt, nrhs), blk, st, a, U, ldu);
        LAUNCH_PD(c, KID_lf_uinv2, k_lf_uinv2, dim3(umax1(mtA * (mtA + 1) / 2 + mtA * ntN + ntN * (ntN + 1) / 2), cnt, nrhs), blk, st, a, U, ldu);
        if (a.namax) launch(c, KID_lf_copy_an, k_lf_copy_an, gcopy, blk, st, a, U, ldu, 0);
        lf_assemble(c, a, cnt, nrhs, U, ldu, 1, st);
        if (a.namax) launch(c, KID_lf_pack_upd, k_lf_pack_upd, dim3(umax1(std::min(64, (a.namax * a.namax + 255) / 256)), cnt, nrhs), blk, st, a);
      } else {
        launch_lds(c, KID_hess_up_inv_mfma_hbm, k_hess_up_inv_mfma<false>, dim3(cnt, nrhs), dim3(thr), 0, st, a, U, ldu);
      }
    });
}

int hessian_impl(csp_ctx* c, const double* L, double* U, int64_t nrhs, int64_t ldu, int adj, int inv,
                 hipStream_t st) {
  TreeArgs a = tree_args(c);
  const int nsn = (int)c->S.nsn;
  auto up = [&]() {
    for_levels_up(c, [&](const int32_t* lev, int cnt) {
      a.lev = lev;
      launch(c, KID_hess_up_level, k_hess_up_level, dim3(cnt, (int)nrhs), dim3(NT), st, a, L, U, ldu);
    });
  };
  auto down = [&]() {
    for_levels_down(c, [&](const int32_t* lev, int cnt) {
      a.lev = lev;
      launch(c, KID_hess_down_level, k_hess_down_level, dim3(cnt, (int)nrhs), dim3(NT), st, a, L, U, ldu);
    });
  };
  auto up_inv = [&]() {
    for_levels_up(c, [&](const int32_t* lev, int cnt) {
      a.lev = lev;
      launch(c, KID_hess_up_inv_level, k_hess_up_inv_level, dim3(cnt, (int)nrhs), dim3(NT), st, a, L, U, ldu);
    });
  };
  auto down_inv = [&]() {
    gather_all(c, U, ldu, (int)nrhs, c->D.upd, st);
    launch(c, KID_hess_down_inv_all, k_hess_down_inv_all, dim3(nsn, (int)nrhs), dim3(NT), st, a, L, U, ldu);
  };
  auto scale = [&](int mode) {
    launch(c, KID_scale_an, k_scale_an, dim3(nsn, (int)nrhs), dim3(NT), st, a, c->D.yaa, c->D.fac, U, ldu, mode);
  };
  if (!inv && !use_generic(c)) {
    // LK must have been prepared for this L (prep_lk) by the caller
    if (adj == 0) hess_up_fast(c, U, (int)nrhs, ldu, c->D.fac, 2, st);
    else if (adj == 1) hess_down_fast(c, U, (int)nrhs, ldu, c->D.fac, 3, st);
    else {
      // (Y: the matrix the cached Y_AA blocks were gathered from -- the pair (L, Y) of this call; its root blocks serve the
      // fronts without separator directly)
      const double* yroot = (const double*)c->D.yaa_tag;
      hess_up_fast(c, U, (int)nrhs, ldu, c->D.yaa, 1, st, 0, -1, nullptr, 0, -1, yroot);
      hess_down_fast(c, U, (int)nrhs, ldu, nullptr, 0, st, 0, yroot);
    }
  } else if (!inv) {
    if (adj == 0) { up(); scale(0); }
    else if (adj == 1) { scale(1); down(); }
    else { up(); scale(4); down(); }
  } else if (!use_generic(c)) {
    if (adj == 0) hess_up_inv_fast(c, L, U, (int)nrhs, ldu, 3, st);
    else if (adj == 1) hess_down_inv_fast(c, L, U, (int)nrhs, ldu, 2, st);
    else { hess_down_inv_fast(c, L, U, (int)nrhs, ldu, 1, st); hess_up_inv_fast(c, L, U, (int)nrhs, ldu, 0, st); }
  } else {
    if (adj == 0) { scale(2); up_inv(); }
    else if (adj == 1) { down_inv(); scale(3); }
    else { down_inv(); scale(5); up_inv(); }
  }
  return 0;
}


bool fam_off() {
  static int off = -1;
  if (off < 0) { const char* e = sw_str("SMCP_FAM"); off = (e && e[0] == '0') ? 1 : 0; }
  return off == 1;
}
// dynamic LDS of the family kernel instantiation that serves (parent separator famna, child separator famcna)
size_t fam_bytes_for(int famna, int famcna, int fampan, int fampk) {
  const int nat = std::max(1, (famna + 15) / 16), natc = std::max(1, (famcna + 15) / 16);
  switch (nat * 2 + natc - 1) {
    case 2: return fam_lds_bytes<1, 1>(fampan, fampk);
    case 3: return fam_lds_bytes<1, 2>(fampan, fampk);
    case 4: return fam_lds_bytes<2, 1>(fampan, fampk);
    case 5: return fam_lds_bytes<2, 2>(fampan, fampk);
    case 6: return fam_lds_bytes<3, 1>(fampan, fampk);
    case 7: return fam_lds_bytes<3, 2>(fampan, fampk);
    case 8: return fam_lds_bytes<4, 1>(fampan, fampk);
    case 9: return fam_lds_bytes<4, 2>(fampan, fampk);
  }
  return (size_t)1 << 30;
}

// Splits the cliques selected by `keep` into per-level lists (LDS-class first -- its family tail last --, large
// fronts after) and records the sizing maxima of each class.  lev2 is the concatenation of the lists, off[l] its start.
template <class Keep>
void classify_levels(const Symbolic& S, Keep keep, std::vector<LevelClass>& lvl, std::vector<int32_t>& lev2,
                     std::vector<int64_t>& off) {
  lvl.assign(S.nlev, LevelClass());
  lev2.clear();
  off.assign(S.nlev + 1, 0);
  auto fits = [&](int64_t k) { return (size_t)mfma_lds_doubles((int)S.nn(k), (int)S.na(k)) * sizeof(double) <= LDS_LIMIT; };
  // pass 1: sizing of the LDS class of every level; the joint maxima may not fit even if every clique does: the
  // level's LDS class is demoted to the large-front class then
  std::vector<uint8_t> demoted(S.nlev, 0);
  for (int64_t l = 0; l < S.nlev; ++l) {
    int nnm = 0, nam = 0;
    bool any = false;
    for (int64_t q = S.levptr[l]; q < S.levptr[l + 1]; ++q) {
      const int64_t k = S.levidx[q];
      if (!keep(k) || !fits(k)) continue;
      any = true;
      nnm = std::max<int>(nnm, (int)S.nn(k));
      nam = std::max<int>(nam, (int)S.na(k));
    }
    if (any && (size_t)mfma_lds_doubles(nnm, nam) * sizeof(double) > LDS_LIMIT) demoted[l] = 1;
  }
  auto small = [&](int64_t k) { return fits(k) && !demoted[S.level[k]]; };
  // pass 2: families.  Parent: small front, nn <= 16, na <= 64, 1..8 children, every child kept, childless, small,
  // nn <= 16, na <= 32; all candidates of a level or none (one launch geometry per level).
  std::vector<uint8_t> special(S.nsn, 0);
  if (!fam_off())
    for (int64_t l = 1; l < S.nlev; ++l) {
      std::vector<int64_t> cand;
      int famna = 0, fampan = 0, fampk = 0, famcna = 0, famnn = 0, famcnn = 0;
      for (int64_t q = S.levptr[l]; q < S.levptr[l + 1]; ++q) {
        const int64_t k = S.levidx[q];
        if (!keep(k) || !small(k) || S.nn(k) > 16 || S.na(k) > 64) continue;
        const int64_t nch = S.chptr[k + 1] - S.chptr[k];
        if (nch < 1 || nch > 8) continue;
        bool ok = true;
        int cna = 0, cnn = 0;
        for (int64_t q2 = S.chptr[k]; q2 < S.chptr[k + 1] && ok; ++q2) {
          const int64_t ch = S.chidx[q2];
          ok = keep(ch) && S.chptr[ch + 1] == S.chptr[ch] && small(ch) && S.nn(ch) <= 16 && S.na(ch) <= 32 && S.na(ch) >= 1;
          cna = std::max<int>(cna, (int)S.na(ch));
          cnn = std::max<int>(cnn, (int)S.nn(ch));
        }
        if (!ok) continue;
        cand.push_back(k);
        famnn = std::max<int>(famnn, (int)S.nn(k));
        famcnn = std::max(famcnn, cnn);
        famna = std::max<int>(famna, (int)S.na(k));
        fampan = std::max<int>(fampan, (int)(S.nf(k) * S.nn(k)));
        fampk = std::max<int>(fampk, (int)(S.na(k) * (S.na(k) + 1) / 2));
        famcna = std::max(famcna, cna);
      }
      if (cand.empty() || fam_bytes_for(famna, famcna, fampan, fampk) > LDS_LIMIT) continue;
      LevelClass& L = lvl[l];
      L.famna = famna; L.fampan = fampan; L.fampk = fampk; L.famcna = famcna; L.famnn = famnn; L.famcnn = famcnn;
      for (int64_t k : cand) {
        special[k] = 1;
        for (int64_t q2 = S.chptr[k]; q2 < S.chptr[k + 1]; ++q2) special[S.chidx[q2]] = 1;
      }
    }
  // pass 3: the lists
  for (int64_t l = 0; l < S.nlev; ++l) {
    LevelClass& L = lvl[l];
    off[l] = (int64_t)lev2.size();
    int64_t b = S.levptr[l], e = S.levptr[l + 1];
    for (int pass = 0; pass < 3; ++pass)
      for (int64_t q = b; q < e; ++q) {
        int64_t k = S.levidx[q];
        if (!keep(k)) continue;
        const bool sm = small(k);
        const int cls = sm ? (special[k] ? 1 : 0) : 2;
        if (cls != pass) continue;
        lev2.push_back((int32_t)k);
        if (sm) {
          L.nI++;
          if (special[k]) L.nS++;
          L.nnmaxI = std::max<int>(L.nnmaxI, (int)S.nn(k));
          L.namaxI = std::max<int>(L.namaxI, (int)S.na(k));
          int64_t rs = 0;
          for (int64_t q2 = S.chptr[k]; q2 < S.chptr[k + 1]; ++q2) rs += S.na(S.chidx[q2]) * (S.na(S.chidx[q2]) + 1) / 2;
          L.plansumI = (int)std::max<int64_t>(L.plansumI, rs);
          L.nchmaxI = std::max<int>(L.nchmaxI, (int)(S.chptr[k + 1] - S.chptr[k]));
          L.panmaxI = std::max<int>(L.panmaxI, (int)(S.nf(k) * S.nn(k)));
          L.pkmaxI = std::max<int>(L.pkmaxI, (int)(S.na(k) * (S.na(k) + 1) / 2));
        } else {
          L.nII++;
          L.nnmaxII = std::max<int>(L.nnmaxII, (int)S.nn(k));
          L.nnminII = std::min<int>(L.nnminII, (int)S.nn(k));
          L.namaxII = std::max<int>(L.namaxII, (int)S.na(k));
          L.nchmaxII = std::max<int>(L.nchmaxII, (int)(S.chptr[k + 1] - S.chptr[k]));
        }
      }
  }
  off[S.nlev] = (int64_t)lev2.size();
}

}  // namespace

extern "C" {

csp_ctx* csp_symbolic_create(int64_t n, const int64_t* colptr, const int64_t* rowind,
                             const int64_t* perm, int64_t* info) {
  csp_ctx* c = new (std::nothrow) csp_ctx();
  if (!c) { if (info) *info = SMCP_ENOMEM; return nullptr; }
  int rc = symbolic_build(n, colptr, rowind, perm, c->S);
  if (info) *info = rc;
  if (rc) { delete c; return nullptr; }
  return c;
}

// K independent copies of the pattern as ONE symbolic object: the clique forest in which copy t owns the cliques
// t * nsn .. (t + 1) * nsn - 1, the columns t * n .., the blkval range [t * blklen, (t + 1) * blklen) and so on.
// Every tree operation on it runs the launches of one factorisation, K times as wide (the concurrent trial
// factorisations of the line searches, smcp_amd/chordal.py probe_cone).
csp_ctx* csp_symbolic_replicate(const csp_ctx* base, int64_t K, int64_t* info) {
  if (!base || K < 1 || K > 16 || base->ntrial != 1 || base->S.n * K > (int64_t)0x7fffffff) { if (info) *info = SMCP_EINVAL; return nullptr; }
  csp_ctx* c = new (std::nothrow) csp_ctx();
  if (!c) { if (info) *info = SMCP_ENOMEM; return nullptr; }
  const Symbolic& B = base->S;
  Symbolic& S = c->S;
  S.n = B.n * K; S.nnz = B.nnz * K; S.nsn = B.nsn * K; S.fill = B.fill * K;
  S.nlev = B.nlev; S.max_nn = B.max_nn; S.max_na = B.max_na; S.max_front = B.max_front;
  auto rep_ptr = [&](const std::vector<int64_t>& src, std::vector<int64_t>& dst) {     // pointer array: shift by the copy's total
    const int64_t cnt = (int64_t)src.size() - 1, tot = src.back();
    dst.resize(cnt * K + 1);
    for (int64_t t = 0; t < K; ++t)
      for (int64_t i = 0; i < cnt; ++i) dst[t * cnt + i] = src[i] + t * tot;
    dst[cnt * K] = tot * K;
  };
  auto rep_val = [&](const auto& src, auto& dst, int64_t shift, bool keep_negative) {  // values: shift by `shift` per copy
    const int64_t cnt = (int64_t)src.size();
    dst.resize(cnt * K);
    for (int64_t t = 0; t < K; ++t)
      for (int64_t i = 0; i < cnt; ++i)
        dst[t * cnt + i] = (keep_negative && src[i] < 0) ? src[i] : (decltype(dst[0] + 0))(src[i] + t * shift);
  };
  rep_val(B.p, S.p, B.n, false);
  rep_val(B.ip, S.ip, B.n, false);
  rep_ptr(B.snptr, S.snptr);
  rep_val(B.snode, S.snode, B.nsn, false);
  rep_val(B.snpar, S.snpar, B.nsn, true);
  rep_ptr(B.rowptr, S.rowptr);
  rep_val(B.rowidx, S.rowidx, B.n, false);
  rep_ptr(B.sepptr, S.sepptr);
  rep_val(B.relidx, S.relidx, 0, false);
  rep_ptr(B.blkptr, S.blkptr);
  rep_ptr(B.updptr, S.updptr);
  rep_ptr(B.updpptr, S.updpptr);
  rep_ptr(B.chptr, S.chptr);
  rep_val(B.chidx, S.chidx, B.nsn, false);
  rep_val(B.level, S.level, 0, false);
  rep_ptr(B.ccsptr, S.ccsptr);
  S.levptr.assign(B.nlev + 1, 0);
  S.levidx.resize(S.nsn);
  int64_t q = 0;
  for (int64_t l = 0; l < B.nlev; ++l) {                 // a level of the forest = that level of every copy
    S.levptr[l] = q;
    for (int64_t t = 0; t < K; ++t)
      for (int64_t e = B.levptr[l]; e < B.levptr[l + 1]; ++e) S.levidx[q++] = B.levidx[e] + t * B.nsn;
  }
  S.levptr[B.nlev] = q;
  c->ntrial = K;
  if (info) *info = 0;
  return c;
}
// failure flags of the copies after csp_cholesky / csp_completion on a replicated context (0 = inside the cone,
// otherwise 1 + the failing clique within the copy); the factorisation call has already synchronised and read them
int csp_trial_flags(csp_ctx* c, int64_t K, int* out) {
  if (!c || !out || K != c->ntrial || c->D.device < 0) return SMCP_EINVAL;
  for (int64_t t = 0; t < K; ++t) out[t] = c->D.info_host[t];
  return 0;
}

int csp_lazy_status(csp_ctx* c, int on) {
  if (int rc = ready(c)) return rc;
  c->lazy_status = on != 0;
  return 0;
}
static int flush_pending_potrf(csp_ctx* c, hipStream_t st, const void* only, bool drop);      // kkt.hip
int csp_status(csp_ctx* c, void* stream) {
  if (int rc = ready(c)) return rc;
  hipStream_t st = (hipStream_t)stream;
  // a Schur complement that kkt_schur_factor left unfactored (deferred status) and nobody has used yet: its verdict belongs
  // to this read-out
  if (int rc = flush_pending_potrf(c, st, nullptr, false)) return rc < 0 ? rc : rc;
  int v = 0;
  HIPCHK(hipMemcpyAsync(c->D.info_host + 16, c->D.info + 16, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  v = c->D.info_host[16];
  if (v) HIPCHK(hipMemsetAsync(c->D.info + 16, 0, sizeof(int), st));
  if (c->launch_err) { c->launch_err = 0; return SMCP_EHIP; }
  return v;
}

void csp_symbolic_destroy(csp_ctx* c) {
  if (!c) return;
  DeviceCtx& D = c->D;
  if (D.device >= 0) {
    hipSetDevice(D.device);
    void* ptrs[] = {D.lfsp_skip, D.famt_skip, D.both_skip, D.trsm_x, D.fp, D.fp_bad, D.gsl_start, D.gsl_len, D.lg_list, D.lg_slot, D.lg_eptr, D.lg_epk, D.lg_ew, D.lg_remap, D.lg_tab, D.sp_rt, D.sp_mk, D.lfsp_list, D.faci, D.lfd, D.lev3idx, D.updp, D.gp_tptr, D.gp_tgt, D.gp_cptr, D.gp_src, D.sw, D.gpart, D.lev2idx, D.lk, D.cl, D.rowidx, D.relidx, D.chidx, D.levidx, D.upd, D.yaa, D.fac, D.tmp, D.tmpptr,
                    D.red, D.info, D.cptr, D.cidx, D.cval, D.cwval, D.rpos, D.rptr, D.rcon, D.rval, D.ustack, D.qr_ws,
                    D.a_r, D.a_c, D.s_rloc, D.s_cloc, D.dlist, D.slist, D.kidx, D.vbuf, D.hd, D.kc_ptr, D.kc_off, D.kc_val, D.hinv, D.kc_ij, D.famc, D.scm_owner};
    if (c->side_fork) { Fork* f = (Fork*)c->side_fork; c->side_fork = nullptr; f->join(); delete f; }
    D.h_pending = nullptr;      // (a deferred factorisation nobody asked for dies with the context)
    for (auto& W : c->flow_ws) for (void* q : {(void*)W.P, (void*)W.dinv, (void*)W.flags}) if (q) hipFree(q);
    for (auto& kv : c->flow_plans) { if (kv.second.own_ptr) hipFree(kv.second.own_ptr); if (kv.second.own_tile) hipFree(kv.second.own_tile); }
    for (void* p : ptrs) if (p) hipFree(p);
    for (auto& G : c->lfsp_grp) { if (G.ptr) hipFree(G.ptr); if (G.list) hipFree(G.list); }
    for (auto& G : c->famt_grp) { if (G.ptr) hipFree(G.ptr); if (G.list) hipFree(G.list); }
    for (int q = 0; q < 2; ++q) {
      if (c->aux_stream[q]) { (void)hipStreamSynchronize(c->aux_stream[q]); (void)hipStreamDestroy(c->aux_stream[q]); }
      if (c->aux_join[q]) (void)hipEventDestroy(c->aux_join[q]);
    }
    if (c->aux_fork) (void)hipEventDestroy(c->aux_fork);
    for (int set = 1; set <= 2; ++set) if (c->sets[set].lev2) hipFree(c->sets[set].lev2);
    for (void* p : {(void*)c->xr_roots, (void*)c->xr_owner, (void*)c->xr_bptr}) if (p) hipFree(p);
    if (D.info_host) hipHostFree(D.info_host);
  }
  delete c;
}

int64_t csp_symbolic_query(const csp_ctx* c, int what, int64_t* out) {
  if (!c) return SMCP_EINVAL;
  const Symbolic& S = c->S;
  auto put64 = [&](const std::vector<int64_t>& v) { if (out) std::copy(v.begin(), v.end(), out); return (int64_t)v.size(); };
  auto put32 = [&](const std::vector<int32_t>& v) { if (out) std::copy(v.begin(), v.end(), out); return (int64_t)v.size(); };
  switch (what) {
    case CSP_Q_SCALARS: {
      int64_t s[10] = {S.n, S.nnz, S.nsn, S.fill, S.blklen(), S.updlen(), S.nlev, S.max_nn, S.max_na, S.max_front};
      if (out) std::copy(s, s + 10, out);
      return 10;
    }
    case CSP_Q_PERM: return put64(S.p);
    case CSP_Q_IPERM: return put64(S.ip);
    case CSP_Q_SNPTR: return put64(S.snptr);
    case CSP_Q_SNPAR: return put64(S.snpar);
    case CSP_Q_ROWPTR: return put64(S.rowptr);
    case CSP_Q_ROWIDX: return put32(S.rowidx);
    case CSP_Q_SEPPTR: return put64(S.sepptr);
    case CSP_Q_RELIDX: return put32(S.relidx);
    case CSP_Q_BLKPTR: return put64(S.blkptr);
    case CSP_Q_UPDPTR: return put64(S.updptr);
    case CSP_Q_CHPTR: return put64(S.chptr);
    case CSP_Q_CHIDX: return put64(S.chidx);
    case CSP_Q_LEVPTR: return put64(S.levptr);
    case CSP_Q_LEVIDX: return put64(S.levidx);
    case CSP_Q_CCSPTR: return put64(S.ccsptr);
    case CSP_Q_SNODE: return put64(S.snode);
    case CSP_Q_FAMILY: {
      std::vector<int64_t> f = c->fam;
      f.resize(S.nsn, 0);
      return put64(f);
    }
  }
  return SMCP_EINVAL;
}

int csp_maxcardsearch(int64_t n, const int64_t* colptr, const int64_t* rowind, int64_t* order) {
  if (n <= 0 || !colptr || !rowind || !order) return SMCP_EINVAL;
  maxcardsearch(n, colptr, rowind, order);
  return 0;
}
int csp_mindegree(int64_t n, const int64_t* colptr, const int64_t* rowind, int64_t* order) {
  if (n <= 0 || !colptr || !rowind || !order) return SMCP_EINVAL;
  mindegree(n, colptr, rowind, order);
  return 0;
}
int csp_index_map(const csp_ctx* c, int64_t cnt, const int64_t* I, const int64_t* J, int64_t* out) {
  if (!c || cnt < 0) return SMCP_EINVAL;
  index_map(c->S, cnt, I, J, out);
  return 0;
}

// copies 1 .. K-1 of the extend-add gather plan of a K-fold replicated pattern: targets repeat, sources shift by the
// packed update length of one copy, the source ranges by the source count of one copy
__global__ void k_replicate_plan(int32_t* tgt, int64_t* cptr, int32_t* src, int64_t nt1, int64_t ns1, int64_t up1, int64_t K) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t e = g; e < nt1 * (K - 1); e += stride) {
    const int64_t t = 1 + e / nt1, q = e % nt1;
    tgt[t * nt1 + q] = tgt[q];
    cptr[t * nt1 + q + 1] = cptr[q + 1] + t * ns1;
  }
  for (int64_t e = g; e < ns1 * (K - 1); e += stride) {
    const int64_t t = 1 + e / ns1, q = e % ns1;
    src[t * ns1 + q] = (int32_t)(src[q] + t * up1);
  }
}

int csp_device_init(csp_ctx* c, int device, int64_t max_rhs) {
  if (!c || max_rhs < 1) return SMCP_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return SMCP_ENODEV;
  if (device < 0 || device >= ndev) return SMCP_EINVAL;
  DeviceCtx& D = c->D;
  const Symbolic& S = c->S;
  if (D.device == device && D.max_rhs >= max_rhs) return 0;
  // One process drives one GPU (one rank per GPU under torch.distributed): the launch helpers cache function
  // attributes, occupancy and the CU count per process, and a context's buffers live on the device it was
  // first initialised on.  A second device -- for this context or for another one of this process -- is refused.
  static int bound_device = -1;
  if (D.device >= 0 && D.device != device) return SMCP_EINVAL;
  if (bound_device >= 0 && bound_device != device) return SMCP_EINVAL;
  bound_device = device;
  HIPCHK(hipSetDevice(device));
  SetupClock clk("csp_device_init");
  if (D.device < 0) {
    std::vector<CliqueDesc> cl(S.nsn);
    for (int64_t k = 0; k < S.nsn; ++k) {
      CliqueDesc& d = cl[k];
      d.blk = S.blkptr[k];
      d.upd = S.updptr[k];
      d.updp = S.updpptr[k];
      d.rows = S.rowptr[k];
      d.rel = S.sepptr[k];
      d.nn = (int32_t)S.nn(k);
      d.na = (int32_t)S.na(k);
      d.parent = (int32_t)S.snpar[k];
      d.chbeg = (int32_t)S.chptr[k];
      d.chend = (int32_t)S.chptr[k + 1];
      d.first = (int32_t)S.snptr[k];
      d.pad = -1;
    }
    std::vector<int32_t> ch(S.chidx.begin(), S.chidx.end()), lev(S.levidx.begin(), S.levidx.end());
    c->h_tmpptr.resize(S.nsn + 1);
    for (int64_t k = 0; k <= S.nsn; ++k) c->h_tmpptr[k] = 2 * S.blkptr[k] + 256 * k;
    D.tmplen = 2 * S.blklen() + 256 * S.nsn;
    std::vector<int32_t> lev2;
    std::vector<int64_t> lev2off;
    classify_levels(S, [](int64_t) { return true; }, c->lvl, lev2, lev2off);
    int rc = 0;
    {
      // slots of the large fronts (per-front 64 x 64 scratch) and the flat list of LDS-class cliques
      int32_t slot = 0;
      std::vector<int32_t> lev3, large;
      c->lev_namax.assign(S.nlev, 0);
      c->fam.assign(S.nsn, 0);
      for (int64_t l = 0; l < S.nlev; ++l) {
        const LevelClass& L = c->lvl[l];
        int64_t b = S.levptr[l];
        for (int64_t q = L.nI - L.nS; q < L.nI; ++q) c->fam[lev2[b + q]] = l ? 2 : 1;
        for (int64_t q = 0; q < L.nI; ++q) lev3.push_back(lev2[b + q]);
        for (int64_t q = L.nI; q < L.nI + L.nII; ++q) { cl[lev2[b + q]].pad = slot++; large.push_back(lev2[b + q]); }
        c->lev_namax[l] = std::max(L.namaxI, L.namaxII);
        if (L.nII) { D.nnmaxII_all = std::max(D.nnmaxII_all, L.nnmaxII); D.namaxII_all = std::max(D.namaxII_all, L.namaxII); }
      }
      c->large_mask.assign((size_t)S.nsn, 0);
      for (int32_t k : large) c->large_mask[(size_t)k] = 1;
      D.nI_total = (int64_t)lev3.size();
      D.nII_total = (int64_t)large.size();
      {
        std::vector<int32_t> sp;       // childless large fronts the sparse-input sweep can take (front_lfsp.hip)
        for (int32_t k : large)
          if (S.chptr[k + 1] == S.chptr[k] && S.nn(k) <= 64 && S.na(k) <= 128 && S.na(k) > 0) sp.push_back(k);
        D.lfsp_cnt = (int64_t)sp.size();
        D.lfsp_exact = !sp.empty();
        for (int32_t k : sp) if (S.nn(k) != 64 || S.na(k) != 128) D.lfsp_exact = false;
        if (!sp.empty() && (rc = dev_upload(&D.lfsp_list, sp, D.bytes))) return rc;
      }
      {
        // sibling groups for the sparse-input sweep (front_lfsp.hip, k_lfsp_up<..., GRP>): members of a large-front class all
        // of whose fronts that sweep can take (childless, nn <= 64, 0 < na <= 128), under one LARGE parent, with identical
        // relative indices; at most eight per group, in list order.  SMCP_LFSP_GROUP=0: none.
        const char* ge = sw_str("SMCP_LFSP_GROUP");
        const bool gon = !(ge && ge[0] == '0');
        std::vector<uint8_t> is_large((size_t)S.nsn, 0), skip((size_t)S.nsn, 0);
        for (int32_t k : large) is_large[(size_t)k] = 1;
        c->lfsp_grp.assign((size_t)S.nlev, csp_ctx::LfspGroups());
        c->lfsp_any_groups = false;
        for (int64_t l = 0; l < S.nlev && gon; ++l) {
          const LevelClass& L = c->lvl[l];
          if (!L.nII || L.nchmaxII != 0 || L.nnmaxII > 64 || L.namaxII > 128) continue;
          const int64_t b = S.levptr[l] + L.nI;
          bool ok = true;
          for (int64_t q = 0; q < L.nII; ++q) ok = ok && S.na(lev2[b + q]) > 0;
          if (!ok) continue;
          std::vector<int32_t> gptr(1, 0), glist;
          std::vector<uint8_t> taken((size_t)L.nII, 0);
          bool shared = false;
          // the members of the class by parent, in list order: a front is only ever compared with its own siblings
          std::unordered_map<int64_t, std::vector<int64_t>> sibs;
          for (int64_t q = 0; q < L.nII; ++q) sibs[S.snpar[lev2[b + q]]].push_back(q);
          for (int64_t q = 0; q < L.nII; ++q) {
            if (taken[(size_t)q]) continue;
            const int32_t k = lev2[b + q];
            taken[(size_t)q] = 1;
            glist.push_back(k);
            int size = 1;
            const int64_t par = S.snpar[k];
            if (par >= 0 && is_large[(size_t)par]) {
              const std::vector<int64_t>& sb = sibs[par];
              for (auto it = std::upper_bound(sb.begin(), sb.end(), q); it != sb.end() && size < 8; ++it) {
                const int64_t q2 = *it;
                const int32_t k2 = lev2[b + q2];
                if (taken[(size_t)q2] || S.na(k2) != S.na(k)) continue;
                if (!std::equal(S.relidx.begin() + S.sepptr[k], S.relidx.begin() + S.sepptr[k + 1], S.relidx.begin() + S.sepptr[k2])) continue;
                taken[(size_t)q2] = 1;
                glist.push_back(k2);
                ++size;
              }
            }
            gptr.push_back((int32_t)glist.size());
            if (size > 1) shared = true;
          }
          if (!shared) continue;
          const int ng = (int)gptr.size() - 1;
          for (int g = 0; g < ng; ++g) {
            const int sz = gptr[(size_t)g + 1] - gptr[(size_t)g];
            for (int q = 0; q < sz; ++q)
              if (q != g % sz) skip[(size_t)glist[(size_t)gptr[(size_t)g] + q]] = 1;
          }
          csp_ctx::LfspGroups& G = c->lfsp_grp[(size_t)l];
          if ((rc = dev_upload(&G.ptr, gptr, D.bytes))) return rc;
          if ((rc = dev_upload(&G.list, glist, D.bytes))) return rc;
          G.ngroups = ng;
          c->lfsp_any_groups = true;
        }
        if (c->lfsp_any_groups && (rc = dev_upload(&D.lfsp_skip, skip, D.bytes))) return rc;
        // sibling groups of FAMILY PARENTS for the entry-driven family sweep (front_famt.hip, k_fam_terms_grp): family parents of a
        // level under one LARGE front with identical relative indices, at most FAMT_GMAX per group, in list order; the lists
        // hold positions in the level's family list (= record indices of k_famt_prep).  Every family parent of the level is in
        // exactly one group (singletons included).  SMCP_FAMT_GROUP=0: none.
        const char* fe = sw_str("SMCP_FAMT_GROUP");
        const bool fon = !(fe && fe[0] == '0');
        std::vector<uint8_t> fskip((size_t)S.nsn, 0);
        c->famt_grp.assign((size_t)S.nlev, csp_ctx::LfspGroups());
        c->famt_any_groups = false;
        for (int64_t l = 1; l < S.nlev && fon; ++l) {
          const LevelClass& L = c->lvl[l];
          if (!L.nS) continue;
          const int nat = std::max(1, (L.famna + 15) / 16);
          if (nat > 4 || !famt_grp_fits(nat, std::max(1, L.famcnn))) continue;
          const int64_t b = S.levptr[l] + (L.nI - L.nS);
          std::vector<int32_t> gptr(1, 0), glist;
          std::vector<uint8_t> taken((size_t)L.nS, 0);
          bool shared = false;
          std::unordered_map<int64_t, std::vector<int64_t>> sibs;
          for (int64_t q = 0; q < L.nS; ++q) sibs[S.snpar[lev2[b + q]]].push_back(q);
          for (int64_t q = 0; q < L.nS; ++q) {
            if (taken[(size_t)q]) continue;
            const int32_t k = lev2[b + q];
            taken[(size_t)q] = 1;
            glist.push_back((int32_t)q);
            int size = 1;
            const int64_t par = S.snpar[k];
            if (par >= 0 && is_large[(size_t)par]) {
              const std::vector<int64_t>& sb = sibs[par];
              for (auto it = std::upper_bound(sb.begin(), sb.end(), q); it != sb.end() && size < FAMT_GMAX; ++it) {
                const int64_t q2 = *it;
                const int32_t k2 = lev2[b + q2];
                if (taken[(size_t)q2] || S.na(k2) != S.na(k)) continue;
                if (!std::equal(S.relidx.begin() + S.sepptr[k], S.relidx.begin() + S.sepptr[k + 1], S.relidx.begin() + S.sepptr[k2])) continue;
                taken[(size_t)q2] = 1;
                glist.push_back((int32_t)q2);
                fskip[(size_t)k2] = 1;
                ++size;
              }
            }
            gptr.push_back((int32_t)glist.size());
            if (size > 1) shared = true;
          }
          if (!shared) {
            for (int64_t q = 0; q < L.nS; ++q) fskip[(size_t)lev2[b + q]] = 0;
            continue;
          }
          csp_ctx::LfspGroups& G = c->famt_grp[(size_t)l];
          if ((rc = dev_upload(&G.ptr, gptr, D.bytes))) return rc;
          if ((rc = dev_upload(&G.list, glist, D.bytes))) return rc;
          G.ngroups = (int)gptr.size() - 1;
          c->famt_any_groups = true;
        }
        if (c->famt_any_groups) {
          if ((rc = dev_upload(&D.famt_skip, fskip, D.bytes))) return rc;
          std::vector<uint8_t> both(fskip);
          if (c->lfsp_any_groups) for (size_t i = 0; i < both.size(); ++i) both[i] |= skip[i];
          if ((rc = dev_upload(&D.both_skip, both, D.bytes))) return rc;
        }
      }
      lev3.insert(lev3.end(), large.begin(), large.end());
      if ((rc = dev_upload(&D.lev3idx, lev3, D.bytes))) return rc;
      D.lfd_len = (int64_t)(slot + 1) * 64 * 64;
      if ((rc = dev_alloc(&D.lfd, (int64_t)(slot + 1) * 64 * 64, D.bytes))) return rc;
      D.lfd_dense = D.lfd + (int64_t)slot * 64 * 64;
    }
    if ((rc = dev_upload(&D.cl, cl, D.bytes))) return rc;
    if ((rc = dev_upload(&D.rowidx, S.rowidx, D.bytes))) return rc;
    if ((rc = dev_upload(&D.relidx, S.relidx, D.bytes))) return rc;
    if ((rc = dev_upload(&D.chidx, ch, D.bytes))) return rc;
    if ((rc = dev_upload(&D.levidx, lev, D.bytes))) return rc;
    if ((rc = dev_upload(&D.lev2idx, lev2, D.bytes))) return rc;
    clk.mark("descriptors + uploads");
    // ---- gather plans for the extend-add
    if (S.updplen() < (int64_t)1 << 31) {
      std::vector<int64_t> tptr(S.nsn + 1, 0), cptr;
      std::vector<int32_t> tgt, src;
      const int64_t nsn1 = S.nsn / c->ntrial;       // a replicated pattern: plan of the first copy, shifted for the others
      // (target code, src offset) pairs of every clique with children, sorted per clique: the cliques are independent,
      // so host threads take them round-robin (5.5 M pairs on synth50k: 0.2 s of the set-up on one thread)
      std::vector<int64_t> par;
      for (int64_t k = 0; k < nsn1; ++k) if (S.chptr[k + 1] > S.chptr[k] || (cl[(size_t)k].pad >= 0 && S.na(k) > 0)) par.push_back(k);
      std::vector<std::vector<std::pair<int32_t, int32_t>>> prs(par.size());
      std::vector<int64_t> ntg(par.size(), 0);     // distinct targets per clique
      // Large fronts list EVERY position of the lower triangle of their update block as a target, with or without
      // contributions (marker pairs, dropped again when the sources are laid out): the plan kernels then assign the whole block
      // and the clear pass before them (k_lf_clear_upd: a 5 us launch in every chain over the top fronts) is not needed
      constexpr int32_t PLAN_MARK = INT32_MIN;
      std::vector<int64_t> nmark(par.size(), 0);
      {
        const unsigned hw = std::thread::hardware_concurrency();
        const int nth = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hw ? hw : 1), (int64_t)16, (int64_t)par.size() / 64 + 1}));
        auto work = [&](int tix) {
          for (size_t x = (size_t)tix; x < par.size(); x += (size_t)nth) {
            const int64_t k = par[x];
            auto& pr = prs[x];
            const int64_t nnp = S.nn(k);
            size_t tot = 0;
            for (int64_t q = S.chptr[k]; q < S.chptr[k + 1]; ++q) { const int64_t nac = S.na(S.chidx[q]); tot += (size_t)(nac * (nac + 1) / 2); }
            pr.reserve(tot);
            for (int64_t q = S.chptr[k]; q < S.chptr[k + 1]; ++q) {
              const int64_t cc = S.chidx[q], nac = S.na(cc);
              const int32_t* rel = &S.relidx[S.sepptr[cc]];
              for (int64_t j = 0; j < nac; ++j)
                for (int64_t i = j; i < nac; ++i) {
                  int32_t ri = rel[i], rj = rel[j];
                  int32_t code = rj < nnp ? (ri | (rj << 15)) : ((1 << 30) | (ri - (int32_t)nnp) | ((rj - (int32_t)nnp) << 15));
                  pr.emplace_back(code, (int32_t)(S.updpptr[cc] + j * nac - j * (j - 1) / 2 + (i - j)));
                }
            }
            if (cl[(size_t)k].pad >= 0) {
              const int64_t nak = S.na(k);
              for (int64_t j = 0; j < nak; ++j)
                for (int64_t i = j; i < nak; ++i) pr.emplace_back((int32_t)((1 << 30) | (int32_t)i | ((int32_t)j << 15)), PLAN_MARK);
              nmark[x] = nak * (nak + 1) / 2;
            }
            std::sort(pr.begin(), pr.end());
            int64_t nd = 0;
            for (size_t e = 0; e < pr.size(); ++e) if (e == 0 || pr[e].first != pr[e - 1].first) ++nd;
            ntg[x] = nd;
          }
        };
        run_threads(nth, work);
      }
      clk.mark("plan: sorted pairs");
      // targets (distinct codes) per clique were counted by the workers: the serial part only lays the pieces out
      std::vector<int64_t> tbase(par.size() + 1, 0), sbase(par.size() + 1, 0);
      for (size_t x = 0; x < par.size(); ++x) { tbase[x + 1] = tbase[x] + ntg[x]; sbase[x + 1] = sbase[x] + (int64_t)prs[x].size() - nmark[x]; }
      const int64_t nt1 = tbase[par.size()], ns1 = sbase[par.size()];
      tgt.resize((size_t)nt1);
      src.resize((size_t)ns1);
      cptr.assign((size_t)nt1 + 1, 0);
      {
        const int nth2 = (int)std::max<int64_t>(1, std::min<int64_t>(16, (int64_t)par.size() / 64 + 1));
        auto fill = [&](int tix) {
          for (size_t x = (size_t)tix; x < par.size(); x += (size_t)nth2) {
            const auto& pr = prs[x];
            int64_t tq = tbase[x];
            const int64_t s0 = sbase[x];
            int64_t w = 0;
            for (size_t e = 0; e < pr.size(); ++e) {
              if (e == 0 || pr[e].first != pr[e - 1].first) { tgt[(size_t)tq] = pr[e].first; cptr[(size_t)tq] = s0 + w; ++tq; }
              if (pr[e].second != PLAN_MARK) src[(size_t)(s0 + w++)] = pr[e].second;
            }
          }
        };
        run_threads(nth2, fill);
      }
      cptr[(size_t)nt1] = ns1;
      {
        size_t x = 0;
        for (int64_t k = 0; k < nsn1; ++k) {
          if (x < par.size() && par[x] == k) ++x;
          tptr[k + 1] = tbase[x];
        }
      }
      prs.clear();
      clk.mark("plan: merge");
      if (c->ntrial > 1) {
        // a replicated pattern: the copies' plans are the first one shifted -- laid out on the device by one kernel
        // (K = 8 on synth50k: 44 M indices; building them on the host and uploading 176 MB took 0.12 s)
        const int64_t K = c->ntrial, up1 = S.updplen() / K;
        for (int64_t t = 1; t < K; ++t)
          for (int64_t k = 0; k < nsn1; ++k) tptr[t * nsn1 + k + 1] = tptr[k + 1] + t * nt1;
        if ((rc = dev_upload(&D.gp_tptr, tptr, D.bytes))) return rc;
        if ((rc = dev_alloc(&D.gp_tgt, nt1 * K, D.bytes))) return rc;
        if ((rc = dev_alloc(&D.gp_cptr, nt1 * K + 1, D.bytes))) return rc;
        if ((rc = dev_alloc(&D.gp_src, ns1 * K, D.bytes))) return rc;
        if (nt1) HIPCHK(hipMemcpy(D.gp_tgt, tgt.data(), sizeof(int32_t) * nt1, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(D.gp_cptr, cptr.data(), sizeof(int64_t) * (nt1 + 1), hipMemcpyHostToDevice));
        if (ns1) HIPCHK(hipMemcpy(D.gp_src, src.data(), sizeof(int32_t) * ns1, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_replicate_plan, dim3(2048), dim3(256), 0, 0, D.gp_tgt, D.gp_cptr, D.gp_src, nt1, ns1, up1, K);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
        clk.mark("plan: replicate (device)");
      } else {
        if ((rc = dev_upload(&D.gp_tptr, tptr, D.bytes))) return rc;
        if ((rc = dev_upload(&D.gp_tgt, tgt, D.bytes))) return rc;
        if ((rc = dev_upload(&D.gp_cptr, cptr, D.bytes))) return rc;
        if ((rc = dev_upload(&D.gp_src, src, D.bytes))) return rc;
      }
    }
    c->plan_full_upd = true;
    clk.mark("plan: upload");
    if ((rc = dev_alloc(&D.lk, S.blklen(), D.bytes))) return rc;
    HIPCHK(hipMemset(D.lk, 0, sizeof(double) * std::max<int64_t>(S.blklen(), 1)));
    static bool attrs_done = false;     // function attributes are per process (one device per process): set them once
    if (!attrs_done) {
      attrs_done = true;
      const int mx = 160 * 1024 - 1024;
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_up_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_up_pad, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_up_n16<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_up_n16<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_up_n16<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_up_n16<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_up_n16<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_up_n16<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_up_n16<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_up_n16<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_partial<true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_partial<false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<3, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<5, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<5, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<6, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<7, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_gram_diag128<9, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_lf_diag, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_factor_yaa_lds, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_down_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_down_mfma<true, WK_DOWN0>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_chol_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_chol_mfma<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_pinv_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_down_inv_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_hess_up_inv_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_llt_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      HIPCHK(hipFuncSetAttribute((const void*)k_completion_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    }
    if ((rc = dev_upload(&D.tmpptr, c->h_tmpptr, D.bytes))) return rc;
    if ((rc = dev_alloc(&D.yaa, S.updlen(), D.bytes))) return rc;
    if ((rc = dev_alloc(&D.fac, S.updlen(), D.bytes))) return rc;
    HIPCHK(hipMemset(D.fac, 0, sizeof(double) * std::max<int64_t>(S.updlen(), 1)));     // strict upper triangles stay zero (prepare_yaa)
    if (!D.sw) {
      if ((rc = dev_alloc(&D.sw, S.blklen(), D.bytes))) return rc;
      hipLaunchKernelGGL(k_fill_sqrt_weights, dim3((unsigned)std::min<int64_t>(S.nsn, 4096)), dim3(256), 0, 0, D.cl, (int)S.nsn, D.sw);
    }
    if ((rc = dev_alloc(&D.faci, S.updlen(), D.bytes))) return rc;
    if ((rc = dev_alloc(&D.red, 4096, D.bytes))) return rc;      // [0, 1024): reduction scratch, [1024, 4096): shares of a split Amap (kkt_solve)
    if ((rc = dev_alloc(&D.info, 32, D.bytes))) return rc;      // [0, 16): failure flags of the copies; [16]: status latch
    HIPCHK(hipMemset(D.info, 0, sizeof(int) * 32));
    // pinned mirror: ints [0, 16) the trial flags (csp_trial_flags), [16] the status latch (csp_status), bytes [96, 104) the
    // scalar of the reductions (csp_dot / csp_logdiagsum): separate slots, so that no call overwrites another's result
    HIPCHK(hipHostMalloc((void**)&D.info_host, 128));
    clk.mark("attributes + buffers");
    { hipDeviceProp_t p; D.ncu = (hipGetDeviceProperties(&p, device) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
    D.device = device;
  } else {
    if (D.upd) { hipFree(D.upd); D.bytes -= D.max_rhs * S.updlen() * 8; D.upd = nullptr; }
    if (D.updp) { hipFree(D.updp); D.bytes -= D.max_rhs * D.updp_stride * 8; D.updp = nullptr; }
    if (D.tmp) { hipFree(D.tmp); D.bytes -= D.max_rhs * D.tmplen * 8; D.tmp = nullptr; }
  }
  int rc = 0;
  if ((rc = dev_alloc(&D.upd, max_rhs * S.updlen(), D.bytes))) return rc;
  { const char* e = sw_str("SMCP_UPDP_PAD"); D.updp_stride = S.updplen() + (e ? std::max(0, atoi(e)) : 0); }
  if ((rc = dev_alloc(&D.updp, max_rhs * D.updp_stride, D.bytes))) return rc;
  if ((rc = dev_alloc(&D.tmp, max_rhs * D.tmplen, D.bytes))) return rc;
  D.max_rhs = max_rhs;
  clk.mark("per-rhs workspaces");
  return 0;
}

int64_t csp_device_bytes(const csp_ctx* c) { return c ? c->D.bytes : 0; }

static int cholesky_impl(csp_ctx* c, double* x, void* stream, int set);
int csp_cholesky(csp_ctx* c, double* x, void* stream) { return cholesky_impl(c, x, stream, 0); }
int csp_cholesky_part(csp_ctx* c, double* x, int set, void* stream) {
  if (set < 1 || set > 2 || !c || !c->sets[set].lev2 || use_generic(c)) return SMCP_EINVAL;
  return cholesky_impl(c, x, stream, set);
}
static int cholesky_impl(csp_ctx* c, double* x, void* stream, int set) {
  if (int rc = ready(c)) return rc;
  invalidate_tags(c, x);
  hipStream_t st = (hipStream_t)stream;
  TreeArgs a = tree_args(c);
  HIPCHK(zero_flag(c, st));
  if (!use_generic(c)) {
    MfmaArgs a0 = mfma_args(c, nullptr, 0, 1);
    a0.LK = nullptr;
    for (int64_t l = 0; l < c->S.nlev; ++l)
      for_level_classes(c, l, a0, [&](bool lds, MfmaArgs am, int cnt, size_t bytes, int thr) {
        if (lds) launch_lds(c, KID_chol_mfma, k_chol_mfma<true>, dim3(cnt), dim3(fact_threads(am, thr, 0)),
                            (size_t)mfma_lds_doubles_for(WK_CHOL, am.nnmax, am.namax) * sizeof(double), st, am, x, (double*)nullptr);   // compact layout: front + update only
        else if (use_large() && c->D.gp_tptr) lf_chol(c, am, cnt, x, st);
        else launch_lds(c, KID_chol_mfma_hbm, k_chol_mfma<false>, dim3(cnt), dim3(thr), 0, st, am, x, (double*)nullptr);
      }, set);
  } else
  for_levels_up(c, [&](const int32_t* lev, int cnt) {
    a.lev = lev;
    launch(c, KID_chol_level, k_chol_level, dim3(cnt), dim3(NT), st, a, x);
  });
  HIPCHK(end_call(c));
  return fetch_info(c, st);
}

int csp_llt(csp_ctx* c, double* x, void* stream) {
  if (int rc = ready(c)) return rc;
  invalidate_tags(c, x);
  hipStream_t st = (hipStream_t)stream;
  TreeArgs a = tree_args(c);
  if (!use_generic(c)) {
    MfmaArgs a0 = mfma_args(c, nullptr, 0, 1);
    a0.LK = nullptr;
    dim3 blk(256);
    for (int64_t l = 0; l < c->S.nlev; ++l)
      for_level_classes(c, l, a0, [&](bool lds, MfmaArgs am, int cnt, size_t bytes, int thr) {
        if (lds) launch_lds(c, KID_llt_mfma, k_llt_mfma<true>, dim3(cnt), dim3(thr), bytes, st, am, x);
        else if (use_large()) {
          const int mtA = tiles64(am.namax), ntN = tiles64(am.nnmax);
          const int ntile = ntN * (ntN + 1) / 2 + mtA * ntN + mtA * (mtA + 1) / 2;
          launch(c, KID_lf_llt, k_lf_llt, dim3(umax1(ntile), cnt), blk, st, am, x, 0);
          launch(c, KID_lf_llt, k_lf_llt, dim3(umax1(ntile), cnt), blk, st, am, x, 1);
          lf_assemble(c, am, cnt, 1, x, 0, 2, st);
          if (am.namax) launch(c, KID_lf_pack_upd, k_lf_pack_upd, dim3(umax1(std::min(64, (am.namax * am.namax + 255) / 256)), cnt), blk, st, am);
        } else launch_lds(c, KID_llt_mfma_hbm, k_llt_mfma<false>, dim3(cnt), dim3(thr), 0, st, am, x);
      });
  } else
  for_levels_up(c, [&](const int32_t* lev, int cnt) {
    a.lev = lev;
    launch(c, KID_llt_level, k_llt_level, dim3(cnt), dim3(NT), st, a, x);
  });
  HIPCHK(end_call(c));
  return 0;
}

static int projected_inverse_impl(csp_ctx* c, double* x, void* stream, int set);
int csp_projected_inverse(csp_ctx* c, double* x, void* stream) { return projected_inverse_impl(c, x, stream, 0); }
int csp_projected_inverse_part(csp_ctx* c, double* x, int set, void* stream) {
  if (set < 1 || set > 2 || !c || !c->sets[set].lev2 || use_generic(c)) return SMCP_EINVAL;
  return projected_inverse_impl(c, x, stream, set);
}
static int projected_inverse_impl(csp_ctx* c, double* x, void* stream, int set) {
  if (int rc = ready(c)) return rc;
  invalidate_tags(c, x);
  hipStream_t st = (hipStream_t)stream;
  TreeArgs a = tree_args(c);
  if (!use_generic(c)) {
    if (set) { prep_lk_set(c, set, x, st); c->D.lk_tag_L = nullptr; c->D.lk_tag_Y = nullptr; }   // partial: no cache claim
    else {
    prep_lk(c, x, st);
    c->D.lk_tag_L = nullptr;   // x is about to be overwritten by Y: LK stays valid for the pair (L, Y = x)
    c->D.lk_tag_Y = x;
    }
    const bool whole = !set;
    MfmaArgs a0 = mfma_args(c, nullptr, 0, 1);
    // The root->leaves pass gathers the separator block Y_AA of every clique into the update workspace (each clique's
    // children read it from there).  With yaa AS that workspace the blocks are where the sweeps that follow look for
    // them: no second gather of all Y_AA (four launches, 72 us on synth50k) and no copy (35 us).
    const bool keep_yaa = !set && !cache_off() && c->D.yaa && c->S.updlen() > 0;
    if (keep_yaa) a0.t.upd = c->D.yaa;
    for (int64_t l = c->S.nlev - 1; l >= 0; --l)
      for_level_classes(c, l, a0, [&](bool lds, MfmaArgs am, int cnt, size_t bytes, int thr) {
        if (lds) launch_lds(c, KID_pinv_mfma, k_pinv_mfma<true>, dim3(cnt), dim3(fact_threads(am, thr, 1)),
                            (size_t)mfma_lds_doubles_for(WK_PINV, am.nnmax, am.namax) * sizeof(double), st, am, x);
        else if (use_large()) lf_pinv(c, am, cnt, x, st);
        else launch_lds(c, KID_pinv_mfma_hbm, k_pinv_mfma<false>, dim3(cnt), dim3(thr), 0, st, am, x);
      }, set);
    if (keep_yaa) {
      c->D.yaa_tag = x;
      c->D.fac_tag = c->D.faci_tag = nullptr;
      c->D.part_valid = false;
      fp_record(c, 2, x, st);
    }
    if (whole) fp_record(c, 1, x, st);
  } else
  for_levels_down(c, [&](const int32_t* lev, int cnt) {
    a.lev = lev;
    launch(c, KID_pinv_level, k_pinv_level, dim3(cnt), dim3(NT), st, a, x);
  });
  HIPCHK(end_call(c));
  return 0;
}

// ---- the dual scaling point in one call (solvers.py:881-891: L = cholesky(S); Y = projected_inverse(L)) ---------------
// The two calls always come as a pair, and as separate entry points they are one dependent chain of ~35 launches, most of
// them a handful of workgroups: the blocked factorisations and inversions of the top fronts.  In one entry point the
// clique-local stages that only need FINISHED levels run on side streams beside those chains:
//   * the inverse-form factor [Li; K] of every level below the last is prepared while the levels above are still being
//     factored (k_prep_lk of the small cliques and lf_prep of each large level beside the top fronts' Cholesky),
//   * Y <- L for everything below the last level is copied there too (the tail after the last level),
//   * with SCALING_FAC: chol(Y_AA) of level l (k_mid_chol / k_factor_yaa_lds, what the Schur sweeps need next:
//     CHOMPACK's factored updates) starts as soon as the root->leaves pass has left level l and runs beside the levels below.
// Results and cache tags are those of csp_cholesky(L); copy; csp_projected_inverse(Y) [; prepare_yaa(Y, fac)].
constexpr int SCALING_FAC = 1;
static bool leafgram_ok(csp_ctx* c, int64_t mcols);      // kkt.hip
static bool scaling_overlap_on() {
  static int on = -1;
  if (on < 0) { const char* e = sw_str("SMCP_SCALING_OVERLAP"); on = (e && e[0] == '0') ? 0 : 1; }
  return on == 1;
}
static int scaling_impl(csp_ctx* c, double* L, double* Y, int flags, hipStream_t st) {
  DeviceCtx& D = c->D;
  const Symbolic& S = c->S;
  const int64_t bl = S.blklen();
  // first level from which on there are no LDS-class cliques any more (the side work starts there) and the blkval offset
  // where the last level begins -- cliques are numbered in postorder, so normally the last level is the tail of blkval; when
  // a clique of a lower level lies behind it the whole copy waits for the factorisation (tail0 = 0).  Computed once.
  if (c->scal_lstar < 0) {
    int64_t ls = S.nlev;
    while (ls > 0 && c->lvl[(size_t)ls - 1].nI == 0) --ls;
    c->scal_lstar = ls;
    std::vector<int> lev_of((size_t)S.nsn, 0);
    for (int64_t l = 0; l < S.nlev; ++l)
      for (int64_t q = S.levptr[l]; q < S.levptr[l + 1]; ++q) lev_of[(size_t)S.levidx[q]] = (int)l;
    int64_t t0 = bl;
    for (int64_t k = 0; k < S.nsn; ++k) if (lev_of[(size_t)k] == S.nlev - 1) t0 = std::min<int64_t>(t0, S.blkptr[k]);
    for (int64_t k = 0; k < S.nsn; ++k) if (S.blkptr[k] >= t0 && lev_of[(size_t)k] != S.nlev - 1) { t0 = 0; break; }
    c->scal_tail0 = t0;
  }
  const int64_t lstar = c->scal_lstar;
  const bool fast = !use_generic(c) && use_large() && D.gp_tptr && !cache_off() && Fork::enabled() && scaling_overlap_on() &&
                    c->ntrial == 1 && lstar < S.nlev && (lstar > 0 || D.nI_total == 0) && D.yaa && S.updlen() > 0 && !c->verify_cache;
  if (!fast) {      // the three steps one after the other (any tree, any switch)
    if (int rc = cholesky_impl(c, L, (void*)st, 0)) return rc;
    HIPCHK(hipMemcpyAsync(Y, L, sizeof(double) * bl, hipMemcpyDeviceToDevice, st));
    if (int rc = projected_inverse_impl(c, Y, (void*)st, 0)) return rc;
    c->D.lk_tag_L = L;                      // L itself is intact: LK is its inverse form as well as the pair's
    if (flags & SCALING_FAC) { if (int rc = prepare_yaa(c, Y, true, st)) return rc; HIPCHK(end_call(c)); }
    return 0;
  }
  invalidate_tags(c, L);
  invalidate_tags(c, Y);
  HIPCHK(zero_flag(c, st));
  const int64_t last = S.nlev - 1;
  const int64_t tail0 = c->scal_tail0;
  MfmaArgs a0 = mfma_args(c, nullptr, 0, 1);
  a0.LK = nullptr;
  TreeArgs t = tree_args(c);
  // small cliques whose supernodes have at most 16 columns leave their inverse-form factor behind as they are factored
  // (k_chol_mfma<true, true>); only when some LDS class is wider does k_prep_lk run, on the side stream (SMCP_CHOL_PREP=0: always)
  int nnI = 0;
  for (const LevelClass& Lc : c->lvl) if (Lc.nI) nnI = std::max(nnI, (int)Lc.nnmaxI);
  static int cprep = -1;
  if (cprep < 0) { const char* e = sw_str("SMCP_CHOL_PREP"); cprep = (e && e[0] == '0') ? 0 : 1; }
  const bool fuse_prep = cprep && nnI <= 16;
  auto chol_level = [&](int64_t l) {
    for_level_classes(c, l, a0, [&](bool lds, MfmaArgs am, int cnt, size_t bytes, int thr) {
      (void)bytes;
      const size_t lb = (size_t)mfma_lds_doubles_for(WK_CHOL, am.nnmax, am.namax) * sizeof(double);
      if (lds && fuse_prep) launch_lds(c, KID_chol_mfma, k_chol_mfma<true, true>, dim3(cnt), dim3(fact_threads(am, thr, 0)), lb, st, am, L, D.lk);
      else if (lds) launch_lds(c, KID_chol_mfma, k_chol_mfma<true>, dim3(cnt), dim3(fact_threads(am, thr, 0)), lb, st, am, L, (double*)nullptr);
      else lf_chol(c, am, cnt, L, st);
    });
  };
  // ---- 1. cholesky, leaves -> root.  Every cross-stream dependency costs ~10 us of idle stream on this stack (rocprofv3
  // timeline: a gap behind each hipEventRecord / hipStreamWaitEvent), so there are exactly two hand-overs to ONE side stream:
  // before the LAST level's chain (one workgroup for 110 us on synth50k) the side stream takes the inverse-form factor of
  // everything below it and the head of the copy into Y; after the root->leaves pass has left the large-only levels it
  // takes the Cholesky factors of their separator blocks.  The side work is queued AFTER the chain it runs beside, so that
  // the chain's few workgroups are dispatched first.
  for (int64_t l = 0; l < last; ++l) chol_level(l);
  std::unique_ptr<Fork> f0;
  if (last > 0) f0.reset(new Fork(c, st, 1));      // (the mark: the filler stream starts behind levels 0 .. last - 1)
  hipStream_t s0 = f0 ? f0->s : st;
  chol_level(last);
  if (D.nI_total && !fuse_prep) {
    TreeArgs ts = t;
    ts.lev = D.lev3idx;
    launch_lds(c, KID_prep_lk, k_prep_lk, dim3((int)D.nI_total), dim3(prep_lk_threads(nnI)), prep_lk_lds_bytes(nnI), s0, ts, (const double*)L, D.lk);
  }
  for (int64_t ll = 0; ll < last; ++ll)
    for_level_classes(c, ll, a0, [&](bool lds, MfmaArgs am, int cnt, size_t, int) { if (!lds) lf_prep(c, am, cnt, L, s0); });
  if (tail0 > 0) (void)hipMemcpyAsync(Y, L, sizeof(double) * tail0, hipMemcpyDeviceToDevice, s0);
  // ---- 2. the last level's inverse form and the tail of Y on the caller's stream, then the side branch is joined
  for_level_classes(c, last, a0, [&](bool lds, MfmaArgs am, int cnt, size_t, int) { if (!lds) lf_prep(c, am, cnt, L, st); });
  (void)hipMemcpyAsync(Y + tail0, L + tail0, sizeof(double) * (bl - tail0), hipMemcpyDeviceToDevice, st);
  if (f0) f0->join();
  D.lk_gen++;
  D.part_valid = false;
  // ---- 3. projected inverse, root -> leaves, the separator blocks kept in yaa; chol(Y_AA) of a level beside the levels below
  MfmaArgs ap = mfma_args(c, nullptr, 0, 1);
  ap.t.upd = D.yaa;
  MfmaArgs af = mfma_args(c, nullptr, 0, 1);
  const bool want_fac = (flags & SCALING_FAC) != 0;
  // the family children's factors are read by nobody when their block of the Schur complement comes from k_leaf_pairs (the
  // decision kkt_schur_* will take for the constraints now set): left out, complete_fac supplies them to whoever asks
  static int skipenv = -1;
  if (skipenv < 0) { const char* e = sw_str("SMCP_FAC_PARTIAL"); skipenv = (e && e[0] == '0') ? 0 : 1; }
  const bool skip_children = want_fac && skipenv && D.m > 0 && !D.ns && leafgram_ok(c, D.m) && c->leafgram_policy != 0;
  bool skipped = false;
  auto factor_level = [&](int64_t l, hipStream_t lds_stream, hipStream_t large_stream) {
    for_level_classes(c, l, af, [&](bool lds, MfmaArgs am, int cnt, size_t, int) {
      if (!am.namax) return;
      if (lds) {
        if (l == 0 && skip_children && am.nS > 0) { cnt -= am.nS; skipped = true; if (!cnt) return; }
        const size_t bytes = ((size_t)padld(am.namax) * am.namax + 256 + 8) * sizeof(double);
        launch_lds(c, KID_factor_yaa_lds, k_factor_yaa_lds, dim3(cnt), dim3(fact_threads(am, 256, 2)), bytes, lds_stream, am, (const double*)D.yaa, D.fac);
      } else {
        launch(c, KID_axpby, k_copy_upd_blocks, dim3(cnt, umax1(std::min(64, (am.namax * am.namax + 1023) / 1024))), dim3(256), large_stream,
               am.t, (const double*)D.yaa, D.fac);
        lf_factor_yaa(c, am, cnt, D.fac, large_stream);
      }
    });
  };
  auto pinv_level = [&](int64_t l) {
    for_level_classes(c, l, ap, [&](bool lds, MfmaArgs am, int cnt, size_t bytes, int thr) {
      (void)bytes;
      if (lds) launch_lds(c, KID_pinv_mfma, k_pinv_mfma<true>, dim3(cnt), dim3(fact_threads(am, thr, 1)),
                          (size_t)mfma_lds_doubles_for(WK_PINV, am.nnmax, am.namax) * sizeof(double), st, am, Y);
      else lf_pinv(c, am, cnt, Y, st);
    });
  };
  // root -> leaves through the large-only levels (lstar .. last); their Y_AA blocks are then all in yaa and their factors go
  // to the side stream (second hand-over) while the pass goes on through the levels with small cliques; the small cliques'
  // own factors follow on the caller's stream (clique-local launches of thousands of workgroups: nothing left to hide behind)
  for (int64_t l = last; l >= lstar; --l) pinv_level(l);
  bool any_large_sep = false;
  for (int64_t l = lstar; l <= last; ++l) if (c->lvl[(size_t)l].nII && c->lvl[(size_t)l].namaxII > 0) any_large_sep = true;
  std::unique_ptr<Fork> f1;
  if (want_fac && any_large_sep && lstar > 0) f1.reset(new Fork(c, st, 0));
  hipStream_t s1 = f1 ? f1->s : st;
  for (int64_t l = lstar - 1; l >= 0; --l) pinv_level(l);
  if (want_fac) {
    for (int64_t l = last; l >= lstar; --l) factor_level(l, s1, s1);
    for (int64_t l = lstar - 1; l >= 0; --l) factor_level(l, st, st);
  }
  if (f1) f1->join();
  D.lk_tag_L = L; D.lk_tag_Y = Y;
  D.yaa_tag = Y;
  D.fac_tag = want_fac ? Y : nullptr;
  D.fac_partial = want_fac && skipped;
  D.faci_tag = nullptr;
  if (want_fac) D.fac_gen++;
  HIPCHK(end_call(c));
  const int rc = fetch_info(c, st);
  if (rc) {       // not positive definite (eager status): nothing derived from these buffers may be advertised as a valid cache
    D.lk_tag_L = D.lk_tag_Y = D.yaa_tag = D.fac_tag = D.faci_tag = nullptr;
    D.part_valid = false;
  }
  return rc;
}
int csp_cholesky_projected_inverse(csp_ctx* c, double* L, double* Y, int with_factors, void* stream) {
  if (int rc = ready(c)) return rc;
  if (!L || !Y || L == Y) return SMCP_EINVAL;
  return scaling_impl(c, L, Y, with_factors ? SCALING_FAC : 0, (hipStream_t)stream);
}

int csp_completion(csp_ctx* c, double* x, void* stream) {
  if (int rc = ready(c)) return rc;
  invalidate_tags(c, x);
  hipStream_t st = (hipStream_t)stream;
  TreeArgs a = tree_args(c);
  HIPCHK(zero_flag(c, st));
  if (!use_generic(c)) {
    // clique-local given chol(X_AA) and its inverse of every clique (taken from the input before it is overwritten)
    prepare_yaa(c, x, true, st, true);
    MfmaArgs a0 = mfma_args(c, c->D.faci, 0, 1);
    a0.LK = nullptr;
    dim3 blk(256);
    for (int64_t l = 0; l < c->S.nlev; ++l)
      for_level_classes(c, l, a0, [&](bool lds, MfmaArgs am, int cnt, size_t bytes, int thr) {
        // compact layout (no inverse-form factor, no update block; the inversion scratch only for wide supernodes)
        const size_t cb = lds ? (size_t)completion_lds_doubles(am.nnmax, am.namax) * sizeof(double) : 0;
        if (lds && cb <= LDS_LIMIT) {
          launch_lds(c, KID_completion_mfma, k_completion_mfma<true>, dim3(cnt), dim3(cb > 48 * 1024 ? 512 : 256), cb, st, am, x);
        }
        else if (lds || !use_large())
          launch_lds(c, KID_completion_mfma_hbm, k_completion_mfma<false>, dim3(cnt), dim3(lds ? 256 : thr), 0, st, am, x);
      });
    if (use_large())
      for_all_large(c, a0, [&](MfmaArgs am, int cnt) {
        const int mtA = tiles64(am.namax), ntN = tiles64(am.nnmax);
        if (am.namax) {
          launch(c, KID_lf_completion, k_lf_completion, dim3(umax1(mtA * ntN), cnt), blk, st, am, x, 0);
          launch(c, KID_lf_completion, k_lf_completion, dim3(umax1(mtA * ntN), cnt), blk, st, am, x, 1);
        }
        launch(c, KID_lf_completion, k_lf_completion, dim3(umax1(ntN * ntN), cnt), blk, st, am, x, 2);
        // Cholesky of the nn x nn matrix at the head of the scratch: fronts of the one-workgroup class in ONE launch (k_mid_chol,
        // mode 3) instead of three per 64 columns -- the chain of a completion on synth50k (tops + root: 27 launches) loses 11
        if (use_mid(am.nnmax))
          launch_lds(c, KID_mid_chol, k_mid_chol, dim3(cnt), dim3(1024), mid_chol_lds(am.nnmax), st, am, (double*)nullptr, (double*)nullptr, 3);
        else
        for (int jb = 0; jb < am.nnmax; jb += LB) {
          launch_lds(c, KID_lf_diag, k_lf_diag, dim3(cnt), diag_blk(), LF_DIAG_LDS, st, am, (double*)nullptr, (double*)nullptr, 3, jb, 1);
          const int mrem = am.nnmax - jb - 1;
          if (mrem > 0) {
            launch(c, KID_lf_chol_panel, k_lf_chol_panel, dim3(umax1(tiles64(mrem)), cnt), blk, st, am, (double*)nullptr, (double*)nullptr, 3, jb);
            const int mt = tiles64(mrem);
            launch(c, KID_lf_chol_trail, k_lf_chol_trail, dim3(umax1(mt * (mt + 1) / 2), cnt), blk, st, am, (double*)nullptr, (double*)nullptr, 3, jb);
          }
        }
        for (int ib = 0; ib < am.nnmax; ib += LB) {
          launch_lds(c, KID_lf_diag, k_lf_diag, dim3(cnt), diag_blk(), LF_DIAG_LDS, st, am, (double*)nullptr, (double*)nullptr, 3, ib, 0);
          if (ib > 0) launch(c, KID_lf_prep_s, k_lf_prep_s, dim3(umax1(tiles64(ib)), cnt), blk, st, am, (const double*)nullptr, (double*)nullptr, ib, 3);
          launch(c, KID_lf_prep_row, k_lf_prep_row, dim3(umax1(tiles64(ib) + 1), cnt), blk, st, am, (const double*)nullptr, (double*)nullptr, ib, 3, 0);
        }
        launch(c, KID_lf_completion, k_lf_completion, dim3(umax1(ntN * ntN + mtA * ntN), cnt), blk, st, am, x, 3);
      });
    invalidate_tags(c, x);   // x now holds the factor, not the matrix the caches were derived from
  } else {
    gather_all(c, x, 0, 1, c->D.upd, st);
    launch(c, KID_completion_all, k_completion_all, dim3((int)c->S.nsn), dim3(NT), st, a, x);
  }
  HIPCHK(end_call(c));
  return fetch_info(c, st);
}

int csp_hessian(csp_ctx* c, const double* L, const double* Y, double* U, int64_t nrhs, int64_t ldu,
                int adj, int inv, void* stream) {
  if (int rc = ready(c)) return rc;
  if (nrhs < 1 || adj < 0 || adj > 2 || (nrhs > 1 && ldu < c->S.blklen())) return SMCP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  bool need_fac = !(adj == 2 && inv == 0);
  invalidate_tags(c, U);
  HIPCHK(zero_flag(c, st));
  const bool refactor = need_fac && (cache_off() || c->D.fac_tag != Y);
  prepare_yaa(c, Y, need_fac, st, inv && !use_generic(c));
  if (!inv && !use_generic(c)) { if (int rc = prep_lk_cached(c, L, Y, st)) return rc; }
  for (int64_t r0 = 0; r0 < nrhs; r0 += c->D.max_rhs) {
    int64_t nr = std::min(c->D.max_rhs, nrhs - r0);
    hessian_impl(c, L, U + r0 * ldu, nr, ldu, adj, inv, st);
  }
  HIPCHK(end_call(c));
  if (refactor) {
    int rc = fetch_info(c, st);
    if (rc) c->D.fac_tag = c->D.faci_tag = nullptr;
    return rc;
  }
  return 0;
}

// Y: the matrix whose pair (L, Y) the cached inverse-form factor may belong to (null: only a factor cached for L itself
// is reused, otherwise it is formed from L)
int trsm_impl(csp_ctx* c, const double* L, const double* Y, double* B, int64_t nrhs, int64_t ldb, int trans, hipStream_t st) {
  static int mm = -1;
  if (mm < 0) { const char* e = sw_str("SMCP_TRSM_MM"); mm = (e && e[0] == '0') ? 0 : 1; }
  if (mm && !use_generic(c) && use_large() && nrhs >= 8) {
    // tile products with the inverse-form factor (front_large.hip: k_trsm_mm_*): the generic kernels below solve every
    // clique's triangle by substitution in one workgroup per sixteen columns -- 0.48 ms per level on config 4
    if (int rc = prep_lk_cached(c, L, Y, st)) return rc;
    const int64_t need = ldb * nrhs;
    DeviceCtx& D = c->D;
    if (D.trsm_x_len < need) {
      if (D.trsm_x) { HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipFree(D.trsm_x)); D.bytes -= D.trsm_x_len * 8; D.trsm_x = nullptr; D.trsm_x_len = 0; }
      if (int rc = dev_alloc(&D.trsm_x, need, D.bytes)) return rc;
      D.trsm_x_len = need;
    }
    MfmaArgs a0 = mfma_args(c, nullptr, 0, (int)nrhs);
    const unsigned ct = (unsigned)tiles64((int)nrhs);
    auto level = [&](int64_t l) {
      const LevelClass& Lc = c->lvl[l];
      const int cnt = (int)(Lc.nI + Lc.nII);
      if (!cnt) return;
      MfmaArgs a = a0;
      a.t.lev = c->D.lev2idx + c->S.levptr[l];
      const int nnm = std::max(Lc.nnmaxI, Lc.nnmaxII), nfm = std::max(Lc.nnmaxI + Lc.namaxI, Lc.nnmaxII + Lc.namaxII);
      if (!trans) launch(c, KID_trsm_fwd_level, k_trsm_mm_fwd, dim3(umax1(tiles64(nfm)), cnt, ct), dim3(256), st, a, B, (int)nrhs, ldb, (const int32_t*)c->D.rowidx, D.trsm_x);
      else launch(c, KID_trsm_bwd_level, k_trsm_mm_bwd, dim3(umax1(tiles64(nnm)), cnt, ct), dim3(256), st, a, B, (int)nrhs, ldb, (const int32_t*)c->D.rowidx, D.trsm_x);
      launch(c, KID_axpby, k_trsm_mm_copy, dim3((unsigned)std::min<int64_t>(64, ((int64_t)nnm * nrhs + 255) / 256), cnt), dim3(256), st, a, B, (int)nrhs, ldb, (const double*)D.trsm_x);
    };
    if (!trans) for (int64_t l = 0; l < c->S.nlev; ++l) level(l);
    else for (int64_t l = c->S.nlev - 1; l >= 0; --l) level(l);
    HIPCHK(end_call(c));
    return 0;
  }
  if (c->S.sepptr[c->S.nsn] * nrhs > c->D.max_rhs * c->D.tmplen) return SMCP_ENOMEM;
  TreeArgs a = tree_args(c);
  if (!trans) {
    for_levels_up(c, [&](const int32_t* lev, int cnt) {
      a.lev = lev;
      launch(c, KID_trsm_fwd_level, k_trsm_fwd_level, dim3(cnt, (unsigned)((nrhs + TRSM_CB - 1) / TRSM_CB)), dim3(NT), st, a, L, B, (int)nrhs, ldb, c->D.rowidx);
    });
  } else {
    for_levels_down(c, [&](const int32_t* lev, int cnt) {
      a.lev = lev;
      launch(c, KID_trsm_bwd_level, k_trsm_bwd_level, dim3(cnt, (unsigned)((nrhs + TRSM_CB - 1) / TRSM_CB)), dim3(NT), st, a, L, B, (int)nrhs, ldb, c->D.rowidx);
    });
  }
  HIPCHK(end_call(c));
  return 0;
}
int csp_trsm(csp_ctx* c, const double* L, double* B, int64_t nrhs, int64_t ldb, int trans, void* stream) {
  if (int rc = ready(c)) return rc;
  if (nrhs < 1 || ldb < c->S.n) return SMCP_EINVAL;
  return trsm_impl(c, L, nullptr, B, nrhs, ldb, trans, (hipStream_t)stream);
}

static int reduce_impl(csp_ctx* c, const double* X, const double* Y, int mode, double* out, hipStream_t st) {
  int nb = 512;
  if (c->D.sw && !use_generic(c)) {
    launch(c, KID_reduce_cliques, k_reduce_flat, dim3(nb), dim3(NT), st, c->S.blklen(), (const double*)c->D.sw, X, Y, mode, c->D.red);
  } else {
    nb = (int)std::min<int64_t>(c->S.nsn, 512);
    launch(c, KID_reduce_cliques, k_reduce_cliques, dim3(nb), dim3(NT), st, c->D.cl, (int)c->S.nsn, X, Y, mode, c->D.red);
  }
  launch(c, KID_reduce_final, k_reduce_final, dim3(1), dim3(NT), st, c->D.red, nb, c->D.red + 512);
  HIPCHK(end_call(c));
  double* h = (double*)(c->D.info_host + 24);
  HIPCHK(hipMemcpyAsync(h, c->D.red + 512, sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *out = *h;
  return 0;
}
int csp_dot(csp_ctx* c, const double* X, const double* Y, double* out, void* stream) {
  if (int rc = ready(c)) return rc;
  return reduce_impl(c, X, Y, 0, out, (hipStream_t)stream);
}
int csp_logdiagsum(csp_ctx* c, const double* X, double* out, void* stream) {
  if (int rc = ready(c)) return rc;
  return reduce_impl(c, X, nullptr, 1, out, (hipStream_t)stream);
}

int csp_axpby(int64_t len, double a, const double* x, double b, double* y, void* stream) {
  if (len < 0 || !y) return SMCP_EINVAL;
  if (len == 0) return 0;
  int nb = (int)std::min<int64_t>((len + NT - 1) / NT, 2048);
  hipLaunchKernelGGL(k_axpby, dim3(nb), dim3(NT), 0, (hipStream_t)stream, len, a, x, b, y);
  HIPCHK(hipGetLastError());
  return 0;
}


int csp_cache_reset(csp_ctx* c) {
  if (!c) return SMCP_EINVAL;
  c->D.lk_tag_L = c->D.lk_tag_Y = c->D.yaa_tag = c->D.fac_tag = c->D.faci_tag = nullptr;
  return 0;
}

int csp_touch(csp_ctx* c, const void* p) {
  if (!c) return SMCP_EINVAL;
  invalidate_tags(c, p);
  return 0;
}

// ---- placement of the packed exchange buffer (CSP_TUNE_PLACEMENT) ------------------------------------------------
// The family sweep of the Schur complement (k_fam_terms) writes, per (family parent, constraint), the parent's panel into the
// constraint stack and its packed update into the exchange buffer -- ~2000 concurrent store streams over two multi-GB
// buffers.  How fast the memory system takes that pattern depends on where the two buffers physically lie: 0.80 to 0.99 ms for
// the same kernel on the same data from one pair of allocations to the next, and a kernel that issues
// only the stores, without the arithmetic, shows the same spread (0.37 to 0.49 ms).  That kernel is the probe here: the
// exchange buffer is moved to fresh allocations, up to `tries` times, and the fastest placement is kept.
__global__ void __launch_bounds__(512) k_probe_family_stores(TreeArgs t, const int32_t* parents, int nrhs, double* ustack, int64_t bl) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const CliqueDesc d = t.cl[parents[blockIdx.x]];
  const int npan = (d.nn + d.na) * d.nn, npk = d.na * (d.na + 1) / 2;
  for (int r = blockIdx.y + gridDim.y * wave; r < nrhs; r += gridDim.y * 8) {
    double* P = ustack + (int64_t)r * bl + d.blk;
    double* U = t.updp + (int64_t)r * t.updplen + d.updp;
    for (int e = lane; e < npan; e += 64) P[e] = 0.0;           // (zeros: the stack's never-written entries must stay finite)
    for (int e = lane; e < npk; e += 64) U[e] = 0.0;
  }
}
static int tune_placement(csp_ctx* c, int tries) {
  if (int rc = ready(c)) return rc;
  DeviceCtx& D = c->D;
  if (!D.ustack || !D.updp || tries < 1) return SMCP_EINVAL;
  std::vector<int32_t> par;
  for (int64_t k = 0; k < c->S.nsn; ++k)
    if (k < (int64_t)c->fam.size() && c->fam[(size_t)k] == 2) par.push_back((int32_t)k);
  if (par.size() < 64) return 0;               // no family sweep worth tuning for
  D.qr_valid = false;                          // the probe overwrites panels of the stack (Q of kkt_qr lives there)
  D.lg_nochild = false;
  int32_t* dpar = nullptr;
  int64_t junk = 0;
  if (int rc = dev_upload(&dpar, par, junk)) return rc;
  HIPCHK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  const int nrhs = (int)std::min<int64_t>(D.max_rhs, D.ustack_cols);
  auto probe = [&](float& ms) -> int {
    TreeArgs t = tree_args(c);
    const dim3 grid((unsigned)par.size(), 2);
    hipLaunchKernelGGL(k_probe_family_stores, grid, dim3(512), 0, 0, t, (const int32_t*)dpar, nrhs, D.ustack, c->S.blklen());
    HIPCHK(hipEventRecord(e0, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_probe_family_stores, grid, dim3(512), 0, 0, t, (const int32_t*)dpar, nrhs, D.ustack, c->S.blklen());
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    return 0;
  };
  float best = 0.f;
  int rc = probe(best);
  const float first = best;
  std::vector<void*> rejected;                 // kept until the end: a freed buffer would be handed out again at once
  // The probe stores into BOTH buffers -- the packed updates and the panels of the swept stack -- and either can lie badly
  // (probe levels seen: 0.38 both well placed, 0.42-0.44 one of them, 0.50 neither): the tries alternate between them, the
  // stack only while the buffers set aside stay under 16 GB.  The CONTENTS of both buffers are NOT preserved: the probe
  // zeroes the family parents' panels in the stack and their slots in the exchange buffer, so whatever was derived from the
  // stack (the Q factor of kkt_qr, the "children's panels left out" mark of the last sweep) is dropped here -- tune between
  // Newton steps, not between a sweep and its consumer.
  const size_t ubytes = sizeof(double) * (size_t)(D.max_rhs * D.updp_stride);
  const size_t sbytes = sizeof(double) * (size_t)(D.ustack_cols * c->S.blklen());
  size_t held = 0;
  for (int q = 0; q < tries && !rc; ++q) {
    const bool stack = (q & 1) && held + sbytes <= ((size_t)16 << 30);
    double*& slot = stack ? D.ustack : D.updp;
    const size_t bytes = stack ? sbytes : ubytes;
    double* old = slot;
    double* nu = nullptr;
    if (hipMalloc((void**)&nu, bytes) != hipSuccess) { (void)hipGetLastError(); break; }   // out of memory: keep what we have (and leave no sticky error behind)
    // (the copy keeps the never-written entries of the stack finite, which is all the sweeps ask of it)
    if (stack && hipMemcpy(nu, old, bytes, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipFree(nu); rc = SMCP_EHIP; break; }
    slot = nu;
    float ms = 0.f;
    rc = probe(ms);
    if (!rc && ms < 0.985f * best) { best = ms; rejected.push_back(old); }
    else { slot = old; rejected.push_back(nu); }
    held += bytes;
    if (best <= 0.80f * first) break;          // from the slow end of the spread to the fast one: good enough
  }
  for (void* p : rejected) (void)hipFree(p);
  (void)hipFree(dpar);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  static int verbose = -1;
  if (verbose < 0) { const char* e = sw_str("SMCP_TIMING"); verbose = (e && e[0] == '1') ? 1 : 0; }
  if (verbose) fprintf(stderr, "smcp_amd: placement of the exchange buffer: store-pattern probe %.3f -> %.3f ms\n", first, best);
  c->placement_probe[0] = first; c->placement_probe[1] = best;
  return rc;
}
// milliseconds of the store-pattern probe before / after the last CSP_TUNE_PLACEMENT (zeros: never run, or nothing to tune);
// out[2]: delay kernels injected so far in this process (CSP_TUNE_RACE)
int csp_tune_report(csp_ctx* c, double* out) {
  if (!c || !out) return SMCP_EINVAL;
  out[0] = c->placement_probe[0]; out[1] = c->placement_probe[1];
  out[2] = (double)race_inject().injected.load();
  return 0;
}

int csp_tune(csp_ctx* c, int what, int64_t value) {
  if (!c) return SMCP_EINVAL;
  switch (what) {
    case CSP_TUNE_PLACEMENT:
      return tune_placement(c, (int)std::min<int64_t>(value, 16));
    case CSP_TUNE_LEAFGRAM:
      if (value < 0 || value > 2) return SMCP_EINVAL;
      c->leafgram_policy = (int)value;
      return 0;
    case CSP_TUNE_VERIFY_CACHE:
      if (c->D.device < 0) return SMCP_ENODEV;
      if (value && !c->D.fp) {
        if (hipMalloc((void**)&c->D.fp, 8 * sizeof(unsigned long long)) != hipSuccess) return SMCP_ENOMEM;
        if (hipMalloc((void**)&c->D.fp_bad, sizeof(int)) != hipSuccess) return SMCP_ENOMEM;
        if (hipMemset(c->D.fp, 0, 8 * sizeof(unsigned long long)) != hipSuccess || hipMemset(c->D.fp_bad, 0, sizeof(int)) != hipSuccess) return SMCP_EHIP;
      }
      c->verify_cache = value != 0;
      // entries created while verification was off carry no fingerprint: start from an empty cache
      c->D.lk_tag_L = c->D.lk_tag_Y = c->D.yaa_tag = c->D.fac_tag = c->D.faci_tag = nullptr;
      return 0;
    case CSP_TUNE_DETERMINISTIC:
      c->deterministic = value != 0;
      return 0;
    case CSP_TUNE_RACE:
      if (value < 0) return SMCP_EINVAL;
      race_inject().max_us.store((value >> 32) > 0 ? (int)std::min<int64_t>(value >> 32, 5000) : 200);
      race_inject().seed((uint64_t)(value & 0xffffffffll));
      return 0;
    case CSP_TUNE_RACE_DROP_JOINS:
      race_inject().drop_joins.store(value != 0);
      if (!value && c->D.device >= 0) (void)hipDeviceSynchronize();      // whatever ran unjoined is over before the next (correct) call
      return 0;
  }
  return SMCP_EINVAL;
}

int csp_profile_enable(csp_ctx* c, int on) {
  if (!c) return SMCP_EINVAL;
  c->prof.on = on != 0;
  return 0;
}

int csp_profile_filter(csp_ctx* c, int kid) {
  if (!c || kid >= KID_COUNT) return SMCP_EINVAL;
  c->prof.filter = kid < 0 ? -1 : kid;
  return 0;
}

int64_t csp_profile_kinds(void) { return KID_COUNT; }

int64_t csp_profile_read(csp_ctx* c, double* ms, int64_t* count) {
  if (int rc = ready(c)) return rc;
  Profiler& P = c->prof;
  HIPCHK(hipDeviceSynchronize());
  if (ms) for (int i = 0; i < KID_COUNT; ++i) ms[i] = 0.0;
  if (count) for (int i = 0; i < KID_COUNT; ++i) count[i] = 0;
  for (size_t i = 0; i < P.kids.size(); ++i) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, P.ev[2 * i], P.ev[2 * i + 1]));
    if (ms) ms[P.kids[i]] += t;
    if (count) count[P.kids[i]]++;
  }
  P.kids.clear();
  P.used = 0;
  return KID_COUNT;
}

// placement studies (scratch/famt_realloc2.py): move one of the big work buffers to a fresh allocation -- 0 the packed
// exchange buffer (updp), 1 the constraint stack (ustack); `shift` bytes are allocated first and freed afterwards so that
// the new buffer lands elsewhere.  Not part of the documented boundary.
// store pattern of the family sweep without its arithmetic: one wave per (slot, right-hand side), eight right-hand sides in
// flight per workgroup, a 9.5 KB run into the stack and a 16.6 KB run into the exchange buffer per pair
__global__ void __launch_bounds__(512) k_probe_streams(double* ustack, int64_t bl, double* updp, int64_t ustride, int nslots, int nrhs,
                                                        int64_t blk0, int64_t upd0) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = blockIdx.x;
  for (int r = blockIdx.y + gridDim.y * wave; r < nrhs; r += gridDim.y * 8) {
    double* P = ustack + (int64_t)r * bl + blk0 + (int64_t)p * 1185;
    double* U = updp + (int64_t)r * ustride + upd0 + (int64_t)p * 2080;
    for (int e = lane; e < 1185; e += 64) P[e] = 1.0;
    for (int e = lane; e < 2080; e += 64) U[e] = 1.0;
  }
}
int csp_debug_realloc(csp_ctx* c, int which, int64_t shift) {
  if (int rc = ready(c)) return rc;
  HIPCHK(hipDeviceSynchronize());
  DeviceCtx& D = c->D;
  if (which == 20) {      // probe: the store pattern of the family sweep on the buffers as they lie (synth50k geometry)
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    const int nrhs = (int)std::min<int64_t>(D.max_rhs, D.ustack_cols);
    const int64_t blk0 = 7168 * 180, upd0 = (int64_t)7168 * 496;
    if ((int64_t)896 * 1185 + blk0 > c->S.blklen() || (int64_t)896 * 2080 + upd0 > c->S.updplen()) return SMCP_EINVAL;
    hipLaunchKernelGGL(k_probe_streams, dim3(896, 2), dim3(512), 0, 0, D.ustack, c->S.blklen(), D.updp, D.updp_stride, 896, nrhs, blk0, upd0);
    HIPCHK(hipEventRecord(e0, 0));
    for (int r = 0; r < 3; ++r)
      hipLaunchKernelGGL(k_probe_streams, dim3(896, 2), dim3(512), 0, 0, D.ustack, c->S.blklen(), D.updp, D.updp_stride, 896, nrhs, blk0, upd0);
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    fprintf(stderr, "smcp_amd: store-pattern probe: %.3f ms per pass  ustack %p updp %p\n", ms / 3, (void*)D.ustack, (void*)D.updp);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
  }
  if (which >= 10) {      // probe: milliseconds of a linear fill of the buffer (which - 10), printed
    double* p = which == 10 ? D.updp : D.ustack;
    const size_t bytes = sizeof(double) * (size_t)(which == 10 ? D.max_rhs * D.updp_stride : D.ustack_cols * c->S.blklen());
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipMemsetAsync(p, 0, bytes, 0));
    HIPCHK(hipEventRecord(e0, 0));
    for (int r = 0; r < 3; ++r) HIPCHK(hipMemsetAsync(p, 0, bytes, 0));
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    fprintf(stderr, "smcp_amd: fill of %s: %.3f ms per pass, %.2f TB/s\n", which == 10 ? "updp" : "ustack", ms / 3, bytes / (ms / 3 * 1e-3) / 1e12);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
  }
  void* dummy = nullptr;
  if (shift > 0 && hipMalloc(&dummy, (size_t)shift) != hipSuccess) return SMCP_ENOMEM;
  int64_t junk = 0;
  if (which == 0) {
    double* nu = nullptr;
    if (dev_alloc(&nu, D.max_rhs * D.updp_stride, junk)) return SMCP_ENOMEM;
    HIPCHK(hipFree(D.updp));
    D.updp = nu;
  } else {
    double* nu = nullptr;
    const int64_t len = D.ustack_cols * c->S.blklen();
    if (dev_alloc(&nu, len, junk)) return SMCP_ENOMEM;
    HIPCHK(hipMemset(nu, 0, sizeof(double) * len));
    HIPCHK(hipFree(D.ustack));
    D.ustack = nu;
  }
  if (dummy) HIPCHK(hipFree(dummy));
  return 0;
}
int csp_debug_stamps(csp_ctx* c, unsigned long long* out, int reset) {
  if (int rc = ready(c)) return rc;
  HIPCHK(hipDeviceSynchronize());
  if (out) HIPCHK(hipMemcpy(out, c->D.red + 768, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (reset) HIPCHK(hipMemset(c->D.red + 768, 0, 32 * sizeof(unsigned long long)));
  return 0;
}

const char* csp_profile_kernel_name(int kid) { return (kid >= 0 && kid < KID_COUNT) ? KID_NAMES[kid] : nullptr; }

}  // extern "C"

#include "kkt.hip"
#include "kkt_qr.hip"
