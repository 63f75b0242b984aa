// Device context behind the opaque csp_ctx of include/smcp_amd.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <map>
#include <vector>

#include "symbolic.hpp"

namespace smcp {

// Per-clique descriptor as the kernels see it (device resident, one per clique).
struct CliqueDesc {
  int64_t blk;     // offset of the (nf x nn) panel in blkval
  int64_t upd;     // offset of the (na x na) update matrix in the update workspace
  int64_t updp;    // offset of its packed lower triangle in the child->parent exchange buffer
  int64_t rows;    // offset into rowidx (nf entries)
  int64_t rel;     // offset into relidx (na entries)
  int32_t nn, na;  // supernode / separator size
  int32_t parent;  // parent clique or -1
  int32_t chbeg, chend;  // children list range in chidx
  int32_t first;   // first permuted column (snptr[k])
  int32_t pad;     // slot of this clique among the large (HBM-class) fronts, -1 otherwise
};

struct DeviceCtx {
  int device = -1;
  int ncu = 256;           // compute units of the device (csp_device_init): launch heuristics
  int64_t max_rhs = 0;
  // index arrays
  CliqueDesc* cl = nullptr;
  int32_t* rowidx = nullptr;
  int32_t* relidx = nullptr;
  int32_t* chidx = nullptr;
  int32_t* levidx = nullptr;  // cliques sorted by level
  int32_t* lev2idx = nullptr; // per level: LDS-class cliques first, then HBM-class
  // extend-add gather plans (deterministic, atomic-free): per clique with children, the list of
  // front positions that receive contributions and, per position, the update-workspace offsets
  // of the children's entries that map onto it
  int64_t* gp_tptr = nullptr;  // nsn+1 : first target of clique k
  int32_t* gp_tgt = nullptr;   // target code: bit 30 = update-matrix block, i | j << 15
  int64_t* gp_cptr = nullptr;  // ntargets+1 : first contribution of target t
  int32_t* gp_src = nullptr;   // offset of the contributing entry in the (per right-hand side) update workspace
  // validity tags of the cached inverse-form factor / Y_AA blocks: the buffers they were derived
  // from.  Every in-place factor operation on a buffer clears the tags that mention it.
  const void* lk_tag_L = nullptr;   // LK was prepared from the factor stored at this address ...
  const void* lk_tag_Y = nullptr;   // ... or from the factor whose projected inverse now lives here
  const void* yaa_tag = nullptr;    // yaa holds the separator blocks of the matrix at this address
  const void* fac_tag = nullptr;    // fac = chol(yaa) of the matrix at this address
  bool fac_partial = false;         // ... except for the family children (childless small cliques whose Gram block comes from
                                    // k_leaf_pairs: nothing reads their factors on that route); complete_fac fills them in
  const void* faci_tag = nullptr;   // faci = fac^-1 of the matrix at this address
  double* lfd = nullptr;      // 64 x 64 doubles per large front: inverse of the current diagonal block
  int32_t* lev3idx = nullptr; // all LDS-class cliques (any level), then all large fronts: lists for clique-local kernels
  int64_t nI_total = 0, nII_total = 0;
  int nnmaxII_all = 0, namaxII_all = 0;   // maxima over the large fronts
  double* lfd_dense = nullptr;            // the 64 x 64 slot used by the dense (Schur complement) Cholesky
  double* lk = nullptr;       // inverse-form factor [L_NN^-1; L_AN L_NN^-1] of the most recent prep
  // workspaces
  double* upd = nullptr;   // max_rhs * updlen : update matrices
  double* updp = nullptr;  // max_rhs * updplen : packed lower triangles handed from children to parents (fast up-sweeps, cholesky)
  double* yaa = nullptr;   // updlen : Y[A_k,A_k] cache (Hessian)
  double* fac = nullptr;   // updlen : chol(Y_AA) cache
  double* faci = nullptr;  // updlen : inverse of fac (lower; positions above the diagonal are scratch)
  double* tmp = nullptr;   // max_rhs * tmplen : per-clique scratch (tmpptr)
  int64_t* tmpptr = nullptr;
  int64_t tmplen = 0;
  double* red = nullptr;   // reduction scratch
  int* info = nullptr;     // device failure flag
  unsigned long long* fp = nullptr; int* fp_bad = nullptr;   // cache verification: latched fingerprints [0..3], scratch [4..7], mismatch flag
  int* info_host = nullptr;  // pinned host mirror
  // constraints
  int64_t m = 0, cnnz = 0;
  int64_t* cptr = nullptr;   // CSC by constraint (m+1)
  int64_t* cidx = nullptr;   // blkval positions
  double* cval = nullptr;    // values
  double* cwval = nullptr;   // values with off-diagonals doubled (Amap weights)
  int64_t rnnz = 0;          // CSR by blkval position (for Aadj)
  int64_t* rpos = nullptr;   // distinct blkval positions (rnnz)
  int64_t* rptr = nullptr;   // rnnz+1
  int32_t* rcon = nullptr;   // constraint index per entry
  double* rval = nullptr;
  double* ustack = nullptr;  // max(m, max_rhs) * blklen : constraint matrices swept by the Hessian
  int64_t ustack_cols = 0;
  // column-sparse constraints (misc.SCMcolumn2 path, solvers.py:489-497): constraint entries in matrix
  // coordinates, the dense / sparse split, and per sparse constraint its nonzero-column set K_s
  int32_t* a_r = nullptr;    // cnnz : row of each constraint entry (permuted matrix coordinates)
  int32_t* a_c = nullptr;    // cnnz : column
  int32_t* s_rloc = nullptr; // cnnz : position of a_r within K_s (sparse constraints only)
  int32_t* s_cloc = nullptr; // cnnz : position of a_c within K_s
  int32_t* dlist = nullptr;  // md : constraints swept through the Hessian (Gram) path
  int32_t* slist = nullptr;  // ns : column-sparse constraints
  int32_t* kidx = nullptr;   // sum |K_s| : the column sets, concatenated in slist order
  int64_t md = 0, ns = 0;
  // constraint entries grouped by clique (sparse input of the Schur-complement sweeps, MfmaArgs::kc_*)
  int32_t* kc_ptr = nullptr; // nsn * (m + 1) : clique k, constraint j -> first entry
  int32_t* kc_off = nullptr; // cnnz : position inside the clique's panel
  double* kc_val = nullptr;  // cnnz
  int32_t* kc_ij = nullptr;  // cnnz : the same position as (row | column << 16) of the clique's panel (k_fam_sparse)
  // fused extend-add of the family parents' updates (MfmaArgs::fz_*): clique -> family number, record slot per family (written by
  // k_famt_prep), static term lists per (family, constraint); fz_ok: every family parent hangs under a front the LDS extend-add takes
  int32_t* fz_no = nullptr; int32_t* fz_slot = nullptr; int32_t* fz_ptr = nullptr; int32_t* fz_pk = nullptr; double* fz_s = nullptr;
  int64_t fz_nfam = 0; bool fz_ok = false;
  double* famc = nullptr;    // children's constants of the family parents of one sweep call, in LDS layout (k_fam2_prep)
  int64_t famc_len = 0;
  int64_t kc_maxlist = 0;    // longest entry list of a (clique, constraint) pair among possible family members
  int64_t fam_maxterms = 0;  // most entries of a (family, constraint) pair: the parent's own + its children's
  double fam_meanterms = 0;   // ... and their mean over all (family, constraint) pairs (what the entry-driven sweeps cost in proportion to)
  double* vbuf = nullptr;    // n x vcols : S^-1[:, K_s] of the chunk in flight
  double* trsm_x = nullptr; int64_t trsm_x_len = 0;   // scratch image of the right-hand sides of csp_trsm (tile-product route)
  int64_t vcols = 0;
  double* hd = nullptr;      // md x md Gram block of the dense constraints (when ns > 0)
  // blocked dense Cholesky of the Schur complement: inverses of its 64 x 64 diagonal blocks (for the blocked potrs)
  double* hinv = nullptr;    // nblocks * 4096 + 2 n doubles (the tail holds the two work vectors of the solve)
  int64_t hinv_cap = 0, hinv_n = 0;
  const void* hinv_tag = nullptr;   // the factor these inverses belong to
  double* h_pending = nullptr; int64_t h_pending_n = 0, h_pending_ld = 0; hipStream_t h_pending_stream = nullptr;   // a Schur complement left unfactored by kkt_schur_factor (deferred status)
  double* sw = nullptr;      // blklen : sqrt of the inner-product weights (Gram path)
  double* gpart = nullptr;   // partial Gram tiles
  int64_t gpart_len = 0;
  // Gram accumulation over slices of the stack (k_gram_diag128) and closed-form blocks of the family children
  // (front_leafgram.hip): tables of the cached set of ranges (csp_ctx::gsl_key)
  int64_t* gsl_start = nullptr; int32_t* gsl_len = nullptr; int64_t gsl_cap = 0; int gsl_n = 0;
  int gsl_spw = 0, gsl_early = 0;    // table built early rows first: slices per chunk and chunks of the early part (0: plain table)
  int32_t* lg_list = nullptr; int32_t* lg_slot = nullptr; int64_t lg_cap = 0; int lg_cnt = 0;     // family children inside the ranges, their index among all of them
  int32_t* lg_eptr = nullptr; int32_t* lg_epk = nullptr; double* lg_ew = nullptr; int32_t* lg_remap = nullptr;   // static entry lists per family child
  double* lg_tab = nullptr; int lg_rec = 0;                           // per-step tables (Psi, Omega), lg_rec doubles per child
  int lg_nf = 0, lg_nn = 0, lg_na = 0;                                // sizing over that list
  int64_t lg_children = 0, lg_maxent = 0, lg_pairs = 0, lg_rows = 0;  // over all family children (kkt_set_constraints)
  const double* part_Y = nullptr;   // the Y of the last kkt_prepare_part (sharded step): valid while part_valid
  int64_t updp_stride = 0;          // doubles between the packed exchange buffers of consecutive right-hand sides (>= updplen)
  int32_t* scm_owner = nullptr;     // per constraint: the part that computes its SCMcolumn2 columns (kkt_schur_gram_part), -1 = swept
  bool kc_sorted = false;    // every per-(clique, constraint) entry list ascends in panel position (columns are contiguous runs)
  bool lg_request = false;   // the running Schur sweep may leave the panels of the family children out
  bool lg_nochild = false;   // ... and did: the stack lacks them, their Gram block comes from k_leaf_gram
  // kkt_qr (csrc/kkt_qr.hip): m x m work matrices, the inverse of the triangular factor, G(bx) and reduction scratch;
  // the orthonormal factor Q itself overwrites ustack
  double* qr_ws = nullptr;
  int64_t qr_len = 0;
  // sparse-input sweep of childless large fronts (front_lfsp.hip): R^T and R^T K of the current factors, the longest
  // entry list of such a front, and the generations of fac / lk they were formed from
  double* sp_rt = nullptr; double* sp_mk = nullptr;
  int32_t* lfsp_list = nullptr; int64_t lfsp_cnt = 0; bool lfsp_exact = false;
  // sibling groups of the sparse-input sweep (front_lfsp.hip): childless large fronts under one large parent whose
  // separators are the same rows of it send ONE summed update per group.  Per clique: 1 = a group member whose slot in the
  // exchange buffer stays unwritten (the parent's extend-add skips it)
  uint8_t* lfsp_skip = nullptr;
  uint8_t* famt_skip = nullptr;   // family parents whose update is summed into their group leader's (k_fam_terms_grp)
  uint8_t* both_skip = nullptr;   // lfsp_skip | famt_skip
  int64_t kc_maxlist_large = 0;
  int64_t fac_gen = 0, lk_gen = 0, sp_fac_gen = -1, sp_lk_gen = -1;
  bool part_valid = false;     // lk / yaa / fac hold the sharded factor prepared by kkt_prepare_part (sets 2 then 1)
  bool qr_valid = false;       // ustack holds Q and qr_ws the factor for the matrices (qr_L, qr_Y)
  const void* qr_L = nullptr; const void* qr_Y = nullptr;
  int64_t lfd_len = 0;         // doubles of lfd (large-front slots + the dense slot)
  int64_t bytes = 0;
};

}  // namespace smcp

namespace smcp {
// optional per-kernel timing with HIP events on the launch stream
struct Profiler {
  bool on = false;
  int filter = -1;   // time only launches of this kernel id (-1: all)
  bool want(int kid) const { return on && (filter < 0 || filter == kid); }
  std::vector<hipEvent_t> ev;
  std::vector<int> kids;
  size_t used = 0;
  hipEvent_t next() {
    if (used == ev.size()) {
      hipEvent_t e;
      (void)hipEventCreate(&e);
      ev.push_back(e);
    }
    return ev[used++];
  }
};
}  // namespace smcp

namespace smcp {
struct LevelClass {
  int64_t nI = 0, nII = 0;     // cliques whose working set fits LDS / does not
  int nnmaxI = 0, namaxI = 0;  // LDS layout sizing for the LDS class
  int nchmaxI = 0, panmaxI = 0, pkmaxI = 0, plansumI = 0;  // index tables of the padded kernels (see pad_layout)
  int nnmaxII = 0, namaxII = 0;  // tile-grid sizing for the large-front (HBM) class
  int nnminII = 1 << 30;         // narrowest supernode of the class (the sparse first phase of wide childless fronts needs all of them wide)
  int nchmaxII = 0;              // most children of a large front (child table of k_lf_assemble_lds)
  // Families (front_fam.hip): the LAST nS cliques of the LDS class of this level are either childless cliques whose
  // parent is a family parent (level 0) or family parents (levels >= 1: small fronts all of whose children are such
  // childless cliques, at most one per wave).  A family-enabled sweep skips the former and hands the latter to
  // k_hess_up_fam; every other operation treats the LDS class as one list.
  int64_t nS = 0;
  int famna = 0, fampan = 0, fampk = 0, famcna = 0;   // sizing over the family parents / their children
  int famnn = 0, famcnn = 0;                          // widest supernode among the parents / among the children
};
}  // namespace smcp

namespace smcp {
// A subset of the cliques organised by level (LDS-class cliques first, then large fronts): set 0 = all
// cliques; sets 1 / 2 = the cliques this rank owns / the replicated top of the tree (multi-GPU sharding)
struct LevelSet {
  std::vector<LevelClass> lvl;     // per level
  std::vector<int64_t> off;        // per level: offset into lev2
  int32_t* lev2 = nullptr;         // device
};
}  // namespace smcp

struct csp_ctx {
  smcp::Symbolic S;
  smcp::LevelSet sets[3];
  std::vector<smcp::LevelClass> lvl;
  smcp::DeviceCtx D;
  smcp::Profiler prof;
  std::vector<int64_t> h_tmpptr;
  std::vector<int> lev_namax;   // per level: largest separator (sizes the gather launches)
  // Work that does not depend on the running leaves->root sweep and may run beside its large-front stage (the closed-form
  // Gram blocks of the family children beside the phase kernels of the top fronts): set by the caller of the sweep, taken
  // (once) by lf_up right after the first extend-add launch and started on a side stream; side_fork = that branch (a Fork,
  // capi.hip), joined by whoever consumes the results.  gpre: what the side work planned for gram_accumulate.
  std::function<void(hipStream_t)> side_work;
  void* side_fork = nullptr;
  struct GramPre { bool valid = false; int ngram = 0, nchunk = 0, spw = 0, nl = 0; int early = 0; bool early_done = false; } gpre;
  // (round 5) the Gram accumulation over the panels that are final once the sweep has passed level 1 -- the family parents',
  // most of the rows -- started on the side branch while the caller's stream goes on with the large fronts: set by schur_gram,
  // taken (once) by hess_up_fast behind level 1; early = the chunks of that part (the slice table lists those rows first)
  std::function<void(hipStream_t)> mid_work;
  struct LfspGroups { int32_t* ptr = nullptr; int32_t* list = nullptr; int ngroups = 0; };
  std::vector<LfspGroups> lfsp_grp;   // per level: the groups of its large-front class (device arrays; ngroups 0 = none)
  bool lfsp_any_groups = false;
  std::vector<LfspGroups> famt_grp;   // per level: sibling groups of its family parents (lists hold positions in the family list)
  bool famt_any_groups = false;
  std::vector<int64_t> fam;     // per clique: family role (CSP_Q_FAMILY)
  std::vector<uint8_t> is_diag_cache;
  std::vector<uint8_t> large_mask;   // per clique: a large (HBM-class) front
  // boundary exchange of the subtree partition: the subtree roots of all ranks (device: clique, owning rank, offset in
  // doubles per right-hand side inside the owner's region), what every rank contributes per right-hand side, the widest block
  int32_t* xr_roots = nullptr; int32_t* xr_owner = nullptr; int64_t* xr_bptr = nullptr;
  int64_t xr_n = 0; int xr_me = -1; int xr_world = 0; int64_t xr_npmax = 1;
  std::vector<int64_t> xr_size;
  int64_t ntrial = 1;                   // copies of the pattern in S (csp_symbolic_replicate): one failure flag per copy
  // side streams for clique-local launches that do not depend on each other (Fork in capi.hip): created on first use
  hipStream_t aux_stream[2] = {nullptr, nullptr};
  hipEvent_t aux_fork = nullptr, aux_join[2] = {nullptr, nullptr};
  int64_t scal_lstar = -1, scal_tail0 = 0;              // scaling_impl: first level without small cliques, start of the last level in blkval
  std::vector<uint8_t> fz_levels;       // per level: holds the parent front of some family parent (those levels' extend-add is the fused one)
  bool fz_set1_ok = false;              // csp_set_partition: every family parent this rank owns has its parent front on this rank too
  bool fz_live = false; int fz_nat = 0, fz_cnn = 0; int64_t fz_recl = 0;   // the running sweep's family launch left the parents' updates to the extend-add above (k_lf_assemble_fz)
  bool plan_full_upd = false;           // the gather plans list every update-block position of the large fronts (no clear pass needed)
  bool lazy_status = false;             // csp_lazy_status: failure flags are latched on the device, read by csp_status
  // one-launch blocked Cholesky with in-launch tile dataflow (front_flow.hip): workspace per stream in use (caller's, side 0,
  // side 1: two such launches may run side by side) and the ownership plans by matrix order
  struct FlowWs { double* P = nullptr; double* dinv = nullptr; unsigned* flags = nullptr; int cap_nt = 0, cap_fronts = 0; unsigned epoch = 0; };
  FlowWs flow_ws[3];
  struct FlowPlanDev { int nwg = 0; int32_t* own_ptr = nullptr; int32_t* own_tile = nullptr; };
  std::map<int, FlowPlanDev> flow_plans;     // key: order * 1024 + workgroups
  double placement_probe[2] = {0.0, 0.0};   // CSP_TUNE_PLACEMENT: probe time before / after, ms
  bool flags_clean = false;             // lazy mode: the last thing done to the flags was k_latch_status (which leaves them zero)
  int launch_err = 0;                   // first failed kernel launch of the running call (launch helpers); read by end_call
  double tnzcols = 0.1;                 // options['tnzcols'] (solvers.py:31,210-216)
  int leafgram_policy = 1;              // csp_tune: closed-form Gram blocks of the family children: 0 never, 1 when cheaper, 2 whenever possible
  bool verify_cache = false;            // csp_tune: fingerprints of the matrices the caches were derived from are checked on reuse
  bool deterministic = false;           // csp_tune: fixed-order summation everywhere (bit-identical results from run to run)
  std::vector<int32_t> lg_slot_of;      // clique -> index among the family children (-1: not one)
  std::vector<int64_t> gsl_key;         // the ranges (+ leaf switch) the slice table in D.gsl_* was built for
  std::vector<int64_t> h_kptr;          // ns + 1 : offsets into kidx, host copy for chunk planning
  std::vector<int32_t> h_slist;
};
