// Every environment switch of the library in ONE table: name, default, meaning.  Every site reads its switch through this
// table (sw_str / sw_on / sw_int: the only getenv of the library), and a variable SMCP_* found in the environment that the table
// does not know is reported on stderr once per process (a mistyped switch would otherwise be silently ignored).  Defaults are the production
// routes; everything else exists for A/B measurements (DESIGN.md section 3 quotes the numbers) and for the parity suite,
// which runs the fallback routes through them.  SMCP_BENCH_* belong to bench.py, SMCP_CXXFLAGS / SMCP_STAMPS to the build.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

extern char** environ;

namespace smcp {

struct Switch { const char* name; const char* dflt; const char* doc; };

static const Switch SWITCHES[] = {
  // ---- routes (1 = production route on, 0 = the fallback)
  {"SMCP_GENERIC", "0", "1: any-size fixed-order kernels of front_generic.hip for every tree operation (as CSP_TUNE_DETERMINISTIC)"},
  {"SMCP_LARGE", "1", "0: no tiled large-front kernels (fronts beyond LDS take the one-workgroup HBM kernels)"},
  {"SMCP_NOCACHE", "0", "1: nothing derived from (L, Y) is reused between calls"},
  {"SMCP_FORK", "1", "0: no side streams, everything on the caller's stream"},
  {"SMCP_AUX_PRIO", "1", "0: side stream 0 (few-workgroup chains) at default instead of highest priority"},
  {"SMCP_AUX_PRIO1", "1", "0: side stream 1 (filler work) at default instead of lowest priority"},
  {"SMCP_EVENT_SYSFENCE", "0", "1: internal events with the default system-scope release (12 us per hand-over instead of 7)"},
  {"SMCP_SCALING_OVERLAP", "1", "0: csp_cholesky_projected_inverse as its three steps one after the other"},
  {"SMCP_CHOL_PREP", "1", "0: k_prep_lk in a launch of its own instead of inside k_chol_mfma (narrow supernodes)"},
  {"SMCP_FAC_PARTIAL", "1", "0: chol(Y_AA) of the family children always formed by the fused scaling point"},
  {"SMCP_MID", "1", "0: per-step kernels instead of k_mid_chol / k_lf_diag_inv for fronts of at most 272 rows"},
  {"SMCP_TRTRI", "1", "0: block-row triangular inversion of large fronts instead of recursive doubling"},
  {"SMCP_N16", "1", "0: no shape-specialised k_hess_up_n16"},
  {"SMCP_OLDLDS", "0", "1: k_hess_up_mfma<true> instead of the padded / n16 kernels"},
  {"SMCP_DOWN_W", "1", "0: k_hess_down_mfma instead of the one-wave k_hess_down_w"},
  {"SMCP_FAM", "1", "0: no family kernels (per-level sweeps with the update exchange through HBM)"},
  {"SMCP_FAM2", "1", "0: no sparse-input family kernels (k_fam_sparse, k_fam_terms): k_hess_up_fam"},
  {"SMCP_FAMT", "1", "0: no entry-driven family sweep k_fam_terms (k_fam_sparse instead)"},
  {"SMCP_FZ", "1", "0: the family parents' packed updates go through HBM (k_fam_terms writes, the extend-add reads) instead of being formed by k_lf_assemble_fz"},
  {"SMCP_FAMT_GROUP", "1", "0: no sibling groups of family parents (k_fam_terms_grp)"},
  {"SMCP_LFSP", "1", "0: dense phases instead of the sparse-input sweep k_lfsp_up of childless large fronts"},
  {"SMCP_LFSP_GROUP", "1", "0: no sibling groups in k_lfsp_up"},
  {"SMCP_ZSP", "1", "0: dense first phase of wide childless fronts instead of k_lf_zsp"},
  {"SMCP_LEAFGRAM", "1", "0: never the closed-form Gram blocks of the family children (k_leaf_tables / k_leaf_pairs)"},
  {"SMCP_LG_SIDE", "1", "0: leaf Gram blocks after the sweep on the caller's stream instead of beside the top phases"},
  {"SMCP_GRAM_EARLY", "0", "1: the Gram chunks of the levels 0 / 1 rows on the side branch beside the top fronts' phase kernels (measured slower: 3.37 against 3.22 ms per step)"},
  {"SMCP_LG_EARLY", "1", "0: leaf Gram blocks beside the phase kernels of the top fronts instead of from the start of the sweep"},
  {"SMCP_FACI_LDS", "1", "0: inverse Y_AA factors of the small fronts by block rows against HBM (k_factor_inverse) instead of in LDS"},
  {"SMCP_FZ_TAIL", "1", "0: every (front, right-hand side) pair of the fused extend-add is one workgroup's task, also in a thin last round"},
  {"SMCP_GRAM", "1", "0: reference (two-sweep) formulation of the Schur complement instead of the Gram formulation"},
  {"SMCP_SCM", "1", "0: column-sparse constraints swept like the others (no SCMcolumn2 route)"},
  {"SMCP_TRSM_MM", "1", "0: csp_trsm through the generic level kernels instead of tile products"},
  {"SMCP_ALDS", "1", "0: extend-add of large fronts by gather plan only (no LDS streaming kernel)"},
  {"SMCP_ALDS_DYN", "1", "0: one workgroup per (front, right-hand side) instead of the task-drawing grid"},
  {"SMCP_ALDS_FILL", "1", "0: k_panel_fill builds the input panels instead of the extend-add (sgn 3)"},
  {"SMCP_ASM", "plan", "t: tiled extend-add kernel instead of the gather plan"},
  {"SMCP_ROOT_FUSED", "0", "1: fused two-product sweep of fronts without separator (Y_NN explicit: condition squared; studies only)"},
  {"SMCP_PD", "0", "1: four-wave tile products everywhere (no sixteen-wave shape for small launches)"},
  {"SMCP_POTRF_OLD", "0", "1: generic one-workgroup dense Cholesky"},
  {"SMCP_FLOW", "1", "0: no one-launch blocked Cholesky with in-launch tile dataflow (front_flow.hip): per-step kernels for dense matrices / single fronts beyond 272 rows"},
  {"SMCP_FLOW_WG", "112", "workgroups of that launch (at most 112: four launches fit the chip side by side)"},
  {"SMCP_POTRS_OLD", "0", "1: generic one-workgroup dense solve"},
  {"SMCP_POTRS_STEPS", "0", "1: per-block launches of the dense solve instead of k_dense_potrs_one"},
  {"SMCP_POTRF_DEFER", "1", "0: kkt_schur_factor factors H at once even under deferred status"},
  {"SMCP_QR_TRSM", "mfma", "f: vector-FMA substitution kernel of kkt_qr instead of the MFMA one"},
  // ---- launch shapes and thresholds (timing studies)
  {"SMCP_FTHR_CHOL", "128", "threads of k_chol_mfma on the leaf class"},
  {"SMCP_FTHR_PINV", "256", "threads of k_pinv_mfma on the leaf class"},
  {"SMCP_FTHR_YAA", "64", "threads of k_factor_yaa_lds on the leaf class"},
  {"SMCP_FTHR_YAA_MID", "256", "threads of k_factor_yaa_lds on separator blocks of 33 .. 64 rows"},
  {"SMCP_N16_THR_LEAF", "128", "threads of k_hess_up_n16 for one or two right-hand sides on childless fronts"},
  {"SMCP_DIAG_THREADS", "512", "threads of k_lf_diag"},
  {"SMCP_POTRF_THREADS", "1024", "threads of k_dense_potrf_small"},
  {"SMCP_FZ_THREADS", "1024", "threads of k_lf_assemble_fz (512: eight waves, no register spills, slower)"},
  {"SMCP_PD_WGS", "1", "largest launch, in workgroups per CU, that takes the sixteen-wave tile shape"},
  {"SMCP_ALDS_Z", "0", "workgroups per (front, right-hand side) of the LDS extend-add (0: automatic)"},
  {"SMCP_RHS_SPLIT", "16", "fewest right-hand sides for the two-stream split of a large-front sweep (plain extend-add launch only)"},
  {"SMCP_RHS_PARTS", "2", "parts of that split (2 or 3)"},
  {"SMCP_RHS_SPLIT_DYN", "0", "1: the split also with the task-drawing extend-add"},
  {"SMCP_FAM_MINRHS", "1", "fewest dense right-hand sides for the family kernel"},
  {"SMCP_FAMT_G", "0", "right-hand-side slices per family of k_fam_terms (0: cost model)"},
  {"SMCP_FAM2_STAG", "0", "stagger of k_fam_sparse's groups (studies)"},
  {"SMCP_FAMT_MEAN", "36", "entry-driven family sweeps (k_fam_terms, fused extend-add) up to this many entries per (family, constraint) list ON AVERAGE (lists up to 384 entries, in chunks); beyond: the dense family sweep"},
  {"SMCP_LFSP_OCC", "2", "waves per SIMD the grouped update phase of k_lfsp_up is compiled for (1: 512 registers, no spills; 2: 256, ~100 spilled)"},
  {"SMCP_LFSP_G", "1", "right-hand sides per workgroup of k_lfsp_up"},
  {"SMCP_GRAM_NW", "16", "waves per workgroup of k_gram_diag128 (4, 8 or 16)"},
  {"SMCP_GRAM_MINCHUNK", "512", "fewest stack rows per workgroup of the Gram accumulation"},
  {"SMCP_UPDP_PAD", "0", "doubles of padding between the packed exchange buffers of consecutive right-hand sides"},
  {"SMCP_QR_PASSES", "0", "Cholesky-QR passes of kkt_qr (0: decided by the deviation test)"},
  {"SMCP_QR_P", "1", "positions per lane of the FMA substitution kernel (1 or 2)"},
  // ---- diagnostics
  {"SMCP_FLOW_STAMPS", "0", "1: per-workgroup time accounts of the one-launch blocked Cholesky on stderr (waits for every launch)"},
  {"SMCP_POISON", "0", "1: every fp64 device buffer of the library (and the exchange buffers of smcp_amd/kkt.py) starts as 4.5e150 (hunting reads of never-written workspace)"},
  {"SMCP_RACE", "0", "seed > 0: delay injection on every internal stream hand-over and one launch in eight (race hunting; results must not change)"},
  {"SMCP_TRACE", "0", "1: every launch named on stderr and waited for (a device fault points at its kernel)"},
  {"SMCP_TIMING", "0", "1: wall-clock marks of the set-up phases on stderr"},
  {"SMCP_OCC", "0", "1: occupancy decisions of k_hess_up_n16 on stderr"},
  {"SMCP_DEBUG_ADDR", "0", "1: addresses of the large allocations on stderr"},
  {"SMCP_CONTIG", "0", "1: large buffers from physically contiguous memory (placement studies: slower)"},
  {"SMCP_SKIP", "0", "phase mask of the sweep kernels (ablation timing; results are wrong)"},
  {"SMCP_GSKIP", "0", "ablation mask of the Gram kernel (results are wrong)"},
  {"SMCP_LGSKIP", "0", "ablation mask of k_leaf_pairs (results are wrong)"},
  {"SMCP_QR_FAKE", "0", "1: timing experiment of the FMA substitution kernel (results are wrong)"},
};

// once per process: report SMCP_* variables of the environment that the table does not know
inline void sw_check_environment() {
  static const bool done = [] {
    for (char** e = environ; e && *e; ++e) {
      if (std::strncmp(*e, "SMCP_", 5) != 0) continue;
      const char* eq = std::strchr(*e, '=');
      const std::string name(*e, eq ? (size_t)(eq - *e) : std::strlen(*e));
      if (name.rfind("SMCP_BENCH_", 0) == 0 || name.rfind("SMCP_FUZZ_", 0) == 0 || name == "SMCP_CXXFLAGS" || name == "SMCP_STAMPS" ||
          name == "SMCP_SHARD_KEEP" || name == "SMCP_SHARD_TOP" || name == "SMCP_GIT_SHA" || name == "SMCP_FS_MODE") continue;
      bool known = false;
      for (const Switch& s : SWITCHES) if (name == s.name) known = true;
      if (!known) std::fprintf(stderr, "smcp_amd: unknown switch %s in the environment (see smcp_amd/csrc/switches.hpp)\n", name.c_str());
    }
    return true;
  }();
  (void)done;
}
// value of a switch OF THE TABLE as it stands in the environment now (most sites cache their answer on first use; the
// set-up entry points -- csp_device_init, kkt_set_constraints -- read theirs per call), or nullptr.  A name the table does not
// list is a programming error and is reported.
inline const char* sw_str(const char* name) {
  sw_check_environment();
  bool known = false;
  for (const Switch& s : SWITCHES) if (!std::strcmp(name, s.name)) { known = true; break; }
  if (!known) std::fprintf(stderr, "smcp_amd: switch %s is read but not listed in switches.hpp\n", name);
  return std::getenv(name);
}
inline int sw_on(const char* name, int dflt) { const char* v = sw_str(name); return v ? (v[0] == '0' ? 0 : (v[0] == '1' ? 1 : dflt)) : dflt; }
inline int sw_int(const char* name, int dflt) { const char* v = sw_str(name); return v ? std::atoi(v) : dflt; }

}  // namespace smcp
