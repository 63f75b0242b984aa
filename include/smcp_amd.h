/*
 * smcp_amd.h -- C ABI of the MI355X-native chordal-matrix Newton-KKT kernels.
 *
 * The reference (cvxopt/smcp) has no plugin registry: its hot path is the set of Python
 * callables it imports from CHOMPACK / CVXOPT / its own misc extension inside
 * chordalsolver_feas/_esd (src/python/solvers.py:77-99).  Every entry point below names the
 * call it replaces.  Conventions (observed at the reference's call sites):
 *   - operations are IN PLACE on a flat fp64 vector `blkval` (clique k owns a dense
 *     column-major (nn+na) x nn block at blkptr[k]); the caller owns all numeric buffers;
 *   - all numeric pointers are DEVICE pointers (HBM) unless a parameter says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return value: 0 ok; k>0 = clique k-1 (or pivot) was not positive definite, which the
 *     Python shim turns into ArithmeticError exactly like CHOMPACK does
 *     (caught at solvers.py:628,643,658,...); <0 = usage / HIP error.
 * No CPU fallback exists: compute entry points fail with SMCP_ENODEV when no GPU is present.
 */
#ifndef SMCP_AMD_H
#define SMCP_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct csp_ctx csp_ctx;

#define SMCP_EINVAL (-1)
#define SMCP_ENODEV (-2)
#define SMCP_EHIP (-3)
#define SMCP_ENOMEM (-4)
#define SMCP_ESTALE (-5)  /* the prepared sharded factor was overwritten by another call (kkt_prepare_part again), or -- with
                             CSP_TUNE_VERIFY_CACHE -- a cached quantity is older than the matrix it was derived from */
#define SMCP_ETIMEOUT (-6) /* a workgroup of the one-launch blocked Cholesky (csrc/front_flow.hip) waited longer than 3 s for a tile
                             of another workgroup: the launch gave up instead of hanging (never seen; the results are invalid) */

/* ---- symbolic layer (host only, no GPU needed) ------------------------------------- */

/* chompack.symbolic(A, p) (solvers.py:305,308,314; analysis.py:49-51).
 * colptr/rowind: CCS lower-triangular pattern of the n x n aggregate sparsity pattern (host,
 * 64-bit indices as cvxopt int_t, src/C/cvxopt.h:46); perm: perm[new] = orig or NULL.
 * On failure returns NULL and writes a negative code to *info. */
csp_ctx* csp_symbolic_create(int64_t n, const int64_t* colptr, const int64_t* rowind,
                             const int64_t* perm, int64_t* info);
void csp_symbolic_destroy(csp_ctx* ctx);
/* K (<= 16) independent copies of the pattern of `base` as one symbolic object: copy t owns cliques t*nsn .., columns
 * t*n .. and the blkval range [t*blklen, (t+1)*blklen).  csp_cholesky / csp_completion on it factor K trial matrices
 * (stored one after the other) with the kernel launches of ONE factorisation, each copy with its own failure flag:
 * the trial factorisations of the reference's step-length searches (solvers.py:615-689, 928-939, 2172-2209), which
 * the reference runs one after the other.  The factorisation returns the first failing copy's code;
 * csp_trial_flags then gives every copy's flag (0 = inside the cone, else 1 + failing clique within the copy). */
csp_ctx* csp_symbolic_replicate(const csp_ctx* base, int64_t K, int64_t* info);
int csp_trial_flags(csp_ctx* ctx, int64_t K, int* out);

/* Deferred status.  The factorisations (csp_cholesky, csp_completion, the chol(Y_AA) inside the Hessians and the Schur
 * sweeps, dense_potrf) report "not positive definite" by reading a device flag back, which synchronises the stream:
 * the reference's calls raise at once (CHOMPACK cholesky -> ArithmeticError, solvers.py:881-891).  With
 * csp_lazy_status(ctx, 1) they return 0 without waiting and the first failure is latched on the device;
 * csp_status(ctx, stream) synchronises, returns the latched code (0 = none; otherwise what the failing call would
 * have returned) and clears it.  After a failure the calls that follow run on meaningless data until the status is
 * read -- safe (no index depends on values), but their results are void.  One KKT solve then costs one host
 * synchronisation instead of four. */
int csp_lazy_status(csp_ctx* ctx, int on);
int csp_status(csp_ctx* ctx, void* stream);

/* symb.p / snode / snptr / relptr / blkptr ... (cspmatrix internals [EXT], SURVEY App. A.1;
 * supernodes()/separators()/cliques() at analysis.py:173-175).  Returns the number of
 * elements of array `what`; when out != NULL copies them as int64 (host). */
enum {
  CSP_Q_SCALARS = 0, /* [n, nnz, nsn, fill, blklen, updlen, nlev, max_nn, max_na, max_front] */
  CSP_Q_PERM = 1, CSP_Q_IPERM = 2, CSP_Q_SNPTR = 3, CSP_Q_SNPAR = 4, CSP_Q_ROWPTR = 5,
  CSP_Q_ROWIDX = 6, CSP_Q_SEPPTR = 7, CSP_Q_RELIDX = 8, CSP_Q_BLKPTR = 9, CSP_Q_UPDPTR = 10,
  CSP_Q_CHPTR = 11, CSP_Q_CHIDX = 12, CSP_Q_LEVPTR = 13, CSP_Q_LEVIDX = 14, CSP_Q_CCSPTR = 15,
  CSP_Q_SNODE = 16,
  CSP_Q_FAMILY = 17  /* nsn, after csp_device_init: 2 = small front swept together with its childless children in one
                        workgroup (family kernel of the sparse-input Schur sweeps), 1 = such a child, 0 = neither */
};
int64_t csp_symbolic_query(const csp_ctx* ctx, int what, int64_t* out);

/* chompack.maxcardsearch (solvers.py:301); order[new] = orig (host). */
int csp_maxcardsearch(int64_t n, const int64_t* colptr, const int64_t* rowind, int64_t* order);
/* fill-reducing ordering for non-chordal input, in place of cvxopt.amd.order (solvers.py:278-279). */
int csp_mindegree(int64_t n, const int64_t* colptr, const int64_t* rowind, int64_t* order);
/* position in blkval of original-coordinate entries (I[e],J[e]); -1 outside V (host arrays).
 * Replaces the LI/lp/Ip/Jp index algebra of solvers.py:310-319. */
int csp_index_map(const csp_ctx* ctx, int64_t cnt, const int64_t* I, const int64_t* J, int64_t* out);

/* ---- device context ---------------------------------------------------------------- */

/* Upload the index arrays to `device` and allocate the internal update-matrix workspace for
 * up to max_rhs simultaneous right-hand sides (>= 1).  Idempotent for the same arguments; a larger max_rhs
 * grows the workspace.  One process drives ONE device: a different device for an initialised context, or for
 * another context of the same process, returns SMCP_EINVAL (one rank per GPU is the multi-GPU model). */
int csp_device_init(csp_ctx* ctx, int device, int64_t max_rhs);
/* Bytes of HBM held by the context (index arrays + workspaces). */
int64_t csp_device_bytes(const csp_ctx* ctx);

/* ---- chordal kernels (all in place, device pointers) ------------------------------- */

/* chompack.cholesky(X): X -> L, L L^T = X with zero fill (solvers.py:640,884,...). */
int csp_cholesky(csp_ctx* ctx, double* blkval, void* stream);
/* chompack.llt(L): L -> L L^T on V (solvers.py:904,1721). */
int csp_llt(csp_ctx* ctx, double* blkval, void* stream);
/* chompack.projected_inverse(L): L -> P_V((L L^T)^-1) (solvers.py:891,2361). */
int csp_projected_inverse(csp_ctx* ctx, double* blkval, void* stream);
/* The dual scaling point in one call (solvers.py:881-891: `L = S.copy(); cholesky(L); Y = projected_inverse(L)`, and
 * 2341-2361 in the embedding driver): on entry L holds S, on return L = chol(S) and Y = P_V(S^-1) -- the results of
 * csp_cholesky(L), a copy into Y and csp_projected_inverse(Y), status as csp_cholesky.  As ONE entry point the clique-local
 * stages that need finished levels only (the inverse-form factor of the lower levels, the copy into Y, and with
 * with_factors != 0 the Cholesky factors of the separator blocks Y_AA that the Hessian / Schur sweeps use next -- CHOMPACK's
 * factored updates) run on side streams beside the few-workgroup chains of the top fronts instead of after them.  L and Y
 * must be distinct blkval buffers. */
int csp_cholesky_projected_inverse(csp_ctx* ctx, double* L, double* Y, int with_factors, void* stream);
/* chompack.completion(X): X -> L with P_V((L L^T)^-1) = X (solvers.py:625,874,...). */
int csp_completion(csp_ctx* ctx, double* blkval, void* stream);
/* chompack.hessian(L, Y, U, adj, inv) (solvers.py:405,415,483,524,531,...): U holds nrhs
 * matrices, matrix r at U + r*ldu (the reference passes a Python list).  adj: 0 = False,
 * 1 = True, 2 = None (both factors).  inv: 0/1. */
int csp_hessian(csp_ctx* ctx, const double* L, const double* Y, double* U, int64_t nrhs,
                int64_t ldu, int adj, int inv, void* stream);
/* chompack.trsm(L, B, trans) (solvers.py:491-492): B is a dense n x nrhs column-major matrix
 * with rows in the permuted (symbolic) order; trans 0: L^-1 B, 1: L^-T B. */
int csp_trsm(csp_ctx* ctx, const double* L, double* B, int64_t nrhs, int64_t ldb, int trans,
             void* stream);
/* chompack.dot(X, Y) = tr(XY) on V (solvers.py:399,836,...); result written to *out (host). */
int csp_dot(csp_ctx* ctx, const double* X, const double* Y, double* out, void* stream);
/* sum(log(X.diag())) (solvers.py:395,925,934); *out host. */
int csp_logdiagsum(csp_ctx* ctx, const double* X, double* out, void* stream);
/* cspmatrix copy / +,- / a*X and blas.scal(a, X.blkval) (solvers.py:407,622,905):
 * y <- a*x + b*y over len doubles. */
int csp_axpby(int64_t len, double a, const double* x, double b, double* y, void* stream);

/* ---- KKT layer --------------------------------------------------------------------- */

/* Constraint data in blkval coordinates (replaces Av/Ip/Jp/Id, solvers.py:323-349):
 * host CSC arrays: constraint j has entries cptr[j]..cptr[j+1]-1 at blkval positions cidx with
 * values cval (lower-triangle values, diagonal positions flagged by the context itself). */
int kkt_set_constraints(csp_ctx* ctx, int64_t m, const int64_t* cptr, const int64_t* cidx,
                        const double* cval);
/* options['tnzcols'] (solvers.py:31,210-216; default 0.1): a constraint whose entries touch at most
 * int(n * tnzcols) distinct rows/columns is handled as misc.SCMcolumn2 does (misc.c:620-663, solvers.py:489-497:
 * S^-1[:, K] by two chompack.trsm calls, then pairwise contractions) instead of a Hessian sweep; misc.nzcolumns /
 * misc.matperm (misc.c:682-773) are folded into kkt_set_constraints.  Call before kkt_set_constraints. */
int kkt_set_tnzcols(csp_ctx* ctx, double tnzcols);
/* Amap (solvers.py:369-380): y[i] = <A_i, X>, i < m;  y device, length m. */
int kkt_amap(csp_ctx* ctx, const double* X, double* y, void* stream);
/* Aadj (solvers.py:382-386): X <- sum_i y[i] A_i (overwrites X). */
int kkt_aadj(csp_ctx* ctx, const double* y, double* X, void* stream);
/* kkt_chol factor step (solvers.py:479-501 + misc.c:620-663): builds the m x m Schur
 * complement H_ij = <A_i, hessian(L,Y)(A_j)> (lower) into H (device, ldh >= m) and factors it
 * in place (lapack.potrf, solvers.py:501).
 * DEFERRED CONTRACT.  Under csp_lazy_status(ctx, 1) (and unless SMCP_POTRF_DEFER=0) the call returns with H holding the
 * RAW Schur complement: its Cholesky factorisation waits for the first call of this library that reads H -- kkt_solve
 * (which runs it on a side stream beside its first Hessian sweep), dense_potrs, csp_status -- and is dropped when
 * dense_potrf / kkt_schur_columns / kkt_schur_factor is called on the same H again.  Until then the caller must keep H
 * alive and unmodified and must not read it by its own means (a copy, an all-reduce, its own potrs would see the raw
 * matrix): call csp_status or dense_potrs first, or factor eagerly (csp_lazy_status(ctx, 0), the default).  ONE matrix
 * per context can wait: kkt_schur_factor on another H, or kkt_set_constraints, first factors the waiting one where it
 * stands.  A caller that frees H while it may still be waiting calls kkt_schur_forget first. */
int kkt_schur_factor(csp_ctx* ctx, const double* L, const double* Y, double* H, int64_t ldh,
                     void* stream);
/* H is about to be freed or reused: drops what the context remembers about it (the deferred factorisation above, the
 * cached inverses of its diagonal blocks).  Nothing is launched. */
int kkt_schur_forget(csp_ctx* ctx, const double* H);
/* Columns j0..j1-1 of the (unfactored) Schur complement only: the unit that is sharded over
 * GPUs (each rank builds its column range, then one RCCL all-gather of H). */
int kkt_schur_columns(csp_ctx* ctx, const double* L, const double* Y, double* H, int64_t ldh,
                      int64_t j0, int64_t j1, void* stream);
/* N > 1 GPUs with REPLICATED factors (L, Y valid on every rank) and column-sparse constraints (the SCMcolumn2 route,
 * solvers.py:489-497, misc.c:620-663): caller `part` of `nparts` computes the columns of its contiguous share of the
 * column-sparse constraints (trsm x 2 + SCMcolumn2 per chunk) and, as part 0, the Gram block of the swept ones, into a
 * CLEARED H; one all-reduce (sum) of H over the callers completes the Schur complement (both triangles; a pair of sparse
 * constraints owned by two callers is written by the owner of the smaller index only).  Then dense_potrf on every rank. */
int kkt_schur_gram_part(csp_ctx* ctx, const double* L, const double* Y, double* H, int64_t ldh, int64_t part,
                        int64_t nparts, void* stream);
/* counts[0] = constraints swept through the Hessian, counts[1] = column-sparse ones (misc.nzcolumns / matperm,
 * misc.c:682-773; as classified by the last kkt_set_constraints under the current kkt_set_tnzcols). */
int kkt_constraint_classes(csp_ctx* ctx, int64_t* counts);
/* lapack.potrf / potrs on a dense device matrix (solvers.py:501,526). */
int dense_potrf(csp_ctx* ctx, double* A, int64_t n, int64_t lda, void* stream);
int dense_potrs(csp_ctx* ctx, const double* A, int64_t n, int64_t lda, double* B, int64_t nrhs,
                int64_t ldb, void* stream);
/* solve_ closure (solvers.py:506-541): given the factored H, overwrites bx (blkval) with x
 * and by (length m) with y for  [-kk*W^-1  A^adj; A 0][x;y] = [bx;by]. */
int kkt_solve(csp_ctx* ctx, const double* L, const double* Y, const double* H, int64_t ldh,
              double kk, double* bx, double* by, void* stream);

/* kkt_qr (solvers.py:413-475 feas, 1843-1905 esd): the QR-based KKT solver.  kkt_qr_factor builds the stack of
 * half-Hessian images of ALL m constraints (call kkt_set_tnzcols(ctx, 0) before kkt_set_constraints: the reference
 * does not split off column-sparse constraints on this path, solvers.py:242,551-556) and factors it, At = Q R, by
 * Cholesky-QR iterations on the device (csrc/kkt_qr.hip; a shifted first pass when chol(At^T At) breaks down).  Q
 * (orthonormal in the Schur complement's inner product) stays on the device in place of the stack, R^T as an m x m
 * lower-triangular matrix.  *passes (optional) receives the number of passes taken, *shift the relative shift of
 * the first pass (0: none).  Returns 0, a negative error, or j+1 > 0 when chol(Y_AA) of a clique or the Gram
 * matrix is not positive definite (lapack.geqrf's ArithmeticError at solvers.py:425-428 has no counterpart: a
 * rank-deficient stack shows up here).  any m (factors wider than 320 columns are processed in panels of 256).
 * kkt_qr_solve is its solve_ closure (solvers.py:430-471): overwrites bx (blkval) with x and by (length m) with y;
 * valid until the next call that rewrites the stack (kkt_qr_factor, kkt_schur_*, kkt_solve, kkt_gram_*).
 * NOT sharded: Q lives on one device.  After csp_set_partition with more than one owning rank kkt_qr_factor returns
 * SMCP_EINVAL (the multi-GPU step uses kkt_chol: kkt_gram_* / csp_*_part); a partition of one rank is accepted. */
int kkt_qr_factor(csp_ctx* ctx, const double* L, const double* Y, int64_t* passes, double* shift, void* stream);
int kkt_qr_solve(csp_ctx* ctx, const double* L, const double* Y, double kk, double* bx, double* by, void* stream);
/* test hook: Rt_host (m*m doubles, host, optional) <- R^T (lower, column-major); G_dev (m*m doubles, device,
 * optional) <- Gram matrix of the current stack in the weighted inner product (Q^T Q; the identity after a
 * factorisation).  Synchronises the stream. */
int kkt_qr_inspect(csp_ctx* ctx, double* Rt_host, double* G_dev, void* stream);

/* ---- subtree-sharded multi-GPU Schur complement (Gram formulation) ---------------------------
 * The elimination tree is cut into subtrees owned by single ranks plus a replicated top
 * (owner[k] = rank, or -1 for the top).  Per solve: kkt_gram_prepare; for every chunk of right-hand
 * sides: kkt_gram_sweep(set 1 = owned) -> exchange of the subtree roots' packed update blocks
 * (csp_exchange_pack -> one all-gather -> csp_exchange_unpack) -> kkt_gram_sweep(set 2 = top); then kkt_gram_accumulate over
 * the blkval ranges this rank owns and one all-reduce of H. */
int csp_set_partition(csp_ctx* ctx, const int32_t* owner, int rank);
int kkt_gram_prepare(csp_ctx* ctx, const double* L, const double* Y, void* stream);
int kkt_gram_sweep(csp_ctx* ctx, int set, int64_t j0, int64_t j1, void* stream);
int kkt_gram_accumulate(csp_ctx* ctx, int64_t nranges, const int64_t* ranges, double* H, int64_t ldh,
                        void* stream);
/* boundary exchange (after csp_set_partition): doubles per right-hand side every rank contributes; pack of THIS rank's
 * subtree roots into buf ([root][rhs][packed block]); unpack of all OTHER ranks' roots from the all-gathered buffers
 * (`width` doubles per rank).  One launch each, no host synchronisation. */
int csp_exchange_sizes(csp_ctx* ctx, int64_t world, int64_t* sizes_per_rhs);
int csp_exchange_pack(csp_ctx* ctx, int64_t nrhs, double* buf, void* stream);
int csp_exchange_unpack(csp_ctx* ctx, int64_t nrhs, const double* buf, int64_t width, void* stream);
/* The exchange BY CONSTRAINT SHARE (the top of the tree sharded by constraint: rank q sweeps the top for its share J_q of the
 * constraints only, instead of every rank for all of them; SURVEY 8e "one exchange step per sweep", as an all-to-all):
 * after kkt_gram_sweep(set 1) over a chunk, csp_exchange_pack_range packs the right-hand sides r0 .. r0 + nrhs - 1 of that chunk
 * -- the share of ONE destination rank -- as [root][rhs][packed block] (once per destination, into that destination's slot of
 * the all-to-all's send buffer); csp_exchange_unpack_all unpacks the roots of EVERY rank, this one's too, from the receive
 * buffer (`width` doubles per source rank) into the slots 0 .. nrhs - 1, where kkt_gram_sweep(set 2, c0, c1) of the own share
 * reads them.  kkt_stack_rows moves the rows [a, b) of the swept stack of the constraints j0 .. j1 - 1 to / from
 * buf[(j - j0) (b - a) + row - a] (dir 0: stack -> buf, 1: buf -> stack): the top's panels of a share travel to the rank that
 * accumulates the top's block of H. */
int csp_exchange_pack_range(csp_ctx* ctx, int64_t r0, int64_t nrhs, double* buf, void* stream);
int csp_exchange_unpack_all(csp_ctx* ctx, int64_t nrhs, const double* buf, int64_t width, void* stream);
int kkt_stack_rows(csp_ctx* ctx, int dir, int64_t j0, int64_t j1, int64_t a, int64_t b, double* buf, void* stream);
/* The boundary blocks of a sweep whose input is a linear combination of inputs that were swept (and exchanged) before
 * need no collective: the second Hessian of solve_ is applied to Aadj(y) - bx (solvers.py:528-531), so the other
 * ranks' root blocks are sum_i y_i (blocks gathered for constraint i during the Schur sweeps) - (blocks gathered for
 * bx).  gbuf: the gathered buffer of a chunk of nrhs constraints (region width gwidth per rank), y: its nrhs
 * multipliers (device), out: a buffer in the layout of a one-right-hand-side gather (region width owidth);
 * mode 0: out <- sum - out (out holds the blocks of bx), mode 1: out <- out + sum (further chunks). */
int csp_exchange_combine(csp_ctx* ctx, int64_t nrhs, const double* y, const double* gbuf, int64_t gwidth, double* out,
                         int64_t owidth, int mode, void* stream);

/* ---- subtree-sharded factorisation and solve (every leaves->root / root->leaves sweep of the path
 * shards; reference call sites of the sweeps: solvers.py:881-891 cholesky + projected_inverse, 521-532 the two
 * Hessians of solve_).  set: 1 = the cliques this rank owns, 2 = the replicated top.
 *   leaves->root (cholesky, hessian dir 0): part(set 1) -> csp_exchange_pack / all-gather / csp_exchange_unpack of the subtree roots' packed updates
 *     (nrhs = 1) around one collective -> part(set 2);
 *   root->leaves (projected_inverse, kkt_prepare_part, hessian dir 1): part(set 2) -> part(set 1), no communication.
 * After them a matrix is valid on the owned cliques and the top; the other blkval ranges keep what they held.
 * kkt_prepare_part: Y_AA blocks and their Cholesky factors (and, with_lk != 0, the inverse-form factor of L) of one set;
 * kkt_gram_prepare_part: what is left of kkt_gram_prepare once both sets are prepared;
 * csp_hessian_sweep_part: one half of hessian(L, Y, U, adj=None, inv=False) over one set (dir 0 up, 1 down).
 * The prepared state lives in the context's shared lk / yaa / fac buffers: any call that rewrites them for other
 * matrices (csp_projected_inverse, csp_hessian, kkt_* on another pair) makes the sweeps return SMCP_ESTALE until
 * kkt_prepare_part (with_lk = 1) has been called again for set 2, then set 1. */
int csp_cholesky_part(csp_ctx* ctx, double* blkval, int set, void* stream);
int csp_projected_inverse_part(csp_ctx* ctx, double* blkval, int set, void* stream);
int kkt_prepare_part(csp_ctx* ctx, const double* L, const double* Y, int set, int with_lk, void* stream);
int kkt_gram_prepare_part(csp_ctx* ctx, void* stream);
int csp_hessian_sweep_part(csp_ctx* ctx, double* U, int64_t nrhs, int64_t ldu, int set, int dir, void* stream);

/* ---- derived-quantity cache ------------------------------------------------------------
 * csp_hessian, csp_completion and the kkt_* entry points keep quantities derived from their (L, Y)
 * arguments on the device (the inverse-form factor of L, the separator blocks Y_AA, their Cholesky
 * factors and the inverses of those), keyed by the DEVICE ADDRESS of L and Y, so that the many
 * Hessian applications of one interior-point iteration (solvers.py:803-1050: kkt_res, Newton
 * decrements, step computation) do not redo them.  Every entry point of this library that writes
 * to a buffer drops the cache entries derived from it.  A caller that changes the contents of L or
 * Y by any other means (its own kernels, memcpy) must call csp_cache_reset() before the next call. */
int csp_cache_reset(csp_ctx* ctx);
/* The explicit form of that contract: the caller has written to the matrix (or right-hand side) stored at `ptr` by
 * means the library cannot see -- the reference does so with blas.scal(a, X.blkval) (solvers.py:407, 905) -- and
 * whatever was derived from the old contents at that address is dropped (cheaper than csp_cache_reset: quantities
 * derived from other matrices stay). */
int csp_touch(csp_ctx* ctx, const void* ptr);

/* ---- tuning / debugging knobs ------------------------------------------------------------ */
#define CSP_TUNE_LEAFGRAM 1       /* closed-form Gram blocks of childless small cliques (front_leafgram.hip): 0 never,
                                     1 when the entry lists are short enough to beat the panel route (default), 2 always */
#define CSP_TUNE_VERIFY_CACHE 2   /* 1: every reuse of a cached derived quantity first checks a fingerprint of the matrix
                                     it was derived from -- every entry of every panel, one pass over blkval + a stream
                                     synchronisation per reuse (which makes csp_lazy_status pointless while it is on: a
                                     debug aid); a caller that forgot csp_touch gets SMCP_ESTALE instead of stale factors */
#define CSP_TUNE_DETERMINISTIC 3  /* 1: every sum in a fixed order (no floating-point atomics): results are bit-identical
                                     from run to run; slower */
#define CSP_TUNE_PLACEMENT 4      /* value = tries (1..16): an ACTION, not a setting -- call it after kkt_set_constraints.  The
                                     family sweep of the Schur complement streams into two multi-GB buffers at once, and how fast
                                     the memory system takes that depends on where the buffers physically lie (+-20 % on that
                                     kernel from one allocation to the next).  A store-only probe of the pattern is timed, the
                                     two buffers -- the packed exchange buffer and the swept stack -- are moved in turn to fresh
                                     allocations, `tries` in all, and the fastest placement is kept (a few ms per try; the buffers
                                     set aside are freed at the end: up to 16 GB of transient memory).  The contents of both
                                     buffers are NOT preserved (the probe zeroes the family parents' panels and update slots):
                                     call it between Newton steps; a kkt_qr factor held in the stack is invalidated and
                                     kkt_qr_solve returns SMCP_EINVAL until the next kkt_qr_factor.  Worth it for runs of many
                                     Newton steps on one problem; off unless called */
#define CSP_TUNE_RACE 5           /* value = seed (low 32 bits; 0: off) | longest delay in us << 32 (0: 200), PROCESS-wide: delay injection for race hunting -- a spin kernel of a
                                     seeded random 5 .. 200 us at the head and tail of every internal side-stream branch, behind every
                                     fork on the caller's stream and before one launch in four (tools/race_hunt.sh; the environment
                                     variable SMCP_RACE=<seed> does the same from the first call on).  Results must not change. */
#define CSP_TUNE_RACE_DROP_JOINS 6 /* 1: the harness's own sensitivity test -- side branches are no longer joined (their join event is
                                     recorded, the caller's stream does not wait for it): results are WRONG by design and must
                                     change under CSP_TUNE_RACE, which is what tests/test_gpu_parity.py asserts.  0: back to normal
                                     (waits for the device).  PROCESS-wide. */
int csp_tune(csp_ctx* ctx, int what, int64_t value);
/* out[0], out[1]: milliseconds of the store-pattern probe of the last CSP_TUNE_PLACEMENT before / after (zeros: not run);
 * out[2]: delay kernels injected so far in this process (CSP_TUNE_RACE).  out holds three doubles. */
int csp_tune_report(csp_ctx* ctx, double* out);

/* ---- measurement hooks (bench.py roofline leg) --------------------------------------- */
/* When enabled every kernel launch is bracketed by HIP events on its own stream. */
int csp_profile_enable(csp_ctx* ctx, int on);
/* Restrict the timing to launches of one kernel id (so that a throughput measurement can carry the HIP events of
 * its dominant kernel without paying for events around every launch); kid < 0 times all kernels again. */
int csp_profile_filter(csp_ctx* ctx, int kid);
/* Number of kernel kinds = required length of the arrays passed to csp_profile_read. */
int64_t csp_profile_kinds(void);
/* Synchronises, then writes per-kernel accumulated milliseconds and launch counts (host arrays of
 * csp_profile_kinds() entries; the return value repeats that length) and clears the record. */
int64_t csp_profile_read(csp_ctx* ctx, double* ms, int64_t* count);
const char* csp_profile_kernel_name(int kid);

#ifdef __cplusplus
}
#endif
#endif
