// Host-side symbolic analysis of a chordal (or chordally-embedded) sparsity pattern.
//
// Replaces what the reference obtains from chompack.symbolic / cspmatrix at
// src/python/solvers.py:301-319 (maxcardsearch/peo/symbolic, sparsity_pattern, second
// symbolic on the permuted pattern) and analysis.py:49-51,173-175 (supernodes/separators/
// cliques).  CHOMPACK itself is not vendored in the reference (pyproject.toml:25-28), so
// this follows the published algorithms: Liu's elimination tree, row-subtree column
// counts, Pothen-Sun maximal supernodes, supernodal postorder, relative indices.
//
// Everything here is plain host C++; the result is an immutable set of index arrays that
// the device context uploads once.
#pragma once
#include <cstdint>
#include <vector>

namespace smcp {

struct Symbolic {
  int64_t n = 0;          // matrix order
  int64_t nnz = 0;        // |V| : lower-triangular nonzeros of the filled pattern
  int64_t nsn = 0;        // number of supernodes (= cliques)
  int64_t fill = 0;       // nnz(filled) - nnz(input, lower incl. diagonal)
  std::vector<int64_t> p;       // p[new] = original index      (length n)
  std::vector<int64_t> ip;      // ip[orig] = new index
  std::vector<int64_t> snptr;   // nsn+1 : supernode k owns permuted columns snptr[k]..snptr[k+1]-1
  std::vector<int64_t> snode;   // n     : supernode of permuted column j
  std::vector<int64_t> snpar;   // nsn   : parent supernode, -1 for roots
  std::vector<int64_t> rowptr;  // nsn+1 : rows of clique k are rowidx[rowptr[k]..rowptr[k+1]) (N_k then A_k, ascending)
  std::vector<int32_t> rowidx;
  std::vector<int64_t> sepptr;  // nsn+1 : sepptr[k] = sum_{j<k} |A_j|
  std::vector<int32_t> relidx;  // |A_k| entries at sepptr[k]: position of each A_k row inside rows(parent(k))
  std::vector<int64_t> blkptr;  // nsn+1 : offset of the (nn+na) x nn column-major block of clique k in blkval
  std::vector<int64_t> updptr;  // nsn+1 : offset of the na x na update matrix of clique k in the update workspace
  std::vector<int64_t> updpptr; // nsn+1 : OFFSET of the packed lower triangle (na(na+1)/2) of clique k in the child->parent exchange
                                //           buffer, siblings side by side (not monotone in k); [nsn] = total length
  std::vector<int64_t> chptr;   // nsn+1 : children lists
  std::vector<int64_t> chidx;   // nsn-#roots
  std::vector<int64_t> level;   // nsn   : height-based level (leaves = 0, parent > all children)
  std::vector<int64_t> levptr;  // nlev+1
  std::vector<int64_t> levidx;  // nsn   : cliques sorted by level
  std::vector<int64_t> ccsptr;  // n+1   : column pointers of the filled lower pattern in permuted order
  int64_t nlev = 0;
  int64_t max_nn = 0, max_na = 0, max_front = 0;

  int64_t nn(int64_t k) const { return snptr[k + 1] - snptr[k]; }
  int64_t nf(int64_t k) const { return rowptr[k + 1] - rowptr[k]; }
  int64_t na(int64_t k) const { return nf(k) - nn(k); }
  int64_t blklen() const { return blkptr[nsn]; }
  int64_t updlen() const { return updptr[nsn]; }
  int64_t updplen() const { return updpptr[nsn]; }
};

// Build from the lower-triangular pattern (CCS, row indices need not be sorted, diagonal
// optional) of an n x n symmetric matrix. perm (nullable): perm[new] = orig, applied first.
// merge_tol >= 0 enables relaxed supernode amalgamation (parent absorbs child when the
// number of explicit zeros introduced is <= merge_tol); 0 disables it.
// Returns 0 on success, negative on malformed input.
int symbolic_build(int64_t n, const int64_t* colptr, const int64_t* rowind,
                   const int64_t* perm, Symbolic& out);

// Maximum cardinality search ordering (Tarjan & Yannakakis). order[new] = orig such that the
// result is a perfect elimination ordering iff the graph is chordal
// (reference call site: src/python/solvers.py:301-303, chompack.maxcardsearch + peo).
void maxcardsearch(int64_t n, const int64_t* colptr, const int64_t* rowind, int64_t* order);

// Greedy minimum-degree ordering with explicit elimination graph (used for non-chordal
// input in place of cvxopt.amd.order, src/python/solvers.py:278-279).
void mindegree(int64_t n, const int64_t* colptr, const int64_t* rowind, int64_t* order);

// Map original (i,j) coordinates (any triangle) to positions in blkval; -1 if outside V.
void index_map(const Symbolic& S, int64_t cnt, const int64_t* I, const int64_t* J, int64_t* out);

}  // namespace smcp
