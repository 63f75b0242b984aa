// Host symbolic analysis. See symbolic.hpp for what it replaces in the reference.
#include "symbolic.hpp"
#include "switches.hpp"

#include <algorithm>
#include <numeric>
#include <set>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace smcp {
namespace {

// adjacency of the symmetric graph in "current" labels, strict lower part stored per
// column (rows > col) and strict upper part per column (rows < col).
struct Graph {
  int64_t n;
  std::vector<int64_t> lptr, uptr;
  std::vector<int32_t> lidx, uidx;
};

// Build Graph from input pattern with labels mapped through ip (orig -> current); only the halves asked for (the
// elimination tree and the column counts walk the upper lists, the clique row structures the lower ones).
static int build_graph(int64_t n, const int64_t* colptr, const int64_t* rowind,
                       const std::vector<int64_t>& ip, Graph& g, bool lower, bool upper, bool tidy_lists = true) {
  g.n = n;
  g.lptr.assign(n + 1, 0);
  g.uptr.assign(n + 1, 0);
  g.lidx.clear();
  g.uidx.clear();
  for (int64_t j = 0; j < n; ++j) {
    const int64_t b = ip[j];
    for (int64_t q = colptr[j]; q < colptr[j + 1]; ++q) {
      int64_t i = rowind[q];
      if (i < 0 || i >= n) return -1;
      if (i == j) continue;
      int64_t a = ip[i];
      int64_t lo = std::min(a, b), hi = std::max(a, b);
      g.lptr[lo + 1]++;  // column lo has row hi below the diagonal
      g.uptr[hi + 1]++;  // column hi has row lo above the diagonal
    }
  }
  for (int64_t j = 0; j < n; ++j) {
    g.lptr[j + 1] += g.lptr[j];
    g.uptr[j + 1] += g.uptr[j];
  }
  if (lower) g.lidx.resize(g.lptr[n]);
  if (upper) g.uidx.resize(g.uptr[n]);
  std::vector<int64_t> lw(g.lptr.begin(), g.lptr.end() - 1), uw(g.uptr.begin(), g.uptr.end() - 1);
  for (int64_t j = 0; j < n; ++j) {
    const int64_t b = ip[j];
    for (int64_t q = colptr[j]; q < colptr[j + 1]; ++q) {
      int64_t i = rowind[q];
      if (i == j) continue;
      int64_t a = ip[i];
      int64_t lo = std::min(a, b), hi = std::max(a, b);
      if (lower) g.lidx[lw[lo]++] = (int32_t)hi;
      if (upper) g.uidx[uw[hi]++] = (int32_t)lo;
    }
  }
  // sort + dedupe each list (duplicates are harmless for etree/counts but not for sizes)
  auto tidy = [n](std::vector<int64_t>& ptr, std::vector<int32_t>& idx) {
    std::vector<int64_t> nptr(n + 1, 0);
    int64_t w = 0;
    for (int64_t j = 0; j < n; ++j) {
      int64_t b = ptr[j], e = ptr[j + 1];
      if (!std::is_sorted(idx.begin() + b, idx.begin() + e)) std::sort(idx.begin() + b, idx.begin() + e);
      int64_t start = w;
      for (int64_t q = b; q < e; ++q)
        if (w == start || idx[w - 1] != idx[q]) idx[w++] = idx[q];
      nptr[j + 1] = w;
    }
    idx.resize(w);
    ptr.swap(nptr);
  };
  if (lower && tidy_lists) tidy(g.lptr, g.lidx);
  if (upper && tidy_lists) tidy(g.uptr, g.uidx);
  return 0;
}

static bool is_identity(const std::vector<int64_t>& order) {
  for (size_t k = 0; k < order.size(); ++k) if (order[k] != (int64_t)k) return false;
  return true;
}

// Liu's elimination tree with path compression.
static void etree(const Graph& g, std::vector<int64_t>& parent) {
  int64_t n = g.n;
  parent.assign(n, -1);
  std::vector<int64_t> anc(n, -1);
  for (int64_t c = 0; c < n; ++c) {
    for (int64_t q = g.uptr[c]; q < g.uptr[c + 1]; ++q) {
      int64_t i = g.uidx[q];
      while (i != -1 && i < c) {
        int64_t nx = anc[i];
        anc[i] = c;
        if (nx == -1) parent[i] = c;
        i = nx;
      }
    }
  }
}

// Postorder of a forest given parent[]; children visited in ascending label order.
static void postorder(const std::vector<int64_t>& parent, std::vector<int64_t>& post) {
  int64_t n = (int64_t)parent.size();
  std::vector<int64_t> head(n, -1), next(n, -1);
  for (int64_t j = n - 1; j >= 0; --j) {
    if (parent[j] >= 0) {
      next[j] = head[parent[j]];
      head[parent[j]] = j;
    }
  }
  post.clear();
  post.reserve(n);
  std::vector<int64_t> stack;
  for (int64_t r = 0; r < n; ++r) {
    if (parent[r] >= 0) continue;
    stack.push_back(r);
    while (!stack.empty()) {
      int64_t v = stack.back();
      int64_t c = head[v];
      if (c == -1) {
        post.push_back(v);
        stack.pop_back();
      } else {
        head[v] = next[c];
        stack.push_back(c);
      }
    }
  }
}

// Column counts of the Cholesky factor via row subtrees: O(|L|).
static void colcounts(const Graph& g, const std::vector<int64_t>& parent, std::vector<int64_t>& cc) {
  int64_t n = g.n;
  cc.assign(n, 1);
  std::vector<int64_t> mark(n, -1);
  for (int64_t i = 0; i < n; ++i) {
    mark[i] = i;
    for (int64_t q = g.uptr[i]; q < g.uptr[i + 1]; ++q) {
      int64_t j = g.uidx[q];
      while (mark[j] != i) {
        mark[j] = i;
        cc[j]++;
        j = parent[j];
      }
    }
  }
}

static void relabel(const std::vector<int64_t>& order /*new->cur*/, std::vector<int64_t>& p /*cur->orig becomes new->orig*/) {
  std::vector<int64_t> np(p.size());
  for (size_t k = 0; k < order.size(); ++k) np[k] = p[order[k]];
  p.swap(np);
}

}  // namespace

int symbolic_build(int64_t n, const int64_t* colptr, const int64_t* rowind, const int64_t* perm,
                   Symbolic& S) {
  if (n <= 0) return -1;
  const bool timing = sw_on("SMCP_TIMING", 0) == 1;
  auto tprev = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "smcp_amd timing: symbolic_build: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tprev).count());
    tprev = now;
  };
  S = Symbolic();
  S.n = n;
  S.p.resize(n);
  S.ip.resize(n);
  if (perm) {
    std::vector<char> seen(n, 0);
    for (int64_t i = 0; i < n; ++i) {
      if (perm[i] < 0 || perm[i] >= n || seen[perm[i]]) return -2;
      seen[perm[i]] = 1;
      S.p[i] = perm[i];
    }
  } else {
    std::iota(S.p.begin(), S.p.end(), 0);
  }
  for (int64_t i = 0; i < n; ++i) S.ip[S.p[i]] = i;

  // ---- pass 1: etree + postorder in the user ordering
  // (each half of the adjacency is built once: the upper lists here for tree and counts, the lower ones in pass 3)
  Graph g;
  if (build_graph(n, colptr, rowind, S.ip, g, false, true)) return -1;
  int64_t nnz_in = g.uptr[n] + n;
  mark("upper lists");
  std::vector<int64_t> parent, post, cc;
  etree(g, parent);
  postorder(parent, post);
  mark("etree + postorder");
  colcounts(g, parent, cc);
  mark("column counts");
  // ---- pass 2: in postorder labels: etree, column counts, maximal supernodes (Pothen-Sun).  A postorder is an equivalent
  // reordering (same filled graph, same elimination tree up to the labels): tree and counts are carried over, not recomputed
  if (!is_identity(post)) {
    std::vector<int64_t> ipost(n), par2(n), cc1(n);
    for (int64_t k = 0; k < n; ++k) ipost[post[k]] = k;
    for (int64_t k = 0; k < n; ++k) {
      const int64_t pv = parent[post[k]];
      par2[k] = pv < 0 ? -1 : ipost[pv];
      cc1[k] = cc[post[k]];
    }
    parent.swap(par2);
    cc.swap(cc1);
    relabel(post, S.p);
    for (int64_t i = 0; i < n; ++i) S.ip[S.p[i]] = i;
    mark("relabel");
  }
  // supernode membership: vertex j joins the supernode of a child c with cc[c] == cc[j]+1
  std::vector<int64_t> sn(n, -1), snlast;  // snlast[s] = last (largest) vertex of supernode s
  std::vector<int64_t> snfirst;
  {
    std::vector<int64_t> pick(n, -1);  // child chosen for merging
    for (int64_t c = 0; c < n; ++c) {
      int64_t j = parent[c];
      if (j >= 0 && pick[j] == -1 && cc[c] == cc[j] + 1) pick[j] = c;
    }
    for (int64_t j = 0; j < n; ++j) {
      if (pick[j] >= 0) {
        sn[j] = sn[pick[j]];
        snlast[sn[j]] = j;
      } else {
        sn[j] = (int64_t)snfirst.size();
        snfirst.push_back(j);
        snlast.push_back(j);
      }
    }
  }
  int64_t nsn = (int64_t)snfirst.size();
  // supernodal tree + its postorder
  std::vector<int64_t> spar(nsn, -1);
  for (int64_t s = 0; s < nsn; ++s) {
    int64_t pv = parent[snlast[s]];
    spar[s] = pv >= 0 ? sn[pv] : -1;
  }
  std::vector<int64_t> spost;
  postorder(spar, spost);
  // vertices of each supernode in ascending order (chain order)
  std::vector<int64_t> cnt(nsn + 1, 0);
  for (int64_t j = 0; j < n; ++j) cnt[sn[j] + 1]++;
  for (int64_t s = 0; s < nsn; ++s) cnt[s + 1] += cnt[s];
  std::vector<int64_t> members(n), w(cnt.begin(), cnt.end() - 1);
  for (int64_t j = 0; j < n; ++j) members[w[sn[j]]++] = j;
  std::vector<int64_t> order2;
  order2.reserve(n);
  S.nsn = nsn;
  S.snptr.assign(nsn + 1, 0);
  std::vector<int64_t> newsn(nsn);  // old supernode id -> new id
  for (int64_t k = 0; k < nsn; ++k) {
    int64_t s = spost[k];
    newsn[s] = k;
    for (int64_t q = cnt[s]; q < cnt[s + 1]; ++q) order2.push_back(members[q]);
    S.snptr[k + 1] = (int64_t)order2.size();
  }
  S.snpar.assign(nsn, -1);
  for (int64_t s = 0; s < nsn; ++s)
    if (spar[s] >= 0) S.snpar[newsn[s]] = newsn[spar[s]];
  std::vector<int64_t> cc2(n);
  for (int64_t k = 0; k < n; ++k) cc2[k] = cc[order2[k]];
  relabel(order2, S.p);
  for (int64_t i = 0; i < n; ++i) S.ip[S.p[i]] = i;
  S.snode.resize(n);
  for (int64_t k = 0; k < nsn; ++k)
    for (int64_t j = S.snptr[k]; j < S.snptr[k + 1]; ++j) S.snode[j] = k;

  // ---- pass 3: final labels: clique row structures
  mark("supernodes");
  if (build_graph(n, colptr, rowind, S.ip, g, true, false, false)) return -1;      // (the marks below skip repeated rows)
  mark("lower lists");
  S.chptr.assign(nsn + 1, 0);
  for (int64_t k = 0; k < nsn; ++k)
    if (S.snpar[k] >= 0) S.chptr[S.snpar[k] + 1]++;
  for (int64_t k = 0; k < nsn; ++k) S.chptr[k + 1] += S.chptr[k];
  S.chidx.resize(S.chptr[nsn]);
  {
    std::vector<int64_t> cw(S.chptr.begin(), S.chptr.end() - 1);
    for (int64_t k = 0; k < nsn; ++k)
      if (S.snpar[k] >= 0) S.chidx[cw[S.snpar[k]]++] = k;
  }
  S.rowptr.assign(nsn + 1, 0);
  for (int64_t k = 0; k < nsn; ++k) S.rowptr[k + 1] = S.rowptr[k] + cc2[S.snptr[k]];
  S.rowidx.resize(S.rowptr[nsn]);
  {
    std::vector<int64_t> mark(n, -1);
    std::vector<int32_t> tmp;
    for (int64_t k = 0; k < nsn; ++k) {
      int64_t f = S.snptr[k], l = S.snptr[k + 1] - 1;
      tmp.clear();
      for (int64_t j = f; j <= l; ++j) {
        for (int64_t q = g.lptr[j]; q < g.lptr[j + 1]; ++q) {
          int64_t r = g.lidx[q];
          if (r > l && mark[r] != k) {
            mark[r] = k;
            tmp.push_back((int32_t)r);
          }
        }
      }
      for (int64_t q = S.chptr[k]; q < S.chptr[k + 1]; ++q) {
        int64_t c = S.chidx[q];
        int64_t cb = S.rowptr[c] + (S.snptr[c + 1] - S.snptr[c]), ce = S.rowptr[c + 1];
        for (int64_t t = cb; t < ce; ++t) {
          int64_t r = S.rowidx[t];
          if (r > l && mark[r] != k) {
            mark[r] = k;
            tmp.push_back((int32_t)r);
          }
        }
      }
      std::sort(tmp.begin(), tmp.end());
      int64_t nn = l - f + 1;
      if ((int64_t)tmp.size() + nn != S.rowptr[k + 1] - S.rowptr[k]) return -3;  // internal inconsistency
      int64_t o = S.rowptr[k];
      for (int64_t j = f; j <= l; ++j) S.rowidx[o++] = (int32_t)j;
      for (auto r : tmp) S.rowidx[o++] = r;
    }
  }
  mark("clique rows");
  // ---- relative indices, block/update pointers, ccs pointers
  S.sepptr.assign(nsn + 1, 0);
  S.blkptr.assign(nsn + 1, 0);
  S.updptr.assign(nsn + 1, 0);
  S.updpptr.assign(nsn + 1, 0);
  for (int64_t k = 0; k < nsn; ++k) {
    int64_t nn = S.nn(k), nf = S.nf(k), na = nf - nn;
    S.sepptr[k + 1] = S.sepptr[k] + na;
    S.blkptr[k + 1] = S.blkptr[k] + nf * nn;
    S.updptr[k + 1] = S.updptr[k] + na * na;
    S.max_nn = std::max(S.max_nn, nn);
    S.max_na = std::max(S.max_na, na);
    S.max_front = std::max(S.max_front, nf);
  }
  // Packed update blocks in the child -> parent exchange buffer: the children of one parent side by side (in the order
  // of its child list), so that a parent reads its children's blocks as ONE contiguous run and a level of siblings
  // writes one -- in clique (postorder) order the blocks of the 112 children of a (64,128) front of synth50k were 16.6 KB
  // pieces at 48 KB strides (their own children's blocks in between), which HBM serves at ~2.8 TB/s.
  // updpptr[k] is the OFFSET of clique k (not monotone in k); updpptr[nsn] the total.
  {
    int64_t off = 0;
    for (int64_t p = 0; p < nsn; ++p)
      for (int64_t q = S.chptr[p]; q < S.chptr[p + 1]; ++q) {
        const int64_t k = S.chidx[q], na = S.na(k);
        S.updpptr[k] = off;
        off += na * (na + 1) / 2;
      }
    for (int64_t k = 0; k < nsn; ++k)
      if (S.snpar[k] < 0) S.updpptr[k] = off;       // roots have no separator: an empty block at the end
    S.updpptr[nsn] = off;
  }
  S.relidx.resize(S.sepptr[nsn]);
  for (int64_t k = 0; k < nsn; ++k) {
    int64_t pk = S.snpar[k];
    int64_t na = S.na(k);
    if (pk < 0) {
      if (na != 0) return -4;
      continue;
    }
    const int32_t* a = &S.rowidx[S.rowptr[k] + S.nn(k)];
    const int32_t* pr = &S.rowidx[S.rowptr[pk]];
    int64_t pnf = S.nf(pk), t = 0;
    for (int64_t i = 0; i < na; ++i) {
      while (t < pnf && pr[t] < a[i]) ++t;
      if (t == pnf || pr[t] != a[i]) return -5;  // separator not contained in parent clique
      S.relidx[S.sepptr[k] + i] = (int32_t)t;
    }
  }
  S.ccsptr.assign(n + 1, 0);
  for (int64_t k = 0; k < nsn; ++k) {
    int64_t nf = S.nf(k);
    for (int64_t t = 0; t < S.nn(k); ++t) S.ccsptr[S.snptr[k] + t + 1] = nf - t;
  }
  for (int64_t j = 0; j < n; ++j) S.ccsptr[j + 1] += S.ccsptr[j];
  S.nnz = S.ccsptr[n];
  S.fill = S.nnz - nnz_in;
  // ---- levels (height based)
  S.level.assign(nsn, 0);
  for (int64_t k = 0; k < nsn; ++k) {
    int64_t pk = S.snpar[k];
    if (pk >= 0) S.level[pk] = std::max(S.level[pk], S.level[k] + 1);
  }
  S.nlev = 0;
  for (int64_t k = 0; k < nsn; ++k) S.nlev = std::max(S.nlev, S.level[k] + 1);
  S.levptr.assign(S.nlev + 1, 0);
  for (int64_t k = 0; k < nsn; ++k) S.levptr[S.level[k] + 1]++;
  for (int64_t l = 0; l < S.nlev; ++l) S.levptr[l + 1] += S.levptr[l];
  S.levidx.resize(nsn);
  {
    std::vector<int64_t> lw(S.levptr.begin(), S.levptr.end() - 1);
    for (int64_t k = 0; k < nsn; ++k) S.levidx[lw[S.level[k]]++] = k;
  }
  mark("relative indices, levels");
  return 0;
}

void maxcardsearch(int64_t n, const int64_t* colptr, const int64_t* rowind, int64_t* order) {
  std::vector<int64_t> id(n);
  std::iota(id.begin(), id.end(), 0);
  Graph g;
  build_graph(n, colptr, rowind, id, g, true, true);
  // bucket structure over cardinalities
  std::vector<int64_t> card(n, 0), head(n + 1, -1), nxt(n, -1), prv(n, -1);
  std::vector<char> done(n, 0);
  auto push = [&](int64_t v, int64_t c) {
    nxt[v] = head[c];
    prv[v] = -1;
    if (head[c] >= 0) prv[head[c]] = v;
    head[c] = v;
  };
  auto pop = [&](int64_t v, int64_t c) {
    if (prv[v] >= 0) nxt[prv[v]] = nxt[v]; else head[c] = nxt[v];
    if (nxt[v] >= 0) prv[nxt[v]] = prv[v];
  };
  for (int64_t v = n - 1; v >= 0; --v) push(v, 0);
  int64_t top = 0;
  for (int64_t k = n - 1; k >= 0; --k) {
    while (top > 0 && head[top] < 0) --top;
    int64_t v = head[top];
    pop(v, top);
    done[v] = 1;
    order[k] = v;  // visited first = eliminated last
    auto bump = [&](int64_t u) {
      if (done[u]) return;
      pop(u, card[u]);
      card[u]++;
      push(u, card[u]);
      if (card[u] > top) top = card[u];
    };
    for (int64_t q = g.lptr[v]; q < g.lptr[v + 1]; ++q) bump(g.lidx[q]);
    for (int64_t q = g.uptr[v]; q < g.uptr[v + 1]; ++q) bump(g.uidx[q]);
  }
}

// Minimum degree on the explicit elimination graph, ties to the smaller vertex.  Up to 32768 vertices the adjacency is one bit
// row per vertex (eliminating v ORs its row into its neighbours' rows: deg(v) * n / 64 word operations, where ordered sets
// of neighbours paid deg(v)^2 log n insertions -- 1.9 s of the 3.1 s interior-point run on the 1000-node max-cut graph,
// whose last cliques have 500 members); larger graphs keep the sets.  Both hold the same graph, so the order is the same.
static void mindegree_bits(int64_t n, const Graph& g, int64_t* order) {
  const int64_t W = (n + 63) / 64;
  std::vector<uint64_t> adj((size_t)(n * W), 0);
  auto row = [&](int64_t v) { return adj.data() + (size_t)(v * W); };
  auto setbit = [&](int64_t v, int64_t u) { row(v)[u >> 6] |= (uint64_t)1 << (u & 63); };
  for (int64_t v = 0; v < n; ++v) {
    for (int64_t q = g.lptr[v]; q < g.lptr[v + 1]; ++q) setbit(v, g.lidx[q]);
    for (int64_t q = g.uptr[v]; q < g.uptr[v + 1]; ++q) setbit(v, g.uidx[q]);
  }
  auto count = [&](int64_t v) {
    int64_t c = 0;
    const uint64_t* r = row(v);
    for (int64_t w = 0; w < W; ++w) c += __builtin_popcountll(r[w]);
    return c;
  };
  std::vector<int64_t> deg((size_t)n);
  std::set<std::pair<int64_t, int64_t>> pq;  // (degree, vertex)
  for (int64_t v = 0; v < n; ++v) { deg[(size_t)v] = count(v); pq.insert({deg[(size_t)v], v}); }
  std::vector<int32_t> nb;
  for (int64_t k = 0; k < n; ++k) {
    auto it = pq.begin();
    const int64_t v = it->second;
    pq.erase(it);
    order[k] = v;
    uint64_t* rv = row(v);
    nb.clear();
    for (int64_t w = 0; w < W; ++w)
      for (uint64_t bits = rv[w]; bits; bits &= bits - 1) nb.push_back((int32_t)(64 * w + __builtin_ctzll(bits)));
    for (int32_t u : nb) {
      pq.erase({deg[(size_t)u], (int64_t)u});
      uint64_t* ru = row(u);
      for (int64_t w = 0; w < W; ++w) ru[w] |= rv[w];
      ru[v >> 6] &= ~((uint64_t)1 << (v & 63));
      ru[u >> 6] &= ~((uint64_t)1 << (u & 63));
      deg[(size_t)u] = count(u);
      pq.insert({deg[(size_t)u], (int64_t)u});
    }
    for (int64_t w = 0; w < W; ++w) rv[w] = 0;
  }
}
void mindegree(int64_t n, const int64_t* colptr, const int64_t* rowind, int64_t* order) {
  std::vector<int64_t> id(n);
  std::iota(id.begin(), id.end(), 0);
  Graph g;
  build_graph(n, colptr, rowind, id, g, true, true);
  if (n <= 32768) { mindegree_bits(n, g, order); return; }
  std::vector<std::set<int32_t>> adj(n);
  for (int64_t v = 0; v < n; ++v) {
    for (int64_t q = g.lptr[v]; q < g.lptr[v + 1]; ++q) adj[v].insert(g.lidx[q]);
    for (int64_t q = g.uptr[v]; q < g.uptr[v + 1]; ++q) adj[v].insert(g.uidx[q]);
  }
  std::set<std::pair<int64_t, int64_t>> pq;  // (degree, vertex)
  for (int64_t v = 0; v < n; ++v) pq.insert({(int64_t)adj[v].size(), v});
  std::vector<char> gone(n, 0);
  for (int64_t k = 0; k < n; ++k) {
    auto it = pq.begin();
    int64_t v = it->second;
    pq.erase(it);
    gone[v] = 1;
    order[k] = v;
    std::vector<int32_t> nb(adj[v].begin(), adj[v].end());
    for (int32_t u : nb) {
      pq.erase({(int64_t)adj[u].size(), (int64_t)u});
      adj[u].erase((int32_t)v);
    }
    for (size_t a = 0; a < nb.size(); ++a)
      for (size_t b = a + 1; b < nb.size(); ++b) {
        adj[nb[a]].insert(nb[b]);
        adj[nb[b]].insert(nb[a]);
      }
    for (int32_t u : nb) pq.insert({(int64_t)adj[u].size(), (int64_t)u});
    adj[v].clear();
  }
}

// entries e0 .. e1-1; an entry that follows another one of the same column further down (the usual order of a
// compressed-column input) is looked for right after it before the binary search
static void index_map_range(const Symbolic& S, int64_t e0, int64_t e1, const int64_t* I, const int64_t* J, int64_t* out) {
  int64_t pc = -1, pr = -1, pq = 0;          // previous hit: column, row, position in the clique's row list
  for (int64_t e = e0; e < e1; ++e) {
    out[e] = -1;
    if (I[e] < 0 || I[e] >= S.n || J[e] < 0 || J[e] >= S.n) { pc = -1; continue; }
    int64_t a = S.ip[I[e]], b = S.ip[J[e]];
    int64_t c = std::min(a, b), r = std::max(a, b);
    int64_t k = S.snode[c];
    const int32_t* rb = &S.rowidx[S.rowptr[k]];
    int64_t nf = S.nf(k);
    const int32_t* pos;
    if (c == pc && r > pr && pq + 1 < nf && rb[pq + 1] == r) pos = rb + pq + 1;
    else pos = std::lower_bound(rb, rb + nf, (int32_t)r);
    if (pos == rb + nf || *pos != r) { pc = -1; continue; }
    pc = c; pr = r; pq = pos - rb;
    out[e] = S.blkptr[k] + (c - S.snptr[k]) * nf + (pos - rb);
  }
}
void index_map(const Symbolic& S, int64_t cnt, const int64_t* I, const int64_t* J, int64_t* out) {
  const int64_t per = 1 << 17;
  int nth = (int)std::min<int64_t>(std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency())), (cnt + per - 1) / per);
  if (nth <= 1) { index_map_range(S, 0, cnt, I, J, out); return; }
  std::vector<std::thread> th;
  const int64_t chunk = (cnt + nth - 1) / nth;
  for (int t = 1; t < nth; ++t)
    th.emplace_back([&, t] { index_map_range(S, t * chunk, std::min(cnt, (t + 1) * chunk), I, J, out); });
  index_map_range(S, 0, std::min(cnt, chunk), I, J, out);
  for (auto& x : th) x.join();
}

}  // namespace smcp
