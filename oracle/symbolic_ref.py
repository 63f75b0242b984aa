"""Pure-Python reference symbolic analysis for SMALL cases (test infrastructure only).

Independent of the product's C++ symbolic: naive dense elimination for the filled pattern,
parent = first sub-diagonal nonzero, Pothen-Sun maximal supernodes, supernodal postorder.
Follows the same published algorithms chompack.symbolic uses ([EXT], solvers.py:305-314).
"""
import numpy as np


def symbolic_ref(n, edges, perm=None):
    """edges: iterable of (i, j) pairs (original coordinates). Returns dict of arrays in the
    shared layout (see oracle.Sym) plus 'p' (p[new] = orig)."""
    p = np.arange(n) if perm is None else np.asarray(perm)
    ip = np.empty(n, dtype=int)
    ip[p] = np.arange(n)
    F = np.eye(n, dtype=bool)
    for i, j in edges:
        a, b = ip[i], ip[j]
        F[a, b] = F[b, a] = True
    nnz_in = int(np.tril(F).sum())
    # naive symbolic elimination (fill)
    for k in range(n):
        nb = [i for i in range(k + 1, n) if F[i, k]]
        for a in nb:
            for b in nb:
                F[a, b] = True
    parent = -np.ones(n, dtype=int)
    for j in range(n):
        below = [i for i in range(j + 1, n) if F[i, j]]
        if below:
            parent[j] = below[0]
    # postorder the etree, relabel
    def postorder(par):
        m = len(par)
        ch = [[] for _ in range(m)]
        for v in range(m):
            if par[v] >= 0:
                ch[par[v]].append(v)
        out = []
        for r in range(m):
            if par[r] >= 0:
                continue
            st = [(r, 0)]
            while st:
                v, i = st.pop()
                if i < len(ch[v]):
                    st.append((v, i + 1))
                    st.append((ch[v][i], 0))
                else:
                    out.append(v)
        return np.array(out, dtype=int)

    post = postorder(parent)
    def relabel(order, F, p):
        return F[np.ix_(order, order)], p[order]
    F, p = relabel(post, F, p)
    n_ = n
    parent = -np.ones(n, dtype=int)
    cc = np.zeros(n, dtype=int)
    for j in range(n):
        below = [i for i in range(j + 1, n) if F[i, j]]
        cc[j] = 1 + len(below)
        if below:
            parent[j] = below[0]
    # maximal supernodes
    sn = -np.ones(n, dtype=int)
    pick = -np.ones(n, dtype=int)
    for c in range(n):
        j = parent[c]
        if j >= 0 and pick[j] < 0 and cc[c] == cc[j] + 1:
            pick[j] = c
    members = []
    for j in range(n):
        if pick[j] >= 0:
            sn[j] = sn[pick[j]]
            members[sn[j]].append(j)
        else:
            sn[j] = len(members)
            members.append([j])
    nsn = len(members)
    spar = np.array([sn[parent[m[-1]]] if parent[m[-1]] >= 0 else -1 for m in members], dtype=int)
    spost = postorder(spar)
    order2 = np.array([v for s in spost for v in members[s]], dtype=int)
    newsn = np.empty(nsn, dtype=int)
    newsn[spost] = np.arange(nsn)
    F, p = relabel(order2, F, p)
    snptr = np.zeros(nsn + 1, dtype=np.int64)
    for k, s in enumerate(spost):
        snptr[k + 1] = snptr[k] + len(members[s])
    snpar = -np.ones(nsn, dtype=np.int64)
    for s in range(nsn):
        if spar[s] >= 0:
            snpar[newsn[s]] = newsn[spar[s]]
    rows = []
    for k in range(nsn):
        f = snptr[k]
        rows.append(np.array([i for i in range(f, n) if F[i, f]], dtype=np.int32))
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    rowidx = np.concatenate(rows).astype(np.int32)
    nn = np.diff(snptr)
    nf = np.diff(rowptr)
    na = nf - nn
    sepptr = np.concatenate([[0], np.cumsum(na)]).astype(np.int64)
    blkptr = np.concatenate([[0], np.cumsum(nf * nn)]).astype(np.int64)
    updptr = np.concatenate([[0], np.cumsum(na * na)]).astype(np.int64)
    relidx = np.zeros(sepptr[-1], dtype=np.int32)
    for k in range(nsn):
        if snpar[k] >= 0:
            pr = list(rows[snpar[k]])
            for t, r in enumerate(rows[k][nn[k]:]):
                relidx[sepptr[k] + t] = pr.index(r)
    ch = [[] for _ in range(nsn)]
    for k in range(nsn):
        if snpar[k] >= 0:
            ch[snpar[k]].append(k)
    chptr = np.concatenate([[0], np.cumsum([len(c) for c in ch])]).astype(np.int64)
    chidx = np.array([c for l in ch for c in l], dtype=np.int64)
    return dict(n=n_, p=p.astype(np.int64), snptr=snptr, snpar=snpar, rowptr=rowptr, rowidx=rowidx,
                sepptr=sepptr, relidx=relidx, blkptr=blkptr, updptr=updptr, chptr=chptr, chidx=chidx,
                nnz=int(np.tril(F).sum()), fill=int(np.tril(F).sum()) - nnz_in)
