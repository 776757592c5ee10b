"""Definitional builder: order-K maximally pruned de Bruijn graph of a small graph.

Follows the *definitions* of the paper (`paper/paper.tex:246-254` path graph,
`:288-299` pruning lemma / maximal pruning, `:534-557` GCSA encoding), not the
reference's disk-based prefix-doubling algorithm: enumerate every K-label of
every start position, build the trie, collapse every subtree whose K-labels
all have the same value set.  Exponential in the number of branches inside a
K-window, so for tests only; `builder.py` is the scalable equivalent and is
cross-checked against this one.
"""
import numpy as np

from .graphs import Graph, SIGMA
from .index_arrays import NodeTable, assemble, redundant_from_lcp


def k_labels(graph: Graph, K: int):
    """labels[v] = set of comp tuples of length K spelled by paths starting at position v."""
    N = graph.size
    comp = [int(c) for c in graph.comp]
    succ = [list(map(int, graph.successors(v))) for v in range(N)]
    cur = [{(comp[v],)} for v in range(N)]
    for _ in range(K - 1):
        nxt = []
        tails = [None] * N
        # labels of length d+1 from v = comp[v] + labels of length d from successors
        for v in range(N):
            s = set()
            for w in succ[v]:
                s |= cur[w]
            tails[v] = s
        for v in range(N):
            nxt.append({(comp[v],) + t for t in tails[v]})
        cur = nxt
    return cur


def path_nodes(graph: Graph, K: int):
    """Sorted list of (key, positions) of the maximally pruned order-K de Bruijn graph."""
    labels = k_labels(graph, K)
    table = {}
    for v, ls in enumerate(labels):
        for l in ls:
            table.setdefault(l, set()).add(v)
    items = sorted((l, frozenset(s)) for l, s in table.items())
    out = []

    def rec(lo, hi, depth):
        if depth >= 1:
            first = items[lo][1]
            if all(items[j][1] == first for j in range(lo + 1, hi)):
                out.append((items[lo][0][:depth], first))
                return
        assert depth < K, "distinct K-labels must differ"
        j = lo
        while j < hi:
            c = items[j][0][depth]
            k = j
            while k < hi and items[k][0][depth] == c:
                k += 1
            rec(j, k, depth + 1)
            j = k

    rec(0, len(items), 0)
    return out


def node_table(graph: Graph, K: int) -> NodeTable:
    nodes = path_nodes(graph, K)
    n = len(nodes)
    keys = [k for k, _ in nodes]
    preds = graph.predecessor_lists()
    comp = graph.comp
    pred_mask = np.zeros(n, dtype=np.uint8)
    outdeg = np.zeros(n, dtype=np.uint32)
    for j, (key, pos) in enumerate(nodes):
        cs = {int(comp[u]) for v in pos for u in preds[v]}
        for c in cs:
            pred_mask[j] |= 1 << c
            target = (c,) + key
            hits = [i for i, k in enumerate(keys)
                    if k == target[:len(k)] or target == k[:len(target)]]
            assert len(hits) == 1, "simplified encoding needs a unique predecessor per label"
            assert len(keys[hits[0]]) <= len(key) + 1
            outdeg[hits[0]] += 1
    lcp = np.zeros(n, dtype=np.uint8)
    for j in range(1, n):
        a, b = keys[j - 1], keys[j]
        l = 0
        while l < len(a) and l < len(b) and a[l] == b[l]:
            l += 1
        assert l < len(a) and l < len(b), "key set must be prefix-free"
        lcp[j] = l
    val_off = np.zeros(n + 1, dtype=np.uint64)
    vals = []
    for j, (_, pos) in enumerate(nodes):
        vs = sorted(int(graph.value[v]) for v in pos)
        assert len(set(vs)) == len(vs)
        vals.extend(vs)
        val_off[j + 1] = len(vals)
    vals = np.asarray(vals, dtype=np.uint64)
    red = redundant_from_lcp(lcp, val_off, vals)
    return NodeTable(order=K, pred_mask=pred_mask, outdeg=outdeg, lcp=lcp, val_off=val_off,
                     vals=vals, redundant=red,
                     key_len=np.asarray([len(k) for k in keys], dtype=np.uint16), keys=keys)


def build(graph: Graph, K: int, sample_period: int = 64, branching: int = 64):
    return assemble(node_table(graph, K), sample_period=sample_period, branching=branching)
