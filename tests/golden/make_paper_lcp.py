#!/usr/bin/env python3
"""Adds the "suffix_tree" section to tests/golden/paper_example.json: the LCP array of the paper's worked example
and the suffix-tree answers it implies, derived from the figure's sorted keys ALONE -- by the definitions, with
string comparisons, not by the oracle, the builder or tests/naive.py:

  LCP[i]  = length of the longest common prefix of key(i - 1) and key(i), LCP[0] = 0          (paper.tex:600;
            the reference computes the same number as lcp_kmers * k + lcp_chars, path_graph.cpp:1204)
  psv(i)  = nearest j < i with LCP[j] < LCP[i];  nsv(i) = nearest j > i with LCP[j] < LCP[i]   (lcp.h:143-160)
  parent([sp, ep]) = the lexicographic range of all keys that share the first l characters of key(sp), where
            l = max(LCP[sp], LCP[ep + 1]) (LCP[n] = 0): the smallest LCP interval that properly contains the range
            (paper.tex:600-604; lcp.cpp:276-295); the root is its own parent
  depth([sp, ep])  = min LCP[sp + 1 .. ep] = the length of the prefix all keys of the range share
  rmq(sp, ep)      = leftmost position of the minimum of LCP[sp .. ep]

    python tests/golden/make_paper_lcp.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "paper_example.json")


def common_prefix(a, b):
    n = 0
    while n < len(a) and n < len(b) and a[n] == b[n]:
        n += 1
    return n


def derive_suffix_tree(keys, lcp, ranges):
    """psv / nsv of every position, parent of every range in `ranges`, depth and rmq cases: from the sorted keys and
    their LCP array by the definitions in the docstring (string comparisons only)."""
    n = len(keys)

    def psv(i):
        return next(([j, lcp[j]] for j in range(i - 1, -1, -1) if lcp[j] < lcp[i]), None) if i > 0 else None

    def nsv(i):
        return next(([j, lcp[j]] for j in range(i + 1, n) if lcp[j] < lcp[i]), None)

    def parent(sp, ep):
        if (sp, ep) == (0, n - 1):
            return {"range": [sp, ep], "parent": [0, n - 1], "left_lcp": 0, "right_lcp": 0, "node_lcp": 0}
        right = lcp[ep + 1] if ep + 1 < n else 0
        depth = max(lcp[sp], right)
        prefix = keys[sp][:depth]
        assert len(prefix) == depth
        members = [i for i in range(n) if keys[i][:depth] == prefix]
        lo, hi = members[0], members[-1]
        assert members == list(range(lo, hi + 1)) and lo <= sp and ep <= hi and (lo, hi) != (sp, ep)
        return {"range": [sp, ep], "parent": [lo, hi], "left_lcp": lcp[lo], "right_lcp": (lcp[hi + 1] if hi + 1 < n else 0),
                "node_lcp": depth}

    seen, parents = set(), []
    for r in ranges:
        if r not in seen:
            seen.add(r)
            parents.append(parent(*r))
    depth = [{"range": list(r), "depth": min(lcp[r[0] + 1: r[1] + 1])} for r in sorted(seen) if r[1] > r[0]]
    rmq = []
    for sp in range(n):
        for ep in range(sp, n, 3):
            window = lcp[sp: ep + 1]
            rmq.append({"range": [sp, ep], "pos": sp + window.index(min(window)), "value": min(window)})
    return {"lcp": lcp, "psv": [psv(i) for i in range(n)], "nsv": [nsv(i) for i in range(n)],
            "parent": parents, "depth": depth, "rmq": rmq}


def suffix_tree_of(gold):
    """The "suffix_tree" section for the example object `gold` (its keys and find() ranges); make_paper_example.py calls this."""
    keys = [node["key"] for node in gold["nodes"]]
    n = len(keys)
    order = gold["comp_order"]
    assert keys == sorted(keys, key=lambda k: [order.index(c) for c in k]), "the figure lists the keys in lexicographic order"
    lcp = [0] + [common_prefix(keys[i - 1], keys[i]) for i in range(1, n)]
    ranges = [(i, i) for i in range(n)] + [tuple(q["range"]) for q in gold["find"] if q["range"][0] <= q["range"][1]]
    ranges += [(2, 4), (2, 3), (9, 12), (10, 11), (13, 15), (14, 15), (1, 4), (5, 6), (7, 8), (0, n - 1), (0, 4), (9, 15)]
    st = derive_suffix_tree(keys, lcp, ranges)
    section = {"_source": "derived from the keys above by tests/golden/make_paper_lcp.py (definitions only; see its docstring)"}
    section.update(st)
    return section


def main():
    with open(PATH) as f:
        gold = json.load(f)
    gold["suffix_tree"] = suffix_tree_of(gold)
    st, lcp = gold["suffix_tree"], gold["suffix_tree"]["lcp"]
    text = json.dumps(gold, indent=2)
    with open(PATH, "w") as f:
        f.write(text + "\n")
    print("LCP =", lcp, ";", len(st["parent"]), "parent cases,", len(st["rmq"]), "rmq cases")


if __name__ == "__main__":
    main()
