"""CPU tests pinning the oracle (oracle/gcsa_oracle.c).

1. the paper's worked example (the only known-answer material in the reference tree),
2. differential tests against tests/naive.py (array-level and input-graph-level brute force)
   on seeded random graphs incl. bubbles, cycles, `N`s and repeated labels,
3. the invariants `verifyIndex` asserts (reference src/algorithms.cpp:131-275).
"""
import os

import numpy as np
import pytest

from workload import graphs
from workload.brute_builder import build, node_table
from workload.index_arrays import unpack_bits
from workload.rng import SplitMix64
from oracle.oracle import OracleIndex
from gcsa2_amd.hostview import concat_patterns
from naive import NaiveIndex, GraphBrute

COMP2CHAR = "$ACGTN#"
UNKNOWN = (1 << 64) - 1


def bits_str(words, n):
    return "".join("1" if b else "0" for b in unpack_bits(words, n))


# ---------------------------------------------------------------------------------------------
# 1. paper example

@pytest.mark.skipif(not os.path.isdir("/root/reference/paper"), reason="the figures are in the reference tree (build container only)")
def test_golden_vectors_are_regenerable():
    """The whole parity pin comes out of committed scripts: the three generators re-read the paper's figures
    (paper/gcsa2_graph_dbg.ipe, gcsa2_pruned_index.ipe, gcsa2_text_indexes.ipe; paper.tex:147-151, 534-557) by the coordinates
    of their objects and reproduce the committed JSON byte for byte -- nothing of it was typed in."""
    import importlib.util
    import json
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(golden, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    with open(os.path.join(golden, "paper_example.json")) as f:
        committed = f.read()
    assert load("make_paper_example").render() == committed
    # make_paper_lcp.py alone (the suffix-tree section from the keys) gives the same section
    gold = json.loads(committed)
    assert load("make_paper_lcp").suffix_tree_of(gold) == gold["suffix_tree"]
    # Figure 1: the generator writes its file; run it on a copy of the module's output path
    text_mod = load("make_text_example")
    with open(os.path.join(golden, "text_example.json")) as f:
        committed_text = f.read()
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        text_mod.OUT = os.path.join(tmp, "text_example.json")
        text_mod.main()
        with open(text_mod.OUT) as f:
            assert f.read() == committed_text


def test_builder_reproduces_paper_figure(paper):
    g = graphs.paper_graph()
    ix = build(g, paper["order"], sample_period=1 << 40)
    t = ix.table
    assert ix.n == len(paper["nodes"]) == 16 and ix.e == 20
    for i, node in enumerate(paper["nodes"]):
        key = "".join(COMP2CHAR[c] for c in t.keys[i])
        # the figure keeps the sink as `$$$`; maximal pruning names it `$`
        assert key == node["key"] or (node["key"] == "$$$" and key == "$")
        vals = [int(v) for v in t.vals[int(t.val_off[i]):int(t.val_off[i + 1])]]
        assert vals == node["values"]
        bwt = "".join(COMP2CHAR[c] for c in range(7) if (int(t.pred_mask[i]) >> c) & 1)
        assert sorted(bwt) == sorted(node["bwt"])
        assert int(t.outdeg[i]) == node["outdegree"]
    assert [int(x) for x in ix.C] == paper["C"]
    assert bits_str(ix.edges, ix.e) == paper["OUT"]
    assert bits_str(ix.sampled_paths, ix.n) == paper["B_S"]
    assert bits_str(ix.samples, ix.sample_count) == paper["B_V"]
    assert [int(v) for v in ix.stored_samples_plain] == paper["V_S"]


def test_oracle_on_paper_example(paper):
    ix = build(graphs.paper_graph(), paper["order"], sample_period=1 << 40)
    o = OracleIndex(ix)
    for q in paper["find"]:
        assert list(o.find(q["pattern"].encode())) == q["range"], q
    for q in paper["locate"]:
        rng = tuple(q["range"])
        assert list(o.locate(rng)) == q["values"]
        assert o.count(rng) == q["count"]
    # the red arrows of Figure 3: one LF step with A from find("T") gives find("AT")
    assert o.LF((9, 12), 1) == (2, 4)
    assert o.find(b"") == (0, 15)


def check_paper_suffix_tree(paper, lcp_values, parent, depth, psv, nsv, rmq, not_found):
    """Shared by the oracle (CPU) and the engine (GPU): the answers tests/golden/make_paper_lcp.py derived from
    the figure's keys by the definitions."""
    st = paper["suffix_tree"]
    assert list(lcp_values) == st["lcp"]
    for i, (p, n) in enumerate(zip(st["psv"], st["nsv"])):
        assert psv(i) == (tuple(p) if p is not None else not_found), ("psv", i)
        assert nsv(i) == (tuple(n) if n is not None else not_found), ("nsv", i)
    for case in st["parent"]:
        got = parent(tuple(case["range"]))
        assert got == (case["parent"][0], case["parent"][1], case["left_lcp"], case["right_lcp"], case["node_lcp"]), case
        if case["parent"][1] > case["parent"][0]:
            assert depth(tuple(case["parent"])) == case["node_lcp"], case
    for case in st["depth"]:
        assert depth(tuple(case["range"])) == case["depth"], case
    for case in st["rmq"]:
        assert rmq(*case["range"]) == (case["pos"], case["value"]), case


def test_oracle_on_paper_suffix_tree(paper):
    """LCPArray family pinned on reference-held material: the LCP array of the worked example follows from the
    figure's sorted keys (definition at paper.tex:600, path_graph.cpp:1204), and parent / depth / psv / nsv / rmq
    follow from it (paper.tex:600-604) -- for several tree shapes (branching 2, 3, 4, 64)."""
    for branching in (2, 3, 4, 64):
        ix = build(graphs.paper_graph(), paper["order"], sample_period=1 << 40, branching=branching)
        o = OracleIndex(ix)
        values = int(ix.lcp_offsets[-1])
        check_paper_suffix_tree(paper, [int(x) for x in ix.lcp_data[: ix.n]], o.parent, o.depth, o.psv, o.nsv, o.rmq, (values, values))


# ---------------------------------------------------------------------------------------------
# 1b. the paper's text-index figure (GCATCATA$): a GCSA of a text is its FM-index

TEXT_NODE_LEN = 3


def text_example_index(fig, sample_period=1 << 40, branching=2):
    """Index of the path graph of the figure's text: vg-style nodes of 3 bases, order high enough for unique keys."""
    seq = graphs.default_char2comp()[np.frombuffer(fig["text"][:-1].encode(), dtype=np.uint8)]
    return build(graphs.linear_graph(len(seq), 0, node_len=TEXT_NODE_LEN, sequence=seq), 16, sample_period=sample_period,
                 branching=branching)


def text_position(value, fig):
    """Text position of a located node_type: (id, offset) of a base, or the sink node = the final '$'."""
    node, offset = int(value) >> graphs.ID_OFFSET, int(value) & ((1 << graphs.OFFSET_BITS) - 1)
    pos = (node - 1) * TEXT_NODE_LEN + offset
    return pos if pos < len(fig["text"]) - 1 else len(fig["text"]) - 1


def check_text_figure(fig, size, pred_char, lcp_values, lf_node, locate, find, suffix_tree_ops):
    """Shared by the oracle (CPU) and the engine (GPU): every column of the figure, and what follows from them."""
    g = fig["gcsa"]
    n = g["path_nodes"]
    assert size == n == len(fig["suffixes"]) + 1
    assert [pred_char(i) for i in range(n)] == g["BWT"]                       # BWT column (+ the source-marker row)
    assert [lf_node(i) for i in range(n)] == g["LF"]                          # LF column
    assert [lf_node(i) for i in range(9) if fig["SA"][i] != 0] == [fig["LF"][i] for i in range(9) if fig["SA"][i] != 0]
    for i in range(9):                                                        # SA column
        assert [text_position(v, fig) for v in locate((i, i))] == [fig["SA"][i]], i
    for q in g["find"]:                                                       # rows whose suffix starts with the pattern
        rng = find(q["pattern"].encode())
        assert list(rng) == q["range"], q
        assert sorted(text_position(v, fig) for v in locate(rng)) == q["positions"], q
    for x in g["absent"]:
        sp, ep = find(x.encode())
        assert sp + 1 > ep + 1 or sp > ep, x                                  # empty (utils.h:93-96)
    st = {"suffix_tree": g["suffix_tree"]}
    assert list(lcp_values)[:9] == fig["LCP"]                                 # LCP column, as printed
    check_paper_suffix_tree(st, lcp_values, *suffix_tree_ops)


def oracle_pred_char(ix):
    return lambda i: "".join(COMP2CHAR[c] for c in range(int(ix.sigma)) if (int(ix.bwt[c][i >> 6]) >> (i & 63)) & 1)


def test_oracle_on_text_figure(text_figure):
    """Second reference-held known-answer instance: Figure 1 (paper.tex:147-151).  Pins LF, locate and -- the part the
    GCSA figure does not carry -- a printed LCP array, with parent / depth / psv / nsv / rmq derived from it."""
    for branching in (2, 3, 64):
        ix = text_example_index(text_figure, branching=branching)
        o = OracleIndex(ix)
        values = int(ix.lcp_offsets[-1])
        check_text_figure(text_figure, ix.n, oracle_pred_char(ix), [int(x) for x in ix.lcp_data[: ix.n]], o.LF, o.locate, o.find,
                          (o.parent, o.depth, o.psv, o.nsv, o.rmq, (values, values)))
    # with samples only every 4th position the located positions are the same
    ix = text_example_index(text_figure, sample_period=4)
    o = OracleIndex(ix)
    for i in range(9):
        assert [text_position(v, text_figure) for v in o.locate((i, i))] == [text_figure["SA"][i]]


# ---------------------------------------------------------------------------------------------
# 2. differential tests

def small_cases():
    cases = [("paper", graphs.paper_graph(), 3)]
    for seed, n, K, pb, back in [(1, 30, 4, 0.25, 0.0), (2, 40, 5, 0.3, 0.0), (3, 25, 3, 0.2, 0.1),
                                 (4, 60, 6, 0.15, 0.0), (5, 35, 8, 0.2, 0.05), (6, 50, 2, 0.3, 0.0),
                                 (7, 12, 16, 0.3, 0.0)]:
        cases.append((f"rand{seed}", graphs.random_graph(n, 0xABC0 + seed, p_branch=pb, p_back=back,
                                                         alphabet=2 + seed % 3), K))
    cases.append(("linear", graphs.linear_graph(200, 0x51, node_len=8), 6))
    cases.append(("snp", graphs.snp_graph(150, 0x52, 0x53, snp_period=8, node_len=8), 6))
    return cases


CASES = small_cases()


@pytest.fixture(scope="module", params=range(len(CASES)), ids=[c[0] for c in CASES])
def case(request):
    name, g, K = CASES[request.param]
    ix = build(g, K, sample_period=8, branching=4)
    return name, g, K, ix, OracleIndex(ix), NaiveIndex(ix), GraphBrute(g)


def random_patterns(g, K, seed, count):
    """Half walks through the graph (hits), half random strings; lengths 1..K+2."""
    rng = SplitMix64(seed)
    pats = []
    for i in range(count):
        L = 1 + rng.below(K + 2)
        if i % 2 == 0:
            v = rng.below(g.size)
            s = []
            for _ in range(L):
                s.append(COMP2CHAR[int(g.comp[v])])
                succ = g.successors(v)
                v = int(succ[rng.below(len(succ))])
            pats.append("".join(s).encode())
        else:
            pats.append("".join("ACGTN"[rng.below(5)] for _ in range(L)).encode())
    return pats


def truncate_at_sink(p):
    """verifyIndex ends a k-mer at its first `$` (reference src/algorithms.cpp:127-129)."""
    k = p.find(b"$")
    return p if k < 0 else p[:k + 1]


def test_find_lf_vs_naive(case):
    name, g, K, ix, o, nv, gb = case
    pats = [truncate_at_sink(p) for p in random_patterns(g, K, 0x77, 300)] + [b"", b"A", b"N", b"#", b"$"]
    for p in pats:
        assert o.find(p) == nv.find(p), (name, p)
    # batched driver, serial and OpenMP
    data, off = concat_patterns(pats)
    r1 = o.find_batch(data, off, threads=1)
    r4 = o.find_batch(data, off, threads=4)
    assert np.array_equal(r1, r4)
    for i, p in enumerate(pats):
        assert tuple(int(x) for x in r1[i]) == nv.find(p)
    # single LF steps on all non-empty char ranges and comps
    for c in range(ix.sigma):
        rng = nv.charRange(c) if ix.C[c + 1] > 0 else None
        if rng is None:
            continue
        assert o.charRange(c) == rng
        if nv.empty(*rng):
            continue
        for c2 in range(ix.sigma):
            assert o.LF(rng, c2) == nv.LF(rng, c2)
    for i in range(ix.n):
        assert o.LF(i) == nv.LF1(i)


def test_find_vs_input_graph(case):
    """No false negatives / no short false positives (paper.tex:270-283): for |X| <= K,
    locate(find(X)) is exactly the set of start positions of paths labelled X."""
    name, g, K, ix, o, nv, gb = case
    c2c = ix.char2comp
    for p in random_patterns(g, K, 0x99, 200):
        p = truncate_at_sink(p)[:K]
        comps = [int(c2c[b]) for b in p]
        rng = o.find(p)
        expected = gb.occurrences(comps)
        got = [int(v) for v in o.locate(rng)]
        assert got == expected, (name, p, rng)
        assert o.count(rng) == len(expected), (name, p, rng)
        # the range is the set of nodes whose key prefix-matches X (Lemma "context length")
        if expected:
            keys = ix.table.keys
            match = [i for i, k in enumerate(keys)
                     if tuple(comps[:len(k)]) == k[:len(comps)]]
            assert match == list(range(rng[0], rng[1] + 1)), (name, p)
        else:
            assert nv.empty(*rng)


def test_locate_count_vs_naive(case):
    name, g, K, ix, o, nv, gb = case
    rng = SplitMix64(0x1234)
    ranges = [(i, i) for i in range(ix.n)]
    for _ in range(100):
        a = rng.below(ix.n)
        b = min(ix.n - 1, a + rng.below(6))
        ranges.append((a, b))
    ranges += [(0, ix.n - 1), (1, 0), (3, 2), (0, ix.n), (ix.n, ix.n + 3)]
    for r in ranges:
        assert [int(v) for v in o.locate(r)] == nv.locate(r), (name, r)
        assert [int(v) for v in o.locate(r, sort=False)] == nv.locate(r, sort=False)
        assert o.count(r) == nv.count(r), (name, r)
    arr = np.array(ranges, dtype=np.uint64)
    offs, vals = o.locate_batch(arr, threads=3)
    for i, r in enumerate(ranges):
        assert [int(v) for v in vals[int(offs[i]):int(offs[i + 1])]] == nv.locate(r)
    assert [int(c) for c in o.count_batch(arr, threads=2)] == [nv.count(r) for r in ranges]
    for i in range(ix.n):
        assert o.sampled(i) == bool(nv.SP[i])
    for j in range(ix.sample_count):
        assert o.sample(j) == nv.VS[j] and o.lastSample(j) == bool(nv.SM[j])


def test_suffix_tree_ops_vs_naive(case):
    name, g, K, ix, o, nv, gb = case
    for pos in range(ix.n + 2):
        assert o.psv(pos) == nv.psv(pos), (name, pos)
        assert o.psev(pos) == nv.psv(pos, equal=True)
        assert o.nsv(pos) == nv.nsv(pos)
        assert o.nsev(pos) == nv.nsv(pos, equal=True)
    rng = SplitMix64(0x4321)
    ranges = [(i, i) for i in range(ix.n)] + [(0, ix.n - 1)]
    for _ in range(200):
        a = rng.below(ix.n)
        b = min(ix.n - 1, a + rng.below(ix.n))
        ranges.append((a, b))
    for r in ranges:
        assert o.rmq(*r) == nv.rmq(*r), (name, r)
        assert o.parent(r) == nv.parent(r), (name, r)
        assert o.depth(r) == nv.depth(r), (name, r)
    assert o.rmq(3, 2) == nv.rmq(3, 2)
    arr = np.array(ranges, dtype=np.uint64)
    pb = o.parent_batch(arr, threads=2)
    db = o.depth_batch(arr, threads=2)
    for i, r in enumerate(ranges):
        assert tuple(int(x) for x in pb[i]) == nv.parent(r)
        assert int(db[i]) == nv.depth(r)


def test_lf_fast_all(case):
    name, g, K, ix, o, nv, gb = case
    rng = SplitMix64(0x55)
    ranges = [(i, i) for i in range(ix.n)] + [(0, ix.n - 1), (1, 0)]
    for _ in range(50):
        a = rng.below(ix.n)
        ranges.append((a, min(ix.n - 1, a + 1 + rng.below(5))))
    for r in ranges:
        for all_, limit in ((0, ix.fast_chars), (1, ix.sigma - 2)):
            got = o.LF_all(r) if all_ else o.LF_fast(r)
            for c in range(ix.sigma):
                if c < 1 or c > limit or nv.empty(*r):
                    assert got[c] == (1, 0)
                elif r[0] == r[1]:
                    exp = nv.LF(r, c) if nv.B[c][r[0]] else (1, 0)
                    assert got[c] == exp
                else:
                    assert got[c] == nv.LF(r, c)


# ---------------------------------------------------------------------------------------------
# 3. verifyIndex invariants (reference src/algorithms.cpp:131-275)

def test_verify_index_invariants(case):
    name, g, K, ix, o, nv, gb = case
    c2c = ix.char2comp
    seen = set()
    for p in random_patterns(g, K, 0x31, 200):
        p = truncate_at_sink(p)[:K]
        if p in seen:
            continue
        seen.add(p)
        rng = o.find(p)
        if nv.empty(*rng):
            continue
        # parent() == re-searching successively shorter prefixes until the range changes
        par = o.parent(rng)
        end = len(p)
        q = rng
        while q == rng:
            end -= 1
            q = o.find(p[:end])
        assert (par[0], par[1]) == q, (name, p)
        assert par[4] == end, (name, p)
        assert o.depth((par[0], par[1])) == par[4]
        # count == |locate| == distinct start nodes; locate(range, 10) is a sorted subset
        occ = [int(v) for v in o.locate(rng)]
        assert o.count(rng) == len(occ)
        sub = [int(v) for v in o.locate(rng, max_positions=3)]
        assert len(sub) == min(3, len(occ)) and sub == sorted(sub) and set(sub) <= set(occ)


def test_count_kmers_vs_input_graph(case):
    """countKMers == number of distinct base-only k-mers spelled by paths of the input graph."""
    import itertools
    name, g, K, ix, o, nv, gb = case
    for k in range(0, min(K, 5) + 1):
        want = 1 if k == 0 else sum(1 for t in itertools.product([1, 2, 3, 4], repeat=k) if gb.starts(list(t)))
        assert o.count_kmers(k) == want, (name, k)
        assert o.count_kmers(k, threads=3) == want
    assert o.count_kmers(K + 1) == 0


def test_match_stats_vs_input_graph(case):
    """LF + parent interplay (paper.tex:344): ms[i] = longest prefix of P[i:] that labels a path of
    the input graph, for lengths up to the order of the index."""
    name, g, K, ix, o, nv, gb = case
    pats = [p for p in random_patterns(g, 2 * K, 0x91, 40) if b"$" not in p]
    data, off = concat_patterns(pats)
    ms, rng, fb = o.match_stats_batch(data, off, threads=2)
    for q, p in enumerate(pats):
        comps = [int(ix.char2comp[b]) for b in p]
        for i in range(len(p)):
            L = 0
            while i + L < len(p) and L < K and gb.starts(comps[i:i + L + 1]):
                L += 1
            assert min(int(ms[int(off[q]) + i]), K) == L, (name, p, i)


def test_compare_kmers_vs_input_graphs():
    """compareKMers == set algebra on the base-only k-mers spelled by the two input graphs."""
    import itertools
    g1 = graphs.snp_graph(120, 0x52, 0x53, snp_period=8, node_len=8)
    g2 = graphs.snp_graph(120, 0x52, 0x99, snp_period=6, node_len=8)
    a, b = OracleIndex(build(g1, 6, sample_period=8, branching=4)), OracleIndex(build(g2, 6, sample_period=8, branching=4))
    A, B = GraphBrute(g1), GraphBrute(g2)
    for k in range(0, 6):
        sa = {t for t in itertools.product([1, 2, 3, 4], repeat=k) if A.starts(list(t))} if k else {()}
        sb = {t for t in itertools.product([1, 2, 3, 4], repeat=k) if B.starts(list(t))} if k else {()}
        assert a.compare_kmers(b, k) == (len(sa & sb), len(sa - sb), len(sb - sa)), k
    assert a.compare_kmers(b, 7) == (0, 0, 0) and a.compare_kmers(a, 4)[1:] == (0, 0)

    def label(rec):      # KMerComparisonState::set, algorithms.cpp:451-457: extension step i (last character first) at bits [3i, 3i+3)
        bits = int(rec[5]) | (int(rec[6]) << 64) | (int(rec[7]) << 128)
        return tuple(reversed([(bits >> (3 * i)) & 7 for i in range(int(rec[4]))]))
    for k in range(1, 6):
        counts, left, right = a.compare_kmers_records(b, k)
        sa = {t for t in itertools.product([1, 2, 3, 4], repeat=k) if A.starts(list(t))}
        sb = {t for t in itertools.product([1, 2, 3, 4], repeat=k) if B.starts(list(t))}
        assert counts == (len(sa & sb), len(sa - sb), len(sb - sa))
        assert {label(r) for r in left} == sa - sb and {label(r) for r in right} == sb - sa
        for r in left:       # the left range is find() of the label in the left index; the label is absent from the right one
            pattern = bytes(b"ACGT"[c - 1] for c in label(r))
            assert (int(r[0]), int(r[1])) == tuple(a.find(pattern))
            assert int(r[2]) == int(r[3]) + 1 and b.find(pattern)[0] > b.find(pattern)[1]


def test_fuzz_oracle_vs_input_graph():
    """The definition-level check of test_find_vs_input_graph on the graph family the GPU fuzz test
    uses (tests/test_gpu_parity.py::test_fuzz_random_graphs): 60 seeded random graphs with bubbles,
    indels, cycles and Ns; locate(find(X)) == start positions of paths labelled X, count == their number."""
    for seed in range(60):
        rng = SplitMix64(0xF00 + seed)
        n = 12 + rng.below(50)
        g = graphs.random_graph(n, 0xF100 + seed, p_branch=0.15 + 0.05 * (seed % 4), p_back=(0.08 if seed % 3 == 0 else 0.0),
                                p_n=0.05, alphabet=(2 if seed % 5 == 0 else 4))
        K = 2 + seed % 5
        ix = build(g, K, sample_period=2 + seed % 7, branching=2 + seed % 5)
        o, gb = OracleIndex(ix), GraphBrute(g)
        for p in random_patterns(g, K, 0xF200 + seed, 60):
            p = truncate_at_sink(p)[:K]
            comps = [int(ix.char2comp[b]) for b in p]
            found = o.find(p)
            expected = gb.occurrences(comps)
            assert [int(v) for v in o.locate(found)] == expected, (seed, p, found)
            assert o.count(found) == len(expected), (seed, p)
