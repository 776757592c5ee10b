"""Workload generators agree with each other: the scalable C++ builder with the definitional
builder, and the torch (GPU-capable) linear-graph generator with the C++ builder."""
import numpy as np
import pytest

from workload import graphs, builder, brute_builder
from workload.rng import SplitMix64, splitmix64_array
from test_oracle import CASES

FIELDS = ("pred_mask", "outdeg", "lcp", "val_off", "vals", "redundant", "key_len")


@pytest.mark.parametrize("case", range(len(CASES)), ids=[c[0] for c in CASES])
def test_cpp_builder_matches_definition(case):
    name, g, K = CASES[case]
    a = brute_builder.node_table(g, K)
    b = builder.node_table(g, K, threads=2)
    for f in FIELDS:
        assert np.array_equal(getattr(a, f).astype(np.uint64), getattr(b, f).astype(np.uint64)), (name, f)


def test_rng_vectorised_matches_scalar():
    r = SplitMix64(0x1234)
    assert [r.next() for _ in range(50)] == [int(x) for x in splitmix64_array(0x1234, 50)]


@pytest.mark.parametrize("n,seed", [(5000, 0x6C5A0040), (777, 3)])
def test_linear_torch_matches_general_builder(n, seed):
    import torch
    from workload import linear_torch
    assert np.array_equal(linear_torch.random_bases_torch(n, seed, torch.device("cpu")).numpy(),
                          graphs.random_bases(n, seed))
    a = builder.build(graphs.linear_graph(n, seed), 32)
    b = linear_torch.build_linear(n, seed, order=32, device=torch.device("cpu"))
    assert (a.n, a.e, a.sample_count, a.sample_width) == (b.n, b.e, b.sample_count, b.sample_width)
    assert np.array_equal(a.C, b.C)
    w = (a.n + 63) // 64
    for c in range(a.sigma):
        assert np.array_equal(a.bwt[c][:w], b.bwt[c][:w]), c
    assert np.array_equal(a.edges[:w], b.edges[:w])
    assert np.array_equal(a.sampled_paths[:w], b.sampled_paths[:w])
    assert np.array_equal(a.stored_samples_plain, b.stored_samples_plain)
    sw = (a.sample_count * a.sample_width + 63) // 64
    assert np.array_equal(a.stored_samples[:sw], b.stored_samples[:sw])
    assert np.array_equal(a.samples[: (a.sample_count + 63) // 64], b.samples[: (a.sample_count + 63) // 64])
    assert np.array_equal(a.lcp_data, b.lcp_data) and np.array_equal(a.lcp_offsets, b.lcp_offsets)
    assert a.extra_values_len == b.extra_values_len == 0 and a.redundant_len == b.redundant_len
    assert np.array_equal(a.redundant[: (a.redundant_len + 63) // 64], b.redundant[: (a.redundant_len + 63) // 64])
    assert not a.extra_filter[:w].any() and not b.extra_filter[:w].any()


@pytest.mark.parametrize("n,seed,family_blocks", [(9000, 0x6C5A0060, 5), (20011, 11, 8)])
def test_repeat_rich_text_matches_general_builder(n, seed, family_blocks):
    """The repeat-rich text of workload/repeats_torch.py (families of interspersed and young repeats, tandem arrays) through
    the prefix-doubling generator equals the general builder on the linear graph of the same text, field by field, and its
    found 32-mers are answered with wide ranges: range width == number of occurrences in the text (the definition)."""
    import torch
    from workload import linear_torch, repeats_torch
    from oracle.oracle import OracleIndex
    from gcsa2_amd.hostview import concat_patterns
    dev = torch.device("cpu")
    seq = repeats_torch.repeat_bases_torch(n, seed, dev, family_blocks=family_blocks)
    assert seq.shape[0] == n and int(seq.min()) >= 1 and int(seq.max()) <= 4
    assert not torch.equal(seq, linear_torch.random_bases_torch(n, seed, dev))
    a = builder.build(graphs.linear_graph(n, seed, sequence=seq.numpy()), 256)
    b = linear_torch.build_linear(n, seed, order=256, device=dev, sequence=seq)
    assert (a.n, a.e, a.sample_count, a.sample_width) == (b.n, b.e, b.sample_count, b.sample_width)
    assert np.array_equal(a.C, b.C)
    w = (a.n + 63) // 64
    for c in range(a.sigma):
        assert np.array_equal(a.bwt[c][:w], b.bwt[c][:w]), c
    assert np.array_equal(a.edges[:w], b.edges[:w]) and np.array_equal(a.sampled_paths[:w], b.sampled_paths[:w])
    assert np.array_equal(a.stored_samples_plain, b.stored_samples_plain)
    assert np.array_equal(a.lcp_data, b.lcp_data) and np.array_equal(a.lcp_offsets, b.lcp_offsets)
    c = linear_torch.build_linear(n, seed, order=256, device=dev, sequence=seq, with_lcp=False)     # what the bench builds
    for comp in range(a.sigma):
        assert np.array_equal(c.bwt[comp][:w], b.bwt[comp][:w])
    assert np.array_equal(c.stored_samples_plain, b.stored_samples_plain) and c.lcp_size == 0
    pats, start = repeats_torch.substring_patterns_device(seq, 400, 32, seed + 1)
    occ = repeats_torch.count_occurrences_device(seq, pats).numpy()
    data, off = concat_patterns([bytes(p) for p in pats.numpy()])
    got = OracleIndex(b).find_batch(data, off)
    assert np.array_equal(got[:, 1] - got[:, 0] + 1, occ.astype(np.uint64))
    assert occ.min() >= 1 and occ.mean() > 1.05 and occ.max() >= 3          # a handful of copies per family at this size
    a2, s2 = repeats_torch.substring_patterns_device(seq, 100, 32, seed + 1, first=300)
    assert torch.equal(a2, pats[300:]) and torch.equal(s2, start[300:])          # any shard of the batch is the same


@pytest.mark.parametrize("degree", [8, 12, 16])
def test_mseq_index_has_analytic_answers(degree):
    """The sort-free m-sequence index: find() of any substring of length >= degree / 2 is the
    closed-form rank of its rotation (checked through the oracle), shorter ones cover 4^(k-m) ranks."""
    import torch
    from workload import mseq_torch
    from oracle.oracle import OracleIndex
    ix, sym_t, rank = mseq_torch.build_mseq(degree, device=torch.device("cpu"))
    assert ix.n == (1 << degree) - 1 and sorted(rank.tolist()) == list(range(ix.n))
    cpu = OracleIndex(ix, with_samples=False, with_counters=False, with_lcp=False)
    for m in (degree // 2, degree // 2 + 3, 32):
        pats, exp = mseq_torch.substring_patterns(sym_t, rank, 500, m, 0xD0 + m)
        flat = np.ascontiguousarray(pats.reshape(-1))
        off = np.arange(501, dtype=np.uint64) * np.uint64(m)
        assert np.array_equal(cpu.find_batch(flat, off), exp)
    # every (degree/2)-mer except A^(degree/2) occurs exactly once: countKMers
    assert cpu.count_kmers(degree // 2, force=True) == ix.n


@pytest.mark.parametrize("degree", [8, 12])
def test_mseq_closed_form_equals_general_builder(degree):
    """The closed-form m-sequence index (samples, counters, LCP included) is exactly what the
    general builder produces for the cycle graph of the same text."""
    import torch
    from workload import mseq_torch
    a = builder.build(mseq_torch.cycle_graph(degree), degree // 2)
    b, _, _ = mseq_torch.build_mseq(degree, device=torch.device("cpu"), full=True)
    assert (a.n, a.e, a.order, a.sample_count, a.sample_width) == (b.n, b.e, b.order, b.sample_count, b.sample_width)
    assert np.array_equal(a.C, b.C)
    w = (a.n + 63) // 64
    for c in range(a.sigma):
        assert np.array_equal(a.bwt[c][:w], b.bwt[c][:w]), c
    assert np.array_equal(a.edges[:w], b.edges[:w])
    assert np.array_equal(a.sampled_paths[:w], b.sampled_paths[:w])
    assert np.array_equal(a.stored_samples_plain, b.stored_samples_plain)
    assert np.array_equal(a.lcp_data, b.lcp_data) and np.array_equal(a.lcp_offsets, b.lcp_offsets)
    assert a.extra_values_len == b.extra_values_len == 0 and a.redundant_len == b.redundant_len
    assert np.array_equal(a.redundant[: (a.redundant_len + 63) // 64], b.redundant[: (b.redundant_len + 63) // 64])


def test_oracle_locate_parent_closed_form():
    """Oracle locate() / parent() / count() against the closed forms of the m-sequence index."""
    import torch
    from workload import mseq_torch
    from oracle.oracle import OracleIndex
    degree, k = 12, 6
    ix, sym_t, rank = mseq_torch.build_mseq(degree, device=torch.device("cpu"), full=True)
    cpu = OracleIndex(ix)
    pos = np.arange(0, ix.n, 7)
    r = rank[pos].astype(np.uint64)
    ranges = np.stack([r, r], axis=1)
    offs, vals = cpu.locate_batch(ranges)
    assert np.array_equal(vals, mseq_torch.node_values(pos)) and np.array_equal(np.diff(offs), np.ones(len(pos), dtype=np.uint64))
    assert np.array_equal(cpu.count_batch(ranges), np.ones(len(pos), dtype=np.uint64))
    lcpv = ix.lcp_data[: ix.n].astype(np.int64)
    ri = r.astype(np.int64)
    L = np.maximum(lcpv[ri], np.where(ri + 1 < ix.n, lcpv[np.minimum(ri + 1, ix.n - 1)], 0))
    shift = 2 * (k - L)
    lo = ((ri + 1) >> shift) << shift
    par = cpu.parent_batch(ranges)
    assert np.array_equal(par["sp"].astype(np.int64), np.maximum(lo, 1) - 1)
    assert np.array_equal(par["ep"].astype(np.int64), np.minimum(lo + (np.int64(1) << shift) - 1, ix.n) - 1)
    assert np.array_equal(par["node_lcp"].astype(np.int64), L)
    # a wider range: all rotations starting with a given 3-mer -> 4^(k-3) values, the start positions of that 3-mer
    sym = sym_t.numpy()
    pat = bytes(b"ACGT"[s] for s in sym[100:103])
    rng = cpu.find(pat)
    occ = [p for p in range(ix.n) if all(sym[(p + j) % ix.n] == sym[100 + j] for j in range(3))]
    assert rng[1] - rng[0] + 1 == len(occ)
    assert cpu.locate(rng).tolist() == sorted(int(v) for v in mseq_torch.node_values(np.array(occ)))


def test_mseq_snp_index_against_the_definition():
    """The branching footprint-scale generator (workload/mseq_torch.py::build_mseq_snp) at degree 12: the index is
    the order-6 de Bruijn graph of {text (k + 1)-mers} + {(k + 1)-mers through a SNP's alternative base}.  Checked
    against that definition, not against any builder: edge counts, out-degrees, and find() of arbitrary patterns
    through the oracle -- non-empty iff every (k + 1)-mer of the pattern is an edge, and then the single node of the
    pattern's first k characters; walks through the graph equal their closed form."""
    import torch
    from workload import mseq_torch
    from workload.index_arrays import unpack_bits
    from workload.rng import SplitMix64
    from oracle.oracle import OracleIndex
    degree, k = 12, 6
    dev = torch.device("cpu")
    ix, sym_t, rank, alt_t = mseq_torch.build_mseq_snp(degree, period=40, device=dev)
    N = ix.n
    sym = sym_t.numpy().astype(np.int64)
    alt = alt_t.numpy().astype(np.int64)
    assert N == 4 ** k - 1 and ix.order == k
    edges = set()
    for p in range(N):
        edges.add(tuple(sym[(p + j) % N] for j in range(k + 1)))
    sites = np.flatnonzero(alt != 255)
    assert len(sites) >= 90 and np.all(np.diff(sites) >= 2 * (k + 1))
    for s in sites:
        for p in range(s - k, s + 1):
            w = [sym[(p + j) % N] for j in range(k + 1)]
            w[s - p] = alt[s]
            edges.add(tuple(w))
    assert ix.e == len(edges) and 1.05 * N < ix.e < 1.15 * N
    value = lambda kmer: sum(int(c) * 4 ** (k - 1 - j) for j, c in enumerate(kmer))       # noqa: E731
    B = [unpack_bits(ix.bwt[c], N) for c in range(7)]
    for c in (0, 5, 6):
        assert not B[c].any()
    preds = sum(int(B[c + 1].sum()) for c in range(4))
    assert preds == ix.e and [int(x) for x in np.diff(ix.C)] == [0] + [int(B[c + 1].sum()) for c in range(4)] + [0, 0]
    for w in list(edges)[:3000]:
        assert B[w[0] + 1][value(w[1:]) - 1]                                               # predecessor label of the target node
    out = unpack_bits(ix.edges, ix.e)
    ends = np.flatnonzero(out)
    assert len(ends) == N
    outdeg = np.diff(np.concatenate([[-1], ends]))
    by_source = {}
    for w in edges:
        by_source[value(w[:k])] = by_source.get(value(w[:k]), 0) + 1
    assert all(outdeg[v - 1] == d for v, d in by_source.items()) and len(by_source) == N

    cpu = OracleIndex(ix, with_samples=False, with_counters=False, with_lcp=False)
    rng = SplitMix64(0x5A1)
    letters = b"ACGT"
    pats = []
    for _ in range(1500):                                     # uniform strings: mostly not in the graph
        pats.append(tuple(rng.below(4) for _ in range(k + 1 + rng.below(5))))
    for _ in range(1500):                                     # walks, some with one substitution
        p, m = rng.below(N), k + 1 + rng.below(12)
        w = [int(alt[(p + j) % N]) if alt[(p + j) % N] != 255 and rng.below(2) else int(sym[(p + j) % N]) for j in range(m)]
        if rng.below(3) == 0:
            w[rng.below(m)] = rng.below(4)
        pats.append(tuple(w))
    hits = 0
    for pat in pats:
        sp, ep = cpu.find(bytes(letters[c] for c in pat))
        inside = all(tuple(pat[j:j + k + 1]) in edges for j in range(len(pat) - k))
        if inside:
            hits += 1
            assert (sp, ep) == (value(pat[:k]) - 1, value(pat[:k]) - 1), pat
        else:
            assert sp > ep or sp == ep + 1 or (sp + 1) % (1 << 64) > (ep + 1) % (1 << 64), pat
    assert 800 < hits < 2600
    rank_t = torch.from_numpy(rank.view(np.int32))
    walks, exp = mseq_torch.walk_patterns_device(sym_t, alt_t, rank_t, 0, 2000, 20, 0x5A2)
    got = cpu.find_batch(walks.reshape(-1).numpy(), np.arange(2001, dtype=np.uint64) * np.uint64(20))
    assert np.array_equal(got[:, 0], exp.numpy().astype(np.uint64)) and np.array_equal(got[:, 1], got[:, 0])


def test_mseq_snp_lcp_against_the_definition():
    """The LCP array the branching generator attaches (the node set is that of the plain text): equal to the common
    prefixes of the lexicographically adjacent k-mers, and consistent with the index the way verifyIndex demands
    (src/algorithms.cpp:146-167): parent(find(X)) is the range of the longest proper prefix of X whose range differs,
    at that prefix's length -- on short patterns, whose ranges are wide, and across SNP bubbles."""
    import torch
    from workload import mseq_torch
    from workload.rng import SplitMix64
    from oracle.oracle import OracleIndex
    degree, k = 12, 6
    ix, sym_t, rank, alt_t = mseq_torch.build_mseq_snp(degree, period=40, device=torch.device("cpu"), with_lcp=True, branching=4)
    N = ix.n
    keys = [tuple((v >> (2 * (k - 1 - j))) & 3 for j in range(k)) for v in range(1, N + 1)]
    assert keys == sorted(keys)
    want = [0] + [next(j for j in range(k) if keys[i - 1][j] != keys[i][j]) for i in range(1, N)]
    assert [int(x) for x in ix.lcp_data[:N]] == want and ix.lcp_size == N
    cpu = OracleIndex(ix, with_samples=False, with_counters=False, with_lcp=True)
    rng = SplitMix64(0x5A3)
    sym, alt = sym_t.numpy(), alt_t.numpy()
    checked = 0
    for _ in range(600):
        p, m = rng.below(N), 1 + rng.below(k)
        w = [int(alt[(p + j) % N]) if alt[(p + j) % N] != 255 and rng.below(2) else int(sym[(p + j) % N]) for j in range(m)]
        pat = bytes(b"ACGT"[c] for c in w)
        rng_x = cpu.find(pat)
        if rng_x[0] > rng_x[1] or rng_x == (0, N - 1):
            continue
        end = m
        shorter = rng_x
        while shorter == rng_x:
            end -= 1
            shorter = cpu.find(pat[:end])
        parent = cpu.parent(rng_x)
        assert (parent[0], parent[1]) == shorter and parent[4] == end, (pat, rng_x, parent, shorter, end)
        assert cpu.depth(shorter) == end or shorter == (0, N - 1)
        checked += 1
    assert checked > 400


# ---- indexes beyond 2^32 path nodes: workload/dbg_torch.py at small degrees --------------------------------------

@pytest.mark.parametrize("degree", [8, 10, 12, 16])
def test_dbg_plain_equals_general_builder(degree):
    """The non-maximal LFSR cycle (a third of the k-mers present, ranks through the universe bitmap): the plain index
    with its closed-form samples, counters and LCP array is exactly what the general builder produces for the cycle
    graph of the same text."""
    import torch
    from workload import dbg_torch
    a = builder.build(dbg_torch.cycle_graph(degree), degree // 2)
    b, wl = dbg_torch.build_dbg(degree, device=torch.device("cpu"), full=True, chunk_bits=10)
    assert b.n == dbg_torch.text_length(degree) == ((1 << degree) - 1) // 3
    assert (a.n, a.e, a.order, a.sample_count, a.sample_width) == (b.n, b.e, b.order, b.sample_count, b.sample_width)
    assert np.array_equal(a.C, b.C)
    w = (a.n + 63) // 64
    for c in range(a.sigma):
        assert np.array_equal(a.bwt[c][:w], b.bwt[c][:w]), c
    assert np.array_equal(a.edges[:w], b.edges[:w])
    assert np.array_equal(a.sampled_paths[:w], b.sampled_paths[:w])
    assert np.array_equal(a.stored_samples_plain, b.stored_samples_plain)
    assert np.array_equal(a.lcp_data, b.lcp_data) and np.array_equal(a.lcp_offsets, b.lcp_offsets)
    assert a.extra_values_len == b.extra_values_len == 0 and a.redundant_len == b.redundant_len
    assert np.array_equal(a.redundant[: (a.redundant_len + 63) // 64], b.redundant[: (b.redundant_len + 63) // 64])
    # the closed form the full-size tests rely on: find() of a substring = the bitmap rank of its first k-mer
    pats, start, exp = dbg_torch.walk_patterns_device(wl, 0, 300, degree // 2 + 5, 0xD1)
    sym = wl.sym_t.numpy().astype(np.int64)
    k = degree // 2
    values = sorted(sum(int(sym[(p + j) % b.n]) << (2 * (k - 1 - j)) for j in range(k)) for p in range(b.n))
    for q in range(300):
        p = int(start[q])
        v = sum(int(sym[(p + j) % b.n]) << (2 * (k - 1 - j)) for j in range(k))
        assert values[int(exp[q])] == v


@pytest.mark.parametrize("degree,period", [(12, 40), (16, 54)])
def test_dbg_snp_index_against_the_definition(degree, period):
    """The branching generator (workload/dbg_torch.py::build_dbg with SNP bubbles): the index is the order-k de Bruijn
    graph of {text (k + 1)-mers} + {(k + 1)-mers through a SNP's alternative base}; its path nodes are ALL k-mers of
    those, text and alternative ones, in lexicographic order.  Checked against that definition, not against any
    builder: node and edge sets, predecessor labels, out-degrees, the LCP array, find() of arbitrary patterns through the
    oracle, and the closed form of walks."""
    import torch
    from workload import dbg_torch
    from workload.index_arrays import unpack_bits
    from workload.rng import SplitMix64
    from oracle.oracle import OracleIndex
    k = degree // 2
    ix, wl = dbg_torch.build_dbg(degree, period=period, device=torch.device("cpu"), with_lcp=True, branching=4, chunk_bits=11)
    P = wl.P
    sym = wl.sym_t.numpy().astype(np.int64)
    alt = wl.alt_t.numpy().astype(np.int64)
    edges = set()
    for p in range(P):
        edges.add(tuple(sym[(p + j) % P] for j in range(k + 1)))
    sites = np.flatnonzero(alt != 255)
    assert len(sites) >= P // period - 2 and np.all(np.diff(sites) >= 2 * (k + 1))
    for s in sites:
        for p in range(s - k, s + 1):
            w = [sym[(p + j) % P] for j in range(k + 1)]
            w[s - p] = alt[s]
            edges.add(tuple(w))
    value = lambda kmer: sum(int(c) * 4 ** (k - 1 - j) for j, c in enumerate(kmer))       # noqa: E731
    node_values = sorted({value(w[:k]) for w in edges} | {value(w[1:]) for w in edges})
    node_id = {v: i for i, v in enumerate(node_values)}
    N = len(node_values)
    assert ix.n == N and ix.e == len(edges) and ix.order == k
    assert N > 1.05 * P and 1.03 * N < ix.e < 1.12 * N                 # alternative k-mers are new path nodes; e / n grows with k (1.08 at k = 17)
    B = [unpack_bits(ix.bwt[c], N) for c in range(7)]
    for c in (0, 5, 6):
        assert not B[c].any()
    assert sum(int(B[c + 1].sum()) for c in range(4)) == ix.e
    assert [int(x) for x in np.diff(ix.C)] == [0] + [int(B[c + 1].sum()) for c in range(4)] + [0, 0]
    for w in edges:
        assert B[w[0] + 1][node_id[value(w[1:])]]
    out = unpack_bits(ix.edges, ix.e)
    ends = np.flatnonzero(out)
    assert len(ends) == N
    outdeg = np.diff(np.concatenate([[-1], ends]))
    by_source = {}
    for w in edges:
        by_source[value(w[:k])] = by_source.get(value(w[:k]), 0) + 1
    assert len(by_source) == N and all(outdeg[node_id[v]] == d for v, d in by_source.items())
    keys = [tuple((v >> (2 * (k - 1 - j))) & 3 for j in range(k)) for v in node_values]
    want = [0] + [next(j for j in range(k) if keys[i - 1][j] != keys[i][j]) for i in range(1, N)]
    assert [int(x) for x in ix.lcp_data[:N]] == want and ix.lcp_size == N

    cpu = OracleIndex(ix, with_samples=False, with_counters=False, with_lcp=True)
    rng = SplitMix64(0x5A1 + degree)
    letters = b"ACGT"
    pats = []
    for _ in range(1000):                                     # uniform strings: mostly not in the graph
        pats.append(tuple(rng.below(4) for _ in range(k + 1 + rng.below(5))))
    for _ in range(1500):                                     # walks, some with one substitution
        p, m = rng.below(P), k + 1 + rng.below(12)
        w = [int(alt[(p + j) % P]) if alt[(p + j) % P] != 255 and rng.below(2) else int(sym[(p + j) % P]) for j in range(m)]
        if rng.below(3) == 0:
            w[rng.below(m)] = rng.below(4)
        pats.append(tuple(w))
    hits = 0
    for pat in pats:
        sp, ep = cpu.find(bytes(letters[c] for c in pat))
        inside = all(tuple(pat[j:j + k + 1]) in edges for j in range(len(pat) - k))
        if inside:
            hits += 1
            assert (sp, ep) == (node_id[value(pat[:k])], node_id[value(pat[:k])]), pat
        else:
            assert (sp + 1) % (1 << 64) > (ep + 1) % (1 << 64), pat
    assert 700 < hits < 2400
    walks, start, exp = dbg_torch.walk_patterns_device(wl, 0, 2000, k + 9, 0x5A2)
    got = cpu.find_batch(walks.reshape(-1).numpy(), np.arange(2001, dtype=np.uint64) * np.uint64(k + 9))
    assert np.array_equal(got[:, 0], exp.numpy().astype(np.uint64)) and np.array_equal(got[:, 1], got[:, 0])
    took_alt = sum(1 for q in range(2000) if any(walks[q, j] != letters[sym[(int(start[q]) + j) % P]] for j in range(k + 9)))
    assert took_alt > 100
    # parent(find(X)) = the range of the longest proper prefix of X with a different range, at that prefix's length
    # (the invariant verifyIndex asserts, src/algorithms.cpp:146-167)
    checked = 0
    for _ in range(400):
        p, m = rng.below(P), 1 + rng.below(k)
        w = [int(alt[(p + j) % P]) if alt[(p + j) % P] != 255 and rng.below(2) else int(sym[(p + j) % P]) for j in range(m)]
        pat = bytes(letters[c] for c in w)
        rng_x = cpu.find(pat)
        if rng_x[0] > rng_x[1] or rng_x == (0, N - 1):
            continue
        end, shorter = m, rng_x
        while shorter == rng_x:
            end -= 1
            shorter = cpu.find(pat[:end])
        parent = cpu.parent(rng_x)
        assert (parent[0], parent[1]) == shorter and parent[4] == end, (pat, rng_x, parent, shorter, end)
        checked += 1
    assert checked > 250


@pytest.mark.parametrize("degree", [12, 16])
def test_dbg_junction_index_against_the_definition(degree):
    """The whole-human-sized branching index (workload/dbg_torch.py::build_dbg with junction edges), at small degrees:
    path nodes = the k-mers of the cyclic text, edges = the text's (k + 1)-mers + the selected junctions u -> u[1..k) x
    between existing nodes.  Checked against that definition through the arrays and through the oracle: predecessor
    labels, out-degrees, find() of arbitrary patterns, the closed forms of walks (find, locate, count), the sampling rule
    of src/gcsa.cpp:621-646 (a node with several predecessors, or whose value is not its predecessor's + 1, is sampled),
    and the parent() invariant of verifyIndex."""
    import torch
    from workload import dbg_torch, mseq_torch
    from workload.index_arrays import unpack_bits
    from workload.rng import SplitMix64
    from oracle.oracle import OracleIndex
    k = degree // 2
    ix, wl = dbg_torch.build_dbg(degree, junctions=80, device=torch.device("cpu"), full=True, branching=4, chunk_bits=11)
    P = wl.P
    sym = wl.sym_t.numpy().astype(np.int64)
    def value(symbols):                                      # base-4 value of a string of any length
        v = 0
        for c in symbols:
            v = v * 4 + int(c)
        return v
    kmer_at = [value([sym[(p + j) % P] for j in range(k)]) for p in range(P)]
    node_values = sorted(kmer_at)
    node_id = {v: i for i, v in enumerate(node_values)}
    pos_of = {v: p for p, v in enumerate(kmer_at)}
    assert ix.n == P == len(node_id) and ix.order == k
    edges = set()
    junctions = 0
    for p in range(P):
        u, succ = kmer_at[p], kmer_at[(p + 1) % P]
        edges.add((u, succ))
        for x in range(4):
            cand = ((u & (4 ** (k - 1) - 1)) << 2) + x
            if cand != succ and cand in node_id and bool(dbg_torch.junction_selected(torch.tensor([(u << 2) + x]), 80)[0]):
                edges.add((u, cand))
                junctions += 1
    assert ix.e == len(edges) == P + junctions and 1.06 * P < ix.e < 1.10 * P
    B = [unpack_bits(ix.bwt[c], P) for c in range(7)]
    for c in (0, 5, 6):
        assert not B[c].any()
    assert sum(int(B[c + 1].sum()) for c in range(4)) == ix.e
    assert [int(x) for x in np.diff(ix.C)] == [0] + [int(B[c + 1].sum()) for c in range(4)] + [0, 0]
    indeg = np.zeros(P, dtype=np.int64)
    outdeg_want = np.zeros(P, dtype=np.int64)
    for u, v in edges:
        assert B[(u >> (2 * (k - 1))) + 1][node_id[v]]
        indeg[node_id[v]] += 1
        outdeg_want[node_id[u]] += 1
    ends = np.flatnonzero(unpack_bits(ix.edges, ix.e))
    assert len(ends) == P and np.array_equal(np.diff(np.concatenate([[-1], ends])), outdeg_want)
    assert int((indeg > 1).sum()) > 0.05 * P and int((outdeg_want > 1).sum()) > 0.05 * P
    # sampling rule
    sampled = unpack_bits(ix.sampled_paths, P)
    want_sampled = np.zeros(P, dtype=bool)
    for p in range(P):
        want_sampled[node_id[kmer_at[p]]] = (p % 32 == 0) or indeg[node_id[kmer_at[p]]] > 1
    assert np.array_equal(sampled, want_sampled) and ix.sample_count == int(want_sampled.sum())

    cpu = OracleIndex(ix)
    # locate / count of every node: its single value
    r = np.arange(P, dtype=np.uint64)
    offs, vals = cpu.locate_batch(np.stack([r, r], axis=1))
    assert np.array_equal(np.diff(offs), np.ones(P, dtype=np.uint64))
    assert np.array_equal(vals, mseq_torch.node_values(np.array([pos_of[v] for v in node_values])))
    assert np.array_equal(cpu.count_batch(np.stack([r, r], axis=1)), np.ones(P, dtype=np.uint64))
    rng = SplitMix64(0x5B1 + degree)
    letters = b"ACGT"
    edge_kmers = {(u << 2) | (v & 3) for u, v in edges}
    pats = [tuple(rng.below(4) for _ in range(k + 1 + rng.below(5))) for _ in range(1000)]
    walks, start, exp = dbg_torch.walk_patterns_device(wl, 0, 2500, k + 14, 0x5B2)
    walks_np = walks.numpy()
    for q in range(1500):                                     # walks, some with one substitution
        w = [letters.index(int(c)) for c in walks_np[q, : k + 1 + rng.below(13)]]
        if rng.below(3) == 0:
            w[rng.below(len(w))] = rng.below(4)
        pats.append(tuple(w))
    hits = 0
    for pat in pats:
        sp, ep = cpu.find(bytes(letters[c] for c in pat))
        inside = value(pat[:k]) in node_id and all(value(pat[j:j + k + 1]) in edge_kmers for j in range(len(pat) - k))
        if inside:
            hits += 1
            assert (sp, ep) == (node_id[value(pat[:k])], node_id[value(pat[:k])]), pat
        else:
            assert (sp + 1) % (1 << 64) > (ep + 1) % (1 << 64), pat
    assert 700 < hits < 2400
    got = cpu.find_batch(walks_np.reshape(-1), np.arange(2501, dtype=np.uint64) * np.uint64(k + 14))
    assert np.array_equal(got[:, 0], exp.numpy().astype(np.uint64)) and np.array_equal(got[:, 1], got[:, 0])
    left_text = sum(1 for q in range(2500)
                    if any(walks_np[q, j] != letters[sym[(int(start[q]) + j) % P]] for j in range(k + 14)))
    assert left_text > 400                                    # the walks do cross junctions
    offs, vals = cpu.locate_batch(got)
    assert np.array_equal(vals, mseq_torch.node_values(start.numpy()))
    # a wide range: every node whose k-mer starts with a 3-mer; its values are the start positions of that 3-mer
    pat = bytes(letters[s] for s in sym[100:103])
    rng3 = cpu.find(pat)
    occ = [p for p in range(P) if all(sym[(p + j) % P] == sym[100 + j] for j in range(3))]
    assert rng3[1] - rng3[0] + 1 == len(occ) == cpu.count(rng3)
    assert cpu.locate(rng3).tolist() == sorted(int(v) for v in mseq_torch.node_values(np.array(occ)))
    checked = 0
    for _ in range(400):
        p, m = rng.below(P), 1 + rng.below(k)
        pat = bytes(letters[sym[(p + j) % P]] for j in range(m))
        rng_x = cpu.find(pat)
        if rng_x == (0, P - 1):
            continue
        end, shorter = m, rng_x
        while shorter == rng_x:
            end -= 1
            shorter = cpu.find(pat[:end])
        parent = cpu.parent(rng_x)
        assert (parent[0], parent[1]) == shorter and parent[4] == end, (pat, rng_x, parent, shorter, end)
        checked += 1
    assert checked > 250


def test_repeat_rich_backbone():
    """workload/graphs.py::repeat_bases: seeded, over {A, C, G, T}, and repeat-rich -- sampled 16-mers occur many times
    (a uniform random backbone of this length has essentially unique 16-mers)."""
    n = 1 << 18
    a, b = graphs.repeat_bases(n, 0x6C5A0020), graphs.repeat_bases(n, 0x6C5A0020)
    assert np.array_equal(a, b) and a.shape == (n,) and int(a.min()) >= 1 and int(a.max()) <= 4
    assert not np.array_equal(a, graphs.repeat_bases(n, 0x6C5A0021))

    def mean_occurrences(seq):
        v = np.zeros(n - 15, dtype=np.uint64)
        for j in range(16):
            v = (v << np.uint64(2)) | (seq[j: n - 15 + j].astype(np.uint64) - np.uint64(1))
        uniq, inv, counts = np.unique(v, return_inverse=True, return_counts=True)
        return float(counts[inv[:: 97]].mean())
    assert mean_occurrences(a) > 20 and mean_occurrences(graphs.random_bases(n, 0x6C5A0020)) < 1.01


@pytest.mark.parametrize("degree,junctions", [(10, 80), (12, 0), (16, 80)])
def test_dbg_kmer_counts_closed_form(degree, junctions):
    """countKMers(k) on the de Bruijn indexes in closed form (workload/dbg_torch.py::distinct_prefixes: the distinct k-prefixes of
    the node labels, read off the bitmap of the k-mer universe) equals what the oracle counts by walking the LF search tree
    (src/algorithms.cpp:387-421), for every k up to the order; junction edges between existing nodes add no k-mer."""
    import torch
    from workload import dbg_torch
    from oracle.oracle import OracleIndex
    order = degree // 2
    ix, wl = dbg_torch.build_dbg(degree, junctions=junctions, device=torch.device("cpu"))
    cpu = OracleIndex(ix)
    for k in range(1, order + 1):
        assert dbg_torch.distinct_prefixes(wl.nodes, order, k) == cpu.count_kmers(k), k
    assert dbg_torch.distinct_prefixes(wl.nodes, order, order) == ix.n


@pytest.mark.parametrize("degree,junctions", [(12, 80), (16, 0)])
def test_dbg_prefix_patterns_closed_form(degree, junctions):
    """find() of the first m < k characters of a path label = the interval of the bitmap ranks of the k-mers with that prefix
    (workload/dbg_torch.py::prefix_patterns_device, the wide-range legs of bench.py): equals the oracle's backward search."""
    import torch
    from workload import dbg_torch
    from oracle.oracle import OracleIndex
    from gcsa2_amd.hostview import concat_patterns
    ix, dbg = dbg_torch.build_dbg(degree, junctions=junctions, device=torch.device("cpu"), full=True)
    cpu = OracleIndex(ix)
    for m in range(1, degree // 2):
        pats, sp, ep = dbg_torch.prefix_patterns_device(dbg, 5, 700, m, 0x77 + m)
        data, off = concat_patterns([bytes(p) for p in pats.numpy()])
        got = cpu.find_batch(data, off)
        assert np.array_equal(got[:, 0], sp.numpy().astype(np.uint64)) and np.array_equal(got[:, 1], ep.numpy().astype(np.uint64)), m
        assert int((ep - sp).min()) >= 0
