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
