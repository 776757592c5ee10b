"""GPU parity at the sizes of BASELINE.json's configs.

  config 1  1-Mbp linear graph, order 64, 100 k 16-mers: the whole query_gcsa phase sequence
            (find -> parent -> depth -> count -> locate, reference benchmark/query_gcsa.cpp:87-169)
            compared with the oracle on every query.
  config 2  (reduced: 2^21 bases) full parity incl. uniform patterns that die early.
  config 2  (full: 2^25 bases, 10 M 32-mers) size-independent properties + oracle on a sample.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from workload import graphs, builder, patterns
from workload.graphs import COMP2CHAR


def phases(gpu, lcp, cpu, flat, off, n, locate_limit=None):
    ranges = gpu.find_batch(flat, off)
    assert np.array_equal(ranges, cpu.find_batch(flat, off, threads=8))
    hit = ranges[(ranges[:, 0] <= ranges[:, 1]) & (ranges[:, 1] < n)]
    par = lcp.parent_batch(hit)
    assert np.array_equal(par, cpu.parent_batch(hit, threads=8))
    pr = np.stack([par["sp"], par["ep"]], axis=1)
    assert np.array_equal(lcp.depth_batch(pr), cpu.depth_batch(pr, threads=8))
    counts = gpu.count_batch(hit)
    assert np.array_equal(counts, cpu.count_batch(hit, threads=8))
    loc = hit if locate_limit is None else hit[(hit[:, 1] - hit[:, 0]) < locate_limit]
    go, gv = gpu.locate_batch(loc)
    co, cv = cpu.locate_batch(loc, threads=8)
    assert np.array_equal(go, co) and np.array_equal(gv, cv)
    assert np.array_equal(np.diff(go), gpu.count_batch(loc))   # query_gcsa.cpp:171-179
    return ranges, hit


def test_config1_query_gcsa_phases():
    from gcsa2_amd.binding import open_index
    from oracle.oracle import OracleIndex
    g = graphs.linear_graph(1_000_000, 0x6C5A0001)
    ix = builder.build(g, 64)
    gpu, lcp = open_index(ix)
    cpu = OracleIndex(ix)
    pats = np.concatenate([patterns.walk_patterns(g, 50_000, 16, 0x6C5A0002),
                           patterns.uniform_patterns(50_000, 16, 0x6C5A0003)])
    flat, off = patterns.as_batch(pats)
    ranges, hit = phases(gpu, lcp, cpu, flat, off, ix.n)
    assert hit.shape[0] >= 50_000
    assert ix.n == 1_000_002 and ix.e == ix.n          # linear graph: an FM-index, all outdegrees 1


def test_config2_reduced_full_parity():
    from gcsa2_amd.binding import open_index
    from oracle.oracle import OracleIndex
    g = graphs.snp_graph(1 << 21, 0x6C5A0010, 0x6C5A0011)
    ix = builder.build(g, 256)
    gpu, lcp = open_index(ix)
    cpu = OracleIndex(ix)
    pats = np.concatenate([patterns.walk_patterns(g, 150_000, 32, 0x6C5A0012),
                           patterns.uniform_patterns(50_000, 32, 0x6C5A0013)])
    flat, off = patterns.as_batch(pats)
    phases(gpu, lcp, cpu, flat, off, ix.n)
    # long patterns (config 5 shape): 256-mers, half of them with substitutions
    long = patterns.walk_patterns(g, 4000, 256, 0x6C5A0050)
    long[::2, 40::41] = ord("A")
    flat, off = patterns.as_batch(long)
    phases(gpu, lcp, cpu, flat, off, ix.n)
    # short / high-occupancy ranges: 4- and 8-mers have thousands of occurrences
    short = patterns.walk_patterns(g, 64, 8, 0x6C5A0051)
    flat, off = patterns.as_batch(short)
    phases(gpu, lcp, cpu, flat, off, ix.n)


def test_config2_full_size_properties():
    import torch
    from gcsa2_amd.binding import open_index
    from oracle.oracle import OracleIndex
    g = graphs.snp_graph(1 << 25, 0x6C5A0010, 0x6C5A0011)
    ix = builder.build(g, 256, keep_table=False)
    gpu, lcp = open_index(ix)
    nq, m = 10_000_000, 32
    pats = patterns.walk_patterns(g, nq, m, 0x6C5A0012)
    flat, off = patterns.as_batch(pats)
    ranges = gpu.find_batch(flat, off)
    # 1. walks through the graph are paths of the graph: no false negatives (paper.tex:270)
    assert bool(np.all(ranges[:, 0] <= ranges[:, 1])) and bool(np.all(ranges[:, 1] < ix.n))
    # 2. find(P) == LF(find(P[1:]), comp(P[0])): the batched single-step kernel composes to find
    tail_flat, tail_off = patterns.as_batch(np.ascontiguousarray(pats[:, 1:]))
    tails = gpu.find_batch(tail_flat, tail_off)
    comps = ix.char2comp[pats[:, 0]]
    assert np.array_equal(gpu.lf_batch(tails, comps), ranges)
    # 3. suffix-tree containment: parent(range) strictly contains the range, shallower depth
    sample = ranges[:: 50]
    par = lcp.parent_batch(sample)
    assert bool(np.all(par["sp"] <= sample[:, 0])) and bool(np.all(par["ep"] >= sample[:, 1]))
    assert bool(np.all((par["ep"] - par["sp"]) > (sample[:, 1] - sample[:, 0])))
    assert bool(np.all(par["node_lcp"] < m))
    # 4. count == |locate| and located values are sorted and distinct (first 200 k queries)
    sub = ranges[:200_000]
    loff, lval = gpu.locate_batch(sub)
    assert np.array_equal(np.diff(loff), gpu.count_batch(sub))
    seg = np.repeat(np.arange(sub.shape[0]), np.diff(loff).astype(np.int64))
    same = seg[1:] == seg[:-1]
    assert bool(np.all(lval[1:][same] > lval[:-1][same]))
    # 4b. both kernel generations agree on the whole batch (device-resident entry points)
    dev = torch.device("cuda", 0)
    d_pat = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    outs = []
    for variant in (1, 2):
        d_out = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
        gpu.find_device_variant(variant, d_pat.data_ptr(), d_off.data_ptr(), nq, d_out.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(d_out.cpu().numpy().view(np.uint64))
    assert np.array_equal(outs[0], ranges) and np.array_equal(outs[1], ranges)
    # 5. oracle on a sample + order-independent checksum of the whole batch vs the oracle's
    cpu = OracleIndex(ix, with_samples=False, with_counters=False, with_lcp=False)
    want = cpu.find_batch(flat, off, threads=64)
    assert np.array_equal(ranges, want)
    with np.errstate(over="ignore"):
        assert int((ranges * np.uint64(0x9E3779B97F4A7C15)).sum()) == int((want * np.uint64(0x9E3779B97F4A7C15)).sum())


def test_mseq_index_closed_form_on_gpu():
    """1 M-node index with analytic answers (workload/mseq_torch.py): every find() of a substring of
    the cyclic text returns the closed-form rank of its rotation; countKMers(10) = all 4^10 - 1."""
    import torch
    from workload import mseq_torch
    from gcsa2_amd.binding import GCSA
    from gcsa2_amd.binding import LCPArray
    ix, sym_t, rank = mseq_torch.build_mseq(20, device=torch.device("cuda", 0), full=True)
    gpu = GCSA(ix)
    lcp = LCPArray(gpu, int(ix.lcp_offsets[-1]), ix.n)
    for m in (10, 17, 32, 100):
        pats, exp = mseq_torch.substring_patterns(sym_t, rank, 200_000, m, 0xE0 + m)
        flat, off = patterns.as_batch(pats)
        assert np.array_equal(gpu.find_batch(flat, off), exp), m
    assert gpu.count_kmers(10, force=True) == ix.n
    # locate: the rotation starting at position p carries exactly the value of p
    starts = (mseq_torch._lsr(mseq_torch.splitmix64_torch(0xE0 + 100, 200_000, sym_t.device), 11) % ix.n).cpu().numpy()
    offs, vals = gpu.locate_batch(exp)
    assert np.array_equal(np.diff(offs), np.ones(exp.shape[0], dtype=np.uint64))
    assert np.array_equal(vals, mseq_torch.node_values(starts))
    assert np.array_equal(gpu.count_batch(exp), np.ones(exp.shape[0], dtype=np.uint64))
    # parent of a singleton: all k-mer values sharing the first L digits, L = max of the two LCPs
    k = 10
    lcpv = ix.lcp_data[: ix.n].astype(np.int64)
    r = exp[:, 0].astype(np.int64)
    L = np.maximum(lcpv[r], np.where(r + 1 < ix.n, lcpv[np.minimum(r + 1, ix.n - 1)], 0))
    shift = 2 * (k - L)
    lo = ((r + 1) >> shift) << shift
    par = lcp.parent_batch(exp)
    assert np.array_equal(par["sp"].astype(np.int64), np.maximum(lo, 1) - 1)
    assert np.array_equal(par["ep"].astype(np.int64), np.minimum(lo + (np.int64(1) << shift) - 1, ix.n) - 1)
    assert np.array_equal(par["node_lcp"].astype(np.int64), L)
