"""GPU parity at the sizes of BASELINE.json's configs.

  config 1  1-Mbp linear graph, order 64, 100 k 16-mers: the whole query_gcsa phase sequence
            (find -> parent -> depth -> count -> locate, reference benchmark/query_gcsa.cpp:87-169)
            compared with the oracle on every query.
  config 2  (reduced: 2^21 bases) full parity incl. uniform patterns that die early.
  config 2/3 (full: 2^25 bases, 10 M 32-mers) size-independent properties, locate() of all 10 M ranges, oracle on the batch.
  config 4/5 (whole-human pangenome size: 5.73 G path nodes, e = 1.08 n -- beyond 2^32) find / locate / count / parent /
            256-bp patterns with parent() on failure against closed-form answers for every query, and against the oracle
            on samples; the u64 wire format of the gather.
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


def test_repeat_rich_index_parity():
    """Wide ranges on the hot path: the SNP graph over a repeat-rich backbone (workload/graphs.py::repeat_bases: interspersed
    families and tandem arrays, so that found k-mers match many path nodes, paper.tex:403,408).  The whole query_gcsa phase
    sequence against the oracle -- find() with endpoints in different blocks at most steps, seed-table entries of wide ranges,
    locate() with segments of every size class incl. the segmented radix sort, count == |locate|."""
    import torch
    from gcsa2_amd.binding import open_index
    from oracle.oracle import OracleIndex
    g = graphs.repeat_graph(1 << 19, 0x6C5A0020, 0x6C5A0021)
    ix = builder.build(g, 256)
    gpu, lcp = open_index(ix)
    cpu = OracleIndex(ix)
    widths = {}
    for m, nq in ((32, 60_000), (16, 30_000), (8, 2_000)):
        pats = np.concatenate([patterns.walk_patterns(g, nq, m, 0x6C5A0022 + m), patterns.uniform_patterns(nq // 4, m, 0x6C5A0023 + m)])
        flat, off = patterns.as_batch(pats)
        ranges, hit = phases(gpu, lcp, cpu, flat, off, ix.n)
        widths[m] = float((hit[:, 1] - hit[:, 0] + 1).mean())
    assert widths[32] > 5 and widths[16] > 50 and widths[8] > 100             # the ranges are wide (x16 at the bench's 2^23 bases)
    # the device-resident instrumented kernel sees second blocks and agrees with the default one
    dev = torch.device("cuda", 0)
    pats = patterns.walk_patterns(g, 100_000, 16, 0x6C5A0024)
    flat, off = patterns.as_batch(pats)
    d_pat = torch.from_numpy(np.concatenate([flat, np.zeros(8, dtype=np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_a = torch.zeros((100_000, 2), dtype=torch.int64, device=dev)
    d_b = torch.zeros_like(d_a)
    d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
    gpu.find_device(d_pat.data_ptr(), d_off.data_ptr(), 100_000, d_a.data_ptr(), 0)
    gpu.find_stats_device(d_pat.data_ptr(), d_off.data_ptr(), 100_000, d_b.data_ptr(), d_stats.data_ptr(), 0)
    torch.cuda.synchronize()
    assert torch.equal(d_a, d_b) and np.array_equal(d_a.cpu().numpy().view(np.uint64), cpu.find_batch(flat, off, threads=8))
    blocks, steps, lookups, jumps, fetch_steps, second, wide, _ = (int(x) for x in d_stats.cpu())
    assert blocks == fetch_steps + second and 0 < second < fetch_steps and lookups > 0


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
    # 4. config 3: count == |locate| and located values are sorted and distinct, for ALL 10 M ranges
    sub = ranges
    loff, lval = gpu.locate_batch(sub)
    assert np.array_equal(np.diff(loff), gpu.count_batch(sub))
    seg = np.repeat(np.arange(sub.shape[0]), np.diff(loff).astype(np.int64))
    same = seg[1:] == seg[:-1]
    assert bool(np.all(lval[1:][same] > lval[:-1][same]))
    # 4b. both launch shapes agree on the whole batch (device-resident entry points)
    dev = torch.device("cuda", 0)
    d_pat = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    outs = []
    for variant in (4, 2):
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


def dbg_parent_closed_form(dbg, lcpv, ranks, values):
    """parent() of the singleton ranges (r, r) on a de Bruijn index of workload/dbg_torch.py: the nodes sharing the first
    L characters with node r, L = max of the two adjacent LCP values -- the k-mer values [lo, lo + 4^(k - L)), whose
    path node ids follow from the bitmap rank.  Returns (sp, ep, L)."""
    import torch
    n, k = dbg.nodes.n, dbg.k
    right = torch.where(ranks + 1 < n, lcpv[torch.clamp(ranks + 1, max=n - 1)].to(torch.int64), torch.zeros_like(ranks))
    L = torch.maximum(lcpv[ranks].to(torch.int64), right)
    L = torch.where(ranks == 0, right, L)                   # LCP[0] = 0 by definition
    shift = 2 * (k - L)
    lo = (values >> shift) << shift
    hi = lo + (torch.ones_like(lo) << shift)                # exclusive; may equal 4^k, one past the universe
    top = torch.full_like(hi, n)
    inside = hi < dbg.nodes.universe
    top[inside] = dbg.nodes.rank(hi[inside])
    return dbg.nodes.rank(lo), top - 1, L


def test_pangenome_index_beyond_32_bits():
    """BASELINE configs[3] and [4] as SURVEY 8(d) wrote them, on one GPU: 5 726 623 061 path nodes and e = 1.08 n
    (paper.tex:380), i.e. path node, edge and LCP positions beyond 2^32 -- the 64-bit side of every kernel: block
    indices, 40-bit seed and locate-table fields, the u64 wire format of the gather.  The index is the order-17 de Bruijn
    graph of a degree-34 LFSR cycle plus junction edges (workload/dbg_torch.py; validated against its definition at small
    degrees in tests/test_workload.py) with samples, counters and LCP array; every find() / locate() / count() /
    parent() answer is known in closed form, the substituted 256-bp patterns (LF + parent interplay, reference
    src/algorithms.cpp:146-167 shape) are checked against the oracle on a sample."""
    import torch
    from workload import dbg_torch, mseq_torch
    from gcsa2_amd import binding
    from gcsa2_amd.binding import GCSA
    from oracle.oracle import OracleIndex
    dev = torch.device("cuda", 0)
    degree, k = 34, 17
    ix, dbg = dbg_torch.build_dbg(degree, junctions=80, device=dev, full=True)
    torch.cuda.empty_cache()
    assert ix.n == 5_726_623_061 and 1.07 * ix.n < ix.e < 1.09 * ix.n and ix.sample_width > 32
    gpu = GCSA(ix, device=0)
    assert gpu.size() == ix.n and gpu.pair_block_bytes() > 0 and gpu.locate_table_bytes() > 0 and gpu.kmer_table_k() >= 15
    st = torch.cuda.current_stream().cuda_stream

    def batch(nq, m, seed):
        pats, start, exp = dbg_torch.walk_patterns_device(dbg, 0, nq, m, seed)
        flat = torch.zeros(nq * m + 8, dtype=torch.uint8, device=dev)
        flat[: nq * m] = pats.reshape(-1)
        return pats, flat, torch.arange(nq + 1, dtype=torch.int64, device=dev) * m, start, exp

    # config 4: 20 M 32-mers, every range = the single node of the walk's first 17 characters
    nq, m = 20_000_000, 32
    _, d_pat, d_off, start, exp = batch(nq, m, 0x6C5A0041)
    assert int(exp.max().item()) > (1 << 32) and int((exp > (1 << 32)).sum().item()) > nq // 8      # sp beyond 32 bits
    d_out = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    gpu.find_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_out.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(d_out[:, 0], exp) and torch.equal(d_out[:, 1], exp)
    # ... also through the length-bucketed launch
    for variant in (4,):
        d_out1 = torch.zeros((2_000_000, 2), dtype=torch.int64, device=dev)
        gpu.find_device_variant(variant, d_pat.data_ptr(), d_off.data_ptr(), 2_000_000, d_out1.data_ptr(), st)
        torch.cuda.synchronize()
        assert torch.equal(d_out1, d_out[:2_000_000]), variant
    # k-mers exactly as long as the order, and uniform random 32-mers (mostly absent: edge-space empty ranges beyond 2^32)
    # against the oracle
    cpu = OracleIndex(ix, with_samples=False, with_counters=False)
    _, d_patk, d_offk, _, expk = batch(1_000_000, k, 0x6C5A0042)
    d_outk = torch.zeros((1_000_000, 2), dtype=torch.int64, device=dev)
    gpu.find_device(d_patk.data_ptr(), d_offk.data_ptr(), 1_000_000, d_outk.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(d_outk[:, 0], expk) and torch.equal(d_outk[:, 1], expk)
    ns = 200_000
    uni = torch.from_numpy(patterns.uniform_patterns(ns, 32, 0x6C5A0043)).to(dev)
    d_patu = torch.zeros(ns * 32 + 8, dtype=torch.uint8, device=dev)
    d_patu[: ns * 32] = uni.reshape(-1)
    d_outu = torch.zeros((ns, 2), dtype=torch.int64, device=dev)
    gpu.find_device(d_patu.data_ptr(), d_off.data_ptr(), ns, d_outu.data_ptr(), st)
    torch.cuda.synchronize()
    want = cpu.find_batch(d_patu[: ns * 32].cpu().numpy(), np.arange(ns + 1, dtype=np.uint64) * np.uint64(32), threads=8)
    got = d_outu.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want) and int((want[:, 0] > np.uint64(1 << 32)).sum()) > ns // 8
    # the u64 wire format of the gather (these ranges do not fit the (sp, len) u32 pairs): world-size-1 communicator
    try:
        comm = binding.Comm(binding.Comm.unique_id(), 0, 1, 0)
    except binding.Gcsa2Error as e:
        comm = None
        assert e.code == -5, e                          # RCCL not loadable on this host
    if comm is not None:
        d_recv = torch.zeros_like(d_out)
        comm.gather(d_out.data_ptr(), [nq * 16], d_recv.data_ptr(), 0, st)
        torch.cuda.synchronize()
        assert torch.equal(d_recv, d_out)
        comm.close()
        del d_recv
    # locate: the node of position p carries exactly the value of p (39-bit values); count = 1
    d_loff = torch.zeros(nq + 1, dtype=torch.int64, device=dev)
    d_lval = torch.zeros(nq, dtype=torch.int64, device=dev)
    assert gpu.locate_into(d_out.data_ptr(), nq, d_loff.data_ptr(), d_lval.data_ptr(), nq, st) == nq
    torch.cuda.synchronize()
    assert torch.equal(d_loff, torch.arange(nq + 1, dtype=torch.int64, device=dev))
    assert np.array_equal(d_lval.cpu().numpy().view(np.uint64), mseq_torch.node_values(start.cpu().numpy()))
    d_cnt = torch.zeros(nq, dtype=torch.int64, device=dev)
    gpu.count_device(d_out.data_ptr(), nq, d_cnt.data_ptr(), st)
    torch.cuda.synchronize()
    assert bool((d_cnt == 1).all())
    # ... and a batch of WIDE ranges (the first 8..12 characters of walks: thousands of nodes each) against the oracle:
    # locate with sort + unique over the big segments, count == |locate|
    nw = 2000
    wide_pats = [bytes(uni[q, : 8 + q % 5].cpu().numpy()) for q in range(nw)]
    from gcsa2_amd.hostview import concat_patterns
    wdata, woff = concat_patterns(wide_pats)
    wr = gpu.find_batch(wdata, woff)
    assert np.array_equal(wr, cpu.find_batch(wdata, woff, threads=8))
    hit = wr[(wr[:, 0] <= wr[:, 1]) & (wr[:, 1] < ix.n)][:300]
    assert hit.shape[0] > 100 and int((hit[:, 1] - hit[:, 0]).max()) > 5000
    go, gv = gpu.locate_batch(hit)
    full_cpu = OracleIndex(ix, with_lcp=False)
    co, cv = full_cpu.locate_batch(hit, threads=8)
    assert np.array_equal(go, co) and np.array_equal(gv, cv)
    assert np.array_equal(np.diff(go), gpu.count_batch(hit)) and np.array_equal(np.diff(go), full_cpu.count_batch(hit, threads=8))
    full_cpu.close()
    # parent of a singleton: all nodes sharing the first L characters, L = max of the two adjacent LCP values
    np_q = 4_000_000
    d_nodes = torch.zeros((np_q, 5), dtype=torch.int64, device=dev)
    gpu.parent_device(d_out.data_ptr(), np_q, d_nodes.data_ptr(), st)
    torch.cuda.synchronize()
    lcpv = torch.from_numpy(ix.lcp_data[: ix.n]).to(dev)
    psp, pep, L = dbg_parent_closed_form(dbg, lcpv, exp[:np_q], dbg.values_at(start[:np_q]))
    assert torch.equal(d_nodes[:, 0], psp) and torch.equal(d_nodes[:, 1], pep) and torch.equal(d_nodes[:, 4], L)
    assert int(d_nodes[:, 1].max().item()) > (1 << 32)
    del lcpv, d_nodes, d_loff, d_lval, d_cnt, d_out1, d_out, d_pat

    # config 5: 256-bp walks, every second one with a substitution every 41 bp
    nq, m = 1_000_000, 256
    pats, _, d_off, start, exp = batch(nq, m, 0x6C5A0050)
    nxt = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"CGTA"):
        nxt[a] = b
    for col in range(37, m, 41):
        pats[1::2, col] = nxt[pats[1::2, col].to(torch.int64)]
    d_pat = torch.zeros(nq * m + 8, dtype=torch.uint8, device=dev)
    d_pat[: nq * m] = pats.reshape(-1)
    d_find = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    gpu.find_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_find.data_ptr(), st)
    d_ms = torch.zeros(nq * m + 8, dtype=torch.int16, device=dev)
    d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    d_fb = torch.zeros(nq, dtype=torch.int64, device=dev)
    gpu.match_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(d_find[0::2, 0], exp[0::2]) and torch.equal(d_find[0::2, 1], exp[0::2])
    # (a substituted walk is found only if its changed 18-mers happen to be junction edges of the graph: almost never)
    assert float((d_find[1::2, 0] > d_find[1::2, 1]).to(torch.float64).mean().item()) > 0.95
    ms2d = d_ms[: nq * m].view(nq, m)
    assert torch.equal(d_rng[0::2, 0], exp[0::2]) and torch.equal(d_rng[0::2, 1], exp[0::2]) and bool((d_fb[0::2] == 0).all())
    assert bool((ms2d[0::2] == (m - torch.arange(m, device=dev)).to(torch.int16).view(1, m)).all())
    assert float(d_fb[1::2].to(torch.float64).mean().item()) > 5
    # locate() of the final ranges: count() values each (query_gcsa.cpp:171-179), closed form for the unmodified half
    d_loff = torch.zeros(nq + 1, dtype=torch.int64, device=dev)
    d_cnt = torch.zeros(nq, dtype=torch.int64, device=dev)
    gpu.count_device(d_rng.data_ptr(), nq, d_cnt.data_ptr(), st)
    torch.cuda.synchronize()
    total = int(d_cnt.sum().item())
    d_lval = torch.zeros(total, dtype=torch.int64, device=dev)
    assert gpu.locate_into(d_rng.data_ptr(), nq, d_loff.data_ptr(), d_lval.data_ptr(), total, st) == total
    torch.cuda.synchronize()
    assert torch.equal(d_loff[1:] - d_loff[:-1], d_cnt)
    assert np.array_equal(d_lval[d_loff[:-1][0::2]].cpu().numpy().view(np.uint64), mseq_torch.node_values(start[0::2].cpu().numpy()))
    # the oracle on a sample: find, matching statistics (LF + parent), parent of the final ranges
    ns = 3000
    flat = d_pat[: ns * m].cpu().numpy()
    off = np.arange(ns + 1, dtype=np.uint64) * np.uint64(m)
    assert np.array_equal(d_find[:ns].cpu().numpy().view(np.uint64), cpu.find_batch(flat, off, threads=8))
    cm, cr, cf = cpu.match_stats_batch(flat, off, threads=8)
    assert np.array_equal(d_ms[: ns * m].cpu().numpy().view(np.uint16), cm)
    assert np.array_equal(d_rng[:ns].cpu().numpy().view(np.uint64), cr)
    assert np.array_equal(d_fb[:ns].cpu().numpy().view(np.uint64), cf)
    d_nodes = torch.zeros((ns, 5), dtype=torch.int64, device=dev)
    gpu.parent_device(d_rng.data_ptr(), ns, d_nodes.data_ptr(), st)
    torch.cuda.synchronize()
    want = cpu.parent_batch(cr, threads=8)
    got = d_nodes.cpu().numpy().view(np.uint64)
    for col, name in enumerate(("sp", "ep", "left_lcp", "right_lcp", "node_lcp")):
        assert np.array_equal(got[:, col], want[name]), name
    # round 4 at this size.  (i) Break points: one record (0, 256, r, r) per unmodified pattern, in closed form; on the sample,
    # exactly what the oracle's dense statistics and find() imply (positions p with p == 0 or ms[p - 1] != ms[p] + 1, ranges
    # beyond 2^32); with a minimum length the same records without the shorter ones.
    from test_gpu_parity import breaks_from_dense
    d_boff = torch.zeros(nq + 1, dtype=torch.int64, device=dev)
    cap = 40 * nq
    d_brk = torch.zeros((cap, 4), dtype=torch.int64, device=dev)
    d_rng2 = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    d_fb2 = torch.zeros(nq, dtype=torch.int64, device=dev)
    n_brk = gpu.match_breaks_device(d_pat.data_ptr(), d_off.data_ptr(), nq, nq * m, d_boff.data_ptr(), d_brk.data_ptr(), cap, d_rng2.data_ptr(), d_fb2.data_ptr(), st)
    assert torch.equal(d_rng2, d_rng) and torch.equal(d_fb2, d_fb) and int(d_boff[-1].item()) == n_brk
    counts = d_boff[1:] - d_boff[:-1]
    first = d_brk[d_boff[:-1][0::2]]
    assert bool((counts[0::2] == 1).all()) and bool((first[:, 0] == 0).all()) and bool((first[:, 1] == m).all())
    assert torch.equal(first[:, 2], exp[0::2]) and torch.equal(first[:, 3], exp[0::2])
    sample_pats = [bytes(flat[q * m:(q + 1) * m]) for q in range(ns)]
    want_off, want_brk = breaks_from_dense(cpu, sample_pats, cm, off)
    assert np.array_equal(d_boff[: ns + 1].cpu().numpy().view(np.uint64), want_off)
    assert np.array_equal(d_brk[: int(want_off[-1])].cpu().numpy().view(np.uint64), want_brk)
    assert int(want_brk[:, 2].max()) > (1 << 32) and want_brk.shape[0] > 10 * ns
    n_mem = gpu.match_breaks_device(d_pat.data_ptr(), d_off.data_ptr(), nq, nq * m, d_boff.data_ptr(), d_brk.data_ptr(), cap, 0, 0, st, min_length=20)
    assert n_mem < n_brk // 3
    keep = want_brk[:, 1] >= 20
    n_keep = int(d_boff[ns].item())
    assert n_keep == int(keep.sum()) and np.array_equal(d_brk[:n_keep].cpu().numpy().view(np.uint64), want_brk[keep])
    del d_brk, d_boff, d_rng2, d_fb2, d_ms
    torch.cuda.empty_cache()
    # (ii) k-mer batches as 2-bit codes: the 20 M 32-mers of config 4 again, every range in closed form
    nq4 = 20_000_000
    pats4, _, _, _, exp4 = batch(nq4, 32, 0x6C5A0041)
    lut = torch.zeros(256, dtype=torch.int64, device=dev)
    for ch, c in zip(b"ACGT", range(4)):
        lut[ch] = c
    comps = lut[pats4.to(torch.int64)]
    code = torch.zeros(nq4, dtype=torch.int64, device=dev)
    for t in range(32):
        code |= comps[:, 31 - t] << (2 * t)
    del comps
    d_out4 = torch.zeros((nq4, 2), dtype=torch.int64, device=dev)
    gpu.find_packed_device(code.data_ptr(), 32, nq4, d_out4.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(d_out4[:, 0], exp4) and torch.equal(d_out4[:, 1], exp4)
    host_ranges = gpu.find_batch_packed(code[:3_000_000].cpu().numpy().view(np.uint64).reshape(-1, 1), 32)        # the chunked host pipeline, 40-bit wire
    assert np.array_equal(host_ranges, d_out4[:3_000_000].cpu().numpy().view(np.uint64))
    # (iii) the marked (wide) seed entries, reference semantics include/gcsa/gcsa.h:96-110: with the seed table cut to k = 4 every
    # entry covers n / 256 = 22 M path nodes, more than the 2^24 - 1 the length field holds, so every pattern is searched from
    # scratch -- and still lands on its closed form; then back to the full table
    k_full = gpu.kmer_table_k()
    gpu.set_tables(kmer_k=4)
    d_flat4 = torch.zeros(nq4 * 32 + 8, dtype=torch.uint8, device=dev)
    d_flat4[: nq4 * 32] = pats4.reshape(-1)
    d_off4 = torch.arange(nq4 + 1, dtype=torch.int64, device=dev) * 32
    d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
    d_out4.zero_()
    nw = 2_000_000
    gpu.find_stats_device(d_flat4.data_ptr(), d_off4.data_ptr(), nw, d_out4.data_ptr(), d_stats.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(d_stats[6].item()) == nw and int(d_stats[1].item()) == 31 * nw            # every seed entry wide: 31 LF steps per query
    assert torch.equal(d_out4[:nw, 0], exp4[:nw]) and torch.equal(d_out4[:nw, 1], exp4[:nw])
    gpu.set_tables(kmer_k=k_full)
    gpu.find_device(d_flat4.data_ptr(), d_off4.data_ptr(), nw, d_out4.data_ptr(), st)
    torch.cuda.synchronize()
    assert gpu.kmer_table_k() == k_full and torch.equal(d_out4[:nw, 0], exp4[:nw])
    # (iv) countKMers (src/algorithms.cpp:387-421) in closed form: the distinct k-prefixes of the node labels, from the bitmap of the
    # 17-mer universe (workload/dbg_torch.py::distinct_prefixes, pinned on the oracle at small degrees in tests/test_workload.py).
    # k = 16: 3.4 G k-mers; k = 17 = the order: one per path node, through a frontier of 3.4 G states cut into 64 M-state pieces
    order = degree // 2
    counted = {k: gpu.count_kmers(k) for k in (11, order - 1, order)}
    for k, got in counted.items():
        assert got == dbg_torch.distinct_prefixes(dbg.nodes, order, k), k
    assert counted[order] == ix.n and gpu.count_kmers(order + 1) == 0                 # beyond the order: 0 unless forced (algorithms.cpp:391-395)


def test_branching_footprint_index_closed_form():
    """The branching variant of the footprint generator (m-sequence text + one SNP bubble per 50 positions: order-k de
    Bruijn graph, e = 1.08 n; validated against its definition on the CPU in tests/test_workload.py) at 268 M path nodes:
    every find() of a walk through the graph -- pair steps that go through a non-last out-edge are replayed singly --
    equals the single node of the walk's first k characters, for 32-mers, 100-mers and k-mers, through the default and
    the length-bucketed launch."""
    import torch
    from workload import mseq_torch
    from gcsa2_amd.binding import GCSA
    dev = torch.device("cuda", 0)
    degree = 28
    ix, sym_t, rank, alt_t = mseq_torch.build_mseq_snp(degree, device=dev, with_lcp=True)
    rank_t = torch.from_numpy(rank.view(np.int32)).to(dev)
    assert 1.06 * ix.n < ix.e < 1.10 * ix.n
    gpu = GCSA(ix, device=0, with_samples=False, with_counters=False, with_lcp=True)
    assert gpu.pair_block_bytes() > 0
    st = torch.cuda.current_stream().cuda_stream
    for m, nq in ((32, 4_000_000), (100, 1_000_000), (degree // 2, 2_000_000)):
        pats, exp = mseq_torch.walk_patterns_device(sym_t, alt_t, rank_t, 0, nq, m, 0x6C5A0041 + m)
        d_pat = torch.zeros(nq * m + 8, dtype=torch.uint8, device=dev)
        d_pat[: nq * m] = pats.reshape(-1)
        d_off = torch.arange(nq + 1, dtype=torch.int64, device=dev) * m
        for variant in (2, 4):
            n = nq if variant == 2 else nq // 8
            d_out = torch.zeros((n, 2), dtype=torch.int64, device=dev)
            gpu.find_device_variant(variant, d_pat.data_ptr(), d_off.data_ptr(), n, d_out.data_ptr(), st)
            torch.cuda.synchronize()
            assert torch.equal(d_out[:, 0], exp[:n]) and torch.equal(d_out[:, 1], exp[:n]), (m, variant)
        d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
        d_out = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
        gpu.find_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_out.data_ptr(), d_stats.data_ptr(), st)
        torch.cuda.synchronize()
        assert torch.equal(d_out[:, 0], exp)
    # matching statistics (LF + parent) on the branching index: 256-bp walks, every second one with a substitution every
    # 41 bp.  Unmodified walks: closed form (full-depth match, no parent() call); the first 3000 patterns: the CPU oracle.
    from oracle.oracle import OracleIndex
    nq, m = 200_000, 256
    pats, exp = mseq_torch.walk_patterns_device(sym_t, alt_t, rank_t, 0, nq, m, 0x6C5A0051)
    nxt = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"CGTA"):
        nxt[a] = b
    for col in range(37, m, 41):
        pats[1::2, col] = nxt[pats[1::2, col].to(torch.int64)]
    d_pat = torch.zeros(nq * m + 8, dtype=torch.uint8, device=dev)
    d_pat[: nq * m] = pats.reshape(-1)
    d_off = torch.arange(nq + 1, dtype=torch.int64, device=dev) * m
    d_ms = torch.zeros(nq * m + 8, dtype=torch.int16, device=dev)
    d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    d_fb = torch.zeros(nq, dtype=torch.int64, device=dev)
    gpu.match_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), st)
    torch.cuda.synchronize()
    want_ms = (m - torch.arange(m, device=dev)).to(torch.int16).view(1, m)
    assert torch.equal(d_rng[0::2, 0], exp[0::2]) and torch.equal(d_rng[0::2, 1], exp[0::2])
    assert bool((d_fb[0::2] == 0).all()) and bool((d_ms[: nq * m].view(nq, m)[0::2] == want_ms).all())
    assert float(d_fb[1::2].to(torch.float64).mean().item()) > 5            # the substituted half does take parent()
    ns = 3000
    cpu = OracleIndex(ix, with_samples=False, with_counters=False)
    cm, cr, cf = cpu.match_stats_batch(d_pat[: ns * m].cpu().numpy(), np.arange(ns + 1, dtype=np.uint64) * np.uint64(m), threads=8)
    assert np.array_equal(d_ms[: ns * m].cpu().numpy().view(np.uint16), cm)
    assert np.array_equal(d_rng[:ns].cpu().numpy().view(np.uint64), cr) and np.array_equal(d_fb[:ns].cpu().numpy().view(np.uint64), cf)


def test_match_stats_large_host_batch_in_pieces(monkeypatch):
    """gcsa2_match_stats_batch cuts a batch of 64 MB or more of pattern bytes into pieces that several host threads send through
    the single-copy path concurrently (GCSA2_MS_PIECES=0: one copy, one launch).  Ragged lengths, so that piece boundaries fall
    anywhere: both ways give the same statistics, ranges and parent() counts, and the oracle agrees on a sample of patterns."""
    from workload import graphs, builder, patterns
    from oracle.oracle import OracleIndex, max_threads
    from gcsa2_amd import binding
    g = graphs.snp_graph(1 << 16, 0x6C5A0030, 0x6C5A0031)
    ix = builder.build(g, 64, keep_table=False)
    nq, m = 330_000, 256
    walks = patterns.walk_patterns(g, nq, m, 0x6C5A0032)
    for col in range(29, m, 47):                               # substitutions in two thirds of the patterns
        rows = np.arange(nq) % 3 != 0
        walks[rows, col] = np.frombuffer(b"CGTA", dtype=np.uint8)[np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), walks[rows, col]) % 4]
    rng = np.random.default_rng(0x33)
    lengths = rng.integers(200, m + 1, size=nq)
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    flat = np.ascontiguousarray(walks[np.arange(m)[None, :] < lengths[:, None]])          # row-major: pattern after pattern
    assert flat.shape[0] == int(off[-1]) and flat.shape[0] >= (64 << 20)
    gpu, _ = binding.open_index(ix)
    gm, gr, gf = gpu.match_stats_batch(flat, off)
    monkeypatch.setenv("GCSA2_MS_PIECES", "0")
    single, _ = binding.open_index(ix)
    sm, sr, sf = single.match_stats_batch(flat, off)
    assert np.array_equal(gm, sm) and np.array_equal(gr, sr) and np.array_equal(gf, sf)
    cpu = OracleIndex(ix)
    pick = np.sort(rng.choice(nq, size=4000, replace=False))
    sub_off = np.concatenate([[0], np.cumsum(lengths[pick])]).astype(np.uint64)
    sub = np.concatenate([flat[int(off[q]): int(off[q + 1])] for q in pick])
    cm, cr, cf = cpu.match_stats_batch(sub, sub_off, threads=max_threads())
    got = np.concatenate([gm[int(off[q]): int(off[q + 1])] for q in pick])
    assert np.array_equal(got, cm) and np.array_equal(gr[pick], cr) and np.array_equal(gf[pick], cf)
    assert int(gf.max()) > 0
    # the break points of the same batch (gcsa2_match_breaks_batch): in pieces that commit their records in order == one copy,
    # one launch; final ranges and parent() counts equal the dense run's; the records expand to the dense statistics; a
    # capacity that is too small is refused with the number of records of the whole batch, which then fits
    for min_length in (0, 12):
        pb, pr, prng, pfb = gpu.match_breaks_batch(flat, off, min_length=min_length, capacity=(5 if min_length == 0 else None))
        sb, sr_, srng, sfb = single.match_breaks_batch(flat, off, min_length=min_length)
        assert np.array_equal(pb, sb) and np.array_equal(pr, sr_) and np.array_equal(prng, srng) and np.array_equal(pfb, sfb), min_length
        assert np.array_equal(prng, gr) and np.array_equal(pfb, gf)
        assert int(pb[-1]) == pr.shape[0] and bool((np.diff(pb.astype(np.int64)) >= 0).all())
        if min_length > 0:
            assert bool((pr[:, 1] >= min_length).all())
    pb, pr, _, _ = gpu.match_breaks_batch(flat, off)
    for q in pick[:1500]:
        rec = pr[int(pb[q]): int(pb[q + 1])]                   # descending positions; ms[i] = length - (i - p) for the record with the largest p <= i
        want = gm[int(off[q]): int(off[q + 1])]
        L = int(lengths[q])
        dense = np.zeros(L, dtype=np.int64)
        end = L
        for pos, length, _sp, _ep in rec:
            pos, length = int(pos), int(length)
            dense[pos:end] = length - (np.arange(pos, end) - pos)
            end = pos
        assert end == 0 and np.array_equal(dense, want.astype(np.int64)), q
    gpu.close(); single.close()
    # pieces whose device-side record buffer was sized too small by the estimate (one record per 8 pattern bytes) run once more
    # with the exact size: random strings break at nearly every position.  (1 MB pieces, so that 3 MB make a pieced batch.)
    monkeypatch.setenv("GCSA2_MS_PIECES", "1")
    monkeypatch.setenv("GCSA2_MS_PIECE_MB", "1")
    small, _ = binding.open_index(ix)
    monkeypatch.setenv("GCSA2_MS_PIECES", "0")
    whole, _ = binding.open_index(ix)
    nr = 30_000
    rflat = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=nr * 100)].copy()
    roff = np.arange(nr + 1, dtype=np.uint64) * np.uint64(100)
    a = small.match_breaks_batch(rflat, roff)
    b = whole.match_breaks_batch(rflat, roff)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and a[1].shape[0] > nr * 100 // 4        # more than one record per 8 bytes
    small.close(); whole.close()
