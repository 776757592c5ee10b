"""GPU parity tests: the HIP engine, called through the C ABI (gcsa2_amd/binding.py -> ctypes ->
libgcsa2_hip.so), must equal the CPU oracle bit for bit on the same seeded inputs.

Small definitional indexes (bubbles, cycles, N, repeats) exercise every edge case the oracle is
pinned on; larger cases live in test_gpu_scale.py."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from workload import graphs
from workload.brute_builder import build
from workload.rng import SplitMix64
from gcsa2_amd.hostview import concat_patterns
from test_oracle import CASES, random_patterns, truncate_at_sink


@pytest.fixture(scope="module")
def engine():
    from gcsa2_amd import binding
    assert binding.device_count() >= 1, "no MI355X visible"
    return binding


@pytest.fixture(scope="module", params=range(len(CASES)), ids=[c[0] for c in CASES])
def case(request, engine):
    from oracle.oracle import OracleIndex
    name, g, K = CASES[request.param]
    ix = build(g, K, sample_period=8, branching=4)
    gpu, lcp = engine.open_index(ix, device=0)
    return name, g, K, ix, gpu, lcp, OracleIndex(ix)


def all_ranges(ix, seed, extra=200):
    rng = SplitMix64(seed)
    ranges = [(i, i) for i in range(ix.n)] + [(0, ix.n - 1)]
    for _ in range(extra):
        a = rng.below(ix.n)
        ranges.append((a, min(ix.n - 1, a + rng.below(8))))
    return ranges


def test_paper_example_on_gpu(engine, paper):
    ix = build(graphs.paper_graph(), paper["order"], sample_period=1 << 40)
    gpu, lcp = engine.open_index(ix)
    for q in paper["find"]:
        assert list(gpu.find(q["pattern"].encode())) == q["range"], q
    for q in paper["locate"]:
        assert list(gpu.locate(tuple(q["range"]))) == q["values"]
        assert gpu.count(tuple(q["range"])) == q["count"]
    assert gpu.LF((9, 12), 1) == (2, 4)
    assert gpu.find(b"") == (0, 15)
    assert gpu.size() == 16 and gpu.edgeCount() == 20 and gpu.order() == 3
    # the LCP family against the answers derived from the figure's keys (tests/golden/make_paper_lcp.py)
    from test_oracle import check_paper_suffix_tree
    for branching in (2, 3, 64):
        ix = build(graphs.paper_graph(), paper["order"], sample_period=1 << 40, branching=branching)
        gpu, lcp = engine.open_index(ix)
        check_paper_suffix_tree(paper, [int(x) for x in lcp.access_batch(np.arange(ix.n, dtype=np.uint64))],
                                lcp.parent, lcp.depth, lcp.psv, lcp.nsv, lcp.rmq, lcp.notFound())


def test_text_figure(engine, text_figure):
    """Figure 1 of the paper (text GCATCATA$: BWT, SA, LCP and LF columns) through the HIP path; the expected values are
    the figure's, the oracle is not consulted (tests/golden/make_text_example.py)."""
    from test_oracle import check_text_figure, text_example_index, text_position
    for branching in (2, 3, 64):
        ix = text_example_index(text_figure, branching=branching)
        gpu, lcp = engine.open_index(ix)
        n = int(ix.n)
        singles = np.array([[i, i] for i in range(n)], dtype=np.uint64)
        steps = [gpu.lf_batch(singles, np.full(n, c, dtype=np.uint8)) for c in range(7)]       # LF((i, i), c) is non-empty iff B_c[i]

        def pred_char(i):
            return "".join("$ACGTN#"[c] for c in range(7) if steps[c][i][0] + 1 <= steps[c][i][1] + 1)
        check_text_figure(text_figure, gpu.size(), pred_char, [int(x) for x in lcp.access_batch(np.arange(n, dtype=np.uint64))],
                          lambda i: int(gpu.lf_node_batch(np.array([i], dtype=np.uint64))[0]), gpu.locate, gpu.find,
                          (lcp.parent, lcp.depth, lcp.psv, lcp.nsv, lcp.rmq, lcp.notFound()))
    gpu, lcp = engine.open_index(text_example_index(text_figure, sample_period=4))
    for i in range(9):
        assert [text_position(v, text_figure) for v in gpu.locate((i, i))] == [text_figure["SA"][i]]


def test_find(case):
    name, g, K, ix, gpu, lcp, cpu = case
    pats = [truncate_at_sink(p) for p in random_patterns(g, K, 0x77, 600)]
    pats += [b"", b"A", b"N", b"#", b"$", b"x", b"acgt", b"\x00", bytes([200, 65])]
    data, off = concat_patterns(pats)
    got = gpu.find_batch(data, off)
    want = cpu.find_batch(data, off)
    assert np.array_equal(got, want), name
    for p in pats[:20]:
        assert gpu.find(p) == cpu.find(p)


def test_lf(case):
    name, g, K, ix, gpu, lcp, cpu = case
    ranges, comps = [], []
    for r in all_ranges(ix, 0x11):
        for c in range(ix.sigma):
            ranges.append(r); comps.append(c)
    ranges = np.array(ranges, dtype=np.uint64)
    comps = np.array(comps, dtype=np.uint8)
    assert np.array_equal(gpu.lf_batch(ranges, comps), cpu.lf_batch(ranges, comps)), name
    nodes = np.arange(ix.n, dtype=np.uint64)
    want = np.array([cpu.LF(int(i)) for i in nodes], dtype=np.uint64)
    assert np.array_equal(gpu.lf_node_batch(nodes), want), name
    for c in range(ix.sigma):
        if ix.C[c + 1] > 0:
            assert gpu.charRange(c) == cpu.charRange(c)
    rr = np.array(all_ranges(ix, 0x12, 50) + [(1, 0)], dtype=np.uint64)
    for all_ in (0, 1):
        got = gpu.lf_all_batch(rr, all_)
        for i, r in enumerate(rr):
            want = cpu.LF_all(tuple(int(x) for x in r)) if all_ else cpu.LF_fast(tuple(int(x) for x in r))
            assert [tuple(int(x) for x in t) for t in got[i]] == want, (name, r, all_)


def test_count_locate(case):
    name, g, K, ix, gpu, lcp, cpu = case
    ranges = all_ranges(ix, 0x13) + [(1, 0), (3, 2), (0, ix.n), (ix.n, ix.n + 3)]
    arr = np.array(ranges, dtype=np.uint64)
    assert np.array_equal(gpu.count_batch(arr), cpu.count_batch(arr)), name
    go, gv = gpu.locate_batch(arr)
    co, cv = cpu.locate_batch(arr)
    assert np.array_equal(go, co), name
    assert np.array_equal(gv, cv), name
    # count() == |locate()| on ranges produced by find (the query_gcsa consistency check,
    # reference benchmark/query_gcsa.cpp:171-179)
    pats = [truncate_at_sink(p)[:K] for p in random_patterns(g, K, 0x78, 300)]
    data, off = concat_patterns(pats)
    found = gpu.find_batch(data, off)
    offs, vals = gpu.locate_batch(found)
    assert np.array_equal(np.diff(offs), gpu.count_batch(found)), name
    # empty batch
    eo, ev = gpu.locate_batch(np.zeros((0, 2), dtype=np.uint64))
    assert eo.tolist() == [0] and ev.shape[0] == 0


def test_suffix_tree_ops(case):
    name, g, K, ix, gpu, lcp, cpu = case
    pos = np.arange(ix.n + 2, dtype=np.uint64)
    for op, fn in ((0, cpu.psv), (1, cpu.psev), (2, cpu.nsv), (3, cpu.nsev)):
        want = np.array([fn(int(p)) for p in pos], dtype=np.uint64)
        assert np.array_equal(lcp._sv_batch(op, pos), want), (name, op)
    ranges = all_ranges(ix, 0x14)
    rng = SplitMix64(0x15)
    for _ in range(200):
        a = rng.below(ix.n)
        ranges.append((a, min(ix.n - 1, a + rng.below(ix.n))))
    arr = np.array(ranges, dtype=np.uint64)
    assert np.array_equal(lcp.parent_batch(arr), cpu.parent_batch(arr)), name
    assert np.array_equal(lcp.depth_batch(arr), cpu.depth_batch(arr)), name
    want = np.array([cpu.rmq(int(a), int(b)) for a, b in arr] , dtype=np.uint64)
    assert np.array_equal(lcp.rmq_batch(arr), want), name
    assert lcp.rmq(3, 2) == cpu.rmq(3, 2)


def test_errors(engine):
    ix = build(graphs.paper_graph(), 3)
    g = engine.GCSA(ix, with_counters=False, with_lcp=False)
    with pytest.raises(engine.Gcsa2Error) as e:
        g.count((0, 1))
    assert e.value.code == -5
    with pytest.raises(engine.Gcsa2Error):
        engine.GCSA(ix, device=99)
    # pattern offsets of the host-pointer entry points must be non-decreasing (they index the pattern buffer)
    data = np.frombuffer(b"ACGTACGT", dtype=np.uint8).copy()
    bad = np.array([0, 5, 3, 8], dtype=np.uint64)
    with pytest.raises(engine.Gcsa2Error) as e:
        g.find_batch(data, bad)
    assert e.value.code == -1 and "non-decreasing" in str(e.value)
    full, lcp = engine.open_index(build(graphs.paper_graph(), 3, sample_period=2, branching=2))
    with pytest.raises(engine.Gcsa2Error) as e:
        full.match_stats_batch(data, bad)
    assert e.value.code == -1


def test_locate_modes_and_samples(case):
    name, g, K, ix, gpu, lcp, cpu = case
    ranges = all_ranges(ix, 0x16, 60)
    arr = np.array(ranges, dtype=np.uint64)
    # sort == false: path order with duplicates (reference src/gcsa.cpp:827-842)
    go, gv = gpu.locate_batch(arr, sort=False)
    for i, r in enumerate(ranges):
        assert gv[int(go[i]):int(go[i + 1])].tolist() == cpu.locate(r, sort=False).tolist(), (name, r)
    # locate(range, max_positions): same mt19937_64 draws as the reference (src/gcsa.cpp:844-878)
    for r in ranges[:: 7]:
        for mx in (1, 2, 5, 1000):
            assert gpu.locate(r, max_positions=mx).tolist() == cpu.locate(r, max_positions=mx).tolist(), (name, r, mx)
    # sampled / firstSample / sampleRange / sample / lastSample (reference gcsa.h:191-210)
    info = gpu.sample_range_batch(np.arange(ix.n, dtype=np.uint64))
    for i in range(ix.n):
        assert bool(info[i, 0]) == cpu.sampled(i)
        assert int(info[i, 1]) == cpu.firstSample(i)
    vals, last = gpu.sample_batch(np.arange(ix.sample_count, dtype=np.uint64))
    assert vals.tolist() == [cpu.sample(j) for j in range(ix.sample_count)]
    assert last.tolist() == [cpu.lastSample(j) for j in range(ix.sample_count)]
    assert gpu.sampledPositions() == sum(cpu.sampled(i) for i in range(ix.n))
    assert lcp.access_batch(np.arange(ix.n, dtype=np.uint64)).tolist() == ix.lcp_data[: ix.n].tolist()
    assert lcp.levels() == ix.lcp_offsets.shape[0] - 1 and lcp.branching() == ix.lcp_branching


def test_group_find_shards(engine):
    """Single-process multi-GPU API; on a 1-GPU box the replicas share device 0."""
    from oracle.oracle import OracleIndex
    name, g, K = CASES[-1]
    ix = build(g, K, sample_period=8, branching=4)
    cpu = OracleIndex(ix)
    pats = [truncate_at_sink(p) for p in random_patterns(g, K, 0x79, 101)] + [b"", b"N"]
    data, off = concat_patterns(pats)
    want = cpu.find_batch(data, off)
    for devices in ([0], [0, 0, 0], [0] * 7):
        grp = engine.GCSAGroup(ix, devices)
        assert grp.size() == len(devices)
        assert np.array_equal(grp.find_batch(data, off), want), devices
        one = concat_patterns(pats[:1])
        assert np.array_equal(grp.find_batch(*one), want[:1])
        grp.close()
    with pytest.raises(engine.Gcsa2Error):
        engine.GCSAGroup(ix, [0, 99])


def test_count_kmers(case):
    """countKMers (reference src/algorithms.cpp:387-421) as a device frontier expansion."""
    name, g, K, ix, gpu, lcp, cpu = case
    for k in range(0, K + 2):
        for ns in (False, True):
            assert gpu.count_kmers(k, include_Ns=ns) == cpu.count_kmers(k, include_Ns=ns, threads=2), (name, k, ns)
    assert gpu.count_kmers(K + 3, force=True) == cpu.count_kmers(K + 3, force=True)


def test_match_stats(case):
    """Fused LF + parent (matching statistics, the MEM-finder interplay) vs the oracle's composition
    of the restated reference primitives."""
    name, g, K, ix, gpu, lcp, cpu = case
    pats = [p for p in random_patterns(g, 3 * K, 0x92, 200)] + [b"", b"N", b"ACGTNACGT", b"$", b"#A"]
    data, off = concat_patterns(pats)
    gm, gr, gf = gpu.match_stats_batch(data, off)
    cm, cr, cf = cpu.match_stats_batch(data, off, threads=2)
    assert np.array_equal(gm, cm), name
    assert np.array_equal(gr, cr) and np.array_equal(gf, cf), name


@pytest.mark.parametrize("variant,knobs", [
    (5, {"GCSA2_MS_GRID": "2", "GCSA2_MS_REFILL_AT": "1"}),             # persistent lanes, eager refill
    (5, {"GCSA2_MS_GRID": "3"}),
    (5, {"GCSA2_MS_GRID": "1", "GCSA2_MS_REFILL_AT": "64"}),            # refill only when a wave is empty
    (5, {}),
    (0, {"GCSA2_COOL_DOWN": "0"}),
    (5, {"GCSA2_MS_GRID": "2", "GCSA2_COOL_DOWN": "12"}),
    (2, {}),                                                            # a lane per pattern
    (2, {"GCSA2_COOL_DOWN": "1000"}),
    (7, {}),                                                            # (round 5's k_match_stats3, variants 6 / 7, was retired in round 6: refused)
])
def test_match_stats_kernel_variants(case, engine, variant, knobs, monkeypatch):
    """Every launch shape of the matching-statistics kernels returns the oracle's statistics, ranges and parent() counts --
    in particular the persistent lanes that draw patterns from a counter, which small batches do not use by default.  The
    tuning knobs are read once, when an index is created; the variant is an argument."""
    import torch
    name, g, K, ix, gpu, lcp, cpu = case
    pats = [p for p in random_patterns(g, 3 * K, 0x95, 1500)] + [b"", b"N", b"", b"ACGTNACGT", b"$", b"#A", b""]
    data, off = concat_patterns(pats)
    cm, cr, cf = cpu.match_stats_batch(data, off, threads=2)
    for key, value in knobs.items():
        monkeypatch.setenv(key, value)
    tuned, _ = engine.open_index(ix, device=0)
    dev = torch.device("cuda", 0)
    nq, total = len(pats), int(off[-1])
    if variant not in (0, 2, 5):                               # an unknown variant is refused, whatever the batch
        with pytest.raises(engine.Gcsa2Error) as e:
            tuned.match_stats_device(0, 0, 0, 0, 0, 0, 0, variant=variant, total_bytes=0)
        assert e.value.code == -1                              # GCSA2_ERR_INVALID_ARGUMENT
        tuned.close()
        return
    d_pat = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
    d_pat[:total] = torch.from_numpy(data[:total]).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    for sized in (False, True):
        d_ms = torch.full((total + 8,), -3, dtype=torch.int16, device=dev)
        d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
        d_fb = torch.zeros(nq, dtype=torch.int64, device=dev)
        tuned.match_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), 0,
                                 variant=variant, total_bytes=(total if sized else None))
        torch.cuda.synchronize()
        assert np.array_equal(d_ms[:total].cpu().numpy().view(np.uint16), cm), (name, variant, knobs)
        assert np.array_equal(d_rng.cpu().numpy().view(np.uint64), cr) and np.array_equal(d_fb.cpu().numpy().view(np.uint64), cf), (name, variant, knobs)
    tuned.close()


def breaks_from_dense(cpu, pats, cm, off):
    """The break-point CSR the dense statistics imply (the definition in include/gcsa2_hip.h): position p is a break iff p == 0 or
    ms[p - 1] != ms[p] + 1; its length is ms[p], its range find() of that substring; per pattern in descending position."""
    offsets, records, subs = [0], [], []
    for q, p in enumerate(pats):
        ms = cm[int(off[q]):int(off[q + 1])].astype(np.int64)
        here = [i for i in range(len(p) - 1, -1, -1) if i == 0 or ms[i - 1] != ms[i] + 1]
        for i in here:
            records.append((i, int(ms[i])))
            subs.append(p[i:i + int(ms[i])])
        offsets.append(len(records))
    data, soff = concat_patterns(subs) if subs else (np.zeros(1, dtype=np.uint8), np.zeros(1, dtype=np.uint64))
    rng = cpu.find_batch(data, soff) if subs else np.zeros((0, 2), dtype=np.uint64)
    out = np.zeros((len(records), 4), dtype=np.uint64)
    for j, (i, length) in enumerate(records):
        out[j] = (i, length, rng[j, 0], rng[j, 1])
    return np.asarray(offsets, dtype=np.uint64), out


@pytest.mark.parametrize("variant", [0, 2, 5])
def test_match_breaks(case, engine, variant):
    """Matching statistics as break points (gcsa2_match_breaks_device): the CSR of left-maximal matches {position, length, sp,
    ep} equals what the oracle's dense statistics and find() imply -- every position where LF emptied and parent() was taken
    (the LF + parent interplay of src/algorithms.cpp:146-167), characters that do not occur (length 0 at the root), position 0,
    nothing for an empty pattern -- with the same final ranges and parent() counts as the dense kernel; a buffer that is too
    small is refused with the number of records needed."""
    import torch
    name, g, K, ix, gpu, lcp, cpu = case
    pats = [p for p in random_patterns(g, 3 * K, 0x9B, 1200)] + [b"", b"N", b"", b"ACGTNACGT", b"$", b"#A", b"", b"NNNN", b"A", b"TTTTTTTTTTTTTTTTTTTTTTTT"]
    data, off = concat_patterns(pats)
    cm, cr, cf = cpu.match_stats_batch(data, off, threads=2)
    want_off, want = breaks_from_dense(cpu, pats, cm, off)
    dev = torch.device("cuda", 0)
    nq, total = len(pats), int(off[-1])
    d_pat = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
    d_pat[:total] = torch.from_numpy(data[:total].copy()).to(dev)
    d_off = torch.from_numpy(off.view(np.int64).copy()).to(dev)
    d_boff = torch.full((nq + 1,), -1, dtype=torch.int64, device=dev)
    d_brk = torch.full((len(want) + 3, 4), -1, dtype=torch.int64, device=dev)
    d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    d_fb = torch.zeros(nq, dtype=torch.int64, device=dev)
    for sized in (total, None):
        d_brk.fill_(-1)
        n = gpu.match_breaks_device(d_pat.data_ptr(), d_off.data_ptr(), nq, sized, d_boff.data_ptr(), d_brk.data_ptr(), d_brk.shape[0],
                                    d_rng.data_ptr(), d_fb.data_ptr(), 0, variant=variant)
        assert n == len(want), (name, n, len(want))
        assert np.array_equal(d_boff.cpu().numpy().view(np.uint64), want_off), name
        got = d_brk.cpu().numpy().view(np.uint64)
        assert np.array_equal(got[:n], want), name
        assert (d_brk[n:] == -1).all()
        assert np.array_equal(d_rng.cpu().numpy().view(np.uint64), cr) and np.array_equal(d_fb.cpu().numpy().view(np.uint64), cf), name
    # a minimum length: the same records without the shorter ones
    for min_length in (1, 2, K, 2 * K):
        keep = want[:, 1] >= min_length
        n = gpu.match_breaks_device(d_pat.data_ptr(), d_off.data_ptr(), nq, total, d_boff.data_ptr(), d_brk.data_ptr(), d_brk.shape[0],
                                    d_rng.data_ptr(), d_fb.data_ptr(), 0, variant=variant, min_length=min_length)
        assert n == int(keep.sum()) and np.array_equal(d_brk.cpu().numpy().view(np.uint64)[:n], want[keep]), (name, min_length)
        kept_per_pattern = np.add.reduceat(np.concatenate([keep, [False]]).astype(np.uint64), np.minimum(want_off[:-1], len(keep)).astype(np.int64))
        kept_per_pattern[want_off[1:] == want_off[:-1]] = 0
        assert np.array_equal(np.diff(d_boff.cpu().numpy().view(np.uint64)), kept_per_pattern), (name, min_length)
        assert np.array_equal(d_rng.cpu().numpy().view(np.uint64), cr) and np.array_equal(d_fb.cpu().numpy().view(np.uint64), cf)
    # without the optional outputs; then a buffer that is too small
    n = gpu.match_breaks_device(d_pat.data_ptr(), d_off.data_ptr(), nq, total, d_boff.data_ptr(), d_brk.data_ptr(), d_brk.shape[0], variant=variant)
    assert n == len(want) and np.array_equal(d_brk.cpu().numpy().view(np.uint64)[:n], want)
    with pytest.raises(engine.Gcsa2Error) as e:
        gpu.match_breaks_device(d_pat.data_ptr(), d_off.data_ptr(), nq, total, d_boff.data_ptr(), d_brk.data_ptr(), len(want) // 2, variant=variant)
    assert e.value.code == -6 and e.value.needed == len(want)
    assert gpu.match_breaks_device(d_pat.data_ptr(), d_off.data_ptr(), 0, 0, d_boff.data_ptr(), d_brk.data_ptr(), 4, variant=variant) == 0


def test_match_stats_ragged_host_batch(engine):
    """A large batch of ragged lengths through the host-pointer entry point, which sends it to the persistent lanes
    (gcsa2_match_stats_batch: longest pattern > 1.25 x the mean, at least 2^19 patterns)."""
    from oracle.oracle import OracleIndex, max_threads
    name, g, K = CASES[-1]
    ix = build(g, K, sample_period=8, branching=4)
    gpu, lcp = engine.open_index(ix)
    cpu = OracleIndex(ix)
    rng = np.random.default_rng(0x96)
    nq = (1 << 19) + 77
    lengths = rng.integers(0, 41, size=nq)
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    data = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.choice(5, size=int(off[-1]), p=[0.245, 0.245, 0.245, 0.245, 0.02])].copy()
    gm, gr, gf = gpu.match_stats_batch(data, off)
    cm, cr, cf = cpu.match_stats_batch(data, off, threads=max_threads())
    assert np.array_equal(gm, cm) and np.array_equal(gr, cr) and np.array_equal(gf, cf)


def test_find_host_pipeline_pageable_and_page_locked(engine):
    """gcsa2_find_batch of 2^19 or more patterns runs chunked over several streams (find_pipelined).  Ragged lengths (chunk
    bases at every phase of the kernel's aligned word reads, empty patterns included), from pageable memory -- staged through
    the lanes' pinned sets -- and from page-locked memory, which the copy engines read and write in place: same ranges as the
    single-launch device path and as the oracle on a sample; each buffer may be page-locked independently."""
    import torch
    from oracle.oracle import OracleIndex
    name, g, K = CASES[-1]
    ix = build(g, K, sample_period=8, branching=4)
    gpu, lcp = engine.open_index(ix)
    cpu = OracleIndex(ix)
    rng = np.random.default_rng(0x97)
    nq = (1 << 19) + 4099
    lengths = rng.integers(0, 37, size=nq)
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    data = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.choice(5, size=int(off[-1]), p=[0.245, 0.245, 0.245, 0.245, 0.02])].copy()
    dev = torch.device("cuda", 0)
    d_pat = torch.zeros(int(off[-1]) + 16, dtype=torch.uint8, device=dev)
    d_pat[: int(off[-1])] = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_out = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    gpu.find_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_out.data_ptr(), 0)
    torch.cuda.synchronize()
    want = d_out.cpu().numpy().view(np.uint64)
    sample = rng.choice(nq, size=3000, replace=False)
    for q in sample:
        assert tuple(int(x) for x in want[q]) == tuple(cpu.find(bytes(data[int(off[q]):int(off[q + 1])]))), q
    assert np.array_equal(gpu.find_batch(data, off), want)                      # pageable
    p_data = torch.empty(data.shape[0], dtype=torch.uint8).pin_memory()
    p_off = torch.empty(nq + 1, dtype=torch.int64).pin_memory()
    p_out = torch.empty((nq, 2), dtype=torch.int64).pin_memory()
    p_data.numpy()[:] = data
    p_off.numpy().view(np.uint64)[:] = off
    pinned = (p_data.numpy(), p_off.numpy().view(np.uint64), p_out.numpy().view(np.uint64))
    for mask in range(1, 8):                                                    # which of the three buffers are page-locked
        a = pinned[0] if mask & 1 else data
        b = pinned[1] if mask & 2 else off
        c = pinned[2] if mask & 4 else np.zeros((nq, 2), dtype=np.uint64)
        c[:] = 7
        got = gpu.find_batch(a, b, out=c)
        assert np.array_equal(got, want), mask
    # a batch that does not start at the beginning of the page-locked arrays
    skip = 1234
    got = gpu.find_batch(pinned[0], pinned[1][skip:], out=pinned[2][skip:])
    assert np.array_equal(got, want[skip:])
    # patterns of one length: the chunk's offsets are generated on the device instead of sent (chunk bases at odd phases)
    m = 15
    assert nq * m <= data.shape[0]
    uni_off = (np.arange(nq + 1, dtype=np.uint64) * np.uint64(m)) + np.uint64(0)
    uni = data[: nq * m]
    d_off2 = torch.from_numpy(uni_off.view(np.int64)).to(dev)
    gpu.find_device(d_pat.data_ptr(), d_off2.data_ptr(), nq, d_out.data_ptr(), 0)
    torch.cuda.synchronize()
    want_uni = d_out.cpu().numpy().view(np.uint64)
    assert np.array_equal(gpu.find_batch(uni, uni_off), want_uni)
    p_off.numpy().view(np.uint64)[:] = uni_off
    assert np.array_equal(gpu.find_batch(pinned[0][: nq * m], pinned[1], out=pinned[2]), want_uni)
    p_off.numpy().view(np.uint64)[:] = off
    bad = pinned[1].copy()
    bad[nq // 2] = bad[nq // 2 + 1] + 5
    with pytest.raises(engine.Gcsa2Error):
        gpu.find_batch(pinned[0], bad)


@pytest.mark.gpu
def test_host_pipeline_shapes(engine):
    """gcsa2_index_set_pipeline: whatever the shape of the host pipeline -- one lane or sixteen, chunks of 2^15 or 2^20
    patterns, lanes that spin or sleep -- gcsa2_find_batch and gcsa2_find_batch_packed return the ranges of the single-launch
    device path (a batch that gives the lanes fewer than two full chunks each is cut finer); arguments out of range are refused
    and leave the shape as it was."""
    import torch
    from gcsa2_amd.hostview import pack_kmers
    name, g, K = CASES[-1]
    ix = build(g, K, sample_period=8, branching=4)
    gpu, lcp = engine.open_index(ix)
    rng = np.random.default_rng(0x98)
    m, nq = 21, (1 << 19) + 777
    arr = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(nq, m))].copy()
    walks = [p for p in random_patterns(g, m, 0x99, 4000) if len(p) == m and all(c in b"ACGT" for c in p)]
    if walks:
        hits = np.frombuffer(b"".join(walks), dtype=np.uint8).reshape(-1, m)
        arr[::3] = hits[rng.integers(0, hits.shape[0], size=arr[::3].shape[0])]
    off = np.arange(nq + 1, dtype=np.uint64) * np.uint64(m)
    dev = torch.device("cuda", 0)
    d_pat = torch.zeros(nq * m + 16, dtype=torch.uint8, device=dev)
    d_pat[: nq * m] = torch.from_numpy(arr.reshape(-1)).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_out = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    gpu.find_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_out.data_ptr(), 0)
    torch.cuda.synchronize()
    want = d_out.cpu().numpy().view(np.uint64)
    codes = pack_kmers(arr, ix.char2comp)
    for lanes, chunk, blocking in ((1, 15, 0), (16, 20, 1), (3, 16, 1), (6, 18, 0), (0, 0, -1)):
        gpu.set_pipeline(lanes, chunk, blocking)
        assert np.array_equal(gpu.find_batch(arr.reshape(-1), off), want), (lanes, chunk, blocking)
        assert np.array_equal(gpu.find_batch_packed(codes, m), want), (lanes, chunk, blocking)
    for bad in ((17, 18, 0), (6, 21, 0), (6, 14, 0)):
        with pytest.raises(engine.Gcsa2Error) as e:
            gpu.set_pipeline(*bad)
        assert e.value.code == -1
    assert np.array_equal(gpu.find_batch_packed(codes, m), want)


def widen_alphabet(ix):
    """Same index over a 9-letter alphabet: two never-occurring comps are inserted after T, so
    N becomes comp 7 and # comp 8.  Exercises sigma != 7 (sigma > 8 disables the pred4 nibbles and
    routes locate through the probing LF(path_node) walk)."""
    import copy
    wide = copy.copy(ix)
    order = [0, 1, 2, 3, 4, None, None, 5, 6]          # new comp -> old comp
    wide.sigma = 9
    zero = np.zeros_like(ix.bwt[0])
    wide.bwt = [zero if o is None else ix.bwt[o] for o in order]
    C = [0]
    for o in order:
        C.append(C[-1] + (0 if o is None else int(ix.C[o + 1]) - int(ix.C[o])))
    wide.C = np.array(C, dtype=np.uint64)
    remap = np.zeros(7, dtype=np.uint8)
    for new, o in enumerate(order):
        if o is not None:
            remap[o] = new
    wide.char2comp = remap[ix.char2comp]
    return wide


def test_other_alphabet_size(engine):
    from oracle.oracle import OracleIndex
    name, g, K = CASES[-1]
    ix = widen_alphabet(build(g, K, sample_period=8, branching=4))
    gpu, lcp = engine.open_index(ix)
    cpu = OracleIndex(ix)
    base = OracleIndex(build(g, K, sample_period=8, branching=4))
    pats = [truncate_at_sink(p) for p in random_patterns(g, K, 0x93, 300)] + [b"", b"N", b"#", b"$", b"ANA"]
    data, off = concat_patterns(pats)
    ranges = gpu.find_batch(data, off)
    assert np.array_equal(ranges, cpu.find_batch(data, off))
    assert np.array_equal(ranges, base.find_batch(data, off))        # relabelling does not change results
    arr = np.array(all_ranges(ix, 0x17, 80), dtype=np.uint64)
    go, gv = gpu.locate_batch(arr)
    co, cv = cpu.locate_batch(arr)
    assert np.array_equal(go, co) and np.array_equal(gv, cv)
    assert np.array_equal(gpu.count_batch(arr), cpu.count_batch(arr))
    nodes = np.arange(ix.n, dtype=np.uint64)
    assert gpu.lf_node_batch(nodes).tolist() == [cpu.LF(int(i)) for i in nodes]
    for all_ in (0, 1):
        got = gpu.lf_all_batch(arr[:50], all_)
        for i, r in enumerate(arr[:50]):
            r = tuple(int(x) for x in r)
            want = cpu.LF_all(r) if all_ else cpu.LF_fast(r)
            assert [tuple(int(x) for x in t) for t in got[i]] == want
    for k in range(0, 5):
        assert gpu.count_kmers(k, include_Ns=True) == cpu.count_kmers(k, include_Ns=True)
    gm, gr, gf = gpu.match_stats_batch(data, off)
    cm, cr, cf = cpu.match_stats_batch(data, off)
    assert np.array_equal(gm, cm) and np.array_equal(gr, cr)


def test_find_variants_on_device(case):
    """Both launch shapes of k_find2 (2 = one lane per query in batch order, 4 = queries ordered by pattern length first)
    through the device-pointer entry point, ragged pattern lengths; the variants dropped in round 3 are refused."""
    import torch
    name, g, K, ix, gpu, lcp, cpu = case
    pats = [truncate_at_sink(p) for p in random_patterns(g, 3 * K, 0x94, 500)] + [b"", b"N", b"$"]
    data, off = concat_patterns(pats)
    want = cpu.find_batch(data, off)
    dev = torch.device("cuda", 0)
    d_pat = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    for variant in (2, 4):
        d_out = torch.full((len(pats), 2), -7, dtype=torch.int64, device=dev)
        gpu.find_device_variant(variant, d_pat.data_ptr(), d_off.data_ptr(), len(pats), d_out.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want), (name, variant)
    from gcsa2_amd.binding import Gcsa2Error
    for variant in (1, 5, 7):
        with pytest.raises(Gcsa2Error):
            gpu.find_device_variant(variant, d_pat.data_ptr(), d_off.data_ptr(), len(pats), d_out.data_ptr(), 0)


def test_create_from_container_file(engine, tmp_path):
    """G2HV file -> device image: same results as creating from the in-memory view."""
    from oracle.oracle import OracleIndex
    name, g, K = CASES[-1]
    ix = build(g, K, sample_period=8, branching=4)
    path = str(tmp_path / "index.g2hv")
    engine.save_host_view(ix, path)
    gpu = engine.GCSA(path)
    cpu = OracleIndex(ix)
    pats = [truncate_at_sink(p) for p in random_patterns(g, K, 0x95, 200)]
    data, off = concat_patterns(pats)
    ranges = gpu.find_batch(data, off)
    assert np.array_equal(ranges, cpu.find_batch(data, off))
    go, gv = gpu.locate_batch(ranges)
    co, cv = cpu.locate_batch(ranges)
    assert np.array_equal(go, co) and np.array_equal(gv, cv)
    assert (gpu.size(), gpu.edgeCount(), gpu.order(), gpu.sigma) == (ix.n, ix.e, ix.order, ix.sigma)
    assert gpu.char2comp.tolist() == ix.char2comp.tolist()


def test_create_from_gcsa_files(engine, tmp_path):
    """`.gcsa` + `.lcp` byte streams -> device image: find / locate / count / parent equal the oracle
    (encodings as restated in workload/sdsl_format.py: format parity unpinned, see sdsl_reader.hpp)."""
    from oracle.oracle import OracleIndex
    from workload import sdsl_format
    name, g, K = CASES[-1]
    ix = build(g, K, sample_period=8, branching=4)
    gcsa_path, _ = sdsl_format.write(ix, str(tmp_path / "index"))
    gpu, lcp = engine.open_index(gcsa_path)
    cpu = OracleIndex(ix)
    pats = [truncate_at_sink(p) for p in random_patterns(g, K, 0x96, 200)]
    data, off = concat_patterns(pats)
    ranges = gpu.find_batch(data, off)
    assert np.array_equal(ranges, cpu.find_batch(data, off))
    go, gv = gpu.locate_batch(ranges)
    co, cv = cpu.locate_batch(ranges)
    assert np.array_equal(go, co) and np.array_equal(gv, cv)
    assert np.array_equal(gpu.count_batch(ranges), cpu.count_batch(ranges))
    found = ranges[ranges[:, 0] <= ranges[:, 1]]
    assert np.array_equal(lcp.parent_batch(found), cpu.parent_batch(found))
    assert (gpu.size(), gpu.edgeCount(), gpu.order(), gpu.sigma, lcp.size()) == (ix.n, ix.e, ix.order, ix.sigma, ix.lcp_size)


@pytest.mark.parametrize("piece", ["", "4", "50"])
def test_compare_kmers(engine, monkeypatch, piece):
    """compareKMers (reference src/algorithms.cpp:534-616) on two graphs sharing a backbone.  The search tree is walked
    depth-first over pieces of the frontier whose children fit one buffer (2^27 states; GCSA2_KMER_PIECE cuts it to 4 / 50
    states here, so that every level is searched in many pieces): counts and record sets are the same however it is cut,
    and a record buffer that is too small is refused with the counts."""
    from oracle.oracle import OracleIndex
    if piece:
        monkeypatch.setenv("GCSA2_KMER_PIECE", piece)
    g1 = graphs.snp_graph(400, 0x52, 0x53, snp_period=8, node_len=8)
    g2 = graphs.snp_graph(400, 0x52, 0x99, snp_period=6, node_len=8)
    i1, i2 = build(g1, 8, sample_period=8, branching=4), build(g2, 8, sample_period=8, branching=4)
    ga, gb = engine.GCSA(i1), engine.GCSA(i2)
    if piece:
        for k in (3, 6, 9):                              # countKMers in pieces too
            assert ga.count_kmers(k) == OracleIndex(i1).count_kmers(k), k
    ca, cb = OracleIndex(i1), OracleIndex(i2)
    for k in range(0, 10):
        for ns in (False, True):
            assert ga.compare_kmers(gb, k, include_Ns=ns) == ca.compare_kmers(cb, k, include_Ns=ns), (k, ns)
    assert ga.compare_kmers(gb, 12, force=True) == ca.compare_kmers(cb, 12, force=True)
    assert ga.compare_kmers(ga, 6) == (ga.count_kmers(6), 0, 0)
    # the states of the unique k-mers (the reference's .left / .right dumps): same set of 64-byte records
    def canon(rows):
        return sorted(tuple(int(x) for x in r) for r in rows)
    for k, ns in ((1, False), (5, False), (7, True), (12, False)):
        gc, gl, gr = ga.compare_kmers_records(gb, k, include_Ns=ns, force=True)
        cc, cl, cr = ca.compare_kmers_records(cb, k, include_Ns=ns, force=True)
        assert gc == cc and canon(gl) == canon(cl) and canon(gr) == canon(cr), (k, ns)
        assert len(gl) == gc[1] and len(gr) == gc[2]
    # record buffers that are too small: refused, with the counts
    import ctypes as C
    counts = ga.compare_kmers(gb, 7)
    assert counts[1] > 1 and counts[2] > 1
    out, small = np.zeros(3, dtype=np.uint64), np.zeros((1, 8), dtype=np.uint64)
    rc = ga._L.gcsa2_compare_kmers_records(ga._h, gb._h, 7, 0, 0, out.ctypes.data_as(C.POINTER(C.c_uint64)), small.ctypes.data_as(C.POINTER(C.c_uint64)), 1,
                                           small.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
    assert rc == -6 and tuple(int(x) for x in out) == counts


def test_locate_table_and_walk_agree(engine, monkeypatch):
    """locate() through the memoised table (default) and through the walk kernel
    (GCSA2_LOCATE_TABLE=0) both equal the oracle, sorted and unsorted, single-node ranges
    (fast path: no duplicate removal needed) and wide ranges alike."""
    from oracle.oracle import OracleIndex
    from workload import builder
    g = graphs.snp_graph(6000, 0xA1, 0xA2, snp_period=10, node_len=16)
    ix = builder.build(g, 16, sample_period=16, branching=8)
    cpu = OracleIndex(ix)
    rng = SplitMix64(0xA3)
    singles = np.array([(v, v) for v in (rng.below(ix.n) for _ in range(4000))], dtype=np.uint64)
    wide = []
    for _ in range(300):
        a = rng.below(ix.n)
        wide.append((a, min(ix.n - 1, a + rng.below(40))))
    wide += [(1, 0), (0, ix.n - 1), (ix.n, ix.n + 1)]
    wide = np.array(wide, dtype=np.uint64)
    engines = []
    with_table = engine.GCSA(ix)
    assert with_table.locate_table_bytes() == 8 * ix.n
    engines.append(with_table)
    monkeypatch.setenv("GCSA2_LOCATE_TABLE", "0")
    without = engine.GCSA(ix)
    assert without.locate_table_bytes() == 0
    engines.append(without)
    for arr in (singles, wide):
        for sort in (True, False):
            co, cv = cpu.locate_batch(arr) if sort else (None, None)
            for gpu in engines:
                go, gv = gpu.locate_batch(arr, sort=sort)
                if sort:
                    assert np.array_equal(go, co) and np.array_equal(gv, cv)
                else:
                    want = [cpu.locate((int(a), int(b)), sort=False) for a, b in arr]
                    assert np.array_equal(np.diff(go), [len(w) for w in want])
                    assert np.array_equal(gv, np.concatenate(want) if len(want) else gv)


def test_memory_ladder(engine, monkeypatch):
    """The optional tables (pair blocks, k-mer seed table, locate table) are memoisations: find(), locate() and the matching
    statistics equal the oracle on every rung of the memory ladder, whether the rung is reached by re-shaping a live image
    (gcsa2_index_set_tables: drop, rebuild, resize) or by creating the image under a cap (GCSA2_MEMORY_BUDGET_MB: seed table
    first, then pair blocks, then a larger seed table, then the locate table)."""
    from oracle.oracle import OracleIndex
    from workload import builder
    g = graphs.snp_graph(40000, 0xB1, 0xB2, snp_period=12, node_len=16)
    ix = builder.build(g, 16, sample_period=16, branching=8)
    cpu = OracleIndex(ix)
    pats = [truncate_at_sink(p) for p in random_patterns(g, 40, 0xB3, 3000)]
    rng = SplitMix64(0xB4)
    pats += [bytes(b"ACGTN"[rng.below(5)] for _ in range(1 + rng.below(20))) for _ in range(500)]
    data, off = concat_patterns(pats)
    want = cpu.find_batch(data, off)
    co, cv = cpu.locate_batch(want)
    cm, cr, cf = cpu.match_stats_batch(data, off, threads=2)

    def check(gpu, tag):
        got = gpu.find_batch(data, off)
        assert np.array_equal(got, want), tag
        go, gv = gpu.locate_batch(got)
        assert np.array_equal(go, co) and np.array_equal(gv, cv), tag
        gm, gr, gf = gpu.match_stats_batch(data, off)
        assert np.array_equal(gm, cm) and np.array_equal(gr, cr) and np.array_equal(gf, cf), tag

    pair_bytes, locate_bytes = 16 * (ix.n // 192 + 1) * 128, 8 * ix.n
    gpu, lcp = engine.open_index(ix, device=0)
    full, k_full = gpu.device_bytes(), gpu.kmer_table_k()
    assert gpu.pair_block_bytes() == pair_bytes and gpu.locate_table_bytes() == locate_bytes and k_full >= 6
    check(gpu, "everything")
    gpu.set_tables(locate_table=0)
    assert gpu.locate_table_bytes() == 0 and gpu.device_bytes() == full - locate_bytes
    check(gpu, "no locate table")
    gpu.set_tables(kmer_k=k_full - 1)
    assert gpu.kmer_table_k() == k_full - 1 and gpu.device_bytes() == full - locate_bytes - 6 * 4 ** k_full
    check(gpu, "smaller seed table")
    # (bench.py's ladder: seed-table levels go before the pair blocks -- a level frees 3/4 of the table for one LF step per query,
    # the pair blocks halve the requests of every remaining step; then the seed table's value without pair blocks)
    gpu.set_tables(kmer_k=k_full - 3)
    assert gpu.kmer_table_k() == k_full - 3 and gpu.pair_block_bytes() == pair_bytes
    check(gpu, "pair blocks, seed table three levels down")
    gpu.set_tables(pair_blocks=0, kmer_k=k_full)
    assert gpu.pair_block_bytes() == 0 and gpu.kmer_table_k() == k_full and gpu.device_bytes() == full - locate_bytes - pair_bytes
    check(gpu, "no pair blocks, full seed table")
    gpu.set_tables(pair_blocks=0, kmer_k=4)
    assert gpu.pair_block_bytes() == 0 and gpu.kmer_table_k() == 4
    bare = gpu.device_bytes() - 8 * 4 ** 4
    check(gpu, "no pair blocks, k = 4")
    gpu.set_tables(kmer_k=0)
    assert gpu.kmer_table_k() == 0 and gpu.device_bytes() == bare
    check(gpu, "bare image")
    gpu.set_tables(pair_blocks=1, kmer_k=k_full, locate_table=1)           # and back up
    assert gpu.device_bytes() == full and gpu.pair_block_bytes() == pair_bytes and gpu.locate_table_bytes() == locate_bytes
    check(gpu, "rebuilt")
    with pytest.raises(engine.Gcsa2Error):
        gpu.set_tables(kmer_k=17)
    gpu.close()

    # the same ladder by creating under a cap; MB with fractions, since this index is small
    mb = 1048576.0
    seen = []
    for budget in (full + 1, full - locate_bytes // 2, bare + pair_bytes + 8 * 4 ** 6 + 64, bare + pair_bytes // 2, bare // 2):
        monkeypatch.setenv("GCSA2_MEMORY_BUDGET_MB", repr(budget / mb))
        capped, _ = engine.open_index(ix, device=0)
        assert capped.device_bytes() <= max(budget, bare), budget          # the image proper is never refused: only tables are dropped
        seen.append((capped.pair_block_bytes() > 0, capped.kmer_table_k(), capped.locate_table_bytes() > 0))
        check(capped, f"budget {budget}")
        capped.close()
    monkeypatch.delenv("GCSA2_MEMORY_BUDGET_MB")
    assert seen[0] == (True, k_full, True), seen
    assert seen[1][0] and not seen[1][2] and seen[1][1] >= k_full - 1, seen        # the locate table goes first
    assert seen[2][0] and not seen[2][2] and 5 <= seen[2][1] < k_full, seen        # then the seed table shrinks
    assert not seen[3][0] and not seen[3][2] and seen[3][1] >= 5, seen             # then the pair blocks go, a seed table stays
    assert seen[4] == (False, 0, False), seen                                      # below the image itself: no tables at all


def test_locate_many_small_calls(engine):
    """Thousands of one-range locate() calls in a row (the access pattern of locate(range, max_positions),
    gcsa.cpp:859-871): every call returns exactly count() values.  Guards the host read-backs of the
    pipeline (stale reads out of stream-ordered pool memory were seen at a rate of ~1 in 5000 calls)."""
    from workload import builder
    g = graphs.snp_graph(2000, 0x71, 0x72, snp_period=12, node_len=16)
    ix = builder.build(g, 16, sample_period=16, branching=8)
    gpu = engine.GCSA(ix)
    rng = SplitMix64(0xB1)
    nodes = np.array([rng.below(ix.n) for _ in range(6000)], dtype=np.uint64)
    counts = gpu.count_batch(np.stack([nodes, nodes], axis=1))
    for v, c in zip(nodes, counts):
        offs, vals = gpu.locate_batch(np.array([(v, v)], dtype=np.uint64))
        assert offs.tolist() == [0, int(c)] and len(vals) == int(c), int(v)


def test_locate_into_caller_buffers(engine):
    """gcsa2_locate_into: same CSR result as locate_run/fetch, written into caller-owned device
    buffers; an undersized buffer is refused and reports the size needed."""
    import torch
    from oracle.oracle import OracleIndex
    name, g, K = CASES[-1]
    ix = build(g, K, sample_period=8, branching=4)
    gpu, cpu = engine.GCSA(ix), OracleIndex(ix)
    ranges = np.array(all_ranges(ix, 0x17) + [(1, 0), (0, ix.n - 1)], dtype=np.uint64)
    dev = torch.device("cuda", 0)
    d_r = torch.from_numpy(ranges.view(np.int64)).to(dev)
    for sort in (True, False):
        if sort:
            co, cv = cpu.locate_batch(ranges)
        else:
            parts = [cpu.locate((int(a), int(b)), sort=False) for a, b in ranges]
            co = np.concatenate([[0], np.cumsum([len(x) for x in parts])]).astype(np.uint64)
            cv = np.concatenate(parts)
        raw = sum(len(cpu.locate((int(a), int(b)), sort=False)) for a, b in ranges)       # values before removeDuplicates
        assert raw > len(cv) + 5 or not sort, "the case has duplicates to remove"
        # capacities: the distinct values and a few more (the pipeline sorts in scratch and compacts into the buffer), and room
        # for the values BEFORE deduplication (round 6: the buffer is then the sorts' target and is compacted in place); what
        # lies between the total and the capacity is unspecified, nothing is written behind the capacity
        for capacity in (len(cv) + 5, raw + 3):
            d_o = torch.full((len(ranges) + 1,), -1, dtype=torch.int64, device=dev)
            d_v = torch.full((capacity + 64,), -1, dtype=torch.int64, device=dev)
            total = gpu.locate_into(d_r.data_ptr(), len(ranges), d_o.data_ptr(), d_v.data_ptr(), capacity, sort=sort)
            assert total == len(cv)
            assert np.array_equal(d_o.cpu().numpy().view(np.uint64), co)
            assert np.array_equal(d_v.cpu().numpy().view(np.uint64)[:total], cv)
            assert (d_v[capacity:] == -1).all()
        d_v.fill_(-1)
        with pytest.raises(engine.Gcsa2Error) as e:
            gpu.locate_into(d_r.data_ptr(), len(ranges), d_o.data_ptr(), d_v.data_ptr(), len(cv) - 1, sort=sort)
        assert e.value.code == -6 and e.value.needed == len(cv) and (d_v[len(cv) - 1:] == -1).all()
    assert gpu.locate_into(d_r.data_ptr(), 0, d_o.data_ptr(), 0, 0) == 0 and int(d_o[0]) == 0


@pytest.mark.parametrize("single", ["1", "0"], ids=["one-kernel-path", "general-pipeline"])
def test_locate_batches_of_one_value_ranges(engine, single, monkeypatch):
    """A batch in which every range is one path node with one value (config 3's 32-mers, the final ranges of long patterns)
    is answered by ONE kernel from the locate table (k_locate_single, GCSA2_LOCATE_SINGLE=0 switches it off): same CSR as the
    general pipeline and the oracle (gcsa.cpp:827-842, 880-896), through every entry point, sorted or not; a single misfit in
    the batch -- an empty range, a wider one, a node with several values -- sends it through the pipeline; too small a buffer
    is refused with the size needed."""
    import torch
    from oracle.oracle import OracleIndex
    monkeypatch.setenv("GCSA2_LOCATE_SINGLE", single)
    name, g, K = CASES[-1]
    ix = build(g, K, sample_period=8, branching=4)
    gpu, cpu = engine.GCSA(ix), OracleIndex(ix)
    nodes = np.arange(ix.n, dtype=np.uint64)
    every = np.stack([nodes, nodes], axis=1)
    eo, ev = cpu.locate_batch(every)
    one = nodes[np.diff(eo) == 1]                                  # the path nodes with exactly one value
    several = nodes[np.diff(eo) > 1]
    assert len(one) > 100 and len(several) > 0
    rng = SplitMix64(0x5151)
    pick = np.array([one[rng.below(len(one))] for _ in range(3000)], dtype=np.uint64)
    dev = torch.device("cuda", 0)
    batches = {"all one value": np.stack([pick, pick], axis=1),
               "one node with several values": np.concatenate([np.stack([pick, pick], axis=1), [[several[0], several[0]]]]).astype(np.uint64),
               "an empty range in front": np.concatenate([[[1, 0]], np.stack([pick, pick], axis=1)]).astype(np.uint64),
               "a wider range": np.concatenate([np.stack([pick[:50], pick[:50]], axis=1), [[0, 3]]]).astype(np.uint64)}
    for tag, ranges in batches.items():
        co, cv = cpu.locate_batch(ranges)
        go, gv = gpu.locate_batch(ranges)
        assert np.array_equal(go, co) and np.array_equal(gv, cv), tag
        d_r = torch.from_numpy(ranges.view(np.int64).copy()).to(dev)
        for sort in (True, False):
            if not sort:
                parts = [cpu.locate((int(a), int(b)), sort=False) for a, b in ranges]
                wo = np.concatenate([[0], np.cumsum([len(x) for x in parts])]).astype(np.uint64)
                wv = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)
            else:
                wo, wv = co, cv
            d_o = torch.full((len(ranges) + 1,), -1, dtype=torch.int64, device=dev)
            d_v = torch.full((len(wv) + 5 + 64,), -1, dtype=torch.int64, device=dev)
            total = gpu.locate_into(d_r.data_ptr(), len(ranges), d_o.data_ptr(), d_v.data_ptr(), len(wv) + 5, sort=sort)
            assert total == len(wv) and np.array_equal(d_o.cpu().numpy().view(np.uint64), wo), (tag, sort)
            assert np.array_equal(d_v.cpu().numpy().view(np.uint64)[:total], wv) and (d_v[len(wv) + 5:] == -1).all(), (tag, sort)
            with pytest.raises(engine.Gcsa2Error) as e:
                gpu.locate_into(d_r.data_ptr(), len(ranges), d_o.data_ptr(), d_v.data_ptr(), len(wv) - 1, sort=sort)
            assert e.value.code == -6 and e.value.needed == len(wv), (tag, sort)
    gpu.close()


@pytest.mark.parametrize("mailbox", ["1", "0"], ids=["resident-wavefront", "launch-per-call"])
def test_scalar_calls_through_the_resident_wavefront(engine, mailbox, monkeypatch):
    """One-query calls -- the reference's caller shape for this path is `range = index.LF(range, comp)` once per character
    (include/gcsa/gcsa.h:155-162), with count() and parent() (src/lcp.cpp:276-301) beside it -- are answered by a resident
    wavefront through a page-locked slot (kernels_mailbox.hpp; GCSA2_MAILBOX=0: a kernel launch per call, as before): the
    oracle's answers either way, for every comp incl. the steps that empty (edge-space pairs), across a pause longer than the
    park interval (the wavefront has left and is launched again), across a re-shaping of the image (it holds the old
    pointers), and the counters say which path ran."""
    import time
    from oracle.oracle import OracleIndex
    from workload import builder
    monkeypatch.setenv("GCSA2_MAILBOX", mailbox)
    monkeypatch.setenv("GCSA2_MAILBOX_PARK_US", "300")
    g = graphs.snp_graph(3000, 0xA1, 0xA2, snp_period=9, node_len=16)
    ix = builder.build(g, 16, sample_period=16, branching=4)
    gpu, lcp = engine.open_index(ix)
    cpu = OracleIndex(ix)
    rng = SplitMix64(0xA3)
    pats = random_patterns(g, 20, 0xA4, 60)
    checked = 0
    for round_ in range(3):
        for p in pats[round_ * 20:(round_ + 1) * 20]:
            r = (0, ix.n - 1)
            for ch in reversed(p):
                comp = int(ix.char2comp[ch])
                want = cpu.LF(r, comp)
                got = gpu.LF(r, comp)
                assert got == want, (p, r, comp)
                if want[0] > want[1] or want[1] >= ix.n:             # emptied: the edge-space pair came back as it is; climb
                    node = lcp.parent(r)
                    assert node == cpu.parent(r), (p, r)
                    r = (node[0], node[1])
                    continue
                r = got
                assert gpu.count(r) == cpu.count(r)
                checked += 1
            v = rng.below(ix.n)
            assert gpu.LF(v) == cpu.LF(v)
        if round_ == 0:
            time.sleep(0.05)                                   # far beyond the park interval: the next call launches the wavefront again
        if round_ == 1:
            gpu.set_tables(pair_blocks=0, kmer_k=2, locate_table=0)        # the image changes under the handle
    assert checked > 300
    calls, launches = gpu.mailbox_stats()
    if mailbox == "1":
        assert calls > 3 * checked // 2 and 3 <= launches < calls // 20, (calls, launches)
    else:
        assert (calls, launches) == (0, 0)
    gpu.close()


def test_concurrent_host_threads(engine):
    """One handle, several host threads (the reference's const query methods are called that way,
    src/algorithms.cpp:113-132, 409-417): every thread gets the oracle's answers."""
    import threading
    from oracle.oracle import OracleIndex
    from workload import builder
    g = graphs.snp_graph(4000, 0xC1, 0xC2, snp_period=10, node_len=16)
    ix = builder.build(g, 16, sample_period=16, branching=8)
    gpu, lcp = engine.open_index(ix)
    cpu = OracleIndex(ix)
    errors = []

    def worker(seed):
        try:
            pats = [truncate_at_sink(p) for p in random_patterns(g, 16, seed, 300)]
            data, off = concat_patterns(pats)
            want = cpu.find_batch(data, off)
            for _ in range(8):
                got = gpu.find_batch(data, off)
                assert np.array_equal(got, want)
                hit = got[got[:, 0] <= got[:, 1]]
                go, gv = gpu.locate_batch(hit)
                co, cv = cpu.locate_batch(hit)
                assert np.array_equal(go, co) and np.array_equal(gv, cv)
                assert np.array_equal(gpu.count_batch(hit), cpu.count_batch(hit))
                assert np.array_equal(lcp.parent_batch(hit), cpu.parent_batch(hit))
        except Exception as e:          # noqa: BLE001  (reported below, in the main thread)
            errors.append((seed, repr(e)))

    threads = [threading.Thread(target=worker, args=(0xD0 + i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:2]


def test_seed_table_sizes(engine, monkeypatch):
    """find() is independent of the k-mer seed table: none, the default size, and forced sizes up to
    patterns longer / shorter than k all equal the oracle (including edge-space empty ranges)."""
    from oracle.oracle import OracleIndex
    from workload import builder
    g = graphs.snp_graph(5000, 0xE1, 0xE2, snp_period=11, node_len=16)
    ix = builder.build(g, 16)
    cpu = OracleIndex(ix)
    pats = [truncate_at_sink(p)[: 1 + q % 16] for q, p in enumerate(random_patterns(g, 16, 0xE3, 1500))]
    rng = SplitMix64(0xE4)
    pats += [bytes(b"ACGTN"[rng.below(5)] for _ in range(1 + rng.below(14))) for _ in range(500)]
    data, off = concat_patterns(pats)
    want = cpu.find_batch(data, off)
    seen = set()
    for setting in (None, "0", "1", "5", "8", "9"):
        if setting is None:
            monkeypatch.delenv("GCSA2_KMER_TABLE", raising=False)
        else:
            monkeypatch.setenv("GCSA2_KMER_TABLE", setting)
        gpu = engine.GCSA(ix, with_samples=False, with_counters=False, with_lcp=False)
        seen.add(gpu.kmer_table_k())
        if setting is not None:
            assert gpu.kmer_table_k() == int(setting)
        assert np.array_equal(gpu.find_batch(data, off), want), setting
    assert len(seen) >= 5


@pytest.mark.parametrize("pairs", ["1", "0"], ids=["pair-blocks", "single-blocks"])
def test_wide_seed_entries(engine, monkeypatch, pairs):
    """A seed-table entry whose range has 2^24 - 1 or more path nodes is marked, not stored, and the pattern is searched from
    scratch, as gcsa.h:96-110 searches every pattern.  On a large index only the shortest k-mers have such ranges, so no run
    met the branch; GCSA2_SEED_WIDE (read at create time) lowers the threshold to 2 / 3 / 30 path nodes here: the table
    builder's restart from charRange (k_seed_level) and the kernels' from-scratch start (k_find2, k_match_stats2) all run,
    `wide_seeds` of the instrumented kernel says how often, and every result equals the oracle."""
    import torch
    from oracle.oracle import OracleIndex
    from workload import builder
    g = graphs.snp_graph(6000, 0xE5, 0xE6, snp_period=9, node_len=16)
    ix = builder.build(g, 16, sample_period=16, branching=4)
    cpu = OracleIndex(ix)
    pats = [truncate_at_sink(p)[: 1 + q % 24] for q, p in enumerate(random_patterns(g, 24, 0xE7, 2000))]
    rng = SplitMix64(0xE8)
    pats += [bytes(b"ACGTN"[rng.below(5)] for _ in range(1 + rng.below(14))) for _ in range(600)] + [b"", b"A", b"AC", b"ACG"]
    data, off = concat_patterns(pats)
    want = cpu.find_batch(data, off)
    cm, cr, cf = cpu.match_stats_batch(data, off, threads=2)
    dev = torch.device("cuda", 0)
    d_pat = torch.zeros(int(off[-1]) + 16, dtype=torch.uint8, device=dev)
    d_pat[: int(off[-1])] = torch.from_numpy(data[: int(off[-1])]).to(dev)
    d_off = torch.from_numpy(off.view(np.int64).copy()).to(dev)
    monkeypatch.setenv("GCSA2_PAIR_BLOCKS", pairs)
    hits = []
    for wide, k in (("2", "3"), ("3", "6"), ("30", "4"), ("2", "8"), (None, "3")):
        monkeypatch.setenv("GCSA2_KMER_TABLE", k)
        if wide is None:
            monkeypatch.delenv("GCSA2_SEED_WIDE", raising=False)
        else:
            monkeypatch.setenv("GCSA2_SEED_WIDE", wide)
        gpu, lcp = engine.open_index(ix, device=0)
        assert gpu.kmer_table_k() == int(k) and (gpu.pair_block_bytes() > 0) == (pairs == "1")
        assert np.array_equal(gpu.find_batch(data, off), want), (wide, k)
        d_out = torch.zeros((len(pats), 2), dtype=torch.int64, device=dev)
        d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
        gpu.find_stats_device(d_pat.data_ptr(), d_off.data_ptr(), len(pats), d_out.data_ptr(), d_stats.data_ptr(), 0)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want), (wide, k)
        hits.append(int(d_stats[6].item()))
        gm, gr, gf = gpu.match_stats_batch(data, off)
        assert np.array_equal(gm, cm) and np.array_equal(gr, cr) and np.array_equal(gf, cf), (wide, k)
        gpu.close()
    assert all(h > 0 for h in hits[:4]) and hits[4] == 0, hits       # the branch ran in every lowered setting, never at the default


@pytest.mark.parametrize("pairs", ["1", "0"], ids=["pair-blocks", "single-blocks"])
def test_find_packed_patterns(engine, monkeypatch, pairs):
    """gcsa2_find_batch_packed / gcsa2_find_packed_device: k-mer batches handed over as 2-bit codes (one length per batch, last
    character first) return exactly the ranges of the byte interface and of the oracle -- hits, misses at every depth (edge-space
    empty ranges, gcsa.h:160), lengths below, at and above the seed table's k, lengths that are not multiples of 32 and span
    several code words; small batches (one copy) and large ones (the chunked host pipeline, pageable and page-locked)."""
    import torch
    from oracle.oracle import OracleIndex
    from workload import builder
    from gcsa2_amd.hostview import pack_kmers
    g = graphs.snp_graph(20000, 0xF1, 0xF2, snp_period=9, node_len=16)
    ix = builder.build(g, 16)
    cpu = OracleIndex(ix)
    monkeypatch.setenv("GCSA2_PAIR_BLOCKS", pairs)
    rng = SplitMix64(0xF3)
    for kmer, m in (("0", 5), ("6", 5), ("6", 6), ("6", 19), (None, 32), ("8", 33), ("3", 64), ("7", 77)):
        if kmer is None:
            monkeypatch.delenv("GCSA2_KMER_TABLE", raising=False)
        else:
            monkeypatch.setenv("GCSA2_KMER_TABLE", kmer)
        gpu = engine.GCSA(ix, with_samples=False, with_counters=False, with_lcp=False)
        walks = [p for p in random_patterns(g, m, 0xF4 + m, 3000) if len(p) == m and all(c in b"ACGT" for c in p)]
        for p in list(walks[:1500]):                              # substitutions: misses at every depth
            k = rng.below(m)
            walks.append(p[:k] + bytes([b"ACGT"[rng.below(4)]]) + p[k + 1:])
        walks += [bytes(b"ACGT"[rng.below(4)] for _ in range(m)) for _ in range(500)]
        arr = np.frombuffer(b"".join(walks), dtype=np.uint8).reshape(len(walks), m)
        data, off = concat_patterns(walks)
        want = cpu.find_batch(data, off)
        codes = pack_kmers(arr, ix.char2comp)
        assert codes.shape == (len(walks), (m + 31) // 32)
        assert np.array_equal(gpu.find_batch(data, off), want), (kmer, m)
        assert np.array_equal(gpu.find_batch_packed(codes, m), want), (kmer, m)
        dev = torch.device("cuda", 0)
        d_codes = torch.from_numpy(codes.view(np.int64).copy()).to(dev)
        d_out = torch.zeros((len(walks), 2), dtype=torch.int64, device=dev)
        gpu.find_packed_device(d_codes.data_ptr(), m, len(walks), d_out.data_ptr(), 0)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want), (kmer, m)
        gpu.close()
    with pytest.raises(ValueError):
        pack_kmers(np.frombuffer(b"ACGN", dtype=np.uint8).reshape(1, 4), ix.char2comp)
    # the chunked pipeline: 2^18 patterns, pageable and page-locked
    monkeypatch.delenv("GCSA2_KMER_TABLE", raising=False)
    gpu = engine.GCSA(ix, with_samples=False, with_counters=False, with_lcp=False)
    m, nq = 32, (1 << 18) + 4321
    base = np.frombuffer(b"".join(p for p in random_patterns(g, m, 0xF9, 6000) if len(p) == m and all(c in b"ACGT" for c in p)), dtype=np.uint8).reshape(-1, m)
    arr = base[np.random.default_rng(0xFA).integers(0, base.shape[0], size=nq)].copy()
    flip = np.random.default_rng(0xFB).integers(0, 4, size=nq) == 0
    arr[flip, 7] = np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(0xFC).integers(0, 4, size=int(flip.sum()))]
    off = np.arange(nq + 1, dtype=np.uint64) * np.uint64(m)
    want = gpu.find_batch(arr.reshape(-1), off)
    sample = np.random.default_rng(0xFD).integers(0, nq, size=3000)
    assert np.array_equal(want[sample], cpu.find_batch(arr[sample].reshape(-1).copy(), np.arange(3001, dtype=np.uint64) * np.uint64(m)))
    codes = pack_kmers(arr, ix.char2comp)
    assert np.array_equal(gpu.find_batch_packed(codes, m), want)
    p_codes = torch.empty((nq, 1), dtype=torch.int64).pin_memory()
    p_out = torch.zeros((nq, 2), dtype=torch.int64).pin_memory()
    p_codes.numpy().view(np.uint64)[:] = codes
    gpu.find_batch_packed(p_codes.numpy().view(np.uint64), m, out=p_out.numpy().view(np.uint64))
    assert np.array_equal(p_out.numpy().view(np.uint64), want)


def test_deep_lcp_tree(engine):
    """A binary range-minimum tree (branching 2, the smallest the reference's constructor accepts) over 70 k values has 18
    levels: more than the 16 an earlier build admitted.  Every LCP operation equals the oracle."""
    from oracle.oracle import OracleIndex
    from workload import builder
    g = graphs.linear_graph(70000, 0x97, node_len=32)
    ix = builder.build(g, 16, sample_period=32, branching=2)
    assert len(ix.lcp_offsets) - 1 > 16
    gpu, lcp = engine.open_index(ix)
    cpu = OracleIndex(ix)
    assert lcp.levels() == len(ix.lcp_offsets) - 1 and lcp.branching() == 2
    rng = np.random.default_rng(0x98)
    a = rng.integers(0, ix.n, size=4000)
    arr = np.stack([a, np.minimum(a + rng.integers(0, 50, size=4000) ** 2, ix.n - 1)], axis=1).astype(np.uint64)
    assert np.array_equal(lcp.parent_batch(arr), cpu.parent_batch(arr))
    assert np.array_equal(lcp.depth_batch(arr), cpu.depth_batch(arr))
    for pos in [0, 1, ix.n - 1] + [int(x) for x in a[:300]]:
        assert lcp.psv(pos) == cpu.psv(pos) and lcp.nsv(pos) == cpu.nsv(pos), pos
        assert lcp.psev(pos) == cpu.psev(pos) and lcp.nsev(pos) == cpu.nsev(pos), pos
    for sp, ep in arr[:300]:
        assert lcp.rmq(int(sp), int(ep)) == cpu.rmq(int(sp), int(ep))


def test_fuzz_random_graphs(engine, monkeypatch):
    """150 seeded random graphs (bubbles, indels, cycles, Ns; orders 2..6): every query type equals the
    oracle, which the CPU suite pins against the definition-level brute force on graphs of this family."""
    from oracle.oracle import OracleIndex
    for seed in range(150):
        rng = SplitMix64(0xF00 + seed)
        n = 12 + rng.below(50)
        g = graphs.random_graph(n, 0xF100 + seed, p_branch=0.15 + 0.05 * (seed % 4), p_back=(0.08 if seed % 3 == 0 else 0.0),
                                p_n=0.05, alphabet=(2 if seed % 5 == 0 else 4))
        K = 2 + seed % 5
        ix = build(g, K, sample_period=2 + seed % 7, branching=2 + seed % 5)
        gpu, lcp = engine.open_index(ix)
        cpu = OracleIndex(ix)
        pats = [truncate_at_sink(p) for p in random_patterns(g, K, 0xF200 + seed, 120)]
        data, off = concat_patterns(pats)
        ranges = gpu.find_batch(data, off)
        assert np.array_equal(ranges, cpu.find_batch(data, off)), seed
        if seed % 3 == 0:                   # the same index with the opt-in tables switched the other way
            monkeypatch.setenv("GCSA2_JUMP_TABLE", "1")
            monkeypatch.setenv("GCSA2_KMER_TABLE", str(seed % 4))
            monkeypatch.setenv("GCSA2_LOCATE_TABLE", "0")
            other = engine.GCSA(ix)
            for name in ("GCSA2_JUMP_TABLE", "GCSA2_KMER_TABLE", "GCSA2_LOCATE_TABLE"):
                monkeypatch.delenv(name)
            assert np.array_equal(other.find_batch(data, off), ranges), seed
            oo, ov = other.locate_batch(ranges)
            go0, gv0 = gpu.locate_batch(ranges)
            assert np.array_equal(oo, go0) and np.array_equal(ov, gv0), seed
        extra = np.array(all_ranges(ix, 0xF300 + seed)[:200], dtype=np.uint64)
        for arr in (ranges, extra):
            assert np.array_equal(gpu.count_batch(arr), cpu.count_batch(arr)), seed
            go, gv = gpu.locate_batch(arr)
            co, cv = cpu.locate_batch(arr)
            assert np.array_equal(go, co) and np.array_equal(gv, cv), seed
            ok = arr[(arr[:, 0] <= arr[:, 1]) & (arr[:, 1] < ix.n)]
            assert np.array_equal(lcp.parent_batch(ok), cpu.parent_batch(ok)), seed
            assert np.array_equal(lcp.depth_batch(ok), cpu.depth_batch(ok)), seed
        for k in (1, K, K + 1):
            assert gpu.count_kmers(k, force=True) == cpu.count_kmers(k, force=True), (seed, k)
        m, r, f = gpu.match_stats_batch(data, off)
        cm, cr, cf = cpu.match_stats_batch(data, off)
        assert np.array_equal(m, cm) and np.array_equal(r, cr) and np.array_equal(f, cf), seed


def test_jump_table(engine, monkeypatch):
    """find() with the memoised unary LF chains (GCSA2_JUMP_TABLE=1) equals the oracle: hits, misses at
    every depth (edge-space empty ranges), patterns with Ns, lengths 1..3 x order; also through the
    instrumented kernel."""
    import torch
    from oracle.oracle import OracleIndex
    from workload import builder
    monkeypatch.setenv("GCSA2_JUMP_TABLE", "1")
    rng = SplitMix64(0x1A0)
    for case, g in enumerate((graphs.snp_graph(20000, 0x1A1, 0x1A2, snp_period=9, node_len=16),
                              graphs.linear_graph(5000, 0x1A3, node_len=8),
                              graphs.random_graph(60, 0x1A4, p_branch=0.2, p_back=0.05))):
        ix = builder.build(g, 16) if case < 2 else build(g, 4)
        monkeypatch.setenv("GCSA2_KMER_TABLE", "3" if case == 0 else "0")
        gpu = engine.GCSA(ix)
        assert gpu.jump_table_bytes() == 16 * ix.n
        cpu = OracleIndex(ix)
        pats = [truncate_at_sink(p) for p in random_patterns(g, 48, 0x1A5 + case, 3000)]
        for p in list(pats[:600]):                      # substitutions at random positions: misses mid-chain
            if len(p) > 2:
                k = rng.below(len(p))
                pats.append(p[:k] + bytes([b"ACGTN"[rng.below(5)]]) + p[k + 1:])
        data, off = concat_patterns(pats)
        want = cpu.find_batch(data, off)
        assert np.array_equal(gpu.find_batch(data, off), want), case
        dev = torch.device("cuda", 0)
        d_pat = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
        d_off = torch.from_numpy(off.view(np.int64).copy()).to(dev)
        d_out = torch.zeros((len(pats), 2), dtype=torch.int64, device=dev)
        d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
        gpu.find_stats_device(d_pat.data_ptr(), d_off.data_ptr(), len(pats), d_out.data_ptr(), d_stats.data_ptr(), 0)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want), case
        d_out.zero_()
        gpu.find_device_variant(4, d_pat.data_ptr(), d_off.data_ptr(), len(pats), d_out.data_ptr(), 0)     # length-bucketed launch
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want), case


def test_very_long_patterns(engine, monkeypatch):
    """Patterns far longer than the order and than any window / seed length (a whole 60 kbp backbone,
    prefixes and suffixes of it, one with a late mismatch): find() equals the oracle, with and without
    the jump table; the matching-statistics values saturate at 65535 only where the oracle's do."""
    from oracle.oracle import OracleIndex
    from workload import builder
    g = graphs.linear_graph(60000, 0x2A1, node_len=8)
    ix = builder.build(g, 32)
    cpu = OracleIndex(ix)
    text = bytes(b"$ACGTN#"[int(c)] for c in g.comp[1:-1])
    broken = bytearray(text[:50000]); broken[137] = ord("A") if broken[137] != ord("A") else ord("C")
    pats = [text, text[:-1], text[1:], text[12345:], text[:40001], bytes(broken), text[-70:], text[:1], b"N" + text[:100]]
    data, off = concat_patterns(pats)
    want = cpu.find_batch(data, off)
    assert want[0][0] <= want[0][1]
    for jump in ("0", "1"):
        monkeypatch.setenv("GCSA2_JUMP_TABLE", jump)
        gpu = engine.GCSA(ix)
        assert np.array_equal(gpu.find_batch(data, off), want), jump
    short = [text[:3000], bytes(broken[:3000]), text[20000:20100]]
    data, off = concat_patterns(short)
    m, r, f = gpu.match_stats_batch(data, off)
    cm, cr, cf = cpu.match_stats_batch(data, off)
    assert np.array_equal(m, cm) and np.array_equal(r, cr) and np.array_equal(f, cf)


def test_arbitrary_bytes(engine, monkeypatch):
    """Patterns over all 256 byte values (lower case, IUPAC codes, control and high bytes): char2comp maps
    them as the reference's Alphabet does (support.h:150-151) and the ranges equal the oracle's."""
    from oracle.oracle import OracleIndex
    from workload import builder
    g = graphs.snp_graph(3000, 0x2B1, 0x2B2, snp_period=10, node_len=16)
    ix = builder.build(g, 16)
    cpu = OracleIndex(ix)
    rng = SplitMix64(0x2B3)
    pats = [bytes(rng.below(256) for _ in range(1 + rng.below(20))) for _ in range(2000)]
    walks = [truncate_at_sink(p) for p in random_patterns(g, 16, 0x2B4, 400)]
    pats += [p.lower() for p in walks] + [p[:-1] + bytes([rng.below(256)]) for p in walks if len(p) > 1]
    data, off = concat_patterns(pats)
    want = cpu.find_batch(data, off)
    for jump in ("0", "1"):
        monkeypatch.setenv("GCSA2_JUMP_TABLE", jump)
        assert np.array_equal(engine.GCSA(ix).find_batch(data, off), want), jump


def test_pair_blocks(engine, monkeypatch):
    """find() through the two-characters-per-step blocks (default) equals find() with GCSA2_PAIR_BLOCKS=0 and
    the oracle: SNP bubbles (pairs whose first step takes a non-last out-edge are replayed), cycles, misses at
    every depth (edge-space empty ranges of gcsa.h:160), Ns between fast characters, odd and even lengths,
    with and without seed / jump tables, and through the instrumented and length-bucketed launches."""
    import torch
    from oracle.oracle import OracleIndex
    from workload import builder
    rng = SplitMix64(0x3C0)
    for case, g in enumerate((graphs.snp_graph(20000, 0x3C1, 0x3C2, snp_period=7, node_len=16),
                              graphs.linear_graph(3000, 0x3C3, node_len=8),
                              graphs.random_graph(80, 0x3C4, p_branch=0.25, p_back=0.08, p_n=0.05))):
        ix = builder.build(g, 16) if case < 2 else build(g, 5)
        cpu = OracleIndex(ix)
        pats = [truncate_at_sink(p) for p in random_patterns(g, 40, 0x3C5 + case, 3000)]
        pats += [p[: 1 + q % 9] for q, p in enumerate(pats[:400])]
        for p in list(pats[:800]):
            if len(p) > 2:
                k = rng.below(len(p))
                pats.append(p[:k] + bytes([b"ACGTN"[rng.below(5)]]) + p[k + 1:])
        data, off = concat_patterns(pats)
        want = cpu.find_batch(data, off)
        for kmer, jump in (("0", "0"), ("3", "0"), (None, "1")):
            monkeypatch.setenv("GCSA2_JUMP_TABLE", jump)
            if kmer is None:
                monkeypatch.delenv("GCSA2_KMER_TABLE", raising=False)
            else:
                monkeypatch.setenv("GCSA2_KMER_TABLE", kmer)
            monkeypatch.setenv("GCSA2_PAIR_BLOCKS", "0")
            plain = engine.GCSA(ix, with_samples=False, with_counters=False, with_lcp=False)
            monkeypatch.delenv("GCSA2_PAIR_BLOCKS")
            gpu = engine.GCSA(ix, with_samples=False, with_counters=False, with_lcp=False)
            assert plain.pair_block_bytes() == 0
            assert gpu.pair_block_bytes() == 16 * (ix.n // 192 + 1) * 128
            assert np.array_equal(plain.find_batch(data, off), want), (case, kmer, jump)
            assert np.array_equal(gpu.find_batch(data, off), want), (case, kmer, jump)
            dev = torch.device("cuda", 0)
            d_pat = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
            d_off = torch.from_numpy(off.view(np.int64).copy()).to(dev)
            d_out = torch.zeros((len(pats), 2), dtype=torch.int64, device=dev)
            stats = []
            for g_ in (plain, gpu):
                d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
                g_.find_stats_device(d_pat.data_ptr(), d_off.data_ptr(), len(pats), d_out.data_ptr(), d_stats.data_ptr(), 0)
                torch.cuda.synchronize()
                assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want), (case, kmer, jump)
                stats.append([int(x) for x in d_stats.cpu()])
            if kmer is not None:
                assert stats[0][1] == stats[1][1]                  # the same LF steps ...
            if jump == "0":
                assert stats[1][0] < 0.75 * stats[0][0]            # ... through far fewer blocks
            for variant in (4,):
                d_out.zero_()
                gpu.find_device_variant(variant, d_pat.data_ptr(), d_off.data_ptr(), len(pats), d_out.data_ptr(), 0)
                torch.cuda.synchronize()
                assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want), (case, kmer, jump, variant)
    monkeypatch.delenv("GCSA2_JUMP_TABLE")


def test_group_find_device_and_comm(engine):
    """The device-resident multi-GPU entry points.  On a 1-GPU box the replicas share device 0, so the
    group gathers with peer copies (RCCL rejects a duplicated device); the RCCL communicator itself is
    exercised with world size 1 (gather = the root's own device copy) and by the (sp, len) u32 wire format.
    world > 1 control flow is covered by the gloo test in test_host.py."""
    import torch
    from oracle.oracle import OracleIndex
    from gcsa2_amd.shard import shard_bounds, slice_batch
    name, g, K = CASES[-1]
    ix = build(g, K, sample_period=8, branching=4)
    cpu = OracleIndex(ix)
    pats = [truncate_at_sink(p) for p in random_patterns(g, K, 0x7A, 203)] + [b"", b"N", b"ACGTTTTTT"]
    data, off = concat_patterns(pats)
    want = cpu.find_batch(data, off)
    dev = torch.device("cuda", 0)
    for devices in ([0], [0, 0], [0] * 5):
        grp = engine.GCSAGroup(ix, devices)
        bounds = shard_bounds(len(pats), len(devices))
        keep, d_pat, d_off = [], [], []
        for b, e in bounds:
            sub, so = slice_batch(data, off, b, e)
            tp = torch.from_numpy(np.concatenate([sub, np.zeros(8, dtype=np.uint8)])).to(dev)
            to = torch.from_numpy(so.view(np.int64).copy()).to(dev)
            keep += [tp, to]; d_pat.append(tp.data_ptr()); d_off.append(to.data_ptr())
        d_out = torch.zeros((len(pats), 2), dtype=torch.int64, device=dev)
        for _ in range(2):
            d_out.zero_()
            grp.find_device(d_pat, d_off, [e - b for b, e in bounds], d_out.data_ptr())
            assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want), devices
        assert not grp.uses_rccl()
        grp.close()
    # RCCL communicator, world 1
    comm = engine.Comm(engine.Comm.unique_id(), 0, 1, 0)
    src = torch.from_numpy(want.view(np.int64).copy()).to(dev)
    dst = torch.zeros_like(src)
    comm.gather(src.data_ptr(), [src.numel() * 8], dst.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    comm.close()
    # wire format: every range find() returns survives (sp, ep) -> (sp, len) u32 -> (sp, ep), incl. wrapped empties
    extra = np.array([[0, 2**64 - 1], [5, 4], [0, ix.n - 1], [2**32 - 2, 2**32 - 3], [7, 2**32 - 2]], dtype=np.uint64)
    both = np.concatenate([want, extra])
    d_in = torch.from_numpy(both.view(np.int64).copy()).to(dev)
    d_packed = torch.zeros((both.shape[0], 2), dtype=torch.int32, device=dev)
    d_back = torch.zeros_like(d_in)
    st = torch.cuda.current_stream().cuda_stream
    engine.pack_ranges32_device(d_in.data_ptr(), both.shape[0], d_packed.data_ptr(), st)
    engine.unpack_ranges32_device(d_packed.data_ptr(), both.shape[0], d_back.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(d_in, d_back)
    # the 10-byte form for indexes below 2^40 (sp, length: 40 bits each)
    wide = np.array([[2**40 - 1, 2**40 - 2], [2**33 + 5, 2**39 + 77], [0, 2**40 - 2], [2**39, 2**39], [1, 0]], dtype=np.uint64)
    both = np.concatenate([both[: want.shape[0] + 3], wide])
    d_in = torch.from_numpy(both.view(np.int64).copy()).to(dev)
    d_packed = torch.zeros(both.shape[0] * 10 + 6, dtype=torch.uint8, device=dev)
    d_back = torch.zeros_like(d_in)
    engine.pack_ranges40_device(d_in.data_ptr(), both.shape[0], d_packed.data_ptr(), st)
    engine.unpack_ranges40_device(d_packed.data_ptr(), both.shape[0], d_back.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(d_in, d_back)
    # the six-byte form (round 6): sp in 40 bits, the length in one byte, the ranges of 255 and more path nodes in a list of
    # fixed capacity behind the shard's ranges.  Exact for every range -- wrapped empties, lengths 254 / 255 / 256 on either
    # side of the byte, lengths beyond 2^32 -- as long as the list holds the long ones; a list that is too short is REPORTED
    # (the count the unpack returns exceeds the capacity), and every range that did fit is still exact.
    edge = np.array([[9, 9 + 253], [9, 9 + 254], [9, 9 + 255], [2**40 - 300, 2**40 - 2], [1, 0], [0, 2**64 - 1]], dtype=np.uint64)
    both = np.concatenate([both, edge, both[::-1]])
    n = both.shape[0]
    lengths = both[:, 1] + np.uint64(1) - both[:, 0]
    long_ones = int((lengths >= 255).sum())
    assert long_ones >= 6 and int((lengths == 0).sum()) >= 4
    d_in = torch.from_numpy(both.view(np.int64).copy()).to(dev)
    d_count = torch.zeros(1, dtype=torch.int64, device=dev)
    for capacity in (long_ones, long_ones + 7, 2):
        size = engine.wire48_bytes(n, capacity)
        assert size == ((6 * n + 15) // 16) * 16 + 16 + 16 * capacity
        d_packed = torch.full((size + 16,), 0x5A, dtype=torch.uint8, device=dev)
        d_back = torch.full_like(d_in, -7)
        engine.pack_ranges48_device(d_in.data_ptr(), n, d_packed.data_ptr(), capacity, st)
        engine.unpack_ranges48_device(d_packed.data_ptr(), n, capacity, d_back.data_ptr(), d_count.data_ptr(), st)
        torch.cuda.synchronize()
        assert int(d_count.item()) == long_ones and (d_packed[size:] == 0x5A).all()
        got = d_back.cpu().numpy().view(np.uint64)
        if capacity >= long_ones:
            assert np.array_equal(got, both)
        else:                                  # reported, and what fits is exact: the short ranges, and `capacity` of the long ones
            short = lengths < 255
            assert np.array_equal(got[short], both[short]) and capacity <= int((got[~short] == both[~short]).all(axis=1).sum()) < long_ones
            assert np.array_equal(got[~short][:, 0], both[~short][:, 0])
    engine.pack_ranges48_device(d_in.data_ptr(), 0, d_packed.data_ptr(), 0, st)              # an empty shard: a count of zero, nothing else
    engine.unpack_ranges48_device(d_packed.data_ptr(), 0, 0, d_back.data_ptr(), d_count.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(d_count.item()) == 0


def test_sharded_match_stats_and_locate(engine):
    """BASELINE configs[4] sharded: matching statistics and locate() of a batch split contiguously over
    replicas, results gathered on the root in query order -- the CSR offsets rebased by the per-shard totals.  On a 1-GPU box
    the replicas of a group share device 0 (peer copies) and the communicator has world size 1; ragged shards, an empty
    shard, and ranges of every width.  Must equal the unsharded oracle (count == |locate| on the gathered result,
    benchmark/query_gcsa.cpp:171-179, is checked on find() ranges by bench.py's sharded config 5 and test_bench.py)."""
    import torch
    from oracle.oracle import OracleIndex
    from gcsa2_amd.shard import shard_bounds, slice_batch
    name, g, K = CASES[-1]
    ix = build(g, K, sample_period=8, branching=4)
    cpu = OracleIndex(ix)
    pats = [p for p in random_patterns(g, 3 * K, 0x7B, 301)] + [b"", b"N", b"ACGTTTTTT", b"A", b"C"]
    data, off = concat_patterns(pats)
    cm, cr, cf = cpu.match_stats_batch(data, off, threads=2)
    ranges = np.array([r for r in all_ranges(ix, 0x7C, 150) if r[0] <= r[1]], dtype=np.uint64)
    lo, lv = cpu.locate_batch(ranges)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    total = int(off[-1])

    def shards(bounds):
        keep, d_pat, d_off, nbytes = [], [], [], []
        for b, e in bounds:
            sub, so = slice_batch(data, off, b, e)
            tp = torch.from_numpy(np.concatenate([sub, np.zeros(16, dtype=np.uint8)])).to(dev)
            to = torch.from_numpy(so.view(np.int64).copy()).to(dev)
            keep += [tp, to]; d_pat.append(tp.data_ptr()); d_off.append(to.data_ptr()); nbytes.append(int(so[-1]))
        return keep, d_pat, d_off, nbytes

    for devices in ([0], [0, 0, 0], [0] * 4):
        G = len(devices)
        grp = engine.GCSAGroup(ix, devices)
        bounds = shard_bounds(len(pats), G)
        if G == 4:                                                # ragged: an empty shard in the middle
            bounds = [(0, 7), (7, 7), (7, 200), (200, len(pats))]
        keep, d_pat, d_off, nbytes = shards(bounds)
        d_ms = torch.full((total + 8,), -5, dtype=torch.int16, device=dev)
        d_rng = torch.zeros((len(pats), 2), dtype=torch.int64, device=dev)
        d_fb = torch.zeros(len(pats), dtype=torch.int64, device=dev)
        grp.match_stats_device(d_pat, d_off, [e - b for b, e in bounds], nbytes, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr())
        assert np.array_equal(d_ms[:total].cpu().numpy().view(np.uint16), cm), devices
        assert np.array_equal(d_rng.cpu().numpy().view(np.uint64), cr) and np.array_equal(d_fb.cpu().numpy().view(np.uint64), cf), devices
        rb = shard_bounds(ranges.shape[0], G) if G != 4 else [(0, 1), (1, 1), (1, 90), (90, ranges.shape[0])]
        parts = [torch.from_numpy(ranges[b:e].view(np.int64).copy()).to(dev) if e > b else torch.zeros((1, 2), dtype=torch.int64, device=dev) for b, e in rb]
        d_loff = torch.full((ranges.shape[0] + 1,), -1, dtype=torch.int64, device=dev)
        for sort in (True, False):
            job, d_val, nval = grp.locate_device([p_.data_ptr() for p_ in parts], [e - b for b, e in rb], d_loff.data_ptr(), sort=sort)
            if sort:
                want_o, want_v = lo, lv
            else:                                                 # path order, duplicates kept
                parts = [cpu.locate((int(a), int(b)), sort=False) for a, b in ranges]
                want_o = np.concatenate([[0], np.cumsum([len(p_) for p_ in parts])]).astype(np.uint64)
                want_v = np.concatenate(parts).astype(np.uint64) if parts else np.zeros(0, dtype=np.uint64)
            assert nval == int(want_o[-1]) and np.array_equal(d_loff.cpu().numpy().view(np.uint64), want_o), (devices, sort)
            assert np.array_equal(engine.fetch_job(job, nval), want_v), (devices, sort)
        grp.close()
    # one rank per GPU: the RCCL communicator with world size 1 (the gathers are the root's own device copies)
    gpu, _ = engine.open_index(ix, device=0)
    try:
        comm = engine.Comm(engine.Comm.unique_id(), 0, 1, 0)
    except engine.Gcsa2Error as e:
        assert e.code == -5
        return
    keep, d_pat, d_off, nbytes = shards([(0, len(pats))])
    d_ms = torch.full((total + 8,), -5, dtype=torch.int16, device=dev)
    d_rng = torch.zeros((len(pats), 2), dtype=torch.int64, device=dev)
    d_fb = torch.zeros(len(pats), dtype=torch.int64, device=dev)
    comm.match_stats(gpu, d_pat[0], d_off[0], [len(pats)], nbytes, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), 0, st)
    torch.cuda.synchronize()
    assert np.array_equal(d_ms[:total].cpu().numpy().view(np.uint16), cm)
    assert np.array_equal(d_rng.cpu().numpy().view(np.uint64), cr) and np.array_equal(d_fb.cpu().numpy().view(np.uint64), cf)
    d_r = torch.from_numpy(ranges.view(np.int64).copy()).to(dev)
    d_loff = torch.zeros(ranges.shape[0] + 1, dtype=torch.int64, device=dev)
    job, d_val, nval = comm.locate(gpu, d_r.data_ptr(), [ranges.shape[0]], d_loff.data_ptr(), 0, st)
    assert nval == int(lo[-1]) and np.array_equal(d_loff.cpu().numpy().view(np.uint64), lo)
    assert np.array_equal(engine.fetch_job(job, nval), lv)
    assert comm.rccl_ranks() == 1
    # a rank whose own part fails (here: an index without samples cannot locate) still goes through the collectives of the call --
    # its total travels as a failure mark, it sends nothing afterwards -- and returns ITS error instead of leaving a peer waiting;
    # the communicator stays usable
    bare = engine.GCSA(ix, device=0, with_samples=False, with_counters=False)
    with pytest.raises(engine.Gcsa2Error) as e:
        comm.locate(bare, d_r.data_ptr(), [ranges.shape[0]], d_loff.data_ptr(), 0, st)
    assert e.value.code == -5                        # GCSA2_ERR_MISSING_COMPONENT, from gcsa2_locate_device
    job, d_val, nval = comm.locate(gpu, d_r.data_ptr(), [ranges.shape[0]], d_loff.data_ptr(), 0, st)
    assert nval == int(lo[-1]) and np.array_equal(engine.fetch_job(job, nval), lv)
    bare.close()
    comm.close()


def test_locate_splits_large_batches(case, engine, monkeypatch):
    """A locate batch with 2^31 or more values before deduplication (the paper's 16-mer batch has 2.5 G) is cut into
    consecutive sub-batches whose CSRs are concatenated.  The limit is a tuning knob read at create time
    (GCSA2_LOCATE_SPLIT), so the splitting runs here on small batches: same offsets and values as the oracle, in both sort
    modes, through the host and the device entry points; only a single range above the limit is refused."""
    import torch
    name, g, K, ix, gpu, lcp, cpu = case
    ranges = np.array([r for r in all_ranges(ix, 0x7D, 120) if r[0] <= r[1]], dtype=np.uint64)
    widest = int((ranges[:, 1] - ranges[:, 0] + 1).max())
    want_o, want_v = cpu.locate_batch(ranges)
    raw = [cpu.locate((int(a), int(b)), sort=False) for a, b in ranges]
    limit = max(len(r) for r in raw) + 3                     # every range fits alone, few fit together
    monkeypatch.setenv("GCSA2_LOCATE_SPLIT", str(limit))
    small, _ = engine.open_index(ix, device=0)
    go, gv = small.locate_batch(ranges)
    assert np.array_equal(go, want_o) and np.array_equal(gv, want_v), name
    go, gv = small.locate_batch(ranges, sort=False)
    assert np.array_equal(go, np.concatenate([[0], np.cumsum([len(r) for r in raw])]).astype(np.uint64)), name
    assert np.array_equal(gv, np.concatenate(raw).astype(np.uint64)), name
    dev = torch.device("cuda", 0)
    d_r = torch.from_numpy(ranges.view(np.int64).copy()).to(dev)
    d_off = torch.zeros(ranges.shape[0] + 1, dtype=torch.int64, device=dev)
    d_val = torch.zeros(max(int(want_o[-1]), 1), dtype=torch.int64, device=dev)
    assert small.locate_into(d_r.data_ptr(), ranges.shape[0], d_off.data_ptr(), d_val.data_ptr(), d_val.shape[0], 0) == int(want_o[-1])
    torch.cuda.synchronize()
    assert np.array_equal(d_off.cpu().numpy().view(np.uint64), want_o) and np.array_equal(d_val[: int(want_o[-1])].cpu().numpy().view(np.uint64), want_v)
    small.close()
    # the same cutting by the NUMBER of ranges (a pass handles at most 2^30; GCSA2_LOCATE_SPLIT_QUERIES lowers that here): empty
    # ranges included, so that a sub-batch can be all empty
    monkeypatch.delenv("GCSA2_LOCATE_SPLIT")
    monkeypatch.setenv("GCSA2_LOCATE_SPLIT_QUERIES", "37")
    padded = np.concatenate([ranges[:50], np.array([(5, 4)] * 90, dtype=np.uint64), ranges[50:]])
    want_po, want_pv = cpu.locate_batch(padded)
    few, _ = engine.open_index(ix, device=0)
    go, gv = few.locate_batch(padded)
    assert np.array_equal(go, want_po) and np.array_equal(gv, want_pv), name
    d_r = torch.from_numpy(padded.view(np.int64).copy()).to(dev)
    d_off = torch.zeros(padded.shape[0] + 1, dtype=torch.int64, device=dev)
    assert few.locate_into(d_r.data_ptr(), padded.shape[0], d_off.data_ptr(), d_val.data_ptr(), d_val.shape[0], 0) == int(want_po[-1])
    assert np.array_equal(d_off.cpu().numpy().view(np.uint64), want_po) and np.array_equal(d_val[: int(want_po[-1])].cpu().numpy().view(np.uint64), want_pv)
    few.close()
    monkeypatch.delenv("GCSA2_LOCATE_SPLIT_QUERIES")
    if widest > 3:
        monkeypatch.setenv("GCSA2_LOCATE_SPLIT", "2")
        tiny, _ = engine.open_index(ix, device=0)
        with pytest.raises(engine.Gcsa2Error) as err:
            tiny.locate_batch(ranges)
        assert err.value.code == -6
        tiny.close()


@pytest.mark.parametrize("dedup_huge", [1, 0, "wide"])
def test_locate_segment_sizes(engine, monkeypatch, dedup_huge):
    """removeDuplicates at every segment size class: 1 value, 2..16 (registers, one lane), 17..1024 (one wavefront in
    LDS; round 6: in registers, incl. the three- and six-register networks for up to 192 and 384 values), 1025..8192 (one workgroup in LDS), more (duplicates removed through an LDS hash set, then the LDS sorts; the
    device-wide radix sort over (segment, value) keys when more than 8192 values are distinct -- the whole-index range here --
    or, with GCSA2_DEDUP_HUGE=0, always), mixed in one batch and in both sort modes; ranges of consecutive path nodes of a repetitive SNP graph (many
    duplicates per segment).  A second batch has no segment beyond 8192 values: the library sort is then not called at all."""
    from oracle.oracle import OracleIndex
    from workload import builder
    g = graphs.snp_graph(30000, 0x4D1, 0x4D2, snp_period=5, node_len=16)
    ix = builder.build(g, 16, sample_period=16)
    # ("wide": the filter's hash table with 64-bit words; an index whose values fit 32 bits gets 32-bit words otherwise)
    monkeypatch.setenv("GCSA2_DEDUP_HUGE", "1" if dedup_huge == "wide" else str(dedup_huge))
    monkeypatch.setenv("GCSA2_DEDUP_NARROW", "0" if dedup_huge == "wide" else "1")
    gpu, lcp = engine.open_index(ix)
    cpu = OracleIndex(ix)
    rng = SplitMix64(0x4D3)
    ranges = []
    for width in (1, 2, 3, 8, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 150, 191, 192, 193, 200, 257, 300, 383, 384, 385, 511, 512, 513, 900, 1023, 1024, 1025, 1500, 2047, 2048, 2049, 3000,
                  4096, 4097, 7000, 8191, 8192, 8193, 9000, 20000):
        for _ in range(6):
            a = rng.below(ix.n - width)
            ranges.append((a, a + width - 1))
    ranges += [(0, ix.n - 1), (5, 4), (ix.n, ix.n + 3)]
    order = list(range(len(ranges)))
    for k in range(len(order) - 1, 0, -1):
        j = rng.below(k + 1)
        order[k], order[j] = order[j], order[k]
    arr = np.array([ranges[k] for k in order], dtype=np.uint64)
    go, gv = gpu.locate_batch(arr)
    co, cv = cpu.locate_batch(arr, threads=8)
    assert np.array_equal(go, co) and np.array_equal(gv, cv)
    # (count() equals the number of located values only for ranges that are suffix-tree nodes, e.g. find() results --
    # these are arbitrary node ranges, so count() is compared with the oracle's instead)
    assert np.array_equal(gpu.count_batch(arr), cpu.count_batch(arr))
    go, gv = gpu.locate_batch(arr, sort=False)
    for q, r in enumerate(arr):
        want = cpu.locate((int(r[0]), int(r[1])), sort=False)
        assert np.array_equal(gv[int(go[q]):int(go[q + 1])], np.asarray(want, dtype=np.uint64)), r
    modest = arr[(arr[:, 1] - arr[:, 0] < 5000) | (arr[:, 0] > arr[:, 1])]
    go, gv = gpu.locate_batch(modest)
    co, cv = cpu.locate_batch(modest, threads=8)
    assert np.array_equal(go, co) and np.array_equal(gv, cv) and int(np.diff(co).max()) > 1024


@pytest.mark.parametrize("knobs", [{}, {"GCSA2_SPLIT_TARGET": "24"}, {"GCSA2_SPLIT_TARGET": "1500", "GCSA2_SPLIT_SKEW": "700"}, {"GCSA2_SPLIT_SKEW": "16"},
                                   {"GCSA2_LOCATE_SPLIT_SORT": "0"}, {"GCSA2_LOCATE_FUSED_COMPACT": "0"}, {"values": "across 2^32"}, {"values": "across 2^32", "GCSA2_SPLIT_TARGET": "24"},
                                   {"values": "across 2^32", "GCSA2_SPLIT_TARGET": "1500", "GCSA2_SPLIT_SKEW": "3000"},
                                   {"GCSA2_LOCATE_FUSE": "0"}, {"GCSA2_LOCATE_IN_PLACE": "0"}, {"GCSA2_LOCATE_FUSE_ABOVE": "600"},
                                   {"GCSA2_LOCATE_FUSE_ABOVE": "2", "GCSA2_SPLIT_TARGET": "24"}, {"values": "across 2^32", "GCSA2_LOCATE_FUSE_ABOVE": "600", "GCSA2_SPLIT_SKEW": "16"}],
                         ids=["split-with-listed-buckets", "split-with-runs", "split-with-listed-and-skewed-buckets", "split-with-skewed-buckets", "radix-sort",
                              "four-kernel-compaction",
                              "64-bit-keys", "64-bit-keys-runs", "64-bit-keys-large-buckets",
                              "table-pass-for-every-range", "sorts-in-scratch", "fused-from-600-nodes",
                              "fused-from-3-nodes-runs", "fused-64-bit-keys-skewed"])
def test_locate_many_large_distinct_segments(engine, knobs, monkeypatch):
    """Ranges of thousands of path nodes whose values are all DISTINCT (a linear text: one value per path node), as found
    16-mers of interspersed repeats have on the 2^30-base text of bench.py: dozens of segments beyond the 8192 distinct values
    the LDS hash set and sorts hold.  Round 5: one workgroup per such segment splits it into buckets of a few dozen values and
    sorts them in registers, a wavefront per bucket (k_over_split); a bucket of more than 64 values is listed for the
    workgroup sort, one that is still too large for that goes to the device-wide radix sort over (segment, value) keys, which
    sorted all of them in round 4 (GCSA2_LOCATE_SPLIT_SORT=0) -- the test knobs make buckets of ~24 (sorted in runs inside the
    split) / ~256 (the default: listed for the register sorts) / ~1500 values and call more than 700 / 16 values too large, so
    that every branch runs, with node_type values below 2^32 and on both sides of it (32- and 64-bit sort keys); through the job interface
    (the value buffer is made once the total is known) and into caller-owned buffers (no wait for the total; a buffer that is
    too small is refused with the size needed and nothing is written behind its end).  Round 6: these ranges are FUSED -- their
    values never pass through the table pass, the split reads the locate table (GCSA2_LOCATE_FUSE=0: as before; the threshold of
    8192 path nodes lowered so that ranges of 700 / 3 nodes take that path too) -- and the caller's buffer is the sorts' target
    (GCSA2_LOCATE_IN_PLACE=0: scratch + compaction)."""
    import torch
    from oracle.oracle import OracleIndex
    from workload import builder
    knobs = dict(knobs)
    across = knobs.pop("values", None)
    for key, value in knobs.items():
        monkeypatch.setenv(key, value)
    g = graphs.linear_graph(70000, 0x4E1, node_len=32)
    if across:
        # node_type values on both sides of 2^32: the register sorts take 32-bit keys only when a segment's values share their
        # upper halves (sort_segment_regs), the run sort of k_over_split when a run lies within 2^32 of its base
        g.value += np.uint64((1 << 32) - int(g.value.max()) // 2)
    ix = builder.build(g, 16, sample_period=32)
    gpu, lcp = engine.open_index(ix)
    cpu = OracleIndex(ix)
    rng = SplitMix64(0x4E2)
    ranges = []
    for width in (8193, 8500, 9000, 12000, 16384, 20000, 33000, 50000, 3, 1, 700, 4097, 5000):
        for _ in range(5):
            a = rng.below(ix.n - width)
            ranges.append((a, a + width - 1))
    ranges += [(0, ix.n - 1), (9, 8)]
    order = list(range(len(ranges)))
    for k in range(len(order) - 1, 0, -1):
        j = rng.below(k + 1)
        order[k], order[j] = order[j], order[k]
    arr = np.array([ranges[k] for k in order], dtype=np.uint64)
    co, cv = cpu.locate_batch(arr, threads=8)
    assert int(np.diff(co).max()) == ix.n and int((np.diff(co) > 8192).sum()) >= 40
    go, gv = gpu.locate_batch(arr)
    assert np.array_equal(go, co) and np.array_equal(gv, cv)
    dev = torch.device("cuda", 0)
    d_r = torch.from_numpy(arr.view(np.int64)).to(dev)
    d_o = torch.full((len(arr) + 1,), -1, dtype=torch.int64, device=dev)
    d_v = torch.full((len(cv) + 7,), -1, dtype=torch.int64, device=dev)
    for _ in range(3):                                         # the scratch pool and the result slots are reused
        total = gpu.locate_into(d_r.data_ptr(), len(arr), d_o.data_ptr(), d_v.data_ptr(), d_v.shape[0])
        assert total == len(cv) and np.array_equal(d_o.cpu().numpy().view(np.uint64), co)
        assert np.array_equal(d_v.cpu().numpy().view(np.uint64)[:total], cv) and (d_v[total:] == -1).all()       # (no duplicates: raw == distinct)
    d_v.fill_(-1)
    with pytest.raises(engine.Gcsa2Error) as e:
        gpu.locate_into(d_r.data_ptr(), len(arr), d_o.data_ptr(), d_v.data_ptr(), len(cv) // 2)
    assert e.value.code == -6 and e.value.needed == len(cv) and (d_v[len(cv) // 2:] == -1).all()
    gpu.trim()                                                 # gives the pools back; the next call builds them again
    go, gv = gpu.locate_batch(arr[:20])
    c2o, c2v = cpu.locate_batch(arr[:20], threads=8)
    assert np.array_equal(go, c2o) and np.array_equal(gv, c2v)


def test_index_from_device_resident_arrays(case, engine):
    """gcsa2_index_create builds the image on the device and reads the bulk arrays of the view from wherever they are: an
    index whose bit arrays, samples and LCP bytes already sit in HBM answers exactly like the one built from host arrays
    (and like the oracle), and both report the same image size."""
    import types
    import torch
    name, g, K, ix, gpu, lcp, cpu = case
    dev = torch.device("cuda", 0)

    def words(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64).copy()).to(dev)

    fields = dict(vars(ix))
    fields["bwt"] = [words(b) for b in ix.bwt]
    for key in ("edges", "sampled_paths", "stored_samples", "samples", "extra_filter", "extra_values", "redundant"):
        fields[key] = words(fields[key])
    fields["lcp_data"] = torch.from_numpy(np.ascontiguousarray(ix.lcp_data, dtype=np.uint8).copy()).to(dev)
    resident, rlcp = engine.open_index(types.SimpleNamespace(**fields), device=0)
    assert resident.device_bytes() == gpu.device_bytes() and resident.sampleCount() == gpu.sampleCount()
    pats = random_patterns(g, K, 0xD1CE, 300)
    cat, off = concat_patterns(pats)
    got = resident.find_batch(cat, off)
    assert np.array_equal(got, gpu.find_batch(cat, off)), name
    for p, r in zip(pats[:60], got):
        assert tuple(int(x) for x in r) == tuple(cpu.find(p)), (name, p)
    ranges = np.array([r for r in all_ranges(ix, 0x5EED, 60) if r[0] <= r[1]], dtype=np.uint64)
    for a, b in zip(resident.locate_batch(ranges), gpu.locate_batch(ranges)):
        assert np.array_equal(a, b), name
    assert np.array_equal(resident.count_batch(ranges), gpu.count_batch(ranges)), name
    assert np.array_equal(rlcp.parent_batch(ranges), lcp.parent_batch(ranges)), name
    resident.close()


def test_small_calls_copy_and_zero_copy(case, engine, monkeypatch):
    """Host-pointer calls that move less than 32 KB run zero-copy (their buffers are page-locked memory the kernel reads and
    writes); GCSA2_ZERO_COPY=0 sends them through the device arena like larger calls.  Same answers either way, and for a
    batch just above the limit."""
    name, g, K, ix, gpu, lcp, cpu = case
    monkeypatch.setenv("GCSA2_ZERO_COPY", "0")
    copied, clcp = engine.open_index(ix, device=0)
    pats = random_patterns(g, K, 0x2C0, 40)
    cat, off = concat_patterns(pats)
    want = gpu.find_batch(cat, off)
    assert np.array_equal(copied.find_batch(cat, off), want), name
    for p, r in zip(pats, want):
        assert tuple(int(x) for x in r) == tuple(cpu.find(p)), (name, p)
    ranges = np.array([r for r in all_ranges(ix, 0x2C1, 30) if r[0] <= r[1]], dtype=np.uint64)[:200]
    comps = (np.arange(ranges.shape[0]) % int(ix.sigma)).astype(np.uint8)
    assert np.array_equal(copied.lf_batch(ranges, comps), gpu.lf_batch(ranges, comps)), name
    assert np.array_equal(copied.count_batch(ranges), gpu.count_batch(ranges)), name
    assert np.array_equal(clcp.parent_batch(ranges), lcp.parent_batch(ranges)), name
    for a, b in zip(copied.locate_batch(ranges), gpu.locate_batch(ranges)):
        assert np.array_equal(a, b), name
    many = [pats[i % len(pats)] for i in range(3000)]                 # ~ 60 KB with offsets and ranges: beyond the zero-copy limit
    cat, off = concat_patterns(many)
    assert np.array_equal(gpu.find_batch(cat, off), np.concatenate([want] * 75)), name
    copied.close()
