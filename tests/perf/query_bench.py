#!/usr/bin/env python3
"""query_gcsa-style phase benchmark on the GPU (reference benchmark/query_gcsa.cpp:87-169):
find -> parent -> depth -> count -> locate over one pattern set, each phase timed on the device
(inputs and outputs resident in HBM) and, beside it, the CPU oracle on the same host.

    python tests/perf/query_bench.py [--config 1|2] [--queries N] [--pattern-len M] [--locate-queries N]

config 1: 1-Mbp linear graph, order 64, 16-mers (50 % substrings, 50 % uniform random)
config 2: chr22-like SNP graph 2^25 bases, order 256, 32-mers from walks (set S)
Prints the reference's "N queries in S seconds (X µs/query)" lines and one JSON object.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def line(name, n, seconds):
    print(f"{name + ':':<12}{n} queries in {seconds:.6f} seconds ({seconds / max(n, 1) * 1e6:.4f} µs/query)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--log2-bases", type=int, default=25)
    ap.add_argument("--queries", type=int, default=0)
    ap.add_argument("--pattern-len", type=int, default=0)
    ap.add_argument("--locate-queries", type=int, default=1_000_000)
    ap.add_argument("--cpu-queries", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()

    import torch
    from workload import graphs, builder, patterns
    from gcsa2_amd.binding import open_index
    from gcsa2_amd.hostview import STNODE_DTYPE
    from oracle.oracle import OracleIndex, max_threads

    t = time.time()
    if args.config == 1:
        g = graphs.linear_graph(1_000_000, 0x6C5A0001)
        ix = builder.build(g, 64)
        nq = args.queries or 100_000
        m = args.pattern_len or 16
        pats = np.concatenate([patterns.walk_patterns(g, nq // 2, m, 0x6C5A0002),
                               patterns.uniform_patterns(nq - nq // 2, m, 0x6C5A0003)])
        name = "config 1: 1-Mbp linear graph, order 64"
    else:
        g = graphs.snp_graph(1 << args.log2_bases, 0x6C5A0010, 0x6C5A0011)
        ix = builder.build(g, 256, keep_table=False)
        nq = args.queries or 10_000_000
        m = args.pattern_len or 32
        pats = patterns.walk_patterns(g, nq, m, 0x6C5A0012)
        name = f"config 2/3: chr22-like SNP graph 2^{args.log2_bases}, order 256"
    print(f"{name}: {ix.n} path nodes, {ix.e} edges, {ix.sample_count} samples; {nq} x {m}-mers ({time.time() - t:.1f} s)")
    flat, off = patterns.as_batch(pats)

    dev = torch.device("cuda", 0)
    gpu, lcp = open_index(ix)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    d_pat = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev)

    def timed(fn, reps=args.reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    res = {"config": name, "queries": nq, "pattern_len": m, "gpu": {}, "cpu": {}}
    t_find = timed(lambda: gpu.find_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_rng.data_ptr(), sp))
    line("find()", nq, t_find)
    hit = d_rng[(d_rng[:, 0] <= d_rng[:, 1])].contiguous()
    nh = int(hit.shape[0])
    print(f"Found {nh} patterns matching {int((hit[:, 1] - hit[:, 0] + 1).sum().item())} paths")
    d_nodes = torch.zeros((nh, 5), dtype=torch.int64, device=dev)
    t_parent = timed(lambda: gpu.parent_device(hit.data_ptr(), nh, d_nodes.data_ptr(), sp))
    line("parent()", nh, t_parent)
    d_cnt = torch.zeros(nh, dtype=torch.int64, device=dev)
    t_count = timed(lambda: gpu.count_device(hit.data_ptr(), nh, d_cnt.data_ptr(), sp))
    line("count()", nh, t_count)
    print(f"{int(d_cnt.sum().item())} occurrences")
    nl = min(nh, args.locate_queries)
    sub = hit[:nl].contiguous()

    def locate_once():
        job, d_o, d_v, total = gpu.locate_device(sub.data_ptr(), nl, sp)
        locate_once.total = total
        gpu.locate_discard(job)
    t0 = time.perf_counter()
    locate_once()
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        locate_once()
    torch.cuda.synchronize()
    t_locate = (time.perf_counter() - t0) / reps
    line("locate()", nl, t_locate)
    print(f"{locate_once.total} occurrences ({t_locate / max(locate_once.total, 1) * 1e6:.4f} µs/occurrence)")
    # the same query into caller-owned buffers (no result allocation inside the call)
    d_lo = torch.zeros(nl + 1, dtype=torch.int64, device=dev)
    d_lv = torch.zeros(max(int(locate_once.total), 1), dtype=torch.int64, device=dev)
    t_into = timed(lambda: gpu.locate_into(sub.data_ptr(), nl, d_lo.data_ptr(), d_lv.data_ptr(), d_lv.shape[0], sp))
    line("locate_into", nl, t_into)
    # ranges spanning several path nodes: duplicates have to be removed (segmented sort + compaction)
    nw = min(nl, 2_000_000)
    wide = sub[:nw].clone()
    wide[:, 1] = torch.clamp(wide[:, 1] + (torch.arange(nw, device=dev) % 16), max=int(ix.n) - 1)
    d_wo = torch.zeros(nw + 1, dtype=torch.int64, device=dev)
    try:
        gpu.locate_into(wide.data_ptr(), nw, d_wo.data_ptr(), d_lv.data_ptr(), 0, sp)
        need = 0
    except Exception as e:
        need = getattr(e, "needed", 0)
    d_wv = torch.zeros(max(need, 1), dtype=torch.int64, device=dev)
    t_wide = timed(lambda: gpu.locate_into(wide.data_ptr(), nw, d_wo.data_ptr(), d_wv.data_ptr(), d_wv.shape[0], sp))
    line("locate wide", nw, t_wide)
    print(f"{need} occurrences ({t_wide / max(need, 1) * 1e6:.4f} µs/occurrence)")
    cpu_w = OracleIndex(ix)
    ns = 20000
    wo, wv = cpu_w.locate_batch(wide[:ns].cpu().numpy().view(np.uint64), threads=max_threads())
    wide_ok = bool(np.array_equal(d_wo[:ns + 1].cpu().numpy().view(np.uint64), wo) and
                   np.array_equal(d_wv[:len(wv)].cpu().numpy().view(np.uint64), wv))
    print(f"wide ranges equal the oracle on the first {ns}: {wide_ok}")
    res["gpu"] = {"locate_wide_qps": nw / t_wide, "locate_wide_values": int(need), "locate_wide_parity": wide_ok, "find_qps": nq / t_find, "parent_qps": nh / t_parent, "count_qps": nh / t_count,
                  "locate_qps": nl / t_locate, "locate_values_per_s": locate_once.total / t_locate,
                  "locate_values": int(locate_once.total), "locate_into_qps": nl / t_into}

    # CPU oracle beside it (bounded sample, all threads)
    cores = max_threads()
    cpu = OracleIndex(ix)
    nc = min(nq, args.cpu_queries)
    r = cpu.find_batch(flat, off[:nc + 1], threads=cores); t_cf = cpu.last_seconds
    h = r[r[:, 0] <= r[:, 1]]
    cpu.parent_batch(h, threads=cores); t_cp = cpu.last_seconds
    cpu.count_batch(h, threads=cores); t_cc = cpu.last_seconds
    hl = h[: min(h.shape[0], 200_000)]
    o, v = cpu.locate_batch(hl, threads=cores); t_cl = cpu.last_seconds
    res["cpu"] = {"cores": cores, "find_qps": nc / t_cf, "parent_qps": h.shape[0] / t_cp, "count_qps": h.shape[0] / t_cc,
                  "locate_qps": hl.shape[0] / t_cl, "locate_values_per_s": int(o[-1]) / t_cl}
    # parity on the CPU sample
    got = d_rng[:nc].cpu().numpy().view(np.uint64)
    res["parity_find"] = bool(np.array_equal(got, r))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
