#!/usr/bin/env python3
"""SURVEY.md 8(d) config 5 shape on one GPU: long patterns, the LF + parent interplay of vg's MEM
finder (fused kernel k_match_stats), then locate() on the final ranges.

    python tests/perf/mem_bench.py [--log2-bases 25] [--queries 1000000] [--pattern-len 256]

Half of the patterns are walks through the graph (full-depth matches), half carry a substitution
every ~40 bp, so their ranges empty mid-pattern and parent() is taken.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-bases", type=int, default=25)
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--pattern-len", type=int, default=256)
    ap.add_argument("--cpu-queries", type=int, default=50_000)
    ap.add_argument("--substituted", type=float, default=0.5, help="fraction of patterns carrying substitutions (0, 0.5 or 1)")
    ap.add_argument("--ragged", action="store_true", help="pattern lengths uniform in [32, pattern-len] instead of all equal")
    ap.add_argument("--knobs", default="", help="';'-separated sets of NAME=value,... kernel knobs (gcsa2_match_stats_device reads them "
                                                "per call); each set is timed on the same index and batch, one JSON line per set")
    ap.add_argument("--also-queries", type=int, default=0, help="with --knobs: time this (smaller) batch size too")
    args = ap.parse_args()
    import torch
    from workload import graphs, builder, patterns
    from gcsa2_amd.binding import open_index
    from oracle.oracle import OracleIndex, max_threads

    g = graphs.snp_graph(1 << args.log2_bases, 0x6C5A0010, 0x6C5A0011)
    ix = builder.build(g, 256, keep_table=False)
    nq, m = args.queries, args.pattern_len
    pats = patterns.walk_patterns(g, nq, m, 0x6C5A0050)
    sub = np.frombuffer(b"ACGT", dtype=np.uint8)
    rows = slice(1, None, 2) if args.substituted == 0.5 else (slice(0, 0) if args.substituted == 0 else slice(None))
    for col in range(37, m, 41):               # substitutions in every second pattern (default)
        pats[rows, col] = sub[(np.searchsorted(sub, pats[rows, col]) + 1) % 4]
    flat, off = patterns.as_batch(pats)
    if args.ragged:
        lengths = np.random.default_rng(0x6C5A0051).integers(32, m + 1, size=nq)
        keep = np.arange(m)[None, :] < lengths[:, None]
        flat = np.ascontiguousarray(pats[keep])
        off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    dev = torch.device("cuda", 0)
    gpu, lcp = open_index(ix)
    stream = torch.cuda.current_stream()
    d_pat = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_ms = torch.zeros(nq * m, dtype=torch.int16, device=dev)
    d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    d_fb = torch.zeros(nq, dtype=torch.int64, device=dev)

    def run():
        gpu.match_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), stream.cuda_stream)
    if args.knobs:
        cpu = OracleIndex(ix)
        nc = min(nq, args.cpu_queries)
        cm, cr, cf = cpu.match_stats_batch(flat, off[:nc + 1], threads=max_threads())
        for spec in [""] + args.knobs.split(";"):
            names = []
            for kv in filter(None, spec.split(",")):
                k, v = kv.split("=")
                os.environ[k] = v
                names.append(k)
            line = {"knobs": spec or "defaults"}
            for count in filter(None, [nq, args.also_queries]):
                def go():
                    gpu.match_stats_device(d_pat.data_ptr(), d_off.data_ptr(), count, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), stream.cuda_stream)
                d_ms.zero_(); go(); torch.cuda.synchronize()
                ok = bool(np.array_equal(d_ms[: int(off[nc])].cpu().numpy().view(np.uint16), cm)) and \
                    bool(np.array_equal(d_rng[:nc].cpu().numpy().view(np.uint64), cr)) and \
                    bool(np.array_equal(d_fb[:nc].cpu().numpy().view(np.uint64), cf))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(3):
                    go()
                e1.record(stream); torch.cuda.synchronize()
                line[f"M_patterns_per_s@{count}"] = round(count / (e0.elapsed_time(e1) * 1e-3 / 3) / 1e6, 1)
                line[f"parity@{count}"] = ok
            for k in names:
                del os.environ[k]
            print(json.dumps(line), flush=True)
        return
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    reps = 3
    for _ in range(reps):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    t_ms = e0.elapsed_time(e1) * 1e-3 / reps
    t0 = time.perf_counter()
    job, d_o, d_v, total = gpu.locate_device(d_rng.data_ptr(), nq, stream.cuda_stream)
    torch.cuda.synchronize()
    t_loc = time.perf_counter() - t0
    gpu.locate_discard(job)

    cpu = OracleIndex(ix)
    nc = min(nq, args.cpu_queries)
    cores = max_threads()
    cm, cr, cf = cpu.match_stats_batch(flat, off[:nc + 1], threads=cores)
    t_cpu = cpu.last_seconds
    parity = bool(np.array_equal(d_ms[: int(off[nc])].cpu().numpy().view(np.uint16), cm)) and \
        bool(np.array_equal(d_rng[:nc].cpu().numpy().view(np.uint64), cr))
    res = {"config": f"config 5 shape: chr22-like SNP graph 2^{args.log2_bases}, {nq} x {m} bp, {int(100 * args.substituted)} % with a substitution every 41 bp",
           "gpu_match_stats_patterns_per_s": nq / t_ms, "gpu_bases_per_s": nq * m / t_ms, "gpu_ms": t_ms * 1e3,
           "parent_calls_per_pattern": float(d_fb.double().mean().item()),
           "locate_final_ranges_s": t_loc, "located_values": int(total),
           "cpu_patterns_per_s": nc / t_cpu, "cpu_cores": cores, "gpu_matches_cpu_on_sample": parity}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
