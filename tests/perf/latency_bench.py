#!/usr/bin/env python3
"""Latency of the host-pointer entry points (`gcsa2_*_batch`: copy in, kernel, copy out, synchronise) by batch
size, on the chr22-like index: what a caller that hands over host buffers -- or calls the facade's scalar
find() / LF() once per query, as a vg-style loop does -- actually pays.

    python tests/perf/latency_bench.py [--log2-bases 22]
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
    ap.add_argument("--log2-bases", type=int, default=22)
    ap.add_argument("--sizes", default="1,64,4096,262144,10000000", help="queries per call, comma separated")
    args = ap.parse_args()
    from workload import graphs, builder, patterns
    from gcsa2_amd.binding import open_index
    g = graphs.snp_graph(1 << args.log2_bases, 0x6C5A0010, 0x6C5A0011)
    ix = builder.build(g, 256, keep_table=False)
    gpu, lcp = open_index(ix)
    rows = []
    for nq in [int(x) for x in args.sizes.split(",")]:
        pats = patterns.walk_patterns(g, nq, 32, 0x6C5A0012)
        flat, off = patterns.as_batch(pats)
        gpu.find_batch(flat, off)                                   # warm-up (arena growth)
        reps = max(3, min(2000, 2_000_000 // nq))
        t0 = time.perf_counter()
        for _ in range(reps):
            ranges = gpu.find_batch(flat, off)
        t_find = (time.perf_counter() - t0) / reps
        comps = np.full(nq, 1, dtype=np.uint8)
        gpu.lf_batch(ranges, comps)
        t0 = time.perf_counter()
        for _ in range(reps):
            gpu.lf_batch(ranges, comps)
        t_lf = (time.perf_counter() - t0) / reps
        gpu.count_batch(ranges)
        t0 = time.perf_counter()
        for _ in range(reps):
            gpu.count_batch(ranges)
        t_count = (time.perf_counter() - t0) / reps
        lcp.parent_batch(ranges)
        t0 = time.perf_counter()
        for _ in range(reps):
            lcp.parent_batch(ranges)
        t_parent = (time.perf_counter() - t0) / reps
        rows.append({"queries": nq, "find_us": t_find * 1e6, "find_queries_per_s": nq / t_find, "lf_us": t_lf * 1e6,
                     "count_us": t_count * 1e6, "parent_us": t_parent * 1e6})
        print(json.dumps(rows[-1]), flush=True)
    print("| queries per call | find_batch | queries/s | lf_batch | count_batch | parent_batch |")
    print("|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['queries']} | {r['find_us']:.1f} us | {r['find_queries_per_s']:.3g} | {r['lf_us']:.1f} us | {r['count_us']:.1f} us | {r['parent_us']:.1f} us |")


if __name__ == "__main__":
    main()
