#!/usr/bin/env python3
"""Round trip of ONE-query calls -- what a caller that steps a search one character at a time pays per step
(include/gcsa/gcsa.h:155-162 in a loop; vg's MEM finder) --, with the resident wavefront of kernels_mailbox.hpp
(GCSA2_MAILBOX=1, the default) and through a kernel launch per call (GCSA2_MAILBOX=0).  The C entry points are called through
ctypes with preallocated arguments, so that the figure is the library's, not the binding's.

    python tests/perf/scalar_latency.py [--log2-bases 22] [--calls 20000]
"""
import argparse
import ctypes as C
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
    ap.add_argument("--calls", type=int, default=20000)
    args = ap.parse_args()
    from workload import graphs, builder, patterns
    from gcsa2_amd import binding
    g = graphs.snp_graph(1 << args.log2_bases, 0x6C5A0010, 0x6C5A0011)
    ix = builder.build(g, 256, keep_table=False)
    pats = patterns.walk_patterns(g, 64, 24, 0x6C5A0012)
    flat, off = patterns.as_batch(pats)
    rows = []
    for mailbox in ("1", "0"):
        os.environ["GCSA2_MAILBOX"] = mailbox
        gpu, lcp = binding.open_index(ix)
        L, h = binding.load_library(), gpu._h
        ranges = gpu.find_batch(flat, off)
        rng = (C.c_uint64 * 2)(int(ranges[0][0]), int(ranges[0][1]))
        out = (C.c_uint64 * 2)()
        comp = (C.c_uint8 * 1)(1)
        cnt = (C.c_uint64 * 1)()
        node = (C.c_uint64 * 5)()
        one = (C.c_uint64 * 1)(int(ranges[0][0]))

        def timed(fn):
            for _ in range(200):
                fn()
            t0 = time.perf_counter()
            for _ in range(args.calls):
                fn()
            return (time.perf_counter() - t0) / args.calls * 1e6

        row = {"mailbox": mailbox == "1",
               "lf_us": timed(lambda: L.gcsa2_lf_batch(h, rng, comp, 1, out)),
               "count_us": timed(lambda: L.gcsa2_count_batch(h, rng, 1, cnt)),
               "parent_us": timed(lambda: L.gcsa2_parent_batch(h, rng, 1, node)),
               "lf_node_us": timed(lambda: L.gcsa2_lf_node_batch(h, one, 1, cnt))}
        # a search stepped one character at a time: LF per character, parent() when it empties (the MEM finder's loop)
        pat = bytes(flat[int(off[1]):int(off[2])])
        c2c = np.asarray(ix.char2comp)
        def walk():
            rng[0], rng[1] = 0, int(ix.n) - 1
            for ch in reversed(pat):
                comp[0] = int(c2c[ch])
                L.gcsa2_lf_batch(h, rng, comp, 1, out)
                rng[0], rng[1] = out[0], out[1]
        t0 = time.perf_counter()
        for _ in range(200):
            walk()
        row["per_character_loop_us_per_step"] = (time.perf_counter() - t0) / (200 * len(pat)) * 1e6
        # the pause between two bursts is longer than the park interval: the next call launches the wavefront again
        time.sleep(0.05)
        t0 = time.perf_counter()
        L.gcsa2_lf_batch(h, rng, comp, 1, out)
        row["first_call_after_a_pause_us"] = (time.perf_counter() - t0) * 1e6
        rows.append(row)
        print(json.dumps(row), flush=True)
        gpu.close()
    print("| path | LF | count | parent | LF(node) | per-character loop, per step | first call after a pause |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {'resident wavefront' if r['mailbox'] else 'launch per call'} | {r['lf_us']:.1f} us | {r['count_us']:.1f} us | {r['parent_us']:.1f} us | "
              f"{r['lf_node_us']:.1f} us | {r['per_character_loop_us_per_step']:.1f} us | {r['first_call_after_a_pause_us']:.0f} us |")


if __name__ == "__main__":
    main()
