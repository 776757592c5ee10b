#!/usr/bin/env python3
"""Matching statistics of a large batch from host memory (gcsa2_match_stats_batch): 1 M x 256-bp walks through the chr22-like
index, every second one with a substitution every 41 bp; with and without the pieced path (GCSA2_MS_PIECES, read at create)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from workload import graphs, builder, patterns
    from gcsa2_amd.binding import open_index
    g = graphs.snp_graph(1 << 22, 0x6C5A0010, 0x6C5A0011)
    ix = builder.build(g, 256, keep_table=False)
    nq, m = 1_000_000, 256
    pats = patterns.walk_patterns(g, nq, m, 0x6C5A0013)
    flat, off = patterns.as_batch(pats)
    flat = flat.copy()
    sub = np.arange(nq // 2) * 2                         # every second pattern: a substitution every 41 bp
    for p in range(20, m, 41):
        at = sub * m + p
        flat[at] = np.frombuffer(b"ACGT", dtype=np.uint8)[(np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), flat[at]) + 1) % 4]
    import torch
    p_flat = torch.empty(flat.shape[0], dtype=torch.uint8).pin_memory(); p_flat.numpy()[:] = flat
    p_off = torch.empty(off.shape[0], dtype=torch.int64).pin_memory(); p_off.numpy().view(np.uint64)[:] = off
    p_out = (torch.empty(nq * m + 8, dtype=torch.int16).pin_memory().numpy().view(np.uint16), torch.empty((nq, 2), dtype=torch.int64).pin_memory().numpy().view(np.uint64),
             torch.empty(nq, dtype=torch.int64).pin_memory().numpy().view(np.uint64))
    want = None
    for pieces, piece_mb in ((1, 16), (0, 16), (1, 32), (1, 64), (1, 16), (1, 128)):
        os.environ["GCSA2_MS_PIECES"] = str(pieces)
        os.environ["GCSA2_MS_PIECE_MB"] = str(piece_mb)
        gpu, lcp = open_index(ix)
        got = gpu.match_stats_batch(flat, off)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            got = gpu.match_stats_batch(flat, off, out=got)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        if want is None:
            want = tuple(a.copy() for a in got)
        same = all(np.array_equal(a, b) for a, b in zip(got, want))
        row = {"pieces": pieces, "piece_mb": piece_mb, "ms": round(best * 1e3, 2), "patterns_per_s": round(nq / best / 1e6, 1), "same": same}
        got = gpu.match_stats_batch(p_flat.numpy(), p_off.numpy().view(np.uint64), out=p_out)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            got = gpu.match_stats_batch(p_flat.numpy(), p_off.numpy().view(np.uint64), out=p_out)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row["page_locked_ms"] = round(best * 1e3, 2); row["page_locked_M_per_s"] = round(nq / best / 1e6, 1)
        row["page_locked_same"] = all(np.array_equal(a[: b.shape[0]], b) for a, b in zip(got, want))
        print(json.dumps(row), flush=True)
        gpu.close()


if __name__ == "__main__":
    main()
