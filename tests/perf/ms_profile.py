#!/usr/bin/env python3
"""Where a round of k_match_stats2 spends its time: config 5's batch (1 M x 256-bp walks, every second one with a
substitution every 41 bp) on the pangenome-sized index, timed with the default kernel and once with the instrumented one
(gcsa2_match_stats_profile_device: shader-clock cycles per phase, event counts).  `--degree 28` for a quick look."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

PHASES = ["stores + loop head + window refill (under the requests in flight)", "step setup + issue of the next requests", "wait for the requests", "first evaluation",
          "second fetch + evaluation", "outcome", "parent() from the LCP window", "parent() tree walk"]
EVENTS = ["rounds (per wave)", "rounds with a second fetch", "lane steps", "pair attempts", "failed pair attempts", "parent() calls",
          "tree walks", "lane second fetches"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree", type=int, default=34)
    ap.add_argument("--junctions", type=int, default=80)
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--len", type=int, default=256)
    ap.add_argument("--substituted", type=float, default=0.5, help="fraction of the patterns that get a substitution every 41 bp")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="0: one lane per pattern; 5: persistent lanes")
    args = ap.parse_args()
    import torch
    from workload import dbg_torch
    from gcsa2_amd.binding import GCSA
    dev = torch.device("cuda", 0)
    t = time.time()
    ix, dbg = dbg_torch.build_dbg(args.degree, junctions=args.junctions, device=dev, with_lcp=True)
    torch.cuda.empty_cache()
    gpu = GCSA(ix, device=0, with_samples=False, with_counters=False, with_lcp=True)
    print(f"index: n = {ix.n}, e = {ix.e}, image {gpu.device_bytes() / 1e9:.1f} GB ({time.time() - t:.0f} s)", file=sys.stderr)
    nq, m = args.queries, args.len
    pats, start, exp = dbg_torch.walk_patterns_device(dbg, 0, nq, m, 0x6C5A0050)
    nxt = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"CGTA"):
        nxt[a] = b
    every = max(1, int(round(1 / args.substituted))) if args.substituted > 0 else 0
    if every:
        for col in range(37, m, 41):
            pats[every - 1::every, col] = nxt[pats[every - 1::every, col].to(torch.int64)]
    d_pat = torch.zeros(nq * m + 8, dtype=torch.uint8, device=dev)
    d_pat[: nq * m] = pats.reshape(-1)
    d_off = torch.arange(nq + 1, dtype=torch.int64, device=dev) * m
    d_ms = torch.zeros(nq * m + 8, dtype=torch.int16, device=dev)
    d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    d_fb = torch.zeros(nq, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def run():
        gpu.match_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), st.cuda_stream,
                               variant=args.variant, total_bytes=nq * m)
    run()
    torch.cuda.synchronize()
    e0.record(st)
    for _ in range(args.reps):
        run()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    out = {"variant": args.variant, "degree": args.degree, "path_nodes": int(ix.n), "edges": int(ix.e), "patterns": nq, "pattern_len": m,
           "substituted_fraction": (1 / every if every else 0), "ms": ms, "patterns_per_s": nq / (ms * 1e-3),
           "parent_calls_per_pattern": float(d_fb.to(torch.float64).mean().item())}
    if not args.no_profile:
        ref_ms, ref_rng, ref_fb = d_ms.clone(), d_rng.clone(), d_fb.clone()
        d_prof = torch.zeros(16, dtype=torch.int64, device=dev)
        gpu.match_stats_profile_device(d_pat.data_ptr(), d_off.data_ptr(), nq, nq * m, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(),
                                       d_prof.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(ref_ms, d_ms) and torch.equal(ref_rng, d_rng) and torch.equal(ref_fb, d_fb), "instrumented kernel differs"
        prof = [int(x) for x in d_prof.cpu()]
        total = sum(prof[:8])
        waves = (nq + 63) // 64
        out["cycles_per_wave_round"] = total / max(prof[8], 1)
        out["phases"] = {PHASES[k]: {"share": prof[k] / total, "cycles_per_round": prof[k] / max(prof[8], 1)} for k in range(8)}
        out["events"] = {EVENTS[k]: prof[8 + k] for k in range(8)}
        out["rounds_per_wave"] = prof[8] / waves
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
