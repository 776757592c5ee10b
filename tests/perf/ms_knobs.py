#!/usr/bin/env python3
"""Sweep of the matching-statistics tuning knobs (read at create) on config 5 batch, pangenome-sized index: the defaults
(cool-down 3, refill at 8 idle lanes, a grid of what the device holds) are at the optimum; final kernel of round 4 (profiles/r04_ms_knobs_final.jsonl):
141 M patterns/s; cool-down 0: 134, 8: 140, 24: 129; refill at 1: 106, 48: 120; twice the grid: 130."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from workload import dbg_torch
from gcsa2_amd.binding import GCSA
dev = torch.device("cuda", 0)
ix, dbg = dbg_torch.build_dbg(34, junctions=80, device=dev, with_lcp=True)
torch.cuda.empty_cache()
nq, m = 1_000_000, 256
pats, _, _ = dbg_torch.walk_patterns_device(dbg, 0, nq, m, 0x6C5A0050)
nxt = torch.zeros(256, dtype=torch.uint8, device=dev)
for a, b in zip(b"ACGT", b"CGTA"): nxt[a] = b
for col in range(37, m, 41): pats[1::2, col] = nxt[pats[1::2, col].to(torch.int64)]
d_pat = torch.zeros(nq * m + 8, dtype=torch.uint8, device=dev); d_pat[: nq * m] = pats.reshape(-1)
d_off = torch.arange(nq + 1, dtype=torch.int64, device=dev) * m
d_ms = torch.zeros(nq * m + 8, dtype=torch.int16, device=dev); d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev); d_fb = torch.zeros(nq, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream()
ref = None
for knob, values in (("GCSA2_COOL_DOWN", (None, 0, 2, 4, 8, 12, 16, 24)), ("GCSA2_MS_REFILL_AT", (1, 2, 4, 8, 16, 24, 32, 48)), ("GCSA2_MS_GRID", (2048, 3072, 4096))):
    for v in values:
        for k in ("GCSA2_COOL_DOWN", "GCSA2_MS_REFILL_AT", "GCSA2_MS_GRID"): os.environ.pop(k, None)
        if v is not None: os.environ[knob] = str(v)
        gpu = GCSA(ix, device=0, with_samples=False, with_counters=False, with_lcp=True)
        run = lambda: gpu.match_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), st.cuda_stream, variant=0, total_bytes=nq * m)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): run()
        e1.record(st); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        same = True
        if ref is None: ref = d_ms.clone()
        else: same = bool(torch.equal(ref, d_ms))
        print(json.dumps({"knob": knob, "value": v, "ms": round(ms, 3), "M_per_s": round(nq / ms / 1e3, 1), "same": same}), flush=True)
        gpu.close()
