#!/usr/bin/env python3
"""A/B of the matching-statistics kernel's knobs on config 5's batch (1 M x 256 bp, every second pattern with a substitution
every 41 bp; pangenome-sized index): dense statistics, break points, and the unmodified patterns alone.  Every configuration
must return the first one's bytes.  One JSON line per configuration.  (Round 5 compared k_match_stats2 with k_match_stats3
through GCSA2_MS_KERNEL here: profiles/r05_ms/; that kernel was retired in round 6.)
  python tests/perf/ms_ab.py [--degree 34] [--configs ';GCSA2_COOL_DOWN=6;GCSA2_MS_GRID=1024']"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from workload import dbg_torch
from gcsa2_amd.binding import GCSA

ap = argparse.ArgumentParser()
ap.add_argument("--degree", type=int, default=34)
ap.add_argument("--patterns", type=int, default=1_000_000)
ap.add_argument("--configs", default=";GCSA2_COOL_DOWN=6;GCSA2_MS_GRID=1024")
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
ix, dbg = dbg_torch.build_dbg(args.degree, junctions=80, device=dev, with_lcp=True)
torch.cuda.empty_cache()
nq, m = args.patterns, 256
pats, _, _ = dbg_torch.walk_patterns_device(dbg, 0, nq, m, 0x6C5A0050)
nxt = torch.zeros(256, dtype=torch.uint8, device=dev)
for a, b in zip(b"ACGT", b"CGTA"): nxt[a] = b
for col in range(37, m, 41): pats[1::2, col] = nxt[pats[1::2, col].to(torch.int64)]
d_pat = torch.zeros(nq * m + 8, dtype=torch.uint8, device=dev); d_pat[: nq * m] = pats.reshape(-1)
nc = nq // 2
d_clean = torch.zeros(nc * m + 8, dtype=torch.uint8, device=dev); d_clean[: nc * m] = pats[0::2].reshape(-1)
del pats
d_off = torch.arange(nq + 1, dtype=torch.int64, device=dev) * m
d_ms = torch.zeros(nq * m + 8, dtype=torch.int16, device=dev); d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev); d_fb = torch.zeros(nq, dtype=torch.int64, device=dev)
cap = 40 * nq
d_boff = torch.zeros(nq + 1, dtype=torch.int64, device=dev); d_brk = torch.zeros((cap, 4), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0.record(st)
    for _ in range(args.reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps


ref = None
knobs = ("GCSA2_COOL_DOWN", "GCSA2_MS_REFILL_AT", "GCSA2_MS_GRID")
for config in args.configs.split(";"):
    for k in knobs: os.environ.pop(k, None)
    for kv in filter(None, config.split(",")):
        k, v = kv.split("="); os.environ[k] = v
    gpu = GCSA(ix, device=0, with_samples=False, with_counters=False, with_lcp=True)
    out = {"config": config}
    t = timed(lambda: gpu.match_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), st.cuda_stream, total_bytes=nq * m))
    out["dense_ms"], out["dense_M_per_s"] = round(t, 3), round(nq / t / 1e3, 1)
    dense = (d_ms.clone(), d_rng.clone(), d_fb.clone())
    total = [0]
    def brk(min_length=0):
        total[0] = gpu.match_breaks_device(d_pat.data_ptr(), d_off.data_ptr(), nq, nq * m, d_boff.data_ptr(), d_brk.data_ptr(), cap, d_rng.data_ptr(), d_fb.data_ptr(), st.cuda_stream, min_length=min_length)
    t = timed(brk)
    out["breaks_ms"], out["breaks_M_per_s"], out["records"] = round(t, 3), round(nq / t / 1e3, 1), total[0]
    breaks = (d_boff.clone(), d_brk[: total[0]].clone(), d_rng.clone(), d_fb.clone())
    t = timed(lambda: brk(20))
    out["breaks20_ms"], out["breaks20_M_per_s"] = round(t, 3), round(nq / t / 1e3, 1)
    t = timed(lambda: gpu.match_stats_device(d_clean.data_ptr(), d_off.data_ptr(), nc, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), st.cuda_stream, total_bytes=nc * m))
    out["clean_ms"], out["clean_M_per_s"] = round(t, 3), round(nc / t / 1e3, 1)
    got = dense + breaks
    if ref is None: ref = got
    out["same_as_first"] = all(bool(torch.equal(a, b)) for a, b in zip(ref, got))
    out["final_equal_dense"] = bool(torch.equal(dense[1], breaks[2])) and bool(torch.equal(dense[2], breaks[3]))
    out["parent_calls_per_pattern"] = float(dense[2].to(torch.float64).mean().item())
    print(json.dumps(out), flush=True)
    gpu.close()
    del dense, breaks, got
