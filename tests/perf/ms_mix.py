#!/usr/bin/env python3
"""Does mixing clean and substituted patterns inside a wavefront cost the matching-statistics kernel anything beyond the sum of
its parts?  1 M x 256 bp on the pangenome-sized index: none substituted / all / every second one (the config-5 batch) / the
second half of the batch (the same patterns as "every second", but wavefronts see one kind only)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from workload import dbg_torch
    from gcsa2_amd.binding import GCSA
    degree = int(sys.argv[1]) if len(sys.argv) > 1 else 34
    dev = torch.device("cuda", 0)
    ix, dbg = dbg_torch.build_dbg(degree, junctions=80, device=dev, with_lcp=True)
    torch.cuda.empty_cache()
    gpu = GCSA(ix, device=0, with_samples=False, with_counters=False, with_lcp=True)
    nq, m = 1_000_000, 256
    base, _, _ = dbg_torch.walk_patterns_device(dbg, 0, nq, m, 0x6C5A0050)
    nxt = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"CGTA"):
        nxt[a] = b
    d_off = torch.arange(nq + 1, dtype=torch.int64, device=dev) * m
    d_ms = torch.zeros(nq * m + 8, dtype=torch.int16, device=dev)
    d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    d_fb = torch.zeros(nq, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream()
    for name, rows in (("none", None), ("all", slice(None)), ("every second", slice(1, None, 2)), ("second half", slice(nq // 2, None))):
        pats = base.clone()
        if rows is not None:
            for col in range(37, m, 41):
                pats[rows, col] = nxt[pats[rows, col].to(torch.int64)]
        d_pat = torch.zeros(nq * m + 8, dtype=torch.uint8, device=dev)
        d_pat[: nq * m] = pats.reshape(-1)
        for variant in (0, 2):
            def run():
                gpu.match_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), st.cuda_stream,
                                       variant=variant, total_bytes=nq * m)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5):
                run()
            e1.record(st)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(json.dumps({"substituted": name, "variant": variant, "ms": round(ms, 3), "M_patterns_per_s": round(nq / ms / 1e3, 1),
                              "parent_calls_per_pattern": round(float(d_fb.to(torch.float64).mean().item()), 2)}), flush=True)


if __name__ == "__main__":
    main()
