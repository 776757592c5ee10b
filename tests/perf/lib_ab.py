#!/usr/bin/env python3
"""A/B of several builds of libgcsa2_hip.so on ONE set of index arrays and batches (pangenome-sized index): find() of 32-mers and
the matching statistics of config 5's batch, clean and half substituted.  Each build is bound through its own copy of the
binding module, created and closed in turn.

  python tests/perf/lib_ab.py gcsa2_amd/lib/libgcsa2_hip.so gcsa2_amd/lib/libgcsa2_hip_prev.so ..."""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def binding_for(lib_path, tag):
    os.environ["GCSA2_HIP_LIB"] = os.path.abspath(lib_path)
    spec = importlib.util.spec_from_file_location(f"gcsa2_amd.binding_{tag}", os.path.join(ROOT, "gcsa2_amd", "binding.py"),
                                                  submodule_search_locations=None)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "gcsa2_amd"
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--degree", type=int, default=34)
    ap.add_argument("--find-queries", type=int, default=100_000_000)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    from workload import dbg_torch
    dev = torch.device("cuda", 0)
    ix, dbg = dbg_torch.build_dbg(args.degree, junctions=80, device=dev, with_lcp=True)
    torch.cuda.empty_cache()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def flat(pats):
        buf = torch.zeros(pats.numel() + 8, dtype=torch.uint8, device=dev)
        buf[: pats.numel()] = pats.reshape(-1)
        return buf
    nf, mf = args.find_queries, 32
    pf, _, exp_f = dbg_torch.walk_patterns_device(dbg, 0, nf, mf, 0x6C5A0041)
    d_pf, d_of = flat(pf), torch.arange(nf + 1, dtype=torch.int64, device=dev) * mf
    del pf
    nq, m = 1_000_000, 256
    pc, _, exp_c = dbg_torch.walk_patterns_device(dbg, 0, nq, m, 0x6C5A0050)
    ps = pc.clone()
    nxt = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"CGTA"):
        nxt[a] = b
    for col in range(37, m, 41):
        ps[1::2, col] = nxt[ps[1::2, col].to(torch.int64)]
    d_pc, d_ps, d_om = flat(pc), flat(ps), torch.arange(nq + 1, dtype=torch.int64, device=dev) * m
    del pc, ps
    d_ms = torch.zeros(nq * m + 8, dtype=torch.int16, device=dev)
    d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    d_fb = torch.zeros(nq, dtype=torch.int64, device=dev)
    d_out = torch.zeros((nf, 2), dtype=torch.int64, device=dev)
    ref = None
    for tag, lib in enumerate(args.libs):
        b = binding_for(lib, tag)
        t = time.time()
        gpu = b.GCSA(ix, device=0, with_samples=False, with_counters=False, with_lcp=True)
        row = {"lib": os.path.basename(lib), "create_s": round(time.time() - t, 1)}
        row["find_ms"] = timed(lambda: gpu.find_device(d_pf.data_ptr(), d_of.data_ptr(), nf, d_out.data_ptr(), st.cuda_stream), args.reps)
        row["find_G_per_s"] = nf / row["find_ms"] / 1e6
        row["find_ok"] = bool(torch.equal(d_out[:, 0], exp_f) and torch.equal(d_out[:, 1], exp_f))
        sized = hasattr(gpu, "match_stats_profile_device")

        def ms(d_pat):
            if sized:
                gpu.match_stats_device(d_pat.data_ptr(), d_om.data_ptr(), nq, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), st.cuda_stream,
                                       total_bytes=nq * m)
            else:
                gpu.match_stats_device(d_pat.data_ptr(), d_om.data_ptr(), nq, d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), st.cuda_stream)
        try:
            row["ms_clean_ms"] = timed(lambda: ms(d_pc), args.reps)
            row["ms_clean_M_per_s"] = nq / row["ms_clean_ms"] / 1e3
            row["ms_clean_ok"] = bool(torch.equal(d_rng[:, 0], exp_c) and (d_fb == 0).all())
            row["ms_mixed_ms"] = timed(lambda: ms(d_ps), args.reps)
            row["ms_mixed_M_per_s"] = nq / row["ms_mixed_ms"] / 1e3
            got = (d_ms.clone(), d_rng.clone(), d_fb.clone())
            if ref is None:
                ref = got
            row["ms_mixed_same_as_first_lib"] = all(bool(torch.equal(x, y)) for x, y in zip(ref, got))
        except Exception as e:          # an older build without the sized entry point / other ABI
            row["ms_error"] = str(e)[:200]
        gpu.close()
        del gpu
        torch.cuda.empty_cache()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
