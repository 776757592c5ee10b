#!/usr/bin/env python3
"""locate() by pattern length on the chr22-like index: short patterns have wide ranges (thousands of path nodes and
values per query, paper.tex:403-418: k = 16 on the real chr22-scale indexes), long ones a value or two.

    python tests/perf/locate_bench.py [--log2-bases 25] [--lengths 8,10,12,16,32]
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
    ap.add_argument("--lengths", default="8,10,12,16,32")
    ap.add_argument("--cache-dir", default=os.environ.get("GCSA2_CACHE", "/tmp/gcsa2_bench_cache"))
    args = ap.parse_args()
    import torch
    from workload import graphs, builder, patterns, cache
    from gcsa2_amd.binding import GCSA
    g = graphs.snp_graph(1 << args.log2_bases, 0x6C5A0010, 0x6C5A0011)
    path = os.path.join(args.cache_dir, f"snp_{args.log2_bases}_256_v2.npz")
    if os.path.exists(path):
        ix = cache.load(path)
    else:
        ix = builder.build(g, 256, keep_table=False)
        os.makedirs(args.cache_dir, exist_ok=True)
        cache.save(path, ix)
    gpu = GCSA(ix)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    print("| pattern length | queries | values per query | find | locate | located values/s | count == located |")
    print("|---|---|---|---|---|---|---|")
    for m in (int(x) for x in args.lengths.split(",")):
        expected = max(1.0, ix.n / 4.0 ** m)
        nq = int(max(2000, min(2_000_000, 60_000_000 / expected)))
        pats = patterns.walk_patterns(g, nq, m, 0x6C5A0060 + m)
        flat, off = patterns.as_batch(pats)
        d_pat = torch.from_numpy(np.concatenate([flat, np.zeros(8, dtype=np.uint8)])).to(dev)
        d_off = torch.from_numpy(off.view(np.int64)).to(dev)
        d_rng = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
        gpu.find_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_rng.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            gpu.find_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_rng.data_ptr(), st)
        torch.cuda.synchronize()
        t_find = (time.perf_counter() - t0) / 3
        d_cnt = torch.zeros(nq, dtype=torch.int64, device=dev)
        gpu.count_device(d_rng.data_ptr(), nq, d_cnt.data_ptr(), st)
        torch.cuda.synchronize()
        total = int(d_cnt.sum().item())
        d_loff = torch.zeros(nq + 1, dtype=torch.int64, device=dev)
        d_val = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
        got = gpu.locate_into(d_rng.data_ptr(), nq, d_loff.data_ptr(), d_val.data_ptr(), d_val.shape[0], st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            gpu.locate_into(d_rng.data_ptr(), nq, d_loff.data_ptr(), d_val.data_ptr(), d_val.shape[0], st)
        torch.cuda.synchronize()
        t_loc = (time.perf_counter() - t0) / 3
        ok = bool(got == total and torch.equal(d_loff[1:] - d_loff[:-1], d_cnt))
        print(f"| {m} | {nq} | {total / nq:.1f} | {t_find * 1e3:.2f} ms | {t_loc * 1e3:.2f} ms | {total / t_loc:.3g} | {ok} |", flush=True)


if __name__ == "__main__":
    main()
