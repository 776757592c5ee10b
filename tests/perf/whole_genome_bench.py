#!/usr/bin/env python3
"""Whole-human-footprint find() on ONE MI355X: a 4.29 G-path-node index (the size of the paper's
whole-genome indexes, paper.tex:378-380) built without suffix sorting from a degree-32 m-sequence
(workload/mseq_torch.py), 10 M 32-mers that are substrings of the text, every result checked against
its closed-form answer find(T[p..p+32)) = (rank[p], rank[p]).

    python tests/perf/whole_genome_bench.py [--degree 32] [--queries 10000000] [--steps 10]
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
    ap.add_argument("--degree", type=int, default=32)
    ap.add_argument("--queries", type=int, default=10_000_000)
    ap.add_argument("--pattern-len", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--oracle-sample", type=int, default=0, help="also check this many queries against the CPU oracle")
    ap.add_argument("--full", action="store_true", help="also build samples / counters / LCP in closed form and run locate + parent")
    ap.add_argument("--full-queries", type=int, default=2_000_000)
    args = ap.parse_args()
    import torch
    from workload import mseq_torch, patterns
    from gcsa2_amd.binding import GCSA

    def log(msg):
        print(f"[{time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    dev = torch.device("cuda", 0)
    t = time.time()
    ix, sym_t, rank = mseq_torch.build_mseq(args.degree, device=dev, verbose=log, full=args.full)
    log(f"index arrays: n = {ix.n} ({time.time() - t:.1f} s)")
    nq, m = args.queries, args.pattern_len
    pats, exp = mseq_torch.substring_patterns(sym_t, rank, nq, m, 0x6C5A0070)
    starts = (mseq_torch._lsr(mseq_torch.splitmix64_torch(0x6C5A0070, nq, dev), 11) % ix.n).cpu().numpy()
    del sym_t
    torch.cuda.empty_cache()
    t = time.time()
    gpu = GCSA(ix, device=0, with_samples=args.full, with_counters=args.full, with_lcp=args.full)
    log(f"device image: {gpu.device_bytes() / 1e9:.2f} GB, seed table k = {gpu.kmer_table_k()} ({time.time() - t:.1f} s)")
    flat, off = patterns.as_batch(pats)
    d_pat = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_out = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream()

    def run():
        gpu.find_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_out.data_ptr(), st.cuda_stream)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(args.steps):
        run()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    got = d_out.cpu().numpy().view(np.uint64)
    exact = bool(np.array_equal(got, exp))
    d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
    d_out2 = torch.zeros_like(d_out)
    gpu.find_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_out2.data_ptr(), d_stats.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    blocks, steps, lookups, jumps = (int(x) for x in d_stats.cpu())
    algo = blocks * gpu.find_block_bytes() + lookups * 8 + jumps * 16 + nq * (m + 16)
    res = {"workload": f"degree-{args.degree} m-sequence cyclic text: {ix.n} path nodes, {nq} x {m}-mer find(), substrings of the text",
           "device_image_GB": gpu.device_bytes() / 1e9, "find_bytes_GB": ix.sigma * (ix.n // 384 + 1) * 128 / 1e9,
           "kernel_ms": ms, "queries_per_s": nq / (ms * 1e-3), "blocks_per_query": blocks / nq, "lf_steps_per_query": steps / nq,
           "algorithmic_GBps": algo / (ms * 1e-3) / 1e9, "frac_of_8TBps": algo / (ms * 1e-3) / 8e12,
           "all_results_equal_closed_form": exact}
    if args.full:
        nf = min(nq, args.full_queries)
        sub = d_out[:nf].contiguous()
        d_loff = torch.zeros(nf + 1, dtype=torch.int64, device=dev)
        d_vals = torch.zeros(nf, dtype=torch.int64, device=dev)          # one value per hit in this index
        total = gpu.locate_into(sub.data_ptr(), nf, d_loff.data_ptr(), d_vals.data_ptr(), nf, st.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            gpu.locate_into(sub.data_ptr(), nf, d_loff.data_ptr(), d_vals.data_ptr(), nf, st.cuda_stream)
        torch.cuda.synchronize()
        t_loc = (time.perf_counter() - t0) / 3
        res["locate_table_GB"] = gpu.locate_table_bytes() / 1e9
        located = d_vals.cpu().numpy().view(np.uint64)
        res["locate_queries_per_s"] = nf / t_loc
        res["locate_equals_closed_form"] = bool(total == nf and np.array_equal(located, mseq_torch.node_values(starts[:nf])))
        d_nodes = torch.zeros((nf, 5), dtype=torch.int64, device=dev)
        gpu.parent_device(sub.data_ptr(), nf, d_nodes.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        e0.record(st)
        for _ in range(3):
            gpu.parent_device(sub.data_ptr(), nf, d_nodes.data_ptr(), st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        res["parent_queries_per_s"] = nf / (e0.elapsed_time(e1) / 3 * 1e-3)
        # closed form: the parent of the singleton with k-mer value v = rank + 1 is the set of values
        # sharing the first L base-4 digits, L = max(lcp[rank], lcp[rank + 1])
        k = args.degree // 2
        lcp = ix.lcp_data[: ix.n].astype(np.int64)
        r = exp[:nf, 0].astype(np.int64)
        right = np.where(r + 1 < ix.n, lcp[np.minimum(r + 1, ix.n - 1)], 0)
        L = np.maximum(lcp[r], right)
        shift = 2 * (k - L)
        lo_val = ((r + 1) >> shift) << shift
        hi_val = lo_val + (np.int64(1) << shift) - 1
        psp = np.maximum(lo_val, 1) - 1
        pep = np.minimum(hi_val, ix.n) - 1
        nodes = d_nodes.cpu().numpy()
        ok = np.array_equal(nodes[:, 0], psp) and np.array_equal(nodes[:, 1], pep) and np.array_equal(nodes[:, 4], L)
        res["parent_equals_closed_form"] = bool(ok)
    if args.oracle_sample > 0:
        from oracle.oracle import OracleIndex
        cpu = OracleIndex(ix, with_samples=False, with_counters=False, with_lcp=False)
        ns = min(nq, args.oracle_sample)
        want = cpu.find_batch(flat, off[:ns + 1], threads=64)
        res["oracle_sample_equal"] = bool(np.array_equal(got[:ns], want))
        res["oracle_queries_per_s_64_threads"] = ns / cpu.last_seconds
    print(json.dumps(res))


if __name__ == "__main__":
    main()
