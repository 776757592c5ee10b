#!/usr/bin/env python3
"""Benchmark of the GCSA2 query hot path on MI355X: batched k-mer find().

One step = one pass of the hot path (`gcsa2_find_device`, kernel k_find) over one batch of
synthetic patterns that already sit in HBM; with N > 1 every rank holds a replica of the index,
searches its own shard (weak scaling) and the hit ranges are gathered on rank 0 with one RCCL
gather per step.  Prints ONE JSON line on rank 0 (contract: see the task statement / DESIGN.md).

Workload (default): "chr22-like" = seeded SNP-bubble graph, 2^25 backbone bases, one SNP per 32 bp,
order-256 maximally pruned de Bruijn graph, 10 M 32-mers per GPU drawn
as random walks through the graph (set S: full-depth matches).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["snp", "linear"], default="snp",
                    help="snp: chr22-like SNP-bubble graph (config 2, fits the Infinity Cache); "
                         "linear: footprint-scale linear graph = FM-index built on the GPU (HBM-bound)")
    ap.add_argument("--log2-bases", type=int, default=25, help="backbone length")
    ap.add_argument("--order", type=int, default=256)
    ap.add_argument("--queries", type=int, default=10_000_000, help="patterns per GPU per step")
    ap.add_argument("--pattern-len", type=int, default=32)
    ap.add_argument("--set", choices=["S", "U"], default="S")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--variant", type=int, default=2, help="find kernel generation (1 = k_find, 2 = k_find2)")
    ap.add_argument("--cache-dir", default=os.environ.get("GCSA2_CACHE", "/tmp/gcsa2_bench_cache"))
    return ap.parse_args()


LINEAR_SEED = 0x6C5A0040


def get_index_and_graph(args, rank, barrier):
    """Rank 0 builds the index once per node and caches it; the others load it."""
    from workload import graphs, builder, cache
    key = f"{args.workload}_{args.log2_bases}_{args.order}_v2"
    path = os.path.join(args.cache_dir, key + ".npz")
    t = time.time()
    graph = None
    if args.workload == "snp":
        graph = graphs.snp_graph(1 << args.log2_bases, 0x6C5A0010, 0x6C5A0011)
        log(f"graph: {graph.size} positions ({time.time() - t:.1f} s)")
    if rank == 0 and not os.path.exists(path):
        os.makedirs(args.cache_dir, exist_ok=True)
        t = time.time()
        if args.workload == "snp":
            ix = builder.build(graph, args.order, keep_table=False)
        else:
            from workload import linear_torch
            ix = linear_torch.build_linear(1 << args.log2_bases, LINEAR_SEED, order=args.order,
                                           with_lcp=False, with_samples=False, verbose=log)
            import torch
            torch.cuda.empty_cache()
        log(f"index built: n={ix.n} e={ix.e} samples={ix.sample_count} ({time.time() - t:.1f} s)")
        cache.save(path + ".tmp.npz", ix)
        os.replace(path + ".tmp.npz", path)
    else:
        ix = None
    barrier()
    if ix is None:
        ix = cache.load(path)
    return ix, graph


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    from workload import patterns
    from gcsa2_amd.binding import GCSA

    ix, graph = get_index_and_graph(args, rank, barrier)
    t = time.time()
    full = args.workload == "snp"
    gpu = GCSA(ix, device=local_rank, with_samples=full, with_counters=full, with_lcp=full)
    log(f"device image: {gpu.device_bytes() / 1e6:.1f} MB in HBM, block = {gpu.block_bits()} payload bits ({time.time() - t:.1f} s)")

    nq, m = args.queries, args.pattern_len
    t = time.time()
    seed = 0x6C5A0012 + 0x1000 * rank
    if args.set == "U":
        pats = patterns.uniform_patterns(nq, m, seed)
    elif args.workload == "snp":
        pats = patterns.walk_patterns(graph, nq, m, seed)
    else:
        from workload import linear_torch
        pats = linear_torch.substring_patterns_torch(1 << args.log2_bases, LINEAR_SEED, nq, m, seed, dev)
        torch.cuda.empty_cache()
    flat, offsets = patterns.as_batch(pats)
    log(f"patterns: {nq} x {m} set {args.set} ({time.time() - t:.1f} s)")
    d_pat = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_out = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    gathered = [torch.zeros_like(d_out) for _ in range(world)] if (world > 1 and rank == 0) else None
    stream = torch.cuda.current_stream()

    def step(record=None):
        if record is not None:
            record[0].record(stream)
        gpu.find_device_variant(args.variant, d_pat.data_ptr(), d_off.data_ptr(), nq, d_out.data_ptr(), stream.cuda_stream)
        if record is not None:
            record[1].record(stream)
        if world > 1:   # the single gather of hit ranges over xGMI (16 B per query)
            dist.gather(d_out, gathered, dst=0)

    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # algorithmic traffic of one launch (instrumented kernel, outside the timed region)
    d_stats = torch.zeros(2, dtype=torch.int64, device=dev)
    d_out2 = torch.zeros_like(d_out)
    gpu.find_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_out2.data_ptr(), d_stats.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(d_out, d_out2), "instrumented and timed kernels disagree"
    blocks, lf_steps = (int(x) for x in d_stats.cpu())
    algo_bytes = blocks * gpu.find_block_bytes() + nq * (m + 16)
    found = int(((d_out[:, 0] <= d_out[:, 1])).sum().item())

    result = None
    if rank == 0:
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        result = {
            "metric": "kmer_find_queries_per_sec", "value": world * nq * args.steps / elapsed, "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": (f"chr22-like SNP graph 2^{args.log2_bases} bases" if args.workload == "snp"
                                    else f"linear graph 2^{args.log2_bases} bases (FM-index shaped GCSA)")
                                   + f", order-{args.order} GCSA, {nq} x {m}-mer find() per GPU, pattern set {args.set}",
                       "path_nodes": int(ix.n), "edges": int(ix.e), "queries_per_gpu": nq,
                       "pattern_len": m, "pattern_set": args.set, "index_bytes_hbm": gpu.device_bytes(),
                       "found": found, "lf_steps_per_query": lf_steps / nq,
                       "blocks_per_query": blocks / nq, "block_bytes": gpu.find_block_bytes(), "parallelism": f"replicated index, query shards x{world}"},
            "roofline": {"bound": "hbm", "kernel": "k_find2" if args.variant == 2 else "k_find", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": kernel_ms},
        }
        if not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(args, ix, flat, offsets, d_out, m)
        print(json.dumps(result), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, ix, flat, offsets, d_out, m):
    """The oracle (CPU restatement of the reference path) timed on this host: a bounded sample of
    the same patterns, all cores with the verifyIndex-style static split, plus one thread."""
    from oracle.oracle import OracleIndex, max_threads
    cpu = OracleIndex(ix, with_samples=False, with_counters=False, with_lcp=False)
    cores = max_threads()
    nq = offsets.shape[0] - 1
    probe = min(nq, 20000)
    cpu.find_batch(flat, offsets[:probe + 1], threads=1)
    per_query = cpu.last_seconds / probe
    n1 = int(min(nq, max(probe, 0.25 * args.cpu_seconds / per_query)))
    r1 = cpu.find_batch(flat, offsets[:n1 + 1], threads=1)
    t1 = cpu.last_seconds
    nall = int(min(nq, max(n1, 0.75 * args.cpu_seconds * cores / per_query * 0.5)))
    rall = cpu.find_batch(flat, offsets[:nall + 1], threads=cores)
    tall = cpu.last_seconds
    got = d_out[:nall].cpu().numpy().view(np.uint64)
    parity = bool(np.array_equal(got, rall)) and bool(np.array_equal(got[:n1], r1))
    return {"value": nall / tall, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"first {nall} of the {nq} patterns, {m}-mers, OpenMP static split over {cores} threads "
                      f"({tall:.1f} s); single thread: first {n1} patterns ({t1:.1f} s)",
            "single_thread_value": n1 / t1, "single_thread_us_per_query": t1 / n1 * 1e6,
            "gpu_matches_cpu_on_sample": parity}


if __name__ == "__main__":
    main()
