#!/usr/bin/env python3
"""Benchmark of the GCSA2 query hot path on MI355X: batched k-mer find().

One step = one pass of the hot path (`gcsa2_find_device`, kernel k_find2) over one batch of
synthetic patterns that already sit in HBM; with N > 1 every rank holds a replica of the index,
searches its own shard (weak scaling) and the hit ranges are gathered on rank 0 with one RCCL
gather per step.  Prints ONE JSON line on rank 0 (contract: task statement / DESIGN.md section 5).

Primary workload (BASELINE.json configs[1]): "chr22-like" seeded
SNP-bubble graph, 2^25 backbone bases, one SNP per 32 bp, order-256 maximally pruned de Bruijn graph
(51 M path nodes), 10 M 32-mers per GPU drawn as random walks through the graph (set S: full-depth
matches).  Its fused blocks (102 MB) fit the 256 MiB Infinity Cache, so at N = 1 the run also
measures an HBM-resident index (linear graph, 2^30 bases, 2.1 GB of fused blocks) and reports it
under "hbm_resident" -- that is where the HBM roofline fraction is meaningful.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
REQUEST_CEILING_GPS = 58.0   # dependent random 128-byte fetches/s measured by gcsa2_amd/lib/gather_bench (profiles/r01_gather_bench.md)
LINEAR_SEED = 0x6C5A0040


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["snp", "linear"], default="snp",
                    help="snp: chr22-like SNP-bubble graph (config 2, Infinity-Cache resident); "
                         "linear: footprint-scale linear graph = FM-index built on the GPU (HBM resident)")
    ap.add_argument("--log2-bases", type=int, default=0, help="backbone length (default 25 for snp, 30 for linear)")
    ap.add_argument("--order", type=int, default=256)
    ap.add_argument("--queries", type=int, default=10_000_000, help="patterns per GPU per step")
    ap.add_argument("--pattern-len", type=int, default=32)
    ap.add_argument("--set", choices=["S", "U"], default="S")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-hbm-resident", action="store_true", help="skip the secondary HBM-resident measurement")
    ap.add_argument("--no-jump-table", action="store_true", help="skip the secondary measurement with GCSA2_JUMP_TABLE=1")
    ap.add_argument("--variant", type=int, default=2, help="find kernel generation (1 = k_find, 2 = k_find2)")
    ap.add_argument("--cache-dir", default=os.environ.get("GCSA2_CACHE", "/tmp/gcsa2_bench_cache"))
    return ap.parse_args()


class Dist:
    """torch.distributed plumbing; a no-op when not launched by torch.distributed.run."""

    def __init__(self, dev):
        import torch.distributed as dist
        self.dist = dist
        self.active = "RANK" in os.environ and "WORLD_SIZE" in os.environ
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # GCSA2_BENCH_BACKEND=gloo is a control-flow check for boxes with fewer GPUs than ranks (the gather
        # then goes through host memory and ranks may share a device); the measured configuration is nccl.
        self.backend = os.environ.get("GCSA2_BENCH_BACKEND", "nccl")
        if self.active:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev)
            else:
                dist.init_process_group(backend=self.backend)

    def barrier(self):
        if self.active:
            self.dist.barrier()

    def gather(self, tensor, parts):
        """Asynchronous gather on rank 0; returns the work handle (None when not distributed)."""
        if self.active and self.backend != "nccl":
            host = [p.cpu() for p in parts] if self.rank == 0 else None
            self.dist.gather(tensor.cpu(), host, dst=0)
            if self.rank == 0:
                for p, h in zip(parts, host):
                    p.copy_(h)
            return None
        if self.active:
            return self.dist.gather(tensor, parts if self.rank == 0 else None, dst=0, async_op=True)
        return None

    def max(self, value, dev):
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=dev if self.backend == "nccl" else "cpu")
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.active:
            self.dist.destroy_process_group()


def build_snp_index(args, log2_bases, rank, barrier):
    """Rank 0 builds the index once per node and caches it; the others load it."""
    from workload import graphs, builder, cache
    path = os.path.join(args.cache_dir, f"snp_{log2_bases}_{args.order}_v2.npz")
    t = time.time()
    graph = graphs.snp_graph(1 << log2_bases, 0x6C5A0010, 0x6C5A0011)
    log(f"graph: {graph.size} positions ({time.time() - t:.1f} s)")
    ix = None
    if rank == 0 and not os.path.exists(path):
        os.makedirs(args.cache_dir, exist_ok=True)
        t = time.time()
        ix = builder.build(graph, args.order, keep_table=False)
        log(f"index built: n={ix.n} e={ix.e} samples={ix.sample_count} ({time.time() - t:.1f} s)")
        cache.save(path + ".tmp.npz", ix)
        os.replace(path + ".tmp.npz", path)
    barrier()
    if ix is None:
        ix = cache.load(path)
    return ix, graph


def build_linear_index(args, log2_bases):
    """Every rank builds its own replica on its own GPU (a few seconds, deterministic)."""
    import torch
    from workload import linear_torch
    t = time.time()
    ix = linear_torch.build_linear(1 << log2_bases, LINEAR_SEED, order=args.order, with_lcp=False,
                                   with_samples=False, verbose=log)
    torch.cuda.empty_cache()
    log(f"linear index built on the GPU: n={ix.n} ({time.time() - t:.1f} s)")
    return ix


def measure(args, D, dev, gpu, flat, offsets, nq, m, steps, warmup, pack32=False):
    """Warm-up, then `steps` timed find() launches (+ gather when distributed).  Returns timings,
    the device result tensor and the algorithmic traffic of one launch."""
    import torch
    d_pat = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    # two result buffers: the RCCL gather of step k (async, on RCCL's own stream) overlaps the
    # find() launch of step k + 1; a buffer is reused only after its gather has completed.
    outs = [torch.zeros((nq, 2), dtype=torch.int64, device=dev) for _ in range(2 if D.active else 1)]
    # pack32: every value of a range is below 2^32 for this index, so the gather carries (sp, ep) as u32
    # pairs (8 bytes per query over xGMI instead of 16); the root keeps the gathered shards in that form
    pack32 = pack32 and D.active
    wire = [torch.zeros((nq, 2), dtype=torch.int32, device=dev) for _ in outs] if pack32 else outs
    parts = [[torch.zeros_like(wire[0]) for _ in range(D.world)] for _ in outs] if (D.active and D.rank == 0) else [None, None]
    pending = [None, None]
    stream = torch.cuda.current_stream()

    def step(k, record=None):
        b = k % len(outs)
        if pending[b] is not None:
            pending[b].wait()          # stream-level wait, no host sync
            pending[b] = None
        if record is not None:
            record[0].record(stream)
        gpu.find_device_variant(args.variant, d_pat.data_ptr(), d_off.data_ptr(), nq, outs[b].data_ptr(), stream.cuda_stream)
        if record is not None:
            record[1].record(stream)
        if pack32:
            wire[b].copy_(outs[b])     # int64 -> int32 keeps the low 32 bits
        pending[b] = D.gather(wire[b], parts[b])   # the single gather of hit ranges over xGMI

    def drain():
        for b in range(len(pending)):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for k in range(warmup):
        step(k)
    drain()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k, events[k])
    drain()
    torch.cuda.synchronize()
    D.barrier()
    d_out = outs[(steps - 1) % len(outs)] if steps > 0 else outs[0]
    elapsed = D.max(time.perf_counter() - t0, dev)
    if D.active and D.rank == 0 and steps > 0:       # the root's own shard came back through the gather intact
        mine = parts[(steps - 1) % len(outs)][0]
        back = (mine.to(torch.int64) & 0xFFFFFFFF) if pack32 else mine
        assert torch.equal(back, d_out), "gathered shard differs from the computed ranges"
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))

    # algorithmic traffic of one launch (instrumented kernel, outside the timed region)
    d_stats = torch.zeros(3, dtype=torch.int64, device=dev)
    d_out2 = torch.zeros_like(d_out)
    gpu.find_stats_device(d_pat.data_ptr(), d_off.data_ptr(), nq, d_out2.data_ptr(), d_stats.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(d_out, d_out2), "instrumented and timed kernels disagree"
    blocks, lf_steps, lookups = (int(x) for x in d_stats.cpu())
    algo_bytes = blocks * gpu.find_block_bytes() + lookups * 16 + nq * (m + 16)
    found = int((d_out[:, 0] <= d_out[:, 1]).sum().item())
    return dict(elapsed=elapsed, kernel_ms=kernel_ms, blocks=blocks, lf_steps=lf_steps, lookups=lookups, algo_bytes=algo_bytes,
                found=found, d_out=d_out)


def measure_locate(gpu, d_ranges, dev, steps):
    """Secondary figure (BASELINE configs[2]): locate() of the ranges the timed find() returned, into
    caller-owned device buffers (gcsa2_locate_into); reported beside the headline, never part of `value`."""
    import torch
    nq = int(d_ranges.shape[0])
    stream = torch.cuda.current_stream()
    d_off = torch.zeros(nq + 1, dtype=torch.int64, device=dev)
    d_val = torch.zeros(1, dtype=torch.int64, device=dev)
    try:
        total = gpu.locate_into(d_ranges.data_ptr(), nq, d_off.data_ptr(), d_val.data_ptr(), 1, stream.cuda_stream)
    except Exception as e:           # BUFFER_TOO_SMALL carries the size needed
        total = getattr(e, "needed", 0)
    d_val = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
    gpu.locate_into(d_ranges.data_ptr(), nq, d_off.data_ptr(), d_val.data_ptr(), d_val.shape[0], stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gpu.locate_into(d_ranges.data_ptr(), nq, d_off.data_ptr(), d_val.data_ptr(), d_val.shape[0], stream.cuda_stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # size-independent check (the query_gcsa consistency test, benchmark/query_gcsa.cpp:160-179): per range,
    # the number of located values equals count()
    d_cnt = torch.zeros(nq, dtype=torch.int64, device=dev)
    gpu.count_device(d_ranges.data_ptr(), nq, d_cnt.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    consistent = bool(torch.equal(d_off[1:] - d_off[:-1], d_cnt)) and int(d_off[-1]) == int(total)
    return {"count_equals_located": consistent,
            "workload": f"locate() of the {nq} ranges found above, sorted distinct values per range, results into caller-owned HBM buffers",
            "value": nq / (ms * 1e-3), "unit": "queries/s", "ms_per_step": ms, "values": int(total),
            "locate_table_bytes": gpu.locate_table_bytes()}


def pmc_traffic(args, key, nq, m, kmer_k):
    """Memory-side read bytes of one launch from the committed rocprofv3 --pmc pass of this exact
    workload (profiles/traffic.json, derivation in profiles/r01_v5_pmc.md).  PMC counters cannot
    be read from inside the timed process, so this is looked up, never estimated: any mismatch in
    workload, batch shape or kernel generation yields None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            entry = json.load(f).get(key)
    except (OSError, ValueError):
        return None
    if not entry or args.variant != 2 or args.set != "S" or entry["queries"] != nq or entry["pattern_len"] != m:
        return None
    if entry.get("kmer_table_k") != kmer_k:        # profiled with another seed table: not this kernel's traffic
        return None
    return entry["read_bytes_per_launch"]


def roofline(args, r, key, nq, m, kmer_k):
    achieved = r["algo_bytes"] / (r["kernel_ms"] * 1e-3) / 1e9
    traffic = pmc_traffic(args, key, nq, m, kmer_k)
    out = {"bound": "hbm", "kernel": "k_find2" if args.variant == 2 else "k_find", "achieved": achieved,
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
           "algorithmic_bytes_per_launch": r["algo_bytes"], "kernel_ms": r["kernel_ms"]}
    # what actually bounds a random-gather kernel beyond L2 is requests per second (profiles/r01_gather_bench.md:
    # 50-58 G dependent random 128-byte fetches/s on this GPU); reported beside the byte roofline
    requests = r["blocks"] + r["lookups"]
    out["request_rate"] = {"achieved_G_per_s": requests / (r["kernel_ms"] * 1e-3) / 1e9, "ceiling_G_per_s": REQUEST_CEILING_GPS,
                           "requests_per_query": requests / nq}
    if traffic is not None:
        out["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc TCC_EA0_RDREQ_*_sum pass of this workload; 128 B x RDREQ_128B + 64 B x RDREQ_64B)"
        out["traffic_GBps"] = traffic / (r["kernel_ms"] * 1e-3) / 1e9
    return out


def main():
    args = parse()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GCSA2_BENCH_BACKEND", "nccl") != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    D = Dist(dev)
    rank, world = D.rank, D.world
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}")

    from workload import patterns, linear_torch
    from gcsa2_amd.binding import GCSA

    nq, m = args.queries, args.pattern_len
    seed = 0x6C5A0012 + 0x1000 * rank
    log2_bases = args.log2_bases or (25 if args.workload == "snp" else 30)
    if args.workload == "snp":
        ix, graph = build_snp_index(args, log2_bases, rank, D.barrier)
        label = f"chr22-like SNP graph 2^{log2_bases} bases"
    else:
        ix, graph = build_linear_index(args, log2_bases), None
        label = f"linear graph 2^{log2_bases} bases (FM-index shaped GCSA)"
    t = time.time()
    full = args.workload == "snp"
    gpu = GCSA(ix, device=local_rank, with_samples=full, with_counters=full, with_lcp=full)
    log(f"device image: {gpu.device_bytes() / 1e6:.1f} MB in HBM ({time.time() - t:.1f} s)")
    t = time.time()
    if args.set == "U":
        pats = patterns.uniform_patterns(nq, m, seed)
    elif args.workload == "snp":
        pats = patterns.walk_patterns(graph, nq, m, seed)
    else:
        pats = linear_torch.substring_patterns_torch(1 << log2_bases, LINEAR_SEED, nq, m, seed, dev)
    flat, offsets = patterns.as_batch(pats)
    log(f"patterns: {nq} x {m} set {args.set} ({time.time() - t:.1f} s)")

    pack32 = max(int(ix.n), int(ix.e)) + 2 < (1 << 32)
    r = measure(args, D, dev, gpu, flat, offsets, nq, m, args.steps, args.warmup, pack32=pack32)

    result = None
    if rank == 0:
        result = {
            "metric": "kmer_find_queries_per_sec", "value": world * nq * args.steps / r["elapsed"], "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{label}, order-{args.order} GCSA, {nq} x {m}-mer find() per GPU, pattern set {args.set}",
                       "path_nodes": int(ix.n), "edges": int(ix.e), "queries_per_gpu": nq, "pattern_len": m,
                       "pattern_set": args.set, "index_bytes_hbm": gpu.device_bytes(),
                       "find_bytes_hbm": int(ix.sigma) * (int(ix.n) // 448 + 1) * 128,
                       "found": r["found"], "lf_steps_per_query": r["lf_steps"] / nq,
                       "blocks_per_query": r["blocks"] / nq, "block_bytes": gpu.find_block_bytes(),
                       "kmer_table_k": gpu.kmer_table_k(),
                       "parallelism": f"replicated index, query shards x{world}, one RCCL gather of ranges per step"
                                      + (" ((sp, ep) as u32 pairs: every value of this index is below 2^32)" if (pack32 and world > 1) else "")},
            "roofline": roofline(args, r, f"{args.workload}_{log2_bases}", nq, m, gpu.kmer_table_k()),
        }
        if args.workload == "snp":
            result["roofline"]["note"] = ("fused blocks of this index fit the 256 MiB Infinity Cache: achieved = algorithmic bytes / "
                                          "kernel time, served mostly on-die; see hbm_resident for the HBM-bound figure")
        if not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(args, ix, flat, offsets, r["d_out"], m)
        if args.workload == "snp" and world == 1:
            result["locate"] = measure_locate(gpu, r["d_out"], dev, max(3, args.steps // 4))
        if args.workload == "snp" and world == 1 and not args.no_jump_table:
            # opt-in acceleration structure (GCSA2_JUMP_TABLE=1, 16 bytes per path node): fewer, smaller
            # memory requests per query; reported beside the headline, which stays the default build
            os.environ["GCSA2_JUMP_TABLE"] = "1"
            try:
                gpu_j = GCSA(ix, device=local_rank, with_samples=False, with_counters=False, with_lcp=False)
            finally:
                del os.environ["GCSA2_JUMP_TABLE"]
            rj = measure(args, D, dev, gpu_j, flat, offsets, nq, m, max(5, args.steps // 2), 2)
            result["jump_table"] = {
                "workload": "the headline workload with the jump table (memoised unary LF chains) enabled",
                "value": nq / (rj["kernel_ms"] * 1e-3), "unit": "queries/s", "kernel_ms": rj["kernel_ms"],
                "table_bytes": gpu_j.jump_table_bytes(), "blocks_per_query": rj["blocks"] / nq,
                "algorithmic_GBps": rj["algo_bytes"] / (rj["kernel_ms"] * 1e-3) / 1e9,
                "requests_per_query": (rj["blocks"] + rj["lookups"]) / nq,
                "request_rate_G_per_s": (rj["blocks"] + rj["lookups"]) / (rj["kernel_ms"] * 1e-3) / 1e9,
                "equals_default_results": bool(torch.equal(rj["d_out"], r["d_out"]))}
            del gpu_j, rj
    del r

    # secondary measurement: the same kernel on an index far larger than the Infinity Cache
    if args.workload == "snp" and world == 1 and not args.no_hbm_resident:
        del gpu, ix, graph
        torch.cuda.empty_cache()
        lb = 30
        ix2 = build_linear_index(args, lb)
        gpu2 = GCSA(ix2, device=local_rank, with_samples=False, with_counters=False, with_lcp=False)
        pats2 = linear_torch.substring_patterns_torch(1 << lb, LINEAR_SEED, nq, m, seed, dev)
        flat2, off2 = patterns.as_batch(pats2)
        r2 = measure(args, D, dev, gpu2, flat2, off2, nq, m, max(5, args.steps // 2), 2)
        result["hbm_resident"] = {
            "workload": f"linear graph 2^{lb} bases (FM-index shaped GCSA, built on the GPU), {nq} x {m}-mer find(), substrings of the text",
            "path_nodes": int(ix2.n), "find_bytes_hbm": int(ix2.sigma) * (int(ix2.n) // 448 + 1) * 128,
            "value": nq / (r2["kernel_ms"] * 1e-3), "unit": "queries/s", "blocks_per_query": r2["blocks"] / nq,
            "kmer_table_k": gpu2.kmer_table_k(),
            "roofline": roofline(args, r2, f"linear_{lb}", nq, m, gpu2.kmer_table_k())}
    if rank == 0:
        print(json.dumps(result), flush=True)
    D.barrier()
    D.close()


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
                      f"= the CPUs this container may use (affinity / cgroup quota; {os.cpu_count()} logical CPUs visible) "
                      f"({tall:.1f} s); single thread: first {n1} patterns ({t1:.1f} s)",
            "single_thread_value": n1 / t1, "single_thread_us_per_query": t1 / n1 * 1e6,
            "gpu_matches_cpu_on_sample": parity}


if __name__ == "__main__":
    main()
