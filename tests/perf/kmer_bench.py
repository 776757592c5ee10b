#!/usr/bin/env python3
"""countKMers / compareKMers on the GPU beside the CPU oracle (reference benchmark/count_kmers.cpp):
chr22-like SNP graph 2^22 (6.1 M path nodes); each k run three times, best time reported."""
import sys
import time

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from workload import graphs, builder
from gcsa2_amd.binding import open_index
from oracle.oracle import OracleIndex, max_threads


def best(fn, reps=3):
    times, value = [], None
    for _ in range(reps):
        t = time.perf_counter()
        value = fn()
        times.append(time.perf_counter() - t)
    return value, min(times)


def main():
    g = graphs.snp_graph(1 << 22, 0x6C5A0010, 0x6C5A0011)
    ix = builder.build(g, 256, keep_table=False)
    g2 = graphs.snp_graph(1 << 22, 0x6C5A0010, 0x6C5A0077)
    ix2 = builder.build(g2, 256, keep_table=False)
    gpu, _ = open_index(ix)
    gpu2, _ = open_index(ix2)
    cpu, cpu2 = OracleIndex(ix), OracleIndex(ix2)
    threads = max_threads()
    for k in (8, 12, 16, 24, 32):
        a, tg = best(lambda: gpu.count_kmers(k))
        b, tc = best(lambda: cpu.count_kmers(k, threads=threads), reps=1)
        print(f"countKMers k={k}: gpu {a} in {tg * 1e3:.2f} ms, cpu({threads} threads) {b} in {tc * 1e3:.1f} ms, equal={a == b}")
    for k in (12, 16):
        a, tg = best(lambda: gpu.compare_kmers(gpu2, k))
        b, tc = best(lambda: cpu.compare_kmers(cpu2, k), reps=1)
        print(f"compareKMers k={k}: gpu {a} in {tg * 1e3:.2f} ms, cpu(1 thread) {b} in {tc * 1e3:.1f} ms, equal={a == b}")


if __name__ == "__main__":
    main()
