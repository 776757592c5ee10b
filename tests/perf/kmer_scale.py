#!/usr/bin/env python3
"""countKMers on the pangenome-sized index (5.73 G path nodes): the level-synchronous frontier with its splits beyond 64 M
states, against the closed form -- the distinct k-prefixes of the node labels, read off the bitmap of the 17-mer universe
(every k-mer of a de Bruijn graph of order 17, k <= 17, is a prefix of a node label; junction edges add none).

    python tests/perf/kmer_scale.py [--degree 34] [--ks 8,12,14,16]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree", type=int, default=34)
    ap.add_argument("--ks", default="8,12,14,16")
    args = ap.parse_args()
    import torch
    from workload import dbg_torch
    from gcsa2_amd.binding import GCSA
    dev = torch.device("cuda", 0)
    ix, dbg = dbg_torch.build_dbg(args.degree, junctions=80, device=dev)
    order = args.degree // 2
    cmp_want = {k: dbg_torch.distinct_prefixes(dbg.nodes, order, k) for k in (10, 14)}
    dbg_torch_count = cmp_want.get
    want = {k: dbg_torch.distinct_prefixes(dbg.nodes, order, k) for k in (int(x) for x in args.ks.split(","))}
    del dbg
    torch.cuda.empty_cache()
    gpu = GCSA(ix, device=0, with_samples=False, with_counters=False, with_lcp=False)
    out = {"path_nodes": int(ix.n), "image_GB": gpu.device_bytes() / 1e9}
    for k, expect in want.items():
        t0 = time.perf_counter()
        got = gpu.count_kmers(k)
        dt = time.perf_counter() - t0
        out[f"k={k}"] = {"kmers": got, "closed_form": expect, "equal": got == expect, "seconds": round(dt, 3), "G_states_per_s": round(got / dt / 1e9, 3)}
        print(json.dumps({f"k={k}": out[f"k={k}"]}), flush=True)
    # compareKMers of the index with itself: every k-mer shared; at k = 14 the frontier (268 M states x 4 children) needs pieces
    for k in (10, 14):
        if k in want or True:
            t0 = time.perf_counter()
            got = gpu.compare_kmers(gpu, k)
            dt = time.perf_counter() - t0
            expect = dbg_torch_count(k)
            out[f"compare k={k}"] = {"result": got, "equal": got == (expect, 0, 0), "seconds": round(dt, 3)}
            print(json.dumps({f"compare k={k}": out[f"compare k={k}"]}), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
