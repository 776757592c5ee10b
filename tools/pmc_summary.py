#!/usr/bin/env python3
"""Summarise the passes of tools/pmc_passes.sh: mean counter value per k_find2 launch and the derived
memory-side read traffic (128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B; MI355X_MICROARCH.md: on
gfx950 FETCH_SIZE x 1024 counts a 128-byte request as 64 bytes).

    python tools/pmc_summary.py <tag> [--write-traffic]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def is_default_find(name):
    """k_find2<STATS=false, REFILL=false[, JUMP=false]>: the timed default kernel."""
    return "k_find2<false, false>" in name or "k_find2<false, false, false" in name


def counters(directory):
    """counter name -> mean per dispatch of k_find2<false, ...>, summed over the XCD instances of a dispatch."""
    per = defaultdict(lambda: defaultdict(float))
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if not is_default_find(row["Kernel_Name"]):
                continue
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {name: sum(d.values()) / len(d) for name, d in per.items()}, max((len(d) for d in per.values()), default=0)


def main():
    tag = sys.argv[1]
    out = {}
    for wl, key in (("snp", "snp_25"), ("linear", "linear_30")):
        merged = {}
        launches = 0
        for group in ("rdreq", "l2", "fetch"):
            c, n = counters(os.path.join(ROOT, "gpurun_out", f"{tag}_{wl}_{group}"))
            merged.update(c)
            launches = max(launches, n)
        if not merged:
            continue
        traffic = 128 * merged.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * merged.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * merged.get("TCC_EA0_RDREQ_32B_sum", 0)
        hit, miss = merged.get("TCC_HIT_sum", 0), merged.get("TCC_MISS_sum", 0)
        print(f"## {wl}: {launches} launches of k_find2<false>")
        for name in sorted(merged):
            print(f"| {wl} | {name} | {merged[name]:.6g} |")
        print(f"read traffic per launch = {traffic / 1e9:.3f} GB; L2 hit rate = {hit / max(hit + miss, 1):.3f}; FETCH_SIZE x 1024 = {merged.get('FETCH_SIZE', 0) * 1024 / 1e9:.3f} GB\n")
        entry = {"kernel": "k_find2", "queries": 10000000, "pattern_len": 32, "read_bytes_per_launch": traffic}
        try:                              # the profiled run's own JSON line: which seed table the launches used
            lines = [l for l in open(os.path.join(ROOT, "gpurun_out", f"{tag}_{wl}_rdreq.log")) if l.startswith("{")]
            cfg = json.loads(lines[-1])["config"]
            entry.update(queries=cfg["queries_per_gpu"], pattern_len=cfg["pattern_len"], kmer_table_k=cfg["kmer_table_k"],
                         blocks_per_query=cfg["blocks_per_query"])
        except (OSError, IndexError, KeyError, ValueError):
            pass
        out[key] = entry
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_trace", "**", "*kernel_stats.csv"), recursive=True)
    for path in stats:
        for row in csv.DictReader(open(path)):
            if is_default_find(row["Name"]):
                print(f"kernel trace: k_find2<false> calls={row['Calls']} average={float(row['AverageNs']) / 1e6:.4f} ms")
    if "--write-traffic" in sys.argv and out:
        out["_source"] = (f"tools/pmc_passes.sh {tag} + tools/pmc_summary.py: rocprofv3 --pmc TCC_EA0_RDREQ_{{32B,64B,128B}}_sum passes of bench.py, "
                          "read bytes per launch of 10 M 32-mers")
        with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
            json.dump(out, f, indent=1)
        print("wrote profiles/traffic.json")


if __name__ == "__main__":
    main()
