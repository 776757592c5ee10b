#!/usr/bin/env python3
"""Summarise the passes of tools/pmc_passes.sh: mean counter value per launch of the timed find kernel and the
derived memory-side read traffic (128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B; MI355X_MICROARCH.md: on
gfx950 FETCH_SIZE x 1024 counts a 128-byte request as 64 bytes).

    python tools/pmc_summary.py <tag> [workload ...] [--write-traffic] [--set U]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# k_find2<STATS = false, JUMP = false, PAIR = *>: the timed default kernel (rounds 1-2: <false, false, false, true, PAIR>)
# (round 4 added a fourth parameter, PACKED; the timed byte-pattern kernel is <false, false, PAIR, false>)
TIMED = re.compile(r"k_find2<false, false, (true|false)(, false)?>|k_find2<false, false, false, true, (true|false)>")


def counters(directory):
    """counter name -> mean per dispatch of the timed kernel, summed over the XCD instances of a dispatch."""
    per = defaultdict(lambda: defaultdict(float))
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if TIMED.search(row["Kernel_Name"]):
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {name: sum(d.values()) / len(d) for name, d in per.items()}, max((len(d) for d in per.values()), default=0)


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag, workloads = argv[0], (argv[1:] or ["human"])
    pattern_set = sys.argv[sys.argv.index("--set") + 1] if "--set" in sys.argv else "S"
    workloads = [w for w in workloads if w != pattern_set or w in ("human", "chr22", "linear")]
    out = {}
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        out = json.load(open(path))
    for wl in workloads:
        merged, launches = {}, 0
        for group in ("rdreq", "l2", "fetch"):
            c, n = counters(os.path.join(ROOT, "gpurun_out", f"{tag}_{wl}_{group}"))
            merged.update(c)
            launches = max(launches, n)
        if not merged:
            continue
        traffic = 128 * merged.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * merged.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * merged.get("TCC_EA0_RDREQ_32B_sum", 0)
        hit, miss = merged.get("TCC_HIT_sum", 0), merged.get("TCC_MISS_sum", 0)
        print(f"## {wl}: {launches} launches of the timed k_find2")
        for name in sorted(merged):
            print(f"| {wl} | {name} | {merged[name]:.6g} |")
        print(f"read traffic per launch = {traffic / 1e9:.3f} GB; L2 hit rate = {hit / max(hit + miss, 1):.3f}; "
              f"FETCH_SIZE x 1024 = {merged.get('FETCH_SIZE', 0) * 1024 / 1e9:.3f} GB")
        entry = {"kernel": "k_find2", "read_bytes_per_launch": traffic}
        try:                              # the profiled run's own JSON line: batch shape, seed table, pair blocks
            lines = [l for l in open(os.path.join(ROOT, "gpurun_out", f"{tag}_{wl}_rdreq.log")) if l.startswith("{")]
            line = json.loads(lines[-1])
            cfg = line["config"]
            entry.update(queries=cfg["queries_per_gpu"], pattern_len=cfg["pattern_len"], kmer_table_k=cfg["kmer_table_k"],
                         pair_blocks=bool(cfg["pair_block_bytes"]), blocks_per_query=cfg["blocks_per_query"],
                         algorithmic_bytes_per_launch=line["roofline"]["algorithmic_bytes_per_launch"])
            size = re.search(r"degree-(\d+)|2\^(\d+)", cfg["workload"])
            key = f"{wl}_{size.group(1) or size.group(2)}_{cfg['pattern_len']}_{pattern_set}"
            print(f"algorithmic bytes per launch = {entry['algorithmic_bytes_per_launch'] / 1e9:.3f} GB "
                  f"(traffic / algorithmic = {traffic / entry['algorithmic_bytes_per_launch']:.3f})")
        except (OSError, IndexError, KeyError, ValueError, AttributeError) as e:
            print(f"(no JSON line for {wl}: {e})")
            continue
        out[key] = entry
        for path2 in glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{wl}_trace", "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(path2)):
                if TIMED.search(row["Name"]):
                    print(f"kernel trace: timed k_find2 calls={row['Calls']} average={float(row['AverageNs']) / 1e6:.4f} ms")
        print()
    if "--write-traffic" in sys.argv and out:
        out["_source"] = ("tools/pmc_passes.sh + tools/pmc_summary.py: rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum passes of bench.py, "
                          "read bytes per launch of the timed k_find2; keys = workload_size_patternlen_set")
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        print("wrote profiles/traffic.json")


if __name__ == "__main__":
    main()
