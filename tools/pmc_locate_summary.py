#!/usr/bin/env python3
"""Memory-side traffic of ONE locate() call from the two passes of tools/pmc_locate.sh:
    python tools/pmc_locate_summary.py gpurun_out/<tag>_locate [calls]
Reads 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B (MI355X_MICROARCH.md, HBM section); writes 64 x WRREQ_64B + 32 x the
other write requests (uncalibrated there: reported apart).  The counters are summed over every dispatch of the engine's locate
kernels and divided by the number of locate() calls the bench leg makes (two that size the buffers, the warm-up, the timed
steps: 5 with --steps 2; pass another count otherwise).  Prints one JSON object; per-kernel shares go to stderr."""
import collections
import csv
import glob
import json
import sys

prefix = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def sums(directory):
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            per_kernel[name][row["Counter_Name"]] += float(row["Counter_Value"])
    return per_kernel


rd, wr = sums(prefix + "_rdreq"), sums(prefix + "_wrreq")
read_of = {k: 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) for k, c in rd.items()}
write_of = {k: 64 * c.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (c.get("TCC_EA0_WRREQ_sum", 0) - c.get("TCC_EA0_WRREQ_64B_sum", 0)) for k, c in wr.items()}
for k in sorted(set(read_of) | set(write_of), key=lambda k: -(read_of.get(k, 0) + write_of.get(k, 0))):
    print(f"{k[:48]:48s} read {read_of.get(k, 0) / calls / 1e9:8.3f} GB  write {write_of.get(k, 0) / calls / 1e9:8.3f} GB per call", file=sys.stderr)
out = {"read_bytes_per_call": sum(read_of.values()) / calls, "write_bytes_per_call": sum(write_of.values()) / calls, "calls": calls,
       "read_requests_per_call": sum(c.get("TCC_EA0_RDREQ_sum", 0) for c in rd.values()) / calls,
       "write_requests_per_call": sum(c.get("TCC_EA0_WRREQ_sum", 0) for c in wr.values()) / calls}
print(json.dumps(out))
