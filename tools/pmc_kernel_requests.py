#!/usr/bin/env python3
"""Per-dispatch sums of the counters of a rocprofv3 --pmc pass (csv), for the kernels whose name contains a pattern:
    python tools/pmc_kernel_requests.py <directory> <pattern>
One line per counter: the values of the first dispatches, in millions."""
import collections
import csv
import glob
import sys

directory, pattern = sys.argv[1], sys.argv[2]
per, names = collections.defaultdict(lambda: collections.defaultdict(float)), {}
for path in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if pattern in row["Kernel_Name"]:
            per[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
            names[int(row["Dispatch_Id"])] = row["Kernel_Name"]
for counter, d in sorted(per.items()):
    vals = sorted(d.items())
    print(counter, " ".join(f"{v / 1e6:.2f}" for _, v in vals[:16]))
for k in sorted(names)[:16]:
    print(k, names[k][:150])
