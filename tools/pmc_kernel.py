#!/usr/bin/env python3
"""Mean counter values per dispatch of the kernels whose name contains a substring, from a rocprofv3 --pmc output
directory (summed over the XCD instances of a dispatch).

    python tools/pmc_kernel.py gpurun_out/<dir> k_parent [k_match_stats2 ...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    directory, names = sys.argv[1], sys.argv[2:]
    per = {n: defaultdict(lambda: defaultdict(float)) for n in names}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            for n in names:
                if n in row["Kernel_Name"]:
                    per[n][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for n in names:
        for counter, d in sorted(per[n].items()):
            print(f"| {n} | {counter} | {len(d)} dispatches | mean {sum(d.values()) / len(d):.6g} |")
    for path in glob.glob(os.path.join(directory, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            for n in names:
                if n in row["Name"]:
                    print(f"| {n} | kernel trace | {row['Calls']} calls | average {float(row['AverageNs']) / 1e6:.4f} ms |")


if __name__ == "__main__":
    main()
