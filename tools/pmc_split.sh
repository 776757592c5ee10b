#!/bin/bash
# SQ counter passes of the locate split and bucket sort (run on the GPU box, from the repo root):
#   tools/pmc_split.sh <tag> <pattern-len> <ranges>
# Where k_over_split / k_sort_bucket spend their wave cycles: parked (s_waitcnt / barrier), issue-stalled, LDS busy and bank conflicts.
set -u
TAG=$1; M=$2; R=$3
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload repeats30 --pattern-len $M --locate-ranges $R --locate --steps 2 --warmup 1 --no-cpu --no-secondary --no-extras --no-measured-traffic --full-json /tmp/pmc_split_full.json"
ONLY="--kernel-include-regex k_over_split|k_sort_bucket"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 $ONLY --pmc $set --output-format csv -d $ROOT/gpurun_out/${TAG}_$name -o x -- $CMD > $ROOT/gpurun_out/${TAG}_$name.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in glob.glob("$ROOT/gpurun_out/${TAG}_SQ_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, c in tot.items():
    print(k)
    for name in sorted(c): print(f"   {name:24s} {c[name]:.4g}")
PY
rm -rf $ROOT/gpurun_out/${TAG}_SQ_*
