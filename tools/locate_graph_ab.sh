#!/bin/bash
# Kernel traces of the locate() legs of bench.py on the 2^23 repeat-rich GRAPH (duplicates: five raw values per distinct one):
#   tools/locate_graph_ab.sh <tag> [ENV=VALUE ...]
# 32-mer and 16-mer batches, 400 k ranges each, five locate() calls under rocprofv3 --kernel-trace --stats.
set -u
TAG=$1
shift
for kv in "$@"; do export "$kv"; done
ROOT=$(pwd); mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
for m in 32 16; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_g$m -o x -- python $ROOT/bench.py --workload repeats --pattern-len $m --locate --steps 5 --warmup 1 --no-cpu --no-secondary --no-extras --full-json $ROOT/gpurun_out/${TAG}_g$m.json > $ROOT/gpurun_out/${TAG}_g$m.log 2>&1
  python - <<PY
import csv,glob
rows=[]
for f in glob.glob("$ROOT/gpurun_out/${TAG}_g$m/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if any(k in n for k in ("k_locate_","k_over_","k_sort_","k_compact","k_mark_","k_dedup","k_collect","k_block_owners","k_word_counts","k_final_offsets","k_classify","DeviceScan","lookback_scan")):
            rows.append((float(r["TotalDurationNs"]),int(r["Calls"]),n[:90]))
rows.sort(reverse=True)
print("== ${TAG}: repeat graph 2^23, $m-mers: kernel, calls, total ms, avg us")
for t,c,n in rows[:9]: print(f"{n:90s} {c:5d} {t/1e6:9.3f} {t/c/1e3:10.1f}")
PY
  grep -o '"ms_per_step": [0-9.]*' $ROOT/gpurun_out/${TAG}_g$m.json | tail -1
  grep -o '"count_equals_located": [a-z]*' $ROOT/gpurun_out/${TAG}_g$m.json | tail -1
  rm -f $ROOT/gpurun_out/${TAG}_g$m/*kernel_trace.csv $ROOT/gpurun_out/${TAG}_g$m/*agent_info.csv $ROOT/gpurun_out/${TAG}_g$m/*domain_stats.csv
done
