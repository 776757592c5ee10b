#!/bin/bash
# Kernel traces of the locate() legs of bench.py on the 2^30-base repeat text (run on the GPU box, from the repo root):
#   tools/locate_ab.sh <tag> [ENV=VALUE ...]      e.g. tools/locate_ab.sh r06_base GCSA2_LOCATE_FUSE=0 GCSA2_LOCATE_IN_PLACE=0
# 16-mer batch (100 k ranges, 544 M values) and 32-mer batch (400 k ranges, 365 M values), five locate() calls each under
# rocprofv3 --kernel-trace --stats; outputs under gpurun_out/<tag>_{16,32}/ and the bench lines in gpurun_out/<tag>_{16,32}.json.
set -u
TAG=$1
shift
for kv in "$@"; do export "$kv"; done
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
ONLY="--kernel-include-regex k_locate_|k_over_|k_sort_|k_compact|k_mark_|k_dedup_huge|k_collect_multi|k_block_owners|k_word_counts|k_final_offsets|k_huge_to_over|k_classify|k_publish|DeviceScan|lookback_scan"
for m in 16 32; do
  if [ $m = 16 ]; then R=100000; else R=400000; fi
  timeout ${PASS_TIMEOUT:-900} rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_$m -o x -- \
    python $ROOT/bench.py --workload repeats30 --pattern-len $m --locate-ranges $R --locate --steps 5 --warmup 1 --no-cpu --no-secondary --no-extras \
      --full-json $ROOT/gpurun_out/${TAG}_$m.json > $ROOT/gpurun_out/${TAG}_$m.log 2>&1
  python - <<PY
import csv, glob
rows = []
for f in glob.glob("$ROOT/gpurun_out/${TAG}_$m/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("k_locate_", "k_over_", "k_sort_", "k_compact", "k_mark_", "k_dedup", "k_collect", "k_block_owners", "k_word_counts", "k_final_offsets", "k_classify", "k_publish", "DeviceScan", "lookback_scan")):
            rows.append((float(r["TotalDurationNs"]), int(r["Calls"]), n[:90]))
rows.sort(reverse=True)
print("== ${TAG} $m-mers: kernel, calls, total ms, avg us")
for t, c, n in rows[:16]:
    print(f"{n:90s} {c:5d} {t/1e6:9.3f} {t/c/1e3:10.1f}")
PY
  grep -o '"ms_per_step": [0-9.]*' $ROOT/gpurun_out/${TAG}_$m.json | tail -1
  rm -f $ROOT/gpurun_out/${TAG}_$m/*kernel_trace.csv $ROOT/gpurun_out/${TAG}_$m/*agent_info.csv $ROOT/gpurun_out/${TAG}_$m/*domain_stats.csv      # (gpurun_out travels back: keep it small)
done
