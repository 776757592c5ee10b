mkdir -p gpurun_out/r05k; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for S in 1 0; do
GCSA2_LOCATE_SPLIT_SORT=$S rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05k/trace_split$S -o t -- python bench.py --workload repeats30 --pattern-len 16 --locate-ranges 100000 --locate --no-cpu --no-secondary --no-extras --steps 2 --warmup 1 --full-json gpurun_out/r05k/tmp.json > /dev/null 2>> gpurun_out/r05k/err
done
python - <<'P'
import csv,glob
for S in (1,0):
  print("split sort", S)
  for f in glob.glob(f"gpurun_out/r05k/trace_split{S}/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[:16]: print("  ", r["Name"][:80], r["Calls"], round(float(r["TotalDurationNs"])/1e6,2), "ms total", round(float(r["AverageNs"])/1e3,1), "us avg")
P
