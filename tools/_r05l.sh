mkdir -p gpurun_out/r05l; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "locate" > gpurun_out/r05l/pytest_loc.log 2>&1; grep -E "passed|failed" gpurun_out/r05l/pytest_loc.log; grep -n "^E " gpurun_out/r05l/pytest_loc.log | head
for W in "repeats30 --pattern-len 16 --locate-ranges 100000" "repeats30 --pattern-len 32 --locate-ranges 400000" "repeats --locate-ranges 400000"; do for S in 1 0; do GCSA2_LOCATE_SPLIT_SORT=$S timeout 400 python bench.py --workload $W --locate --no-cpu --no-secondary --no-extras --steps 3 --warmup 1 --full-json gpurun_out/r05l/tmp.json > /dev/null 2>> gpurun_out/r05l/err.log; python -c "
import json,sys; d=json.load(open('gpurun_out/r05l/tmp.json')); l=d.get('locate',{}); print('$W', 'split=$S', l.get('ms_per_step'), l.get('values'), l.get('count_equals_located'))"; done; done
GCSA2_LOCATE_SPLIT_SORT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05l/trace -o t -- python bench.py --workload repeats30 --pattern-len 16 --locate-ranges 100000 --locate --no-cpu --no-secondary --no-extras --steps 2 --warmup 1 --full-json gpurun_out/r05l/tmp.json > /dev/null 2>> gpurun_out/r05l/err.log
python - <<'P'
import csv,glob
for f in glob.glob("gpurun_out/r05l/trace/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows:
        n=r["Name"]
        if any(x in n for x in ("k_over","k_sort","k_dedup","k_locate_tab","k_compact","k_mark","400200")) and float(r["TotalDurationNs"])>2e5:
            print("  ", n[:90], r["Calls"], round(float(r["TotalDurationNs"])/1e6,2), "ms total", round(float(r["AverageNs"])/1e3,1), "us avg")
P
