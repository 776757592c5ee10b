mkdir -p gpurun_out/r05n; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05n
for K in "2" "3"; do
  for PASS in rd wr; do
    if [ $PASS = rd ]; then PMC="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum"; else PMC="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; fi
    GCSA2_MS_KERNEL=$K rocprofv3 --kernel-include-regex "k_match_stats" --pmc $PMC --output-format csv -d /tmp/pmc_k${K}_$PASS -o t -- python tests/perf/ms_ab.py --configs "GCSA2_MS_KERNEL=$K" --reps 1 > $OUT/ab_k${K}_$PASS.jsonl 2>> $OUT/err.log
    echo "## kernel $K pass $PASS" >> $OUT/pmc_summary.txt; python tools/pmc_kernel_requests.py /tmp/pmc_k${K}_$PASS k_match_stats | cut -c1-400 >> $OUT/pmc_summary.txt
  done
done
for K in "2" "3"; do
  GCSA2_MS_KERNEL=$K rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_k$K -o t -- python tests/perf/ms_ab.py --configs "GCSA2_MS_KERNEL=$K" --reps 5 > $OUT/ab_trace_k$K.jsonl 2>> $OUT/err.log
  python - >> $OUT/trace_summary.txt <<P
import csv,glob
for f in glob.glob("/tmp/trace_k$K/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(x in r["Name"] for x in ("k_match_stats","k_pack_records","k_breaks")): print("K=$K", r["Name"][:140], "calls", r["Calls"], "avg_ms", round(float(r["AverageNs"])/1e6,4))
P
done
GCSA2_MS_KERNEL=2 python tests/perf/ms_profile.py > $OUT/profile_k2.json 2>> $OUT/err.log
GCSA2_MS_KERNEL=3 python tests/perf/ms_profile.py > $OUT/profile_k3.json 2>> $OUT/err.log
cat $OUT/pmc_summary.txt $OUT/trace_summary.txt | cut -c1-300
