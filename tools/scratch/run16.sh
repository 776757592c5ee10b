# the same passes as run15, keeping only what the summaries need (counter CSVs are deleted after summarising)
mkdir -p gpurun_out/keep
bash tools/pmc_passes.sh r02 human human_snp chr22 linear > /dev/null 2>&1
echo '{}' > profiles/traffic.json
python tools/pmc_summary.py r02 human human_snp chr22 linear --write-traffic > gpurun_out/keep/summary_main.txt
for W in human human_snp chr22 linear; do cp gpurun_out/r02_${W}_trace/x_kernel_stats.csv gpurun_out/keep/r02_${W}_kernel_stats.csv; grep "^{" gpurun_out/r02_${W}_trace.log > gpurun_out/keep/r02_${W}_line.json; done
for M in 16 64 128; do
  Q=100000000; [ $M -ge 64 ] && Q=20000000
  PASSES="rdreq trace" EXTRA="--pattern-len $M --queries $Q" bash tools/pmc_passes.sh t3m$M human > /dev/null 2>&1
  PASSES="rdreq trace" EXTRA="--pattern-len $M --queries 10000000" bash tools/pmc_passes.sh t3m$M linear > /dev/null 2>&1
  python tools/pmc_summary.py t3m$M human linear --write-traffic >> gpurun_out/keep/summary_sweep.txt
  for W in human linear; do cp gpurun_out/t3m${M}_${W}_trace/x_kernel_stats.csv gpurun_out/keep/r02_${W}_m${M}_kernel_stats.csv; grep "^{" gpurun_out/t3m${M}_${W}_trace.log > gpurun_out/keep/r02_${W}_m${M}_line.json; done
done
PASSES="rdreq trace" EXTRA="--set U" bash tools/pmc_passes.sh setU human linear chr22 > /dev/null 2>&1
python tools/pmc_summary.py setU human linear chr22 --set U --write-traffic >> gpurun_out/keep/summary_sweep.txt
for W in human linear chr22; do cp gpurun_out/setU_${W}_trace/x_kernel_stats.csv gpurun_out/keep/r02_${W}_setU_kernel_stats.csv; grep "^{" gpurun_out/setU_${W}_trace.log > gpurun_out/keep/r02_${W}_setU_line.json; done
cp profiles/traffic.json gpurun_out/keep/traffic.json
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --queries 10000000 --steps 2 --warmup 1 --no-cpu --secondary config5"
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --output-format csv -d $ROOT/gpurun_out/c5_rdreq -o x -- $CMD > $ROOT/gpurun_out/c5_rdreq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/c5_trace -o x -- $CMD > $ROOT/gpurun_out/c5_trace.log 2>&1
cd $ROOT
python tools/pmc_kernel.py gpurun_out/c5_rdreq k_parent k_match_stats2 k_locate_tab > gpurun_out/keep/summary_config5.txt
python tools/pmc_kernel.py gpurun_out/c5_trace k_parent k_match_stats2 k_locate_tab >> gpurun_out/keep/summary_config5.txt
cp gpurun_out/c5_trace/x_kernel_stats.csv gpurun_out/keep/r02_config5_kernel_stats.csv
grep "^{" gpurun_out/c5_trace.log > gpurun_out/keep/r02_config5_line.json
# the default bench line, unprofiled
python bench.py > gpurun_out/keep/r02_bench.json 2> gpurun_out/keep/r02_bench.err
mv gpurun_out/keep /tmp/keep; rm -rf gpurun_out/*; mv /tmp/keep gpurun_out/keep
cat gpurun_out/keep/summary_main.txt | grep -E "^##|traffic|trace"; du -sh gpurun_out
