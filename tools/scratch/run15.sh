bash tools/pmc_passes.sh r02 human human_snp chr22 linear > gpurun_out/pmc_r02.log 2>&1
echo '{}' > profiles/traffic.json
python tools/pmc_summary.py r02 human human_snp chr22 linear --write-traffic
for M in 16 64 128; do
  Q=100000000; [ $M -ge 64 ] && Q=20000000
  PASSES="rdreq trace" EXTRA="--pattern-len $M --queries $Q" bash tools/pmc_passes.sh t3m$M human > /dev/null 2>&1
  python tools/pmc_summary.py t3m$M human --write-traffic | grep -E "launches|traffic|kernel trace"
  PASSES="rdreq trace" EXTRA="--pattern-len $M --queries 10000000" bash tools/pmc_passes.sh t3m$M linear > /dev/null 2>&1
  python tools/pmc_summary.py t3m$M linear --write-traffic | grep -E "launches|traffic|kernel trace"
done
PASSES="rdreq trace" EXTRA="--set U" bash tools/pmc_passes.sh setU human linear chr22 > /dev/null 2>&1
python tools/pmc_summary.py setU human linear chr22 --set U --write-traffic | grep -E "launches|traffic|kernel trace"
cp profiles/traffic.json gpurun_out/traffic_final.json
for f in gpurun_out/r02_*_trace.log gpurun_out/t3m*_trace.log gpurun_out/setU*_trace.log; do echo $f; grep "^{" $f | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print(c['pattern_len'], c['pattern_set'], c['queries_total'], 'value %.4g q/s' % d['value'], 'kernel_ms %.3f' % r['kernel_ms'], 'frac %.3f' % r['frac'], 'req/q %.2f' % r['request_rate']['requests_per_query'], 'reqrate %.1f' % r['request_rate']['achieved_G_per_s'], 'steps/q %.2f' % c['lf_steps_per_query'], 'blocks/q %.2f' % c['blocks_per_query'], 'k', c['kmer_table_k'], 'found', c['found'], 'closed', c.get('all_ranges_equal_closed_form'))"; done
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --queries 10000000 --steps 2 --warmup 1 --no-cpu --secondary config5"
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --output-format csv -d $ROOT/gpurun_out/c5_rdreq -o x -- $CMD > $ROOT/gpurun_out/c5_rdreq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/c5_trace -o x -- $CMD > $ROOT/gpurun_out/c5_trace.log 2>&1
cd $ROOT
python tools/pmc_kernel.py gpurun_out/c5_rdreq k_parent k_match_stats2 | grep -E "128B|RDREQ_sum"
python tools/pmc_kernel.py gpurun_out/c5_trace k_parent k_match_stats2
