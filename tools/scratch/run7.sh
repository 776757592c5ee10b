# two ranks on one GPU through the host (gloo): control flow of the strong-sharded bench only
GCSA2_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --degree 20 --queries 1000003 --steps 3 --warmup 1 --no-cpu 2>gpurun_out/two_rank.err | tail -1 | cut -c1-900
tail -3 gpurun_out/two_rank.err
for M in 16 64 128; do
  Q=100000000; [ $M -ge 64 ] && Q=20000000
  PASSES="rdreq trace" EXTRA="--pattern-len $M --queries $Q" bash tools/pmc_passes.sh t3m$M human > /dev/null 2>&1
  python tools/pmc_summary.py t3m$M human --write-traffic | grep -E "launches|traffic|kernel trace"
  Q=10000000
  PASSES="rdreq trace" EXTRA="--pattern-len $M --queries $Q" bash tools/pmc_passes.sh t3m$M linear > /dev/null 2>&1
  python tools/pmc_summary.py t3m$M linear --write-traffic | grep -E "launches|traffic|kernel trace"
done
PASSES="rdreq trace" EXTRA="--set U" bash tools/pmc_passes.sh setU human linear > /dev/null 2>&1
python tools/pmc_summary.py setU human linear --set U --write-traffic | grep -E "launches|traffic|kernel trace"
cp profiles/traffic.json gpurun_out/traffic_sweep.json
for f in gpurun_out/t3m*_trace.log gpurun_out/setU*_trace.log; do echo $f; grep "^{" $f | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print(c['pattern_len'], c['pattern_set'], c['queries_total'], 'value %.4g q/s' % d['value'], 'kernel_ms %.3f' % r['kernel_ms'], 'frac %.3f' % r['frac'], 'req/q %.2f' % r['request_rate']['requests_per_query'], 'reqrate %.1f' % r['request_rate']['achieved_G_per_s'], 'steps/q %.2f' % c['lf_steps_per_query'], 'found', c['found'], 'closed', c['all_ranges_equal_closed_form'])"; done
