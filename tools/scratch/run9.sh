ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --queries 10000000 --steps 2 --warmup 1 --no-cpu --secondary config5"
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --output-format csv -d $ROOT/gpurun_out/c5_rdreq -o x -- $CMD > $ROOT/gpurun_out/c5_rdreq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/c5_trace -o x -- $CMD > $ROOT/gpurun_out/c5_trace.log 2>&1
cd $ROOT
python tools/pmc_kernel.py gpurun_out/c5_rdreq k_parent k_match_stats2 k_locate_tab k_count
python tools/pmc_kernel.py gpurun_out/c5_trace k_parent k_match_stats2 k_locate_tab k_count
grep "^{" gpurun_out/c5_trace.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['config5']))"
