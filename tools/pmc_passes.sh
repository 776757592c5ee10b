#!/bin/bash
# rocprofv3 passes of bench.py for the roofline traffic figure (run on the GPU box, from the repo root):
#   tools/pmc_passes.sh <tag> [workload ...]     e.g. tools/pmc_passes.sh r02 human chr22 linear
# One counter group per pass (separate --pmc runs; no tracing domain besides the kernel trace of the
# last pass), outputs under gpurun_out/<tag>_<workload>_<group>/; summarise with tools/pmc_summary.py.
# EXTRA="--set U" (or any bench.py flags) applies to every pass.
set -u
TAG=${1:-p}
shift
WORKLOADS=${*:-human}
EXTRA=${EXTRA:-}
PASSES=${PASSES:-rdreq l2 fetch trace}      # e.g. PASSES="rdreq trace" for sweeps
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for WL in $WORKLOADS; do
  CMD="python $ROOT/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --no-secondary --no-extras $EXTRA"
  ONLY="--kernel-include-regex k_find2"      # counters for the find kernels only (collecting them for every torch kernel of the generator is slow, and crashed once)
  for P in $PASSES; do
    case $P in
      rdreq) timeout ${PASS_TIMEOUT:-420} rocprofv3 $ONLY --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum \
                --output-format csv -d $ROOT/gpurun_out/${TAG}_${WL}_rdreq -o x -- $CMD > $ROOT/gpurun_out/${TAG}_${WL}_rdreq.log 2>&1 ;;
      l2)    timeout ${PASS_TIMEOUT:-420} rocprofv3 $ONLY --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $ROOT/gpurun_out/${TAG}_${WL}_l2 -o x -- $CMD > $ROOT/gpurun_out/${TAG}_${WL}_l2.log 2>&1 ;;
      fetch) timeout ${PASS_TIMEOUT:-420} rocprofv3 $ONLY --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/${TAG}_${WL}_fetch -o x -- $CMD > $ROOT/gpurun_out/${TAG}_${WL}_fetch.log 2>&1 ;;
      trace) timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_${WL}_trace -o x -- \
                python $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu --no-secondary --no-extras $EXTRA > $ROOT/gpurun_out/${TAG}_${WL}_trace.log 2>&1 ;;
    esac
  done
done
echo done
