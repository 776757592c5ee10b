#!/bin/bash
# rocprofv3 passes of bench.py for the roofline traffic figure (run on the GPU box, from the repo root):
#   tools/pmc_passes.sh <tag>        e.g. tools/pmc_passes.sh p5
# One counter group per pass (separate --pmc runs, no tracing domains besides the kernel trace of the
# last pass), outputs under gpurun_out/<tag>_<workload>_<group>/; summarise with tools/pmc_summary.py.
set -u
TAG=${1:-p}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for WL in snp linear; do
  CMD="python $ROOT/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --no-hbm-resident --no-jump-table"
  rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum \
            --output-format csv -d $ROOT/gpurun_out/${TAG}_${WL}_rdreq -o x -- $CMD > $ROOT/gpurun_out/${TAG}_${WL}_rdreq.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $ROOT/gpurun_out/${TAG}_${WL}_l2 -o x -- $CMD > $ROOT/gpurun_out/${TAG}_${WL}_l2.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/${TAG}_${WL}_fetch -o x -- $CMD > $ROOT/gpurun_out/${TAG}_${WL}_fetch.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_trace -o x -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-jump-table > $ROOT/gpurun_out/${TAG}_trace.log 2>&1
echo done
