#!/bin/bash
# rocprofv3 counter passes of a locate() leg of bench.py (run on the GPU box, from the repo root):
#   tools/pmc_locate.sh <tag> <bench.py flags ...>     e.g. tools/pmc_locate.sh r05 --workload repeats30 --pattern-len 16 --locate-ranges 100000
# Two separate --pmc runs (memory-side read requests by size; write requests), counters for the engine's locate kernels only,
# outputs under gpurun_out/<tag>_locate_{rdreq,wrreq}/; summarise with tools/pmc_locate_summary.py.
set -u
TAG=$1
shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py $* --locate --steps 2 --warmup 1 --no-cpu --no-secondary --no-extras --full-json /tmp/pmc_locate_full.json"
ONLY="--kernel-include-regex k_locate_|k_over_|k_sort_|k_compact|k_mark_|k_dedup_huge|k_collect_multi|k_block_owners|k_word_counts|k_final_offsets|k_huge_to_over|k_classify_fused"
timeout ${PASS_TIMEOUT:-600} rocprofv3 $ONLY --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum \
    --output-format csv -d $ROOT/gpurun_out/${TAG}_locate_rdreq -o x -- $CMD > $ROOT/gpurun_out/${TAG}_locate_rdreq.log 2>&1
timeout ${PASS_TIMEOUT:-600} rocprofv3 $ONLY --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum \
    --output-format csv -d $ROOT/gpurun_out/${TAG}_locate_wrreq -o x -- $CMD > $ROOT/gpurun_out/${TAG}_locate_wrreq.log 2>&1
echo done
