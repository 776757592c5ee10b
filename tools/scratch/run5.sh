python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "locate or segment or fuzz or config2_reduced or paper" 2>&1 | grep -E "passed|failed|rror" | tail -5
python tests/perf/locate_bench.py 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02_locate_trace -o x -- python /root/repo/tests/perf/locate_bench.py --lengths 10,12 > /root/repo/gpurun_out/r02_locate_trace.log 2>&1; head -14 /root/repo/gpurun_out/r02_locate_trace/x_kernel_stats.csv | cut -c1-200
