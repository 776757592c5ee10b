mkdir -p gpurun_out/r05zy
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r05zy/pytest_all.log 2>&1; grep -E "passed|failed" gpurun_out/r05zy/pytest_all.log; grep -n "^E " gpurun_out/r05zy/pytest_all.log | head -10
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 --full-json gpurun_out/r05zy/bench_full.json > gpurun_out/r05zy/bench_line.json 2> gpurun_out/r05zy/bench_stderr.log ) 2>&1 | tail -3; wc -c gpurun_out/r05zy/bench_line.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o x -- python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary --no-extras --full-json /tmp/f.json > /dev/null 2> gpurun_out/r05zy/trace.log
find /tmp/tr -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05zy/pangenome_kernel_stats.csv \;
grep "k_find2<false" gpurun_out/r05zy/pangenome_kernel_stats.csv | cut -c1-60,190-260
