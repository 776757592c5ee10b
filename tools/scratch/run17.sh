python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo rc=$?; tail -3 gpurun_out/r02_bench.err
