python -m pytest tests -m gpu -x -q 2>&1 | tail -4
./gcsa2_amd/lib/pool_readback_repro 100000 2>&1 | tail -15
python tests/perf/latency_bench.py 2>&1 | tail -12
