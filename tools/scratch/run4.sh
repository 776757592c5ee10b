python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -5
python tests/perf/latency_bench.py 2>&1 | tail -7
