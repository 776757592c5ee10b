python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "segment" 2>&1 | grep -E "passed|failed|rror" | tail -3
echo "--- wave-per-segment sort (default)"; python tests/perf/locate_bench.py --lengths 8,9,10,11 2>&1 | tail -4
echo "--- GCSA2_SORT_MEDIUM=0 (segmented radix sort for 17..1024 values)"; GCSA2_SORT_MEDIUM=0 python tests/perf/locate_bench.py --lengths 8,9,10,11 2>&1 | tail -4
