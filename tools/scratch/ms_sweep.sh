python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "match_stats or fuzz or very_long or suffix_tree" 2>&1 | tail -3
for B in 1 4 12 24 48; do
  GCSA2_PARENT_BATCH=$B python bench.py --queries 10000000 --steps 2 --warmup 1 --no-cpu --secondary config5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config5']; print('batch $B', c['match_stats_ms'], c['patterns_per_s'], c['unmodified_half_equals_closed_form'], c['parent_queries_per_s'])"
done
GCSA2_MATCH_STATS=1 python bench.py --queries 10000000 --steps 2 --warmup 1 --no-cpu --secondary config5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config5']; print('gen1', c['match_stats_ms'], c['patterns_per_s'], c['unmodified_half_equals_closed_form'])"
