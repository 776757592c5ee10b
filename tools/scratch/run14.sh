python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
python bench.py --workload human_snp --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('human_snp', d['value'], d['ms_per_step'], c['all_ranges_equal_closed_form'], c['blocks_per_query'], r['frac'], r['request_rate'])"
python bench.py --steps 10 --warmup 2 --no-cpu --secondary config5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('human', d['value'], d['ms_per_step'], c['all_ranges_equal_closed_form'], c['blocks_per_query'], c['index_bytes_hbm'], r['frac'], r['request_rate']); print('config5', d['config5']['patterns_per_s'], d['config5']['unmodified_half_equals_closed_form'])"
python bench.py --workload chr22 --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('chr22', d['value'], c['blocks_per_query'], r['frac'], r['request_rate'])"
python bench.py --workload chr22 --set U --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('chr22 U', d['value'], c['blocks_per_query'], c['lf_steps_per_query'], r['request_rate'])"
