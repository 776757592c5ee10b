python bench.py --workload human_snp --degree 24 --queries 2000000 --steps 3 --warmup 1 --no-cpu 2>gpurun_out/snp24.err | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['all_ranges_equal_closed_form'], d['config']['edges']/d['config']['path_nodes'], d['config']['blocks_per_query'])"
tail -3 gpurun_out/snp24.err
python bench.py --workload human_snp --steps 5 --warmup 2 --no-cpu 2>gpurun_out/snp32.err > gpurun_out/snp32.json; tail -12 gpurun_out/snp32.err; python -c "
import json; d=json.load(open('gpurun_out/snp32.json')); r=d['roofline']; c=d['config']; print(d['value'], d['ms_per_step'], c['all_ranges_equal_closed_form'], c['edges']/c['path_nodes'], c['blocks_per_query'], c['lf_steps_per_query'], r['frac'], r['request_rate'])"
