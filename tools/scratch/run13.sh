python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k branching 2>&1 | grep -E "passed|failed|rror" | tail -3
python bench.py --workload human_snp --steps 5 --warmup 2 --no-cpu 2>gpurun_out/snp32.err > gpurun_out/snp32.json; grep -E "SNP|edges" gpurun_out/snp32.err; python -c "
import json; d=json.load(open('gpurun_out/snp32.json')); r=d['roofline']; c=d['config']; print(d['value'], d['ms_per_step'], c['all_ranges_equal_closed_form'], c['edges']/c['path_nodes'], c['blocks_per_query'], c['lf_steps_per_query'], r['frac'], r['request_rate'])"
