GCSA2_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --degree 28 --queries 4000001 --steps 3 --warmup 1 --no-cpu 2>gpurun_out/two_rank.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['scaling'], d['value'], d['config']['all_ranges_equal_closed_form'], d['config']['parallelism'])"
grep -v "^\[bench" gpurun_out/two_rank.err | grep -i -E "error|Traceback" | head -5
