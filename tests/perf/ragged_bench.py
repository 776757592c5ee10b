import sys, time, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from workload import graphs, builder, patterns
from gcsa2_amd.binding import open_index
from oracle.oracle import OracleIndex
g = graphs.snp_graph(1 << 23, 0x6C5A0010, 0x6C5A0011)
ix = builder.build(g, 256, keep_table=False)
gpu, lcp = open_index(ix)
nq = 2_000_000
full = patterns.walk_patterns(g, nq, 256, 0x6C5A0060)
lens = 16 + (np.arange(nq) * 2654435761 % 241)          # 16..256, scattered
flat = np.concatenate([full[q, :lens[q]] for q in range(0, nq)]) if False else None
# vectorised ragged concatenation
mask = np.arange(256)[None, :] < lens[:, None]
flat = full[mask]
off = np.zeros(nq + 1, dtype=np.uint64); off[1:] = np.cumsum(lens)
dev = torch.device("cuda", 0)
d_pat = torch.from_numpy(np.ascontiguousarray(flat)).to(dev); d_off = torch.from_numpy(off.view(np.int64)).to(dev)
st = torch.cuda.current_stream()
res = {}
outs = {}
for v in (2, 4, 5):
    d_out = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
    gpu.find_device_variant(v, d_pat.data_ptr(), d_off.data_ptr(), nq, d_out.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        gpu.find_device_variant(v, d_pat.data_ptr(), d_off.data_ptr(), nq, d_out.data_ptr(), st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    res[v] = e0.elapsed_time(e1) / 5
    outs[v] = d_out.cpu().numpy().view(np.uint64)
cpu = OracleIndex(ix, with_samples=False, with_counters=False, with_lcp=False)
want = cpu.find_batch(flat, off[:200001], threads=64)
print(json.dumps({"ragged 16..256-mers, 2 M queries": {"k_find2 ms": res[2], "length-bucketed ms": res[4], "refill ms": res[5]},
                  "equal": bool(np.array_equal(outs[2], outs[4])) and bool(np.array_equal(outs[2], outs[5])),
                  "oracle_sample": bool(np.array_equal(outs[4][:200000], want))}))
# mixed hits and misses, equal lengths: every second 128-mer carries one substitution at a random place
nq2 = 4_000_000
mix = patterns.walk_patterns(g, nq2, 128, 0x6C5A0061)
pos = (np.arange(nq2) * 40503 % 128)
sub = np.frombuffer(b"ACGT", dtype=np.uint8)
rows = np.arange(1, nq2, 2)
mix[rows, pos[rows]] = sub[(np.searchsorted(sub, mix[rows, pos[rows]]) + 1) % 4]
flat2, off2 = patterns.as_batch(mix)
d_pat2 = torch.from_numpy(flat2).to(dev); d_off2 = torch.from_numpy(off2.view(np.int64)).to(dev)
res2, outs2 = {}, {}
for v in (2, 5):
    d_out = torch.zeros((nq2, 2), dtype=torch.int64, device=dev)
    gpu.find_device_variant(v, d_pat2.data_ptr(), d_off2.data_ptr(), nq2, d_out.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        gpu.find_device_variant(v, d_pat2.data_ptr(), d_off2.data_ptr(), nq2, d_out.data_ptr(), st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    res2[v] = e0.elapsed_time(e1) / 5
    outs2[v] = d_out.cpu().numpy().view(np.uint64)
want2 = cpu.find_batch(flat2, off2[:200001], threads=64)
print(json.dumps({"mixed hit/miss 128-mers, 4 M queries": {"k_find2 ms": res2[2], "refill ms": res2[5]},
                  "equal": bool(np.array_equal(outs2[2], outs2[5])), "oracle_sample": bool(np.array_equal(outs2[5][:200000], want2))}))
