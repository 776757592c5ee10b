import sys, time
sys.path.insert(0, "/root/repo")
from workload import graphs, builder
from gcsa2_amd.binding import open_index
from oracle.oracle import OracleIndex, max_threads
g = graphs.snp_graph(1 << 22, 0x6C5A0010, 0x6C5A0011)
ix = builder.build(g, 256, keep_table=False)
gpu, lcp = open_index(ix)
cpu = OracleIndex(ix)
for k in (8, 12, 16, 24):
    t = time.time(); a = gpu.count_kmers(k); tg = time.time() - t
    t = time.time(); b = cpu.count_kmers(k, threads=max_threads()); tc = time.time() - t
    print(f"k={k}: gpu {a} in {tg*1e3:.1f} ms, cpu({max_threads()} threads) {b} in {tc*1e3:.1f} ms, equal={a==b}")
