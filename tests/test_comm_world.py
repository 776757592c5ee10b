"""gcsa2_comm_gather / gcsa2_comm_match_stats / gcsa2_comm_locate with MORE THAN ONE RANK, on one GPU: two or three processes
share device 0 (RCCL refuses that, `Duplicate GPU detected`), so the communicator is made over the application-transport
entry point gcsa2_comm_create_custom with a gather through host memory (gcsa2_amd/host_transport.py over gloo).  Everything
above the transport is the library's C++ exactly as under RCCL: contiguous shards, ragged and empty ones; the three gathers of
the matching statistics; locate()'s totals -> offsets -> values exchange, the CSR rebasing on the root, both sort modes; and the
protocol for a rank whose own part fails.  The root compares with the single-process engine on the whole batch and with the CPU
oracle (reference shape: the static split of verifyIndex, src/algorithms.cpp:106-114)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _bounds(n, world):
    base, extra = divmod(n, world)
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < extra else 0)
        out.append((b, e))
        b = e
    return out


def _worker(rank, world, port, nq, fail_rank, out_path, transport_kind="blocking"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from workload import graphs
    from workload.brute_builder import build
    from workload.rng import SplitMix64
    from gcsa2_amd import binding
    from gcsa2_amd.hostview import concat_patterns
    from gcsa2_amd.host_transport import HostGather, AsyncHostGather
    from oracle.oracle import OracleIndex
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ix = build(graphs.snp_graph(300, 0x52, 0x53, snp_period=8, node_len=8), 8, sample_period=8, branching=4)
    gpu, lcp = binding.open_index(ix)
    transport = (AsyncHostGather if transport_kind == "async" else HostGather)(dist, rank, world)
    comm = binding.Comm.custom(rank, world, 0, transport)
    assert comm.rccl_ranks() == 0                     # not an RCCL communicator
    rng = SplitMix64(11 + nq)
    g = graphs.snp_graph(300, 0x52, 0x53, snp_period=8, node_len=8)
    from test_oracle import random_patterns
    walks = [p for m in (3, 9, 17, 30) for p in random_patterns(g, m, 0x60 + m, 40)]
    pats = []
    for q in range(nq):                               # walks through the graph (hits), some with a substitution, random strings, empties
        kind = rng.below(4)
        if kind < 2 and walks:
            p = bytearray(walks[rng.below(len(walks))])
            if kind == 1 and len(p) > 0:
                p[rng.below(len(p))] = b"ACGTN"[rng.below(5)]
            pats.append(bytes(p))
        else:
            pats.append("".join("ACGT"[rng.below(4)] for _ in range(rng.below(12))).encode())
    flat, off = concat_patterns(pats)
    bounds = _bounds(nq, world)
    b, e = bounds[rank]
    counts = [hi - lo for lo, hi in bounds]
    pbytes = [int(off[hi] - off[lo]) for lo, hi in bounds]
    st = torch.cuda.current_stream().cuda_stream
    # this rank's shard on the device (offsets rebased to the shard)
    sub = np.ascontiguousarray(flat[int(off[b]):int(off[e])])
    d_pat = torch.zeros(sub.shape[0] + 16, dtype=torch.uint8, device=dev)
    d_pat[: sub.shape[0]] = torch.from_numpy(sub.copy()).to(dev)
    d_off = torch.from_numpy((off[b:e + 1] - off[b]).astype(np.int64)).to(dev)
    root = 0
    is_root = rank == root
    total = int(off[-1])
    # 1. find() on the shard + the plain gather of ranges
    d_mine = torch.zeros((max(e - b, 1), 2), dtype=torch.int64, device=dev)
    gpu.find_device(d_pat.data_ptr(), d_off.data_ptr(), e - b, d_mine.data_ptr(), st)
    d_all = torch.zeros((max(nq, 1), 2), dtype=torch.int64, device=dev) if is_root else None
    comm.gather(d_mine.data_ptr(), [16 * c for c in counts], d_all.data_ptr() if is_root else 0, root, st)
    # 2. matching statistics, gathered
    d_ms = torch.zeros(total + 8, dtype=torch.int16, device=dev) if is_root else None
    d_rng = torch.zeros((max(nq, 1), 2), dtype=torch.int64, device=dev) if is_root else None
    d_fb = torch.zeros(max(nq, 1), dtype=torch.int64, device=dev) if is_root else None
    comm.match_stats(gpu, d_pat.data_ptr(), d_off.data_ptr(), counts, pbytes, d_ms.data_ptr() if is_root else 0,
                     d_rng.data_ptr() if is_root else 0, d_fb.data_ptr() if is_root else 0, root, st)
    torch.cuda.synchronize()
    # 3. locate() of the shard's non-empty ranges, CSR gathered; every rank derives the same global list from the oracle
    cpu = OracleIndex(ix)
    found = cpu.find_batch(flat, off)
    keep = (found[:, 0] <= found[:, 1]) & (found[:, 1] < ix.n)
    ranges = np.ascontiguousarray(found[keep])
    nr = ranges.shape[0]
    rb = _bounds(nr, world)
    lo, hi = rb[rank]
    rcounts = [y - x for x, y in rb]
    mine = np.ascontiguousarray(ranges[lo:hi])
    # a rank whose own part fails: its image has no samples, so gcsa2_locate_device refuses (MISSING_COMPONENT) on that rank only
    loc_gpu = binding.GCSA(ix, device=0, with_samples=False, with_counters=False) if rank == fail_rank else gpu
    d_r = torch.zeros((max(hi - lo, 1), 2), dtype=torch.int64, device=dev)
    if hi > lo:
        d_r[: hi - lo] = torch.from_numpy(mine.view(np.int64)).to(dev)
    results = {}
    for sort in (True, False):
        d_loff = torch.zeros(nr + 1, dtype=torch.int64, device=dev) if is_root else None
        try:
            res = comm.locate(loc_gpu, d_r.data_ptr(), rcounts, d_loff.data_ptr() if is_root else 0, root, st, sort=sort)
            err = None
        except binding.Gcsa2Error as ex:
            res, err = None, str(ex)
        if fail_rank >= 0:
            # the failing rank reports its own error, the root names the rank, the healthy peers finish: nobody hangs
            if rank == fail_rank:
                assert err is not None and "MISSING_COMPONENT" in err, (rank, err)
            elif is_root:
                assert err is not None and f"rank {fail_rank}" in err, (rank, err)
            else:
                assert err is None, err
            continue
        assert err is None, err
        if is_root:
            job, d_val, tot = res
            vals = binding.fetch_job(job, tot)
            results[sort] = (d_loff.cpu().numpy().view(np.uint64), vals)
    if is_root:
        # against the single-process engine on the whole batch ...
        got_find = gpu.find_batch(flat, off)
        assert np.array_equal(d_all[:nq].cpu().numpy().view(np.uint64), got_find)
        gm, gr, gf = gpu.match_stats_batch(flat, off)
        assert np.array_equal(d_ms[:total].cpu().numpy().view(np.uint16), gm)
        assert np.array_equal(d_rng[:nq].cpu().numpy().view(np.uint64), gr)
        assert np.array_equal(d_fb[:nq].cpu().numpy().view(np.uint64), gf)
        # ... and the oracle
        assert np.array_equal(got_find, found)
        cm, cr, cf = cpu.match_stats_batch(flat, off)
        assert np.array_equal(gm, cm) and np.array_equal(gr, cr) and np.array_equal(gf, cf)
        if fail_rank < 0:
            co, cv = cpu.locate_batch(ranges)
            assert np.array_equal(results[True][0], co) and np.array_equal(results[True][1], cv)
            assert np.array_equal(np.diff(results[True][0]), cpu.count_batch(ranges))      # benchmark/query_gcsa.cpp:171-179
            uo, uv = results[False]                   # path order, duplicates kept: per range the same multiset at least
            assert uo.shape == co.shape and int(uo[-1]) >= int(co[-1])
            for q in range(0, nr, max(1, nr // 50)):
                assert np.array_equal(np.unique(uv[int(uo[q]):int(uo[q + 1])]), cv[int(co[q]):int(co[q + 1])]), q
        assert transport.calls >= 5
        open(out_path, "w").write("ok")
    if fail_rank >= 0:
        # the communicator is still usable after the failed calls: one more gather
        d_one = torch.full((2,), rank + 1, dtype=torch.int64, device=dev)
        d_got = torch.zeros(2 * world, dtype=torch.int64, device=dev) if is_root else None
        comm.gather(d_one.data_ptr(), [16] * world, d_got.data_ptr() if is_root else 0, root, st)
        torch.cuda.synchronize()
        if is_root:
            assert d_got.cpu().tolist() == [x for r in range(world) for x in (r + 1, r + 1)]
    comm.close()
    if hasattr(transport, "close"):
        transport.close()
    if loc_gpu is not gpu:
        loc_gpu.close()
    gpu.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nq,fail_rank,transport", [(2, 301, -1, "blocking"), (3, 50, -1, "blocking"), (3, 2, -1, "blocking"), (2, 1, -1, "blocking"),
                                                          (3, 120, 1, "blocking"), (2, 90, 0, "blocking"),
                                                          # the target world size, ragged shards (8 k + 3) and fewer queries than ranks, over the
                                                          # transport that only enqueues (gcsa2_amd/host_transport.py::AsyncHostGather); a failing rank
                                                          (8, 8 * 37 + 3, -1, "async"), (8, 5, -1, "async"), (3, 50, -1, "async"), (8, 8 * 11 + 3, 5, "async")])
def test_comm_entry_points_beyond_one_rank(tmp_path, world, nq, fail_rank, transport):
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = tmp_path / "ok.txt"
    mp.spawn(_worker, args=(world, _free_port(), nq, fail_rank, str(out), transport), nprocs=world, join=True)
    assert out.read_text() == "ok"
