#!/usr/bin/env python3
"""Differential campaign on mid-size random graphs (bubbles, indels, cycles, Ns, small alphabets): thousands of path nodes,
so that ranges straddle the 192-position pair blocks and the 384-position single blocks, steps empty in either character of a
pair, and locate() meets every segment-size class.  Every query kind of the engine against the CPU oracle.

    python tests/perf/fuzz_campaign.py [--seeds 40] [--first 0]

Prints one line per graph and a final verdict; exits non-zero at the first difference.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--cpu-only", action="store_true", help="build the indexes and run the oracle only (timing of the harness)")
    args = ap.parse_args()
    from workload import graphs, builder, patterns
    from workload.rng import SplitMix64
    from oracle.oracle import OracleIndex, max_threads
    if not args.cpu_only:
        from gcsa2_amd.binding import open_index
    threads = max_threads()
    t_start = time.time()
    for seed in range(args.first, args.first + args.seeds):
        rng = SplitMix64(0xCA11 + seed)
        n = 1500 + rng.below(30000)
        alphabet = 2 + rng.below(3)
        p_back = (0.0, 0.0, 0.001, 0.004)[rng.below(4)]
        g = graphs.random_graph(n, 0xCA5000 + seed, p_branch=0.03 + 0.01 * rng.below(15), p_back=p_back,
                                p_n=(0.0, 0.01, 0.03)[rng.below(3)], alphabet=alphabet)
        K = (4, 8, 16, 32)[rng.below(4)] if p_back == 0.0 else (6, 8)[rng.below(2)]
        ix = builder.build(g, K, sample_period=(1 << 40 if seed % 7 == 0 else 2 + rng.below(40)), branching=2 + rng.below(63))
        cpu = OracleIndex(ix)
        m = 4 + rng.below(60)
        walks = patterns.walk_patterns(g, 12000, m, 0xCA6000 + seed)
        mutate = np.random.default_rng(seed)
        rows = []
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        for q in range(walks.shape[0]):
            row = walks[q].copy()
            if q % 3 == 1:                                   # substitutions: steps that empty mid-pattern, parent() in matching statistics
                for pos in mutate.integers(0, m, size=1 + q % 4):
                    row[pos] = lut[mutate.integers(0, alphabet)]
            rows.append(bytes(row[: 1 + mutate.integers(0, m)]) if q % 5 == 2 else bytes(row))
        rows += [bytes(p) for p in patterns.uniform_patterns(2000, 3 + rng.below(12), 0xCA7000 + seed)]
        rows += [b"", b"N", b"$", b"#", b"A" * 70]
        from gcsa2_amd.hostview import concat_patterns
        data, off = concat_patterns(rows)
        c_find = cpu.find_batch(data, off, threads=threads)
        nonempty = c_find[(c_find[:, 0] <= c_find[:, 1]) & (c_find[:, 1] < ix.n)]
        line = f"seed {seed}: n={ix.n} e={ix.e} K={K} sigma={alphabet} back={p_back} m={m} patterns={len(rows)} hits={len(nonempty)}"
        if args.cpu_only:
            print(line, flush=True)
            continue
        # round 6: on every second graph the fused split of locate() takes ranges of a few dozen path nodes (its threshold is 8192)
        # and the split aims at small buckets, so that the table-reading split, its LDS tiles and the run phase meet these graphs
        for key in ("GCSA2_LOCATE_FUSE_ABOVE", "GCSA2_SPLIT_TARGET", "GCSA2_SPLIT_SKEW"):
            os.environ.pop(key, None)
        if seed % 2 == 1:
            os.environ["GCSA2_LOCATE_FUSE_ABOVE"] = str((8, 64, 300)[seed % 3])
            os.environ["GCSA2_SPLIT_TARGET"] = str((24, 256, 6)[(seed // 3) % 3])
            if seed % 5 == 0:
                os.environ["GCSA2_SPLIT_SKEW"] = "40"
        gpu, lcp = open_index(ix)
        assert np.array_equal(gpu.find_batch(data, off), c_find), (seed, "find")
        wide = np.array([[a, min(a + w, ix.n - 1)] for a, w in zip(mutate.integers(0, ix.n, size=3000), mutate.integers(0, 3000, size=3000))],
                        dtype=np.uint64)
        for name, arr in (("found", nonempty), ("wide", wide)):
            assert np.array_equal(gpu.count_batch(arr), cpu.count_batch(arr)), (seed, "count", name)
            go, gv = gpu.locate_batch(arr)
            co, cv = cpu.locate_batch(arr)
            assert np.array_equal(go, co) and np.array_equal(gv, cv), (seed, "locate", name)
            assert np.array_equal(lcp.parent_batch(arr), cpu.parent_batch(arr)), (seed, "parent", name)
            assert np.array_equal(lcp.depth_batch(arr), cpu.depth_batch(arr)), (seed, "depth", name)
        comps = mutate.integers(0, ix.sigma, size=len(wide)).astype(np.uint8)
        want = np.array([cpu.LF((int(a), int(b)), int(c)) for (a, b), c in zip(wide[:400], comps[:400])], dtype=np.uint64).reshape(-1, 2)
        assert np.array_equal(gpu.lf_batch(wide[:400], comps[:400]), want), (seed, "lf")
        gm, gr, gf = gpu.match_stats_batch(data, off)
        cm, cr, cf = cpu.match_stats_batch(data, off, threads=threads)
        assert np.array_equal(gm, cm) and np.array_equal(gr, cr) and np.array_equal(gf, cf), (seed, "match_stats")
        # round 4: break points (all, and with a minimum length) against what the dense statistics + find() imply; k-mers as 2-bit
        # codes; locate() into caller-owned buffers; all of it again on a re-shaped image (tables dropped / resized at random)
        import torch
        from test_gpu_parity import breaks_from_dense
        from gcsa2_amd.hostview import pack_kmers
        want_off, want_brk = breaks_from_dense(cpu, rows, cm, off)
        fixed = [r for r in rows if len(r) == m and all(c in b"ACGT" for c in r)]
        fixed_arr = np.frombuffer(b"".join(fixed), dtype=np.uint8).reshape(len(fixed), m) if fixed else None
        dev = torch.device("cuda", 0)
        raw_wide = sum(len(cpu.locate((int(a), int(b)), sort=False)) for a, b in wide)       # values before removeDuplicates
        for shape in (None, (int(mutate.integers(0, 2)), int(mutate.integers(0, 9)), int(mutate.integers(0, 2)))):
            if shape is not None:
                gpu.set_tables(pair_blocks=shape[0], kmer_k=shape[1], locate_table=shape[2])
                assert np.array_equal(gpu.find_batch(data, off), c_find), (seed, "find", shape)
            for min_length in (0, 1 + int(mutate.integers(0, 2 * K))):
                boff, brk, brng, bfb = gpu.match_breaks_batch(data, off, min_length=min_length)
                keep = want_brk[:, 1] >= min_length
                assert np.array_equal(brk, want_brk[keep]), (seed, "match_breaks", shape, min_length)
                assert int(boff[-1]) == int(keep.sum()) and np.array_equal(brng, cr) and np.array_equal(bfb, cf), (seed, "match_breaks tail", shape)
            if fixed_arr is not None:
                fd, fo = concat_patterns(fixed)
                assert np.array_equal(gpu.find_batch_packed(pack_kmers(fixed_arr, ix.char2comp), m), cpu.find_batch(fd, fo, threads=threads)), (seed, "packed", shape)
            co, cv = cpu.locate_batch(wide)
            d_r = torch.from_numpy(wide.view(np.int64).copy()).to(dev)
            d_o = torch.zeros(len(wide) + 1, dtype=torch.int64, device=dev)
            d_v = torch.zeros(len(cv) + 1, dtype=torch.int64, device=dev)
            total = gpu.locate_into(d_r.data_ptr(), len(wide), d_o.data_ptr(), d_v.data_ptr(), d_v.shape[0])
            assert total == len(cv) and np.array_equal(d_o.cpu().numpy().view(np.uint64), co) and np.array_equal(d_v.cpu().numpy().view(np.uint64)[:total], cv), (seed, "locate_into", shape)
            # round 6: a buffer with room for the values BEFORE deduplication is the sorts' work space and is compacted in place
            raw = raw_wide
            d_w = torch.full((raw + 3 + 32,), -1, dtype=torch.int64, device=dev)
            total = gpu.locate_into(d_r.data_ptr(), len(wide), d_o.data_ptr(), d_w.data_ptr(), raw + 3)
            assert total == len(cv) and np.array_equal(d_o.cpu().numpy().view(np.uint64), co) and np.array_equal(d_w.cpu().numpy().view(np.uint64)[:total], cv), (seed, "locate_into in place", shape)
            assert bool((d_w[raw + 3:] == -1).all()), (seed, "locate_into wrote behind the capacity", shape)
        # round 5: every launch shape of both matching-statistics kernels (2 / 5: k_match_stats2, 6 / 7: k_match_stats3) on the device
        # buffers, and locate() of batches of one-node ranges (the one-kernel path when every node has one value, the pipeline otherwise)
        total_bytes = int(off[-1])
        d_pat = torch.zeros(total_bytes + 16, dtype=torch.uint8, device=dev)
        d_pat[:total_bytes] = torch.from_numpy(np.ascontiguousarray(data[:total_bytes]).copy()).to(dev)
        d_off = torch.from_numpy(off.view(np.int64).copy()).to(dev)
        for variant in (2, 5):
            d_ms = torch.full((total_bytes + 8,), -3, dtype=torch.int16, device=dev)
            d_rng = torch.zeros((len(rows), 2), dtype=torch.int64, device=dev)
            d_fb = torch.zeros(len(rows), dtype=torch.int64, device=dev)
            gpu.match_stats_device(d_pat.data_ptr(), d_off.data_ptr(), len(rows), d_ms.data_ptr(), d_rng.data_ptr(), d_fb.data_ptr(), 0, variant=variant,
                                   total_bytes=total_bytes)
            torch.cuda.synchronize()
            assert np.array_equal(d_ms[:total_bytes].cpu().numpy().view(np.uint16), cm), (seed, "match_stats variant", variant)
            assert np.array_equal(d_rng.cpu().numpy().view(np.uint64), cr) and np.array_equal(d_fb.cpu().numpy().view(np.uint64), cf), (seed, "variant tail", variant)
        # round 6: one-query calls (the resident wavefront of kernels_mailbox.hpp answers them)
        for (a, b), c in zip(wide[:60], comps[:60]):
            r = (int(a), int(b))
            assert gpu.LF(r, int(c)) == cpu.LF(r, int(c)), (seed, "scalar LF", r, int(c))
            assert gpu.count(r) == cpu.count(r) and lcp.parent(r) == cpu.parent(r), (seed, "scalar count / parent", r)
        nodes = mutate.integers(0, ix.n, size=4000).astype(np.uint64)
        singles = np.stack([nodes, nodes], axis=1)
        so, sv = cpu.locate_batch(singles)
        go, gv = gpu.locate_batch(singles)
        assert np.array_equal(go, so) and np.array_equal(gv, sv), (seed, "locate one-node ranges")
        ones = singles[np.diff(so) == 1]
        if len(ones):
            oo, ov = cpu.locate_batch(ones)
            go, gv = gpu.locate_batch(ones)
            assert np.array_equal(go, oo) and np.array_equal(gv, ov), (seed, "locate one-value ranges")
        print(line + f"  breaks={len(want_brk)} packed={len(fixed)} one-value={len(ones)}  ok", flush=True)
        gpu.close()
    print(f"campaign of {args.seeds} graphs: no difference ({time.time() - t_start:.0f} s)")


if __name__ == "__main__":
    main()
