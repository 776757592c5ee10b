#!/usr/bin/env python3
"""Generates tests/golden/snp3000.json: inputs (seeds) and expected outputs of every query type on
a small seeded SNP graph, computed by the CPU oracle (oracle/gcsa_oracle.c), after the oracle has
been pinned by tests/test_oracle.py.  The reference cannot be run in this image (SDSL absent), so
these vectors are produced by our restatement; the paper-derived vectors are in paper_example.json.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from workload import graphs, builder, patterns  # noqa: E402
from oracle.oracle import OracleIndex  # noqa: E402

SPEC = dict(bases=3000, seq_seed=0x6C5A0101, snp_seed=0x6C5A0102, snp_period=16, node_len=16,
            order=16, sample_period=16, branching=8, walk_seed=0x6C5A0103, uniform_seed=0x6C5A0104,
            n_walk=150, n_uniform=100, pattern_len=12)


def make_inputs(spec=SPEC):
    g = graphs.snp_graph(spec["bases"], spec["seq_seed"], spec["snp_seed"], snp_period=spec["snp_period"],
                         node_len=spec["node_len"])
    ix = builder.build(g, spec["order"], sample_period=spec["sample_period"], branching=spec["branching"])
    pats = np.concatenate([patterns.walk_patterns(g, spec["n_walk"], spec["pattern_len"], spec["walk_seed"]),
                           patterns.uniform_patterns(spec["n_uniform"], spec["pattern_len"], spec["uniform_seed"])])
    # ragged lengths: pattern q keeps its first 1 + q % pattern_len characters; plus specials
    plist = [bytes(p[: 1 + (q % spec["pattern_len"])]) for q, p in enumerate(pats)]
    plist += [b"", b"N", b"$", b"#", b"acgtn", bytes(pats[0]) + bytes(pats[1])]
    return g, ix, plist


def main():
    from gcsa2_amd.hostview import concat_patterns
    g, ix, plist = make_inputs()
    cpu = OracleIndex(ix)
    flat, off = concat_patterns(plist)
    ranges = cpu.find_batch(flat, off)
    counts = cpu.count_batch(ranges)
    loff, lval = cpu.locate_batch(ranges)
    nonempty = ranges[(ranges[:, 0] <= ranges[:, 1]) & (ranges[:, 1] < ix.n)]
    parents = cpu.parent_batch(nonempty)
    depths = cpu.depth_batch(np.stack([parents["sp"], parents["ep"]], axis=1))
    out = {
        "_generator": "tests/golden/make_golden.py (CPU oracle; reference not runnable here)",
        "spec": SPEC, "path_nodes": int(ix.n), "edges": int(ix.e), "samples": int(ix.sample_count),
        "patterns": [p.decode("latin1") for p in plist],
        "find": ranges.tolist(), "count": counts.tolist(),
        "locate_offsets": loff.tolist(), "locate_values": lval.tolist(),
        "parent": [[int(x) for x in row] for row in parents.tolist()],
        "parent_depth": depths.tolist(),
    }
    path = os.path.join(ROOT, "tests", "golden", "snp3000.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes;", len(plist), "patterns;", int(loff[-1]), "located values")


if __name__ == "__main__":
    main()
