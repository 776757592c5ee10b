"""Committed golden vectors (tests/golden/snp3000.json, made by tests/golden/make_golden.py):
the oracle must keep reproducing them (CPU), and the HIP engine must match them (GPU) without
consulting the live oracle."""
import importlib.util
import json
import os

import numpy as np
import pytest

from gcsa2_amd.hostview import concat_patterns

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "snp3000.json")) as f:
        gold = json.load(f)
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.SPEC == gold["spec"]
    g, ix, plist = mod.make_inputs(gold["spec"])
    assert [p.decode("latin1") for p in plist] == gold["patterns"]
    assert (ix.n, ix.e, ix.sample_count) == (gold["path_nodes"], gold["edges"], gold["samples"])
    return gold, ix, plist


def check(gold, find_batch, count_batch, locate_batch, parent_batch, depth_batch, n, plist):
    flat, off = concat_patterns(plist)
    ranges = find_batch(flat, off)
    assert ranges.tolist() == gold["find"]
    assert count_batch(ranges).tolist() == gold["count"]
    loff, lval = locate_batch(ranges)
    assert loff.tolist() == gold["locate_offsets"] and lval.tolist() == gold["locate_values"]
    nonempty = ranges[(ranges[:, 0] <= ranges[:, 1]) & (ranges[:, 1] < n)]
    parents = parent_batch(nonempty)
    assert [[int(x) for x in row] for row in parents.tolist()] == gold["parent"]
    pr = np.stack([parents["sp"], parents["ep"]], axis=1)
    assert depth_batch(pr).tolist() == gold["parent_depth"]


def test_oracle_reproduces_golden(golden):
    from oracle.oracle import OracleIndex
    gold, ix, plist = golden
    cpu = OracleIndex(ix)
    check(gold, cpu.find_batch, cpu.count_batch, cpu.locate_batch, cpu.parent_batch, cpu.depth_batch, ix.n, plist)


@pytest.mark.gpu
def test_engine_matches_golden(golden):
    from gcsa2_amd.binding import open_index
    gold, ix, plist = golden
    gpu, lcp = open_index(ix)
    check(gold, gpu.find_batch, gpu.count_batch, gpu.locate_batch, lcp.parent_batch, lcp.depth_batch, ix.n, plist)
