"""The C++ facade (include/gcsa2_hip/gcsa.hpp) keeps the reference's class API; this compiles a
small client against it (CPU) and, on the GPU box, runs it and compares every printed result with
the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_test.cpp")


def compile_client(out):
    import __graft_entry__ as entry
    entry.build()
    libdir = os.path.join(ROOT, "gcsa2_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", out,
           "-L", libdir, "-lgcsa2_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)
    return out


def test_facade_compiles_and_links(tmp_path):
    exe = compile_client(str(tmp_path / "facade_test"))
    assert os.path.exists(exe)


def dump_index(ix, path):
    lv = int(ix.lcp_offsets.shape[0]) - 1
    head = np.array([ix.n, ix.e, ix.order, ix.sigma, ix.fast_chars, ix.sample_count, ix.sample_width,
                     ix.extra_values_len, ix.redundant_len, ix.lcp_size, ix.lcp_branching, lv], dtype=np.uint64)
    blobs = [ix.char2comp, ix.C] + list(ix.bwt) + [ix.edges, ix.sampled_paths, ix.stored_samples, ix.samples,
                                                    ix.extra_filter, ix.extra_values, ix.redundant, ix.lcp_offsets, ix.lcp_data]
    with open(path, "wb") as f:
        f.write(head.tobytes())
        for b in blobs:
            raw = np.ascontiguousarray(b).tobytes()
            f.write(np.uint64(len(raw)).tobytes())
            f.write(raw)


@pytest.mark.gpu
def test_facade_matches_oracle(tmp_path):
    from workload import graphs, builder, patterns
    from oracle.oracle import OracleIndex
    g = graphs.snp_graph(2000, 0x71, 0x72, snp_period=12, node_len=16)
    ix = builder.build(g, 16, sample_period=16, branching=8)
    cpu = OracleIndex(ix)
    pats = [bytes(p[: 2 + q % 10]) for q, p in enumerate(patterns.walk_patterns(g, 40, 12, 0x73))]
    pats += [bytes(p) for p in patterns.uniform_patterns(10, 9, 0x74)]
    dump_index(ix, tmp_path / "index.bin")
    (tmp_path / "patterns.txt").write_text("\n".join(p.decode() for p in pats) + "\n")
    exe = compile_client(str(tmp_path / "facade_test"))
    env = dict(os.environ)
    try:
        import torch
        env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    except Exception:
        pass
    out = subprocess.run([exe, str(tmp_path / "index.bin"), str(tmp_path / "patterns.txt")], capture_output=True,
                         text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split("\n")
    it = iter(lines)
    head = next(it).split()
    assert [int(x) for x in head[1:6]] == [ix.n, ix.e, ix.order, ix.sample_count, ix.sample_width]
    for k in range(7):
        assert next(it) == f"kmers {k} {cpu.count_kmers(k)}"
    ranges = []
    for p in pats:
        want = cpu.find(p)
        assert next(it) == f"find {want[0]} {want[1]}"
        if want[0] <= want[1] < ix.n:
            ranges.append(want)
    assert ranges
    for r in ranges:
        par = cpu.parent(r)
        assert next(it) == "parent " + " ".join(str(x) for x in par)
        assert next(it) == f"depth {cpu.depth((par[0], par[1]))}"
        left = int(ix.lcp_data[r[0]]); right = int(ix.lcp_data[r[1] + 1]) if r[1] + 1 < ix.n else 0
        assert next(it) == f"nodeFor {left} {right}"
        assert next(it) == f"count {cpu.count(r)}"
        assert next(it).split()[1:] == [str(int(v)) for v in cpu.locate(r)]
        assert next(it).split()[1:] == [str(int(v)) for v in cpu.locate(r, sort=False)]
        assert next(it).split()[1:] == [str(int(v)) for v in cpu.locate(r, max_positions=3)]
        fast = cpu.LF_fast(r)
        assert next(it).split()[1:] == [str(x) for c in range(1, ix.fast_chars + 1) for x in fast[c]]
        lf = cpu.LF(r, 1)
        assert next(it) == f"LF {lf[0]} {lf[1]} {cpu.LF(r[0])}"
        assert next(it) == f"sample {int(cpu.sampled(r[0]))} {cpu.firstSample(r[0])} {cpu.sample(0)} {int(cpu.lastSample(0))}"
        assert next(it) == f"sv {cpu.psv(r[0])[0]} {cpu.nsv(r[0])[0]} {cpu.rmq(*r)[0]} {int(ix.lcp_data[r[0]])}"
