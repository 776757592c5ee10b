"""The C++ facade (include/gcsa2_hip/gcsa.hpp) keeps the reference's class API; this compiles a
small client against it (CPU) and, on the GPU box, runs it and compares every printed result with
the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_test.cpp")


def compile_client(out, src=SRC):
    import __graft_entry__ as entry
    entry.build()
    libdir = os.path.join(ROOT, "gcsa2_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", out,
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
    # round 4's additions through the facade: packed k-mers, break points (against the oracle's dense statistics + find()), the ladder
    from gcsa2_amd.hostview import concat_patterns
    from test_gpu_parity import breaks_from_dense
    packed = next(it).split()
    assert packed[0] == "packed" and int(packed[1]) > 10 and packed[2] == "same"
    data, off = concat_patterns(pats)
    cm, cr, cf = cpu.match_stats_batch(data, off, threads=2)
    want_off, want = breaks_from_dense(cpu, pats, cm, off)
    for min_length in (0, 3):
        keep = want[:, 1] >= min_length
        line = next(it)
        head, _, rest = line.partition(" |")
        assert head.split() == ["breaks", str(min_length), str(int(keep.sum()))]
        groups = [g.split() for g in (" |" + rest).split(" |")[1:]]
        assert len(groups) == len(pats)
        for q, g in enumerate(groups):
            rows = want[int(want_off[q]):int(want_off[q + 1])]
            rows = rows[rows[:, 1] >= min_length]
            assert g == [":".join(str(int(x)) for x in r) for r in rows], (q, pats[q])
    ladder = next(it).split()
    assert ladder[:3] == ["ladder", "same", "ok"] and int(ladder[3]) == cpu.find(pats[0])[0]


def _run_env():
    env = dict(os.environ)
    try:
        import torch
        env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    except Exception:
        pass
    return env


def test_query_gcsa_cli_builds():
    from gcsa2_amd import build
    exe = build.build_query_gcsa()
    assert os.access(exe, os.X_OK)
    out = subprocess.run([exe], env=_run_env(), capture_output=True, text=True)
    assert out.returncode == 0 and "usage: query_gcsa base_name [patterns]" in out.stderr      # query_gcsa.cpp:37-43


@pytest.mark.gpu
def test_query_gcsa_cli_matches_oracle(tmp_path):
    """`query_gcsa base_name patterns` on .gcsa / .lcp files: the totals of every phase
    (reference benchmark/query_gcsa.cpp:87-169) equal the oracle's."""
    import re
    from gcsa2_amd import build
    from workload import graphs, builder, patterns, sdsl_format
    from oracle.oracle import OracleIndex
    g = graphs.snp_graph(3000, 0x81, 0x82, snp_period=12, node_len=16)
    ix = builder.build(g, 16, sample_period=16, branching=8)
    cpu = OracleIndex(ix)
    pats = [bytes(p[: 3 + q % 12]) for q, p in enumerate(patterns.walk_patterns(g, 300, 14, 0x83))]
    pats += [bytes(p) for p in patterns.uniform_patterns(60, 7, 0x84)]
    rows = [p.decode() for p in pats]
    rows[5:5] = ["", "NNNN", "N"]                       # skipped / filtered rows (query_gcsa.cpp:80-85,186-204)
    base = str(tmp_path / "index")
    sdsl_format.write(ix, base)
    (tmp_path / "patterns.txt").write_text("\n".join(rows) + "\n")
    exe = build.build_query_gcsa()

    stats = subprocess.run([exe, base], env=_run_env(), capture_output=True, text=True, timeout=300)
    assert stats.returncode == 0, stats.stderr
    assert re.search(rf"Paths:\s+{ix.n}\n", stats.stdout) and re.search(rf"Edges:\s+{ix.e}\n", stats.stdout)
    assert re.search(rf"Max query:\s+{ix.order}\n", stats.stdout)

    run = subprocess.run([exe, base, str(tmp_path / "patterns.txt")], env=_run_env(), capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    text = run.stdout
    ranges = [cpu.find(p) for p in pats]
    found = [(r, len(p)) for r, p in zip(ranges, pats) if r[0] <= r[1]]
    matching = sum(((r[1] + 1 - r[0]) % (1 << 64)) for r in ranges) % (1 << 64)
    assert re.search(rf"Patterns:\s+{len(pats)} \(total", text)
    assert f"Found {len(found)} patterns matching {matching} paths" in text
    parents = [cpu.parent(r) for r, _ in found]
    pdist = sum(m - node[4] for (_, m), node in zip(found, parents)) / len(found)
    ddist = sum(m - cpu.depth((node[0], node[1])) for (_, m), node in zip(found, parents)) / len(found)
    got = [float(x) for x in re.findall(r"Average distance ([0-9.e+-]+) characters", text)]
    assert len(got) == 2 and abs(got[0] - pdist) < 1e-4 * max(1.0, pdist) and abs(got[1] - ddist) < 1e-4 * max(1.0, ddist)
    occurrences = sum(cpu.count(r) for r, _ in found)
    located = sum(len(cpu.locate(r)) for r, _ in found)
    assert re.search(rf"count\(\):\s+{occurrences} occurrences", text)
    assert re.search(rf"locate\(\):\s+{located} occurrences", text)
    assert "inconsistent" not in text

    missing = subprocess.run([exe, str(tmp_path / "nothing")], env=_run_env(), capture_output=True, text=True)
    assert missing.returncode != 0 and "Cannot load the index" in missing.stderr        # query_gcsa.cpp:55-59


def test_count_kmers_cli_builds():
    from gcsa2_amd import build
    exe = build.build_count_kmers()
    out = subprocess.run([exe], env=_run_env(), capture_output=True, text=True)
    assert out.returncode == 0 and "usage: count_kmers [options] base_name [base_name2]" in out.stderr     # count_kmers.cpp:45-59


@pytest.mark.gpu
def test_count_kmers_cli_matches_oracle(tmp_path):
    """`count_kmers -k K base` and `count_kmers -k K -o X left right` on .gcsa files
    (reference benchmark/count_kmers.cpp): counts and the dumped symmetric difference equal the oracle's."""
    import re
    from gcsa2_amd import build
    from workload import graphs, sdsl_format
    from workload import builder
    from oracle.oracle import OracleIndex
    g1 = graphs.snp_graph(1500, 0x91, 0x92, snp_period=9, node_len=16)
    g2 = graphs.snp_graph(1500, 0x91, 0x99, snp_period=7, node_len=16)
    i1, i2 = builder.build(g1, 16), builder.build(g2, 16)
    sdsl_format.write(i1, str(tmp_path / "left"))
    sdsl_format.write(i2, str(tmp_path / "right"))
    c1, c2 = OracleIndex(i1), OracleIndex(i2)
    exe = build.build_count_kmers()
    for k, flags in ((8, []), (11, ["-N"]), (20, ["-f"]), (20, [])):
        run = subprocess.run([exe, "-k", str(k)] + flags + [str(tmp_path / "left")], env=_run_env(), capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, run.stderr
        want = c1.count_kmers(k, include_Ns="-N" in flags, force="-f" in flags)
        assert re.search(rf"Kmers:\s+{want}\n", run.stdout), (k, flags, run.stdout)
        assert re.search(rf"GCSA:\s+{i1.n} paths, order 16\n", run.stdout)
    out = str(tmp_path / "diff")
    run = subprocess.run([exe, "-k", "9", "-o", out, str(tmp_path / "left"), str(tmp_path / "right")], env=_run_env(),
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    counts, left, right = c1.compare_kmers_records(c2, 9)
    assert re.search(rf"Shared:\s+{counts[0]} kmers\nLeft:\s+{counts[1]} unique kmers\nRight:\s+{counts[2]} unique kmers", run.stdout)
    for path, want in ((out + ".left", left), (out + ".right", right)):
        got = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
        assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, want.tolist()))
    bad = subprocess.run([exe, "-k", "9", str(tmp_path / "nothing")], env=_run_env(), capture_output=True, text=True)
    assert bad.returncode != 0 and "Cannot load the index" in bad.stderr


REF_CLIENT = os.path.join(ROOT, "tests", "cpp", "ref_api_client.cpp")


def test_reference_api_client_compiles_and_links(tmp_path):
    """A program that includes <gcsa/gcsa.h>, <gcsa/lcp.h>, <gcsa/algorithms.h> and uses the reference's names only
    (tests/cpp/ref_api_client.cpp) builds against include/ and libgcsa2_hip.so; it mentions nothing of the engine."""
    text = open(REF_CLIENT).read()
    code = "\n".join(line.split("//")[0] for line in text.split("\n") if not line.startswith("#define GCSA2_HIP_SDSL_IO"))
    assert "gcsa2_" not in code and "retainHostView" in code        # one engine knob, for serialize()
    exe = compile_client(str(tmp_path / "ref_api_client"), REF_CLIENT)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_reference_api_client_matches_oracle(tmp_path):
    """The same program run on the GPU against .gcsa / .lcp files: header, alphabet, find (all three overloads and the
    charRange + LF loop), parent / depth / nodeFor, count, locate, countKMers, copies, serialize() byte-identical to the
    files it loaded, load() errors, two structures in one stream -- every line compared with the oracle."""
    from workload import graphs, builder, patterns, sdsl_format
    from oracle.oracle import OracleIndex
    g = graphs.snp_graph(3000, 0x81, 0x82, snp_period=10, node_len=16)
    ix = builder.build(g, 16, sample_period=16, branching=8)
    cpu = OracleIndex(ix)
    pats = [bytes(p[: 3 + q % 12]) for q, p in enumerate(patterns.walk_patterns(g, 60, 16, 0x83))]
    pats += [bytes(p) for p in patterns.uniform_patterns(12, 9, 0x84)] + [b"ACGTN", b"A"]
    base = str(tmp_path / "index")
    sdsl_format.write(ix, base)
    (tmp_path / "patterns.txt").write_text("\n".join(p.decode() for p in pats) + "\n")
    exe = compile_client(str(tmp_path / "ref_api_client"), REF_CLIENT)
    out = subprocess.run([exe, base, str(tmp_path / "patterns.txt")], capture_output=True, text=True, env=_run_env(), timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    it = iter(out.stdout.strip().split("\n"))
    assert next(it) == f"header 1 3 {ix.n} {ix.e} {ix.order} 0"
    assert next(it) == f"index {ix.n} {ix.e} {ix.order} {ix.sample_count} {ix.sample_width} 0"
    assert next(it) == f"alpha {ix.sigma} {ix.fast_chars} $ACGTN# " + " ".join(str(int(c)) for c in ix.C)
    assert next(it) == f"lcp 1 {ix.lcp_size} {int(ix.lcp_offsets[-1])} {len(ix.lcp_offsets) - 1} {ix.lcp_branching}"
    located = 0
    for p in pats:
        sp, ep = cpu.find(p)
        length = (ep + 1 - sp) % (1 << 64)
        assert next(it) == f"find {sp} {ep} 1 1 {length} {int(sp > ep)}", p
        if sp > ep or ep >= ix.n:
            continue
        par = cpu.parent((sp, ep))
        assert next(it) == f"parent {par[0]} {par[1]} {par[4]} {cpu.depth((par[0], par[1]))} 1"
        vals = cpu.locate((sp, ep))
        text = " ".join(f"{int(v) >> 11}:{'-' if (int(v) >> 10) & 1 else ''}{int(v) & 1023}" for v in vals)
        assert next(it) == (f"locate {cpu.count((sp, ep))} {len(vals)} " + text).rstrip()
        located += len(vals)
    assert located > 0
    assert next(it) == f"kmers {cpu.count_kmers(3)}"
    first = cpu.find(pats[0])
    assert next(it) == f"copies 0 {ix.n} 1 0 0"          # `copy` was swapped away: an empty index answers (0, size() - 1)
    assert next(it) == "serialize 1 1 1 1"
    assert next(it) == "gcsa error 1"
    assert next(it) == "lcp error 1"
    assert next(it) == f"stream {ix.n} {ix.lcp_size} 1"
    assert first[0] <= first[1]


# ---- verifyIndex() over a k-mer array (include/gcsa/algorithms.h; reference src/algorithms.cpp:101-295) --------------------

VERIFY_SRC = os.path.join(ROOT, "tests", "cpp", "verify_client.cpp")


def graph_kmers(g, k):
    """Every (k-mer label, start node) of the input graph, the reference's verification input: labels are cut after the
    first endmarker and padded with '$' to k characters, as the k-mer extraction leaves them (paper figure: key `$$$`)."""
    from workload.brute_builder import k_labels
    rows = set()
    for v, labels in enumerate(k_labels(g, k)):
        for label in labels:
            text = "".join("$ACGTN#"[c] for c in label)
            if "$" in text:
                text = text[: text.index("$") + 1].ljust(k, "$")
            rows.add((text, int(g.value[v])))
    return sorted(rows)


def test_verify_client_compiles(tmp_path):
    assert os.path.exists(compile_client(str(tmp_path / "verify_client"), VERIFY_SRC))


@pytest.mark.gpu
def test_verify_index_through_the_reference_api(tmp_path):
    """The reference's own test of this path: every distinct k-mer of the input graph is searched, located, counted and --
    with the LCP array -- its parent() compared with re-searching shorter prefixes.  Complete on a correct index; fails,
    with the reference's messages, when the k-mer array and the index disagree."""
    from workload import graphs, builder, sdsl_format
    g = graphs.snp_graph(1500, 0x91, 0x92, snp_period=10, node_len=16)
    ix = builder.build(g, 16, sample_period=8, branching=4)
    base = str(tmp_path / "index")
    sdsl_format.write(ix, base)
    k = 8
    rows = graph_kmers(g, k)
    assert len(rows) > 1500 and any(r[0].endswith("$$") for r in rows) and any(r[0].startswith("#") for r in rows)
    (tmp_path / "kmers.txt").write_text("".join(f"{label} {value}\n" for label, value in rows))
    exe = compile_client(str(tmp_path / "verify_client"), VERIFY_SRC)

    def run(*mode):
        return subprocess.run([exe, base, str(tmp_path / "kmers.txt"), str(k), *mode], env=_run_env(), capture_output=True, text=True,
                              timeout=600)
    good = run()
    assert good.returncode == 0, good.stderr + good.stdout
    distinct = len({label for label, value in rows})
    assert f"Queried the index with {distinct} patterns in " in good.stdout
    assert "Index verification complete" in good.stdout and "result 1" in good.stdout and "verifyIndex()" not in good.stderr
    plain = run("nolcp")
    assert plain.returncode == 0 and "Index verification complete" in plain.stdout
    dropped = run("drop")
    assert dropped.returncode == 3 and "Index verification failed for 1 patterns" in dropped.stdout
    assert "verifyIndex(): count(" in dropped.stderr and " occurrences, got " in dropped.stderr
    altered = run("alter")
    assert altered.returncode == 3 and "Index verification failed for 1 patterns" in altered.stdout
    assert "verifyIndex(): locate(" in altered.stderr and "failed: Expected " in altered.stderr
