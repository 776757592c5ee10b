"""bench.py's contract (task statement / DESIGN.md section 5) on small footprints: the JSON line and its objects, the
strong-sharded multi-rank path through the library's RCCL communicator (world size 1 under torch.distributed.run -- all a
1-GPU box can hold), its torch.distributed fallback, and two ranks sharing the GPU through the host (gloo: control flow)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


LINE_LIMIT = 4096


def strict(text):
    """Strict JSON: NaN / Infinity are refused (json.loads accepts them by default)."""
    def refuse(name):
        raise ValueError(f"non-finite constant {name} in the line")
    return json.loads(text, parse_constant=refuse)


def check_compact(line, full):
    """The stdout line is what the driver parses: under 4 KB, strict JSON, the contract's keys with
    `roofline` and `cpu_baseline`, one small object per secondary; the full objects are in --full-json."""
    assert len(line) < LINE_LIMIT, len(line)
    c = strict(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "secondary"):
        assert key in c, key
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"] and c["metric"] == full["metric"]
    assert "workload" in c["config"] and "model" not in c["config"]
    r = c["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert ("cpu_baseline" in c) == ("cpu_baseline" in full)
    for key, leg in c["secondary"].items():
        assert len(json.dumps(leg)) < 1200, key
    return c


def run(cmd, env=None, timeout=900, rc=0, tmp=None):
    import tempfile
    full_env = dict(os.environ)
    full_env.update(env or {})
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "full.json")
        out = subprocess.run(cmd + ["--full-json", path], cwd=ROOT, env=full_env, capture_output=True, text=True, timeout=timeout)
        assert out.returncode == rc, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        with open(path) as f:
            full = strict(f.read())
    full["_line"] = check_compact(lines[0], full)
    return full


def check_line(d, gpus, steps):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["metric"] == "kmer_find_queries_per_sec" and d["unit"] == "queries/s" and d["n_gpus"] == gpus and d["steps"] == steps
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "u64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["kernel_ms"] * d["steps"] <= d["ms_per_step"] * d["steps"] * 1.02 + 0.5
    assert d["config"]["all_ranges_equal_closed_form"] is True


def test_single_gpu_line_with_secondaries():
    d = run([sys.executable, "bench.py", "--degree", "20", "--queries", "300000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1",
             "--secondary", "config5"])
    check_line(d, 1, 3)
    assert d["scaling"] == "strong" and d["config"]["queries_total"] == 300000
    assert d["config"]["path_nodes"] == ((1 << 20) - 1) // 3 and 1.06 < d["config"]["edges"] / d["config"]["path_nodes"] < 1.10
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["gpu_matches_cpu_on_sample"] is True and c["value"] > 0
    c5 = d["config5"]
    assert c5["unmodified_half_equals_closed_form"] and c5["find_unmodified_half_equals_closed_form"] and c5["locate"]["count_equals_located"]
    assert c5["locate_unmodified_half_equals_closed_form"] and c5["cpu_baseline"]["gpu_matches_cpu_on_sample"]
    big = c5["four_times_the_batch"]
    assert big["patterns"] == 4_000_000 and big["unmodified_half_equals_closed_form"] is True and big["patterns_per_s"] > 0
    assert c5["roofline"]["requests_per_pattern"] > 100 and 0 < c5["roofline"]["request_rate"]["frac_of_ceiling"] < 1.5
    mb = c5["match_breaks"]
    assert mb["unmodified_half_has_one_record_in_closed_form"] and mb["expands_to_the_dense_statistics_on_the_first_patterns"]
    assert mb["final_ranges_and_parent_counts_equal_dense"] and mb["records"] > 1_000_000 and mb["clean_batch"]["breaks_patterns_per_s"] > 0


def test_branching_workload_line():
    d = run([sys.executable, "bench.py", "--workload", "human_snp", "--degree", "20", "--queries", "300000", "--steps", "2", "--warmup", "1", "--no-cpu"])
    check_line(d, 1, 2)
    assert 1.05 < d["config"]["edges"] / d["config"]["path_nodes"] < 1.12


@pytest.mark.parametrize("no_comm", ["", "1"], ids=["library-rccl-gather", "torch-gather-fallback"])
def test_distributed_path_world_size_1(no_comm):
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
             "--master-port", str(free_port()), "bench.py", "--gpus", "1", "--degree", "24", "--queries", "1000001", "--steps", "3",
             "--warmup", "1", "--no-cpu", "--secondary", "config5"], env={"GCSA2_BENCH_NO_COMM": no_comm})
    check_line(d, 1, 3)
    assert ("gcsa2_comm_gather" in d["config"]["parallelism"]) == (no_comm == "")
    assert "6 bytes" in d["config"]["parallelism"]


@pytest.mark.parametrize("wire", ["", "32", "40", "64"], ids=["six-bytes", "u32-pairs", "40-bit-pairs", "u64-pairs"])
def test_two_ranks_share_the_gpu_through_the_host(wire):
    """(the wire format of the gathered ranges: six bytes per range -- 40-bit sp, a length byte, a list for the long ranges -- below
    2^40 path nodes, u64 pairs beyond; GCSA2_BENCH_WIRE forces the u32 pairs, the 40-bit pairs (the fallback when a batch has more
    long ranges than the list holds) and the u64 pairs on this small index)"""
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--degree", "24", "--queries", "1000001", "--steps", "2",
             "--warmup", "1", "--no-cpu"] + (["--secondary", "config5"] if wire == "" else ["--no-secondary"]),
            env={"GCSA2_BENCH_BACKEND": "gloo", "GCSA2_BENCH_WIRE": wire})
    check_line(d, 2, 2)
    assert ("40-bit pairs" in d["config"]["parallelism"]) == (wire == "40") and ("u32 pairs" in d["config"]["parallelism"]) == (wire == "32")
    assert ("6 bytes" in d["config"]["parallelism"]) == (wire == "")
    mg = d["multi_gpu"]
    assert mg["asserted"] is True and mg["gathered_shards_verified"] == 2 and "problems" not in mg
    assert mg["gather_only"]["reps"] >= 2 and mg["root_ingest_GBps"] > 0           # the gather alone, after the timed steps
    if wire != "":
        return
    # one peer: its 500 000 ranges in six bytes each, the count of its list and 500 000 // 64 + 64 entries of 16 bytes
    assert mg["wire_bytes_per_query"] == 6 and mg["bytes_into_root_per_step"] == 6 * 500_000 + 16 + 16 * (500_000 // 64 + 64)
    assert d["scaling"] == "strong" and d["config"]["queries_per_gpu"] == 500001 and d["config"]["queries_total"] == 1000001
    # config 5 sharded over the two ranks: matching statistics and the CSR of located values gathered on the root
    c5 = d["config5"]
    assert c5["n_gpus"] == 2 and c5["unmodified_half_equals_closed_form"] is True
    # both went through the library's C++ (gcsa2_comm_gather / _match_stats / _locate over the host-memory transport of
    # gcsa2_comm_create_custom), not through the Python mirror of the sharding
    assert d["multi_gpu"]["gather"].startswith("gcsa2_comm_gather over a host-memory transport") and d["multi_gpu"]["rccl_ranks"] == 0
    assert "gcsa2_comm_match_stats / gcsa2_comm_locate over a host-memory transport" in c5["workload"]
    assert c5["locate"]["count_equals_located"] is True and c5["locate"]["unmodified_half_equals_closed_form"] is True


def test_eight_ranks_share_the_gpu_with_an_asynchronous_transport():
    """The target world size as far as one GPU allows: eight ranks, the six-byte wire format of the headline
    index (round 6; 40-bit pairs until then), ragged shards (8 k + 3 queries), the library's gcsa2_comm_gather over a transport that only
    ENQUEUES on the gather stream -- so the gather of step k really runs under the kernel of step k + 1 and the root's
    gather calls return before their bytes have moved (with a transport that completes inside the call nothing could overlap).
    Reference shape of the only data-parallel query path: src/algorithms.cpp:106-114 (a static split)."""
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
             "--master-port", str(free_port()), "bench.py", "--gpus", "8", "--degree", "20", "--queries", str(8 * 100_000 + 3), "--steps", "6",
             "--warmup", "2", "--no-cpu", "--no-secondary"], env={"GCSA2_BENCH_BACKEND": "gloo"}, timeout=1500)
    check_line(d, 8, 6)
    assert "6 bytes" in d["config"]["parallelism"] and d["config"]["queries_total"] == 8 * 100_000 + 3
    mg = d["multi_gpu"]
    assert len(mg["per_rank"]) == 8 and sorted(x["queries"] for x in mg["per_rank"]) == [100_000] * 5 + [100_001] * 3
    assert "asynchronous" in mg["gather"] and mg["rccl_ranks"] == 0 and mg["wire_bytes_per_query"] == 6
    assert mg["gathered_shards_verified"] == 8 and mg["asserted"] is True          # the root holds every shard against the checksums its rank computed
    peers = [x["queries"] for x in mg["per_rank"][1:]]
    assert mg["bytes_into_root_per_step"] == sum((6 * c + 15) // 16 * 16 + 16 + 16 * (c // 64 + 64) for c in peers)
    assert mg["gather_only"]["bytes_into_root_per_gather"] == mg["bytes_into_root_per_step"]
    # the overlap: no gather call waits for its bytes -- on the root most calls return before the seven parts have arrived (a
    # transport that completes inside the call returns late every time, and then nothing of gather k can run under kernel
    # k + 1).  `gather_hidden_frac` is reported per rank; with eight processes time-slicing ONE GPU and a gather through host
    # memory that takes 30 x the kernel, the share it can hide is a few per cent and moves from run to run: not asserted.
    root = mg["per_rank"][0]
    assert root["transport_calls"] >= 8 and root["transport_early_returns"] >= root["transport_calls"] // 2, root
    assert all(x["gather_hidden_frac"] is not None and 0.0 <= x["gather_hidden_frac"] <= 1.0 for x in mg["per_rank"])
    assert len(d["_line"]["multi_gpu"]["kernel_ms_per_rank"]) == 8


def test_a_communicator_that_fails_its_probe_is_dropped_by_every_rank():
    """Before anything is timed one small ragged gather of known bytes goes through the library's communicator; a rank that
    sees an error there (GCSA2_BENCH_FAIL_PROBE: rank 1) makes EVERY rank fall back to the other data path, and the line says
    which one ran -- the RCCL transport has never run with peers on the box this is developed on."""
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--degree", "20", "--queries", "200001", "--steps", "2",
             "--warmup", "1", "--no-cpu", "--no-secondary"], env={"GCSA2_BENCH_BACKEND": "gloo", "GCSA2_BENCH_FAIL_PROBE": "1"})
    check_line(d, 2, 2)
    mg = d["multi_gpu"]
    assert mg["gather"].startswith("host copies") and mg["gathered_shards_verified"] == 2


def test_plain_invocation_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's N = 1 command) starts two ranks by
    itself and prints n_gpus: 2 with the per-rank kernel / pack / gather breakdown (reference shape of the
    only data-parallel query path: src/algorithms.cpp:106-114, a static split)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"GCSA2_BENCH_BACKEND": "gloo"})
    d = run([sys.executable, "bench.py", "--gpus", "2", "--degree", "24", "--queries", "1000001", "--steps", "3", "--warmup", "1",
             "--no-cpu", "--no-secondary"], env=env)
    check_line(d, 2, 3)
    assert len(d["_line"]["multi_gpu"]["kernel_ms_per_rank"]) == 2
    mg = d["multi_gpu"]
    assert mg["launched_by"].startswith("bench.py itself") and len(mg["per_rank"]) == 2
    assert [x["rank"] for x in mg["per_rank"]] == [0, 1] and sum(x["queries"] for x in mg["per_rank"]) == 1000001
    for x in mg["per_rank"]:
        assert x["kernel_ms"] > 0 and x["pack_ms"] >= 0 and x["gather_ms"] >= 0 and x["pace_ms"] > 0
        assert x["gather_hidden_frac"] is None or 0.0 <= x["gather_hidden_frac"] <= 1.0


def test_world_mismatch_is_refused():
    """More GPUs asked for than the launcher's world holds, or than the host has: a non-zero exit code, never a line that
    quietly says n_gpus: 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "64", "--degree", "20", "--queries", "1000", "--steps", "1", "--warmup", "0",
                          "--no-cpu", "--no-secondary"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "refusing" in out.stderr
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--degree", "20", "--queries", "1000", "--steps", "1",
                          "--warmup", "0", "--no-cpu", "--no-secondary"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_wide_range_and_ladder_secondaries():
    """The wide-range legs (prefixes of path labels in closed form; the batch with a short seed table), the memory ladder
    (one image re-shaped rung by rung, bit-exact on each) and the repeat-rich text through the prefix-doubling generator, at
    small footprints."""
    d = run([sys.executable, "bench.py", "--degree", "20", "--queries", "300000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1",
             "--secondary", "wide"])
    w = d["wide_ranges"]
    assert len(w) == 4 and all(leg["all_ranges_equal_closed_form"] is True for leg in w.values())
    widths = [w[f"{m}-mers"]["mean_range_width_path_nodes"] for m in (5, 7, 9)]
    assert widths[0] > 250 and widths[0] > widths[1] > widths[2] >= 1.0
    assert w["5-mers"]["second_fetch_fraction_of_steps"] > 0.3
    cut = [leg for name, leg in w.items() if "seed table" in name][0]
    assert cut["kmer_table_k"] == 5 and cut["lf_steps_per_query"] == 27.0
    d = run([sys.executable, "bench.py", "--degree", "20", "--queries", "300000", "--steps", "2", "--warmup", "1", "--no-cpu", "--secondary", "ladder"])
    # the ladder goes down in the order of least find() throughput lost per gigabyte freed: locate table, three seed-table
    # levels with the pair blocks kept, then without pair blocks at three seed-table sizes; the smallest image's
    # rate is carried into the line
    rungs = d["memory_ladder"]["rungs"]
    assert len(rungs) == 8 and all(r["all_ranges_equal_closed_form"] is True and r["locate"]["count_equals_located"] is True for r in rungs)
    assert rungs[0]["locate_table_bytes"] > 0 and all(r["locate_table_bytes"] == 0 for r in rungs[1:])
    assert all(r["pair_block_bytes"] > 0 for r in rungs[:5]) and all(r["pair_block_bytes"] == 0 for r in rungs[5:])
    k = rungs[0]["kmer_table_k"]
    assert [r["kmer_table_k"] for r in rungs] == [k, k, k - 1, k - 2, k - 3, k, k - 1, k - 3]
    for part in (rungs[:5], rungs[5:]):
        assert [r["image_bytes_hbm"] for r in part] == sorted((r["image_bytes_hbm"] for r in part), reverse=True)
    assert rungs[7]["image_bytes_hbm"] == min(r["image_bytes_hbm"] for r in rungs)
    assert rungs[7]["requests_per_query"] > 1.5 * rungs[0]["requests_per_query"]
    ref = d["value_at_reference_footprint"]
    assert ref["value"] == rungs[7]["value"] and abs(d["_line"]["value_at_reference_footprint"]["value"] / ref["value"] - 1) < 1e-4
    d = run([sys.executable, "bench.py", "--degree", "20", "--queries", "300000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1",
             "--secondary", "repeats30"])
    r = d["repeats_hbm"]
    for m in ("32-mers", "16-mers"):
        leg = r[m]
        assert leg["range_width_equals_occurrences_in_text_on_sample"] is True and leg["cpu_baseline"]["gpu_matches_cpu_on_sample"] is True
        assert leg["locate"]["count_equals_located"] is True and leg["mean_range_width_path_nodes"] > 1.5
        assert leg["locate"]["sorted_distinct_every_range"] is True and leg["locate"]["widest_ranges_equal_occurrences_in_text"]["equal"] is True


def test_a_failing_secondary_does_not_cost_the_line():
    """A leg that raises becomes {"error": ...} under its key, in the line and in the full object; the headline, its roofline
    and the legs before and after it are there, and the exit code stays 0."""
    d = run([sys.executable, "bench.py", "--degree", "20", "--queries", "300000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1",
             "--secondary", "config5"], env={"GCSA2_BENCH_FAIL_LEG": "config5"})
    check_line(d, 1, 2)
    assert "GCSA2_BENCH_FAIL_LEG" in d["config5"]["error"] and "error" in d["_line"]["secondary"]["config5"]
    assert d["cpu_baseline"]["gpu_matches_cpu_on_sample"] is True and d["host_batch"]["value"] > 0
    d = run([sys.executable, "bench.py", "--degree", "20", "--queries", "300000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1",
             "--no-secondary"], env={"GCSA2_BENCH_FAIL_LEG": "cpu_baseline"})
    check_line(d, 1, 2)
    assert "error" in d["cpu_baseline"] and "error" in d["_line"]["cpu_baseline"]


def test_a_crash_inside_a_library_call_still_prints_the_line():
    """A secondary that dies of SIGSEGV inside a C call (the interpreter lock is released there): the watcher thread prints the
    line -- headline, roofline, the legs before the crash, the crashed leg as an error -- and the process exits with 128 + 11."""
    d = run([sys.executable, "bench.py", "--degree", "20", "--queries", "300000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1",
             "--secondary", "config5"], env={"GCSA2_BENCH_FAIL_LEG": "crash:config5"}, rc=139)
    check_line(d, 1, 2)
    assert "SIGSEGV" in d["config5"]["error"] and "SIGSEGV in leg config5" in d["errors"][0]
    assert d["cpu_baseline"]["gpu_matches_cpu_on_sample"] is True and "error" in d["_line"]["secondary"]["config5"]
