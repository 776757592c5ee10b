"""The stdout line of bench.py, checked without a GPU: `compact_line` turns the full result object of a run (the committed
`profiles/r05_bench_full.json`, and a variant with a failed leg, non-finite numbers and eight ranks) into what the driver parses
-- under 4 KB, strict JSON, the contract's keys with `roofline` and `cpu_baseline`, one small object per secondary (VERDICT r04
#1: round 4's 23 KB line came back as `parsed: null`).  The metric mirrors the find() phase of the reference's benchmark
(benchmark/query_gcsa.cpp:87-100)."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def strict(text):
    def refuse(name):
        raise ValueError(f"non-finite constant {name}")
    return json.loads(text, parse_constant=refuse)


def load_full():
    with open(os.path.join(ROOT, "profiles", "r05_bench_full.json")) as f:
        return json.load(f)


def check(line, full):
    import bench
    assert len(line) < bench.LINE_LIMIT and "\n" not in line
    c = strict(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "secondary"):
        assert key in c, key
    assert c["metric"] == "kmer_find_queries_per_sec" and c["unit"] == "queries/s" and c["vs_baseline"] is None and c["dtype"] == "u64"
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"]
    assert "workload" in c["config"] and "model" not in c["config"]
    r = c["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["served"] in ("HBM", "L2 / Infinity Cache (partly)")
    return c


def test_the_committed_run_gives_a_line_the_driver_can_parse():
    import bench
    full = load_full()
    c = check(bench.compact_line(bench.finite(full)), full)
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] >= 1 and c["cpu_baseline"]["value"] > 0
    ref = c["value_at_reference_footprint"]
    assert ref["image_GB"] < 40 and 0 < ref["value"] < c["value"] * 1.01
    sec = c["secondary"]
    assert sec["config5"]["patterns_per_s"] > 0 and sec["config5"]["match_stats_frac_of_request_ceiling"] is not None
    assert len(sec["memory_ladder"]["image_GB"]) == len(sec["memory_ladder"]["G_queries_per_s"]) == 8
    assert sec["chr22"]["locate_values_per_s"] > 0
    # no leg claims an HBM fraction above one under the HBM label
    def legs(o):
        if isinstance(o, dict):
            if "served" in o and "frac" in o:
                yield o
            for v in o.values():
                yield from legs(v)
    seen = list(legs(full))
    assert len(seen) >= 10
    for leg in seen:
        if leg["served"] == "HBM":
            assert leg["frac"] <= 1.0, leg.get("workload", leg.get("kernel"))
        elif leg["frac"] > 1.0:
            assert "frac_is" in leg or "working_set_note" in leg


def test_failed_legs_non_finite_numbers_and_eight_ranks_still_fit():
    import bench
    full = load_full()
    bad = copy.deepcopy(full)
    bad["config5"] = {"error": "RuntimeError: " + "x" * 1000}
    bad["repeats"] = {"error": "hipErrorOutOfMemory"}
    bad["errors"] = ["SIGTERM before the last leg finished"]
    bad["roofline"]["traffic"] = float("nan")
    bad["host_batch"]["value"] = float("inf")
    bad["n_gpus"] = 8
    bad["multi_gpu"] = {"backend": "nccl", "gather": "gcsa2_comm_gather (library RCCL communicator)", "wire_bytes_per_query": 10,
                        "bytes_into_root_per_step": 875000000, "rccl_ranks": 8, "slowest_kernel_ms": 2.7, "root_gather_ms": 1.9, "root_gather_hidden_frac": 0.93,
                        "note": "n" * 500,
                        "per_rank": [dict(rank=r, device=r, queries=12500000, kernel_ms=2.6 + r / 100, pack_ms=0.06, gather_ms=1.8, pace_ms=2.7,
                                          wall_ms_per_step=2.7, gather_hidden_frac=0.9, wire_bytes_sent=125000000, rccl_ranks=8) for r in range(8)]}
    c = check(bench.compact_line(bench.finite(bad)), bad)
    assert "error" in c["secondary"]["config5"] and len(c["secondary"]["config5"]["error"]) <= 120
    assert c["roofline"]["traffic"] is None and c["secondary"]["host_batch"]["queries_per_s"] is None
    assert len(c["multi_gpu"]["kernel_ms_per_rank"]) == 8 and c["multi_gpu"]["rccl_ranks"] == 8
    # a pathological object: the guard drops secondaries rather than exceed the limit
    huge = copy.deepcopy(full)
    huge["wide_ranges"] = {f"{k}-mers of a very long leg name number {k}": {"value": 1.0 * k, "served": "HBM"} for k in range(200)}
    line = bench.compact_line(bench.finite(huge))
    assert len(line) < bench.LINE_LIMIT and strict(line)["secondary_dropped_for_size"] is True


def test_locate_counter_summary_adds_up(tmp_path):
    """tools/pmc_locate_summary.py: bytes per locate() call from the two counter passes (reads by request size, writes as 64- and
    32-byte requests), summed over the dispatches of the engine's locate kernels and divided by the calls of the bench leg."""
    import csv
    import subprocess
    import sys
    header = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]
    rows_rd = [[1, "(anonymous namespace)::k_over_split(unsigned long const*)", "TCC_EA0_RDREQ_128B_sum", 1000],
               [1, "(anonymous namespace)::k_over_split(unsigned long const*)", "TCC_EA0_RDREQ_64B_sum", 10],
               [1, "(anonymous namespace)::k_over_split(unsigned long const*)", "TCC_EA0_RDREQ_32B_sum", 4],
               [1, "(anonymous namespace)::k_over_split(unsigned long const*)", "TCC_EA0_RDREQ_sum", 1014],
               [2, "void (anonymous namespace)::k_sort_big<4096u, 0u>(unsigned long const*)", "TCC_EA0_RDREQ_128B_sum", 500],
               [2, "void (anonymous namespace)::k_sort_big<4096u, 0u>(unsigned long const*)", "TCC_EA0_RDREQ_sum", 500]]
    rows_wr = [[1, "(anonymous namespace)::k_over_split(unsigned long const*)", "TCC_EA0_WRREQ_sum", 300],
               [1, "(anonymous namespace)::k_over_split(unsigned long const*)", "TCC_EA0_WRREQ_64B_sum", 200]]
    for suffix, rows in (("_rdreq", rows_rd), ("_wrreq", rows_wr)):
        d = tmp_path / ("x_locate" + suffix) / "host"
        d.mkdir(parents=True)
        with open(d / "x_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(header)
            w.writerows(rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_locate_summary.py"), str(tmp_path / "x_locate"), "5"],
                         capture_output=True, text=True, check=True)
    d = json.loads(out.stdout)
    assert d["read_bytes_per_call"] == (128 * 1500 + 64 * 10 + 32 * 4) / 5 and d["write_bytes_per_call"] == (64 * 200 + 32 * 100) / 5
    assert d["read_requests_per_call"] == 1514 / 5 and d["write_requests_per_call"] == 60 and "k_over_split" in out.stderr
