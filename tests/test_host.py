"""CPU tests of the host side: C-ABI surface, loud failure without a device, query sharding +
gather over gloo (world_size 2)."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as entry
    entry.build()
    from gcsa2_amd import binding
    return binding


def declared_functions():
    text = open(os.path.join(ROOT, "include", "gcsa2_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gcsa2_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built):
    names = declared_functions()
    assert len(names) >= 30
    lib = ctypes.CDLL(built.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/gcsa2_hip.h but not exported"
    assert sorted(built.EXPORTS) == names, "binding.EXPORTS out of sync with the header"


def test_fails_loudly_without_device(built):
    from workload import graphs
    from workload.brute_builder import build
    if built.load_library().gcsa2_device_count() > 0:
        pytest.skip("a GPU is visible here")
    ix = build(graphs.paper_graph(), 3)
    with pytest.raises(built.Gcsa2Error) as e:
        built.GCSA(ix)
    assert e.value.code == -2   # GCSA2_ERR_NO_DEVICE: no silent CPU path


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing in the product (gcsa2_amd/, include/, tools/) mentions it,
    and in bench.py only the cpu_baseline leg does."""
    for top in ("gcsa2_amd", "include", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".sh")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "oracle" not in text.lower().replace("no cpu fallback", ""), f"{top}/{f} mentions the oracle"
    bench = open(os.path.join(ROOT, "bench.py")).read()
    head, _, tail = bench.partition("def cpu_baseline(")
    assert "from oracle" not in head and "import oracle" not in head
    assert "from oracle" in tail          # the cpu_baseline leg is where the oracle is timed


def test_shard_bounds():
    from gcsa2_amd.shard import shard_bounds, slice_batch
    assert shard_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    flat = np.frombuffer(b"AACCCGT", dtype=np.uint8)
    off = np.array([0, 2, 5, 6, 7], dtype=np.uint64)
    f, o = slice_batch(flat, off, 1, 3)
    assert f.tobytes() == b"CCCG" and o.tolist() == [0, 3, 4]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, nq, out_path):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from workload import graphs
    from workload.brute_builder import build
    from workload.rng import SplitMix64
    from gcsa2_amd.hostview import concat_patterns
    from gcsa2_amd.shard import find_sharded
    from oracle.oracle import OracleIndex
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ix = build(graphs.snp_graph(120, 0x52, 0x53, snp_period=8, node_len=8), 6, sample_period=8, branching=4)
    cpu = OracleIndex(ix)   # stands in for the GPU engine: this test covers sharding + gather only
    rng = SplitMix64(7)
    pats = ["".join("ACGT"[rng.below(4)] for _ in range(1 + rng.below(7))).encode() for _ in range(nq)]
    flat, off = concat_patterns(pats)

    def compute(sub_flat, sub_off):
        return torch.from_numpy(cpu.find_batch(sub_flat, sub_off).view(np.int64))

    res = find_sharded(compute, flat, off)
    if rank == 0:
        want = cpu.find_batch(flat, off)
        assert np.array_equal(res.numpy().view(np.uint64), want)
        open(out_path, "w").write("ok")
    else:
        assert res is None
    dist.destroy_process_group()


@pytest.mark.parametrize("nq", [101, 1])
def test_sharded_find_gloo_world2(tmp_path, nq):
    import torch.multiprocessing as mp
    out = tmp_path / "ok.txt"
    mp.spawn(_worker, args=(2, _free_port(), nq, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def test_host_view_file_round_trip(built, tmp_path):
    """G2HV container: save -> load -> identical arrays (host only, no device)."""
    import ctypes as C
    from workload import graphs
    from workload.brute_builder import build
    ix = build(graphs.snp_graph(150, 0x52, 0x53, snp_period=8, node_len=8), 6, sample_period=8, branching=4)
    path = str(tmp_path / "index.g2hv")
    built.save_host_view(ix, path)
    loaded = built.LoadedHostView(path)
    v = loaded.view
    assert (v.path_nodes, v.edges, v.order, v.sigma, v.fast_chars) == (ix.n, ix.e, ix.order, ix.sigma, ix.fast_chars)
    assert (v.sample_count, v.sample_width, v.extra_values_len, v.redundant_len) == \
        (ix.sample_count, ix.sample_width, ix.extra_values_len, ix.redundant_len)
    words = (ix.n + 63) // 64
    for c in range(ix.sigma):
        assert np.ctypeslib.as_array(v.bwt[c], shape=(words,)).tolist() == ix.bwt[c][:words].tolist()
    assert np.ctypeslib.as_array(v.edge_bits, shape=((ix.e + 63) // 64,)).tolist() == ix.edges[: (ix.e + 63) // 64].tolist()
    assert np.ctypeslib.as_array(v.C, shape=(ix.sigma + 1,)).tolist() == ix.C.tolist()
    assert np.ctypeslib.as_array(v.char2comp, shape=(256,)).tolist() == ix.char2comp.tolist()
    nvals = int(ix.lcp_offsets[-1])
    assert np.ctypeslib.as_array(v.lcp_data, shape=(nvals,)).tolist() == ix.lcp_data.tolist()
    sw = (ix.sample_count * ix.sample_width + 63) // 64
    assert np.ctypeslib.as_array(v.stored_samples, shape=(sw,)).tolist() == ix.stored_samples[:sw].tolist()
    loaded.close()
    # a file with a bad tag is refused, like GCSA::load does for an invalid header
    bad = tmp_path / "bad.g2hv"
    bad.write_bytes(b"XXXX" + open(path, "rb").read()[4:])
    with pytest.raises(built.Gcsa2Error):
        built.LoadedHostView(str(bad))
    with pytest.raises(built.Gcsa2Error):
        built.LoadedHostView(str(tmp_path / "missing.g2hv"))
    # view without LCP / counters
    built.save_host_view(ix, path, with_lcp=False, with_counters=False)
    v2 = built.LoadedHostView(path).view
    assert not v2.lcp_data and not v2.extra_filter_bits and bool(v2.sampled_path_bits)


def _view_matches(v, ix):
    def words(ptr, nbits):
        n = (nbits + 63) // 64
        return np.ctypeslib.as_array(ptr, shape=(n,)).tolist() if n else []

    def plain(arr, nbits):
        n = (nbits + 63) // 64
        return np.asarray(arr, dtype=np.uint64)[:n].tolist()

    assert (v.path_nodes, v.edges, v.order, v.sigma, v.fast_chars) == (ix.n, ix.e, ix.order, ix.sigma, ix.fast_chars)
    assert (v.sample_count, v.sample_width, v.extra_values_len, v.redundant_len) == \
        (ix.sample_count, ix.sample_width, ix.extra_values_len, ix.redundant_len)
    for c in range(ix.sigma):
        assert words(v.bwt[c], ix.n) == plain(ix.bwt[c], ix.n), c
    assert words(v.edge_bits, ix.e) == plain(ix.edges, ix.e)
    assert words(v.sampled_path_bits, ix.n) == plain(ix.sampled_paths, ix.n)
    assert words(v.stored_samples, ix.sample_count * ix.sample_width) == plain(ix.stored_samples, ix.sample_count * ix.sample_width)
    assert words(v.sample_bits, ix.sample_count) == plain(ix.samples, ix.sample_count)
    assert words(v.extra_filter_bits, ix.n) == plain(ix.extra_filter, ix.n)
    assert words(v.extra_values_bits, ix.extra_values_len) == plain(ix.extra_values, ix.extra_values_len)
    assert words(v.redundant_bits, ix.redundant_len) == plain(ix.redundant, ix.redundant_len)
    assert np.ctypeslib.as_array(v.C, shape=(ix.sigma + 1,)).tolist() == ix.C.tolist()
    assert np.ctypeslib.as_array(v.char2comp, shape=(256,)).tolist() == ix.char2comp.tolist()


def test_gcsa_file_loader(built, tmp_path):
    """`.gcsa` / `.lcp` byte streams (GCSA::serialize order, SDSL container encodings as restated in
    workload/sdsl_format.py) -> gcsa2_host_view_load_gcsa -> the same members.  Host only."""
    from workload import graphs, sdsl_format
    from workload import builder
    from workload.brute_builder import build
    cases = [build(graphs.paper_graph(), 3, sample_period=2, branching=2),
             build(graphs.snp_graph(150, 0x52, 0x53, snp_period=8, node_len=8), 6, sample_period=8, branching=4),
             builder.build(graphs.snp_graph(40000, 0x61, 0x62), 16)]       # > 4096 ones per select directory, several LCP levels
    for k, ix in enumerate(cases):
        gcsa_path, lcp_path = sdsl_format.write(ix, str(tmp_path / f"case{k}"))
        loaded = built.LoadedHostView(gcsa_path, lcp_path)
        v = loaded.view
        _view_matches(v, ix)
        assert (v.lcp_size, v.lcp_branching, v.lcp_levels) == (ix.lcp_size, ix.lcp_branching, len(ix.lcp_offsets) - 1)
        assert np.ctypeslib.as_array(v.lcp_offsets, shape=(len(ix.lcp_offsets),)).tolist() == ix.lcp_offsets.tolist()
        nvals = int(ix.lcp_offsets[-1])
        assert np.ctypeslib.as_array(v.lcp_data, shape=(nvals,)).tolist() == ix.lcp_data.tolist()
        loaded.close()
        without = built.LoadedHostView(gcsa_path)
        assert not without.view.lcp_data and bool(without.view.redundant_bits)
        without.close()

    # refusals: wrong tag / version (GCSA::load "Invalid header", src/gcsa.cpp:188-193), truncation, trailing bytes
    raw = open(gcsa_path, "rb").read()
    for name, data in (("tag", b"\0\0\0\0" + raw[4:]), ("version", raw[:4] + b"\x02\0\0\0" + raw[8:]),
                       ("truncated", raw[: len(raw) // 2]), ("trailing", raw + b"\0" * 8), ("empty", b"")):
        bad = tmp_path / f"bad_{name}.gcsa"
        bad.write_bytes(data)
        with pytest.raises(built.Gcsa2Error):
            built.LoadedHostView(str(bad))
    flipped = bytearray(raw)
    flipped[len(raw) // 3] ^= 0x10          # a payload or count word somewhere inside the BWT vectors
    bad = tmp_path / "bad_flip.gcsa"
    bad.write_bytes(bytes(flipped))
    try:
        corrupted = built.LoadedHostView(str(bad))
    except built.Gcsa2Error:
        corrupted = None
    if corrupted is not None:             # a flipped low bit of an sd_vector can still decode; it must then differ
        with pytest.raises(AssertionError):
            _view_matches(corrupted.view, ix)
    with pytest.raises(built.Gcsa2Error):
        built.LoadedHostView(gcsa_path, str(tmp_path / "missing.lcp"))
    lcp_raw = open(lcp_path, "rb").read()
    (tmp_path / "bad.lcp").write_bytes(lcp_raw[:-8])
    with pytest.raises(built.Gcsa2Error):
        built.LoadedHostView(gcsa_path, str(tmp_path / "bad.lcp"))


def test_sdsl_select_directory_shapes(built, tmp_path):
    """The reader must step over both kinds of select_support_mcl superblocks: a sparse `redundant`
    vector forces "long" blocks, a dense one "mini" blocks (loader checks exact end of file)."""
    import copy
    from workload import graphs, sdsl_format
    from workload.brute_builder import build
    ix = copy.copy(build(graphs.snp_graph(150, 0x52, 0x53, snp_period=8, node_len=8), 6, sample_period=8, branching=4))
    nbits = 1 << 23
    pos = np.unique((np.arange(9000, dtype=np.uint64) * np.uint64(911)) % np.uint64(nbits))
    dense = np.arange(5000, dtype=np.uint64) + np.uint64(nbits - 6000)
    bits = np.zeros(nbits, dtype=np.uint8)
    bits[pos.astype(np.int64)] = 1
    bits[dense.astype(np.int64)] = 1
    ix.redundant = np.packbits(bits, bitorder="little").view(np.uint64)
    ix.redundant_len = nbits
    raw = sdsl_format.select_support_mcl(ix.redundant, nbits, 1)
    gcsa_path, _ = sdsl_format.write(ix, str(tmp_path / "shapes"))
    loaded = built.LoadedHostView(gcsa_path)
    v = loaded.view
    assert v.redundant_len == nbits
    assert np.ctypeslib.as_array(v.redundant_bits, shape=(nbits // 64,)).tolist() == ix.redundant.tolist()
    args = int.from_bytes(raw[:8], "little")
    sb = (args + 4095) // 4096
    sb_bits = int.from_bytes(raw[8:16], "little")
    kinds = raw[8 + 9 + 8 * ((sb_bits + 63) // 64):]           # after arg_cnt and the superblock vector: mini_or_long
    assert int.from_bytes(kinds[:8], "little") == sb and kinds[8] not in (0, (1 << sb) - 1)      # both kinds present
    loaded.close()
