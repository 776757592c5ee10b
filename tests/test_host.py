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


def _worker_config5(rank, world, port, nq, out_path):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from workload import graphs
    from workload.brute_builder import build
    from workload.rng import SplitMix64
    from gcsa2_amd.hostview import concat_patterns
    from gcsa2_amd.shard import locate_sharded, match_stats_sharded
    from oracle.oracle import OracleIndex
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ix = build(graphs.snp_graph(120, 0x52, 0x53, snp_period=8, node_len=8), 6, sample_period=8, branching=4)
    cpu = OracleIndex(ix)   # stands in for the GPU engine: this test covers the sharding and the CSR gather only
    rng = SplitMix64(11)
    pats = ["".join("ACGTN"[rng.below(5)] for _ in range(rng.below(9))).encode() for _ in range(nq)]
    flat, off = concat_patterns(pats)
    res = match_stats_sharded(lambda f, o: cpu.match_stats_batch(f, o), flat, off)
    ranges = cpu.find_batch(flat, off)
    ranges = ranges[(ranges[:, 0] <= ranges[:, 1]) & (ranges[:, 1] < ix.n)]
    loc = locate_sharded(lambda r: cpu.locate_batch(r), ranges)
    if rank == 0:
        cm, cr, cf = cpu.match_stats_batch(flat, off)
        assert np.array_equal(res[0], cm) and np.array_equal(res[1], cr) and np.array_equal(res[2], cf)
        lo, lv = cpu.locate_batch(ranges)
        assert np.array_equal(loc[0], lo) and np.array_equal(loc[1], lv)
        assert np.array_equal(np.diff(loc[0]), cpu.count_batch(ranges))          # benchmark/query_gcsa.cpp:171-179
        open(out_path, "w").write("ok")
    else:
        assert res is None and loc is None
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nq", [(2, 101), (3, 2), (2, 1)])
def test_sharded_config5_gloo(tmp_path, world, nq):
    """The control flow of the sharded matching statistics and locate (gcsa2_comm_match_stats / gcsa2_comm_locate):
    contiguous shards, per-rank totals, CSR offsets rebased on the root -- with the oracle standing in for the engine."""
    import torch.multiprocessing as mp
    out = tmp_path / "ok.txt"
    mp.spawn(_worker_config5, args=(world, _free_port(), nq, str(out)), nprocs=world, join=True)
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


def test_serialize_matches_the_file_format_restatement(built):
    """GCSA::serialize / LCPArray::serialize in C++ (csrc/sdsl_writer.hpp, behind gcsa2_host_view_serialize_*) emit
    byte for byte what the independent Python restatement of the format (workload/sdsl_format.py) emits, and the
    in-memory parser (what GCSA::load(std::istream&) of the facade calls) reads them back to the same members,
    with and without trailing bytes in the stream.  Host only."""
    from gcsa2_amd.hostview import make_host_view
    from workload import graphs, sdsl_format, builder
    from workload.brute_builder import build
    cases = [build(graphs.paper_graph(), 3, sample_period=2, branching=2),
             build(graphs.snp_graph(150, 0x52, 0x53, snp_period=8, node_len=8), 6, sample_period=8, branching=4),
             builder.build(graphs.snp_graph(40000, 0x61, 0x62), 16)]
    for ix in cases:
        holder = make_host_view(ix)
        raw = built.serialize_view(holder.ref(), "gcsa")
        assert raw == sdsl_format.gcsa_bytes(ix)
        lcp_raw = built.serialize_view(holder.ref(), "lcp")
        assert lcp_raw == sdsl_format.lcp_bytes(ix)
        h, vp, used = built.parse_view(raw, "gcsa")
        _view_matches(vp.contents, ix)
        built.free_view(h)
        h, vp, used = built.parse_view(raw + b"TRAILING", "gcsa", exact=False)
        assert used == len(raw)
        _view_matches(vp.contents, ix)
        built.free_view(h)
        with pytest.raises(built.Gcsa2Error):
            built.parse_view(raw + b"TRAILING", "gcsa")
        h, vp, used = built.parse_view(lcp_raw + b"xx", "lcp", exact=False)
        v = vp.contents
        assert used == len(lcp_raw) and (v.lcp_size, v.lcp_branching, v.lcp_levels) == (ix.lcp_size, ix.lcp_branching, len(ix.lcp_offsets) - 1)
        assert np.ctypeslib.as_array(v.lcp_data, shape=(int(ix.lcp_offsets[-1]),)).tolist() == ix.lcp_data.tolist()
        assert not v.bwt and v.path_nodes == 0
        built.free_view(h)
    # a view without samples / counters cannot become a .gcsa file
    with pytest.raises(built.Gcsa2Error):
        built.serialize_view(make_host_view(cases[0], with_counters=False).ref(), "gcsa")
    # alpha.comp2char survives load + serialize (GCSA::load keeps the file's): a file whose comp2char differs
    # from what char2comp would give -- here lower-case letters -- is reproduced byte for byte
    raw = built.serialize_view(make_host_view(cases[1]).ref(), "gcsa")
    at = raw.index(b"$ACGTN#")
    custom = raw[:at] + b"$acgtn#" + raw[at + 7:]
    h, vp, used = built.parse_view(custom, "gcsa")
    assert bytes(vp.contents.comp2char[:7]) == b"$acgtn#"
    assert built.serialize_view(vp, "gcsa") == custom
    built.free_view(h)
    # ... and the one derivation rule (facade Alphabet::read and the serializer of a view without comp2char)
    import ctypes as C
    out = (C.c_uint8 * 7)()
    c2c = np.ascontiguousarray(cases[1].char2comp, dtype=np.uint8)
    built.load_library().gcsa2_derive_comp2char(c2c.ctypes.data_as(C.POINTER(C.c_uint8)), 7, out)
    assert bytes(out) == b"$ACGTN#"
    odd = np.full(256, 2, dtype=np.uint8)
    odd[0] = 0; odd[ord("x")] = 1; odd[ord("X")] = 1
    out3 = (C.c_uint8 * 3)()
    built.load_library().gcsa2_derive_comp2char(odd.ctypes.data_as(C.POINTER(C.c_uint8)), 3, out3)
    assert bytes(out3) == bytes([0, ord("X"), 1])          # comp 0 holds only NUL; comp 2's first byte that is not NUL / lower case


def test_header_bytes_from_the_reference_definitions(built, tmp_path):
    """GCSAHeader / LCPHeader byte layouts hand-assembled from the reference's definitions, not from any writer of
    this repository: GCSAHeader = u32 tag 0x6C5A6C5A, u32 version 3, u64 path_nodes, edges, order, flags
    (include/gcsa/files.h:135-156, src/files.cpp:513-537: 40 bytes); LCPHeader = u32 tag 0x6C5A7C94, u32 version 1,
    u64 size, branching, flags (files.h:169-190, files.cpp:581-603: 32 bytes).  check() accepts exactly
    tag + version (+ flags == 0); GCSA::load / LCPArray::load throw "Invalid header" otherwise (gcsa.cpp:188-193,
    lcp.cpp:134-139)."""
    import struct
    from workload import graphs, sdsl_format
    from workload.brute_builder import build
    ix = build(graphs.paper_graph(), 3, sample_period=2, branching=2)
    body = sdsl_format.gcsa_bytes(ix)[40:]
    lcp_body = sdsl_format.lcp_bytes(ix)[32:]
    good = bytes.fromhex("5a6c5a6c" "03000000") + struct.pack("<QQQQ", 16, 20, 3, 0)       # the paper's example: 16 path nodes, 20 edges, order 3
    assert len(good) == 40
    h, vp, _ = built.parse_view(good + body)
    assert (vp.contents.path_nodes, vp.contents.edges, vp.contents.order) == (16, 20, 3)
    built.free_view(h)
    good_lcp = bytes.fromhex("947c5a6c" "01000000") + struct.pack("<QQQ", 16, 2, 0)
    assert len(good_lcp) == 32
    h, vp, _ = built.parse_view(good_lcp + lcp_body, "lcp")
    assert (vp.contents.lcp_size, vp.contents.lcp_branching) == (16, 2)
    built.free_view(h)
    bad_headers = {
        "tag of the LCP file": bytes.fromhex("947c5a6c" "03000000") + good[8:],
        "byte-swapped tag": bytes.fromhex("6c5a6c5a" "03000000") + good[8:],
        "version 2 (old format)": good[:4] + struct.pack("<I", 2) + good[8:],
        "version 4": good[:4] + struct.pack("<I", 4) + good[8:],
        "flags set": good[:32] + struct.pack("<Q", 1),
    }
    for name, head in bad_headers.items():
        with pytest.raises(built.Gcsa2Error, match="Invalid header"):
            built.parse_view(head + body)
    with pytest.raises(built.Gcsa2Error):
        built.parse_view(good[:39] + body)           # a 39-byte header shifts every member
    with pytest.raises(built.Gcsa2Error, match="truncated"):
        built.parse_view(good[:39])
    for name, head in {"GCSA tag": good[:8] + good_lcp[8:], "version 0": good_lcp[:4] + struct.pack("<I", 0) + good_lcp[8:],
                       "flags set": good_lcp[:24] + struct.pack("<Q", 2)}.items():
        with pytest.raises(built.Gcsa2Error, match="Invalid header"):
            built.parse_view(head + lcp_body, "lcp")
    # header fields that contradict the members are refused too: one path node too many, one edge too few
    for head in (good[:8] + struct.pack("<QQQQ", 17, 20, 3, 0), good[:8] + struct.pack("<QQQQ", 16, 19, 3, 0)):
        with pytest.raises(built.Gcsa2Error):
            built.parse_view(head + body)


def test_create_validates_the_view(built):
    """gcsa2_index_create refuses inconsistent views before touching a device: char2comp out of range, C not
    monotone / not ending at `edges`, an LCP array of another index (size, branching, offsets)."""
    import copy
    from workload import graphs
    from workload.brute_builder import build
    ix = build(graphs.snp_graph(150, 0x52, 0x53, snp_period=8, node_len=8), 6, sample_period=8, branching=4)
    other = build(graphs.snp_graph(90, 0x54, 0x55, snp_period=8, node_len=8), 6, sample_period=8, branching=4)

    def refuses(mutate, text):
        bad = copy.copy(ix)
        mutate(bad)
        with pytest.raises(built.Gcsa2Error, match=text) as e:
            built.GCSA(bad)
        assert e.value.code == -1           # GCSA2_ERR_INVALID_ARGUMENT, not NO_DEVICE: checked before any device work

    def c2c(b):
        b.char2comp = b.char2comp.copy(); b.char2comp[200] = 7
    refuses(c2c, "char2comp")

    def cmono(b):
        b.C = b.C.copy(); b.C[2], b.C[3] = b.C[3], b.C[2] - 1
    refuses(cmono, "non-decreasing")

    def cend(b):
        b.C = b.C.copy(); b.C[-1] += 1
    refuses(cend, "edges")

    def stale_lcp(b):
        b.lcp_data, b.lcp_offsets, b.lcp_size = other.lcp_data, other.lcp_offsets, other.lcp_size
    refuses(stale_lcp, "path nodes")

    def branching(b):
        b.lcp_branching = 1
    refuses(branching, "branching")

    def offsets(b):
        b.lcp_offsets = b.lcp_offsets.copy(); b.lcp_offsets[1] -= 1
    refuses(offsets, "offsets")

    def wrong_branching(b):
        b.lcp_branching = 8
    refuses(wrong_branching, "offsets|root")


def test_gcsa_inspect_lists_every_member(built, tmp_path):
    """tools/cpp/gcsa_inspect: the walk over a .gcsa / .lcp pair (here written by the independent Python restatement of the
    container encodings, workload/sdsl_format.py) accounts for every byte and names each member of GCSA::serialize /
    LCPArray::serialize in order (reference src/gcsa.cpp:140-179, src/lcp.cpp:116-128); a damaged file stops the walk at the
    member and byte offset in question.  This is the tool to run first on a file written by the real library (SURVEY 8(f)-1:
    the encodings are unpinned here)."""
    import subprocess
    from gcsa2_amd import build as engine_build
    from workload import graphs, builder, sdsl_format
    tool = engine_build.build_gcsa_inspect()
    ix = builder.build(graphs.snp_graph(3000, 0x21, 0x22, snp_period=7, node_len=16), 16, sample_period=16, branching=4)
    base = str(tmp_path / "g")
    sdsl_format.write(ix, base)
    out = subprocess.run([tool, base + ".gcsa", base + ".lcp"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr
    lines = out.stdout.splitlines()
    assert sum("every byte accounted for" in l for l in lines) == 2
    members = [l.split()[0] for l in lines if l and not l.startswith(("#", "member"))]
    want = (["header", "alpha.char2comp", "alpha.comp2char", "alpha.C", "alpha.sigma,"] + [f"fast_bwt[{c}]" for c in range(ix.sigma)] + ["fast_rank[0..sigma)"]
            + [f"sparse_bwt[{c}]" for c in range(ix.sigma)] + ["sparse_rank[0..sigma)", "edges", "sampled_paths", "stored_samples", "samples",
            "sample_select", "extra_pointers.filter", "extra_pointers.values", "redundant_pointers.data", "redundant_pointers.select", "header", "data", "offsets"])
    assert members == want
    head = [l for l in lines if l.startswith("header")][0]
    assert f"path_nodes {ix.n}," in head and f"edges {ix.e}," in head and "tag 0x6C5A6C5A" in head
    # offsets are contiguous: every member starts where the one before ended
    rows = [l.split() for l in lines if l and not l.startswith(("#", "member"))]
    gcsa_rows = rows[: len(rows) - 3]
    at = 0
    for r in gcsa_rows:
        nums = [int(x) for x in r if x.isdigit()][:2]
        assert nums[0] == at, r
        at += nums[1]
    assert at == os.path.getsize(base + ".gcsa")
    # a damaged file: the walk stops and says where
    blob = bytearray(open(base + ".gcsa", "rb").read())
    blob[len(blob) // 2] ^= 0xFF
    open(base + "_bad.gcsa", "wb").write(bytes(blob[: len(blob) - 9]))
    out = subprocess.run([tool, base + "_bad.gcsa"], capture_output=True, text=True)
    assert out.returncode != 0 and ("STOPPED" in out.stdout or "NOT ACCOUNTED" in out.stdout)


def test_editing_a_facade_header_rebuilds_the_cli(built):
    """build_cli's dependency list is every header a facade client is compiled from (include/gcsa/*.h through
    include/gcsa2_hip/gcsa.hpp): a newer include/gcsa/gcsa.h makes query_gcsa stale and it is recompiled."""
    import glob
    from gcsa2_amd import build as engine_build
    src = os.path.join(ROOT, "tools", "cpp", "query_gcsa.cpp")
    deps = engine_build.cli_deps(src)
    for h in glob.glob(os.path.join(ROOT, "include", "gcsa", "*.h")) + glob.glob(os.path.join(ROOT, "include", "gcsa2_hip", "*.hpp")):
        assert h in deps, h
    out = engine_build.build_query_gcsa()
    first = os.stat(out).st_mtime_ns
    assert engine_build.build_query_gcsa() == out and os.stat(out).st_mtime_ns == first          # up to date: left alone
    header = os.path.join(ROOT, "include", "gcsa", "gcsa.h")
    st = os.stat(header)
    try:
        os.utime(header, ns=(st.st_atime_ns, max(first, st.st_mtime_ns) + 2_000_000_000))
        engine_build.build_query_gcsa()
        assert os.stat(out).st_mtime_ns > first
    finally:
        os.utime(header, ns=(st.st_atime_ns, st.st_mtime_ns))
    # (the rebuilt binary is now newer than the restored header: nothing is stale afterwards)
    second = os.stat(out).st_mtime_ns
    assert engine_build.build_query_gcsa() == out and os.stat(out).st_mtime_ns == second
