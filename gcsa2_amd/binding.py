"""ctypes binding of the C ABI (include/gcsa2_hip.h) + a host-side mirror of the reference's
query interface.

`GCSA` / `LCPArray` keep the reference's method names and argument meaning
(`include/gcsa/gcsa.h:96-210`, `include/gcsa/lcp.h:137-178`) so that parity tests read like the
reference's own checks; every method runs on the GPU through `libgcsa2_hip.so`.  There is no CPU
fallback: if the library or a device is missing, construction raises.
"""
import ctypes as C
import os
import sys
import numpy as np

from .hostview import (HostView, STNode, STNODE_DTYPE, UNKNOWN, make_host_view, concat_patterns,
                       u64p, u8p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GCSA2_HIP_LIB") or os.path.join(_HERE, "lib", "libgcsa2_hip.so")   # override: A/B runs of two builds

STATUS = {0: "OK", -1: "INVALID_ARGUMENT", -2: "NO_DEVICE", -3: "OUT_OF_MEMORY", -4: "HIP",
          -5: "MISSING_COMPONENT", -6: "BUFFER_TOO_SMALL"}

EXPORTS = [
    "gcsa2_device_count", "gcsa2_index_create", "gcsa2_index_destroy", "gcsa2_index_set_tables", "gcsa2_index_trim", "gcsa2_index_set_pipeline", "gcsa2_last_error",
    "gcsa2_size", "gcsa2_edge_count", "gcsa2_order", "gcsa2_sample_count", "gcsa2_sample_bits",
    "gcsa2_device", "gcsa2_device_bytes", "gcsa2_block_bits",
    "gcsa2_find_batch", "gcsa2_find_batch_packed", "gcsa2_find_packed_device", "gcsa2_find_device", "gcsa2_find_stats_device", "gcsa2_find_device_variant",
    "gcsa2_find_block_bytes", "gcsa2_kmer_table_k", "gcsa2_locate_table_bytes", "gcsa2_jump_table_bytes", "gcsa2_pair_block_bytes", "gcsa2_lf_batch", "gcsa2_lf_device",
    "gcsa2_lf_node_batch", "gcsa2_char_range", "gcsa2_lf_all_batch",
    "gcsa2_count_batch", "gcsa2_count_device",
    "gcsa2_locate_run", "gcsa2_locate_fetch", "gcsa2_locate_discard", "gcsa2_locate_device", "gcsa2_locate_into",
    "gcsa2_parent_batch", "gcsa2_parent_device", "gcsa2_depth_batch", "gcsa2_sv_batch",
    "gcsa2_rmq_batch", "gcsa2_locate_max", "gcsa2_sample_range_batch", "gcsa2_sample_batch",
    "gcsa2_sampled_positions", "gcsa2_sigma", "gcsa2_fast_chars", "gcsa2_alphabet", "gcsa2_derive_comp2char",
    "gcsa2_lcp_size", "gcsa2_lcp_values", "gcsa2_lcp_levels", "gcsa2_lcp_branching",
    "gcsa2_lcp_access_batch",
    "gcsa2_group_create", "gcsa2_group_destroy", "gcsa2_group_size", "gcsa2_group_index",
    "gcsa2_group_find_batch", "gcsa2_group_find_device", "gcsa2_group_uses_rccl",
    "gcsa2_group_match_stats_device", "gcsa2_group_locate_device", "gcsa2_comm_match_stats", "gcsa2_comm_locate",
    "gcsa2_comm_unique_id", "gcsa2_comm_create", "gcsa2_comm_create_custom", "gcsa2_comm_destroy", "gcsa2_comm_rank", "gcsa2_comm_world", "gcsa2_comm_rccl_ranks", "gcsa2_comm_gather",
    "gcsa2_pack_ranges32_device", "gcsa2_unpack_ranges32_device", "gcsa2_pack_ranges40_device", "gcsa2_unpack_ranges40_device", "gcsa2_wire48_bytes", "gcsa2_mailbox_stats", "gcsa2_pack_ranges48_device", "gcsa2_unpack_ranges48_device", "gcsa2_count_kmers", "gcsa2_compare_kmers", "gcsa2_compare_kmers_records", "gcsa2_match_stats_batch", "gcsa2_match_stats_device", "gcsa2_match_stats_device_variant", "gcsa2_match_stats_device_sized", "gcsa2_match_stats_profile_device", "gcsa2_match_breaks_device", "gcsa2_match_breaks_batch",
    "gcsa2_host_view_save", "gcsa2_host_view_load", "gcsa2_host_view_get", "gcsa2_host_view_free",
    "gcsa2_index_create_from_file", "gcsa2_host_view_load_gcsa", "gcsa2_index_create_from_gcsa",
    "gcsa2_host_view_parse_gcsa", "gcsa2_host_view_parse_lcp", "gcsa2_host_view_serialize_gcsa", "gcsa2_host_view_serialize_lcp",
    "gcsa2_lcp_create",
]


SINK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint64)     # gcsa2_sink


class Gcsa2Error(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"gcsa2_hip: {STATUS.get(code, code)}: {message}")
        self.code = code


_lib = None


def load_library():
    """dlopen libgcsa2_hip.so.  When torch is installed it is imported first so that both share
    the one HIP runtime already mapped into the process (same soname, libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Gcsa2Error(-2, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        import torch  # noqa: F401  (plumbing only: one HIP runtime per process)
        # ... and one RCCL: the gather entry points bind RCCL at run time (csrc/comm.hpp) and are pointed at
        # the copy torch ships and loads for its own "nccl" backend, when there is one
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(bundled):
            os.environ.setdefault("GCSA2_RCCL_LIB", bundled)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.gcsa2_last_error.restype = C.c_char_p
    L.gcsa2_index_create.argtypes = [C.POINTER(HostView), i32, C.POINTER(vp)]
    L.gcsa2_index_set_tables.argtypes = [vp, i32, i32, i32]
    L.gcsa2_index_trim.argtypes = [vp]
    L.gcsa2_index_set_pipeline.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.gcsa2_index_destroy.argtypes = [vp]
    L.gcsa2_index_destroy.restype = None
    for name in ("gcsa2_size", "gcsa2_edge_count", "gcsa2_order", "gcsa2_sample_count",
                 "gcsa2_sample_bits", "gcsa2_device_bytes", "gcsa2_block_bits", "gcsa2_find_block_bytes", "gcsa2_kmer_table_k", "gcsa2_locate_table_bytes", "gcsa2_jump_table_bytes", "gcsa2_pair_block_bytes",
                 "gcsa2_sampled_positions", "gcsa2_sigma", "gcsa2_fast_chars", "gcsa2_lcp_size",
                 "gcsa2_lcp_values", "gcsa2_lcp_levels", "gcsa2_lcp_branching"):
        getattr(L, name).restype = u64
        getattr(L, name).argtypes = [vp]
    L.gcsa2_device.argtypes = [vp]
    L.gcsa2_find_batch.argtypes = [vp, u8p, u64p, u64, u64p]
    L.gcsa2_find_device.argtypes = [vp, vp, vp, u64, vp, vp]
    L.gcsa2_find_batch_packed.argtypes = [vp, u64p, u64, u64, u64p]
    L.gcsa2_find_packed_device.argtypes = [vp, vp, u64, u64, vp, vp]
    L.gcsa2_find_stats_device.argtypes = [vp, vp, vp, u64, vp, vp, vp]
    L.gcsa2_find_device_variant.argtypes = [vp, i32, vp, vp, u64, vp, vp]
    L.gcsa2_lf_batch.argtypes = [vp, u64p, u8p, u64, u64p]
    L.gcsa2_lf_device.argtypes = [vp, vp, vp, u64, vp, vp]
    L.gcsa2_lf_node_batch.argtypes = [vp, u64p, u64, u64p]
    L.gcsa2_char_range.argtypes = [vp, C.c_uint8, u64p, u64p]
    L.gcsa2_lf_all_batch.argtypes = [vp, u64p, u64, i32, u64p]
    L.gcsa2_count_batch.argtypes = [vp, u64p, u64, u64p]
    L.gcsa2_count_device.argtypes = [vp, vp, u64, vp, vp]
    L.gcsa2_locate_run.argtypes = [vp, u64p, u64, i32, u64p, C.POINTER(vp)]
    L.gcsa2_locate_fetch.argtypes = [vp, u64p, u64]
    L.gcsa2_locate_discard.argtypes = [vp]
    L.gcsa2_locate_discard.restype = None
    L.gcsa2_locate_device.argtypes = [vp, vp, u64, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                      u64p, vp]
    L.gcsa2_locate_into.argtypes = [vp, vp, u64, i32, vp, vp, u64, u64p, vp]
    L.gcsa2_parent_batch.argtypes = [vp, u64p, u64, vp]
    L.gcsa2_parent_device.argtypes = [vp, vp, u64, vp, vp]
    L.gcsa2_depth_batch.argtypes = [vp, u64p, u64, u64p]
    L.gcsa2_sv_batch.argtypes = [vp, i32, u64p, u64, u64p]
    L.gcsa2_rmq_batch.argtypes = [vp, u64p, u64, u64p]
    L.gcsa2_locate_max.argtypes = [vp, u64, u64, u64, u64p, u64, u64p]
    L.gcsa2_sample_range_batch.argtypes = [vp, u64p, u64, u64p]
    L.gcsa2_sample_batch.argtypes = [vp, u64p, u64, u64p, u8p]
    L.gcsa2_alphabet.argtypes = [vp, u8p, u64p]
    L.gcsa2_alphabet.restype = None
    L.gcsa2_lcp_access_batch.argtypes = [vp, u64p, u64, u64p]
    L.gcsa2_count_kmers.argtypes = [vp, u64, i32, i32, u64p]
    L.gcsa2_compare_kmers.argtypes = [vp, vp, u64, i32, i32, u64p]
    L.gcsa2_compare_kmers_records.argtypes = [vp, vp, u64, i32, i32, u64p, u64p, u64, u64p, u64]
    L.gcsa2_host_view_save.argtypes = [C.POINTER(HostView), C.c_char_p]
    L.gcsa2_host_view_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.gcsa2_host_view_get.argtypes = [vp]
    L.gcsa2_host_view_get.restype = C.POINTER(HostView)
    L.gcsa2_host_view_free.argtypes = [vp]
    L.gcsa2_host_view_free.restype = None
    L.gcsa2_index_create_from_file.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.gcsa2_host_view_load_gcsa.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.gcsa2_index_create_from_gcsa.argtypes = [C.c_char_p, C.c_char_p, i32, C.POINTER(vp)]
    L.gcsa2_host_view_parse_gcsa.argtypes = [vp, u64, u64p, C.POINTER(vp)]
    L.gcsa2_host_view_parse_lcp.argtypes = [vp, u64, u64p, C.POINTER(vp)]
    L.gcsa2_host_view_serialize_gcsa.argtypes = [C.POINTER(HostView), SINK, vp, u64p]
    L.gcsa2_host_view_serialize_lcp.argtypes = [C.POINTER(HostView), SINK, vp, u64p]
    L.gcsa2_lcp_create.argtypes = [C.POINTER(HostView), i32, C.POINTER(vp)]
    L.gcsa2_match_stats_batch.argtypes = [vp, u8p, u64p, u64, vp, u64p, u64p]
    L.gcsa2_match_stats_device.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp]
    L.gcsa2_match_stats_device_variant.argtypes = [vp, C.c_int, vp, vp, u64, vp, vp, vp, vp]
    L.gcsa2_match_stats_device_sized.argtypes = [vp, C.c_int, vp, vp, u64, u64, vp, vp, vp, vp]
    L.gcsa2_match_stats_profile_device.argtypes = [vp, vp, vp, u64, u64, vp, vp, vp, vp, vp]
    L.gcsa2_match_breaks_batch.argtypes = [vp, u8p, u64p, u64, u64, u64p, vp, u64, u64p, vp, vp]
    L.gcsa2_match_breaks_device.argtypes = [vp, vp, vp, u64, u64, i32, u64, vp, vp, u64, u64p, vp, vp, vp]
    L.gcsa2_group_create.argtypes = [C.POINTER(HostView), C.POINTER(i32), i32, C.POINTER(vp)]
    L.gcsa2_group_destroy.argtypes = [vp]
    L.gcsa2_group_destroy.restype = None
    L.gcsa2_group_size.argtypes = [vp]
    L.gcsa2_group_index.argtypes = [vp, i32]
    L.gcsa2_group_index.restype = vp
    L.gcsa2_group_find_batch.argtypes = [vp, u8p, u64p, u64, u64p]
    L.gcsa2_group_find_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), u64p, vp]
    L.gcsa2_group_uses_rccl.argtypes = [vp]
    L.gcsa2_group_match_stats_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), u64p, u64p, vp, vp, vp]
    L.gcsa2_group_locate_device.argtypes = [vp, C.POINTER(vp), u64p, i32, vp, C.POINTER(vp), C.POINTER(vp), u64p]
    L.gcsa2_comm_match_stats.argtypes = [vp, vp, vp, vp, u64p, u64p, i32, vp, vp, vp, vp]
    L.gcsa2_comm_locate.argtypes = [vp, vp, vp, u64p, i32, i32, vp, C.POINTER(vp), C.POINTER(vp), u64p, vp]
    L.gcsa2_comm_unique_id.argtypes = [u8p]
    L.gcsa2_comm_create.argtypes = [u8p, i32, i32, i32, C.POINTER(vp)]
    L.gcsa2_comm_create_custom.argtypes = [i32, i32, i32, vp, vp, C.POINTER(vp)]
    L.gcsa2_comm_destroy.argtypes = [vp]
    L.gcsa2_comm_destroy.restype = None
    L.gcsa2_comm_rank.argtypes = [vp]
    L.gcsa2_comm_world.argtypes = [vp]
    L.gcsa2_comm_rccl_ranks.argtypes = [vp, C.POINTER(C.c_int)]
    L.gcsa2_comm_gather.argtypes = [vp, vp, u64p, vp, i32, vp]
    L.gcsa2_pack_ranges32_device.argtypes = [vp, u64, vp, vp]
    L.gcsa2_unpack_ranges32_device.argtypes = [vp, u64, vp, vp]
    L.gcsa2_pack_ranges40_device.argtypes = [vp, u64, vp, vp]
    L.gcsa2_unpack_ranges40_device.argtypes = [vp, u64, vp, vp]
    L.gcsa2_mailbox_stats.argtypes = [vp, u64p, u64p]
    L.gcsa2_wire48_bytes.argtypes = [u64, u64]
    L.gcsa2_wire48_bytes.restype = u64
    L.gcsa2_pack_ranges48_device.argtypes = [vp, u64, vp, u64, vp]
    L.gcsa2_unpack_ranges48_device.argtypes = [vp, u64, u64, vp, vp, vp]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise Gcsa2Error(rc, load_library().gcsa2_last_error().decode(errors="replace"))


def device_count():
    rc = load_library().gcsa2_device_count()
    if rc < 0:
        _check(rc)
    return rc


def _p64(a):
    return a.ctypes.data_as(u64p)


def _p8(a):
    return a.ctypes.data_as(u8p)


def _ranges(r):
    r = np.ascontiguousarray(r, dtype=np.uint64)
    if r.ndim == 1:
        r = r.reshape(-1, 2)
    return r


class Range:
    """`gcsa::Range` (reference include/gcsa/utils.h:84-117)."""

    @staticmethod
    def length(r):
        return (int(r[1]) + 1 - int(r[0])) & UNKNOWN

    @staticmethod
    def empty(r):
        return ((int(r[0]) + 1) & UNKNOWN) > ((int(r[1]) + 1) & UNKNOWN)

    @staticmethod
    def empty_range():
        return (1, 0)


class Node:
    """`gcsa::Node` (reference include/gcsa/support.h:443-471): id << 11 | rc << 10 | offset."""
    OFFSET_BITS = 10
    ID_OFFSET = 11

    @staticmethod
    def encode(node_id, offset=0, rc=False):
        return (node_id << 11) | (int(bool(rc)) << 10) | offset

    @staticmethod
    def id(node):
        return int(node) >> 11

    @staticmethod
    def rc(node):
        return bool((int(node) >> 10) & 1)

    @staticmethod
    def offset(node):
        return int(node) & 1023


def save_host_view(index_arrays, path, **view_kwargs):
    """Write the G2HV container file of an index (host only)."""
    holder = make_host_view(index_arrays, **view_kwargs)
    _check(load_library().gcsa2_host_view_save(holder.ref(), os.fsencode(path)))


class LoadedHostView:
    """A G2HV file, or a `.gcsa` (+ `.lcp`) pair, read back into host memory (host only); `.view` is
    the ctypes HostView."""

    def __init__(self, path, lcp_path=None):
        self._L = load_library()
        h = C.c_void_p()
        if os.fspath(path).endswith(".gcsa"):
            _check(self._L.gcsa2_host_view_load_gcsa(os.fsencode(path), os.fsencode(lcp_path) if lcp_path else None, C.byref(h)))
        else:
            _check(self._L.gcsa2_host_view_load(os.fsencode(path), C.byref(h)))
        self._h = h
        self.view = self._L.gcsa2_host_view_get(h).contents

    def close(self):
        if getattr(self, "_h", None):
            self._L.gcsa2_host_view_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GCSA:
    """Device-resident index with the reference's query methods.

    Scalar methods (`find`, `LF`, `count`, `locate`, ...) are one-element batches of the batched
    ones; the batched ones take / return numpy arrays.  `*_device` methods take raw device
    pointers (ints) and a stream pointer and only enqueue work."""

    def __init__(self, index_arrays, device=0, **view_kwargs):
        L = load_library()
        h = C.c_void_p()
        if isinstance(index_arrays, (str, bytes, os.PathLike)):      # a G2HV container or a `.gcsa` file (+ `.lcp` beside it)
            path = os.fspath(index_arrays)
            path = path.decode() if isinstance(path, bytes) else path
            if path.endswith(".gcsa"):     # GCSA::EXTENSION / LCPArray::EXTENSION (gcsa.h:63, lcp.h:110)
                # query_gcsa names the pair base.gcsa + base.lcp (benchmark/query_gcsa.cpp:53-63), vg base.gcsa + base.gcsa.lcp
                beside = [p for p in (path[:-5] + ".lcp", path + ".lcp") if os.path.exists(p)]
                lcp = view_kwargs.pop("lcp_path", beside[0] if beside else None)
                _check(L.gcsa2_index_create_from_gcsa(os.fsencode(path), os.fsencode(lcp) if lcp else None, device, C.byref(h)))
            else:
                _check(L.gcsa2_index_create_from_file(os.fsencode(path), device, C.byref(h)))
            self._h, self._L = h, L
            self.sigma = int(L.gcsa2_sigma(h))
            self.fast_chars = int(L.gcsa2_fast_chars(h))
            self.char2comp = np.zeros(256, dtype=np.uint8)
            L.gcsa2_alphabet(h, _p8(self.char2comp), None)
            return
        holder = make_host_view(index_arrays, **view_kwargs)
        _check(L.gcsa2_index_create(holder.ref(), device, C.byref(h)))
        self._h = h
        self._L = L
        self.char2comp = np.asarray(index_arrays.char2comp, dtype=np.uint8).copy()
        self.sigma = int(index_arrays.sigma)
        self.fast_chars = int(index_arrays.fast_chars)

    def close(self):
        if getattr(self, "_h", None):
            self._L.gcsa2_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # header accessors (gcsa.h:137-148)
    def size(self):
        return int(self._L.gcsa2_size(self._h))

    def empty(self):
        return self.size() == 0

    def edgeCount(self):
        return int(self._L.gcsa2_edge_count(self._h))

    def order(self):
        return int(self._L.gcsa2_order(self._h))

    def sampleCount(self):
        return int(self._L.gcsa2_sample_count(self._h))

    def sampleBits(self):
        return int(self._L.gcsa2_sample_bits(self._h))

    def device_bytes(self):
        return int(self._L.gcsa2_device_bytes(self._h))

    def block_bits(self):
        return int(self._L.gcsa2_block_bits(self._h))

    # ---- find ---------------------------------------------------------------------------
    def find_batch(self, patterns, offsets, out=None):
        """`out`: an (nq, 2) uint64 array to receive the ranges (a caller that reuses its result buffer avoids the page
        faults of a fresh one, which cost as much as the batch itself at 10 M queries)."""
        patterns = np.ascontiguousarray(patterns, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nq = offsets.shape[0] - 1
        if out is None:
            out = np.zeros((nq, 2), dtype=np.uint64)
        assert out.dtype == np.uint64 and out.shape == (nq, 2) and out.flags["C_CONTIGUOUS"]
        _check(self._L.gcsa2_find_batch(self._h, _p8(patterns), _p64(offsets), nq, _p64(out)))
        return out

    def find(self, pattern):
        data, off = concat_patterns([pattern])
        r = self.find_batch(data, off)[0]
        return (int(r[0]), int(r[1]))

    def find_batch_packed(self, codes, pattern_length, out=None):
        """find() of patterns of one length given as 2-bit codes (pack_kmers): (nq, W) uint64, W = ceil(pattern_length / 32)."""
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        words = (int(pattern_length) + 31) // 32
        nq = codes.size // words
        if out is None:
            out = np.zeros((nq, 2), dtype=np.uint64)
        assert out.dtype == np.uint64 and out.shape == (nq, 2) and out.flags["C_CONTIGUOUS"]
        _check(self._L.gcsa2_find_batch_packed(self._h, _p64(codes), int(pattern_length), nq, _p64(out)))
        return out

    def find_packed_device(self, d_codes, pattern_length, nq, d_ranges, stream=0):
        _check(self._L.gcsa2_find_packed_device(self._h, d_codes, int(pattern_length), nq, d_ranges, stream))

    def find_device(self, d_patterns, d_offsets, nq, d_ranges, stream=0):
        _check(self._L.gcsa2_find_device(self._h, d_patterns, d_offsets, nq, d_ranges, stream))

    def find_device_variant(self, variant, d_patterns, d_offsets, nq, d_ranges, stream=0):
        _check(self._L.gcsa2_find_device_variant(self._h, variant, d_patterns, d_offsets, nq, d_ranges, stream))

    def jump_table_bytes(self):
        return int(self._L.gcsa2_jump_table_bytes(self._h))

    def set_tables(self, pair_blocks=-1, kmer_k=-1, locate_table=-1):
        """Drop (0) / build (1) / leave (-1) the pair blocks and the locate table, resize the k-mer seed table (0 drops it):
        gcsa2_index_set_tables.  Results of every query stay the same; no queries may run on the handle meanwhile."""
        _check(self._L.gcsa2_index_set_tables(self._h, int(pair_blocks), int(kmer_k), int(locate_table)))

    def trim(self):
        """Give back the host pipeline, the staging objects and the scratch pool of this handle (gcsa2_index_trim)."""
        _check(self._L.gcsa2_index_trim(self._h))

    def set_pipeline(self, lanes=0, chunk_log2=0, blocking=-1):
        """Shape of the host pipeline of find_batch / find_batch_packed (gcsa2_index_set_pipeline; 0 leaves a value)."""
        _check(self._L.gcsa2_index_set_pipeline(self._h, int(lanes), int(chunk_log2), int(blocking)))

    def pair_block_bytes(self):
        return int(self._L.gcsa2_pair_block_bytes(self._h))

    def mailbox_stats(self):
        """(scalar calls answered by the resident wavefront, launches of it): gcsa2_mailbox_stats."""
        calls, launches = C.c_uint64(0), C.c_uint64(0)
        _check(self._L.gcsa2_mailbox_stats(self._h, C.byref(calls), C.byref(launches)))
        return int(calls.value), int(launches.value)

    def locate_table_bytes(self):
        return int(self._L.gcsa2_locate_table_bytes(self._h))

    def kmer_table_k(self):
        return int(self._L.gcsa2_kmer_table_k(self._h))

    def find_block_bytes(self):
        return int(self._L.gcsa2_find_block_bytes(self._h))

    def find_stats_device(self, d_patterns, d_offsets, nq, d_ranges, d_stats, stream=0):
        _check(self._L.gcsa2_find_stats_device(self._h, d_patterns, d_offsets, nq, d_ranges, d_stats, stream))

    # ---- LF -----------------------------------------------------------------------------
    def lf_batch(self, ranges, comps):
        ranges = _ranges(ranges)
        comps = np.ascontiguousarray(comps, dtype=np.uint8)
        out = np.zeros_like(ranges)
        _check(self._L.gcsa2_lf_batch(self._h, _p64(ranges), _p8(comps), ranges.shape[0], _p64(out)))
        return out

    def lf_node_batch(self, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
        out = np.zeros_like(nodes)
        _check(self._L.gcsa2_lf_node_batch(self._h, _p64(nodes), nodes.shape[0], _p64(out)))
        return out

    def LF(self, arg, comp=None):
        if comp is None:
            return int(self.lf_node_batch(np.array([arg], dtype=np.uint64))[0])
        r = self.lf_batch(np.array([arg], dtype=np.uint64), np.array([comp], dtype=np.uint8))[0]
        return (int(r[0]), int(r[1]))

    def charRange(self, comp):
        sp, ep = C.c_uint64(), C.c_uint64()
        _check(self._L.gcsa2_char_range(self._h, comp, C.byref(sp), C.byref(ep)))
        return (sp.value, ep.value)

    def lf_all_batch(self, ranges, all_comps):
        ranges = _ranges(ranges)
        out = np.zeros((ranges.shape[0], self.sigma, 2), dtype=np.uint64)
        _check(self._L.gcsa2_lf_all_batch(self._h, _p64(ranges), ranges.shape[0], int(all_comps), _p64(out)))
        return out

    def LF_fast(self, rng):
        return [(int(a), int(b)) for a, b in self.lf_all_batch(np.array([rng], dtype=np.uint64), 0)[0]]

    def LF_all(self, rng):
        return [(int(a), int(b)) for a, b in self.lf_all_batch(np.array([rng], dtype=np.uint64), 1)[0]]

    # ---- count --------------------------------------------------------------------------
    def count_batch(self, ranges):
        ranges = _ranges(ranges)
        out = np.zeros(ranges.shape[0], dtype=np.uint64)
        _check(self._L.gcsa2_count_batch(self._h, _p64(ranges), ranges.shape[0], _p64(out)))
        return out

    def count(self, rng):
        return int(self.count_batch(np.array([rng], dtype=np.uint64))[0])

    # ---- locate -------------------------------------------------------------------------
    def locate_batch(self, ranges, sort=True):
        """CSR (offsets[nq+1], values): sorted distinct node_type values per range
        (sort=False: path order with duplicates, as the reference's sort == false)."""
        ranges = _ranges(ranges)
        nq = ranges.shape[0]
        offsets = np.zeros(nq + 1, dtype=np.uint64)
        job = C.c_void_p()
        _check(self._L.gcsa2_locate_run(self._h, _p64(ranges), nq, int(sort), _p64(offsets), C.byref(job)))
        values = np.zeros(max(int(offsets[nq]), 1), dtype=np.uint64)
        _check(self._L.gcsa2_locate_fetch(job, _p64(values), values.shape[0]))
        return offsets, values[: int(offsets[nq])]

    def locate(self, rng, sort=True, max_positions=None):
        """locate(path_node) / locate(range) / locate(range, max_positions) (gcsa.h:126-128)."""
        if isinstance(rng, (int, np.integer)):
            rng = (int(rng), int(rng))
        if max_positions is None:
            return self.locate_batch(np.array([rng], dtype=np.uint64), sort)[1]
        cap = max(1, min(int(max_positions), self.count(rng)))
        values = np.zeros(cap, dtype=np.uint64)
        cnt = C.c_uint64()
        _check(self._L.gcsa2_locate_max(self._h, rng[0], rng[1], max_positions, _p64(values), cap, C.byref(cnt)))
        return values[: cnt.value]

    def compare_kmers(self, other, k, include_Ns=False, force=False):
        """`compareKMers(self, other, k)` (reference src/algorithms.cpp:534-616): (shared, left, right)."""
        out = np.zeros(3, dtype=np.uint64)
        _check(self._L.gcsa2_compare_kmers(self._h, other._h, k, int(include_Ns), int(force), _p64(out)))
        return tuple(int(x) for x in out)

    def compare_kmers_records(self, other, k, include_Ns=False, force=False):
        """compareKMers with parameters.output: (counts, left_states, right_states), states as
        (count, 8) uint64 arrays in the layout of the reference's .left / .right dumps."""
        counts = self.compare_kmers(other, k, include_Ns, force)
        left = np.zeros((max(counts[1], 1), 8), dtype=np.uint64)
        right = np.zeros((max(counts[2], 1), 8), dtype=np.uint64)
        out = np.zeros(3, dtype=np.uint64)
        _check(self._L.gcsa2_compare_kmers_records(self._h, other._h, k, int(include_Ns), int(force), _p64(out),
                                                   _p64(left), counts[1], _p64(right), counts[2]))
        return tuple(int(x) for x in out), left[: counts[1]], right[: counts[2]]

    def match_stats_batch(self, patterns, offsets, out=None):
        """Matching statistics by fused LF + parent (needs the LCP array):
        (ms uint16[total bytes], ranges (nq, 2), parent() calls per pattern).  `out`: the three arrays of an earlier call
        of the same shape, to be filled again (fresh arrays cost their page faults: 0.5 GB for 1 M x 256 bp)."""
        patterns = np.ascontiguousarray(patterns, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nq = offsets.shape[0] - 1
        if out is not None:
            ms, ranges, fallbacks = out
            if isinstance(ms.base, np.ndarray) and ms.base.dtype == np.uint16 and ms.base.ndim == 1 and ms.base.shape[0] >= max(int(offsets[nq]), 1):
                ms = ms.base              # the slice an earlier call returned: its whole array
            assert ms.dtype == np.uint16 and ms.shape[0] >= int(offsets[nq]) and ranges.shape == (nq, 2) and fallbacks.shape == (nq,)
        else:
            ms = np.zeros(max(int(offsets[nq]), 1), dtype=np.uint16)
            ranges = np.zeros((nq, 2), dtype=np.uint64)
            fallbacks = np.zeros(nq, dtype=np.uint64)
        _check(self._L.gcsa2_match_stats_batch(self._h, _p8(patterns), _p64(offsets), nq, ms.ctypes.data,
                                               _p64(ranges), _p64(fallbacks)))
        return ms[: int(offsets[nq])], ranges, fallbacks

    def match_stats_device(self, d_patterns, d_offsets, nq, d_ms, d_ranges, d_fallbacks=0, stream=0, variant=0, total_bytes=None):
        """total_bytes = offsets[nq] when the caller knows it: the call then only enqueues (otherwise it reads that
        value back from the device, which waits for the stream once)."""
        if total_bytes is None:
            _check(self._L.gcsa2_match_stats_device_variant(self._h, variant, d_patterns, d_offsets, nq, d_ms, d_ranges, d_fallbacks, stream))
        else:
            _check(self._L.gcsa2_match_stats_device_sized(self._h, variant, d_patterns, d_offsets, nq, int(total_bytes), d_ms, d_ranges,
                                                          d_fallbacks, stream))

    def match_breaks_batch(self, patterns, offsets, min_length=0, capacity=None, out=None):
        """Break points of a batch in host memory: (break_offsets (nq + 1), breaks (total, 4) = {position, length, sp, ep}, ranges,
        parent() counts).  The record buffer is sized from the refusal when `capacity` is too small (default: 4 per pattern).
        `out` = (break_offsets, breaks, ranges, fallbacks) arrays of the caller (a caller that reuses them avoids the page faults
        of fresh ones); too few rows in `breaks` raise BUFFER_TOO_SMALL with `needed`."""
        patterns = np.ascontiguousarray(patterns, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nq = offsets.shape[0] - 1
        if out is not None:
            boff, brk, rng, fb = out
            assert boff.dtype == np.uint64 and boff.shape[0] >= nq + 1 and brk.dtype == np.uint64 and brk.flags.c_contiguous and brk.shape[1] == 4
            assert rng.dtype == np.uint64 and rng.shape[0] >= nq and fb.dtype == np.uint64 and fb.shape[0] >= nq
            total = C.c_uint64()
            rc = self._L.gcsa2_match_breaks_batch(self._h, _p8(patterns), _p64(offsets), nq, int(min_length), _p64(boff), brk.ctypes.data, brk.shape[0],
                                                  C.byref(total), rng.ctypes.data, fb.ctypes.data)
            if rc == -6:
                err = Gcsa2Error(rc, self._L.gcsa2_last_error().decode(errors="replace"))
                err.needed = total.value
                raise err
            _check(rc)
            return boff[: nq + 1], brk[: total.value], rng[:nq], fb[:nq]
        cap = int(capacity if capacity is not None else 4 * nq + 16)
        boff = np.zeros(nq + 1, dtype=np.uint64)
        rng = np.zeros((max(nq, 1), 2), dtype=np.uint64)
        fb = np.zeros(max(nq, 1), dtype=np.uint64)
        total = C.c_uint64()
        for _ in range(2):
            brk = np.zeros((max(cap, 1), 4), dtype=np.uint64)
            rc = self._L.gcsa2_match_breaks_batch(self._h, _p8(patterns), _p64(offsets), nq, int(min_length), _p64(boff), brk.ctypes.data, cap,
                                                  C.byref(total), rng.ctypes.data, fb.ctypes.data)
            if rc == -6 and total.value > cap:
                cap = total.value
                continue
            _check(rc)
            return boff, brk[: total.value], rng[:nq], fb[:nq]
        _check(rc)

    def match_breaks_device(self, d_patterns, d_offsets, nq, total_bytes, d_break_offsets, d_breaks, capacity, d_ranges=0, d_fallbacks=0,
                            stream=0, variant=0, min_length=0):
        """Matching statistics as break points (gcsa2_match_breaks_device): the CSR of the left-maximal matches, records of four
        u64 {position, length, sp, ep}; returns the number of records.  Raises Gcsa2Error (BUFFER_TOO_SMALL, `.needed`) when
        `capacity` records are not enough."""
        total = C.c_uint64()
        tb = 0xFFFFFFFFFFFFFFFF if total_bytes is None else int(total_bytes)
        rc = self._L.gcsa2_match_breaks_device(self._h, d_patterns, d_offsets, nq, tb, variant, int(min_length), d_break_offsets, d_breaks, capacity, C.byref(total),
                                               d_ranges, d_fallbacks, stream)
        if rc != 0:
            err = Gcsa2Error(rc, self._L.gcsa2_last_error().decode(errors="replace"))
            err.needed = total.value
            raise err
        return total.value

    def match_stats_profile_device(self, d_patterns, d_offsets, nq, total_bytes, d_ms, d_ranges, d_fallbacks, d_prof, stream=0):
        """Diagnostic: the instrumented matching-statistics kernel (cycles per phase and event counts into d_prof[16])."""
        _check(self._L.gcsa2_match_stats_profile_device(self._h, d_patterns, d_offsets, nq, int(total_bytes), d_ms, d_ranges, d_fallbacks, d_prof, stream))

    def count_kmers(self, k, include_Ns=False, force=False):
        """`countKMers` (reference src/algorithms.cpp:387-421)."""
        res = C.c_uint64()
        _check(self._L.gcsa2_count_kmers(self._h, k, int(include_Ns), int(force), C.byref(res)))
        return res.value

    # ---- samples (gcsa.h:191-210) ---------------------------------------------------------
    def sample_range_batch(self, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
        out = np.zeros((nodes.shape[0], 3), dtype=np.uint64)
        _check(self._L.gcsa2_sample_range_batch(self._h, _p64(nodes), nodes.shape[0], _p64(out)))
        return out

    def sampled(self, node):
        return bool(self.sample_range_batch([node])[0, 0])

    def sampleRange(self, node):
        r = self.sample_range_batch([node])[0]
        return (int(r[1]), int(r[2]))

    def firstSample(self, node):
        return int(self.sample_range_batch([node])[0, 1])

    def sample_batch(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        values = np.zeros(idx.shape[0], dtype=np.uint64)
        last = np.zeros(idx.shape[0], dtype=np.uint8)
        _check(self._L.gcsa2_sample_batch(self._h, _p64(idx), idx.shape[0], _p64(values), _p8(last)))
        return values, last.astype(bool)

    def sample(self, i):
        return int(self.sample_batch([i])[0][0])

    def lastSample(self, i):
        return bool(self.sample_batch([i])[1][0])

    def sampledPositions(self):
        return int(self._L.gcsa2_sampled_positions(self._h))

    def locate_device(self, d_ranges, nq, stream=0, sort=True):
        """Returns (job, d_offsets, d_values, total); free with locate_discard(job)."""
        job, d_off, d_val = C.c_void_p(), C.c_void_p(), C.c_void_p()
        total = C.c_uint64()
        _check(self._L.gcsa2_locate_device(self._h, d_ranges, nq, int(sort), C.byref(job), C.byref(d_off),
                                           C.byref(d_val), C.byref(total), stream))
        return job, d_off.value, d_val.value, total.value

    def locate_discard(self, job):
        self._L.gcsa2_locate_discard(job)

    def locate_into(self, d_ranges, nq, d_offsets, d_values, capacity, stream=0, sort=True):
        """locate() into caller-owned device buffers; returns the number of values.  Raises
        Gcsa2Error (BUFFER_TOO_SMALL, `.needed` = values required) when capacity is insufficient."""
        total = C.c_uint64()
        rc = self._L.gcsa2_locate_into(self._h, d_ranges, nq, int(sort), d_offsets, d_values, capacity, C.byref(total), stream)
        if rc != 0:
            err = Gcsa2Error(rc, self._L.gcsa2_last_error().decode(errors="replace"))
            err.needed = total.value
            raise err
        return total.value

    def count_device(self, d_ranges, nq, d_counts, stream=0):
        _check(self._L.gcsa2_count_device(self._h, d_ranges, nq, d_counts, stream))

    def lf_device(self, d_in, d_comps, nq, d_out, stream=0):
        _check(self._L.gcsa2_lf_device(self._h, d_in, d_comps, nq, d_out, stream))

    def parent_device(self, d_ranges, nq, d_nodes, stream=0):
        _check(self._L.gcsa2_parent_device(self._h, d_ranges, nq, d_nodes, stream))


class LCPArray:
    """`gcsa::LCPArray` queries (reference include/gcsa/lcp.h:137-178) over the LCP part of the
    same device image as a `GCSA`."""

    def __init__(self, gcsa: GCSA, lcp_values: int, lcp_size: int):
        self._g = gcsa
        self._L = gcsa._L
        self._values = int(lcp_values)
        self._size = int(lcp_size)

    def size(self):
        return self._size

    def values(self):
        return self._values

    def notFound(self):
        return (self._values, self._values)

    def root(self):
        return (0, self._size - 1, 0, 0, 0)

    def parent_batch(self, ranges):
        ranges = _ranges(ranges)
        out = np.zeros(ranges.shape[0], dtype=STNODE_DTYPE)
        _check(self._L.gcsa2_parent_batch(self._g.handle, _p64(ranges), ranges.shape[0], out.ctypes.data))
        return out

    def parent(self, rng):
        return tuple(int(x) for x in self.parent_batch(np.array([rng], dtype=np.uint64))[0])

    def depth_batch(self, ranges):
        ranges = _ranges(ranges)
        out = np.zeros(ranges.shape[0], dtype=np.uint64)
        _check(self._L.gcsa2_depth_batch(self._g.handle, _p64(ranges), ranges.shape[0], _p64(out)))
        return out

    def depth(self, rng):
        return int(self.depth_batch(np.array([rng], dtype=np.uint64))[0])

    def _sv_batch(self, op, positions):
        positions = np.ascontiguousarray(positions, dtype=np.uint64)
        out = np.zeros((positions.shape[0], 2), dtype=np.uint64)
        _check(self._L.gcsa2_sv_batch(self._g.handle, op, _p64(positions), positions.shape[0], _p64(out)))
        return out

    def psv_batch(self, positions):
        return self._sv_batch(0, positions)

    def psev_batch(self, positions):
        return self._sv_batch(1, positions)

    def nsv_batch(self, positions):
        return self._sv_batch(2, positions)

    def nsev_batch(self, positions):
        return self._sv_batch(3, positions)

    def psv(self, pos):
        return tuple(int(x) for x in self._sv_batch(0, [pos])[0])

    def psev(self, pos):
        return tuple(int(x) for x in self._sv_batch(1, [pos])[0])

    def nsv(self, pos):
        return tuple(int(x) for x in self._sv_batch(2, [pos])[0])

    def nsev(self, pos):
        return tuple(int(x) for x in self._sv_batch(3, [pos])[0])

    def rmq_batch(self, ranges):
        ranges = _ranges(ranges)
        out = np.zeros((ranges.shape[0], 2), dtype=np.uint64)
        _check(self._L.gcsa2_rmq_batch(self._g.handle, _p64(ranges), ranges.shape[0], _p64(out)))
        return out

    def rmq(self, sp, ep):
        return tuple(int(x) for x in self.rmq_batch(np.array([[sp, ep]], dtype=np.uint64))[0])

    def levels(self):
        return int(self._L.gcsa2_lcp_levels(self._g.handle))

    def branching(self):
        return int(self._L.gcsa2_lcp_branching(self._g.handle))

    def access_batch(self, positions):
        positions = np.ascontiguousarray(positions, dtype=np.uint64)
        out = np.zeros(positions.shape[0], dtype=np.uint64)
        _check(self._L.gcsa2_lcp_access_batch(self._g.handle, _p64(positions), positions.shape[0], _p64(out)))
        return out

    def __getitem__(self, i):
        return int(self.access_batch([i])[0])

    def nodeFor(self, rng):
        """LCPArray::nodeFor (lcp.h:163-175)."""
        sp, ep = int(rng[0]), int(rng[1])
        right = self[ep + 1] if ep + 1 < self._size else 0
        return (sp, ep, self[sp], right, UNKNOWN)


class Comm:
    """RCCL communicator of the library (`gcsa2_comm_*`): one rank per GPU, one gather of hit ranges on the root.
    The 128-byte id is made on one rank (`Comm.unique_id()`) and handed to the others by the launcher."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = np.zeros(Comm.ID_BYTES, dtype=np.uint8)
        _check(load_library().gcsa2_comm_unique_id(_p8(buf)))
        return buf.tobytes()

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int):
        self._L = load_library()
        buf = np.frombuffer(unique_id, dtype=np.uint8).copy()
        assert buf.shape[0] == Comm.ID_BYTES
        h = C.c_void_p()
        _check(self._L.gcsa2_comm_create(_p8(buf), rank, world, device, C.byref(h)))
        self._h = h
        self.rank, self.world = rank, world

    GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.c_int, C.c_void_p)

    @classmethod
    def custom(cls, rank: int, world: int, device: int, gather):
        """A communicator over the application's own transport (gcsa2_comm_create_custom): `gather(d_send, sizes, d_recv, root,
        stream) -> int` with device pointers as integers and `sizes` the list of bytes per rank; 0 = success.  Everything above
        the transport -- sharded matching statistics, sharded locate() with its CSR rebasing -- is the library's C++ as with
        RCCL (gcsa2_amd/host_transport.py holds a gather through host memory over torch.distributed)."""
        self = cls.__new__(cls)
        self._L = load_library()
        self.rank, self.world = rank, world

        def trampoline(user, d_send, sizes, d_recv, root, stream):
            try:
                return int(gather(d_send or 0, [int(sizes[r]) for r in range(world)], d_recv or 0, int(root), stream or 0))
            except Exception as e:                   # an exception must not cross the C frames
                print(f"gcsa2 custom gather failed on rank {rank}: {e!r}", file=sys.stderr, flush=True)
                return 1
        self._callback = Comm.GATHER_FN(trampoline)             # kept alive with the communicator
        h = C.c_void_p()
        _check(self._L.gcsa2_comm_create_custom(rank, world, device, C.cast(self._callback, C.c_void_p), None, C.byref(h)))
        self._h = h
        return self

    def rccl_ranks(self) -> int:
        """The number of ranks RCCL itself reports for this communicator (ncclCommCount)."""
        n = C.c_int(0)
        _check(self._L.gcsa2_comm_rccl_ranks(self._h, C.byref(n)))
        return n.value

    def gather(self, d_send, bytes_per_rank, d_recv, root=0, stream=0):
        """Enqueue the gather: rank r sends bytes_per_rank[r] bytes from device pointer d_send; the root receives
        all parts back to back at device pointer d_recv."""
        sizes = np.asarray(bytes_per_rank, dtype=np.uint64)
        assert sizes.shape[0] == self.world
        _check(self._L.gcsa2_comm_gather(self._h, d_send, _p64(sizes), d_recv, root, stream))

    def match_stats(self, gcsa, d_patterns, d_offsets, counts, pattern_bytes, d_ms_root, d_ranges_root, d_fallbacks_root, root=0, stream=0):
        """Matching statistics of this rank's shard, gathered on the root in query order (enqueue-only).  counts[r] /
        pattern_bytes[r] = patterns / pattern bytes of rank r's shard; the *_root pointers are read on the root only."""
        cnt, pb = np.asarray(counts, dtype=np.uint64), np.asarray(pattern_bytes, dtype=np.uint64)
        assert cnt.shape[0] == self.world and pb.shape[0] == self.world
        _check(self._L.gcsa2_comm_match_stats(self._h, gcsa.handle, d_patterns, d_offsets, _p64(cnt), _p64(pb), root, d_ms_root, d_ranges_root,
                                              d_fallbacks_root, stream))

    def locate(self, gcsa, d_ranges, counts, d_offsets_root, root=0, stream=0, sort=True):
        """locate() of this rank's shard; on the root returns (job, d_values, total) for the whole batch in query order
        (offsets written to d_offsets_root: sum(counts) + 1 entries; free the job with gcsa.locate_discard), None elsewhere."""
        cnt = np.asarray(counts, dtype=np.uint64)
        assert cnt.shape[0] == self.world
        job, d_val, total = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(self._L.gcsa2_comm_locate(self._h, gcsa.handle, d_ranges, _p64(cnt), int(sort), root, d_offsets_root, C.byref(job), C.byref(d_val),
                                         C.byref(total), stream))
        return (job, d_val.value, total.value) if self.rank == root else None

    def close(self):
        if getattr(self, "_h", None):
            self._L.gcsa2_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def fetch_job(job, total: int) -> np.ndarray:
    """The values of a locate job (gcsa2_locate_device / group / comm locate) on the host; frees the job (gcsa2_locate_fetch)."""
    values = np.zeros(max(int(total), 1), dtype=np.uint64)
    _check(load_library().gcsa2_locate_fetch(job, _p64(values), values.shape[0]))
    return values[: int(total)]


def pack_ranges32_device(d_ranges, nq, d_packed, stream=0):
    _check(load_library().gcsa2_pack_ranges32_device(d_ranges, nq, d_packed, stream))


def unpack_ranges32_device(d_packed, nq, d_ranges, stream=0):
    _check(load_library().gcsa2_unpack_ranges32_device(d_packed, nq, d_ranges, stream))


def pack_ranges40_device(d_ranges, nq, d_packed, stream=0):
    _check(load_library().gcsa2_pack_ranges40_device(d_ranges, nq, d_packed, stream))


def unpack_ranges40_device(d_packed, nq, d_ranges, stream=0):
    _check(load_library().gcsa2_unpack_ranges40_device(d_packed, nq, d_ranges, stream))


def wire48_bytes(nq, capacity):
    """Size of a shard's block in the six-byte wire format: nq ranges and an overflow list of `capacity` entries."""
    return int(load_library().gcsa2_wire48_bytes(nq, capacity))


def pack_ranges48_device(d_ranges, nq, d_packed, capacity, stream=0):
    _check(load_library().gcsa2_pack_ranges48_device(d_ranges, nq, d_packed, capacity, stream))


def unpack_ranges48_device(d_packed, nq, capacity, d_ranges, d_overflow_count=0, stream=0):
    _check(load_library().gcsa2_unpack_ranges48_device(d_packed, nq, capacity, d_ranges, d_overflow_count, stream))


class GCSAGroup:
    """One replica per device; `find_batch` shards the batch contiguously over the replicas
    (single-process multi-GPU, `gcsa2_group_*`)."""

    def __init__(self, index_arrays, devices):
        L = load_library()
        holder = make_host_view(index_arrays)
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(L.gcsa2_group_create(holder.ref(), devs, len(devices), C.byref(h)))
        self._h, self._L = h, L

    def size(self):
        return int(self._L.gcsa2_group_size(self._h))

    def find_batch(self, patterns, offsets, out=None):
        """`out`: an (nq, 2) uint64 array to receive the ranges (a caller that reuses its result buffer avoids the page
        faults of a fresh one, which cost as much as the batch itself at 10 M queries)."""
        patterns = np.ascontiguousarray(patterns, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nq = offsets.shape[0] - 1
        if out is None:
            out = np.zeros((nq, 2), dtype=np.uint64)
        assert out.dtype == np.uint64 and out.shape == (nq, 2) and out.flags["C_CONTIGUOUS"]
        _check(self._L.gcsa2_group_find_batch(self._h, _p8(patterns), _p64(offsets), nq, _p64(out)))
        return out

    def uses_rccl(self):
        return bool(self._L.gcsa2_group_uses_rccl(self._h))

    def find_device(self, d_patterns, d_offsets, counts, d_ranges_root):
        """Shards already in HBM: d_patterns[r] / d_offsets[r] are device pointers on the device of replica r,
        d_ranges_root a device pointer on the device of replica 0 (2 x sum(counts) words); complete on return."""
        G = self.size()
        assert len(d_patterns) == G and len(d_offsets) == G and len(counts) == G
        pats = (C.c_void_p * G)(*[C.c_void_p(int(p)) for p in d_patterns])
        offs = (C.c_void_p * G)(*[C.c_void_p(int(p)) for p in d_offsets])
        cnt = np.asarray(counts, dtype=np.uint64)
        _check(self._L.gcsa2_group_find_device(self._h, pats, offs, _p64(cnt), d_ranges_root))

    def match_stats_device(self, d_patterns, d_offsets, counts, pattern_bytes, d_ms_root, d_ranges_root, d_fallbacks_root=0):
        """Matching statistics of a batch sharded over the replicas (shards in HBM as for find_device; pattern_bytes[r] =
        pattern bytes of shard r); statistics, ranges and parent() counts gathered on replica 0's device.  Complete on return."""
        G = self.size()
        pats = (C.c_void_p * G)(*[C.c_void_p(int(p)) for p in d_patterns])
        offs = (C.c_void_p * G)(*[C.c_void_p(int(p)) for p in d_offsets])
        cnt, pb = np.asarray(counts, dtype=np.uint64), np.asarray(pattern_bytes, dtype=np.uint64)
        _check(self._L.gcsa2_group_match_stats_device(self._h, pats, offs, _p64(cnt), _p64(pb), d_ms_root, d_ranges_root, d_fallbacks_root))

    def locate_device(self, d_ranges, counts, d_offsets_root, sort=True):
        """locate() of a batch of ranges sharded over the replicas; returns (job, d_values, total) on replica 0's device,
        offsets of the whole batch in d_offsets_root (sum(counts) + 1 entries).  Free with locate_discard(job)."""
        G = self.size()
        rng = (C.c_void_p * G)(*[C.c_void_p(int(p)) for p in d_ranges])
        cnt = np.asarray(counts, dtype=np.uint64)
        job, d_val, total = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(self._L.gcsa2_group_locate_device(self._h, rng, _p64(cnt), int(sort), d_offsets_root, C.byref(job), C.byref(d_val), C.byref(total)))
        return job, d_val.value, total.value

    def locate_discard(self, job):
        self._L.gcsa2_locate_discard(job)

    def close(self):
        if getattr(self, "_h", None):
            self._L.gcsa2_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def serialize_view(view_ref, what="gcsa") -> bytes:
    """GCSA::serialize / LCPArray::serialize of a host view (`gcsa2_host_view_serialize_*`) as bytes."""
    L = load_library()
    chunks = []
    sink = SINK(lambda ctx, data, n: chunks.append(C.string_at(data, n)))
    fn = L.gcsa2_host_view_serialize_gcsa if what == "gcsa" else L.gcsa2_host_view_serialize_lcp
    written = C.c_uint64()
    _check(fn(view_ref, sink, None, C.byref(written)))
    out = b"".join(chunks)
    assert len(out) == written.value
    return out


def parse_view(data: bytes, what="gcsa", exact=True):
    """`gcsa2_host_view_parse_*`: (storage handle, HostView pointer, bytes consumed); free with free_view()."""
    L = load_library()
    buf = C.create_string_buffer(data, len(data))
    h = C.c_void_p()
    consumed = C.c_uint64()
    fn = L.gcsa2_host_view_parse_gcsa if what == "gcsa" else L.gcsa2_host_view_parse_lcp
    _check(fn(buf, len(data), None if exact else C.byref(consumed), C.byref(h)))
    return h, L.gcsa2_host_view_get(h), (len(data) if exact else consumed.value)


def free_view(h):
    load_library().gcsa2_host_view_free(h)


def open_index(index_arrays, device=0):
    """(GCSA, LCPArray) over one device image; `index_arrays` may be an IndexArrays, a G2HV file or a
    `.gcsa` file with its `.lcp` beside it."""
    g = GCSA(index_arrays, device=device)
    lcp = LCPArray(g, int(g._L.gcsa2_lcp_values(g.handle)), int(g._L.gcsa2_lcp_size(g.handle)))
    return g, lcp
