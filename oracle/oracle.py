"""ctypes wrapper of the CPU oracle (oracle/gcsa_oracle.c).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, `__graft_entry__.smoke()` and the `cpu_baseline` leg of bench.py -- never by
the product package gcsa2_amd/.  Method names mirror the reference's `gcsa::GCSA` /
`gcsa::LCPArray` members so that parity tests read like the reference's own checks.
"""
import ctypes as C
import os
import subprocess
import numpy as np

from gcsa2_amd.hostview import (HostView, STNode, STNODE_DTYPE, make_host_view, concat_patterns,
                                u64p, u8p)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgcsa_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "gcsa_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, u64, i32, dbl = C.c_void_p, C.c_uint64, C.c_int, C.c_double
        L.oracle_create.restype = vp
        L.oracle_create.argtypes = [C.POINTER(HostView)]
        L.oracle_destroy.argtypes = [vp]
        L.oracle_free.argtypes = [vp]
        L.oracle_find.argtypes = [vp, u8p, u64, u64p, u64p]
        L.oracle_char_range.argtypes = [vp, C.c_uint8, u64p, u64p]
        L.oracle_lf_range.argtypes = [vp, u64p, u64p, C.c_uint8]
        L.oracle_lf_node.restype = u64
        L.oracle_lf_node.argtypes = [vp, u64]
        L.oracle_lf_all.argtypes = [vp, u64, u64, i32, u64p]
        L.oracle_count.restype = u64
        L.oracle_count.argtypes = [vp, u64, u64]
        L.oracle_locate.restype = vp
        L.oracle_locate.argtypes = [vp, u64, u64, i32, u64p]
        L.oracle_locate_max.restype = vp
        L.oracle_locate_max.argtypes = [vp, u64, u64, u64, u64p]
        L.oracle_sampled.restype = i32
        L.oracle_sampled.argtypes = [vp, u64]
        L.oracle_first_sample.restype = u64
        L.oracle_first_sample.argtypes = [vp, u64]
        L.oracle_last_sample.restype = i32
        L.oracle_last_sample.argtypes = [vp, u64]
        L.oracle_sample.restype = u64
        L.oracle_sample.argtypes = [vp, u64]
        L.oracle_parent.argtypes = [vp, u64, u64, C.POINTER(STNode)]
        L.oracle_depth.restype = u64
        L.oracle_depth.argtypes = [vp, u64, u64]
        L.oracle_sv.argtypes = [vp, i32, u64, u64p, u64p]
        L.oracle_rmq.argtypes = [vp, u64, u64, u64p, u64p]
        for name in ("oracle_find_batch",):
            getattr(L, name).restype = dbl
        L.oracle_find_batch.argtypes = [vp, u8p, u64p, u64, u64p, i32]
        L.oracle_lf_batch.restype = dbl
        L.oracle_lf_batch.argtypes = [vp, u64p, u8p, u64, u64p, i32]
        L.oracle_count_batch.restype = dbl
        L.oracle_count_batch.argtypes = [vp, u64p, u64, u64p, i32]
        L.oracle_parent_batch.restype = dbl
        L.oracle_parent_batch.argtypes = [vp, u64p, u64, vp, i32]
        L.oracle_depth_batch.restype = dbl
        L.oracle_depth_batch.argtypes = [vp, u64p, u64, u64p, i32]
        L.oracle_locate_batch.restype = dbl
        L.oracle_locate_batch.argtypes = [vp, u64p, u64, u64p, C.POINTER(vp), i32]
        L.oracle_find_traffic.argtypes = [vp, u8p, u64p, u64, u64, u64p, u64p]
        L.oracle_count_kmers.restype = u64
        L.oracle_count_kmers.argtypes = [vp, u64, i32, i32, u64, i32]
        L.oracle_match_stats_batch.restype = dbl
        L.oracle_match_stats_batch.argtypes = [vp, u8p, u64p, u64, vp, u64p, u64p, i32]
        L.oracle_compare_kmers.argtypes = [vp, vp, u64, i32, i32, u64p]
        L.oracle_compare_kmers_records.argtypes = [vp, vp, u64, i32, i32, u64p, C.POINTER(vp), C.POINTER(vp)]
        L.oracle_max_threads.restype = i32
        _lib = L
    return _lib


def _p64(a):
    return a.ctypes.data_as(u64p)


def _p8(a):
    return a.ctypes.data_as(u8p)


class OracleIndex:
    """CPU restatement of `gcsa::GCSA` + `gcsa::LCPArray` queries over one index."""

    def __init__(self, ix, **kw):
        self._holder = make_host_view(ix, **kw)
        self._h = lib().oracle_create(self._holder.ref())
        if not self._h:
            raise RuntimeError("oracle_create failed")
        self.n = int(ix.n)
        self.sigma = int(ix.sigma)
        self.char2comp = np.asarray(ix.char2comp)
        self.last_seconds = 0.0
        # what the view was made with: a query that needs a component the view left out is refused here (the C restatement
        # indexes the arrays it is given; bench.py opens find()-only oracles for its larger legs)
        self._parts = {"samples": kw.get("with_samples", True), "counters": kw.get("with_counters", True), "lcp": kw.get("with_lcp", True)}

    def _need(self, part, what):
        if not self._parts[part]:
            raise RuntimeError(f"OracleIndex: {what} needs the {part}, and this oracle was opened without them")

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- single queries -------------------------------------------------------------------
    def find(self, pattern):
        p = np.frombuffer(bytes(pattern), dtype=np.uint8)
        if p.shape[0] == 0:
            p = np.zeros(1, dtype=np.uint8)
            length = 0
        else:
            length = p.shape[0]
        sp, ep = C.c_uint64(), C.c_uint64()
        lib().oracle_find(self._h, _p8(p), length, C.byref(sp), C.byref(ep))
        return (sp.value, ep.value)

    def charRange(self, comp):
        sp, ep = C.c_uint64(), C.c_uint64()
        lib().oracle_char_range(self._h, comp, C.byref(sp), C.byref(ep))
        return (sp.value, ep.value)

    def LF(self, arg, comp=None):
        if comp is None:
            return int(lib().oracle_lf_node(self._h, int(arg)))
        sp, ep = C.c_uint64(arg[0]), C.c_uint64(arg[1])
        lib().oracle_lf_range(self._h, C.byref(sp), C.byref(ep), comp)
        return (sp.value, ep.value)

    def _lf_all(self, rng, all_):
        out = np.zeros(2 * self.sigma, dtype=np.uint64)
        lib().oracle_lf_all(self._h, rng[0], rng[1], all_, _p64(out))
        return [(int(out[2 * c]), int(out[2 * c + 1])) for c in range(self.sigma)]

    def LF_fast(self, rng):
        return self._lf_all(rng, 0)

    def LF_all(self, rng):
        return self._lf_all(rng, 1)

    def count(self, rng):
        return int(lib().oracle_count(self._h, rng[0], rng[1]))

    def _take(self, ptr, count):
        out = np.ctypeslib.as_array(C.cast(ptr, u64p), shape=(max(count, 1),))[:count].copy()
        lib().oracle_free(ptr)
        return out

    def locate(self, rng, sort=True, max_positions=None):
        self._need("samples", "locate()")
        cnt = C.c_uint64()
        if isinstance(rng, (int, np.integer)):
            rng = (int(rng), int(rng))
        if max_positions is None:
            ptr = lib().oracle_locate(self._h, rng[0], rng[1], int(sort), C.byref(cnt))
        else:
            ptr = lib().oracle_locate_max(self._h, rng[0], rng[1], max_positions, C.byref(cnt))
        return self._take(ptr, cnt.value)

    def sampled(self, node):
        return bool(lib().oracle_sampled(self._h, node))

    def firstSample(self, node):
        return int(lib().oracle_first_sample(self._h, node))

    def lastSample(self, i):
        return bool(lib().oracle_last_sample(self._h, i))

    def sample(self, i):
        return int(lib().oracle_sample(self._h, i))

    def parent(self, rng):
        node = STNode()
        lib().oracle_parent(self._h, rng[0], rng[1], C.byref(node))
        return node.astuple()

    def depth(self, rng):
        return int(lib().oracle_depth(self._h, rng[0], rng[1]))

    def _sv(self, op, pos):
        a, b = C.c_uint64(), C.c_uint64()
        lib().oracle_sv(self._h, op, pos, C.byref(a), C.byref(b))
        return (a.value, b.value)

    def psv(self, pos):
        return self._sv(0, pos)

    def psev(self, pos):
        return self._sv(1, pos)

    def nsv(self, pos):
        return self._sv(2, pos)

    def nsev(self, pos):
        return self._sv(3, pos)

    def rmq(self, sp, ep):
        a, b = C.c_uint64(), C.c_uint64()
        lib().oracle_rmq(self._h, sp, ep, C.byref(a), C.byref(b))
        return (a.value, b.value)

    # ---- batches (numpy in / out) ------------------------------------------------------------
    def find_batch(self, patterns, offsets, threads=1):
        nq = offsets.shape[0] - 1
        out = np.zeros((nq, 2), dtype=np.uint64)
        self.last_seconds = lib().oracle_find_batch(self._h, _p8(patterns), _p64(offsets), nq,
                                                    _p64(out), threads)
        return out

    def lf_batch(self, ranges, comps, threads=1):
        ranges = np.ascontiguousarray(ranges, dtype=np.uint64)
        comps = np.ascontiguousarray(comps, dtype=np.uint8)
        out = np.zeros_like(ranges)
        self.last_seconds = lib().oracle_lf_batch(self._h, _p64(ranges), _p8(comps),
                                                  ranges.shape[0], _p64(out), threads)
        return out

    def count_batch(self, ranges, threads=1):
        self._need("counters", "count_batch()")
        ranges = np.ascontiguousarray(ranges, dtype=np.uint64)
        out = np.zeros(ranges.shape[0], dtype=np.uint64)
        self.last_seconds = lib().oracle_count_batch(self._h, _p64(ranges), ranges.shape[0],
                                                     _p64(out), threads)
        return out

    def parent_batch(self, ranges, threads=1):
        ranges = np.ascontiguousarray(ranges, dtype=np.uint64)
        out = np.zeros(ranges.shape[0], dtype=STNODE_DTYPE)
        self.last_seconds = lib().oracle_parent_batch(self._h, _p64(ranges), ranges.shape[0],
                                                      out.ctypes.data, threads)
        return out

    def depth_batch(self, ranges, threads=1):
        ranges = np.ascontiguousarray(ranges, dtype=np.uint64)
        out = np.zeros(ranges.shape[0], dtype=np.uint64)
        self.last_seconds = lib().oracle_depth_batch(self._h, _p64(ranges), ranges.shape[0],
                                                     _p64(out), threads)
        return out

    def locate_batch(self, ranges, threads=1):
        self._need("samples", "locate_batch()")
        ranges = np.ascontiguousarray(ranges, dtype=np.uint64)
        nq = ranges.shape[0]
        offsets = np.zeros(nq + 1, dtype=np.uint64)
        ptr = C.c_void_p()
        self.last_seconds = lib().oracle_locate_batch(self._h, _p64(ranges), nq, _p64(offsets),
                                                      C.byref(ptr), threads)
        values = self._take(ptr.value, int(offsets[nq]))
        return offsets, values

    def count_kmers(self, k, include_Ns=False, force=False, threads=1):
        """`countKMers` (reference src/algorithms.cpp:387-421)."""
        return int(lib().oracle_count_kmers(self._h, k, int(include_Ns), int(force), 5, threads))

    def match_stats_batch(self, patterns, offsets, threads=1):
        """Matching statistics by LF + parent: (ms uint16[total bytes], ranges (nq, 2), fallbacks)."""
        nq = offsets.shape[0] - 1
        ms = np.zeros(max(int(offsets[nq]), 1), dtype=np.uint16)
        ranges = np.zeros((nq, 2), dtype=np.uint64)
        fallbacks = np.zeros(nq, dtype=np.uint64)
        self.last_seconds = lib().oracle_match_stats_batch(self._h, _p8(patterns), _p64(offsets), nq, ms.ctypes.data,
                                                           _p64(ranges), _p64(fallbacks), threads)
        return ms[: int(offsets[nq])], ranges, fallbacks

    def compare_kmers(self, other, k, include_Ns=False, force=False):
        """`compareKMers(self, other, k)` (reference src/algorithms.cpp:534-616): (shared, left, right)."""
        out = np.zeros(3, dtype=np.uint64)
        lib().oracle_compare_kmers(self._h, other._h, k, int(include_Ns), int(force), _p64(out))
        return tuple(int(x) for x in out)

    def compare_kmers_records(self, other, k, include_Ns=False, force=False):
        """compareKMers with parameters.output set: (counts, left_states, right_states); a state is the
        8-u64 KMerComparisonState the reference writes to output.left / output.right."""
        out = np.zeros(3, dtype=np.uint64)
        lp, rp = C.c_void_p(), C.c_void_p()
        lib().oracle_compare_kmers_records(self._h, other._h, k, int(include_Ns), int(force), _p64(out), C.byref(lp), C.byref(rp))
        res = []
        for ptr, cnt in ((lp, int(out[1])), (rp, int(out[2]))):
            if ptr.value:
                res.append(np.ctypeslib.as_array(C.cast(ptr, u64p), shape=(cnt * 8,)).copy().reshape(cnt, 8))
                lib().oracle_free(ptr)
            else:
                res.append(np.zeros((0, 8), dtype=np.uint64))
        return tuple(int(x) for x in out), res[0], res[1]

    def find_traffic(self, patterns, offsets, block_bits):
        blocks, steps = C.c_uint64(), C.c_uint64()
        lib().oracle_find_traffic(self._h, _p8(patterns), _p64(offsets), offsets.shape[0] - 1,
                                  block_bits, C.byref(blocks), C.byref(steps))
        return blocks.value, steps.value


def max_threads():
    """Threads the oracle should be timed with: what OpenMP would start, capped by the CPU affinity mask
    and by the cgroup CPU quota of the container (a box with 256 logical CPUs and a 16-CPU quota runs 128
    threads slower than 16)."""
    import math
    import os
    n = int(lib().oracle_max_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())           # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return max(1, n)
