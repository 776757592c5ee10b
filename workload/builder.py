"""ctypes front end of workload/builder.cpp (scalable path-node table construction)."""
import ctypes as C
import os
import subprocess
import numpy as np

from .graphs import Graph
from .index_arrays import NodeTable, assemble

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "builder.cpp")
_LIB = os.path.join(_HERE, "_build", "libgcsa2_workload.so")
_lib = None


class _Table(C.Structure):
    _fields_ = [("n", C.c_uint64), ("total_vals", C.c_uint64),
                ("pred_mask", C.POINTER(C.c_uint8)), ("outdeg", C.POINTER(C.c_uint32)),
                ("lcp", C.POINTER(C.c_uint8)), ("val_off", C.POINTER(C.c_uint64)),
                ("vals", C.POINTER(C.c_uint64)), ("redundant", C.POINTER(C.c_uint32)),
                ("key_len", C.POINTER(C.c_uint16)),
                ("trie_nodes", C.c_uint64), ("max_members", C.c_uint64), ("seconds", C.c_double)]


def build_library(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        subprocess.check_call(["g++", "-O3", "-march=x86-64-v3", "-std=c++17", "-fopenmp", "-fPIC",
                               "-shared", "-Wall", "-o", _LIB, _SRC])
    return _LIB


def _load():
    global _lib
    if _lib is None:
        build_library()
        L = C.CDLL(_LIB)
        L.gcsa_build_nodes.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_uint64, C.c_int, C.c_int, C.POINTER(_Table)]
        L.gcsa_build_free.argtypes = [C.POINTER(_Table)]
        L.gcsa_pack_ints.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        _lib = L
    return _lib


def node_table(graph: Graph, K: int, threads: int = 0, verbose: bool = False) -> NodeTable:
    L = _load()
    comp = np.ascontiguousarray(graph.comp, dtype=np.uint8)
    value = np.ascontiguousarray(graph.value, dtype=np.uint64)
    soff = np.ascontiguousarray(graph.succ_off, dtype=np.uint64)
    succ = np.ascontiguousarray(graph.succ, dtype=np.uint32)
    t = _Table()
    rc = L.gcsa_build_nodes(graph.size, comp.ctypes.data, value.ctypes.data, soff.ctypes.data,
                            succ.ctypes.data, K, threads, int(verbose), C.byref(t))
    if rc != 0:
        raise RuntimeError(f"gcsa_build_nodes failed with {rc}")
    n = int(t.n)

    def take(ptr, count, dtype):
        return np.ctypeslib.as_array(ptr, shape=(max(count, 1),))[:count].astype(dtype, copy=True)

    table = NodeTable(order=K,
                      pred_mask=take(t.pred_mask, n, np.uint8), outdeg=take(t.outdeg, n, np.uint32),
                      lcp=take(t.lcp, n, np.uint8), val_off=take(t.val_off, n + 1, np.uint64),
                      vals=take(t.vals, int(t.total_vals), np.uint64),
                      redundant=take(t.redundant, max(n - 1, 0), np.uint32),
                      key_len=take(t.key_len, n, np.uint16))
    table.stats = {"trie_nodes": int(t.trie_nodes), "max_members": int(t.max_members),
                   "seconds": float(t.seconds)}
    L.gcsa_build_free(C.byref(t))
    return table


def build(graph: Graph, K: int, sample_period: int = 64, branching: int = 64, threads: int = 0,
          verbose: bool = False, keep_table: bool = True):
    return assemble(node_table(graph, K, threads, verbose), sample_period=sample_period,
                    branching=branching, keep_table=keep_table)
