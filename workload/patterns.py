"""Seeded query sets.

  set "S": substrings spelled by random walks through the input graph -> match to full depth
           (mirrors the `vg sim` methodology of paper.tex:395); primary number.
  set "U": uniform random ACGT strings -> most die after ~log4(n) steps.
Patterns of one set all have the same length m, so the concatenated buffer is (nq, m) bytes and
offsets are q * m.
"""
import numpy as np

from .graphs import COMP2CHAR
from .rng import splitmix64_array

_C2B = np.frombuffer(COMP2CHAR, dtype=np.uint8)


def walk_patterns(graph, nq: int, m: int, seed: int) -> np.ndarray:
    """(nq, m) uint8 bytes.  Starts are uniform over positions from which an m-step walk cannot
    reach the sink (positions 1 .. source-side backbone), successors are chosen uniformly."""
    N = graph.size
    backbone = graph.sink - 1  # positions 1..backbone are the backbone in workload.graphs layouts
    limit = max(1, backbone - m - 1)
    r = splitmix64_array(seed, nq * 2)
    cur = (1 + (r[:nq] >> np.uint64(11)) % np.uint64(limit)).astype(np.int64)
    choice = r[nq:]
    out = np.empty((nq, m), dtype=np.uint8)
    soff = graph.succ_off.astype(np.int64)
    succ = graph.succ
    comp = graph.comp
    for i in range(m):
        out[:, i] = _C2B[comp[cur]]
        begin = soff[cur]
        deg = soff[cur + 1] - begin
        pick = ((choice >> np.uint64((i * 2) % 60)) % deg.astype(np.uint64)).astype(np.int64)
        # fresh randomness every 30 steps
        if (i * 2) % 60 == 58:
            choice = splitmix64_array(seed ^ (0x9E37 * (i + 1)), nq)
        cur = succ[begin + pick].astype(np.int64)
    return out


def uniform_patterns(nq: int, m: int, seed: int) -> np.ndarray:
    r = splitmix64_array(seed, nq * ((m + 31) // 32))
    r = r.reshape(nq, -1)
    out = np.empty((nq, m), dtype=np.uint8)
    for i in range(m):
        out[:, i] = _C2B[1 + ((r[:, i // 32] >> np.uint64(2 * (i % 32))) & np.uint64(3)).astype(np.int64)]
    return out


def as_batch(pats: np.ndarray):
    """(nq, m) -> (flat bytes, offsets)."""
    nq, m = pats.shape
    offsets = np.arange(nq + 1, dtype=np.uint64) * np.uint64(m)
    return np.ascontiguousarray(pats.reshape(-1)), offsets
