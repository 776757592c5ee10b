"""Whole-genome-footprint index without suffix sorting.

A binary m-sequence of even degree d read two bits at a time is a cyclic text over {A,C,G,T} of
length N = 2^d - 1 in which every d/2-mer except A^(d/2) occurs exactly once, so the rank of every
rotation is known in closed form (workload/builder.cpp: gcsa_mseq_text).  The GCSA of that cyclic
graph (one cycle of N positions, every order-d/2 path label unique, all out-degrees 1, no `$` / `#`)
is an FM-index shaped index with N path nodes that can be written down directly: degree 32 gives
4.29 G path nodes -- the size of the whole-human indexes of the paper (paper.tex:378-380) -- in
about a minute, with an analytic answer for every query: find(T[p .. p + m)) = (rank[p], rank[p])
for m >= d/2.  Workload generation only; find()-only (no samples / counters / LCP).
"""
import ctypes as C

import numpy as np
import torch

from .graphs import SIGMA, FAST_CHARS, ID_OFFSET, default_char2comp
from .index_arrays import IndexArrays, bit_length, build_lcp_tree
from .linear_torch import pack_bits_torch, splitmix64_torch, _lsr

# primitive polynomials x^d + ... + 1 as tap lists (verified at run time by gcsa_mseq_text)
TAPS = {8: [8, 6, 5, 4], 10: [10, 7], 12: [12, 6, 4, 1], 16: [16, 15, 13, 4], 20: [20, 17],
        24: [24, 23, 22, 17], 28: [28, 25], 32: [32, 22, 2, 1]}


def mseq_text(degree: int):
    """(sym uint8[N] in 0..3, rank uint32[N]) of the cyclic text."""
    from . import builder
    lib = builder._load()
    lib.gcsa_mseq_text.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p]
    taps = TAPS[degree]
    N = (1 << degree) - 1
    sym = np.empty(N, dtype=np.uint8)
    rank = np.empty(N, dtype=np.uint32)
    rc = lib.gcsa_mseq_text(degree, (C.c_int * len(taps))(*taps), len(taps), sym.ctypes.data, rank.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"gcsa_mseq_text({degree}) failed with {rc}")
    return sym, rank


NODE_LEN = 32      # vg-style nodes of 32 positions: value = (p // 32 + 1) << 11 | p % 32


def node_values(pos: np.ndarray) -> np.ndarray:
    pos = pos.astype(np.uint64)
    return ((pos // np.uint64(NODE_LEN) + np.uint64(1)) << np.uint64(ID_OFFSET)) | (pos % np.uint64(NODE_LEN))


def build_mseq(degree: int, device=None, verbose=None, full: bool = False, branching: int = 64):
    """Returns (IndexArrays, sym tensor on `device`, rank numpy uint32).

    full = True also derives, in closed form, the samples (position p carries the value
    node_values(p); a node is sampled iff p % 32 == 0, which is what the rules of
    src/gcsa.cpp:621-646 give for these values), the counters (one value per node: A = 0, R = 0)
    and the LCP array: adjacent rotations j-1, j hold the k-mer values j, j+1, so their common
    prefix is k - 1 - (number of trailing base-4 digits of j equal to 3)."""
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    sym, rank = mseq_text(degree)
    N = sym.shape[0]
    if verbose:
        verbose(f"m-sequence text: {N} symbols")
    sym_t = torch.from_numpy(sym).to(device)
    bwt_sorted = torch.empty(N, dtype=torch.uint8, device=device)
    chunk = 1 << 27
    for b in range(0, N, chunk):
        e = min(N, b + chunk)
        idx = torch.from_numpy(rank[b:e].view(np.int32)).to(device).to(torch.int64) & 0xFFFFFFFF
        # BWT character of rotation i = symbol preceding position i (cyclically)
        prev = sym_t[b - 1:e - 1] if b > 0 else torch.cat([sym_t[N - 1:], sym_t[:e - 1]])
        bwt_sorted.index_put_((idx,), prev)
        del idx, prev
    if verbose:
        verbose("BWT scattered into rotation order")
    counts = np.zeros(4, dtype=np.int64)
    for b in range(0, N, chunk):          # chunked: bincount wants int64 input
        counts += torch.bincount(bwt_sorted[b:b + chunk].to(torch.int64), minlength=4).cpu().numpy()
    if verbose:
        verbose(f"character counts {counts.tolist()}")
    Carr = np.zeros(SIGMA + 1, dtype=np.uint64)
    per_comp = np.zeros(SIGMA, dtype=np.uint64)
    per_comp[1:5] = counts
    Carr[1:] = np.cumsum(per_comp)
    zero = np.zeros((N + 63) // 64 + 1, dtype=np.uint64)
    bwt = [zero] + [pack_bits_torch(bwt_sorted == s) for s in range(4)] + [zero, zero]
    del bwt_sorted
    edges = pack_bits_torch(torch.ones(N, dtype=torch.bool, device=device))
    if verbose:
        verbose("B_c and edges packed")
    extras = dict(sampled_paths=zero, sample_count=0, sample_width=1,
                  stored_samples=np.zeros(2, dtype=np.uint64), stored_samples_plain=np.zeros(0, dtype=np.uint64),
                  samples=np.zeros(2, dtype=np.uint64), extra_filter=zero, extra_values_len=0,
                  extra_values=np.zeros(2, dtype=np.uint64), redundant_len=0, redundant=np.zeros(2, dtype=np.uint64),
                  lcp_size=0, lcp_branching=branching, lcp_offsets=np.zeros(2, dtype=np.uint64),
                  lcp_data=np.zeros(1, dtype=np.uint8))
    if full:
        k = degree // 2
        # samples: positions p = 0, 32, 64, ... in rank order
        spos = np.arange(0, N, NODE_LEN, dtype=np.int64)
        srank = rank[spos].astype(np.int64)
        order = np.argsort(srank, kind="stable")
        stored = node_values(spos[order])
        S = int(stored.shape[0])
        width = bit_length(int(stored.max()))
        from . import builder as _b
        lib = _b._load()
        lib.gcsa_pack_ints.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        packed = np.zeros((S * width + 63) // 64 + 2, dtype=np.uint64)
        lib.gcsa_pack_ints(stored.ctypes.data, S, width, packed.ctypes.data)
        sampled = torch.zeros(N, dtype=torch.bool, device=device)
        sampled[torch.from_numpy(srank).to(device)] = True
        # LCP bytes in closed form
        lcp = torch.empty(N, dtype=torch.uint8, device=device)
        for b in range(0, N, chunk):
            e = min(N, b + chunk)
            j = torch.arange(b, e, dtype=torch.int64, device=device)
            q = torch.zeros(e - b, dtype=torch.int64, device=device)
            run = torch.ones(e - b, dtype=torch.bool, device=device)
            for m in range(k):
                run &= ((j >> (2 * m)) & 3) == 3
                q += run.to(torch.int64)
            lcp[b:e] = (k - 1 - q).clamp(min=0).to(torch.uint8)
        lcp[0] = 0
        lcp_data, lcp_offsets = build_lcp_tree(lcp.cpu().numpy(), branching)
        extras = dict(sampled_paths=pack_bits_torch(sampled), sample_count=S, sample_width=width,
                      stored_samples=packed, stored_samples_plain=stored,
                      samples=pack_bits_torch(torch.ones(S, dtype=torch.bool, device=device)),
                      extra_filter=zero, extra_values_len=0, extra_values=np.zeros(2, dtype=np.uint64),
                      redundant_len=N - 1, redundant=pack_bits_torch(torch.ones(N - 1, dtype=torch.bool, device=device)),
                      lcp_size=N, lcp_branching=branching, lcp_offsets=lcp_offsets,
                      lcp_data=np.ascontiguousarray(lcp_data))
        del sampled, lcp
        if verbose:
            verbose(f"samples ({S}), counters and LCP derived in closed form")
    ix = IndexArrays(n=N, e=N, order=degree // 2, sigma=SIGMA, fast_chars=FAST_CHARS, char2comp=default_char2comp(),
                     C=Carr, bwt=bwt, edges=edges, table=None, **extras)
    return ix, sym_t, rank


def cycle_graph(degree: int):
    """The input graph of the m-sequence text: one cycle of N positions (no source / sink)."""
    from .graphs import Graph
    sym, rank = mseq_text(degree)
    N = sym.shape[0]
    succ = np.roll(np.arange(N, dtype=np.uint32), -1)
    return Graph(comp=(sym + 1).astype(np.uint8), value=node_values(np.arange(N)), succ_off=np.arange(N + 1, dtype=np.uint64),
                 succ=succ, source=0, sink=N - 1)


def substring_patterns(sym_t: torch.Tensor, rank: np.ndarray, nq: int, m: int, seed: int):
    """(patterns (nq, m) uint8 bytes, expected uint64 (nq, 2)): substrings of the cyclic text at
    splitmix64 positions and their analytic find() result (valid for m >= degree / 2)."""
    device = sym_t.device
    N = sym_t.shape[0]
    r = splitmix64_torch(seed, nq, device)
    start = _lsr(r, 11) % N
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty((nq, m), dtype=torch.uint8, device=device)
    for j in range(m):
        out[:, j] = lut[sym_t[(start + j) % N].to(torch.int64)]
    pos = start.cpu().numpy()
    exp = rank[pos].astype(np.uint64)
    return out.cpu().numpy(), np.stack([exp, exp], axis=1)


def splitmix64_range_torch(seed: int, first: int, count: int, device) -> torch.Tensor:
    """Outputs first .. first + count - 1 (0-based) of SplitMix64(seed) as int64 bit patterns."""
    from .linear_torch import _s64
    idx = torch.arange(first + 1, first + count + 1, dtype=torch.int64, device=device)
    z = idx * _s64(0x9E3779B97F4A7C15) + _s64(seed)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def substring_patterns_device(sym_t: torch.Tensor, first: int, count: int, m: int, seed: int):
    """Queries first .. first + count - 1 of the global batch `seed`, kept on the device: (patterns (count, m)
    uint8 bytes, start positions int64).  Query q starts at position (splitmix64(seed)[q] >> 11) % N of the
    cyclic text, so a shard of the batch is the same whichever rank generates it."""
    device = sym_t.device
    N = sym_t.shape[0]
    start = _lsr(splitmix64_range_torch(seed, first, count, device), 11) % N
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty((count, m), dtype=torch.uint8, device=device)
    chunk = 1 << 24
    for b in range(0, count, chunk):
        e = min(count, b + chunk)
        idx = (start[b:e].view(-1, 1) + torch.arange(m, dtype=torch.int64, device=device).view(1, -1)) % N
        out[b:e] = lut[sym_t[idx].to(torch.int64)]
        del idx
    return out, start
