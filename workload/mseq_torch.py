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

from .graphs import SIGMA, FAST_CHARS, default_char2comp
from .index_arrays import IndexArrays
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


def build_mseq(degree: int, device=None, verbose=None):
    """Returns (IndexArrays, sym tensor on `device`, rank numpy uint32)."""
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
    ix = IndexArrays(
        n=N, e=N, order=degree // 2, sigma=SIGMA, fast_chars=FAST_CHARS, char2comp=default_char2comp(), C=Carr,
        bwt=bwt, edges=edges, sampled_paths=zero, sample_count=0, sample_width=1,
        stored_samples=np.zeros(2, dtype=np.uint64), stored_samples_plain=np.zeros(0, dtype=np.uint64),
        samples=np.zeros(2, dtype=np.uint64), extra_filter=zero, extra_values_len=0,
        extra_values=np.zeros(2, dtype=np.uint64), redundant_len=0, redundant=np.zeros(2, dtype=np.uint64),
        lcp_size=0, lcp_branching=64, lcp_offsets=np.zeros(2, dtype=np.uint64), lcp_data=np.zeros(1, dtype=np.uint8),
        table=None)
    return ix, sym_t, rank


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
