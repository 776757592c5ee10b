"""Footprint-scale GCSA of a linear graph, built with torch ops (GPU when available).

A linear graph `#` + n random bases + `$` whose order-K paths are all distinct has one path node
per position, sorted like the rotations of the cyclic text (the artificial edge `$` -> `#` closes
it): its GCSA degenerates to an FM-index -- every out-degree is 1, B_c is the indicator of the BWT
character, one value per node.  That is what this module builds, by prefix
doubling with `torch.unique` (a sort per round), so that indexes far larger than the 256 MiB
Infinity Cache can be produced on the GPU box in seconds.  It is workload generation, not product
code; for small n it is cross-checked field by field against the general builder
(tests/test_workload.py).

Positions and values follow workload.graphs.linear_graph: position 0 = `#`, 1..n = backbone,
n+1 = `$`; vg-style ids (32 bases per node), source id = last + 1, sink id = last + 2.
"""
import numpy as np
import torch

from .graphs import SIGMA, FAST_CHARS, ID_OFFSET, default_char2comp
from .index_arrays import IndexArrays, bit_length, build_lcp_tree

_M = (1 << 64) - 1


def _s64(x: int) -> int:
    """Python int (mod 2^64) -> the same bit pattern as a signed 64-bit value."""
    x &= _M
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(z: torch.Tensor, k: int) -> torch.Tensor:
    """Logical right shift of int64 tensors (torch's >> is arithmetic)."""
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix64_torch(seed: int, count: int, device) -> torch.Tensor:
    """First `count` outputs of SplitMix64(seed) as int64 bit patterns (== workload.rng)."""
    idx = torch.arange(1, count + 1, dtype=torch.int64, device=device)
    z = idx * _s64(0x9E3779B97F4A7C15) + _s64(seed)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def random_bases_torch(n: int, seed: int, device) -> torch.Tensor:
    """== workload.graphs.random_bases: comps 1..4."""
    out = torch.empty(n, dtype=torch.uint8, device=device)
    chunk = 1 << 26
    for b in range(0, n, chunk):
        e = min(n, b + chunk)
        idx = torch.arange(b + 1, e + 1, dtype=torch.int64, device=device)
        z = idx * _s64(0x9E3779B97F4A7C15) + _s64(seed)
        z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
        z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
        z = z ^ _lsr(z, 31)
        out[b:e] = ((_lsr(z, 33) % 4) + 1).to(torch.uint8)
    return out


def pack_bits_torch(bits: torch.Tensor, pad_words: int = 1) -> np.ndarray:
    """bool[n] -> uint64 words (LSB first), as numpy on the host."""
    n = bits.shape[0]
    nwords = (n + 63) // 64
    weights = (torch.ones(64, dtype=torch.int64, device=bits.device) << torch.arange(64, device=bits.device))
    out = torch.zeros(nwords + pad_words, dtype=torch.int64, device=bits.device)
    chunk = 1 << 24   # words per chunk
    for w0 in range(0, nwords, chunk):
        w1 = min(nwords, w0 + chunk)
        seg = bits[w0 * 64: min(n, w1 * 64)]
        if seg.shape[0] < (w1 - w0) * 64:
            seg = torch.cat([seg, torch.zeros((w1 - w0) * 64 - seg.shape[0], dtype=seg.dtype, device=seg.device)])
        out[w0:w1] = (seg.view(-1, 64).to(torch.int64) * weights).sum(dim=1)
    return out.cpu().numpy().view(np.uint64)


def build_linear(n: int, seed: int, order: int = 256, node_len: int = 32, sample_period: int = 64,
                 branching: int = 64, device=None, with_lcp: bool = True, with_samples: bool = True,
                 verbose=None, sequence=None) -> IndexArrays:
    """`sequence`: the n backbone comps (1..4) as a tensor instead of random_bases_torch(n, seed) -- e.g. the repeat-rich
    text of workload/repeats_torch.py; every order-`order` path must still be distinct."""
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    N = n + 2
    assert N < (1 << 31), "ranks are kept below 2^31 so that rank pairs fit 62 bits"
    assert sample_period == 64 and node_len <= 64

    def log(msg):
        if verbose:
            verbose(msg)

    text = torch.empty(N, dtype=torch.uint8, device=device)
    text[0] = 6
    text[1:n + 1] = random_bases_torch(n, seed, device) if sequence is None else sequence.to(device)
    text[n + 1] = 0

    # ---- prefix doubling over the cyclic text -------------------------------------------------
    pos = torch.arange(N, dtype=torch.int64, device=device)
    rank = text.to(torch.int64)
    levels = []           # rank arrays: level l orders rotations by their first 2^l characters
    h = 1
    rank = torch.unique(rank, return_inverse=True)[1]
    if with_lcp:
        levels.append(rank.to(torch.int32))
    distinct = int(rank.max().item()) + 1
    while distinct < N:
        if h >= order:
            raise RuntimeError(f"paths of length {order} are not all distinct (seed {seed}); "
                               "this generator only handles the all-unique case")
        nxt = rank[(pos + h) % N]
        key = rank * distinct + nxt
        del nxt
        rank = torch.unique(key, return_inverse=True)[1]
        del key
        h *= 2
        if with_lcp:
            levels.append(rank.to(torch.int32))
        distinct = int(rank.max().item()) + 1
        log(f"doubling: {h} characters -> {distinct} / {N} distinct")
    sa = torch.empty(N, dtype=torch.int64, device=device)
    sa[rank] = pos
    del rank

    # ---- BWT indicators, C, edges ---------------------------------------------------------------
    bwt_comp = text[(sa - 1) % N]
    counts = torch.bincount(bwt_comp.to(torch.int64), minlength=SIGMA).cpu().numpy().astype(np.uint64)
    C = np.zeros(SIGMA + 1, dtype=np.uint64)
    C[1:] = np.cumsum(counts)
    bwt = [pack_bits_torch(bwt_comp == c) for c in range(SIGMA)]
    del bwt_comp
    ones = torch.ones(N, dtype=torch.bool, device=device)
    edges = pack_bits_torch(ones)

    # ---- values and samples: value = id << 11 | offset, sampled iff offset == 0 -----------------
    if with_samples:
        nid = (n + node_len - 1) // node_len
        b = sa - 1                                           # backbone index of a position
        value = (((b // node_len) + 1) << ID_OFFSET) | (b % node_len)
        value = torch.where(sa == 0, torch.full_like(value, (nid + 1) << ID_OFFSET), value)
        value = torch.where(sa == n + 1, torch.full_like(value, (nid + 2) << ID_OFFSET), value)
        sampled = (value & ((1 << ID_OFFSET) - 1)) == 0
        stored = value[sampled]
        S = int(stored.shape[0])
        width = bit_length(int(stored.max().item()))
        stored_np = stored.cpu().numpy().astype(np.uint64)
        from . import builder as _b          # C helper for packing large int vectors
        lib = _b._load()
        packed = np.zeros((S * width + 63) // 64 + 2, dtype=np.uint64)
        lib.gcsa_pack_ints(stored_np.ctypes.data, S, width, packed.ctypes.data)
        sampled_bits = pack_bits_torch(sampled)
        samples_bits = pack_bits_torch(torch.ones(S, dtype=torch.bool, device=device))
        del value, sampled, stored
    else:
        S, width = 0, 1
        stored_np = np.zeros(0, dtype=np.uint64)
        packed = np.zeros(2, dtype=np.uint64)
        sampled_bits = np.zeros(N // 64 + 2, dtype=np.uint64)
        samples_bits = np.zeros(2, dtype=np.uint64)

    # ---- LCP of adjacent rotations from the stored rank levels ----------------------------------
    if with_lcp:
        a = sa[:-1].clone()
        bpos = sa[1:].clone()
        lcp = torch.zeros(N - 1, dtype=torch.int64, device=device)
        for lev in range(len(levels) - 1, -1, -1):
            r = levels[lev]
            same = r[a % N] == r[bpos % N]
            step = same.to(torch.int64) << lev
            lcp += step
            a += step
            bpos += step
        lcp_full = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), lcp]).clamp(max=255)
        lcp_np = lcp_full.to(torch.uint8).cpu().numpy()
        lcp_data, lcp_offsets = build_lcp_tree(lcp_np, branching)
    else:
        lcp_data = np.zeros(1, dtype=np.uint8)
        lcp_offsets = np.zeros(2, dtype=np.uint64)
    del levels, sa

    zeros_n = np.zeros(N // 64 + 2, dtype=np.uint64)
    red = pack_bits_torch(torch.ones(N - 1, dtype=torch.bool, device=device))
    return IndexArrays(
        n=N, e=N, order=order, sigma=SIGMA, fast_chars=FAST_CHARS, char2comp=default_char2comp(), C=C,
        bwt=bwt, edges=edges, sampled_paths=sampled_bits, sample_count=S, sample_width=width,
        stored_samples=packed, stored_samples_plain=stored_np, samples=samples_bits,
        extra_filter=zeros_n, extra_values_len=0, extra_values=np.zeros(2, dtype=np.uint64),
        redundant_len=N - 1, redundant=red,
        lcp_size=N if with_lcp else 0, lcp_branching=branching, lcp_offsets=lcp_offsets,
        lcp_data=np.ascontiguousarray(lcp_data), table=None)


def substring_patterns_torch(n: int, seed: int, nq: int, m: int, pat_seed: int, device=None) -> np.ndarray:
    """(nq, m) bytes: substrings of the backbone at splitmix64 positions (full-depth matches)."""
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    seq = random_bases_torch(n, seed, device)
    r = splitmix64_torch(pat_seed, nq, device)
    start = _lsr(r, 11) % (n - m)
    idx = start.view(-1, 1) + torch.arange(m, dtype=torch.int64, device=device).view(1, -1)
    comps = seq[idx]
    lut = torch.tensor(list(b"$ACGTN#"), dtype=torch.uint8, device=device)
    return lut[comps.to(torch.int64)].cpu().numpy()
