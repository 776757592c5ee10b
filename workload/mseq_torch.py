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
from .linear_torch import pack_bits_torch, splitmix64_torch, _lsr, _s64

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
        lcp_data, lcp_offsets = mseq_lcp(degree, device, branching)
        extras = dict(sampled_paths=pack_bits_torch(sampled), sample_count=S, sample_width=width,
                      stored_samples=packed, stored_samples_plain=stored,
                      samples=pack_bits_torch(torch.ones(S, dtype=torch.bool, device=device)),
                      extra_filter=zero, extra_values_len=0, extra_values=np.zeros(2, dtype=np.uint64),
                      redundant_len=N - 1, redundant=pack_bits_torch(torch.ones(N - 1, dtype=torch.bool, device=device)),
                      lcp_size=N, lcp_branching=branching, lcp_offsets=lcp_offsets,
                      lcp_data=np.ascontiguousarray(lcp_data))
        del sampled
        if verbose:
            verbose(f"samples ({S}), counters and LCP derived in closed form")
    ix = IndexArrays(n=N, e=N, order=degree // 2, sigma=SIGMA, fast_chars=FAST_CHARS, char2comp=default_char2comp(),
                     C=Carr, bwt=bwt, edges=edges, table=None, **extras)
    return ix, sym_t, rank


def mseq_lcp(degree: int, device, branching: int = 64):
    """(LCP bytes + range-minimum tree, level offsets) of the path nodes = all k-mers but A^k in lexicographic order:
    adjacent nodes j - 1, j hold the k-mer values j, j + 1, whose common prefix is k - 1 - (number of trailing base-4
    digits of j equal to 3).  Depends on the node set only, so the branching variant (same nodes, more edges) shares it."""
    k = degree // 2
    N = (1 << degree) - 1
    chunk = 1 << 27
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
    return build_lcp_tree(lcp.cpu().numpy(), branching)


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


# ---- the same text with SNP bubbles: a branching whole-genome-footprint index ------------------------------
# Input graph: the cycle of the m-sequence text plus, every `period` positions on average, an alternative base
# (a two-node bubble).  Its order-k de Bruijn graph (k = degree / 2) has the same path nodes as the plain text's --
# every k-mer but A^k is there already -- and k + 1 additional (k + 1)-mers per SNP (the windows covering the
# alternative base).  About a quarter of them are new edges (the text obeys a linear recurrence, so a window whose
# changed symbol misses the recurrence's taps is a (k + 1)-mer the text already has): e ~ (1 + 4 / period) n, i.e.
# 1.08 n at one SNP per 50 positions, with out-degrees and in-degrees above one wherever an alternative k-mer
# leaves or rejoins the text's k-mers (anywhere in the index).  An unpruned order-k de Bruijn graph is a valid GCSA path graph (every
# node is a k-mer; src/gcsa.cpp prunes only to save space), and find() has a closed form on it: a pattern
# spelled by a walk through the graph (every (k + 1)-mer of it is an edge) ends in the single node of its first
# k characters, find(P) = (v - 1, v - 1) with v the base-4 value of P[0 .. k).

SNP_SEED = 0x6C5A0042


def snp_sites(N: int, k: int, period: int, device):
    """(positions int64, alternative symbols uint8): one site per `period` positions, at least 2 k + 2 apart (so that
    a window of k + 1 symbols covers at most one), alternative base != reference base chosen by SplitMix64."""
    count = N // period - 1
    i = torch.arange(count, dtype=torch.int64, device=device)
    r = splitmix64_range_torch(SNP_SEED, 0, count, device)
    room = period - 2 * (k + 1)
    assert room >= 1
    pos = i * period + (k + 1) + _lsr(r, 11) % room
    shift = 1 + _lsr(r, 7) % 3                         # 1..3: added to the reference symbol mod 4
    return pos, shift.to(torch.uint8)


def build_mseq_snp(degree: int, period: int = 50, device=None, verbose=None, with_lcp: bool = False, branching: int = 64):
    """Returns (IndexArrays [no samples / counters; the LCP array with with_lcp -- the path nodes are those of the plain
    text, so it is mseq_lcp()], sym tensor, rank numpy uint32, alt tensor uint8[N]: the alternative symbol at a SNP
    site, 255 elsewhere)."""
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    sym, rank = mseq_text(degree)
    N, k = sym.shape[0], degree // 2
    sym_t = torch.from_numpy(sym).to(device)
    rank_t = torch.from_numpy(rank.view(np.int32)).to(device)          # value of the k-mer at p = rank + 1
    chunk = 1 << 27

    def value_at(p):                                   # base-4 value of T[p .. p + k), p int64 (any shape), 1 .. 4^k - 1
        return (rank_t[p % N].to(torch.int64) & 0xFFFFFFFF) + 1

    # B_c of the plain text: the node of rotation p has the predecessor label T[p - 1]
    B = torch.zeros((4, N), dtype=torch.bool, device=device)
    for b in range(0, N, chunk):
        e = min(N, b + chunk)
        idx = rank_t[b:e].to(torch.int64) & 0xFFFFFFFF
        prev = sym_t[b - 1:e - 1] if b > 0 else torch.cat([sym_t[N - 1:], sym_t[:e - 1]])
        B[prev.to(torch.int64), idx] = True
        del idx, prev
    if verbose:
        verbose("B_c of the text scattered into rotation order")

    # the windows of every SNP: W = T[p .. p + k] with position s replaced, p = s - k .. s
    pos, shift = snp_sites(N, k, period, device)
    ref = sym_t[pos].to(torch.int64)
    alt = (ref + shift.to(torch.int64)) % 4
    keep = torch.ones(pos.shape[0], dtype=torch.bool, device=device)
    targets = []
    for w in range(k + 1):
        p = pos - k + w                                # window start; the alternative base sits at offset k - w
        if w == k:                                     # the window starts AT the alternative base: target = the reference k-mer behind it
            v = value_at(pos + 1)
            label = alt
        else:                                          # target node = T[p + 1 .. p + k] with the digit of position s replaced:
            v = value_at(p + 1) + (alt - ref) * (4 ** w)   # s sits at offset k - w - 1 of the target k-mer, weight 4^w
            label = sym_t[p % N].to(torch.int64)
        keep &= v > 0                                  # A^k does not exist: a bubble that would need it is dropped whole
        targets.append((label, v))
    # the source k-mer of the first window is a reference k-mer; the sources of the others are alternative k-mers
    # = the targets of the previous window, so v > 0 for all targets covers them
    for label, v in targets:
        B[label[keep], (v[keep] - 1)] = True
    pos, alt = pos[keep], alt[keep]
    alt_t = torch.full((N,), 255, dtype=torch.uint8, device=device)
    alt_t[pos] = alt.to(torch.uint8)
    del targets, keep, ref
    if verbose:
        verbose(f"{pos.shape[0]} SNP bubbles added")

    # out-degree of node u = number of its four possible successors u[1..k) x that have the predecessor label u[0]
    outdeg = torch.empty(N, dtype=torch.uint8, device=device)
    low_mask = (1 << (2 * (k - 1))) - 1
    x = torch.arange(4, dtype=torch.int64, device=device).view(1, 4)
    for b in range(0, N, chunk):
        e = min(N, b + chunk)
        val = torch.arange(b + 1, e + 1, dtype=torch.int64, device=device)
        c = val >> (2 * (k - 1))
        succ = ((val & low_mask) << 2).view(-1, 1) + x             # values of the four successors
        ok = succ > 0
        outdeg[b:e] = (B[c.view(-1, 1).expand(-1, 4), torch.clamp(succ - 1, min=0)] & ok).sum(dim=1).to(torch.uint8)
        del val, c, succ, ok
    counts = np.zeros(4, dtype=np.uint64)                  # chunked: sum() of a bool tensor materialises int64
    for b in range(0, N, chunk):
        counts += B[:, b:b + chunk].sum(dim=1).cpu().numpy().astype(np.uint64)
    e_total = int(counts.sum())
    # edges: the last outgoing edge of every node marked (0^(outdeg - 1) 1)
    edge_bits = torch.zeros(e_total, dtype=torch.bool, device=device)
    carry = 0
    for b in range(0, N, chunk):
        e = min(N, b + chunk)
        cum = torch.cumsum(outdeg[b:e].to(torch.int64), dim=0) + carry
        edge_bits[cum - 1] = True
        carry = int(cum[-1].item())
        del cum
    assert carry == e_total, (carry, e_total)
    assert int(outdeg.min().item()) >= 1
    if verbose:
        verbose(f"edges: {e_total} = {e_total / N:.4f} n; branching nodes: {int((outdeg > 1).sum().item())}")
    Carr = np.zeros(SIGMA + 1, dtype=np.uint64)
    per_comp = np.zeros(SIGMA, dtype=np.uint64)
    per_comp[1:5] = counts
    Carr[1:] = np.cumsum(per_comp)
    zero = np.zeros((N + 63) // 64 + 1, dtype=np.uint64)
    bwt = [zero] + [pack_bits_torch(B[s]) for s in range(4)] + [zero, zero]
    edges = pack_bits_torch(edge_bits)
    del B, edge_bits, outdeg
    extras = dict(sampled_paths=zero, sample_count=0, sample_width=1,
                  stored_samples=np.zeros(2, dtype=np.uint64), stored_samples_plain=np.zeros(0, dtype=np.uint64),
                  samples=np.zeros(2, dtype=np.uint64), extra_filter=zero, extra_values_len=0,
                  extra_values=np.zeros(2, dtype=np.uint64), redundant_len=0, redundant=np.zeros(2, dtype=np.uint64),
                  lcp_size=0, lcp_branching=branching, lcp_offsets=np.zeros(2, dtype=np.uint64),
                  lcp_data=np.zeros(1, dtype=np.uint8))
    if with_lcp:
        lcp_data, lcp_offsets = mseq_lcp(degree, device, branching)
        extras.update(lcp_size=N, lcp_offsets=lcp_offsets, lcp_data=np.ascontiguousarray(lcp_data))
        if verbose:
            verbose("LCP array of the node set in closed form")
    ix = IndexArrays(n=N, e=e_total, order=k, sigma=SIGMA, fast_chars=FAST_CHARS, char2comp=default_char2comp(),
                     C=Carr, bwt=bwt, edges=edges, table=None, **extras)
    return ix, sym_t, rank, alt_t


def walk_patterns_device(sym_t: torch.Tensor, alt_t: torch.Tensor, rank_t: torch.Tensor, first: int, count: int, m: int, seed: int):
    """Queries first .. first + count - 1 of the global batch `seed` on the SNP graph: walks of m positions from
    SplitMix64 starts that take the alternative base at a SNP site when the site's coin (a hash of query and
    position) says so.  Returns (patterns (count, m) uint8 bytes, expected node int64): find() of the walk is
    (expected, expected) for m >= k."""
    device = sym_t.device
    N = sym_t.shape[0]
    degree = int(N).bit_length()
    k = degree // 2
    assert m >= k
    r = splitmix64_range_torch(seed, first, count, device)
    start = _lsr(r, 11) % N
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty((count, m), dtype=torch.uint8, device=device)
    expected = torch.empty(count, dtype=torch.int64, device=device)
    chunk = 1 << 23
    offs = torch.arange(m, dtype=torch.int64, device=device).view(1, -1)
    for b in range(0, count, chunk):
        e = min(count, b + chunk)
        idx = (start[b:e].view(-1, 1) + offs) % N
        ref = sym_t[idx].to(torch.int64)
        alt = alt_t[idx].to(torch.int64)
        coin = ((r[b:e].view(-1, 1) ^ (idx * _s64(0x9E3779B97F4A7C15))) >> 17) & 1
        take = (alt != 255) & (coin == 1)
        chosen = torch.where(take, alt, ref)
        out[b:e] = lut[chosen]
        weights = (4 ** torch.arange(k - 1, -1, -1, dtype=torch.int64, device=device)).view(1, -1)
        delta = ((chosen[:, :k] - ref[:, :k]) * weights).sum(dim=1)
        expected[b:e] = (rank_t[start[b:e]].to(torch.int64) & 0xFFFFFFFF) + delta
        del idx, ref, alt, coin, take, chosen, delta
    return out, expected
