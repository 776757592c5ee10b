"""Whole-human-pangenome-sized indexes: more than 2^32 path nodes, without suffix sorting.

SURVEY.md 8(d) config 4 / paper.tex:380: the whole-human index of the paper has 5.73 G path nodes.  The generator of
workload/mseq_torch.py stops at 2^32 - 1 (a maximal LFSR of degree 32: every 16-mer but one, ranks in closed form).
This module takes a NON-maximal cycle of a degree-34 LFSR -- x^34 + x^7 + 1 is irreducible of order (2^34 - 1) / 3 =
5 726 623 061, so its state sequence is a cyclic binary text of exactly that period in which every 34-bit window is
distinct; read two bits at a time (the period is odd) it is a cyclic text over {A, C, G, T} with 5.73 G positions and
pairwise distinct 17-mers.  Only a third of the 17-mers exist, so the rank of a 17-mer among the path nodes is no
longer its value: it is the number of set bits below it in a bitmap of the 4^17 = 2^34 possible 17-mers (2 GB packed).

Two indexes over that text, both order-17 de Bruijn graphs (every path label a distinct 17-mer; an unpruned de Bruijn
graph is a valid GCSA path graph, see workload/mseq_torch.py):

  build_dbg(degree)                 the plain cyclic text: n = e = period, FM-index shaped, with samples (every 32nd
                                    position), counters and LCP array in closed form when full=True
  build_dbg(degree, junctions=80)   the same path nodes plus JUNCTION edges: node u gets, besides its text successor, the
                                    edge to u[1..k) x for every other x whose k-mer is a path node and whose hash selects
                                    it (80 per mille: a third of the 3 n candidates exist) -- n = 5.73 G path nodes and
                                    e = 1.08 n, the paper's whole-human figures (paper.tex:380), with out-degrees and
                                    in-degrees above one at 8 % of the nodes.  Every node keeps its single value, so
                                    samples (every 32nd position and every node with several predecessors, the rule of
                                    src/gcsa.cpp:621-646), counters and the LCP array stay in closed form.
  build_dbg(degree, period=54)      the text plus one SNP bubble per `period` positions: the alternative 17-mers are NEW
                                    path nodes two times out of three (for degree 32 they all existed already), so
                                    n = 1.21 x 5.73 G = 6.9 G path nodes; in a k-mer space this sparse a bubble stays a
                                    bubble, e = 1.03 n (find() and LCP only)

Every answer stays analytic: find() of a pattern spelled by a walk of >= 17 positions is the single node of its first
17 characters, i.e. (r, r) with r = bitmap rank of that 17-mer; locate() on the plain index is the node value of the
start position; the LCP of lexicographically adjacent nodes is the common prefix of their 17-mers.

Workload generation only (torch ops; runs on the GPU box at full size, on the CPU at degrees 8-20 where
tests/test_workload.py checks it against the general builder and against the definition).
"""
import ctypes as C

import numpy as np
import torch

from .graphs import SIGMA, FAST_CHARS, default_char2comp
from .index_arrays import IndexArrays, bit_length
from .linear_torch import pack_bits_torch, _lsr, _s64
from .mseq_torch import NODE_LEN, node_values, splitmix64_range_torch

# tap lists of a[n + d] = XOR a[n + d - tap] whose cycle through state 1 has (2^d - 1) / 3 states (verified at run time
# by gcsa_lfsr_text; found by exhaustive search over 2- and 4-tap recurrences)
LFSR = {8: [8, 7, 3, 1], 10: [10, 3, 2, 1], 12: [12, 11, 2, 1], 16: [16, 6, 2, 1], 20: [20, 3, 2, 1], 24: [24, 22, 11, 1],
        28: [28, 3, 2, 1], 34: [34, 7]}

SNP_SEED = 0x6C5A0043


def text_length(degree: int) -> int:
    return ((1 << degree) - 1) // 3


def lfsr_text(degree: int) -> np.ndarray:
    """sym uint8[P] in 0..3 of the cyclic text, P = (2^degree - 1) / 3."""
    from . import builder
    lib = builder._load()
    lib.gcsa_lfsr_text.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_uint64, C.c_void_p]
    taps = LFSR[degree]
    P = text_length(degree)
    sym = np.empty(P, dtype=np.uint8)
    rc = lib.gcsa_lfsr_text(degree, (C.c_int * len(taps))(*taps), len(taps), P, sym.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"gcsa_lfsr_text({degree}) failed with {rc}")
    return sym


def lfsr_next_kmer(v: torch.Tensor, degree: int) -> torch.Tensor:
    """k-mer value of the next text position: two steps of the recurrence on the window of `degree` bits
    (workload/builder.cpp: gcsa_lfsr_text)."""
    mask = (1 << degree) - 1
    for _ in range(2):
        fb = torch.zeros_like(v)
        for t in LFSR[degree]:
            fb ^= (v >> (t - 1)) & 1
        v = ((v << 1) | fb) & mask
    return v


JUNCTION_SEED = 0x6C5A0044


def junction_selected(edge: torch.Tensor, permille: int) -> torch.Tensor:
    """Is the candidate edge (value of the (k + 1)-mer u x) part of the graph?  A SplitMix64 hash of the edge decides."""
    z = edge * _s64(0x9E3779B97F4A7C15) + _s64(JUNCTION_SEED)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    z = z ^ _lsr(z, 31)
    return (_lsr(z, 11) % 1000) < permille


def popcount64(x: torch.Tensor) -> torch.Tensor:
    """Set bits of every int64 element (torch has no popcount)."""
    x = x - (_lsr(x, 1) & 0x5555555555555555)
    x = (x & 0x3333333333333333) + (_lsr(x, 2) & 0x3333333333333333)
    x = (x + _lsr(x, 4)) & 0x0F0F0F0F0F0F0F0F
    return _lsr(x * 0x0101010101010101, 56)


def pack_bits_device(bits: torch.Tensor) -> torch.Tensor:
    """bool[n] -> int64 words on the same device, bit i in word i >> 6 at position i & 63."""
    n = bits.shape[0]
    nwords = (n + 63) // 64
    dev = bits.device
    weights = torch.ones(64, dtype=torch.int64, device=dev) << torch.arange(64, device=dev)
    out = torch.zeros(nwords, dtype=torch.int64, device=dev)
    chunk = 1 << 22
    for w0 in range(0, nwords, chunk):
        w1 = min(nwords, w0 + chunk)
        seg = bits[w0 * 64: min(n, w1 * 64)]
        if seg.shape[0] < (w1 - w0) * 64:
            seg = torch.cat([seg, torch.zeros((w1 - w0) * 64 - seg.shape[0], dtype=seg.dtype, device=dev)])
        out[w0:w1] = (seg.view(-1, 64).to(torch.int64) * weights).sum(dim=1)
    return out


class NodeSet:
    """The path nodes as a subset of the k-mer universe [0, 4^k): packed bitmap + word-granular prefix counts.
    rank(v) = number of nodes with a smaller k-mer value = the path node id of k-mer v when v is a node."""

    def __init__(self, isnode: torch.Tensor):
        self.universe = int(isnode.shape[0])
        self.words = pack_bits_device(isnode)
        counts = torch.empty_like(self.words)
        chunk = 1 << 24
        for b in range(0, self.words.shape[0], chunk):
            counts[b:b + chunk] = popcount64(self.words[b:b + chunk])
        self.cum = torch.cumsum(counts, dim=0) - counts             # exclusive
        self.n = int((self.cum[-1] + counts[-1]).item())
        del counts

    def rank(self, v: torch.Tensor) -> torch.Tensor:
        w = v >> 6
        below = (torch.ones_like(v) << (v & 63)) - 1
        return self.cum[w] + popcount64(self.words[w] & below)

    def contains(self, v: torch.Tensor) -> torch.Tensor:
        return ((self.words[v >> 6] >> (v & 63)) & 1) == 1

    def values(self, a: int, b: int):
        """(k-mer values of the nodes in [a, b), id of the first of them); a, b multiples of 64."""
        w = self.words[a >> 6: b >> 6]
        bits = ((w.view(-1, 1) >> torch.arange(64, device=w.device).view(1, 64)) & 1).view(-1)
        vals = torch.nonzero(bits).view(-1) + a
        return vals, int(self.cum[a >> 6].item())


class DbgWorkload:
    """What the closed forms need after the index has been handed over: the text, the alternative symbols, the node set."""

    def __init__(self, degree, sym_t, alt_t, nodes, junctions=0):
        self.degree, self.k = degree, degree // 2
        self.sym_t, self.alt_t, self.nodes, self.junctions = sym_t, alt_t, nodes, junctions
        self.P = int(sym_t.shape[0])
        self.sym_ext = torch.cat([sym_t, sym_t[: self.k + 1]])        # the cyclic text unrolled by one k-mer

    def values_at(self, p: torch.Tensor) -> torch.Tensor:
        """k-mer value of T[p .. p + k) for positions p (int64, any values: taken modulo P)."""
        p = p % self.P
        v = torch.zeros_like(p)
        for j in range(self.k):
            v = (v << 2) | self.sym_ext[p + j].to(torch.int64)
        return v

    def values_range(self, b: int, e: int) -> torch.Tensor:
        """k-mer values of positions b .. e - 1 (0 <= b <= e <= P), by slices."""
        v = torch.zeros(e - b, dtype=torch.int64, device=self.sym_t.device)
        for j in range(self.k):
            v = (v << 2) | self.sym_ext[b + j: e + j].to(torch.int64)
        return v


def snp_sites(P: int, k: int, period: int, device):
    """(positions int64, shift 1..3): one site per `period` positions, at least 2 k + 2 apart (a window of k + 1 symbols
    covers at most one), alternative base = reference + shift mod 4."""
    count = P // period - 1
    i = torch.arange(count, dtype=torch.int64, device=device)
    r = splitmix64_range_torch(SNP_SEED, 0, count, device)
    room = period - 2 * (k + 1)
    assert room >= 1, "SNP period too short for the order"
    pos = i * period + (k + 1) + _lsr(r, 11) % room
    shift = 1 + _lsr(r, 7) % 3
    return pos, shift


def dbg_lcp(nodes: NodeSet, k: int, device, branching: int = 64, chunk_bits: int = 27):
    """LCP bytes + range-minimum tree of the node set: LCP[i] = common prefix (in characters) of the k-mers of nodes
    i - 1 and i; LCP[0] = 0."""
    lcp = torch.empty(nodes.n, dtype=torch.uint8, device=device)
    step = min(nodes.universe, 1 << chunk_bits)
    prev = None
    for a in range(0, nodes.universe, step):
        vals, first = nodes.values(a, min(nodes.universe, a + step))
        if vals.shape[0] == 0:
            continue
        left = torch.cat([prev if prev is not None else vals[:1], vals[:-1]])
        x = vals ^ left
        q = torch.zeros_like(x)
        for m in range(1, k + 1):
            q += (x >> (2 * (k - m))) == 0
        lcp[first: first + vals.shape[0]] = q.clamp(max=k - 1).to(torch.uint8)     # (distinct k-mers share < k characters)
        prev = vals[-1:].clone()
        del vals, left, x, q
    lcp[0] = 0
    return lcp_tree_torch(lcp, branching)


def lcp_tree_torch(lcp: torch.Tensor, branching: int):
    """== index_arrays.build_lcp_tree (levels and data as LCPArray::LCPArray lays them out, src/lcp.cpp:224-259), with the
    level minima taken on the device; returns host arrays."""
    n = int(lcp.shape[0])
    sizes = [n]
    while sizes[-1] > 1:
        sizes.append((sizes[-1] + branching - 1) // branching)
    offsets = np.zeros(len(sizes) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(sizes)
    data = np.empty(int(offsets[-1]), dtype=np.uint8)
    level = lcp
    at = 0
    for size in sizes:
        assert int(level.shape[0]) == size
        data[at: at + size] = level.cpu().numpy()
        at += size
        if size == 1:
            break
        whole = size // branching
        parts = [level[: whole * branching].view(whole, branching).amin(dim=1)]
        if whole * branching < size:
            parts.append(level[whole * branching:].amin().view(1))
        level = torch.cat(parts)
    return data, offsets


def _empty_extras(zero, branching):
    return dict(sampled_paths=zero, sample_count=0, sample_width=1,
                stored_samples=np.zeros(2, dtype=np.uint64), stored_samples_plain=np.zeros(0, dtype=np.uint64),
                samples=np.zeros(2, dtype=np.uint64), extra_filter=zero, extra_values_len=0,
                extra_values=np.zeros(2, dtype=np.uint64), redundant_len=0, redundant=np.zeros(2, dtype=np.uint64),
                lcp_size=0, lcp_branching=branching, lcp_offsets=np.zeros(2, dtype=np.uint64),
                lcp_data=np.zeros(1, dtype=np.uint8))


def build_dbg(degree: int, period: int = 0, junctions: int = 0, device=None, verbose=None, full: bool = False,
              with_lcp: bool = False, branching: int = 64, chunk_bits: int = 27):
    """Returns (IndexArrays, DbgWorkload).

    period = 0, junctions = 0: the plain cyclic text (n = e = P).
    junctions = j > 0: the same path nodes with j per mille of the candidate junction edges (module docstring).
    For both, full = True adds, in closed form, the samples (position p carries node_values(p); a node is sampled iff
    p % 32 == 0 or it has several predecessors, the rule of src/gcsa.cpp:621-646 for these values), the counters (one
    value per node: A = 0, R = 0) and the LCP array.
    period > 0: one SNP bubble per `period` positions (find() and, with_lcp, the LCP array; no samples)."""
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    assert not (period > 0 and junctions > 0)
    say = verbose or (lambda msg: None)
    k = degree // 2
    U = 1 << degree
    sym_t = torch.from_numpy(lfsr_text(degree)).to(device)
    P = int(sym_t.shape[0])
    say(f"LFSR text: {P} symbols, order {k}")
    wl = DbgWorkload(degree, sym_t, None, None, junctions)
    chunk = 1 << chunk_bits
    step = min(U, chunk)
    low_mask = (1 << (2 * (k - 1))) - 1

    # ---- node set -------------------------------------------------------------------------------------------
    isnode = torch.zeros(U, dtype=torch.bool, device=device)
    for b in range(0, P, chunk):
        e = min(P, b + chunk)
        isnode[wl.values_range(b, e)] = True
    pos = alt = ref = None
    if period > 0:
        pos, shift = snp_sites(P, k, period, device)
        ref = sym_t[pos].to(torch.int64)
        alt = (ref + shift) % 4
        for j in range(k):                             # the k-mer starting at pos - j holds the site at offset j
            isnode[wl.values_at(pos - j) + (alt - ref) * (4 ** (k - 1 - j))] = True
    nodes = NodeSet(isnode)
    del isnode
    wl.nodes = nodes
    n = nodes.n
    say(f"node set: {n} of {U} k-mers" + (f" ({pos.shape[0]} SNP bubbles)" if period > 0 else ""))

    # ---- B_c: node v has the predecessor label c iff the (k + 1)-mer c v is an edge -----------------------------
    B = torch.zeros((4, n), dtype=torch.bool, device=device)
    outdeg = None
    if junctions > 0:
        # one sweep over the nodes in lexicographic order: node u (first symbol c) -> its text successor and the selected
        # junction successors u[1..k) x; out-degrees on the way
        outdeg = torch.empty(n, dtype=torch.int64, device=device)
        for a in range(0, U, step):
            vals, first = nodes.values(a, min(U, a + step))
            if vals.shape[0] == 0:
                continue
            c = vals >> (2 * (k - 1))
            succ = lfsr_next_kmer(vals, degree)
            B[c, nodes.rank(succ)] = True
            deg = torch.ones_like(vals)
            base = (vals & low_mask) << 2
            for x in range(4):
                cand = base + x
                ok = (cand != succ) & nodes.contains(cand) & junction_selected((vals << 2) + x, junctions)
                idx = torch.nonzero(ok).view(-1)
                B[c[idx], nodes.rank(cand[idx])] = True
                deg[idx] += 1
                del cand, ok, idx
            outdeg[first: first + vals.shape[0]] = deg
            del vals, c, succ, deg, base
    else:
        for b in range(0, P, chunk):
            e = min(P, b + chunk)
            prev = sym_t[b - 1:e - 1] if b > 0 else torch.cat([sym_t[P - 1:], sym_t[:e - 1]])
            B[prev.to(torch.int64), nodes.rank(wl.values_range(b, e))] = True
            del prev
    if period > 0:
        for w in range(k + 1):                         # the window of k + 1 symbols starting at q = pos - k + w
            q = pos - k + w
            if w == k:                                 # it starts AT the alternative base: the target is the reference k-mer behind it
                label, v = alt, wl.values_at(pos + 1)
            else:                                      # the target T[q + 1 .. q + k] holds the site at offset k - w - 1
                label, v = sym_t[q % P].to(torch.int64), wl.values_at(q + 1) + (alt - ref) * (4 ** w)
            B[label, nodes.rank(v)] = True
        alt_t = torch.full((P,), 255, dtype=torch.uint8, device=device)
        alt_t[pos] = alt.to(torch.uint8)
        wl.alt_t = alt_t
    say("B_c scattered")

    counts = np.zeros(4, dtype=np.uint64)
    for b in range(0, n, chunk):
        counts += B[:, b:b + chunk].sum(dim=1).cpu().numpy().astype(np.uint64)
    e_total = int(counts.sum())

    # ---- edges: out-degree of node u = number of successors u[1..k) x that have the predecessor label u[0] ---------
    if period > 0:
        outdeg = torch.zeros(n, dtype=torch.int64, device=device)
        for a in range(0, U, step):
            vals, first = nodes.values(a, min(U, a + step))
            if vals.shape[0] == 0:
                continue
            c = vals >> (2 * (k - 1))
            base = (vals & low_mask) << 2
            deg = torch.zeros_like(vals)
            for x in range(4):
                succ = base + x
                has = nodes.contains(succ)
                idx = torch.nonzero(has).view(-1)
                hit = B[c[idx], nodes.rank(succ[idx])]
                deg[idx] += hit.to(torch.int64)
                del succ, has, idx, hit
            outdeg[first: first + vals.shape[0]] = deg
            del vals, c, base, deg
    if outdeg is not None:
        assert int(outdeg.min().item()) >= 1 and int(outdeg.sum().item()) == e_total
        edge_bits = torch.zeros(e_total, dtype=torch.bool, device=device)
        carry = 0
        for b in range(0, n, chunk):
            cum = torch.cumsum(outdeg[b:b + chunk], dim=0) + carry
            edge_bits[cum - 1] = True
            carry = int(cum[-1].item())
            del cum
        say(f"edges: {e_total} = {e_total / n:.4f} n; branching nodes: {int((outdeg > 1).sum().item())}")
        del outdeg
    else:
        assert e_total == n == P
        edge_bits = torch.ones(n, dtype=torch.bool, device=device)

    Carr = np.zeros(SIGMA + 1, dtype=np.uint64)
    per_comp = np.zeros(SIGMA, dtype=np.uint64)
    per_comp[1:5] = counts
    Carr[1:] = np.cumsum(per_comp)
    zero = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
    bwt = [zero] + [pack_bits_torch(B[s]) for s in range(4)] + [zero, zero]
    edges = pack_bits_torch(edge_bits)
    del edge_bits
    say("B_c and edges packed")

    extras = _empty_extras(zero, branching)
    if full and period == 0:
        # sampled: p % 32 == 0 (a new vg node: the value is not the predecessor's + 1) or several predecessors
        ranks, places = [], []
        for b in range(0, P, chunk):
            e = min(P, b + chunk)
            r = nodes.rank(wl.values_range(b, e))
            p = torch.arange(b, e, dtype=torch.int64, device=device)
            keep = (p % NODE_LEN) == 0
            if junctions > 0:
                keep |= B[:, r].sum(dim=0) > 1
            ranks.append(r[keep]); places.append(p[keep])
            del r, p, keep
        srank, spos = torch.cat(ranks), torch.cat(places)
        del ranks, places
        order = torch.argsort(srank)
        stored = node_values(spos[order].cpu().numpy())
        S = int(stored.shape[0])
        width = bit_length(int(stored.max()))
        from . import builder as _b
        lib = _b._load()
        lib.gcsa_pack_ints.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        packed = np.zeros((S * width + 63) // 64 + 2, dtype=np.uint64)
        lib.gcsa_pack_ints(stored.ctypes.data, S, width, packed.ctypes.data)
        sampled = torch.zeros(n, dtype=torch.bool, device=device)
        sampled[srank] = True
        extras.update(sampled_paths=pack_bits_torch(sampled), sample_count=S, sample_width=width,
                      stored_samples=packed, stored_samples_plain=stored,
                      samples=pack_bits_torch(torch.ones(S, dtype=torch.bool, device=device)),
                      redundant_len=n - 1, redundant=pack_bits_torch(torch.ones(n - 1, dtype=torch.bool, device=device)))
        del sampled, srank, order, spos
        say(f"samples ({S}, {width} bits each) and counters in closed form")
    del B
    if with_lcp or (full and period == 0):
        lcp_data, lcp_offsets = dbg_lcp(nodes, k, device, branching, chunk_bits)
        extras.update(lcp_size=n, lcp_offsets=lcp_offsets, lcp_data=np.ascontiguousarray(lcp_data))
        say("LCP array of the node set")
    ix = IndexArrays(n=n, e=e_total, order=k, sigma=SIGMA, fast_chars=FAST_CHARS, char2comp=default_char2comp(),
                     C=Carr, bwt=bwt, edges=edges, table=None, **extras)
    return ix, wl


def distinct_prefixes(nodes, order, k):
    """Number of distinct k-prefixes of the node set (k <= order), from its packed bitmap (value = first character highest): the
    closed form of countKMers(k) on these indexes -- every k-mer of a de Bruijn graph of that order is a prefix of a node label,
    and junction edges between existing nodes add none (reference: src/algorithms.cpp:387-421 counts the non-empty states at
    depth k of the LF search tree)."""
    drop = 2 * (order - k)                      # low bits of the value that a prefix ignores
    w = nodes.words
    if drop >= 6:
        group = 1 << (drop - 6)                 # whole words per prefix
        total = 0
        chunk = (1 << 24) // max(1, group) * group or group
        for b in range(0, w.shape[0], chunk):
            part = w[b:b + chunk]
            pad = (-part.shape[0]) % group
            if pad:
                part = torch.cat([part, torch.zeros(pad, dtype=part.dtype, device=part.device)])
            total += int((part.view(-1, group) != 0).any(dim=1).sum().item())
        return total
    width = 1 << drop                           # bits per prefix inside a word: 1, 4 or 16
    total = 0
    mask = {1: -1, 4: 0x1111111111111111, 16: 0x0001000100010001}[width]
    for b in range(0, w.shape[0], 1 << 24):
        x = w[b:b + (1 << 24)].clone()
        s = 1
        while s < width:
            x |= (x >> s) & ((1 << (64 - s)) - 1 if s else -1)       # logical shift on int64
            s <<= 1
        if mask != -1:
            x &= mask - (1 << 64) if mask >= (1 << 63) else mask
        total += int(popcount64(x).sum().item())
    return total



def cycle_graph(degree: int):
    """The input graph of the plain text: one cycle of P positions (no source / sink)."""
    from .graphs import Graph
    sym = lfsr_text(degree)
    P = sym.shape[0]
    succ = np.roll(np.arange(P, dtype=np.uint32), -1)
    return Graph(comp=(sym + 1).astype(np.uint8), value=node_values(np.arange(P)), succ_off=np.arange(P + 1, dtype=np.uint64),
                 succ=succ, source=0, sink=P - 1)


def walk_patterns_device(wl: DbgWorkload, first: int, count: int, m: int, seed: int):
    """Queries first .. first + count - 1 of the global batch `seed`: walks of m >= k characters from SplitMix64 start
    positions.  Plain index: substrings of the text.  Junction index: at every node the walk takes a junction edge when
    one exists for the symbol its coin (a hash of query and step) names.  SNP index: the alternative base at a site when
    the site's coin says so.  Returns (patterns (count, m) uint8 bytes, start positions int64, expected node int64):
    find() of the walk is (expected, expected), and on the plain / junction index locate() of it is
    node_values(start)."""
    device = wl.sym_t.device
    P, k = wl.P, wl.k
    assert m >= k
    r = splitmix64_range_torch(seed, first, count, device)
    start = _lsr(r, 11) % P
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty((count, m), dtype=torch.uint8, device=device)
    expected = torch.empty(count, dtype=torch.int64, device=device)
    chunk = 1 << 22
    offs = torch.arange(m, dtype=torch.int64, device=device).view(1, -1)
    low_mask = (1 << (2 * (k - 1))) - 1
    for b in range(0, count, chunk):
        e = min(count, b + chunk)
        if wl.junctions > 0:
            v = wl.values_at(start[b:e])
            expected[b:e] = wl.nodes.rank(v)
            for j in range(k):
                out[b:e, j] = lut[(v >> (2 * (k - 1 - j))) & 3]
            for j in range(k, m):
                nxt = lfsr_next_kmer(v, wl.degree)
                h = r[b:e] ^ _s64((j * 0x9E3779B97F4A7C15) & ((1 << 64) - 1))
                h = (h ^ _lsr(h, 29)) * _s64(0xBF58476D1CE4E5B9)
                x = _lsr(h, 40) & 3
                cand = ((v & low_mask) << 2) + x
                take = (cand != nxt) & wl.nodes.contains(cand) & junction_selected((v << 2) + x, wl.junctions)
                v = torch.where(take, cand, nxt)
                out[b:e, j] = lut[v & 3]
                del nxt, h, x, cand, take
            del v
            continue
        idx = (start[b:e].view(-1, 1) + offs) % P
        chosen = wl.sym_t[idx].to(torch.int64)
        if wl.alt_t is not None:
            alt = wl.alt_t[idx].to(torch.int64)
            coin = ((r[b:e].view(-1, 1) ^ (idx * _s64(0x9E3779B97F4A7C15))) >> 17) & 1
            chosen = torch.where((alt != 255) & (coin == 1), alt, chosen)
            del alt, coin
        out[b:e] = lut[chosen]
        v = torch.zeros(e - b, dtype=torch.int64, device=device)
        for j in range(k):
            v = (v << 2) | chosen[:, j]
        expected[b:e] = wl.nodes.rank(v)
        del idx, chosen, v
    return out, start, expected


def prefix_patterns_device(wl: DbgWorkload, first: int, count: int, m: int, seed: int):
    """Queries first .. first + count - 1 of the global batch `seed`: the first m < k characters of the k-mer at a
    SplitMix64 text position.  Every path label is a distinct k-mer and the path nodes are the k-mers in lexicographic
    order, so find() of such a prefix is the interval of the nodes whose k-mer starts with it: with lo = prefix << 2 (k - m)
    and hi = lo + 4^(k - m), (sp, ep) = (rank(lo), rank(hi) - 1) through the bitmap of the k-mer universe -- about
    n / 4^m path nodes (341 / 21 / 1.3 for m = 12 / 14 / 16 at 5.73 G nodes): the wide ranges a pattern meets in the first
    steps of its search.  Returns (patterns (count, m) uint8 bytes, sp int64, ep int64)."""
    device = wl.sym_t.device
    P, k = wl.P, wl.k
    assert 1 <= m < k
    r = splitmix64_range_torch(seed, first, count, device)
    start = _lsr(r, 11) % P
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty((count, m), dtype=torch.uint8, device=device)
    sp = torch.empty(count, dtype=torch.int64, device=device)
    ep = torch.empty(count, dtype=torch.int64, device=device)
    shift = 2 * (k - m)
    chunk = 1 << 22
    last_word = wl.nodes.words.shape[0] - 1
    for b in range(0, count, chunk):
        e = min(count, b + chunk)
        v = wl.values_at(start[b:e])
        for j in range(m):
            out[b:e, j] = lut[(v >> (2 * (k - 1 - j))) & 3]
        lo = (v >> shift) << shift
        hi = lo + (1 << shift)
        sp[b:e] = wl.nodes.rank(lo)
        # rank(4^k) = n: the position one past the bitmap
        inside = hi < wl.nodes.universe
        ep[b:e] = torch.where(inside, wl.nodes.rank(torch.where(inside, hi, torch.zeros_like(hi))), torch.full_like(hi, wl.nodes.n)) - 1
        del v, lo, hi, inside
    return out, sp, ep
