"""Turn a lexicographically sorted path-node table into the GCSA arrays.

`NodeTable` is what either builder produces (one row per path node, in key
order); `assemble()` applies the encoding rules of the reference's constructor
(`src/gcsa.cpp:568-704`, restated, not copied) to obtain exactly the members a
`gcsa::GCSA` + `gcsa::LCPArray` pair holds (`include/gcsa/gcsa.h:214-240`,
`include/gcsa/lcp.h:188-190`) as plain numpy arrays:

  bwt[c]            n bits   B_c[i] = node i has a predecessor labelled c
  edges             e bits   last outgoing edge of each node marked (0^{outdeg-1} 1)
  C                 sigma+1  prefix sums of per-comp edge counts
  sampled_paths     n bits   node i stores its values
  samples           S bits   last value of each sampled node marked
  stored_samples    S ints   concatenated sorted values of sampled nodes (packed, width w)
  extra_*           Sada-S encoding of A[i] = |values(i)| - 1     (`support.h:342-364`)
  redundant         Sadakane encoding of R[0..n-2] (k -> 0^k 1)   (`support.h:265-279`)
  lcp_data/offsets  byte LCP + b-ary range-minimum tree          (`src/lcp.cpp:224-259`)
"""
from dataclasses import dataclass, field
import numpy as np

from .graphs import SIGMA, FAST_CHARS, default_char2comp


def pack_bits(bits: np.ndarray, pad_words: int = 1) -> np.ndarray:
    """bool[n] -> uint64 words, bit i in word i>>6 at position i&63 (SDSL order).
    `pad_words` extra zero words are appended so that word-granular readers may overrun."""
    n = int(bits.shape[0])
    nwords = (n + 63) // 64 + pad_words
    by = np.packbits(bits.astype(np.uint8), bitorder="little")
    out = np.zeros(nwords * 8, dtype=np.uint8)
    out[: by.shape[0]] = by
    return out.view(np.uint64)


def unpack_bits(words: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)


def unary_bits(counts: np.ndarray, zeros_first_offset: int) -> np.ndarray:
    """Encode counts k as 0^{k - zeros_first_offset... } 1: offset 0 -> `0^k 1`, offset 1 -> `0^{k-1} 1`."""
    counts = np.asarray(counts, dtype=np.int64)
    lens = counts + 1 - zeros_first_offset
    total = int(lens.sum())
    bits = np.zeros(total, dtype=bool)
    if total:
        ends = np.cumsum(lens) - 1
        bits[ends] = True
    return bits


def bit_length(x: int) -> int:
    """Reference `bit_length()` = `sdsl::bits::hi(x) + 1` (so 0 -> 1)."""
    return max(1, int(x).bit_length())


def pack_ints(values: np.ndarray, width: int, pad_words: int = 1) -> np.ndarray:
    """`sdsl::int_vector<0>` layout: element i occupies bits [i*w, (i+1)*w) LSB-first."""
    values = np.asarray(values, dtype=np.uint64)
    m = int(values.shape[0])
    nwords = (m * width + 63) // 64 + pad_words
    out = np.zeros(nwords, dtype=np.uint64)
    if m == 0:
        return out
    pos = np.arange(m, dtype=np.uint64) * np.uint64(width)
    word = (pos >> np.uint64(6)).astype(np.int64)
    shift = pos & np.uint64(63)
    with np.errstate(over="ignore"):
        lo = values << shift
        np.add.at(out, word, lo)  # bit ranges are disjoint, so add == or
        spill = (shift + np.uint64(width)) > np.uint64(64)
        if spill.any():
            hi = values[spill] >> (np.uint64(64) - shift[spill])
            np.add.at(out, word[spill] + 1, hi)
    return out


def build_lcp_tree(lcp: np.ndarray, branching: int):
    """Levels and data exactly as `LCPArray::LCPArray` lays them out (`src/lcp.cpp:224-259`)."""
    n = int(lcp.shape[0])
    sizes = [n]
    while sizes[-1] > 1:
        sizes.append((sizes[-1] + branching - 1) // branching)
    offsets = np.zeros(len(sizes) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(sizes)
    levels = [lcp.astype(np.uint8)]
    for _ in range(1, len(sizes)):
        prev = levels[-1]
        m = (prev.shape[0] + branching - 1) // branching
        padded = np.full(m * branching, 255, dtype=np.uint8)
        padded[: prev.shape[0]] = prev
        levels.append(padded.reshape(m, branching).min(axis=1))
    data = np.concatenate(levels) if n else np.zeros(0, dtype=np.uint8)
    return data, offsets


@dataclass
class NodeTable:
    order: int
    pred_mask: np.ndarray   # uint8[n]  bit c = predecessor with comp c
    outdeg: np.ndarray      # uint32[n]
    lcp: np.ndarray         # uint8[n]  LCP of key[i-1], key[i]; lcp[0] = 0
    val_off: np.ndarray     # uint64[n+1]
    vals: np.ndarray        # uint64[...] sorted per node
    redundant: np.ndarray   # uint32[n-1]  R[] of `src/gcsa.cpp:590-619`
    key_len: np.ndarray = None  # uint16[n], informational
    keys: list = None           # tuples of comps, small builds only


@dataclass
class IndexArrays:
    n: int
    e: int
    order: int
    sigma: int
    fast_chars: int
    char2comp: np.ndarray
    C: np.ndarray
    bwt: list
    edges: np.ndarray
    sampled_paths: np.ndarray
    sample_count: int
    sample_width: int
    stored_samples: np.ndarray       # packed
    stored_samples_plain: np.ndarray  # uint64[S], for tests
    samples: np.ndarray
    extra_filter: np.ndarray
    extra_values_len: int
    extra_values: np.ndarray
    redundant_len: int
    redundant: np.ndarray
    lcp_size: int
    lcp_branching: int
    lcp_offsets: np.ndarray
    lcp_data: np.ndarray
    table: NodeTable = field(default=None, repr=False)

    def core_bytes(self) -> int:
        """Bytes of everything find() touches (B_c + edges), as plain bits."""
        return (self.sigma * self.n + self.e) // 8


def lf_single_pred(pred_mask, outdeg, C, nodes):
    """Vectorised LF(path_node) for nodes with exactly one predecessor: returns predecessor ids."""
    n = pred_mask.shape[0]
    # comp of the single predecessor
    comp = np.zeros(nodes.shape[0], dtype=np.int64)
    pm = pred_mask[nodes]
    for c in range(SIGMA):
        comp[(pm >> c) & 1 == 1] = c
    # rank(B_c, i) for each node: per-comp exclusive cumsum
    edge_pos = np.zeros(nodes.shape[0], dtype=np.int64)
    for c in range(SIGMA):
        sel = comp == c
        if not sel.any():
            continue
        bc = ((pred_mask >> c) & 1).astype(np.int64)
        excl = np.cumsum(bc) - bc
        edge_pos[sel] = int(C[c]) + excl[nodes[sel]]
    # rank(edges, x) = number of nodes whose last edge index < x  -> searchsorted on cumulative outdeg
    last_edge = np.cumsum(outdeg.astype(np.int64)) - 1  # position of each node's 1-bit
    return np.searchsorted(last_edge, edge_pos, side="left")


def assemble(table: NodeTable, sample_period: int = 64, branching: int = 64,
             char2comp=None, keep_table: bool = True) -> IndexArrays:
    n = int(table.pred_mask.shape[0])
    pm = table.pred_mask
    outdeg = table.outdeg.astype(np.int64)
    assert (outdeg >= 1).all(), "every path node needs an outgoing edge"
    e = int(outdeg.sum())
    counts = np.array([int(((pm >> c) & 1).sum()) for c in range(SIGMA)], dtype=np.uint64)
    assert int(counts.sum()) == e, "indegree/outdegree mismatch"
    C = np.zeros(SIGMA + 1, dtype=np.uint64)
    C[1:] = np.cumsum(counts)

    bwt = [pack_bits(((pm >> c) & 1).astype(bool)) for c in range(SIGMA)]
    edge_bits = np.zeros(e, dtype=bool)
    edge_bits[np.cumsum(outdeg) - 1] = True

    # --- sampling rules, `src/gcsa.cpp:621-646` ---
    val_off = table.val_off.astype(np.int64)
    vals = table.vals
    cnt = np.diff(val_off)
    indeg = np.zeros(n, dtype=np.int64)
    for c in range(SIGMA):
        indeg += (pm >> c) & 1
    sampled = (indeg > 1) | ((pm & 1) == 1)
    owner = np.repeat(np.arange(n), cnt)
    mod0 = (vals % np.uint64(sample_period)) == 0
    sampled[owner[mod0]] = True
    cand = np.flatnonzero(~sampled)
    if cand.shape[0]:
        assert (indeg[cand] == 1).all(), "unsampled node without a unique predecessor"
        pred = lf_single_pred(pm, table.outdeg, C, cand)
        differs = cnt[cand] != cnt[pred]
        same = ~differs
        cs, ps = cand[same], pred[same]
        if cs.shape[0]:
            k = cnt[cs]
            rep = np.repeat(np.arange(cs.shape[0]), k)
            within = np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k)
            cur = vals[val_off[cs][rep] + within]
            prv = vals[val_off[ps][rep] + within]
            bad = cur != prv + np.uint64(1)
            bad_nodes = np.zeros(cs.shape[0], dtype=bool)
            bad_nodes[rep[bad]] = True
            sampled[cs[bad_nodes]] = True
        sampled[cand[differs]] = True

    # --- samples, `src/gcsa.cpp:648-658, 701-703` ---
    sel = sampled[owner]
    stored_plain = vals[sel].astype(np.uint64)
    S = int(stored_plain.shape[0])
    samples_bits = np.zeros(S, dtype=bool)
    scnt = cnt[sampled]
    if S:
        samples_bits[np.cumsum(scnt) - 1] = True
    width = bit_length(int(stored_plain.max())) if S else 1

    # --- counters: A[i] = |values| - 1 as Sada-S, R as Sadakane ---
    A = cnt - 1
    filt = A > 0
    extra_values = unary_bits(A[filt], 1)
    red = unary_bits(table.redundant.astype(np.int64), 0) if n > 1 else np.zeros(0, dtype=bool)

    lcp_data, lcp_offsets = build_lcp_tree(table.lcp, branching)

    return IndexArrays(
        n=n, e=e, order=table.order, sigma=SIGMA, fast_chars=FAST_CHARS,
        char2comp=default_char2comp() if char2comp is None else char2comp, C=C, bwt=bwt,
        edges=pack_bits(edge_bits), sampled_paths=pack_bits(sampled),
        sample_count=S, sample_width=width,
        stored_samples=pack_ints(stored_plain, width), stored_samples_plain=stored_plain,
        samples=pack_bits(samples_bits),
        extra_filter=pack_bits(filt), extra_values_len=int(extra_values.shape[0]),
        extra_values=pack_bits(extra_values),
        redundant_len=int(red.shape[0]), redundant=pack_bits(red),
        lcp_size=n, lcp_branching=branching, lcp_offsets=lcp_offsets,
        lcp_data=np.ascontiguousarray(lcp_data),
        table=table if keep_table else None,
    )


def redundant_from_lcp(lcp, val_off, vals) -> np.ndarray:
    """R[] by the in-order suffix-tree traversal of `src/gcsa.cpp:590-619` (restated).

    A stack holds, for every open LCP-interval, its LCP value and the first and last node index
    at which that value was seen.  When value x re-occurs, the lowest common ancestor of the
    previous and the current occurrence is the first stack entry whose last_time reaches back to
    the previous occurrence; R is incremented at that interval's first visit.
    """
    import bisect
    n = len(lcp)
    red = np.zeros(max(n - 1, 0), dtype=np.uint32)
    node_lcp, first_time, last_time = [], [], []
    prev_occ = {}
    for i in range(n):
        cur = int(lcp[i]) + (1 if i > 0 else 0)  # LCP[0] acts as -1
        while node_lcp and node_lcp[-1] > cur:
            node_lcp.pop(); first_time.pop(); last_time.pop()
        if node_lcp and node_lcp[-1] == cur:
            last_time[-1] = i
        else:
            node_lcp.append(cur); first_time.append(i); last_time.append(i)
        for v in vals[int(val_off[i]):int(val_off[i + 1])]:
            v = int(v)
            p = prev_occ.get(v, 0)
            if p > 0:
                pos = bisect.bisect_left(last_time, p)
                red[first_time[pos] - 1] += 1
            prev_occ[v] = i + 1
    return red
