"""Seeded input graphs, one character per position.

A graph is what GCSA2 construction sees after vg expands node sequences into
single-base positions: every position has a comp code (default alphabet
`$ACGTN#` = 0..6, reference `src/support.cpp:69-92`), a `node_type` value
(`id << 11 | offset`, reference `include/gcsa/support.h:443-471`), and
successor positions.  The sink `$` has the artificial edge back to the source
`#` (the (t, s) edge of `paper/paper.tex:262`), so every position has a
successor and every order-K path label is well defined.
"""
from dataclasses import dataclass
import numpy as np

from .rng import SplitMix64, splitmix64_array

COMP2CHAR = b"$ACGTN#"
SIGMA = 7
FAST_CHARS = 4
OFFSET_BITS = 10
ID_OFFSET = OFFSET_BITS + 1


def default_char2comp() -> np.ndarray:
    """Reference `Alphabet::DEFAULT_CHAR2COMP`, restated from its rule: NUL and `$` -> 0,
    `ACGT`/`acgt` -> 1..4, `#` -> 6, every other byte -> 5 (`N`)."""
    t = np.full(256, 5, dtype=np.uint8)
    t[0] = 0
    t[ord("$")] = 0
    for i, ch in enumerate("ACGT"):
        t[ord(ch)] = i + 1
        t[ord(ch.lower())] = i + 1
    t[ord("#")] = 6
    return t


def encode_node(node_id, offset=0, rc=False):
    return (int(node_id) << ID_OFFSET) | (int(bool(rc)) << OFFSET_BITS) | int(offset)


@dataclass
class Graph:
    comp: np.ndarray      # uint8[N]
    value: np.ndarray     # uint64[N]   node_type of each position
    succ_off: np.ndarray  # uint64[N+1] CSR offsets
    succ: np.ndarray      # uint32[...] successor positions
    source: int
    sink: int

    @property
    def size(self):
        return int(self.comp.shape[0])

    def successors(self, v):
        return self.succ[int(self.succ_off[v]):int(self.succ_off[v + 1])]

    def predecessor_lists(self):
        preds = [[] for _ in range(self.size)]
        for v in range(self.size):
            for w in self.successors(v):
                preds[int(w)].append(v)
        return preds


def _from_adj(comp, value, adj, source, sink):
    off = np.zeros(len(adj) + 1, dtype=np.uint64)
    flat = []
    for i, a in enumerate(adj):
        a = sorted(set(a))
        flat.extend(a)
        off[i + 1] = len(flat)
    return Graph(np.asarray(comp, dtype=np.uint8), np.asarray(value, dtype=np.uint64), off,
                 np.asarray(flat, dtype=np.uint32), source, sink)


def paper_graph() -> Graph:
    """The worked example of `paper/gcsa2_graph_dbg.ipe` / `gcsa2_pruned_index.ipe`
    (Figures 2-3 of the paper; transcribed in SURVEY.md §4.3).

    Nodes 0..11 labelled `# G C A T T C A G T A $`.  The figure pads the source with the
    abstract positions `0:1`, `0:2` (keys `##G`, `###`); to make them concrete integers that
    still satisfy "value = predecessor value + 1" every value is shifted by +2 and the two
    padding positions get values 1 and 0.  So a value v here is the figure's node v - 2.
    """
    labels = "#GCATTCAGTA$"
    edges = [(0, 1), (1, 2), (2, 3), (2, 4), (3, 5), (4, 5), (5, 6), (5, 8), (6, 7), (7, 9),
             (8, 9), (9, 10), (10, 11)]
    c2c = default_char2comp()
    comp = [6, 6] + [int(c2c[ord(ch)]) for ch in labels]
    value = [0, 1] + [i + 2 for i in range(len(labels))]
    adj = [[] for _ in comp]
    adj[0].append(1)
    adj[1].append(2)
    for a, b in edges:
        adj[a + 2].append(b + 2)
    adj[11 + 2].append(0)  # (t, s)
    return _from_adj(comp, value, adj, 0, 13)


def random_bases(n: int, seed: int) -> np.ndarray:
    """n comp codes uniform over A,C,G,T (1..4)."""
    return ((splitmix64_array(seed, n) >> np.uint64(33)) % np.uint64(4)).astype(np.uint8) + 1


def linear_graph(n: int, seed: int, node_len: int = 32, sequence=None) -> Graph:
    """`#` + n bases + `$` as one path, chunked into vg-style nodes of node_len bases
    (ids 1.., offsets 0..node_len-1); source id = last id + 1, sink id = last id + 2."""
    return snp_graph(n, seed, snp_seed=None, node_len=node_len, sequence=sequence)


def snp_graph(n: int, seed: int, snp_seed, snp_period: int = 32, node_len: int = 32,
              sequence=None) -> Graph:
    """Backbone of n bases with single-base substitution bubbles.

    Each backbone position 1..n-2 is a SNP site with probability 1/snp_period (splitmix64 of
    snp_seed); the alternative base differs from the reference base.  vg-style ids: a SNP
    splits the backbone, the reference and alternative alleles are one-base nodes of their own,
    plain runs are cut every node_len bases.  Layout of positions: 0 = source `#`, 1..n =
    backbone, n+1 = sink `$`, n+2.. = alternative alleles in backbone order.
    """
    seq = random_bases(n, seed) if sequence is None else np.asarray(sequence, dtype=np.uint8)
    n = int(seq.shape[0])
    is_snp = np.zeros(n, dtype=bool)
    alt = np.zeros(n, dtype=np.uint8)
    if snp_seed is not None and n > 2:
        r = splitmix64_array(snp_seed, n)
        is_snp = ((r >> np.uint64(20)) % np.uint64(snp_period)) == 0
        is_snp[0] = False
        is_snp[-1] = False
        shift = ((r >> np.uint64(8)) % np.uint64(3)).astype(np.uint8) + 1  # 1..3
        alt = ((seq - 1 + shift) % 4 + 1).astype(np.uint8)
    sites = np.flatnonzero(is_snp)
    n_alt = int(sites.shape[0])

    # vg-style node ids and offsets along the backbone.
    starts_node = np.zeros(n, dtype=bool)
    starts_node[0] = True
    starts_node[sites] = True
    after = sites + 1
    starts_node[after[after < n]] = True
    # also cut plain runs every node_len bases
    run_start = np.flatnonzero(starts_node)
    run_id = np.cumsum(starts_node) - 1
    off_in_run = np.arange(n) - run_start[run_id]
    starts_node |= (off_in_run % node_len) == 0
    node_index = np.cumsum(starts_node)  # 1-based index among backbone nodes
    # alt allele nodes get the id right after their reference allele's node: renumber.
    alt_before = np.cumsum(is_snp) - is_snp  # alts strictly before position i
    ref_id = node_index + alt_before
    node_start = np.flatnonzero(starts_node)
    offset = np.arange(n) - node_start[node_index - 1]
    alt_id = ref_id[sites] + 1
    last_id = int(ref_id[-1])  # the last backbone position is never a site
    source_id, sink_id = last_id + 1, last_id + 2

    N = n + 2 + n_alt
    comp = np.empty(N, dtype=np.uint8)
    value = np.empty(N, dtype=np.uint64)
    comp[0] = 6
    comp[1:n + 1] = seq
    comp[n + 1] = 0
    comp[n + 2:] = alt[sites]
    value[0] = encode_node(source_id)
    value[1:n + 1] = (ref_id.astype(np.uint64) << np.uint64(ID_OFFSET)) | offset.astype(np.uint64)
    value[n + 1] = encode_node(sink_id)
    value[n + 2:] = alt_id.astype(np.uint64) << np.uint64(ID_OFFSET)

    # successors: position p (1..n) = backbone i=p-1.
    # out-degree: backbone i -> i+1 (and alt of i+1 if site); alt of i -> i+1 (and alt of i+1).
    deg = np.ones(N, dtype=np.uint64)
    nxt_is_site = np.zeros(n, dtype=bool)
    nxt_is_site[:-1] = is_snp[1:]
    deg[1:n + 1] += nxt_is_site
    deg[0] = 1 + (1 if n > 0 and is_snp[0] else 0)
    deg[n + 1] = 1
    deg[n + 2:] = 1 + nxt_is_site[sites]
    succ_off = np.zeros(N + 1, dtype=np.uint64)
    np.cumsum(deg, out=succ_off[1:])
    succ = np.empty(int(succ_off[-1]), dtype=np.uint32)
    alt_pos_of_site = np.zeros(n, dtype=np.int64)
    alt_pos_of_site[sites] = n + 2 + np.arange(n_alt)
    # source
    succ[int(succ_off[0])] = 1 if n > 0 else n + 1
    # backbone
    bpos = np.arange(1, n + 1)
    first = succ_off[bpos].astype(np.int64)
    nxt = np.where(np.arange(n) == n - 1, n + 1, bpos + 1)
    succ[first] = nxt
    idx = np.flatnonzero(nxt_is_site)
    succ[first[idx] + 1] = alt_pos_of_site[idx + 1]
    # sink -> source
    succ[int(succ_off[n + 1])] = 0
    # alts
    if n_alt:
        afirst = succ_off[n + 2:N].astype(np.int64)
        succ[afirst] = nxt[sites]
        aidx = np.flatnonzero(nxt_is_site[sites])
        succ[afirst[aidx] + 1] = alt_pos_of_site[sites[aidx] + 1]
    # CSR rows must be sorted (backbone successor < alt successor already holds).
    return Graph(comp, value, succ_off, succ, 0, n + 1)


def random_graph(n: int, seed: int, p_branch: float = 0.25, p_back: float = 0.0,
                 p_n: float = 0.05, alphabet: int = 4) -> Graph:
    """Small arbitrary graph for definitional tests: a backbone 0 -> 1 -> ... -> n+1 with random
    forward skip edges (bubbles, indels), optional back edges (cycles), occasional `N` labels and
    repeated labels.  Values follow the vg convention loosely: value+1 along simple chains,
    a fresh id after every branch point."""
    rng = SplitMix64(seed)
    N = n + 2
    comp = [6] + [0] * n + [0]
    for i in range(1, n + 1):
        if rng.below(1000) < int(p_n * 1000):
            comp[i] = 5
        else:
            comp[i] = 1 + rng.below(alphabet)
    adj = [[] for _ in range(N)]
    for i in range(N - 1):
        adj[i].append(i + 1)
    for i in range(0, N - 2):
        if rng.below(1000) < int(p_branch * 1000):
            j = i + 2 + rng.below(3)
            if j <= N - 1:
                adj[i].append(min(j, N - 1))
    for i in range(2, N - 1):
        if rng.below(1000) < int(p_back * 1000):
            j = 1 + rng.below(i - 1)
            adj[i].append(j)
    adj[N - 1] = [0]
    indeg = [0] * N
    for a in adj:
        for w in set(a):
            indeg[w] += 1
    value = [0] * N
    node_id, off = 1, 0
    for i in range(N):
        simple = i > 0 and indeg[i] == 1 and len(set(adj[i - 1])) == 1 and adj[i - 1][0] == i \
            and off < 7
        if simple:
            off += 1
        else:
            node_id += 1
            off = 0
        value[i] = encode_node(node_id, off)
    return _from_adj(comp, value, adj, 0, N - 1)


# ---- a repeat-rich backbone: the range widths of real genomes -------------------------------------------------------
# The paper's human indexes answer a found 32-mer with 336 path nodes on average and a 16-mer with 7129
# (paper/paper.tex:403,408): interspersed repeat families and tandem arrays.  A uniform random backbone has unique
# 16-mers, so every range of the other workloads is a singleton after a few steps.  This backbone plants, in blocks of
# REPEAT_BLOCK bases: an Alu-like family (a 300-bp consensus, one copy per block, 7 % of the bases of every copy
# substituted independently), a younger family (a consensus filling one block in 16, 2 % divergence) and short
# tandem arrays (unit of 2..7 bases repeated over 150..400 bases in one block in 16).  At 2^23 bases a found 32-mer
# matches ~300 path nodes on average and a 16-mer ~1300 (the means are carried by the repeats: the median is 1).
REPEAT_BLOCK = 600


def repeat_bases(n: int, seed: int, alu_divergence: float = 0.07, young_divergence: float = 0.02) -> np.ndarray:
    """n comp codes (1..4): random bases with the planted repeat families described above."""
    seq = random_bases(n, seed)
    blocks = n // REPEAT_BLOCK
    if blocks == 0:
        return seq
    h = splitmix64_array(seed ^ 0x5EED5EED, blocks)
    alu = random_bases(300, seed ^ 0xA1)
    young = random_bases(REPEAT_BLOCK, seed ^ 0xA2)
    base = np.arange(blocks, dtype=np.int64) * REPEAT_BLOCK
    kind = (h >> np.uint64(4)) % np.uint64(16)                    # 0: young family, 1: tandem array, else: Alu-like copy

    def mutate(copy_positions, consensus_bases, divergence, salt):
        r = splitmix64_array(seed ^ salt, copy_positions.size).reshape(copy_positions.shape)
        hit = (r >> np.uint64(11)) % np.uint64(10000) < np.uint64(int(divergence * 10000))
        shift = ((r >> np.uint64(40)) % np.uint64(3)).astype(np.uint8) + 1
        return np.where(hit, (consensus_bases - 1 + shift) % 4 + 1, consensus_bases).astype(np.uint8)

    sel = np.flatnonzero(kind >= 2)
    if sel.size:
        off = ((h[sel] >> np.uint64(20)) % np.uint64(REPEAT_BLOCK - 300)).astype(np.int64)
        pos = (base[sel] + off)[:, None] + np.arange(300, dtype=np.int64)[None, :]
        seq[pos] = mutate(pos, np.broadcast_to(alu, pos.shape), alu_divergence, 0xB1)
    sel = np.flatnonzero(kind == 0)
    if sel.size:
        pos = base[sel][:, None] + np.arange(REPEAT_BLOCK, dtype=np.int64)[None, :]
        seq[pos] = mutate(pos, np.broadcast_to(young, pos.shape), young_divergence, 0xB2)
    sel = np.flatnonzero(kind == 1)
    for j in sel:                                                 # one block in 16: a loop over ~n / 16000 arrays
        hj = int(h[j])
        unit_len = 2 + (hj >> 24) % 6
        length = 150 + (hj >> 32) % 251
        start = int(base[j]) + (hj >> 44) % (REPEAT_BLOCK - length)
        unit = random_bases(unit_len, (seed ^ hj) & 0xFFFFFFFFFFFFFFFF)
        seq[start:start + length] = np.resize(unit, length)
    return seq


def repeat_graph(n: int, seed: int, snp_seed, snp_period: int = 32, node_len: int = 32) -> Graph:
    """The chr22-like SNP graph over a repeat-rich backbone (repeat_bases)."""
    return snp_graph(n, seed, snp_seed, snp_period=snp_period, node_len=node_len, sequence=repeat_bases(n, seed))
