"""Naive references used to pin the oracle (tests only).

Two independent levels:
  * NaiveIndex  -- the Appendix-A pseudocode of SURVEY.md evaluated with linear-time rank /
                   select over unpacked bit arrays, and *semantic* psv / nsv / rmq (plain scans),
                   with none of the oracle's data structures.
  * GraphBrute  -- definition-level answers computed on the INPUT GRAPH itself: which positions
                   start a path labelled X (paper.tex:262-268 "path graph as an index").
"""
import numpy as np

from workload.index_arrays import unpack_bits


class NaiveIndex:
    def __init__(self, ix):
        self.ix = ix
        self.n, self.e, self.sigma, self.fast_chars = ix.n, ix.e, ix.sigma, ix.fast_chars
        self.C = [int(x) for x in ix.C]
        self.char2comp = ix.char2comp
        self.B = [unpack_bits(ix.bwt[c], ix.n) for c in range(ix.sigma)]
        self.OUT = unpack_bits(ix.edges, ix.e)
        self.SP = unpack_bits(ix.sampled_paths, ix.n)
        self.SM = unpack_bits(ix.samples, ix.sample_count)
        self.VS = [int(v) for v in ix.stored_samples_plain]
        self.filter = unpack_bits(ix.extra_filter, ix.n)
        self.values = unpack_bits(ix.extra_values, ix.extra_values_len)
        self.red = unpack_bits(ix.redundant, ix.redundant_len)
        self.lcp = [int(x) for x in ix.lcp_data[: ix.n]]

    @staticmethod
    def rank(bits, i):
        return int(np.count_nonzero(bits[:i]))

    @staticmethod
    def select(bits, j):
        return int(np.flatnonzero(bits)[j - 1])

    @staticmethod
    def empty(sp, ep):
        M = 1 << 64
        return (sp + 1) % M > (ep + 1) % M

    def charRange(self, c):
        M = 1 << 64
        return (self.rank(self.OUT, self.C[c]), self.rank(self.OUT, (self.C[c + 1] - 1) % M))

    def LF(self, rng, c):
        M = 1 << 64
        sp, ep = rng
        a = self.C[c] + self.rank(self.B[c], sp)
        b = (self.C[c] + self.rank(self.B[c], ep + 1) - 1) % M
        if self.empty(a, b):
            return (a, b)
        return (self.rank(self.OUT, a), self.rank(self.OUT, b))

    def find(self, pattern):
        M = 1 << 64
        if len(pattern) == 0 or self.n == 0:
            return (0, (self.n - 1) % M)
        i = len(pattern) - 1
        rng = self.charRange(int(self.char2comp[pattern[i]]))
        while not self.empty(*rng) and i > 0:
            i -= 1
            rng = self.LF(rng, int(self.char2comp[pattern[i]]))
        return rng

    def LF1(self, i):
        order = list(range(1, self.fast_chars + 1)) + list(range(self.fast_chars + 1, self.sigma))
        for c in order:
            if self.B[c][i]:
                return self.rank(self.OUT, self.C[c] + self.rank(self.B[c], i))
        return self.rank(self.OUT, self.C[0] + self.rank(self.B[0], i))

    def locate1(self, i):
        steps = 0
        while not self.SP[i]:
            i = self.LF1(i)
            steps += 1
        r = self.rank(self.SP, i)
        j = self.select(self.SM, r) + 1 if r > 0 else 0
        out = []
        while True:
            out.append(self.VS[j] + steps)
            j += 1
            if self.SM[j - 1]:
                break
        return out

    def locate(self, rng, sort=True):
        sp, ep = rng
        if self.empty(sp, ep) or ep >= self.n:
            return []
        out = []
        for i in range(sp, ep + 1):
            out.extend(self.locate1(i))
        return sorted(set(out)) if sort else out

    def count(self, rng):
        sp, ep = rng
        if self.empty(sp, ep) or ep >= self.n:
            return 0
        # A[i] decoded from (filter, values): i-th filtered node has value = gap of unary code
        a, b = self.rank(self.filter, sp), self.rank(self.filter, ep + 1)
        extra = 0
        if b > a:
            extra = (self.select(self.values, b) + 1) - (self.select(self.values, a) + 1 if a > 0 else 0)
        res = extra + (ep + 1 - sp)
        if ep > sp:
            s, e = sp, ep - 1
            red = (self.select(self.red, e + 1) - e) - (self.select(self.red, s) + 1 - s if s > 0 else 0)
            res -= red
        return res % (1 << 64)  # count() is unsigned arithmetic; only suffix-tree-node ranges are meaningful

    # --- semantic suffix-tree operations over the LCP array ---
    def psv(self, pos, equal=False):
        nf = (self.ix.lcp_data.shape[0],) * 2
        if pos == 0 or pos >= self.n:
            return nf
        v = self.lcp[pos]
        for i in range(pos - 1, -1, -1):
            if self.lcp[i] < v or (equal and self.lcp[i] == v):
                return (i, self.lcp[i])
        return nf

    def nsv(self, pos, equal=False):
        nf = (self.ix.lcp_data.shape[0],) * 2
        if pos + 1 >= self.n:
            return nf
        v = self.lcp[pos]
        for i in range(pos + 1, self.n):
            if self.lcp[i] < v or (equal and self.lcp[i] == v):
                return (i, self.lcp[i])
        return nf

    def rmq(self, sp, ep):
        nf = (self.ix.lcp_data.shape[0],) * 2
        if sp > ep or ep >= self.n:
            return nf
        seg = self.lcp[sp:ep + 1]
        m = min(seg)
        return (sp + seg.index(m), m)

    def parent(self, rng):
        sp, ep = rng
        nf = (self.ix.lcp_data.shape[0],) * 2
        left_lcp = self.lcp[sp]
        right_lcp = self.lcp[ep + 1] if ep + 1 < self.n else 0
        if sp == 0 and ep == self.n - 1:
            return (0, self.n - 1, 0, 0, 0)
        node_lcp = max(left_lcp, right_lcp)
        left, right = (sp, left_lcp), (ep + 1, right_lcp)
        if left_lcp == node_lcp:
            left = self.psv(sp)
            if left == nf:
                left = (0, 0)
        if right_lcp == node_lcp:
            right = self.nsv(ep + 1)
            if right == nf:
                right = (self.n, 0)
        return (left[0], right[0] - 1, left[1], right[1], node_lcp)

    def depth(self, rng):
        sp, ep = rng
        if (ep + 1 - sp) % (1 << 64) <= 1:
            return (1 << 64) - 1
        r = self.rmq(sp + 1, ep)
        if r == (self.ix.lcp_data.shape[0],) * 2:
            return (1 << 64) - 1
        return r[1]


class GraphBrute:
    """Answers derived from the input graph alone."""

    def __init__(self, graph):
        self.g = graph
        self.comp = [int(c) for c in graph.comp]
        self.succ = [list(map(int, graph.successors(v))) for v in range(graph.size)]

    def starts(self, comps):
        """Positions v such that some path starting at v spells `comps`."""
        if len(comps) == 0:
            return set(range(self.g.size))
        cur = {v for v in range(self.g.size) if self.comp[v] == comps[-1]}
        for c in reversed(comps[:-1]):
            cur = {v for v in range(self.g.size) if self.comp[v] == c and any(w in cur for w in self.succ[v])}
        return cur

    def occurrences(self, comps):
        return sorted(int(self.g.value[v]) for v in self.starts(comps))
