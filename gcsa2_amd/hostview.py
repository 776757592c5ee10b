"""ctypes mirror of `gcsa2_host_view` / `gcsa2_stnode` (include/gcsa2_hip.h).

A host view describes the data members of `gcsa::GCSA` (reference `include/gcsa/gcsa.h:214-240`)
and `gcsa::LCPArray` (`include/gcsa/lcp.h:188-190`) as plain LSB-first bit arrays and integer
arrays.  `make_host_view()` accepts any object with the attribute names used by
`workload.index_arrays.IndexArrays`; the numpy arrays it references are kept alive by the
returned holder.  The bulk arrays (bwt[c], edges, the sample / counter bit arrays, stored_samples,
lcp_data) may also be torch tensors in device memory: the image is built on the device and reads
them from wherever they are.  char2comp, C and lcp_offsets are read by the host.
"""
import ctypes as C
import numpy as np

MAX_SIGMA = 16
UNKNOWN = (1 << 64) - 1

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


class HostView(C.Structure):
    _fields_ = [
        ("path_nodes", C.c_uint64), ("edges", C.c_uint64), ("order", C.c_uint64),
        ("sigma", C.c_uint64), ("fast_chars", C.c_uint64),
        ("char2comp", u8p), ("C", u64p),
        ("bwt", C.POINTER(u64p)), ("edge_bits", u64p), ("sampled_path_bits", u64p),
        ("sample_count", C.c_uint64), ("sample_width", C.c_uint64),
        ("stored_samples", u64p), ("sample_bits", u64p),
        ("extra_filter_bits", u64p), ("extra_values_len", C.c_uint64), ("extra_values_bits", u64p),
        ("redundant_len", C.c_uint64), ("redundant_bits", u64p),
        ("lcp_size", C.c_uint64), ("lcp_branching", C.c_uint64), ("lcp_levels", C.c_uint64),
        ("lcp_offsets", u64p), ("lcp_data", u8p),
        ("comp2char", u8p),
    ]


class STNode(C.Structure):
    """Field for field `gcsa::STNode` (reference `include/gcsa/lcp.h:40-79`)."""
    _fields_ = [("sp", C.c_uint64), ("ep", C.c_uint64), ("left_lcp", C.c_uint64),
                ("right_lcp", C.c_uint64), ("node_lcp", C.c_uint64)]

    def range(self):
        return (self.sp, self.ep)

    def lcp(self):
        return self.node_lcp

    def astuple(self):
        return (self.sp, self.ep, self.left_lcp, self.right_lcp, self.node_lcp)


STNODE_DTYPE = np.dtype([("sp", "<u8"), ("ep", "<u8"), ("left_lcp", "<u8"), ("right_lcp", "<u8"),
                         ("node_lcp", "<u8")])


def _is_device_tensor(a):
    return hasattr(a, "data_ptr") and hasattr(a, "is_cuda")


def _u64(a):
    if _is_device_tensor(a):
        # a bulk array already in device (or pinned host) memory: gcsa2_index_create copies it with hipMemcpyDefault
        assert a.is_contiguous() and a.element_size() == 8, "device arrays must be contiguous 64-bit words"
        return a, C.cast(a.data_ptr(), u64p)
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def _u8(a):
    if _is_device_tensor(a):
        assert a.is_contiguous() and a.element_size() == 1, "device byte arrays must be contiguous"
        return a, C.cast(a.data_ptr(), u8p)
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(u8p)


class HostViewHolder:
    """Owns the numpy buffers a HostView points into."""

    def __init__(self, ix, with_samples=True, with_counters=True, with_lcp=True):
        keep = []
        v = HostView()
        v.path_nodes, v.edges, v.order = int(ix.n), int(ix.e), int(ix.order)
        v.sigma, v.fast_chars = int(ix.sigma), int(ix.fast_chars)
        a, v.char2comp = _u8(ix.char2comp); keep.append(a)
        a, v.C = _u64(ix.C); keep.append(a)
        ptrs = (u64p * int(ix.sigma))()
        for c in range(int(ix.sigma)):
            a, ptrs[c] = _u64(ix.bwt[c]); keep.append(a)
        keep.append(ptrs)
        v.bwt = C.cast(ptrs, C.POINTER(u64p))
        a, v.edge_bits = _u64(ix.edges); keep.append(a)
        if with_samples:
            a, v.sampled_path_bits = _u64(ix.sampled_paths); keep.append(a)
            v.sample_count, v.sample_width = int(ix.sample_count), int(ix.sample_width)
            a, v.stored_samples = _u64(ix.stored_samples); keep.append(a)
            a, v.sample_bits = _u64(ix.samples); keep.append(a)
        if with_counters:
            a, v.extra_filter_bits = _u64(ix.extra_filter); keep.append(a)
            v.extra_values_len = int(ix.extra_values_len)
            a, v.extra_values_bits = _u64(ix.extra_values); keep.append(a)
            v.redundant_len = int(ix.redundant_len)
            a, v.redundant_bits = _u64(ix.redundant); keep.append(a)
        if with_lcp:
            v.lcp_size, v.lcp_branching = int(ix.lcp_size), int(ix.lcp_branching)
            v.lcp_levels = int(ix.lcp_offsets.shape[0]) - 1
            a, v.lcp_offsets = _u64(ix.lcp_offsets); keep.append(a)
            a, v.lcp_data = _u8(ix.lcp_data); keep.append(a)
        self.view = v
        self._keep = keep

    def ref(self):
        return C.byref(self.view)


def make_host_view(ix, **kw) -> HostViewHolder:
    return HostViewHolder(ix, **kw)


def concat_patterns(patterns):
    """List of bytes-like -> (uint8 concatenation, uint64 offsets[nq+1])."""
    lens = np.fromiter((len(p) for p in patterns), dtype=np.uint64, count=len(patterns))
    offsets = np.zeros(len(patterns) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    data = np.frombuffer(b"".join(bytes(p) for p in patterns), dtype=np.uint8)
    if data.shape[0] == 0:
        data = np.zeros(1, dtype=np.uint8)
    return np.ascontiguousarray(data), offsets


def pack_kmers(patterns: np.ndarray, char2comp=None) -> np.ndarray:
    """(nq, m) pattern bytes over the fast characters -> (nq, ceil(m / 32)) uint64 code words in the layout of
    gcsa2_find_batch_packed (include/gcsa2_hip.h): last character first, comp - 1 in two bits.  Raises ValueError when a
    pattern holds a character outside comps 1..4 (such patterns go through the byte interface)."""
    patterns = np.ascontiguousarray(patterns, dtype=np.uint8)
    nq, m = patterns.shape
    if char2comp is None:
        char2comp = np.full(256, 5, dtype=np.uint8)
        for ch, c in ((b"$", 0), (b"\0", 0), (b"A", 1), (b"a", 1), (b"C", 2), (b"c", 2), (b"G", 3), (b"g", 3), (b"T", 4), (b"t", 4), (b"#", 6)):
            char2comp[ch[0]] = c
    comps = np.asarray(char2comp, dtype=np.uint8)[patterns].astype(np.int64) - 1
    if ((comps < 0) | (comps > 3)).any():
        raise ValueError("pack_kmers: a pattern holds a character outside comps 1..4")
    words = (m + 31) // 32
    out = np.zeros((nq, words), dtype=np.uint64)
    rev = comps[:, ::-1].astype(np.uint64)                    # distance t from the end at column t
    for t in range(m):
        out[:, t >> 5] |= rev[:, t] << np.uint64(2 * (t & 31))
    return out
