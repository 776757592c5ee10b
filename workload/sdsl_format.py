"""Writer for `.gcsa` / `.lcp` files in the byte format of GCSA::serialize / LCPArray::serialize
(reference src/gcsa.cpp:140-179, src/lcp.cpp:116-128), from an IndexArrays (test infrastructure).

Member order and the GCSA-level headers are the reference's.  The encodings of the SDSL containers
(int_vector, bit_vector_il<512>, sd_vector<>, select_support_mcl) are NOT in the reference tree: they
are restated from sdsl-lite 2.1.1 as summarised in SURVEY.md section 8(f)-1, including the layout of the
select_support_mcl directories (4096-argument superblocks, "long" blocks holding every position, "mini"
blocks holding every 64th offset) and bit_vector_il's binary-search rank samples.  FORMAT PARITY
UNPINNED: no file written by the real library exists here to compare with, so these files pin the
product's reader (gcsa2_amd/csrc/sdsl_reader.hpp) against this restatement only.
"""
import struct

import numpy as np

GCSA_TAG, GCSA_VERSION = 0x6C5A6C5A, 3      # include/gcsa/files.h:144-145, utils.h:165
LCP_TAG, LCP_VERSION = 0x6C5A7C94, 1        # include/gcsa/files.h:178-179, utils.h:166


def _hi(x):
    """sdsl::bits::hi: position of the most significant set bit (0 for x == 0)."""
    return int(x).bit_length() - 1 if x > 0 else 0


def _words(bits_words, nbits):
    w = np.zeros((nbits + 63) // 64, dtype=np.uint64)
    src = np.asarray(bits_words, dtype=np.uint64)[: len(w)]
    w[: len(src)] = src
    if nbits & 63 and len(w):
        w[-1] &= np.uint64((1 << (nbits & 63)) - 1)
    return w


def _positions(words, nbits):
    bits = np.unpackbits(_words(words, nbits).view(np.uint8), bitorder="little")[:nbits]
    return np.flatnonzero(bits).astype(np.uint64)


def pack(values, width):
    """LSB-first packing of `values` into 64-bit words, `width` bits each."""
    values = np.asarray(values, dtype=np.uint64)
    n = len(values)
    if n == 0:
        return np.zeros(0, dtype=np.uint64)
    bits = ((values[:, None] >> np.arange(width, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint8).reshape(-1)
    pad = (-len(bits)) % 64
    bits = np.concatenate([bits, np.zeros(pad, dtype=np.uint8)])
    return np.packbits(bits, bitorder="little").view(np.uint64)


def int_vector(values, width, fixed):
    """sdsl::int_vector<w>: u64 size in bits, [u8 width if w == 0], ceil(bits / 64) words."""
    values = np.asarray(values, dtype=np.uint64)
    head = struct.pack("<Q", len(values) * width) + (b"" if fixed else struct.pack("<B", width))
    if width == 64:
        return head + values.tobytes()
    if width == 8 and len(values) % 8 == 0:
        return head + values.astype(np.uint8).tobytes()
    return head + pack(values, width).tobytes()


def bit_vector(words, nbits):
    return struct.pack("<Q", nbits) + _words(words, nbits).tobytes()


def bit_vector_il(words, nbits):
    """sdsl::bit_vector_il<512>: size, block_num, superblocks, block_shift, data, rank_samples."""
    payload = (nbits + 64) // 64
    superblocks = (nbits + 512) // 512
    mem = payload + superblocks + 1
    w = np.zeros(payload, dtype=np.uint64)
    src = _words(words, nbits)
    w[: len(src)] = src
    counts = np.concatenate([[0], np.cumsum(np.bitwise_count(w) if hasattr(np, "bitwise_count") else _popcount(w))]).astype(np.uint64)
    data = np.zeros(mem, dtype=np.uint64)
    idx = np.arange(payload)
    data[idx + idx // 8 + 1] = w
    sb = np.arange(superblocks)
    data[sb * 9] = counts[np.minimum(sb * 8, payload)]
    data[mem - 1] = counts[payload]
    samples = np.zeros(0, dtype=np.uint64)
    if mem > 1024 * 64:                       # init_rank_samples: breadth-first midpoints of the binary search
        want = min(1024 * 64, superblocks)
        out, queue, head = [], [(0, superblocks)], 0
        while head < len(queue) and len(out) < want:
            lb, rb = queue[head]
            head += 1
            mid = lb + (rb - lb) // 2
            out.append(int(data[mid * 9]) if mid * 9 < mem else 0)
            queue.append((lb, mid))
            queue.append((mid + 1, rb))
        samples = np.asarray(out, dtype=np.uint64)
    return (struct.pack("<QQQQ", nbits, mem, superblocks, 9) + int_vector(data, 64, True) + int_vector(samples, 64, True))


def empty_bit_vector_il():
    return struct.pack("<QQQQ", 0, 0, 0, 0) + int_vector([], 64, True) + int_vector([], 64, True)


def _popcount(w):
    return np.unpackbits(w.view(np.uint8)).reshape(len(w), 64).sum(axis=1)


def select_support_mcl(words, nbits, bit):
    """sdsl::select_support_mcl<bit, 1> over a bit_vector of `nbits` bits."""
    bits = np.unpackbits(_words(words, nbits).view(np.uint8), bitorder="little")[:nbits]
    args = np.flatnonzero(bits == bit).astype(np.int64)
    out = struct.pack("<Q", len(args))
    if len(args) == 0:
        return out
    capacity = ((nbits + 63) // 64) * 64
    logn = _hi(capacity) + 1
    logn4 = logn ** 4
    sb = (len(args) + 4095) // 4096
    out += int_vector(args[::4096], logn, False)
    blocks, kinds = [], []
    for s in range(sb):
        a = args[4096 * s: 4096 * (s + 1)]
        diff = int(a[-1] - a[0])
        if diff > logn4:                      # long superblock: every position, absolute
            full = np.zeros(4096, dtype=np.uint64)
            full[: len(a)] = a
            blocks.append(int_vector(full, _hi(int(a[-1])) + 1, False))
            kinds.append(0)
        else:                                 # mini blocks: every 64th position, relative to the first
            mini = np.zeros(64, dtype=np.uint64)
            sub = (a[::64] - a[0]).astype(np.uint64)
            mini[: len(sub)] = sub
            blocks.append(int_vector(mini, _hi(diff) + 1, False))
            kinds.append(1)
    if 0 in kinds:
        out += bit_vector(pack(np.asarray(kinds, dtype=np.uint64), 1), sb)
    else:
        out += bit_vector([], 0)
    return out + b"".join(blocks)


def sd_vector(words, nbits):
    """sdsl::sd_vector<>: size, wl, low, high, select_1 and select_0 over high."""
    pos = _positions(words, nbits)
    m = len(pos)
    logm, logn = _hi(m) + 1, _hi(nbits) + 1
    if logm == logn:
        logm -= 1
    wl = logn - logm
    low = pos & np.uint64((1 << wl) - 1)
    high_len = m + (1 << logm)
    high_pos = (pos >> np.uint64(wl)) + np.arange(m, dtype=np.uint64)
    high_bits = np.zeros(((high_len + 63) // 64) * 64, dtype=np.uint8)
    high_bits[high_pos.astype(np.int64)] = 1
    high = np.packbits(high_bits, bitorder="little").view(np.uint64)
    return (struct.pack("<QB", nbits, wl) + int_vector(low, wl, False) + bit_vector(high, high_len)
            + select_support_mcl(high, high_len, 1) + select_support_mcl(high, high_len, 0))


def empty_sd_vector():
    return struct.pack("<QB", 0, 0) + int_vector([], 64, False) + bit_vector([], 0) + struct.pack("<QQ", 0, 0)


def gcsa_bytes(ix):
    """The `.gcsa` byte stream of an IndexArrays (GCSA::serialize, src/gcsa.cpp:140-179)."""
    sigma, n = int(ix.sigma), int(ix.n)
    out = [struct.pack("<IIQQQQ", GCSA_TAG, GCSA_VERSION, n, int(ix.e), int(ix.order), 0)]       # files.cpp:513-525
    comp2char = np.zeros(sigma, dtype=np.uint64)
    c2c = np.asarray(ix.char2comp, dtype=np.uint64)
    for c in range(sigma):
        hits = np.flatnonzero(c2c == c)
        # the reference's default alphabet maps both cases to one comp and prints the upper-case one
        upper = [int(h) for h in hits if chr(int(h)).upper() == chr(int(h))]
        comp2char[c] = (upper[0] if upper else (int(hits[0]) if len(hits) else 0))
    if sigma == 7 and all(int(c2c[b]) == c for c, b in enumerate(b"$ACGTN#")):
        comp2char[:] = np.frombuffer(b"$ACGTN#", dtype=np.uint8)     # Alphabet::DEFAULT_COMP2CHAR (src/support.cpp:69-92)
    out += [int_vector(c2c, 8, True), int_vector(comp2char, 8, True), int_vector(np.asarray(ix.C, dtype=np.uint64), 64, True),
            struct.pack("<QQ", sigma, int(ix.fast_chars))]                                         # support.cpp:229-240
    fast = [1 <= c <= int(ix.fast_chars) for c in range(sigma)]
    out += [bit_vector_il(ix.bwt[c], n) if fast[c] else empty_bit_vector_il() for c in range(sigma)]     # fast_bwt; fast_rank is empty
    out += [empty_sd_vector() if fast[c] else sd_vector(ix.bwt[c], n) for c in range(sigma)]             # sparse_bwt; sparse_rank is empty
    out += [bit_vector_il(ix.edges, int(ix.e)), bit_vector_il(ix.sampled_paths, n)]
    width = int(ix.sample_width)
    out += [struct.pack("<QB", int(ix.sample_count) * width, width),
            _words(ix.stored_samples, int(ix.sample_count) * width).tobytes(),
            bit_vector(ix.samples, int(ix.sample_count)), select_support_mcl(ix.samples, int(ix.sample_count), 1)]
    out += [sd_vector(ix.extra_filter, n), sd_vector(ix.extra_values, int(ix.extra_values_len))]   # SadaSparse, support.cpp:493-503
    out += [bit_vector(ix.redundant, int(ix.redundant_len)),
            select_support_mcl(ix.redundant, int(ix.redundant_len), 1)]                                  # SadaCount, support.cpp:401-408
    return b"".join(out)


def lcp_bytes(ix):
    """The `.lcp` byte stream (LCPArray::serialize, src/lcp.cpp:116-128); the data vector is bit-compressed to the
    width of its largest value, as the reference's constructor leaves it (sdsl::util::bit_compress, src/lcp.cpp:258)."""
    data = np.asarray(ix.lcp_data, dtype=np.uint8)
    width = _hi(int(data.max()) if len(data) else 0) + 1
    head = struct.pack("<IIQQQ", LCP_TAG, LCP_VERSION, int(ix.lcp_size), int(ix.lcp_branching), 0)   # files.cpp:581-593
    if width == 8:
        padded = np.zeros(((len(data) + 7) // 8) * 8, dtype=np.uint8)
        padded[: len(data)] = data
        body = struct.pack("<QB", len(data) * 8, 8) + padded.tobytes()
    else:
        body = int_vector(data.astype(np.uint64), width, False)
    return head + body + int_vector(np.asarray(ix.lcp_offsets, dtype=np.uint64), 64, True)


def write(ix, base):
    """Writes base.gcsa and base.lcp (the pair benchmark/query_gcsa.cpp:53-63 opens)."""
    with open(base + ".gcsa", "wb") as f:
        f.write(gcsa_bytes(ix))
    with open(base + ".lcp", "wb") as f:
        f.write(lcp_bytes(ix))
    return base + ".gcsa", base + ".lcp"
