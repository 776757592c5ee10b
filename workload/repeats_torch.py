"""A repeat-rich text at HBM footprint: the range widths of real genomes on an index far beyond the Infinity Cache.

The paper's human indexes answer a found 32-mer with 336 path nodes on average and a 16-mer with 7129
(paper/paper.tex:403,408).  workload/graphs.py::repeat_bases plants repeat families in a 2^23-base backbone that goes
through the general builder; this module is its torch twin for 2^30 bases, built for workload/linear_torch.py, whose
prefix doubling needs every order-256 path to be distinct.  Hence FAMILIES: the backbone is cut into runs of
`family_blocks` blocks of 600 bases and every run has its own consensus sequences, so that a 32-mer of a repeat copy
matches the thousands of copies of its family that carry no substitution in those 32 bases, while no two copies agree
over 256 bases (divergence 7 % / 6 %: the chance of two copies identical over a 256-base window is below 1e-13 per pair).

Per block of 600 bases (kind = a hash of the block number):
  14 in 16   one copy of the family's 300-base interspersed repeat at a random offset, 7 % of its bases substituted
   1 in 16   one copy of the family's 600-base young repeat, 6 % substituted
   1 in 16   a tandem array: a unit of 2..7 bases repeated over 100..200 bases (shorter than the order, so that every
             256-base window reaches past the array)

Workload generation only (torch ops; on the GPU at full size, on the CPU in tests/test_workload.py, where the index of
such a text equals what the general builder produces).
"""
import torch

from .linear_torch import _lsr, _s64, random_bases_torch, splitmix64_torch

REPEAT_BLOCK = 600
ALU_LEN = 300
FAMILY_BLOCKS = 1 << 17          # 78.6 M bases per family: 115 k interspersed copies, 8 k young copies


def _mix(z: torch.Tensor) -> torch.Tensor:
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _hash(idx: torch.Tensor, seed: int) -> torch.Tensor:
    return _mix(idx * _s64(0x9E3779B97F4A7C15) + _s64(seed))


def _consensus(fam: torch.Tensor, length: int, seed: int) -> torch.Tensor:
    """(len(fam), length) comps 1..4: the family's consensus, a hash of (family, position)."""
    j = torch.arange(length, dtype=torch.int64, device=fam.device).view(1, -1)
    return (_lsr(_hash(fam.view(-1, 1) * 4096 + j, seed), 33) % 4) + 1


def _mutate(pos: torch.Tensor, cons: torch.Tensor, divergence: float, seed: int) -> torch.Tensor:
    """Every base of a copy is substituted with probability `divergence`, independently (a hash of its text position)."""
    r = _hash(pos, seed)
    hit = (_lsr(r, 11) % 10000) < int(divergence * 10000)
    shift = (_lsr(r, 40) % 3) + 1
    return torch.where(hit, (cons - 1 + shift) % 4 + 1, cons).to(torch.uint8)


def repeat_bases_torch(n: int, seed: int, device, family_blocks: int = FAMILY_BLOCKS, alu_divergence: float = 0.07,
                       young_divergence: float = 0.06) -> torch.Tensor:
    """n comp codes (1..4) on `device`: random bases with the planted repeat families described above."""
    seq = random_bases_torch(n, seed, device)
    blocks = n // REPEAT_BLOCK
    if blocks == 0:
        return seq
    h_all = splitmix64_torch(seed ^ 0x5EED5EED, blocks, device)
    chunk = 1 << 17
    for b0 in range(0, blocks, chunk):
        b1 = min(blocks, b0 + chunk)
        blk = torch.arange(b0, b1, dtype=torch.int64, device=device)
        h = h_all[b0:b1]
        kind = _lsr(h, 4) % 16
        fam = blk // family_blocks
        base = blk * REPEAT_BLOCK
        sel = torch.nonzero(kind >= 2).view(-1)
        if sel.numel():
            off = _lsr(h[sel], 20) % (REPEAT_BLOCK - ALU_LEN)
            pos = (base[sel] + off).view(-1, 1) + torch.arange(ALU_LEN, dtype=torch.int64, device=device).view(1, -1)
            seq[pos] = _mutate(pos, _consensus(fam[sel], ALU_LEN, seed ^ 0xA1), alu_divergence, seed ^ 0xB1)
        sel = torch.nonzero(kind == 0).view(-1)
        if sel.numel():
            pos = base[sel].view(-1, 1) + torch.arange(REPEAT_BLOCK, dtype=torch.int64, device=device).view(1, -1)
            seq[pos] = _mutate(pos, _consensus(fam[sel], REPEAT_BLOCK, seed ^ 0xA2), young_divergence, seed ^ 0xB2)
        sel = torch.nonzero(kind == 1).view(-1)
        if sel.numel():
            hj = h[sel]
            unit_len = 2 + _lsr(hj, 24) % 6
            length = 100 + _lsr(hj, 32) % 101
            start = base[sel] + _lsr(hj, 44) % (REPEAT_BLOCK - length)
            unit = (_lsr(_hash(hj.view(-1, 1) * 8 + torch.arange(7, dtype=torch.int64, device=device).view(1, -1), seed ^ 0xA3), 33) % 4) + 1
            t = torch.arange(200, dtype=torch.int64, device=device).view(1, -1)
            vals = torch.gather(unit, 1, t % unit_len.view(-1, 1)).to(torch.uint8)
            keep = t < length.view(-1, 1)
            pos = start.view(-1, 1) + t
            seq[pos[keep]] = vals[keep]
    return seq


def substring_patterns_device(seq: torch.Tensor, nq: int, m: int, pat_seed: int, first: int = 0):
    """Queries first .. first + nq - 1 of the batch `pat_seed`: (nq, m) ASCII bytes, substrings of the backbone at
    SplitMix64 positions (every pattern occurs), and their start positions (backbone index)."""
    from .mseq_torch import splitmix64_range_torch
    n = int(seq.shape[0])
    r = splitmix64_range_torch(pat_seed, first, nq, seq.device)
    start = _lsr(r, 11) % (n - m)
    lut = torch.tensor(list(b"$ACGTN#"), dtype=torch.uint8, device=seq.device)
    out = torch.empty((nq, m), dtype=torch.uint8, device=seq.device)
    chunk = 1 << 22
    offs = torch.arange(m, dtype=torch.int64, device=seq.device).view(1, -1)
    for b in range(0, nq, chunk):
        e = min(nq, b + chunk)
        out[b:e] = lut[seq[start[b:e].view(-1, 1) + offs].to(torch.int64)]
    return out, start


def count_occurrences_device(seq: torch.Tensor, patterns: torch.Tensor) -> torch.Tensor:
    """Definition-level check of find(): the number of positions of the backbone where each pattern (a row of ASCII
    bytes, length m <= 32) occurs, by comparing the packed 2-bit code of every m-base window with the pattern's."""
    n, m = int(seq.shape[0]), int(patterns.shape[1])
    assert m <= 32
    comp = torch.zeros(256, dtype=torch.int64, device=seq.device)
    for ch, c in zip(b"ACGT", range(4)):
        comp[ch] = c
    want = torch.zeros(patterns.shape[0], dtype=torch.int64, device=seq.device)
    for j in range(m):
        want = (want << 2) | comp[patterns[:, j].to(torch.int64)]
    counts = torch.zeros(patterns.shape[0], dtype=torch.int64, device=seq.device)
    chunk = 1 << 27
    for b in range(0, n - m + 1, chunk):
        e = min(n - m + 1, b + chunk)
        code = torch.zeros(e - b, dtype=torch.int64, device=seq.device)
        for j in range(m):
            code = (code << 2) | (seq[b + j: e + j].to(torch.int64) - 1)
        for q in range(patterns.shape[0]):
            counts[q] += (code == want[q]).sum()
    return counts


def occurrence_values_device(seq: torch.Tensor, pattern: torch.Tensor, node_len: int = 32, id_offset: int = 11) -> torch.Tensor:
    """Definition-level check of locate() on the linear graph of workload/linear_torch.py: the sorted values of the path nodes
    whose suffixes start with `pattern` (one row of ASCII bytes, length m <= 32) -- an occurrence at backbone index p has the value
    ((p // node_len) + 1) << id_offset | (p % node_len) -- found by comparing the packed 2-bit code of every m-base window."""
    n, m = int(seq.shape[0]), int(pattern.shape[0])
    assert m <= 32
    comp = torch.zeros(256, dtype=torch.int64, device=seq.device)
    for ch, c in zip(b"ACGT", range(4)):
        comp[ch] = c
    want = 0
    for j in range(m):
        want = (want << 2) | int(comp[int(pattern[j])])
    found = []
    chunk = 1 << 27
    for b in range(0, n - m + 1, chunk):
        e = min(n - m + 1, b + chunk)
        code = torch.zeros(e - b, dtype=torch.int64, device=seq.device)
        for j in range(m):
            code = (code << 2) | (seq[b + j: e + j].to(torch.int64) - 1)
        found.append(torch.nonzero(code == want).reshape(-1) + b)
    p = torch.cat(found) if found else torch.zeros(0, dtype=torch.int64, device=seq.device)
    values = ((torch.div(p, node_len, rounding_mode="floor") + 1) << id_offset) | (p % node_len)
    return torch.sort(values).values
