"""Multi-GPU driver: replicated index, query batch sharded over ranks, one gather of hit ranges.

The reference's only data-parallel query path is the static contiguous split of `verifyIndex`
(reference src/algorithms.cpp:106-114); queries are independent and the index is read-only, so
the path shards with no data-path collective.  One process per GPU; the single collective is the
final gather of `(sp, ep)` pairs (16 B per query) on rank 0 -- RCCL over xGMI when the tensors
are on the GPU (`backend="nccl"`), gloo in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_queries: int, world: int):
    """Contiguous shards, sizes differing by at most one: [(begin, end)] per rank."""
    base, rem = divmod(n_queries, world)
    bounds, start = [], 0
    for r in range(world):
        size = base + (1 if r < rem else 0)
        bounds.append((start, start + size))
        start += size
    return bounds


def slice_batch(flat: np.ndarray, offsets: np.ndarray, begin: int, end: int):
    """Sub-batch [begin, end) of a concatenated pattern batch, offsets rebased to 0."""
    lo, hi = int(offsets[begin]), int(offsets[end])
    sub = flat[lo:hi] if hi > lo else np.zeros(1, dtype=np.uint8)
    return np.ascontiguousarray(sub), (offsets[begin:end + 1] - offsets[begin]).astype(np.uint64)


def gather_ranges(local: torch.Tensor, n_queries: int, group=None):
    """Gather per-rank (n_r, 2) int64 range tensors on rank 0 in shard order.

    Shards are padded to the largest shard so that one fixed-size gather suffices.
    Returns the (n_queries, 2) tensor on rank 0 and None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = shard_bounds(n_queries, world)
    width = max(e - b for b, e in bounds)
    padded = torch.zeros((width, 2), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.zeros_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, parts, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([parts[r][: e - b] for r, (b, e) in enumerate(bounds)], dim=0)


def find_sharded(compute, flat: np.ndarray, offsets: np.ndarray, group=None):
    """Every rank holds the whole batch description; rank r searches shard r with
    `compute(flat_r, offsets_r) -> (n_r, 2) int64 tensor` and rank 0 receives all ranges."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nq = int(offsets.shape[0]) - 1
    b, e = shard_bounds(nq, world)[rank]
    sub_flat, sub_off = slice_batch(flat, offsets, b, e)
    local = compute(sub_flat, sub_off)
    return gather_ranges(local, nq, group)


def gpu_find_compute(gpu, device):
    """`compute` for find_sharded that runs the HIP engine on `device` with inputs staged through
    torch tensors (device memory + current stream are torch plumbing)."""
    def compute(sub_flat, sub_off):
        n = int(sub_off.shape[0]) - 1
        # 8 spare bytes: the device entry points read patterns in aligned 8-byte words (include/gcsa2_hip.h)
        d_pat = torch.from_numpy(np.concatenate([sub_flat, np.zeros(8, dtype=np.uint8)])).to(device)
        d_off = torch.from_numpy(sub_off.view(np.int64)).to(device)
        d_out = torch.zeros((n, 2), dtype=torch.int64, device=device)
        if n > 0:
            gpu.find_device(d_pat.data_ptr(), d_off.data_ptr(), n, d_out.data_ptr(),
                            torch.cuda.current_stream(device).cuda_stream)
        return d_out
    return compute


def gather_variable(local: np.ndarray, sizes, group=None):
    """Gather 1-D numpy arrays of per-rank lengths `sizes` (known on every rank) on rank 0, concatenated in rank order.
    The parts travel as bytes, padded to the longest (torch's gather wants equal shapes; the library's own gather,
    gcsa2_comm_gather, sends exact sizes)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    item = local.dtype.itemsize
    width = max(max(sizes), 1) * item
    padded = torch.zeros(width, dtype=torch.uint8)
    raw = np.ascontiguousarray(local).view(np.uint8).reshape(-1)
    padded[: raw.shape[0]] = torch.from_numpy(raw.copy())
    parts = [torch.zeros_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, parts, dst=0, group=group)
    if rank != 0:
        return None
    return np.concatenate([parts[r][: sizes[r] * item].numpy() for r in range(world)]).view(local.dtype)


def locate_sharded(compute, ranges: np.ndarray, group=None):
    """locate() of a batch of ranges split contiguously over the ranks, the CSR result gathered on rank 0 in query order
    -- the control flow of gcsa2_comm_locate: per-rank totals first, then the offsets of every shard
    (rebased by the totals of the shards before it) and the values.  `compute(sub_ranges) -> (offsets uint64[n + 1],
    values uint64[offsets[n]])`.  Returns (offsets, values) on rank 0, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nq = int(ranges.shape[0])
    bounds = shard_bounds(nq, world)
    b, e = bounds[rank]
    off, val = compute(ranges[b:e])
    totals = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(totals, torch.tensor([int(off[-1])], dtype=torch.int64), group=group)
    totals = [int(t.item()) for t in totals]
    offsets = gather_variable(off[:-1].astype(np.uint64), [hi - lo for lo, hi in bounds], group)
    values = gather_variable(val.astype(np.uint64), totals, group)
    if rank != 0:
        return None
    out = np.zeros(nq + 1, dtype=np.uint64)
    base = 0
    for r, (lo, hi) in enumerate(bounds):
        out[lo:hi] = offsets[lo:hi] + np.uint64(base)
        base += totals[r]
    out[nq] = base
    return out, values


def match_stats_sharded(compute, flat: np.ndarray, offsets: np.ndarray, group=None):
    """Matching statistics of a batch split contiguously over the ranks (the control flow of gcsa2_comm_match_stats):
    `compute(flat_r, offsets_r) -> (ms uint16[bytes_r], ranges (n_r, 2) uint64, fallbacks uint64[n_r])`; rank 0 receives
    the three arrays of the whole batch in query order."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nq = int(offsets.shape[0]) - 1
    bounds = shard_bounds(nq, world)
    b, e = bounds[rank]
    sub_flat, sub_off = slice_batch(flat, offsets, b, e)
    ms, rng, fb = compute(sub_flat, sub_off)
    counts = [hi - lo for lo, hi in bounds]
    nbytes = [int(offsets[hi] - offsets[lo]) for lo, hi in bounds]
    g_ms = gather_variable(ms.astype(np.uint16), nbytes, group)
    g_rng = gather_variable(rng.astype(np.uint64).reshape(-1), [2 * c for c in counts], group)
    g_fb = gather_variable(fb.astype(np.uint64), counts, group)
    if rank != 0:
        return None
    return g_ms, g_rng.reshape(-1, 2), g_fb
