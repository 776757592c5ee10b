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
