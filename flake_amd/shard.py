"""Frame sharding across ranks.

Frames are independent units: a frame's encoding depends only on its own block
and the static parameters (encode.c:919-947).  A job of `total_frames` is cut
into contiguous ranges, one per rank, so that the job's output is the
concatenation of the ranks' outputs in rank order and the frame number of
frame i is just i (encode.c:740, :970-974).  No data-path collective exists;
the only exchange is the final reduction of a few counters.
"""
from __future__ import annotations


def shard_range(total_frames: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [first, last) frame range of `rank`; sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world or total_frames < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total_frames, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def reduce_job_stats(frames: int, residual_bits: int, max_frame_bytes: int = 0, device=None):
    """All-reduce of the job counters over the default process group (RCCL on
    GPUs, gloo on CPU): (sum frames, sum residual bits, max frame bytes) -- the
    cross-frame state libflake keeps in its context (encode.c:967-974)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return frames, residual_bits, max_frame_bytes
    sums = torch.tensor([frames, residual_bits], dtype=torch.int64, device=device)
    mx = torch.tensor([max_frame_bytes], dtype=torch.int64, device=device)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    return int(sums[0]), int(sums[1]), int(mx[0])


def gather_frame_sizes(sizes, device=None):
    """All-gather of per-frame byte counts so that every rank can place its
    frames in the job's output stream (exclusive prefix sum over ranks)."""
    import torch
    import torch.distributed as dist

    t = torch.as_tensor(sizes, dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return [t]
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(dist.get_world_size())]
    dist.all_gather(counts, torch.tensor([t.numel()], dtype=torch.int64, device=device))
    width = int(max(int(c) for c in counts))
    pad = torch.zeros(width, dtype=torch.int64, device=device)
    pad[:t.numel()] = t
    out = [torch.zeros(width, dtype=torch.int64, device=device) for _ in counts]
    dist.all_gather(out, pad)
    return [o[:int(c)] for o, c in zip(out, counts)]
