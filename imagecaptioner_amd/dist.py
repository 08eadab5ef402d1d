"""Multi-GPU layout of a batch: one process per GPU, files sharded by contiguous ranges.

Files are independent (the reference processes them with Parallel.ForEachAsync, Program.cs:197-199), so the
only exchange step is the sum of the per-GPU tag histograms and batch counters: one all-reduce over RCCL
(torch.distributed backend "nccl") on the GPU, or gloo in CPU tests.  Concatenating the shards' captions in
rank order reproduces the single-GPU output, because shards are contiguous ranges of the input order.
"""


def shard_range(n_files, rank, world):
    """Contiguous [begin, end) of files owned by `rank`; sizes differ by at most one, earlier ranks get the extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_files, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_offsets(offsets, begin, end):
    """Offsets of files [begin, end) rebased to their first byte, plus that byte position in the full arena."""
    first = int(offsets[begin])
    return offsets[begin:end + 1] - offsets[begin], first


def allreduce_sum_(tensors, group=None):
    """In-place sum of each tensor over all ranks (histogram, counters).  No-op when torch.distributed is not initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tensors
