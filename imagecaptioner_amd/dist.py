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


def shard_digest(captions, caption_offsets=None):
    """(sha256 hex digest, byte count, file count) of one shard's caption bytes: what a rank hands to rank 0."""
    import hashlib
    import numpy as np
    b = np.ascontiguousarray(captions, dtype=np.uint8)
    n = 0 if caption_offsets is None else len(caption_offsets) - 1
    return hashlib.sha256(b.tobytes()).hexdigest(), int(b.size), int(n)


def check_shards_against_whole(shard_digests, whole_captions, whole_offsets, world):
    """The ranks' shards, concatenated in rank order, are the single-GPU output: shard r's digest and size equal those of
    bytes [offsets[begin_r], offsets[end_r]) of the whole run's captions (shard_range gives begin_r, end_r).  shard_digests:
    shard_digest() of every rank, in rank order.  Returns None, or a message naming the first difference."""
    import hashlib
    import numpy as np
    n = len(whole_offsets) - 1
    if len(shard_digests) != world:
        return "%d shard digests for %d ranks" % (len(shard_digests), world)
    whole = np.ascontiguousarray(whole_captions, dtype=np.uint8)
    for r, (digest, nbytes, nfiles) in enumerate(shard_digests):
        b, e = shard_range(n, r, world)
        lo, hi = int(whole_offsets[b]), int(whole_offsets[e])
        if nfiles != e - b:
            return "rank %d processed %d files, its shard has %d" % (r, nfiles, e - b)
        if nbytes != hi - lo:
            return "rank %d wrote %d caption bytes, the whole run has %d for its files" % (r, nbytes, hi - lo)
        if hashlib.sha256(whole[lo:hi].tobytes()).hexdigest() != digest:
            return "rank %d's captions differ from the whole run's captions of files %d..%d" % (r, b, e)
    if int(whole_offsets[n]) != sum(d[1] for d in shard_digests):
        return "the shards' bytes do not add up to the whole run's"
    return None


def free_port():
    """A TCP port on 127.0.0.1 that was free a moment ago (for MASTER_PORT)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(world, argv, env=None, timeout=None, grace=5.0):
    """Start `world` fresh child processes running `argv`, one per GPU, and wait for all of them.

    Every child gets RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1 and one common MASTER_PORT in its
    environment, the variables torch.distributed's env:// rendezvous reads.  The caller must not have touched the
    GPU: the children are started as ordinary child processes (never by replacing this one), and this process
    only waits.  Rank 0 inherits stdout, so the one line it prints is this command's output; the other ranks'
    stdout goes to stderr.  When a rank fails the rest get `grace` seconds to end by themselves (they may be about to
    report the same trouble) and are then terminated (the very processes started here, by pid).
    Returns the first non-zero exit status, or 0.
    """
    import os
    import subprocess
    import sys
    import time
    if world < 1:
        raise ValueError("world must be at least 1")
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(base.get("MASTER_PORT") or free_port()))
    procs = []
    for r in range(world):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(argv), env=e, stdout=None if r == 0 else sys.stderr))
    deadline = None if timeout is None else time.monotonic() + timeout
    status = 0
    failed_at = None
    live = set(range(world))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and status == 0:
                status = rc
                failed_at = time.monotonic()
        if live and ((status != 0 and time.monotonic() - failed_at > grace) or (deadline is not None and time.monotonic() > deadline)):
            if status == 0:
                status = 124
            for r in live:
                procs[r].terminate()
            for r in live:
                try:
                    procs[r].wait(10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            break
        if live:
            time.sleep(0.05)
    return status
