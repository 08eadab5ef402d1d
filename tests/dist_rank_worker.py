"""One rank of the two-rank CPU rehearsal started by imagecaptioner_amd.dist.spawn_ranks (see test_dist_gloo.py).

Reads RANK / WORLD_SIZE / MASTER_* from the environment like bench.py, joins a gloo group, runs the oracle on its
contiguous shard of a seeded corpus and sums histogram and counters over the ranks.  argv: n_files out_dir [failing_rank
[strong]].  With "strong" the rank builds only its own range of the ONE corpus (make_corpus(..., file_range=...)), as
`bench.py --scaling strong` does, the ranks' caption digests are gathered, and rank 0 checks them against its own run
over the whole corpus with the function bench.py uses.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from helpers import Config, Oracle  # noqa: E402
from imagecaptioner_amd.corpus import make_corpus  # noqa: E402
from imagecaptioner_amd.dist import allreduce_sum_, check_shards_against_whole, shard_digest, shard_offsets, shard_range  # noqa: E402


def main():
    n_files, out_dir = int(sys.argv[1]), sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
    dist.init_process_group("gloo")          # env:// rendezvous from the variables the launcher set
    strong = len(sys.argv) > 4 and sys.argv[4] == "strong"
    c = make_corpus("c1", n_files, seed=11)
    o = Oracle(Config(**c.config_kwargs))
    b, e = shard_range(n_files, rank, world)
    if strong:
        part = make_corpus("c1", n_files, seed=11, file_range=(b, e))
        r = o.run_lines(part.arena, part.offsets)
    else:
        offs, first = shard_offsets(c.offsets, b, e)
        r = o.run_lines(c.arena[first:first + int(offs[-1])], offs)
    whole = o.run_lines(c.arena, c.offsets)
    vocab = sorted(whole.histogram)
    hist = torch.tensor([r.histogram.get(t, 0) for t in vocab], dtype=torch.int64)
    ctr = torch.tensor([r.counters[k] for k in sorted(r.counters)], dtype=torch.int64)
    joined = torch.zeros(world, dtype=torch.int64)
    joined[rank] = 1 + rank
    allreduce_sum_([hist, ctr, joined])
    np.save(os.path.join(out_dir, "cap%d.npy" % rank), r.captions)
    shard_check = "not run"
    if strong:
        digests = [None] * world
        dist.all_gather_object(digests, shard_digest(r.captions, r.caption_offsets))
        if rank == 0:
            shard_check = check_shards_against_whole(digests, whole.captions, whole.caption_offsets, world)
            broken = [(d[0][::-1], d[1], d[2]) for d in digests]          # a wrong digest must be noticed
            assert check_shards_against_whole(broken, whole.captions, whole.caption_offsets, world) is not None
    if rank == 0:
        print(json.dumps({"world": dist.get_world_size(), "joined": joined.tolist(), "hist": hist.tolist(), "ctr": ctr.tolist(), "shard_check": shard_check}))
    dist.barrier()
    dist.destroy_process_group()
    if len(sys.argv) > 3 and int(sys.argv[3]) == rank:
        sys.exit(7)                          # the launcher must report a failing rank


if __name__ == "__main__":
    main()
