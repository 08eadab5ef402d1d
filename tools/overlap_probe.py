"""Does running the pipeline of two file ranges at once (two contexts, two HIP streams, two host threads) finish sooner than
one range after the other?  If it does, the caption-emit kernels of one range can hide behind the wave kernel of the next."""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import Config, TagPipe  # noqa: E402
from imagecaptioner_amd._abi import tp_device_io  # noqa: E402
from imagecaptioner_amd.corpus import make_corpus  # noqa: E402

n, parts = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
c = make_corpus("c2", n, seed=1234)


def make(lo, hi, stream):
    pipe = TagPipe(Config(**c.config_kwargs))
    o0, o1 = int(c.offsets[lo]), int(c.offsets[hi])
    arena = torch.cat([torch.from_numpy(c.arena[o0:o1]).to(dev), torch.zeros(64, dtype=torch.uint8, device=dev)])
    offsets = torch.from_numpy((c.offsets[lo:hi + 1] - c.offsets[lo]).astype(np.int64)).to(dev)
    m = hi - lo
    cap = max(4 * (o1 - o0), 1 << 20)
    bufs = dict(arena=arena, offsets=offsets, captions=torch.empty(cap, dtype=torch.uint8, device=dev), co=torch.empty(m + 1, dtype=torch.int64, device=dev),
                ver=torch.empty(m * 16, dtype=torch.uint8, device=dev), hist=torch.zeros(pipe.info.n_strings + 1, dtype=torch.int64, device=dev),
                ctr=torch.zeros(8, dtype=torch.int64, device=dev))
    io = tp_device_io()
    io.arena, io.offsets, io.n_files = arena.data_ptr(), offsets.data_ptr(), m
    io.captions, io.caption_capacity, io.caption_offsets = bufs["captions"].data_ptr(), cap, bufs["co"].data_ptr()
    io.verdicts, io.histogram, io.counters = bufs["ver"].data_ptr(), bufs["hist"].data_ptr(), bufs["ctr"].data_ptr()
    io.stream = stream.cuda_stream
    return pipe, io, bufs


def timed(work, reps=10):
    for _ in range(2):
        work()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        work()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


whole = make(0, n, torch.cuda.Stream())
t_whole = timed(lambda: whole[0].run_lines_device(whole[1]))
edges = [n * k // parts for k in range(parts + 1)]
ctxs = [make(edges[k], edges[k + 1], torch.cuda.Stream()) for k in range(parts)]


def serial():
    for p, io, _ in ctxs:
        p.run_lines_device(io)


def concurrent():
    th = [threading.Thread(target=lambda p=p, io=io: p.run_lines_device(io)) for p, io, _ in ctxs]
    for t in th:
        t.start()
    for t in th:
        t.join()


t_serial, t_conc = timed(serial), timed(concurrent)
print("c2 %d files: one call %.2f ms (%.1f M files/s); %d ranges one after the other %.2f ms; the same ranges at once on %d streams %.2f ms (%.1f M files/s)" % (
    n, t_whole, n / t_whole / 1e3, parts, t_serial, parts, t_conc, n / t_conc / 1e3))
