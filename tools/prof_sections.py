#!/usr/bin/env python3
"""Per-section cycle shares of k_fast, from the diagnostic build (python -m imagecaptioner_amd.build --prof).

usage: TAGPIPE_LIBRARY=imagecaptioner_amd/libtagpipe_hip_prof.so python3 tools/prof_sections.py [config] [files]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagecaptioner_amd import Config, TagPipe
from imagecaptioner_amd._abi import tp_device_io
from imagecaptioner_amd.corpus import make_corpus
from imagecaptioner_amd.tagpipe import load_library

NAMES = ["S1 copy", "S2 scan", "S4 tokens+dedupe", "S5 candidates", "S6 buckets", "S7 coverage", "S8 evidence", "S9 closure", "S10 rules", "S11 facets", "S12 verdict"]


LEAN_NAMES = ["between files", "load+scan", "tokens", "dedupe", "select", "evidence", "rules+facets", "position+final", "verdict"]


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
    corpus = make_corpus(cfgname, n, seed=1234)
    min_tokens = int(os.environ.get("PROF_MIN_TOKENS", "0"))        # only the files with at least so many tokens (the long-file stage's share)
    if min_tokens:
        o = corpus.offsets.astype(np.int64)
        commas = np.add.reduceat((corpus.arena == ord(",")).astype(np.int64), o[:-1])
        keep = np.nonzero(commas + 1 >= min_tokens)[0]
        lens = (o[1:] - o[:-1])[keep]
        arena = np.concatenate([corpus.arena[o[i]:o[i + 1]] for i in keep])
        corpus.arena, corpus.offsets = arena, np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        n = len(keep)
        print("files with >=", min_tokens, "tokens:", n)
    pipe = TagPipe(Config(device=0, **corpus.config_kwargs))
    dev = torch.device("cuda", 0)
    arena = torch.cat([torch.from_numpy(corpus.arena).to(dev), torch.zeros(64, dtype=torch.uint8, device=dev)])
    offsets = torch.from_numpy(corpus.offsets.astype(np.int64)).to(dev)
    cap = max(4 * int(corpus.offsets[-1]), 1 << 20)
    captions = torch.empty(cap, dtype=torch.uint8, device=dev)
    cap_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    verdicts = torch.empty(n * 16, dtype=torch.uint8, device=dev)
    io = tp_device_io()
    io.arena, io.offsets, io.n_files = arena.data_ptr(), offsets.data_ptr(), n
    io.captions, io.caption_capacity, io.caption_offsets = captions.data_ptr(), cap, cap_off.data_ptr()
    io.verdicts, io.histogram, io.counters, io.fired, io.facet_choice, io.stream = verdicts.data_ptr(), None, None, None, None, None
    lib = load_library()
    buf = (C.c_ulonglong * 32)()
    pipe.run_lines_device(io)
    has_prof = hasattr(lib, "tp_debug_prof")
    if has_prof:
        lib.tp_debug_prof(buf, 1)
    pipe.run_lines_device(io)
    if has_prof:
        lib.tp_debug_prof(buf, 1)
    tot = float(sum(buf)) or 1.0
    print("k_fast ms", pipe.last_timing()["select_ms"], "files", n, "config", cfgname)
    print("no histogram:", pipe.last_timing())
    hist = torch.zeros(pipe.info.n_strings + 1, dtype=torch.int64, device=dev)
    io.histogram = hist.data_ptr()
    pipe.run_lines_device(io)
    print("with histogram:", pipe.last_timing())
    for k, name in enumerate(NAMES):
        print("%-20s %6.2f %%  %8.0f cycles/file" % (name, 100.0 * buf[k + 1] / tot, buf[k + 1] / n))
    for k, name in enumerate(LEAN_NAMES):
        print("lean %-15s %6.2f %%  %8.0f cycles/file" % (name, 100.0 * buf[16 + k] / tot, buf[16 + k] / n))
    for k, name in enumerate(["long scan", "long tokens", "long candidates+evidence", "long buckets"]):
        print("%-26s %8.0f cycles/file" % (name, buf[25 + k] / n))


if __name__ == "__main__":
    main()
