#!/usr/bin/env python3
"""Times the select stage (wave kernels) of one library build on a resident corpus; prints select_ms.  No result check:
meant for the diagnostic builds that stop the pipeline early.  usage: TAGPIPE_LIBRARY=... time_select.py [config] [files]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagecaptioner_amd import Config, TagPipe
from imagecaptioner_amd._abi import tp_device_io
from imagecaptioner_amd.corpus import make_corpus

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
corpus = make_corpus(cfgname, n, seed=1234)
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
sel = []
for k in range(6):
    try:
        pipe.run_lines_device(io)
    except Exception as e:      # a build that stops early leaves sizes the later steps may refuse
        pass
    sel.append(pipe.last_timing()["select_ms"])
print(os.environ.get("TAGPIPE_LIBRARY", "default"), cfgname, n, "select_ms min %.3f  all %s" % (min(sel[1:]), ["%.3f" % x for x in sel]), pipe.last_timing())
