"""Device time of the triage entry point on a corpus config (host arena in, one byte per file out)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import Config, TagPipe  # noqa: E402
from imagecaptioner_amd.corpus import make_corpus  # noqa: E402

name, n = sys.argv[1], int(sys.argv[2])
c = make_corpus(name, n, seed=1)
p = TagPipe(Config(**c.config_kwargs))
r = p.run_lines(c.arena, c.offsets)
for what, arena, offs in (("input lines", c.arena, c.offsets), ("own captions", r.captions, r.caption_offsets)):
    for _ in range(3):
        t0 = time.time()
        bad = p.triage_lines(arena, offs)
        wall = time.time() - t0
        tm = p.last_timing()
    print("%s %s: bad %d of %d, unsupported %d; device %.3f ms (wave stages %.3f, general %.3f on %d files), wall %.1f ms" % (
        name, what, int((bad == 1).sum()), n, int((bad == 2).sum()), tm["total_ms"], tm["select_ms"], tm["general_ms"], tm["general_files"], wall * 1e3))
