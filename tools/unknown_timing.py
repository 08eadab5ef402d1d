"""k_fast time on a c2 corpus whose lines carry extra tags that are in no table (strict vocabulary drops them)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import Config, TagPipe  # noqa: E402
from imagecaptioner_amd.corpus import make_corpus  # noqa: E402

n, extra = int(sys.argv[1]), int(sys.argv[2])
c = make_corpus("c2", n, seed=1)
a = bytes(c.arena)
rng = np.random.default_rng(5)
lines = []
for i in range(n):
    line = a[int(c.offsets[i]):int(c.offsets[i + 1])].rstrip(b"\n")
    for k in rng.integers(0, 1 << 30, size=extra):
        line += b", tagger_made_this_up_%d" % k
    lines.append(line)
offs = np.zeros(n + 1, dtype=np.uint64)
offs[1:] = np.cumsum([len(l) for l in lines])
arena = np.frombuffer(b"".join(lines), dtype=np.uint8).copy()
p = TagPipe(Config(**c.config_kwargs))
for _ in range(3):
    r = p.run_lines(arena, offs)
    tm = p.last_timing()
print("files", n, "extra unknown tags per file", extra, "wave stages ms", round(tm["select_ms"], 3), "general files", tm["general_files"], "mean tags", float(r.verdicts["n_tags"].mean()))
