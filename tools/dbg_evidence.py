import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from imagecaptioner_amd import Config, TagPipe
from imagecaptioner_amd.corpus import make_corpus
from oracle.oracle_py import Oracle
c = make_corpus("c2", 6000, seed=4242)
cfg = Config(**c.config_kwargs)
p = TagPipe(cfg)
got = p.run_lines(c.arena, c.offsets)
print(p.last_timing())
want = Oracle(cfg).run_lines(c.arena, c.offsets, threads=16)
d = np.nonzero(got.verdicts["n_evidence"] != want.verdicts["n_evidence"])[0]
print("files differing in n_evidence:", len(d), d[:20])
for i in d[:5]:
    print(i, got.verdicts[i], want.verdicts[i])
for name in got.verdicts.dtype.names:
    print(name, int((got.verdicts[name] != want.verdicts[name]).sum()))
