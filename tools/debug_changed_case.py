"""Which line of the crafted changed-action-tag case takes the general kernel after another configuration ran in the process."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import changed_action_case  # noqa: E402
from test_audit_cpu import review_args  # noqa: E402
from imagecaptioner_amd import Config, TagPipe  # noqa: E402
from imagecaptioner_amd.corpus import make_corpus  # noqa: E402
def au(n):
    a = review_args(n); a["used_profile"] = "sgrui"; return a
c = make_corpus("c2", 6000, seed=77)
p0 = TagPipe(Config(**c.config_kwargs))
p0.run_lines(c.arena, c.offsets, audit=au(6000)); print("c2 audit:", p0.last_timing()["general_files"])
p0.run_lines(c.arena, c.offsets); print("c2 plain:", p0.last_timing()["general_files"])
del p0
import gc; gc.collect()
cfg, lines = changed_action_case()
p = TagPipe(cfg)
p.run_texts(lines, audit=au(len(lines))); print("crafted, audit all:", p.last_timing())
for i, l in enumerate(lines):
    p.run_texts([l], audit=au(1)); print(i, p.last_timing()["general_files"], l)
