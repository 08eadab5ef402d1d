"""Device time of a run in audit mode (tp_set_audit: the audit row's tag lists kept by the lean stage, or by the thread-per-file kernel
for the files it declines, and emitted as text) beside the same run without it, and the host time of tp_format_audit_rows."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import Config, TagPipe  # noqa: E402
from imagecaptioner_amd.corpus import make_corpus  # noqa: E402

name, n = sys.argv[1], int(sys.argv[2])
c = make_corpus(name, n, seed=1)
p = TagPipe(Config(**c.config_kwargs))
audit = {"names": ["%08d.png" % i for i in range(n)], "vocab_file": "top_tags.txt", "used_profile": "sgrui"}
for _ in range(2):
    p.run_lines(c.arena, c.offsets)
    plain = p.last_timing()
for _ in range(2):
    t0 = time.time()
    r = p.run_lines(c.arena, c.offsets, audit=audit)
    wall = time.time() - t0
    tm = p.last_timing()
rows = sum(len(x) for x in r.audit_rows)
print("%s %d files: plain run device %.2f ms (%.1f M files/s); audit run device %.2f ms (general kernel %.2f ms, emit %.2f ms, other incl. audit text %.2f ms) "
      "= %.2f M files/s; audit.csv %.1f MB (%.0f B/row); whole audit call incl. copies and host formatting %.2f s" % (
          name, n, plain["total_ms"], n / plain["total_ms"] / 1e3, tm["total_ms"], tm["general_ms"], tm["emit_ms"], tm["other_ms"],
          n / tm["total_ms"] / 1e3, rows / 1e6, rows / n, wall))
