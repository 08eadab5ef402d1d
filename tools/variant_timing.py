"""Wave-stage time on variations of a c2 corpus that real tag files show: tags in no table, blanks for underscores,
capital letters, CRLF line ends, a trailing comma."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import Config, TagPipe  # noqa: E402
from imagecaptioner_amd.corpus import make_corpus  # noqa: E402

n = int(sys.argv[1])
modes = sys.argv[2:] or ["plain", "unknown3", "blanks", "capitals", "crlf", "trailing_comma", "cr_inside_2pc"]
c = make_corpus("c2", n, seed=1)
a = bytes(c.arena)
base = [a[int(c.offsets[i]):int(c.offsets[i + 1])].rstrip(b"\n") for i in range(n)]
p = TagPipe(Config(**c.config_kwargs))
rng = np.random.default_rng(5)
for mode in modes:
    if mode == "plain":
        lines = base
    elif mode == "unknown3":
        lines = [l + b"".join(b", tagger_made_this_up_%d" % k for k in rng.integers(0, 1 << 30, size=3)) for l in base]
    elif mode == "blanks":
        lines = [l.replace(b"_", b" ") for l in base]
    elif mode == "capitals":
        lines = [l.title() for l in base]
    elif mode == "crlf":
        lines = [l + b"\r\n" for l in base]
    elif mode == "cr_inside_2pc":                          # every 50th file is broken over lines the Windows way: general kernel
        lines = [l.replace(b", ", b",\r\n", 3) if i % 50 == 0 else l for i, l in enumerate(base)]
    elif mode == "lone_cr_2pc":                            # every 50th file holds a carriage return of its own: the general kernel takes it
        lines = [l.replace(b", ", b",\r ", 1) if i % 50 == 0 else l for i, l in enumerate(base)]
    elif mode == "trailing_comma":
        lines = [l + b"," for l in base]
    else:
        raise SystemExit("unknown mode " + mode)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(l) for l in lines])
    arena = np.frombuffer(b"".join(lines), dtype=np.uint8).copy()
    for _ in range(3):
        r = p.run_lines(arena, offs)
        tm = p.last_timing()
    print("%-15s files %d: wave stages %.3f ms, general kernel %.3f ms on %d files, emit %.3f ms; mean tags %.1f" % (
        mode, n, tm["select_ms"], tm["general_ms"], tm["general_files"], tm["emit_ms"], float(r.verdicts["n_tags"].mean())), flush=True)
