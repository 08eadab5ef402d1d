"""Device time of the scored entry points (tp_run_scored, tp_run_merged) on the tags of a corpus config with made-up scores."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import Config, TagPipe  # noqa: E402
from imagecaptioner_amd.corpus import make_corpus  # noqa: E402

name, n = sys.argv[1], int(sys.argv[2])
c = make_corpus(name, n, seed=1)
a = bytes(c.arena)
rng = np.random.default_rng(3)
tags, begin, mid = [], [0], []
for i in range(n):
    toks = a[int(c.offsets[i]):int(c.offsets[i + 1])].rstrip(b"\n").split(b", ")
    k = max(1, (2 * len(toks)) // 3)                       # the primary reply holds two thirds of the tags, the secondary one all of them
    tags += toks[:k]
    mid.append(len(tags))
    tags += toks
    begin.append(len(tags))
offs = np.zeros(len(tags) + 1, dtype=np.uint64)
offs[1:] = np.cumsum([len(t) for t in tags])
arena = np.frombuffer(b"".join(tags), dtype=np.uint8).copy()
scores = np.round(rng.random(len(tags)) * 0.65 + 0.35, 3)
begin, mid = np.array(begin, dtype=np.uint64), np.array(mid, dtype=np.uint64)
p = TagPipe(Config(**c.config_kwargs))
# one reply per file: the secondary replies alone (every tag once), each in the order ParseJoyTags hands a reply with scores on
# (score descending, then as the tags came: Program.cs:487)
sec_idx = np.concatenate([np.arange(int(mid[i]), int(begin[i + 1])) for i in range(n)])
file_of = np.repeat(np.arange(n), (begin[1:] - mid).astype(np.int64))
sec_idx = sec_idx[np.lexsort((sec_idx, -scores[sec_idx], file_of))]
one_tags = [tags[k] for k in sec_idx]
one_offs = np.zeros(len(one_tags) + 1, dtype=np.uint64)
one_offs[1:] = np.cumsum([len(t) for t in one_tags])
one_arena = np.frombuffer(b"".join(one_tags), dtype=np.uint8).copy()
one_begin = np.zeros(n + 1, dtype=np.uint64)
one_begin[1:] = np.cumsum(begin[1:] - mid)
for _ in range(2):
    r = p.run_scored(one_arena, one_offs, one_begin, scores[sec_idx], np.zeros(len(one_tags), np.uint8))
    tm = p.last_timing()
print("%s scored: %d files, %.1f tags/file in: device %.2f ms (wave kernel %.2f ms, general kernel %.2f ms on %d files, emit %.2f ms) = %.2f M files/s; mean caption tags %.1f" % (
    name, n, len(one_tags) / n, tm["total_ms"], tm["select_ms"], tm["general_ms"], tm["general_files"], tm["emit_ms"], n / tm["total_ms"] / 1e3,
    float(r.verdicts["n_tags"].mean())))
for _ in range(2):
    r = p.run_merged(arena, offs, begin, mid, scores, 15)
    tm = p.last_timing()
print("%s merged: %d files, %.1f tags/file in: device %.2f ms (wave kernel %.2f ms, general kernel %.2f ms on %d files, emit %.2f ms) = %.2f M files/s; mean caption tags %.1f" % (
    name, n, len(tags) / n, tm["total_ms"], tm["select_ms"], tm["general_ms"], tm["general_files"], tm["emit_ms"], n / tm["total_ms"] / 1e3,
    float(r.verdicts["n_tags"].mean())))
# the same two calls in audit mode (the reference appends an audit row per image whatever the options: Program.cs:334)
audit = {"names": ["%d.png" % i for i in range(n)], "vocab_file": "top_tags.txt", "used_profile": "none"}
for what, call in (("scored", lambda: p.run_scored(one_arena, one_offs, one_begin, scores[sec_idx], np.zeros(len(one_tags), np.uint8), audit=audit)),
                   ("merged", lambda: p.run_merged(arena, offs, begin, mid, scores, 15, audit=audit))):
    for _ in range(2):
        call()
        tm = p.last_timing()
    print("%s %s, audit mode: device %.2f ms (wave kernel %.2f ms, general kernel %.2f ms on %d files, emit %.2f ms) = %.2f M files/s" % (
        name, what, tm["total_ms"], tm["select_ms"], tm["general_ms"], tm["general_files"], tm["emit_ms"], n / tm["total_ms"] / 1e3))
