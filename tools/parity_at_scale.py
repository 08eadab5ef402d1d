"""Parity at the benchmark's sizes: every corpus configuration at `files` files through the library (host-pointer entry point, so
the run in pieces and every stage is exercised), compared with the CPU oracle on the first `prefix` files and on `prefix` files
from the middle (captions, offsets, verdicts, fired rules, facet choices byte for byte; the histogram and the counters of a
run over exactly those files), and by sums over the whole run (counters add up to the file count, histogram total = tag total).
usage: parity_at_scale.py [files] [prefix]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import Oracle, assert_same_results  # noqa: E402
from imagecaptioner_amd import Config, TagPipe  # noqa: E402
from imagecaptioner_amd.corpus import make_corpus  # noqa: E402

files = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
prefix = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
for name in ("c1", "c2", "c2n", "c3", "c3c", "c4", "c5"):
    t0 = time.time()
    c = make_corpus(name, files, seed=2026)
    cfg = Config(**c.config_kwargs)
    p, o = TagPipe(cfg), Oracle(cfg)
    got = p.run_lines(c.arena, c.offsets)
    n = len(c.offsets) - 1
    assert got.counters["files"] == n and got.counters["unsupported"] == 0, (name, got.counters)
    assert int(got.histogram.sum()) + got.histogram_other == int(got.verdicts["n_tags"].astype(np.int64).sum()), name   # (other: tags in no table)
    assert np.array_equal(np.diff(got.caption_offsets.astype(np.int64)) > 0, np.ones(n, bool)), name
    for lo in (0, (n - prefix) // 2):
        hi = lo + prefix
        b0, b1 = int(c.offsets[lo]), int(c.offsets[hi])
        arena, offsets = c.arena[b0:b1], c.offsets[lo:hi + 1] - c.offsets[lo]
        want = o.run_lines(arena, offsets, threads=16)
        part = p.run_lines(arena, offsets)                                        # the same files on their own: histogram and counters of exactly these
        assert_same_results(part, want, p.histogram_dict(part.histogram), "%s files %d..%d" % (name, lo, hi))
        c0, c1 = int(got.caption_offsets[lo]), int(got.caption_offsets[hi])
        assert bytes(got.captions[c0:c1]) == bytes(want.captions), (name, lo)
        assert np.array_equal(got.caption_offsets[lo:hi + 1] - got.caption_offsets[lo], want.caption_offsets), (name, lo)
        assert got.verdicts[lo:hi].tobytes() == want.verdicts.tobytes(), (name, lo)
    print("%s: %d files, %.1f tags/caption, general kernel %d files; equal to the oracle on files 0..%d and %d..%d (%.0f s)" % (
        name, n, float(got.verdicts["n_tags"].mean()), p.last_timing()["general_files"], prefix, (n - prefix) // 2, (n - prefix) // 2 + prefix, time.time() - t0), flush=True)

# Audit mode (the audit row's lists and the review records, Program.cs:334-341) on `audit_files` files of three configurations,
# and tagger replies made of the c2 files' tags with scores of their own (one reply per file, and two to merge)
from test_gpu_parity import _check_audit, _check_review, _review_args  # noqa: E402

audit_files = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000
for name in ("c2", "c3c", "c5"):
    t0 = time.time()
    c = make_corpus(name, audit_files, seed=2027)
    cfg = Config(**c.config_kwargs)
    p, o = TagPipe(cfg), Oracle(cfg)
    audit = _review_args(audit_files)
    got, want = p.run_lines(c.arena, c.offsets, audit=audit), o.run_lines(c.arena, c.offsets, threads=16, detail=True, audit=audit)
    _check_audit(got, want)
    _check_review(got, want)
    assert_same_results(got, want, p.histogram_dict(got.histogram), name + " audit")
    print("%s in audit mode: %d files, general kernel %d files; audit rows and review records equal to the oracle's (%.0f s)" % (
        name, audit_files, p.last_timing()["general_files"], time.time() - t0), flush=True)
    if name != "c2":
        continue
    raw = bytes(c.arena)
    rng = np.random.default_rng(5)
    tags, begin, mid = [], [0], []
    for i in range(audit_files):
        toks = raw[int(c.offsets[i]):int(c.offsets[i + 1])].rstrip(b"\n").split(b", ")
        tags += toks[:max(1, 2 * len(toks) // 3)]
        mid.append(len(tags))
        tags += toks
        begin.append(len(tags))
    toffs = np.zeros(len(tags) + 1, dtype=np.uint64)
    toffs[1:] = np.cumsum([len(t) for t in tags])
    arena = np.frombuffer(b"".join(tags), dtype=np.uint8).copy()
    scores = np.round(rng.random(len(tags)) * 0.8 + 0.2, 3)              # three decimals: equal scores within and between the replies
    begin, mid = np.array(begin, dtype=np.uint64), np.array(mid, dtype=np.uint64)
    for what in ("one reply per file", "two replies per file"):
        t0 = time.time()
        if what.startswith("one"):
            src = np.zeros(len(tags), np.uint8)
            got = p.run_scored(arena, toffs, begin, scores, src, audit=audit)
            want = o.run_scored(arena, toffs, begin, scores, src, threads=16, detail=True, audit=audit)
        else:
            got = p.run_merged(arena, toffs, begin, mid, scores, 15, audit=audit)
            want = o.run_merged(arena, toffs, begin, mid, scores, 15, threads=16, detail=True, audit=audit)
        _check_audit(got, want)
        _check_review(got, want)
        assert_same_results(got, want, p.histogram_dict(got.histogram), "c2 replies, " + what)
        print("c2 tagger replies in audit mode, %s (%.1f tags a file, tags twice and equal scores among them): general kernel %d of %d files; rows, records and captions equal to the oracle's (%.0f s)" % (
            what, len(tags) / audit_files, p.last_timing()["general_files"], audit_files, time.time() - t0), flush=True)
