"""Many more fuzzed inputs than the test suite holds, through the library and the CPU oracle, on every table configuration of the
tests: fuzzed tag lines (plain run, and a part of them in audit mode), fuzzed tagger replies (one per file, two to merge, caps 15 / 2).
usage: fuzz_at_scale.py [lines_per_seed] [seeds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import EDGE_LINES, Oracle, assert_same_results, pack_lines  # noqa: E402
from imagecaptioner_amd import TagPipe  # noqa: E402
from test_gpu_parity import _check_audit, _check_review, _review_args  # noqa: E402
from test_merged_cpu import merged_inputs  # noqa: E402
from test_pipeline_logic_cpu import CONFIGS, fuzz_lines  # noqa: E402

per_seed = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for name in sorted(CONFIGS):
    cfg = CONFIGS[name]()
    p, o = TagPipe(cfg), Oracle(cfg)
    vf = None if name == "novocab_w07_crlf" else "top_tags.txt"
    nl = b"\r\n" if name == "novocab_w07_crlf" else b"\n"
    for seed in range(1000, 1000 + seeds):
        t0 = time.time()
        lines = EDGE_LINES + fuzz_lines(seed, per_seed)
        arena, offsets = pack_lines(lines)
        got, want = p.run_lines(arena, offsets), o.run_lines(arena, offsets, threads=16)
        assert_same_results(got, want, p.histogram_dict(got.histogram), "%s seed %d" % (name, seed))
        general = p.last_timing()["general_files"]
        k = min(len(lines), 8000)
        a2, o2 = pack_lines(lines[:k])
        audit = _review_args(k, vocab_file=vf)
        got, want = p.run_lines(a2, o2, audit=audit), o.run_lines(a2, o2, threads=16, detail=True, audit=audit)
        _check_audit(got, want, nl)
        _check_review(got, want, nl)
        ar, toffs, begin, mid, scores = merged_inputs(seed, per_seed // 8)
        m = len(begin) - 1
        audit = _review_args(m, vocab_file=vf)
        for cap in (15, 2):
            got = p.run_merged(ar, toffs, begin, mid, scores, cap, audit=audit)
            want = o.run_merged(ar, toffs, begin, mid, scores, cap, threads=16, detail=True, audit=audit)
            assert_same_results(got, want, p.histogram_dict(got.histogram), "%s merged seed %d cap %d" % (name, seed, cap))
            _check_audit(got, want, nl)
            _check_review(got, want, nl)
        src = (np.arange(len(scores)) % 3).astype(np.uint8)
        got = p.run_scored(ar, toffs, begin, scores, src, audit=audit)
        want = o.run_scored(ar, toffs, begin, scores, src, threads=16, detail=True, audit=audit)
        assert_same_results(got, want, p.histogram_dict(got.histogram), "%s scored seed %d" % (name, seed))
        _check_audit(got, want, nl)
        _check_review(got, want, nl)
        got, want = p.run_scored(ar, toffs, begin, scores, src), o.run_scored(ar, toffs, begin, scores, src, threads=16)
        assert_same_results(got, want, p.histogram_dict(got.histogram), "%s scored plain seed %d" % (name, seed))
        print("%s seed %d: %d lines (%d in the general kernel), %d of them in audit mode, %d files of tagger replies x 4 runs: equal to the oracle (%.0f s)" % (
            name, seed, len(lines), general, k, m, time.time() - t0), flush=True)
