#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from the CPU oracle.

The reference itself (C#, .NET 8) cannot run in this image, so these are restatement-generated goldens: they
freeze the oracle's outputs (which tests/test_oracle_selfcheck.py pins to the reference's own self-check and to
hand-derived known answers) so that a later change to the oracle or to the HIP path shows up as a diff.
Inputs and expected outputs only; no reference source text is stored here.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from helpers import EDGE_LINES, Config, Oracle, shipped_config  # noqa: E402
from imagecaptioner_amd.corpus import make_corpus  # noqa: E402

CONFIGS = {
    "shipped": lambda: shipped_config(),
    "shipped_nonstrict": lambda: shipped_config(strict_known=False, strict_known_explicit=True),
    "builtin": lambda: Config(trigger="trigger_token", minimal_subject_tags="1girl", no_profile=True),
}
CORPORA = {"c1": 1000, "c2": 1000, "c3": 500, "c4": 500, "c5": 300}


def main():
    edge = {"lines": EDGE_LINES, "expected": {}}
    for name, mk in CONFIGS.items():
        r = Oracle(mk()).run_texts(EDGE_LINES)
        edge["expected"][name] = [{"caption": r.caption(i).decode("utf-8"), "verdict": [int(x) for x in r.verdicts[i]]} for i in range(len(EDGE_LINES))]
    with open(os.path.join(HERE, "edge_lines.json"), "w", encoding="utf-8") as f:
        json.dump(edge, f, ensure_ascii=False, indent=0)
    digests = {}
    for name, n in CORPORA.items():
        c = make_corpus(name, n)
        r = Oracle(Config(**c.config_kwargs)).run_lines(c.arena, c.offsets, threads=8)
        top = sorted(r.histogram.items(), key=lambda kv: (-kv[1], kv[0]))[:20]
        digests[name] = {"n_files": n, "input_sha256": hashlib.sha256(c.arena.tobytes()).hexdigest(),
                         "captions_sha256": hashlib.sha256(r.captions.tobytes()).hexdigest(),
                         "verdicts_sha256": hashlib.sha256(r.verdicts.tobytes()).hexdigest(),
                         "counters": r.counters, "top_tags": top, "first_caption": r.caption(0).decode("utf-8")}
    with open(os.path.join(HERE, "corpus_digests.json"), "w", encoding="utf-8") as f:
        json.dump(digests, f, indent=1)


if __name__ == "__main__":
    main()
