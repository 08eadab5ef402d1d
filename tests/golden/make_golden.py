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
CORPORA = {"c1": 1000, "c2": 1000, "c3": 500, "c3c": 500, "c4": 500, "c5": 300}


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
    with open(os.path.join(HERE, "replies.json"), "w", encoding="utf-8") as f:
        json.dump(replies_golden(), f, ensure_ascii=False, indent=0)
    write_audit_golden()


def audit_args(n):
    """File names, paths and option texts the audit and review lines are made with."""
    return {"names": ["img_%03d.png" % i for i in range(n)], "vocab_file": "top_tags.txt", "used_profile": "sgrui",
            "image_paths": ["D:\\set \"x\"\\img_%03d.png" % i for i in range(n)], "caption_paths": ["D:\\set \"x\"\\img_%03d.txt" % i for i in range(n)],
            "thresholds": "primary=0.40;secondary=0.30"}


def audit_inputs():
    """The first 40 files of replies.json, both replies."""
    import numpy as np
    with open(os.path.join(HERE, "replies.json"), encoding="utf-8") as f:
        files = json.load(f)["files"][:40]
    a, offs, begin, scores = pack_tags([[tuple(x) for x in f["primary"] + f["secondary"]] for f in files])
    mid = begin[:-1] + np.array([len(f["primary"]) for f in files], dtype=np.uint64)
    return a, offs, begin, mid, scores


def write_audit_golden():
    """audit.csv and review_candidates.jsonl/.csv lines (Program.cs:871-890, 958-969) of the merged replies (cap 15), without newlines."""
    a, offs, begin, mid, scores = audit_inputs()
    out = {}
    for name in ("shipped", "shipped_nonstrict"):
        r = Oracle(CONFIGS[name]()).run_merged(a, offs, begin, mid, scores, 15, audit=audit_args(len(begin) - 1))
        out[name] = {"audit": [x.decode("utf-8") for x in r.audit_rows], "review_jsonl": [x.decode("utf-8") for x in r.review_jsonl],
                     "review_csv": [x.decode("utf-8") for x in r.review_csv]}
    with open(os.path.join(HERE, "audit_rows.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)


def pack_tags(files):
    """files: per file a list of (tag, score) -> (arena, tag_offsets, tag_begin, scores)"""
    import numpy as np
    tags = [t for fl in files for t, _ in fl]
    enc = [t.encode("utf-8") for t in tags]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offs[1:] = np.cumsum([len(b) for b in enc])
    arena = np.frombuffer(b"".join(enc), dtype=np.uint8).copy() if enc else np.zeros(0, np.uint8)
    begin = np.zeros(len(files) + 1, dtype=np.uint64)
    begin[1:] = np.cumsum([len(fl) for fl in files])
    return arena, offs, begin, np.array([sc for fl in files for _, sc in fl], dtype=np.float64)


def replies_golden():
    """Tagger replies (primary and secondary per file) and existing captions, with what the oracle makes of them: one reply
    (tp_run_scored on the primary reply), both merged with caps 15 and 2 (tp_run_merged), triage of the texts (tp_triage_lines)."""
    import random
    import numpy as np
    rng = random.Random(20)
    vocab = [l.strip() for l in open(os.path.join(HERE, "reference_data", "top_tags.txt"), encoding="utf-8") if l.strip()][:1200]
    odd = [";)", "zz_unknown", "best_quality_zz", "some thing new", "Long Hair", "'quoted'", "trailing.", ""]
    files = []
    for i in range(80):
        k = rng.choice([0, 1, 3, 8, 20, 45])
        prim = [(t, round(0.4 + 0.6 * rng.random(), 3)) for t in rng.sample(vocab, k)]
        if i % 4 == 0:
            prim += [(rng.choice(odd), round(rng.random(), 2)) for _ in range(2)]
        if i % 5 == 0 and prim:
            prim.append((prim[0][0].replace("_", " "), round(prim[0][1] - 0.02, 3)))
        sec = [(t, round(sc + rng.choice([0.0, 0.01, -0.03, 0.05]), 3)) for t, sc in prim if rng.random() < 0.8]
        sec += [(t, round(0.3 + 0.1 * rng.random(), 3)) for t in rng.sample(vocab, rng.choice([0, 2, 6, 18]))]
        files.append({"primary": prim, "secondary": sec})
    texts = ["", "1girl, smile, dress", "sorry, I can't help with that", "1girl", "looking_at_viewer, very_long_hair_1234567890123456",
             "a, b\nc, d", "tag one,\r\ntag two, tag three", "POSITIVE: 1girl, smile, hat\nNEGATIVE: blurry", "```\n1girl, smile, dress\n```"]
    texts += [", ".join(t for t, _ in f["primary"]) for f in files[:20]]
    out = {"files": files, "texts": texts, "expected": {}}
    for name in ("shipped", "builtin"):
        o = Oracle(CONFIGS[name]())
        exp = {}
        a, offs, begin, scores = pack_tags([f["primary"] for f in files])
        r = o.run_scored(a, offs, begin, scores, np.zeros(len(scores), np.uint8))
        exp["scored"] = [{"caption": r.caption(i).decode("utf-8"), "verdict": [int(x) for x in r.verdicts[i]]} for i in range(len(files))]
        a, offs, begin, scores = pack_tags([f["primary"] + f["secondary"] for f in files])
        mid = begin[:-1] + np.array([len(f["primary"]) for f in files], dtype=np.uint64)
        for cap in (15, 2):
            r = o.run_merged(a, offs, begin, mid, scores, cap)
            exp["merged_cap%d" % cap] = [{"caption": r.caption(i).decode("utf-8"), "verdict": [int(x) for x in r.verdicts[i]]} for i in range(len(files))]
        exp["triage"] = [int(x) for x in o.triage_texts(texts)]
        out["expected"][name] = exp
    return out


if __name__ == "__main__":
    if "--audit-only" in sys.argv:
        write_audit_golden()
    else:
        main()
