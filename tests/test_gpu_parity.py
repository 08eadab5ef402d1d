"""Parity tests proper: the HIP path, called through the C ABI, against the oracle — bit-exact on caption bytes,
caption offsets, verdict records, fired-rule bitmaps, facet choices, counters and the tag histogram."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import EDGE_LINES, Config, Oracle, assert_same_results, check_non_ascii, non_ascii_config, pack_lines, shipped_config
from test_pipeline_logic_cpu import CONFIGS, fuzz_lines, scored_inputs

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pipe_for(cfg):
    from imagecaptioner_amd import TagPipe
    return TagPipe(cfg)


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_edge_and_fuzz_lines(name):
    cfg = CONFIGS[name]()
    lines = EDGE_LINES + fuzz_lines(11, 3000)
    p = pipe_for(cfg)
    got, want = p.run_texts(lines), Oracle(cfg).run_texts(lines)
    assert got.counters["unsupported"] == 0
    assert_same_results(got, want, p.histogram_dict(got.histogram), name)


def test_edge_lines_match_committed_golden():
    with open(os.path.join(GOLDEN, "edge_lines.json"), encoding="utf-8") as f:
        g = json.load(f)
    from golden.make_golden import CONFIGS as GC
    for name in sorted(GC):
        got = pipe_for(GC[name]()).run_texts(g["lines"])
        for i, exp in enumerate(g["expected"][name]):
            assert got.caption(i).decode("utf-8") == exp["caption"], (name, i)
            assert [int(x) for x in got.verdicts[i]] == exp["verdict"], (name, i)


@pytest.mark.parametrize("name", ["shipped", "builtin"])
def test_replies_match_committed_golden(name):
    from golden.make_golden import CONFIGS as GC
    from test_golden import check_replies_golden
    with open(os.path.join(GOLDEN, "replies.json"), encoding="utf-8") as f:
        g = json.load(f)
    p = pipe_for(GC[name]())
    check_replies_golden(p.run_scored, p.run_merged, p.triage_texts, name, g)


@pytest.mark.parametrize("name,n", [("c1", 4000), ("c2", 6000), ("c3", 3000), ("c3c", 4000), ("c4", 3000), ("c5", 1500)])
def test_benchmark_corpora(name, n):
    from imagecaptioner_amd.corpus import make_corpus
    c = make_corpus(name, n, seed=4242)
    cfg = Config(**c.config_kwargs)
    p = pipe_for(cfg)
    got, want = p.run_lines(c.arena, c.offsets), Oracle(cfg).run_lines(c.arena, c.offsets, threads=16)
    assert got.counters["unsupported"] == 0
    assert_same_results(got, want, p.histogram_dict(got.histogram), name)
    if name == "c3c":
        # the alias chains (depth 2-8) and 2-cycles are inside the vocabulary here: tags move one hop per NormalizeTag site
        from test_pipeline_logic_cpu import check_chains_are_walked
        check_chains_are_walked(c, got)
        general = p.last_timing()["general_files"]
        print("c3c: %d of %d files in the general kernel" % (general, n))
        assert general <= n // 50


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c3c", "c4", "c5"])
def test_corpus_digests_match_committed_golden(name):
    from imagecaptioner_amd.corpus import make_corpus
    with open(os.path.join(GOLDEN, "corpus_digests.json")) as f:
        g = json.load(f)[name]
    c = make_corpus(name, g["n_files"])
    p = pipe_for(Config(**c.config_kwargs))
    r = p.run_lines(c.arena, c.offsets)
    assert hashlib.sha256(r.captions.tobytes()).hexdigest() == g["captions_sha256"]
    assert hashlib.sha256(r.verdicts.tobytes()).hexdigest() == g["verdicts_sha256"]
    assert r.counters == g["counters"]
    top = sorted(p.histogram_dict(r.histogram).items(), key=lambda kv: (-kv[1], kv[0]))[:20]
    assert top == [tuple(x) for x in g["top_tags"]]


@pytest.mark.parametrize("name", ["shipped", "shipped_nonstrict", "builtin"])
def test_scored_ingest(name):
    cfg = CONFIGS[name]()
    args = scored_inputs(5, 1500)
    p = pipe_for(cfg)
    got, want = p.run_scored(*args), Oracle(cfg).run_scored(*args)
    assert_same_results(got, want, p.histogram_dict(got.histogram), name + " scored")


@pytest.mark.parametrize("name", ["shipped", "shipped_nonstrict", "builtin"])
@pytest.mark.parametrize("cap", [15, 2, 0])
def test_merged_tagger_replies(name, cap):
    """tp_run_merged against MergeCandidateTags(primary, LimitSecondaryTags(...)) of the oracle (Program.cs:214-221)."""
    from test_merged_cpu import merged_inputs
    cfg = CONFIGS[name]()
    args = merged_inputs(41 + cap, 1500)
    p = pipe_for(cfg)
    got, want = p.run_merged(*args, cap), Oracle(cfg).run_merged(*args, cap, threads=8)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "%s merged cap %d" % (name, cap))
    empty = np.zeros(1, np.uint64)
    assert p.run_merged(np.zeros(0, np.uint8), empty, empty, np.zeros(0, np.uint64), np.zeros(0), cap).caption_offsets.tolist() == [0]


@pytest.mark.parametrize("name", ["shipped", "novocab_w07_crlf", "c5"])
def test_audit_strings_and_side_files(name):
    """firedRules text, .needtag decision and .reject.txt content (Program.cs:319-334, 947-956) against the oracle's own lists."""
    if name == "c5":
        from imagecaptioner_amd.corpus import make_corpus
        c = make_corpus("c5", 1500, seed=3)
        cfg = Config(**c.config_kwargs)
        arena, offsets = c.arena, c.offsets
    else:
        cfg = CONFIGS[name]() if name != "shipped" else shipped_config(safe_mode=True)
        lines = EDGE_LINES + fuzz_lines(57, 1500) + ["1girl, nude, smile, dress, long_hair", "1girl, smile, dress, penis, feet, barefoot, soles, toes, footjob"]
        arena, offsets = pack_lines(lines)
    p, o = pipe_for(cfg), Oracle(cfg)
    got, want = p.run_lines(arena, offsets), o.run_lines(arena, offsets, threads=8, detail=True)
    n = len(want.verdicts)
    fired = p.fired_rules(got)
    for i in range(n):
        if got.verdicts["flags"][i] & 0x20:
            continue
        assert fired[i] == ";".join(want.details[i]["fired"]), (i, fired[i], want.details[i]["fired"])
    assert any(fired)
    nl = (cfg.newline if isinstance(cfg.newline, bytes) else cfg.newline.encode()) if getattr(cfg, "newline", None) else None
    for keep in (0, 1):
        act, rej = p.side_files(got.verdicts, keep)
        for i in range(n):
            fl = int(want.verdicts["flags"][i])
            if got.verdicts["flags"][i] & 0x20:
                assert act[i] == 2 and rej[i] == b""
                continue
            assert act[i] == ((1 if keep else 2) if fl & 0x01 else 0)
            reason = want.details[i]["reject_reason"]
            if reason:
                assert rej[i].startswith(reason.encode()) and rej[i][len(reason):] in (b"\n", b"\r\n"), (rej[i], reason)
                if nl:
                    assert rej[i] == reason.encode() + nl
            else:
                assert rej[i] == b""
    if name == "shipped":
        assert any(rej)
    assert p.needtag_text() == b"low tags; short caption or missing core; check later"
    act, rej = p.side_files(got.verdicts[:0], 1)
    assert len(act) == 0 and rej == []


def _audit_args(n, vocab_file="top_tags.txt", used_profile="sgrui"):
    return {"names": ['img "%d", x.png' % i for i in range(n)], "vocab_file": vocab_file, "used_profile": used_profile}


def _check_audit(got, want, nl=b"\n"):
    """audit.csv lines (Program.cs:966-967) and the three tag lists behind them against the oracle's."""
    from helpers import oracle_audit_lists
    exp_lists = oracle_audit_lists(want)
    n = len(exp_lists)
    assert len(got.audit_rows) == n
    for i in range(n):
        if got.verdicts["flags"][i] & 0x20:
            assert got.audit_rows[i] == b"" and got.audit_lists[i] == (b"",) * 5
            continue
        assert got.audit_lists[i][:3] == exp_lists[i], (i, got.audit_lists[i], exp_lists[i])
        assert got.audit_rows[i] == want.audit_rows[i] + nl, (i, got.audit_rows[i], want.audit_rows[i])


@pytest.mark.parametrize("name", ["shipped", "shipped_nonstrict"])
def test_audit_and_review_rows_match_committed_golden(name):
    from golden.make_golden import CONFIGS as GC
    from test_golden import check_audit_golden
    check_audit_golden(pipe_for(GC[name]()).run_merged, name, nl="\n")


def _review_args(n, vocab_file="top_tags.txt", used_profile="sgrui"):
    from test_audit_cpu import review_args
    a = review_args(n, vocab_file)
    a["used_profile"] = used_profile
    return a


def _check_review(got, want, nl=b"\n"):
    """review_candidates.jsonl / .csv lines (Program.cs:871-890) against the oracle's, file by file; b"" where no record is written."""
    n = len(want.review_jsonl)
    assert len(got.review_jsonl) == n and len(got.review_csv) == n
    for i in range(n):
        if got.verdicts["flags"][i] & 0x20:
            assert got.review_jsonl[i] == b"" and got.review_csv[i] == b""
            continue
        assert got.review_jsonl[i] == (want.review_jsonl[i] + nl if want.review_jsonl[i] else b""), (i, got.review_jsonl[i], want.review_jsonl[i])
        assert got.review_csv[i] == (want.review_csv[i] + nl if want.review_csv[i] else b""), (i, got.review_csv[i], want.review_csv[i])


def test_review_rows_known_answer():
    from test_audit_cpu import REVIEW_CSV, REVIEW_JSON, review_args, review_scored_input
    p = pipe_for(shipped_config())
    got = p.run_scored(*review_scored_input(), audit=review_args(1))
    assert got.review_jsonl[0].decode("utf-8") == REVIEW_JSON + "\n"
    assert got.review_csv[0].decode("utf-8") == REVIEW_CSV + "\n"
    assert p.review_csv_header() == (b"image_path,caption_path,reasons,banned_tokens_detected,dropped_tags,final_tag_count,short_caption,thresholds_used,"
                                     b"top_joy_tags,unknown_final_tags,known_dict_found,known_dict_count,known_dict_path,coverage_stop_reason,coverage_trace")
    p = pipe_for(shipped_config(vocabulary=None))
    got = p.run_texts(["girl, blue_hair, blue_eyes, dress, standing, classroom"], audit=review_args(1, vocab_file=None))
    assert got.review_jsonl == [b""] and got.review_csv == [b""]


@pytest.mark.parametrize("name", ["shipped", "shipped_nonstrict", "novocab_w07_crlf", "c5"])
def test_review_rows(name):
    from test_audit_cpu import SUBJECT_LINES
    from test_merged_cpu import merged_inputs
    nl = b"\r\n" if name == "novocab_w07_crlf" else b"\n"
    vf = None if name == "novocab_w07_crlf" else "top_tags.txt"
    if name == "c5":
        from imagecaptioner_amd.corpus import make_corpus
        c = make_corpus("c5", 1500, seed=13)
        cfg = Config(**c.config_kwargs)
        arena, offsets = c.arena, c.offsets
    else:
        cfg = CONFIGS[name]()
        arena, offsets = pack_lines(SUBJECT_LINES + EDGE_LINES + fuzz_lines(5, 1000))
    p, o = pipe_for(cfg), Oracle(cfg)
    audit = _review_args(len(offsets) - 1, vocab_file=vf)
    got, want = p.run_lines(arena, offsets, audit=audit), o.run_lines(arena, offsets, threads=8, detail=True, audit=audit)
    _check_review(got, want, nl)
    _check_audit(got, want, nl)
    assert any(got.review_jsonl)
    if name != "c5":
        arena, toffs, begin, mid, scores = merged_inputs(41, 800)
        audit = _review_args(len(begin) - 1, vocab_file=vf)
        got = p.run_merged(arena, toffs, begin, mid, scores, 15, audit=audit)
        want = o.run_merged(arena, toffs, begin, mid, scores, 15, threads=8, detail=True, audit=audit)
        _check_review(got, want, nl)
        assert any(b"male_subject_token" in x for x in got.review_jsonl)


def test_audit_mode_stays_in_the_wave_stage_on_c2():
    """c2 in audit mode: the lean stage lists what the audit row and the review record want, so only files it cannot take
    (more than 64 tags) reach the thread-per-file kernel.  Among its files are those whose final list holds an action tag
    NormalizeTag changes (Program.cs:738): the row's final_tags keep the tag as the rule names it, the caption has its
    normalised form (Program.cs:972)."""
    from imagecaptioner_amd.corpus import make_corpus
    n = 6000
    c = make_corpus("c2", n, seed=77)
    cfg = Config(**c.config_kwargs)
    p, o = pipe_for(cfg), Oracle(cfg)
    audit = _review_args(n)
    got, want = p.run_lines(c.arena, c.offsets, audit=audit), o.run_lines(c.arena, c.offsets, threads=16, detail=True, audit=audit)
    assert p.last_timing()["general_files"] <= n // 500
    _check_audit(got, want)
    _check_review(got, want)
    assert np.array_equal(got.captions, want.captions) and got.verdicts.tobytes() == want.verdicts.tobytes()
    differ = sum(1 for i in range(n) if got.audit_lists[i][0].replace(b";", b", ") != got.caption(i).rstrip(b"\n"))
    assert differ >= 5, differ
    plain = p.run_lines(c.arena, c.offsets)
    assert plain.verdicts.tobytes() == got.verdicts.tobytes() and np.array_equal(plain.captions, got.captions)


def test_audit_rows_with_action_tags_normalize_tag_changes():
    """The audit row keeps a rule's tag as the rule names it; the caption and the unknown-tag list have its normalised form
    (Program.cs:738, 972, 819).  Crafted collisions, all in the lean stage's detail build."""
    from helpers import changed_action_case
    cfg, lines = changed_action_case()
    p, o = pipe_for(cfg), Oracle(cfg)
    audit = _review_args(len(lines))
    got, want = p.run_texts(lines, audit=audit), o.run_texts(lines, detail=True, audit=audit)
    assert p.last_timing()["general_files"] == 0
    _check_audit(got, want)
    _check_review(got, want)
    assert np.array_equal(got.captions, want.captions) and got.verdicts.tobytes() == want.verdicts.tobytes()
    assert b";bj" in got.audit_lists[0][0] and b"bj" not in got.caption(0).replace(b"blowjob", b"")
    plain = p.run_texts(lines)
    assert plain.verdicts.tobytes() == got.verdicts.tobytes() and np.array_equal(plain.captions, got.captions)


def test_audit_mode_takes_text_outside_ascii():
    """Audit mode (thread-per-file kernel) on lines with characters outside ASCII; the JSON line escapes them the way
    System.Text.Json does."""
    from helpers import NON_ASCII_LINES
    cfg = non_ascii_config()
    p, o = pipe_for(cfg), Oracle(cfg)
    texts = list(NON_ASCII_LINES)
    audit = _review_args(len(texts))
    audit["image_paths"] = ["/data/\u9752 %d.png" % i for i in range(len(texts))]
    got, want = p.run_texts(texts, audit=audit), o.run_texts(texts, detail=True, audit=audit)
    assert not (got.verdicts["flags"] & 0x20).any()
    _check_audit(got, want)
    _check_review(got, want)
    assert any(b"\\u9752" in x for x in got.review_jsonl)
    plain = p.run_texts(texts)
    assert plain.verdicts.tobytes() == got.verdicts.tobytes() and np.array_equal(plain.captions, got.captions)


def test_audit_row_known_answer():
    from test_audit_cpu import KNOWN_AUDIT, KNOWN_ROW, known_scored_input
    p = pipe_for(shipped_config())
    got = p.run_scored(*known_scored_input(), audit=KNOWN_AUDIT)
    assert got.audit_rows[0].decode("utf-8") == KNOWN_ROW + "\n"
    assert p.audit_header() == (b"filename,final_tags,evidence_tags,fired_rules,dropped_tags_with_reason,reject_reason,used_vocab_file,used_profile,"
                                b"coverage_stop_reason,coverage_trace")


@pytest.mark.parametrize("name,n", [("c3c", 4000), ("c1", 4000), ("c4", 3000), ("c5", 3000)])
def test_audit_mode_stays_in_the_wave_stages(name, n):
    """Audit mode beyond the lean stage: a file it declines (c3c: every file holds a tag that moves one alias hop per NormalizeTag
    site; c1: spellings the lean stage does not take) gets its lists from the general wave stages, in the reference's insertion
    order (evidence: seeds, then what each ExpandImpliedTags round appends), so hardly any file needs the thread-per-file kernel."""
    from imagecaptioner_amd.corpus import make_corpus
    c = make_corpus(name, n, seed=515)
    cfg = Config(**c.config_kwargs)
    p, o = pipe_for(cfg), Oracle(cfg)
    audit = _review_args(n)
    got, want = p.run_lines(c.arena, c.offsets, audit=audit), o.run_lines(c.arena, c.offsets, threads=16, detail=True, audit=audit)
    general = p.last_timing()["general_files"]
    print("%s in audit mode: %d of %d files in the general kernel" % (name, general, n))
    assert general <= n // 100
    _check_audit(got, want)
    _check_review(got, want)
    assert_same_results(got, want, p.histogram_dict(got.histogram), name + " audit")
    if name == "c3c":
        deep = sum(1 for i in range(n) if got.audit_lists[i][1].count(b";") > got.audit_lists[i][0].count(b";"))
        assert deep > n // 4          # evidence lists longer than the final lists: closures of several rounds were listed


def test_audit_mode_of_tagger_replies_stays_in_the_wave_kernel():
    """Tagger replies in audit mode (the reference appends an audit row per image whatever the options, Program.cs:334): the wave
    kernel lists the dropped tags, the evidence and the top tagger tags itself - one reply per file as it came (ParseJoyTags sorts
    it) and two replies to merge - so hardly any file needs the thread-per-file kernel; rows and review records equal the oracle's."""
    from imagecaptioner_amd.corpus import make_corpus
    n = 3000
    c = make_corpus("c2", n, seed=808)
    cfg = Config(**c.config_kwargs)
    p, o = pipe_for(cfg), Oracle(cfg)
    raw = bytes(c.arena)
    rng = np.random.default_rng(4)
    tags, begin, mid = [], [0], []
    for i in range(n):
        toks = raw[int(c.offsets[i]):int(c.offsets[i + 1])].rstrip(b"\n").split(b", ")
        tags += toks[:max(1, 2 * len(toks) // 3)]
        mid.append(len(tags))
        tags += toks
        begin.append(len(tags))
    toffs = np.zeros(len(tags) + 1, dtype=np.uint64)
    toffs[1:] = np.cumsum([len(t) for t in tags])
    arena = np.frombuffer(b"".join(tags), dtype=np.uint8).copy()
    scores = rng.random(len(tags)) * 0.8 + 0.2
    scores[rng.random(len(tags)) < 0.05] = 0.5                       # some equal scores, within a reply and between the two
    begin, mid = np.array(begin, dtype=np.uint64), np.array(mid, dtype=np.uint64)
    audit = _review_args(n)
    idx = np.concatenate([np.arange(int(mid[i]), int(begin[i + 1])) for i in range(n)])       # the second replies alone: every tag once
    offs1 = np.zeros(len(idx) + 1, dtype=np.uint64)
    offs1[1:] = np.cumsum([len(tags[k]) for k in idx])
    arena1 = np.frombuffer(b"".join(tags[k] for k in idx), dtype=np.uint8).copy()
    begin1 = np.zeros(n + 1, dtype=np.uint64)
    begin1[1:] = np.cumsum(begin[1:] - mid)
    for what in ("one reply", "one reply that holds most tags twice", "two replies"):
        if what != "two replies":
            a1, o1, b1, s1 = (arena1, offs1, begin1, scores[idx]) if what == "one reply" else (arena, toffs, begin, scores)
            src = np.zeros(len(s1), np.uint8)
            got = p.run_scored(a1, o1, b1, s1, src, audit=audit)
            want = o.run_scored(a1, o1, b1, s1, src, threads=16, detail=True, audit=audit)
        else:
            got = p.run_merged(arena, toffs, begin, mid, scores, 15, audit=audit)
            want = o.run_merged(arena, toffs, begin, mid, scores, 15, threads=16, detail=True, audit=audit)
        general = p.last_timing()["general_files"]
        print("c2 tagger replies in audit mode, %s: %d of %d files in the general kernel" % (what, general, n))
        assert general <= n // 10, what
        _check_audit(got, want)
        _check_review(got, want)
        assert_same_results(got, want, p.histogram_dict(got.histogram), "c2 replies audit, " + what)


@pytest.mark.parametrize("name", ["shipped", "novocab_w07_crlf", "c3", "c5"])
def test_audit_rows_lines(name):
    if name in ("c3", "c5"):
        from imagecaptioner_amd.corpus import make_corpus
        c = make_corpus(name, 2000, seed=9)
        cfg = Config(**c.config_kwargs)
        arena, offsets = c.arena, c.offsets
    else:
        cfg = CONFIGS[name]() if name != "shipped" else shipped_config(safe_mode=True)
        lines = EDGE_LINES + fuzz_lines(91, 1500) + ["1girl, nude, smile, dress, long_hair", "1girl, smile, dress, penis, feet, barefoot, soles, toes, footjob"]
        arena, offsets = pack_lines(lines)
    n = len(offsets) - 1
    audit = _audit_args(n, vocab_file=None if name != "shipped" else "top_tags.txt", used_profile="none" if name != "shipped" else "sgrui")
    p, o = pipe_for(cfg), Oracle(cfg)
    got, want = p.run_lines(arena, offsets, audit=audit), o.run_lines(arena, offsets, threads=8, detail=True, audit=audit)
    nl = b"\r\n" if name == "novocab_w07_crlf" else b"\n"
    _check_audit(got, want, nl)
    assert_same_results(got, want, p.histogram_dict(got.histogram), name)     # audit mode changes nothing else
    plain = p.run_lines(arena, offsets)                                      # and the context is back to the wave kernels afterwards
    assert plain.audit_rows is None and np.array_equal(plain.captions, got.captions) and plain.verdicts.tobytes() == got.verdicts.tobytes()
    assert p.last_timing()["general_files"] < n


@pytest.mark.parametrize("name", ["shipped", "shipped_nonstrict", "builtin"])
def test_audit_rows_tagger_replies(name):
    from test_merged_cpu import merged_inputs
    cfg = CONFIGS[name]()
    p, o = pipe_for(cfg), Oracle(cfg)
    arena, toffs, begin, mid, scores = merged_inputs(77, 1200)
    audit = _audit_args(len(begin) - 1)
    got = p.run_merged(arena, toffs, begin, mid, scores, 15, audit=audit)
    want = o.run_merged(arena, toffs, begin, mid, scores, 15, threads=8, detail=True, audit=audit)
    _check_audit(got, want)
    assert any(l[2] for l in got.audit_lists)                                # dropped tags with their reasons occur
    src = (np.arange(len(scores)) % 3).astype(np.uint8)
    got = p.run_scored(arena, toffs, begin, scores, src, audit=audit)
    want = o.run_scored(arena, toffs, begin, scores, src, threads=8, detail=True, audit=audit)
    _check_audit(got, want)
    assert_same_results(got, want, p.histogram_dict(got.histogram), name)


def test_audit_rows_chunked_and_empty(monkeypatch):
    from imagecaptioner_amd.corpus import make_corpus
    c = make_corpus("c2", 3000, seed=5)
    cfg = Config(**c.config_kwargs)
    audit = _audit_args(3000, vocab_file=None, used_profile="none")
    whole = pipe_for(cfg).run_lines(c.arena, c.offsets, audit=audit)
    monkeypatch.setenv("TAGPIPE_WORKSPACE_BYTES", str(4 << 20))               # the range is halved until a piece fits: several audit passes
    p = pipe_for(cfg)
    chunked = p.run_lines(c.arena, c.offsets, audit=audit)
    assert chunked.audit_rows == whole.audit_rows and chunked.audit_lists == whole.audit_lists
    want = Oracle(cfg).run_lines(c.arena, c.offsets, threads=8, detail=True, audit=audit)
    _check_audit(whole, want)
    empty = p.run_lines(np.zeros(0, np.uint8), np.zeros(1, np.uint64), audit=_audit_args(0))
    assert empty.audit_rows == [] and empty.audit_lists == []


@pytest.mark.parametrize("merged", [False, True])
def test_tagger_replies_take_the_wave_kernel(merged):
    """Replies as a tagger sends them (known tags, each once) stay in the wave kernel's scored front; results equal the oracle's."""
    from test_merged_cpu import clean_replies
    cfg = shipped_config()
    p, o = pipe_for(cfg), Oracle(cfg)
    arena, offs, begin, mid, scores = clean_replies(29, 6000, merged)
    if merged:
        got, want = p.run_merged(arena, offs, begin, mid, scores, 15), o.run_merged(arena, offs, begin, mid, scores, 15, threads=8)
    else:
        src = np.zeros(len(scores), np.uint8)
        got, want = p.run_scored(arena, offs, begin, scores, src), o.run_scored(arena, offs, begin, scores, src, threads=8)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "clean replies merged=%s" % merged)
    assert p.last_timing()["general_files"] < (2000 if merged else 400)


def test_empty_and_ragged_batches():
    cfg = shipped_config()
    p, o = pipe_for(cfg), Oracle(cfg)
    r = p.run_texts([])
    assert r.n_files == 0 and len(r.captions) == 0 and r.counters["files"] == 0
    for lines in (["1girl, smile"], [""] * 65, ["1girl, smile, dress"] * 64 + [""], [", ".join(["long_hair"] * 3000)], ["x" * 70000]):
        got, want = p.run_texts(lines), o.run_texts(lines)
        assert_same_results(got, want, p.histogram_dict(got.histogram), "ragged")


def test_text_outside_ascii_with_the_shipped_tables():
    cfg = shipped_config()
    lines = ["1girl, smile, dress", "1girl, caf\u00e9, dress", "1girl, \u9752\u3044\u9aea"]
    p = pipe_for(cfg)
    got, want = p.run_texts(lines), Oracle(cfg).run_texts(lines)
    assert got.counters["unsupported"] == 0
    assert_same_results(got, want, p.histogram_dict(got.histogram), "shipped tables")


def test_long_files():
    from test_pipeline_logic_cpu import long_lines
    cfg = shipped_config()
    p, o = pipe_for(cfg), Oracle(cfg)
    lines = long_lines(7, 200)
    got, want = p.run_texts(lines), o.run_texts(lines)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "long lines")
    from imagecaptioner_amd.corpus import make_corpus
    c = make_corpus("c5", 3000, seed=5)
    cfg = Config(**c.config_kwargs)
    p, o = pipe_for(cfg), Oracle(cfg)
    got, want = p.run_lines(c.arena, c.offsets), o.run_lines(c.arena, c.offsets, threads=8)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "c5")


def test_common_spellings():
    """Blanks for underscores, capitals, tabs, quotes and full stops around tags: the byte-to-byte route and the byte-wise normaliser."""
    from test_pipeline_logic_cpu import spelling_variants
    lines, cfg = spelling_variants(4000, seed=9)
    p, o = pipe_for(cfg), Oracle(cfg)
    got, want = p.run_texts(lines), o.run_texts(lines, threads=8)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "spellings")
    assert p.last_timing()["general_files"] == 0


def test_multi_line_texts():
    """Line feeds inside the text are the blanks they become (Program.cs:595), "\r\n" pairs shrink to them first (Program.cs:591)."""
    from test_pipeline_logic_cpu import multi_line_texts
    lines, cfg = multi_line_texts(4000, seed=6)
    p, o = pipe_for(cfg), Oracle(cfg)
    got, want = p.run_texts(lines), o.run_texts(lines, threads=8)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "multi-line")
    assert p.last_timing()["general_files"] == 0
    assert np.array_equal(p.triage_texts(lines), o.triage_texts(lines))
    odd = ["a\rb, c, d, e", "1girl,\r\nsmile,\r\n\r\ndress", "1girl, sm\r\nile, dress", "i\r\ncan't do that, x, y", "1girl\r, smile, dress", "tag`x, a\r\nb, c"]
    got, want = p.run_texts(odd), o.run_texts(odd)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "carriage returns")


@pytest.mark.parametrize("strict", [True, False])
def test_text_outside_ascii(strict):
    """The Unicode half of NormalizeTag (Trim and \\s over Unicode white space, the ideographic stops, ToLowerInvariant;
    Program.cs:1150-1154) on the device: no such line is unsupported, and all equal the oracle's."""
    cfg = non_ascii_config() if strict else non_ascii_config(strict_known=False, strict_known_explicit=True)
    p = pipe_for(cfg)
    check_non_ascii(p.run_lines, Oracle(cfg), p.histogram_dict)
    if strict:
        from helpers import check_case_twins
        check_case_twins(pipe_for)


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_existing_caption_triage(name):
    """tp_triage_lines against LooksBad(NormalizeRawTagLine(text)) of the oracle (Program.cs:203-210)."""
    from helpers import triage_lines
    from test_pipeline_logic_cpu import long_lines
    cfg = CONFIGS[name]()
    p, o = pipe_for(cfg), Oracle(cfg)
    lines = triage_lines(5, 4000) + fuzz_lines(31, 1500) + EDGE_LINES + long_lines(9, 40)
    got, want = p.triage_texts(lines), o.triage_texts(lines)
    assert not (got == 2).any()
    diff = np.nonzero(got != want)[0]
    assert diff.size == 0, [(lines[i], int(got[i]), int(want[i])) for i in diff[:5]]
    assert 0 < want.sum() < len(lines)
    assert p.triage_texts([]).size == 0


def test_triage_of_the_pipelines_own_captions():
    """What the pipeline writes is what a later run finds on disk: triage of those captions agrees with the oracle, and a
    caption of three or more tags is never called bad."""
    from imagecaptioner_amd.corpus import make_corpus
    c = make_corpus("c2", 20000, seed=13)
    cfg = Config(**c.config_kwargs)
    p, o = pipe_for(cfg), Oracle(cfg)
    r = p.run_lines(c.arena, c.offsets)
    got = p.triage_lines(r.captions, r.caption_offsets)
    texts = [r.caption(i) for i in range(20000)]
    want = o.triage_texts(texts)
    assert np.array_equal(got, want)
    assert not got[r.verdicts["n_tags"] >= 3].any()


def test_workspace_chunking_gives_identical_results(monkeypatch):
    from imagecaptioner_amd.corpus import make_corpus
    c = make_corpus("c2", 5000, seed=31)
    cfg = Config(**c.config_kwargs)
    whole = pipe_for(cfg).run_lines(c.arena, c.offsets)
    monkeypatch.setenv("TAGPIPE_WORKSPACE_BYTES", str(4 << 20))   # forces many chunks
    p = pipe_for(cfg)
    chunked = p.run_lines(c.arena, c.offsets)
    assert np.array_equal(whole.captions, chunked.captions) and np.array_equal(whole.caption_offsets, chunked.caption_offsets)
    assert whole.verdicts.tobytes() == chunked.verdicts.tobytes() and whole.counters == chunked.counters
    assert np.array_equal(whole.histogram, chunked.histogram)


@pytest.mark.parametrize("name", ["shipped", "shipped_nonstrict"])
def test_marks_at_the_scan_step_boundaries(name):
    """Long lines with a token the line scan has to mark, count or refuse over at every byte offset around its step boundaries
    (16 bytes a lane, a kilobyte a step in the stages for long lines): equal to the oracle, most of them in the wave stages."""
    from test_pipeline_logic_cpu import scan_boundary_lines
    cfg = CONFIGS[name]()
    lines = scan_boundary_lines()
    p = pipe_for(cfg)
    got, want = p.run_texts(lines), Oracle(cfg).run_texts(lines)
    assert_same_results(got, want, p.histogram_dict(got.histogram), name + " scan boundaries")
    assert p.last_timing()["general_files"] < len(lines) * 3 // 4


@pytest.mark.parametrize("name", ["c5", "c1"])
def test_side_chain_gives_identical_results(monkeypatch, name):
    """The files the lean stage's front half declines go through the later stages on a stream of their own beside the back half,
    the files the back half declines after it (the stage's list is filled from both ends): the same arrays as the stages in a
    row on one stream, and the oracle's, on corpora where both halves decline files and every later stage has work."""
    from imagecaptioner_amd.corpus import make_corpus
    c = make_corpus(name, 30000, seed=99)
    cfg = Config(**c.config_kwargs)
    runs = {}
    for v in ("1", "0"):
        monkeypatch.setenv("TAGPIPE_SIDE_CHAIN", v)
        p = pipe_for(cfg)
        runs[v] = [p.run_lines(c.arena, c.offsets) for _ in range(2)][1]
        if v == "1":
            hist = p.histogram_dict(runs[v].histogram)
    a, b = runs["1"], runs["0"]
    assert np.array_equal(a.caption_offsets, b.caption_offsets) and np.array_equal(a.captions, b.captions)
    assert a.verdicts.tobytes() == b.verdicts.tobytes() and a.counters == b.counters
    assert np.array_equal(a.histogram, b.histogram) and np.array_equal(a.fired, b.fired) and np.array_equal(a.facet_choice, b.facet_choice)
    want = Oracle(cfg).run_lines(c.arena, c.offsets, threads=16)
    assert_same_results(a, want, hist, name + " with the side chain")


def test_host_run_in_pieces_equals_the_run_in_turn(monkeypatch):
    """tp_run_lines over many files works in pieces (the next piece's lines going in and the last one's captions coming back while
    the kernels run): the same arrays as copying in, computing and copying out in turn, and the oracle's captions for a sample."""
    from imagecaptioner_amd.corpus import make_corpus
    c = make_corpus("c5", 300000, seed=77)                        # three pieces, the last one short; long files, so every stage has work
    cfg = Config(**c.config_kwargs)
    p = pipe_for(cfg)
    first = p.run_lines(c.arena, c.offsets)
    again = p.run_lines(c.arena, c.offsets)                          # the second call finds its buffers at their final sizes
    monkeypatch.setenv("TAGPIPE_NO_OVERLAP", "1")
    turn = pipe_for(cfg).run_lines(c.arena, c.offsets)
    for got in (first, again):
        assert np.array_equal(got.caption_offsets, turn.caption_offsets) and np.array_equal(got.captions, turn.captions)
        assert got.verdicts.tobytes() == turn.verdicts.tobytes() and got.counters == turn.counters
        assert np.array_equal(got.histogram, turn.histogram) and np.array_equal(got.fired, turn.fired) and np.array_equal(got.facet_choice, turn.facet_choice)
    k = 2000
    want = Oracle(cfg).run_lines(c.arena[:int(c.offsets[k])], c.offsets[:k + 1], threads=16)
    assert bytes(first.captions[:int(first.caption_offsets[k])]) == bytes(want.captions[:int(want.caption_offsets[k])])


def test_device_api_capacity_and_accumulation():
    import torch
    from imagecaptioner_amd._abi import TP_E_CAPACITY, tp_device_io
    from imagecaptioner_amd import TagPipeError
    from imagecaptioner_amd.corpus import make_corpus
    c = make_corpus("c1", 3000, seed=8)
    cfg = Config(**c.config_kwargs)
    p = pipe_for(cfg)
    host = p.run_lines(c.arena, c.offsets)
    dev = torch.device("cuda", 0)
    arena = torch.cat([torch.from_numpy(c.arena).to(dev), torch.zeros(64, dtype=torch.uint8, device=dev)])
    offsets = torch.from_numpy(c.offsets.astype(np.int64)).to(dev)
    n = c.n_files
    caps = torch.empty(len(host.captions), dtype=torch.uint8, device=dev)
    cap_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ver = torch.empty(n * 16, dtype=torch.uint8, device=dev)
    hist = torch.zeros(p.info.n_strings + 1, dtype=torch.int64, device=dev)
    ctr = torch.zeros(8, dtype=torch.int64, device=dev)
    io = tp_device_io()
    io.arena, io.offsets, io.n_files = arena.data_ptr(), offsets.data_ptr(), n
    io.captions, io.caption_capacity, io.caption_offsets = caps.data_ptr(), len(host.captions) - 1, cap_off.data_ptr()
    io.verdicts, io.histogram, io.counters = ver.data_ptr(), hist.data_ptr(), ctr.data_ptr()
    io.stream = torch.cuda.current_stream().cuda_stream
    with pytest.raises(TagPipeError) as e:
        p.run_lines_device(io)
    assert e.value.code == TP_E_CAPACITY and e.value.required == len(host.captions)
    io.caption_capacity = len(host.captions)
    hist.zero_(); ctr.zero_()
    assert p.run_lines_device(io) == len(host.captions)
    p.run_lines_device(io)            # histogram and counters accumulate across calls
    torch.cuda.synchronize()
    assert np.array_equal(caps.cpu().numpy(), host.captions)
    assert np.array_equal(cap_off.cpu().numpy().astype(np.uint64), host.caption_offsets)
    got_ver = ver.cpu().numpy().copy().view(host.verdicts.dtype)
    assert np.all(got_ver["n_evidence"] == 0xFFFF)    # no detail arrays were asked for: the evidence set is not counted
    got_ver["n_evidence"] = host.verdicts["n_evidence"]
    assert got_ver.tobytes() == host.verdicts.tobytes()
    # with the detail arrays the count is there
    fired = torch.zeros(max(n * p.info.fired_words, 1), dtype=torch.int32, device=dev)
    io.fired = fired.data_ptr()
    hist2, ctr2 = torch.zeros_like(hist), torch.zeros_like(ctr)
    io.histogram, io.counters = hist2.data_ptr(), ctr2.data_ptr()
    p.run_lines_device(io)
    torch.cuda.synchronize()
    assert ver.cpu().numpy().tobytes() == host.verdicts.tobytes()
    assert np.array_equal(fired.cpu().numpy().view(np.uint32)[:n * p.info.fired_words].reshape(n, -1), host.fired)
    assert np.array_equal(hist.cpu().numpy()[:-1].astype(np.uint64), 2 * host.histogram)
    assert int(ctr[0]) == 2 * n


def test_full_size_properties():
    """BASELINE config 2 at full size (1M files): properties that need no oracle run — shard concatenation equals the
    whole batch, the histogram sums to the number of caption tags, reruns are deterministic — plus oracle parity on a prefix."""
    from imagecaptioner_amd.corpus import make_corpus
    n = 1_000_000
    c = make_corpus("c2", n, seed=1234)
    cfg = Config(**c.config_kwargs)
    p = pipe_for(cfg)
    whole = p.run_lines(c.arena, c.offsets)
    assert whole.counters["files"] == n and whole.counters["unsupported"] == 0
    assert int(whole.histogram.sum()) + whole.histogram_other == int(whole.verdicts["n_tags"].astype(np.int64).sum())
    again = p.run_lines(c.arena, c.offsets)
    assert hashlib.sha256(again.captions.tobytes()).digest() == hashlib.sha256(whole.captions.tobytes()).digest()
    # two shards, concatenated
    half = n // 2
    first = int(c.offsets[half])
    a = p.run_lines(c.arena[:first], c.offsets[:half + 1])
    b = p.run_lines(c.arena[first:], c.offsets[half:] - c.offsets[half])
    assert np.array_equal(np.concatenate([a.captions, b.captions]), whole.captions)
    assert np.array_equal(a.histogram + b.histogram, whole.histogram)
    m = 20000
    want = Oracle(cfg).run_lines(c.arena, c.offsets[:m + 1], threads=16)
    assert np.array_equal(whole.captions[:int(whole.caption_offsets[m])], want.captions)
    assert whole.verdicts[:m].tobytes() == want.verdicts.tobytes()


# ---- table files in every shape the reference's loaders accept (shapes and reference lines: loader_shapes.py) ----
def _shape_names():
    from loader_shapes import SHAPES
    return sorted(SHAPES)


@pytest.mark.parametrize("name", _shape_names())
def test_loader_shapes_match_the_oracle(name):
    from loader_shapes import SHAPES, shape_lines
    cfg = SHAPES[name]()
    p = pipe_for(cfg)
    lines = shape_lines()
    o = Oracle(cfg)
    got, want = p.run_texts(lines), o.run_texts(lines)
    assert_same_results(got, want, p.histogram_dict(got.histogram), name)
    if name.startswith("force_allowed_commas"):
        # a force-allowed tag with a comma: its pieces are the candidates (Program.cs:604, 578), in audit mode and triage too
        assert want.verdicts["n_candidates"][lines.index("ab")] == 2
        audit = _review_args(len(lines))
        _check_audit(p.run_texts(lines, audit=audit), o.run_texts(lines, detail=True, audit=audit))
        texts = lines + ["xy, " + "z" * 40, "x, " + "z" * 40, "gone, " + "z" * 40, "ab" + " " * 50 + ", smile"]
        assert list(p.triage_texts(texts)) == list(o.triage_texts(texts))


@pytest.mark.parametrize("name", ["case_mapping_twins", "five_tags_with_commas", "nine_pieces"])
def test_tables_this_build_refuses(name):
    """Tables the reference loads and this build cannot represent: tp_create fails with TP_E_CONFIG and says why."""
    from imagecaptioner_amd import TagPipeError
    from imagecaptioner_amd._abi import TP_E_CONFIG
    from loader_shapes import REJECTED
    with pytest.raises(TagPipeError) as e:
        pipe_for(REJECTED[name]())
    assert e.value.code == TP_E_CONFIG and "not representable" in str(e.value)


def test_top_tags_file_order_and_round_trip():
    """tp_format_top_tags through the C ABI, with the histogram the kernels counted."""
    from helpers import check_top_tags_file
    check_top_tags_file(pipe_for)


@pytest.mark.parametrize("kind", ["nan", "inf", "-inf"])
def test_non_finite_scores_are_refused(kind):
    """A NaN or an infinity in `scores` fails tp_run_scored / tp_run_merged with TP_E_INVALID_ARG before anything runs."""
    from imagecaptioner_amd import TagPipeError
    from imagecaptioner_amd._abi import TP_E_INVALID_ARG
    from test_pipeline_logic_cpu import odd_scores
    arena, offs, begin, scores, sources = odd_scores(kind)
    p = pipe_for(CONFIGS["shipped"]())
    with pytest.raises(TagPipeError) as e:
        p.run_scored(arena, offs, begin, scores, sources)
    assert e.value.code == TP_E_INVALID_ARG and "finite" in str(e.value)
    with pytest.raises(TagPipeError) as e:
        p.run_merged(arena, offs, begin, begin[:-1].copy(), scores, 15)
    assert e.value.code == TP_E_INVALID_ARG


def test_unusual_finite_scores():
    from test_pipeline_logic_cpu import odd_scores
    cfg = CONFIGS["shipped"]()
    args = odd_scores("finite")
    p = pipe_for(cfg)
    got, want = p.run_scored(*args), Oracle(cfg).run_scored(*args)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "finite scores")


def test_tagger_reply_tags_outside_ascii():
    """Scored ingest keeps unknown tags with characters outside ASCII as candidates; the dropped list spells them as NormalizeTag
    leaves them (Program.cs:1150-1154, 630-631)."""
    from test_audit_cpu import unicode_reply_tags
    cfg = non_ascii_config()
    p, o = pipe_for(cfg), Oracle(cfg)
    arena, toffs, begin, scores, src = unicode_reply_tags()
    audit = _review_args(len(begin) - 1)
    got = p.run_scored(arena, toffs, begin, scores, src, audit=audit)
    want = o.run_scored(arena, toffs, begin, scores, src, detail=True, audit=audit)
    assert got.counters["unsupported"] == 0
    assert_same_results(got, want, p.histogram_dict(got.histogram), "reply tags outside ASCII")
    _check_audit(got, want)
    _check_review(got, want)
    assert "caf\u00e9_au_lait:invalid_tag".encode("utf-8") in got.audit_lists[0][2]


def test_cached_tagger_replies_decode_like_parse_joy_tags():
    """tp_decode_joytag_replies through the C ABI, its arrays through tp_run_scored / tp_run_merged, against the oracle reading
    the same JSON bodies (CallJoyTagAsync from the body on and ParseJoyTags, Program.cs:463-489, 500-505)."""
    from helpers import check_joytag_replies
    cfg = shipped_config()
    check_joytag_replies(pipe_for(cfg), Oracle(cfg))


def test_replies_without_scores_keep_their_order():
    """tp_run_replies: a reply in which no element carries a score is not sorted by ParseJoyTags (Program.cs:487), so the audit
    row lists its evidence and its dropped tags in the reply's order; the secondary reply of a pair is sorted by
    LimitSecondaryTags whatever it holds (Program.cs:805)."""
    from helpers import UNSORTED_REPLIES, UNSORTED_SECOND
    cfg = shipped_config()
    p, o = pipe_for(cfg), Oracle(cfg)
    audit = _review_args(len(UNSORTED_REPLIES))
    d = p.decode_joytag_replies(UNSORTED_REPLIES)
    assert list(d["status"]) == [2, 2, 0, 0] and list(d["has_score"]) == [0, 0, 0, 1]
    got = p.run_replies(d, audit=audit)
    want = o.run_replies(UNSORTED_REPLIES, detail=True, audit=audit)
    _check_audit(got, want)
    _check_review(got, want)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "unsorted replies")
    assert_same_results(p.run_replies(d), want, p.histogram_dict(got.histogram), "unsorted replies, plain run")
    sorted_run = p.run_scored(d["arena"], d["tag_offsets"], d["tag_begin"], d["scores"], d["sources"], audit=audit)
    assert sorted_run.audit_lists[0][1] != got.audit_lists[0][1] and sorted_run.audit_lists[0][0] == got.audit_lists[0][0]
    d = p.decode_joytag_replies(UNSORTED_REPLIES, UNSORTED_SECOND)
    for cap in (15, 1):
        got = p.run_replies(d, secondary_cap=cap, audit=audit)
        want = o.run_replies(UNSORTED_REPLIES, UNSORTED_SECOND, secondary_cap=cap, detail=True, audit=audit)
        _check_audit(got, want)
        _check_review(got, want)
        assert_same_results(got, want, p.histogram_dict(got.histogram), "unsorted replies, merged, cap %d" % cap)


def test_bench_two_ranks_share_this_gpu():
    """`bench.py --gpus 2` as its own launcher, both ranks on the one GPU of this box, sums through gloo (RCCL refuses two ranks on
    one GPU): the ranks' shards go through tp_run_lines_device, the summed histogram adds up, and rank 0 prints the line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-gpu", "--files", "30000", "--steps", "2",
                        "--warmup", "1", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["files_per_gpu"] == 30000 and line["value"] > 0 and line["scaling"] == "weak"


def test_bench_strong_scaling_two_ranks_share_this_gpu():
    """`bench.py --gpus 2 --scaling strong`: ONE corpus (the committed golden size of c2) cut into two contiguous shards, both
    ranks on this box's GPU, sums through gloo.  Rank 0 checks the shards against its own run over the whole corpus, the summed
    histogram and counters against that run's, and the whole run's digest against tests/golden/corpus_digests.json."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-gpu", "--scaling", "strong", "--files", "1000",
                        "--steps", "2", "--warmup", "1", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    chk = line["config"]["strong_scaling_check"]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["collective"]["world_size_seen"] == 2
    assert chk["files_in_all"] == 1000 and chk["shard_files"] == [500, 500] and chk["whole_run_checked"] and chk["golden_checked"]


def test_strings_in_no_table_are_learned():
    """c2 without a vocabulary file: tags in no table are kept (Program.cs:1121-1133 with KnownTagDictionaryFound false), and a
    file that holds one takes the thread-per-file kernel.  The library notes such strings and compiles its tables again
    with them interned behind the configuration's own, so the second call over the same kind of input stays in the wave
    kernels; results, ids and histogram are the same before and after (a learned tag counts with the tags in no table)."""
    from imagecaptioner_amd.corpus import make_corpus
    n = 6000
    c = make_corpus("c2n", n, seed=4243)
    cfg = Config(**c.config_kwargs)
    want = Oracle(cfg).run_lines(c.arena, c.offsets, threads=16)
    p = pipe_for(cfg)
    n_strings = p.info.n_strings
    got = p.run_lines(c.arena, c.offsets)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "c2n, first call")
    first = p.last_timing()["general_files"]
    assert first > n // 2, "nearly every file holds a tag in no table"
    got = p.run_lines(c.arena, c.offsets)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "c2n, second call")
    second = p.last_timing()["general_files"]
    print("c2n: %d of %d files in the general kernel on the first call, %d on the second" % (first, n, second))
    assert second <= n // 100
    assert p.info.n_strings == n_strings and got.histogram_other > 0
    # other files of the same corpus: mostly strings met before, the rest is learned in turn
    c2 = make_corpus("c2n", n, seed=999)
    got = p.run_lines(c2.arena, c2.offsets)
    assert_same_results(got, Oracle(cfg).run_lines(c2.arena, c2.offsets, threads=16), p.histogram_dict(got.histogram), "c2n, other files")


def test_fallback_caption_is_the_failure_path_line():
    """tp_fallback_caption = string.Join(", ", SelectTags(ParseTagCandidatesFromLine(minimalFallback, null, 1.0, OpenAI)).Tags)
    + newline (Program.cs:349-350), against the oracle's SelectTags on its own minimalFallback; several configurations."""
    for name in sorted(CONFIGS):
        cfg = CONFIGS[name]()
        p, o = pipe_for(cfg), Oracle(cfg)
        nl = cfg.struct.newline or b"\n"
        assert p.fallback_caption() == ", ".join(o.select_tags_from_line(o.minimal_fallback, 1.0)).encode("utf-8") + nl, name
    assert pipe_for(CONFIGS["shipped"]()).error_needtag_text() == b"error; check later"      # Program.cs:353


def test_noted_strings_do_not_point_into_a_freed_arena():
    """Tags in no table that tp_run_lines_device_noalloc meets are noted as copies of their bytes: the caller frees (here:
    overwrites, then drops) the arena, and the next allocating call still learns the right strings from what was noted."""
    import torch
    from imagecaptioner_amd.corpus import make_corpus
    from test_zz_no_alloc import DeviceRun
    n = 3000
    c = make_corpus("c2n", n, seed=4244)
    cfg = Config(**c.config_kwargs)
    p = pipe_for(cfg)
    r = DeviceRun(p, c).run(noalloc=True)
    r.arena.fill_(0x41)                                       # what the notes used to point at is gone
    del r
    torch.cuda.empty_cache()
    other = make_corpus("c2", 50, seed=5)                     # strict-free tables, another input: the call collects what was noted before
    got = p.run_lines(other.arena, other.offsets)
    assert got.counters["files"] == 50
    got = p.run_lines(c.arena, c.offsets)                     # the noted strings are interned now: the wave kernels take the corpus
    want = Oracle(cfg).run_lines(c.arena, c.offsets, threads=16)
    assert_same_results(got, want, p.histogram_dict(got.histogram), "c2n after a freed arena")
    assert p.last_timing()["general_files"] <= n // 100


def test_hand_derived_known_answers():
    """known_answers.py (coverage strings, facet choices, closure order, unknown-tag order, the 0.03 window) through the C ABI."""
    import known_answers as ka
    assert ka.check_known_answers(pipe_for, shipped_config) == 6
