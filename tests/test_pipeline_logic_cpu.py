"""Logic check of the kernels' per-file pipeline source without a GPU.

tests/hostsim compiles imagecaptioner_amd/csrc/tp_pipeline.h and the table compiler for the host (test-only;
the product library never does) and runs files through the same lane-interleaved workspace layout the kernel
uses.  Results must be bit-identical to the oracle's.
"""
import os
import random

import numpy as np
import pytest

from helpers import DATA, EDGE_LINES, NON_ASCII_LINES, Config, HostSim, Oracle, assert_same_results, check_non_ascii, non_ascii_config, shipped_config
from imagecaptioner_amd.corpus import make_corpus

CONFIGS = {
    "shipped": lambda: shipped_config(),
    "shipped_nonstrict": lambda: shipped_config(strict_known=False, strict_known_explicit=True),
    "builtin": lambda: Config(trigger="trigger_token", minimal_subject_tags="1girl", no_profile=True),
    "novocab_w07_crlf": lambda: shipped_config(vocabulary=None, profile=None, trigger="My Trigger", auto_prefix_tags="1girl, solo",
                                               minimal_subject_tags="smile", safe_mode=False, newline="\r\n", line_weight=0.7),
    "allow_deny": lambda: shipped_config(allowlist=b"# comment\n1girl\nsmile\n", denylist=b"dress\nlong hair\n"),
}


def fuzz_lines(seed, n):
    rng = random.Random(seed)
    vocab = [l.strip() for l in open(DATA + "/top_tags.txt") if l.strip()][:600]
    pool = vocab + ["pedal stimulation", "foot", "soles", "fingers", "licking", "fellatio", "sexual_intercourse", "barefoot", "toes", "penis", "feet",
                    "hand", "grip", "between_feet", "cum", "ejaculation", "male_pov", "no_panties", "1boy", "text", "unknown_thing", "zzz_not_in_vocab",
                    "my_oc_name", "some random words", "watermark_big", "rating_x"]
    junk = ["'", '"', "`", ".", ";", " ", "  ", "_", "__", "\t", "\n", "\r\n", ":", "!", "(", ")", "/", "-", ",", "A", "Z", "x", "9", "sorry",
            "POSITIVE:", "NEGATIVE:", "```", "i can't", "watermark", "rating"]
    out = []
    for _ in range(n):
        k = rng.choice([0, 1, 2, 3, 5, 8, 13, 30, 60, 100])
        toks = []
        for _ in range(k):
            r = rng.random()
            t = rng.choice(pool)
            if r < 0.15:
                t = t.upper()
            elif r < 0.25:
                t = t.replace("_", " ")
            elif r < 0.35:
                t = rng.choice(junk) + t + rng.choice(junk)
            elif r < 0.40:
                t = t + rng.choice(["'", "'.", "\".", "';.", "`..", ".'.'"])
            elif r < 0.45:
                t = "".join(rng.choice("abcxyz_09 '.-") for _ in range(rng.randint(0, 12)))
            elif r < 0.47:
                t = rng.choice(junk)
            toks.append(t)
        line = rng.choice([", ", ",", " , ", ",  "]).join(toks)
        if rng.random() < 0.1:
            line = rng.choice(junk) + line + rng.choice(junk)
        out.append(line)
    return out


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_edge_and_fuzz_lines(name):
    cfg = CONFIGS[name]()
    o, s = Oracle(cfg), HostSim(cfg)
    lines = EDGE_LINES + fuzz_lines(11, 1200)
    want, got = o.run_texts(lines), s.run_texts(lines)
    assert got.counters["unsupported"] == 0
    assert_same_results(got, want, s.histogram_dict(got.histogram), name)


def chain_tags(text):
    """Members of c3c's alias chains and cycles among the tags of a line."""
    return [t for t in text.strip().split(", ") if t.startswith(("syn", "cyc"))]


def check_chains_are_walked(corpus, result):
    """c3c: chain members occur in the input, many of them reach the plain tag at the end of their chain on the way
    through the NormalizeTag sites, and some are still chain members in the caption."""
    n_in = sum(len(chain_tags(corpus.line(i).decode())) for i in range(corpus.n_files))
    n_out = sum(len(chain_tags(result.caption(i).decode())) for i in range(corpus.n_files))
    assert n_in > corpus.n_files and 0 < n_out < n_in, (n_in, n_out)


@pytest.mark.parametrize("name,n", [("c1", 600), ("c2", 800), ("c3", 500), ("c3c", 500), ("c4", 500), ("c5", 200)])
def test_benchmark_corpora(name, n):
    c = make_corpus(name, n, seed=77)
    cfg = Config(**c.config_kwargs)
    o, s = Oracle(cfg), HostSim(cfg)
    want, got = o.run_lines(c.arena, c.offsets, threads=8), s.run_lines(c.arena, c.offsets)
    assert got.counters["unsupported"] == 0
    assert_same_results(got, want, s.histogram_dict(got.histogram), name)
    if name == "c3c":
        check_chains_are_walked(c, want)


def scored_inputs(seed, n_files):
    rng = random.Random(seed)
    vocab = [l.strip() for l in open(DATA + "/top_tags.txt") if l.strip()][:400] + ["foot", "fingers", "Pedal Stimulation", "zz_unknown", "x'.", "", " ", "a"]
    tags, begin, scores, sources = [], [0], [], []
    for _ in range(n_files):
        k = rng.choice([0, 1, 3, 10, 40, 90])
        for _ in range(k):
            tags.append(rng.choice(vocab))
            scores.append(round(rng.random(), 2) if rng.random() < 0.8 else rng.choice([0.28, 0.6, 0.5, 0.53, 0.47]))
            sources.append(rng.choice([0, 0, 1, 2]))
        begin.append(len(tags))
    enc = [t.encode() for t in tags]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offs[1:] = np.cumsum([len(b) for b in enc])
    arena = np.frombuffer(b"".join(enc), dtype=np.uint8).copy() if enc else np.zeros(0, np.uint8)
    return arena, offs, np.array(begin, dtype=np.uint64), np.array(scores, dtype=np.float64), np.array(sources, dtype=np.uint8)


@pytest.mark.parametrize("name", ["shipped", "shipped_nonstrict", "builtin"])
def test_scored_ingest(name):
    cfg = CONFIGS[name]()
    o, s = Oracle(cfg), HostSim(cfg)
    args = scored_inputs(5, 400)
    want, got = o.run_scored(*args), s.run_scored(*args)
    assert_same_results(got, want, s.histogram_dict(got.histogram), name + " scored")


@pytest.mark.parametrize("strict", [True, False])
def test_text_outside_ascii(strict):
    """The Unicode half of NormalizeTag (Program.cs:1150-1154) in the thread-per-file pipeline."""
    cfg = non_ascii_config() if strict else non_ascii_config(strict_known=False, strict_known_explicit=True)
    s = HostSim(cfg)
    check_non_ascii(s.run_lines, Oracle(cfg), s.histogram_dict)
    cfg = shipped_config()
    lines = ["1girl, smile, dress", "1girl, café, dress", "1girl, 青い髪"]
    s = HostSim(cfg)
    got, want = s.run_texts(lines), Oracle(cfg).run_texts(lines)
    assert got.counters["unsupported"] == 0
    assert_same_results(got, want, s.histogram_dict(got.histogram), "shipped tables")
    if strict:
        from helpers import check_case_twins
        check_case_twins(HostSim)


def long_lines(seed, count):
    """Lines of 65 .. 1025 tokens from the shipped vocabulary, with repeats, unknown tags, upper case and odd spacing."""
    import random
    vocab = [l.strip() for l in open(os.path.join(DATA, "top_tags.txt"), encoding="utf-8") if l.strip()]
    rng = random.Random(seed)
    lines = []
    for _ in range(count):
        n = rng.choice([65, 100, 129, 200, 400, 700, 1000, 1024, 1025])
        toks = []
        for _i in range(n):
            r = rng.random()
            if r < 0.85: tok = rng.choice(vocab[:3000])
            elif r < 0.92: tok = rng.choice(toks) if toks else "smile"
            elif r < 0.96: tok = "zz_unknown_%d" % rng.randrange(50)
            else: tok = rng.choice(vocab).upper()
            toks.append(tok)
        lines.append(rng.choice([", ", ",", " , ", ",  "]).join(toks) + rng.choice(["", "\n", "  \n"]))
    return lines


def scan_boundary_lines():
    """Long lines (200 to 330 tokens) in which one token the line scan has to mark, count or refuse over - capitals, quotes, runs of
    blanks or underscores, trailing punctuation, a refusal phrase, a keyword with its colon, a tab, a character outside ASCII, a
    fence - begins at every byte offset around the scan's step boundaries (multiples of 16 bytes up to 1024 + 48 into the line)."""
    marks = ["Smile", "'smile'", "long  hair", "long__hair", "smile.", "sorry", "i can't", "positive: smile", "long\thair", "caf\u00e9", "smi`le", "SMILE;", "smile ,dress"]
    lines = []
    for at in list(range(0, 36)) + list(range(1000, 1072)):
        head = ""
        while len(head) + 8 <= at:
            head += "dress, "[: min(7, at - len(head))] if at - len(head) < 7 else "dress, "
        head = head + "x" * (at - len(head)) if at > len(head) else head
        mark = marks[at % len(marks)]
        tail = ", ".join(["smile", "long_hair", "1girl", "dress", "blue_eyes", "standing", "hat", "solo"] * (25 + at % 17))
        lines.append((head + ", " if head and not head.endswith(", ") else head) + mark + ", " + tail)
    return lines


@pytest.mark.parametrize("name", ["shipped", "shipped_nonstrict"])
def test_marks_at_the_scan_step_boundaries(name):
    cfg = CONFIGS[name]()
    o, s = Oracle(cfg), HostSim(cfg)
    lines = scan_boundary_lines()
    want, got = o.run_texts(lines), s.run_texts_fast(lines)
    assert_same_results(got, want, s.histogram_dict(got.histogram), name + " scan boundaries")
    assert got.n_general < len(lines) * 3 // 4          # (refusal phrases, keywords and fences are the general kernel's)


@pytest.mark.parametrize("name", ["shipped", "shipped_nonstrict"])
def test_long_files_take_the_long_wave_stage(name):
    cfg = CONFIGS[name]()
    o, s = Oracle(cfg), HostSim(cfg)
    lines = long_lines(3, 60)
    want, got = o.run_texts(lines), s.run_texts_fast(lines)
    assert_same_results(got, want, s.histogram_dict(got.histogram), name + " long")
    if name == "shipped":
        assert got.n_general < len(lines) // 2      # most of them fit the long-file stage


def test_text_outside_ascii_through_the_wave_stages():
    """A token with bytes outside ASCII that is a known string as it stands stays in the wave kernels; every other such file is
    handed to the thread-per-file pipeline, none is reported as unsupported."""
    cfg = non_ascii_config()
    s = HostSim(cfg)
    check_non_ascii(s.run_lines_fast, Oracle(cfg), s.histogram_dict)
    check_non_ascii(s.run_lines_lean, Oracle(cfg), s.histogram_dict)
    got = s.run_texts_fast(NON_ASCII_LINES[:3])
    assert got.n_general == 0


# ---- the wave-per-file kernel body (tp_fast.h) on an emulated wavefront ---------------------------------------------
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_wave_kernel_edge_and_fuzz_lines(name):
    cfg = CONFIGS[name]()
    o, s = Oracle(cfg), HostSim(cfg)
    lines = EDGE_LINES + fuzz_lines(23, 300)
    want, got = o.run_texts(lines), s.run_texts_fast(lines)
    assert_same_results(got, want, s.histogram_dict(got.histogram), name + " wave kernel")


@pytest.mark.parametrize("name,n", [("c1", 250), ("c2", 250), ("c3", 250), ("c3c", 400), ("c4", 250), ("c5", 120)])
def test_wave_kernel_benchmark_corpora(name, n):
    c = make_corpus(name, n, seed=91)
    cfg = Config(**c.config_kwargs)
    o, s = Oracle(cfg), HostSim(cfg)
    want, got = o.run_lines(c.arena, c.offsets, threads=8), s.run_lines_fast(c.arena, c.offsets)
    assert_same_results(got, want, s.histogram_dict(got.histogram), name + " wave kernel")
    if name in ("c2", "c3", "c3c", "c4"):
        assert got.n_general <= n // 20, "the wave-per-file kernel should take nearly every file of %s (declined %d)" % (name, got.n_general)
    if name == "c3c":
        check_chains_are_walked(c, got)


# ---- the lean stage (tp_lean.h) in front of the wave stages, on an emulated wavefront ---------------------------------
@pytest.mark.parametrize("details", [False, True])
@pytest.mark.parametrize("name,n,least", [("c1", 300, 0.85), ("c2", 400, 0.95), ("c3", 300, 0.9), ("c3c", 300, 0.0), ("c4", 300, 0.8), ("c5", 150, 0.6)])
def test_lean_stage_benchmark_corpora(name, n, least, details):
    """The flow the library runs: lean stage, then the wave stages for what it declines, then the general kernel.  Without the
    detail arrays (details=False: fired / facet_choice null) the evidence set is not counted and more files stay lean."""
    c = make_corpus(name, n, seed=123)
    cfg = Config(**c.config_kwargs)
    o, s = Oracle(cfg), HostSim(cfg)
    want, got = o.run_lines(c.arena, c.offsets, threads=8), s.run_lines_lean(c.arena, c.offsets, details)
    if not details:
        skipped = got.verdicts["n_evidence"] == 0xFFFF          # only the emulation's general pipeline counts it regardless
        assert int(skipped.sum()) >= n - got.n_general
        got.verdicts["n_evidence"][skipped] = want.verdicts["n_evidence"][skipped]
        got.fired = want.fired
    assert_same_results(got, want, s.histogram_dict(got.histogram), "%s lean stage" % name)
    if not details:
        assert got.n_lean >= least * n, "the lean stage should take most files of %s (took %d of %d)" % (name, got.n_lean, n)
    assert got.n_general <= n // 10


def test_lean_stage_edge_fuzz_and_spellings():
    for name in sorted(CONFIGS):
        cfg = CONFIGS[name]()
        o, s = Oracle(cfg), HostSim(cfg)
        lines = EDGE_LINES + fuzz_lines(29, 300)
        want, got = o.run_texts(lines), s.run_texts_lean(lines)
        assert_same_results(got, want, s.histogram_dict(got.histogram), name + " lean stage")
    for lines, cfg in (spelling_variants(200, seed=8), multi_line_texts(200, seed=9)):
        o, s = Oracle(cfg), HostSim(cfg)
        want, got = o.run_texts(lines), s.run_texts_lean(lines)
        assert_same_results(got, want, s.histogram_dict(got.histogram), "respelled lines, lean stage")


def spelling_variants(n, seed=3):
    """c2 lines respelled the ways real tag files are: blanks for underscores, capitals, tabs, runs of blanks and underscores,
    quotes and full stops at the ends of tags, blanks around them."""
    c = make_corpus("c2", n, seed=seed)
    a = bytes(c.arena)
    base = [a[int(c.offsets[i]):int(c.offsets[i + 1])].rstrip(b"\n").decode() for i in range(n)]
    rng = random.Random(seed)

    def respell(tag):
        r = rng.random()
        if r < 0.25: return tag.replace("_", " ")
        if r < 0.40: return tag.title()
        if r < 0.50: return tag.replace("_", " ").upper()
        if r < 0.55: return tag.replace("_", "  ")
        if r < 0.60: return tag.replace("_", "\t")
        if r < 0.65: return tag.replace("_", " _")
        if r < 0.70: return "'" + tag.replace("_", " ") + "'"
        if r < 0.75: return tag.replace("_", " ") + rng.choice([".", ";", "..", " ."])
        if r < 0.80: return "  " + tag.replace("_", " ") + " "
        if r < 0.83: return "_" + tag
        if r < 0.86: return tag + "_"
        return tag
    return [", ".join(respell(t) for t in line.split(", ")) for line in base], Config(**c.config_kwargs)


def test_wave_kernel_common_spellings():
    lines, cfg = spelling_variants(300)
    o, s = Oracle(cfg), HostSim(cfg)
    want, got = o.run_texts(lines), s.run_texts_fast(lines)
    assert_same_results(got, want, s.histogram_dict(got.histogram), "spellings")
    assert got.n_general == 0


def multi_line_texts(n, seed=4):
    """c2 lines broken over several lines (line feeds after commas, inside tags, doubled), refusals split by a line feed, and
    texts broken the Windows way."""
    c = make_corpus("c2", n, seed=seed)
    a = bytes(c.arena)
    rng = random.Random(seed)
    out = []
    for i in range(n):
        toks = a[int(c.offsets[i]):int(c.offsets[i + 1])].rstrip(b"\n").decode().split(", ")
        r = rng.random()
        if r < 0.3: text = ",\n".join(toks)
        elif r < 0.5: text = ", ".join(t.replace("_", "\n", 1) for t in toks)
        elif r < 0.6: text = ",\n\n".join(toks[:10]) + "\n" + ", ".join(toks[10:])
        elif r < 0.7: text = "\n" + ",\n ".join(toks) + "\n\n"
        elif r < 0.8: text = ",\r\n".join(toks)
        elif r < 0.85: text = ", ".join(toks[:5]) + ", i\ncan't do that, " + ", ".join(toks[5:])
        elif r < 0.9: text = ", ".join(toks[:5]) + ", as\nan\nai, " + ", ".join(toks[5:])
        elif r < 0.95: text = "\n".join(toks[:2])
        else: text = toks[0] + "\n" + "x" * 45
        out.append(text)
    return out, Config(**c.config_kwargs)


def test_wave_kernel_multi_line_texts():
    lines, cfg = multi_line_texts(300)
    o, s = Oracle(cfg), HostSim(cfg)
    want, got = o.run_texts(lines), s.run_texts_fast(lines)
    assert_same_results(got, want, s.histogram_dict(got.histogram), "multi-line")
    assert got.n_general == 0                                   # "\r\n" inside is taken out of the LDS copy, line feeds are blanks
    assert (want.verdicts["flags"] & 0x10).any()                # the refusals split by a line feed are rejected


def odd_scores(kind):
    """scored_inputs with a few scores replaced: 'finite' keeps them finite but unusual, the others plant one bad value."""
    arena, offs, begin, scores, sources = scored_inputs(13, 300)
    scores = scores.copy()
    if kind == "finite":
        scores[::7] = [(-0.0, 5e-324, 1.5, -3.0, 1e308, 0.28, 0.6)[i % 7] for i in range(len(scores[::7]))]
    else:
        scores[len(scores) // 2] = {"nan": float("nan"), "inf": float("inf"), "-inf": float("-inf")}[kind]
    return arena, offs, begin, scores, sources


@pytest.mark.parametrize("kind", ["nan", "inf", "-inf"])
def test_non_finite_scores_are_refused(kind):
    """NaN and the infinities cannot come out of the reference's JSON ingest; both the library (TP_E_INVALID_ARG, GPU test of
    the same name) and the oracle refuse them instead of defining an order for them."""
    from oracle.oracle_py import OracleError
    arena, offs, begin, scores, sources = odd_scores(kind)
    o = Oracle(CONFIGS["shipped"]())
    with pytest.raises(OracleError) as e:
        o.run_scored(arena, offs, begin, scores, sources)
    assert "finite" in str(e.value)
    with pytest.raises(OracleError):
        o.run_merged(arena, offs, begin, begin[:-1].copy(), scores, 15)


def test_unusual_finite_scores():
    cfg = CONFIGS["shipped"]()
    args = odd_scores("finite")
    o, s = Oracle(cfg), HostSim(cfg)
    want = o.run_scored(*args)
    got = s.run_scored(*args)
    assert_same_results(got, want, s.histogram_dict(got.histogram), "finite scores")
    got = s.run_scored_fast(*args)
    assert_same_results(got, want, s.histogram_dict(got.histogram), "finite scores, wave kernel")


# ---- strings in no table, learned: the tables compiled again with them interned (what the library does between calls) ------
@pytest.mark.parametrize("strict_off_with_vocabulary", [False, True])
def test_learned_strings_change_nothing_but_the_path(strict_off_with_vocabulary):
    """c2 without its vocabulary file (or with it and strict vocabulary switched off): tags in no table are kept, and the wave
    stages hand a file that holds one to the general pipeline.  With the corpus's strings learned - interned behind the
    configuration's own, ids and histogram entries of the configuration's strings unchanged, a learned tag counted with
    the tags in no table - the results are the same and the lean stage takes the files."""
    n = 300
    c = make_corpus("c2", n, seed=77)
    kw = dict(c.config_kwargs)
    if strict_off_with_vocabulary:
        # half the vocabulary, strict vocabulary switched off explicitly: unknown tags are kept and counted (GetUnknownTags)
        vocab = kw["vocabulary"].decode().split("\n")
        kw["vocabulary"] = ("\n".join(vocab[::2]) + "\n").encode()
        kw.update(strict_known=False, strict_known_explicit=True)
    else:
        kw["vocabulary"] = None
    cfg = Config(**kw)
    want = Oracle(cfg).run_lines(c.arena, c.offsets, threads=8)
    plain = HostSim(cfg)
    before = plain.run_lines_lean(c.arena, c.offsets, details=False)
    assert before.n_lean < n // 10, "without learning nearly every file holds a tag in no table"
    learned = HostSim(cfg, learned=c.pool + ["Not_Plain", "red_hair"])     # strings the tables hold already are skipped
    assert learned.info.n_strings == plain.info.n_strings
    assert learned.strings() == plain.strings()
    got = learned.run_lines_lean(c.arena, c.offsets, details=False)
    skipped = got.verdicts["n_evidence"] == 0xFFFF
    got.verdicts["n_evidence"][skipped] = want.verdicts["n_evidence"][skipped]
    got.fired = want.fired
    assert_same_results(got, want, learned.histogram_dict(got.histogram), "c2 with learned strings")
    assert got.n_lean >= 0.9 * n and got.n_general <= n // 50
    assert got.histogram_other > 0                                        # (compared with the oracle's pooled count above)


def test_candidates_that_end_in_one_tag_stay_in_the_wave_stages():
    """c3c: alias chains inside the vocabulary, so AddOne's own NormalizeTag (Program.cs:693) moves tags, and now and then two
    candidates of a file end in one tag: the second AddOne fails (Program.cs:696) without using up its bucket's share.  The wave
    stages repeat the bucket fills without the candidate that fails first; no such file is left to the general pipeline."""
    c = make_corpus("c3c", 4000, seed=4242)
    cfg = Config(**c.config_kwargs)
    s = HostSim(cfg)
    got, want = s.run_lines_lean(c.arena, c.offsets, details=True), Oracle(cfg).run_lines(c.arena, c.offsets, threads=8)
    assert_same_results(got, want, s.histogram_dict(got.histogram), "c3c")
    assert got.n_general == 0


def test_action_tags_that_normalize_tag_changes_stay_in_the_lean_stage():
    """InferActionTags adds a rule's tag as the rule names it (Program.cs:738, 745); FormatCaptionSingleLine and GetUnknownTags
    normalise the final list again and keep first occurrences (Program.cs:972, 819).  The lean stage's detail build takes such
    files: the changed tag meets an earlier element (the built-in subtype "fellatio" behind its parent "blowjob"), a later one
    (a facet's default tag), another changed tag, or nothing."""
    from helpers import changed_action_case
    cfg, lines = changed_action_case()
    o, s = Oracle(cfg), HostSim(cfg)
    want, got = o.run_texts(lines), s.run_texts_lean(lines)
    assert_same_results(got, want, s.histogram_dict(got.histogram), "changed action tags")
    assert got.n_general == 0 and got.n_lean == len(lines)
    assert b"bj" not in want.caption(0).replace(b"blowjob", b"") and want.caption(0).count(b"blowjob") == 1
