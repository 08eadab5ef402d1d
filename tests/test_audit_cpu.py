"""The audit row (AppendAudit, Program.cs:334, 958-969): the tag lists the general pipeline keeps in audit mode, checked
against the oracle's lists on the host build of the pipeline source, and the oracle's own row against hand-derived lines."""
import numpy as np
import pytest

from helpers import EDGE_LINES, HostSim, Oracle, oracle_audit_lists, pack_lines, shipped_config
from imagecaptioner_amd._abi import Config
from imagecaptioner_amd.corpus import make_corpus


def _check_lists(sim, orc, texts):
    arena, offsets = pack_lines(texts)
    want = orc.run_lines(arena, offsets, detail=True)
    lists, ver = sim.run_audit(arena, offsets)
    exp = oracle_audit_lists(want)
    for i, t in enumerate(texts):
        if ver["flags"][i] & 0x20:
            continue                                   # outside what the device code takes: no lists
        assert lists[i][:3] == exp[i], (t, lists[i], exp[i])
        assert ver[i].tobytes() == want.verdicts[i].tobytes(), t


def test_audit_lists_edge_lines():
    cfg = shipped_config()
    _check_lists(HostSim(cfg), Oracle(cfg), EDGE_LINES)


def test_audit_lists_dropped_and_invalid():
    cfg = shipped_config(denylist=b"smile\nhat\n", allowlist=b"dress\n")
    texts = ["1girl, smile, :bad, a!b, rating:safe, watermark, dress, hat, blue_hair, SMILE, sorry_tag",
             "x'., y';, 1girl, (paren), -bad, smile, smile",
             "zz_unknown_one, zz_unknown_two, 1girl, long_hair, dress, penis, feet"]
    _check_lists(HostSim(cfg), Oracle(cfg), texts)
    strict = shipped_config(strict_known=True, strict_known_explicit=True)
    _check_lists(HostSim(strict), Oracle(strict), texts)


@pytest.mark.parametrize("name", ["c1", "c3", "c5"])
def test_audit_lists_corpus(name):
    c = make_corpus(name, 300, seed=11)
    cfg = Config(**c.config_kwargs)
    arena, offsets = c.arena, c.offsets
    sim, orc = HostSim(cfg), Oracle(cfg)
    want = orc.run_lines(arena, offsets, detail=True)
    lists, ver = sim.run_audit(arena, offsets)
    exp = oracle_audit_lists(want)
    for i in range(len(exp)):
        assert not (ver["flags"][i] & 0x20)
        assert lists[i][:3] == exp[i], i
    # evidence and dropped counts of the verdict agree with the lists
    for i in range(len(exp)):
        assert ver["n_dropped"][i] == (lists[i][2].count(b";") + 1 if lists[i][2] else 0)
        assert ver["n_evidence"][i] == (lists[i][1].count(b";") + 1 if lists[i][1] else 0)


KNOWN_TAGS = ["1girl", "blue_hair", "blue_eyes", "dress", "standing", "classroom", ":bad", 'a"b', "watermark", "Blue Hair"]
KNOWN_SCORES = [0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.35, 0.3, 0.29, 0.2]
# Hand-derived from Program.cs on the shipped tables: SelectTags keeps the trigger, "1girl" and the five valid tags (tags=7);
# ":bad" and 'a"b' fail ValidTagRegex, "watermark" is a policy drop (Program.cs:630-631); evidence holds every tag scored 0.28 or
# more, dropped ones included (Program.cs:721-729); no facet rule matches, so each of the seven required facets adds and logs its
# default (Program.cs:760-772); every coverage category is met (Program.cs:228, 292-294); Csv doubles the quotes (Program.cs:976-981).
KNOWN_ROW = ('"pic ""1"".png",'
             '"sgrui;1girl;blue_hair;blue_eyes;dress;standing;classroom;unknown_pov;unknown_gaze;unknown_underwear;unknown_fluids;unknown_ejaculation;'
             'unknown_expression;unknown_mouth",'
             '"1girl;blue_hair;blue_eyes;dress;standing;classroom;:bad;a""b;watermark",'
             '"facet:viewpoint:unknown_pov:default;facet:gaze:unknown_gaze:default;facet:underwear:unknown_underwear:default;facet:fluids:unknown_fluids:default;'
             'facet:ejaculation:unknown_ejaculation:default;facet:expression:unknown_expression:default;facet:mouth_shape:unknown_mouth:default",'
             '":bad:invalid_tag;a""b:invalid_tag;watermark:policy_drop","","top_tags.txt","sgrui","coverage_satisfied",'
             '"pass0_joy missing= required_missing= details=identity=ok|hair=ok|eyes=ok|outfit=ok|pose_action=ok|background_scene=ok tags=7"')
KNOWN_AUDIT = {"names": ['pic "1".png'], "vocab_file": "top_tags.txt", "used_profile": "sgrui"}


def known_scored_input():
    arena, toffs = pack_lines(KNOWN_TAGS)
    return arena, toffs, np.array([0, len(KNOWN_TAGS)], dtype=np.uint64), np.array(KNOWN_SCORES), np.zeros(len(KNOWN_TAGS), np.uint8)


def test_oracle_audit_row_known_answer():
    orc = Oracle(shipped_config())
    res = orc.run_scored(*known_scored_input(), audit=KNOWN_AUDIT)
    assert res.audit_rows[0].decode("utf-8") == KNOWN_ROW
    # a line with nothing the categories ask for: everything but the identity is missing, the OpenAI passes are off
    arena, offsets = pack_lines(["cat"])
    res = orc.run_lines(arena, offsets, audit={"names": ["b.png"], "vocab_file": None, "used_profile": "none"})
    f = res.audit_rows[0].decode("utf-8").split('","')
    assert f[6] == "" and f[7] == "none"
    assert f[8] == "openai_disabled_or_missing_api_key"
    assert f[9].startswith("pass0_joy missing=hair|eyes|outfit|pose_action|background_scene required_missing=hair|eyes|outfit|pose_action details=identity=ok|hair=missing|")


def test_audit_lists_known_scored():
    sim = HostSim(shipped_config())
    arena, toffs, tb, scores, src = known_scored_input()
    lists, ver = sim.run_audit(arena, toffs, tag_begin=tb, scores=scores, sources=src)
    f = KNOWN_ROW[1:-1].split('","')
    assert [x.decode() for x in lists[0][:3]] == [f[1], f[2].replace('""', '"'), f[4].replace('""', '"')]
    assert ver["n_selected"][0] == 7


@pytest.mark.parametrize("cap", [15, 2])
def test_audit_lists_scored_and_merged(cap):
    from test_merged_cpu import merged_inputs
    for kw in (dict(), dict(strict_known=True, strict_known_explicit=True), dict(denylist=b"smile\nlong_hair\n")):
        cfg = shipped_config(**kw)
        sim, orc = HostSim(cfg), Oracle(cfg)
        arena, toffs, begin, mid, scores = merged_inputs(23, 150)
        want = orc.run_merged(arena, toffs, begin, mid, scores, cap, detail=True)
        lists, ver = sim.run_audit(arena, toffs, tag_begin=begin, secondary_begin=mid, scores=scores, secondary_cap=cap)
        exp = oracle_audit_lists(want)
        n_dropped = 0
        for i in range(len(exp)):
            assert not (ver["flags"][i] & 0x20)
            assert lists[i][:3] == exp[i], (i, lists[i], exp[i])
            assert ver[i].tobytes() == want.verdicts[i].tobytes()
            n_dropped += len(want.details[i]["dropped"])
        assert n_dropped > 0
        # one reply per file, sources given
        src = (np.arange(len(scores)) % 3).astype(np.uint8)
        want = orc.run_scored(arena, toffs, begin, scores, src, detail=True)
        lists, ver = sim.run_audit(arena, toffs, tag_begin=begin, scores=scores, sources=src)
        assert [l[:3] for l in lists] == oracle_audit_lists(want)


def test_audit_lists_of_replies_that_stay_unsorted():
    """A reply without scores keeps its order (Program.cs:487): the evidence set and the dropped tags follow that order."""
    from helpers import UNSORTED_REPLIES, UNSORTED_SECOND, unsorted_reply_lists
    cfg = shipped_config()
    sim, orc = HostSim(cfg), Oracle(cfg)
    d = sim.decode_joytag_replies(UNSORTED_REPLIES)
    assert list(d["status"]) == [2, 2, 0, 0] and list(d["has_score"]) == [0, 0, 0, 1]
    want, exp = unsorted_reply_lists(orc)
    lists, ver = sim.run_audit(d["arena"], d["tag_offsets"], tag_begin=d["tag_begin"], scores=d["scores"], sources=d["sources"], reply_flags=d["has_score"])
    assert [l[:3] for l in lists] == exp
    assert all(ver[i].tobytes() == want.verdicts[i].tobytes() for i in range(len(exp)))
    # sorted, the same files list the same tags in another order (and the same final tags)
    lists2, ver2 = sim.run_audit(d["arena"], d["tag_offsets"], tag_begin=d["tag_begin"], scores=d["scores"], sources=d["sources"])
    assert [l[0] for l in lists2] == [e[0] for e in exp] and ver2.tobytes() == ver.tobytes()
    assert lists2[0][1] != exp[0][1] and sorted(lists2[0][1].split(b";")) == sorted(exp[0][1].split(b";"))
    assert [l[:3] for l in lists2[2:]] == exp[2:]
    # two replies per file: the primary one keeps its order, the secondary one is sorted by LimitSecondaryTags in any case
    d = sim.decode_joytag_replies(UNSORTED_REPLIES, UNSORTED_SECOND)
    for cap in (15, 1):
        want, exp = unsorted_reply_lists(orc, secondary=True, cap=cap)
        lists, ver = sim.run_audit(d["arena"], d["tag_offsets"], tag_begin=d["tag_begin"], secondary_begin=d["secondary_begin"], scores=d["scores"],
                                   secondary_cap=cap, reply_flags=d["has_score"])
        assert [l[:3] for l in lists] == exp, cap
        assert all(ver[i].tobytes() == want.verdicts[i].tobytes() for i in range(len(exp)))


# ---- review records (BuildReviewRecord / AppendReviewLog, Program.cs:826-890) -----------------------------------------------
def review_args(n, vocab_file="top_tags.txt"):
    return {"names": ["%d.png" % i for i in range(n)], "vocab_file": vocab_file, "used_profile": "sgrui",
            "image_paths": ['C:\\data\\set "a"\\%d+x.png' % i for i in range(n)], "caption_paths": ["C:\\data\\set\\%d.txt" % i for i in range(n)],
            "thresholds": "primary=0.35;secondary=0.25"}


SUBJECT_LINES = [
    "1girl, smile, dress, long_hair",
    "1boy, 1girl, smile, dress",
    "2girls, smile, dress, long_hair, blue_eyes, standing, classroom",
    "multiple_girls, multiple_boys, smile",
    "3girls, 4boys, yaoi, male_focus, male, BOY, smile, dress",
    "zz_girls_club, smile, dress, long_hair",
    "zz_boys_team, dress, long_hair",
    "12girls, 7boy, 3girl, x2girls, 2girlsx",
    "multiple_zz, multiple_girls_x",
    "girl, solo, dress, blue_hair",
    "1girl, blue_hair, blue_eyes, dress, standing, classroom",
    "1girl, blue_hair, blue_eyes, dress, standing, classroom, zz_unknown_b, Zz_unknown_A, _zz, zz-y",
    "cat",
]


@pytest.mark.parametrize("kw", [dict(), dict(strict_known=False, strict_known_explicit=True), dict(vocabulary=None, profile=None)])
def test_review_facts_lines(kw):
    cfg = shipped_config(**kw)
    sim, orc = HostSim(cfg), Oracle(cfg)
    texts = SUBJECT_LINES + EDGE_LINES
    arena, offsets = pack_lines(texts)
    want = orc.run_lines(arena, offsets, detail=True, audit=review_args(len(texts)))
    lists, ver = sim.run_audit(arena, offsets)
    from helpers import check_review_facts
    keep = [i for i in range(len(texts)) if not (ver["flags"][i] & 0x20)]
    assert len(keep) == len(texts)
    check_review_facts(lists, sim.review_bits, ver, want)
    assert any(want.review_jsonl)


def test_review_facts_tagger_replies():
    from helpers import check_review_facts
    from test_merged_cpu import merged_inputs
    for kw in (dict(), dict(strict_known=False, strict_known_explicit=True)):
        cfg = shipped_config(**kw)
        sim, orc = HostSim(cfg), Oracle(cfg)
        arena, toffs, begin, mid, scores = merged_inputs(31, 200)
        n = len(begin) - 1
        want = orc.run_merged(arena, toffs, begin, mid, scores, 15, detail=True, audit=review_args(n))
        lists, ver = sim.run_audit(arena, toffs, tag_begin=begin, secondary_begin=mid, scores=scores, secondary_cap=15)
        check_review_facts(lists, sim.review_bits, ver, want)
        src = (np.arange(len(scores)) % 3).astype(np.uint8)
        want = orc.run_scored(arena, toffs, begin, scores, src, detail=True, audit=review_args(n))
        lists, ver = sim.run_audit(arena, toffs, tag_begin=begin, scores=scores, sources=src)
        check_review_facts(lists, sim.review_bits, ver, want)


# Hand-derived from Program.cs:826-890 on the shipped tables for the tagger reply below: "1boy" is banned (male_subject_token; with
# "2girls" also mixed_subject_tokens and multi_person); no outfit tag, so the caption is short and outfit is missing; "zz_b" and "Zz_A"
# normalise to "zz_b" / "zz_a", are valid and not in the vocabulary -- under the shipped (strict) vocabulary they are dropped; the
# unknown final tags are the trigger and the seven facet defaults, none of which top_tags.txt holds, in ordinal-ignore-case order
# (Program.cs:817-823); the top tags are the candidates by score, then order; System.Text.Json writes " as \u0022, + as \u002B
# and \ as \; Csv doubles the quotes.
REVIEW_TAGS = ["2girls", "1boy", "blue_hair", "blue_eyes", "standing", "classroom", "zz_b", "Zz_A"]
REVIEW_SCORES = [0.9, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.95]
REVIEW_JSON = ('{"ImagePath":"C:\\\\data\\\\set \\u0022a\\u0022\\\\0\\u002Bx.png","CaptionPath":"C:\\\\data\\\\set\\\\0.txt",'
               '"Reasons":["male_subject_token","mixed_subject_tokens","multi_person","short_caption"],"BannedTokensDetected":["1boy"],"DroppedTags":["1boy"],'
               '"FinalTagCount":7,"ShortCaption":true,"ThresholdsUsed":"primary=0.35;secondary=0.25",'
               '"TopJoyTags":["zz_a","2girls","1boy","blue_hair","blue_eyes","standing","classroom","zz_b"],'
               '"UnknownFinalTags":["sgrui","unknown_ejaculation","unknown_expression","unknown_fluids","unknown_gaze","unknown_mouth","unknown_pov","unknown_underwear"],'
               '"KnownTagDictionaryFound":true,"KnownTagDictionaryCount":5811,"KnownTagDictionaryPath":"top_tags.txt",'
               '"CoverageTrace":["pass0_joy missing=outfit required_missing=outfit details=identity=ok|hair=ok|eyes=ok|outfit=missing|pose_action=ok|background_scene=ok tags=7"],'
               '"CoverageStopReason":"openai_disabled_or_missing_api_key"}')
REVIEW_CSV = ('"C:\\data\\set ""a""\\0+x.png","C:\\data\\set\\0.txt","male_subject_token;mixed_subject_tokens;multi_person;short_caption","1boy","1boy","7","true",'
              '"primary=0.35;secondary=0.25","zz_a;2girls;1boy;blue_hair;blue_eyes;standing;classroom;zz_b",'
              '"sgrui;unknown_ejaculation;unknown_expression;unknown_fluids;unknown_gaze;unknown_mouth;unknown_pov;unknown_underwear","true","5811","top_tags.txt",'
              '"openai_disabled_or_missing_api_key",'
              '"pass0_joy missing=outfit required_missing=outfit details=identity=ok|hair=ok|eyes=ok|outfit=missing|pose_action=ok|background_scene=ok tags=7"')


def review_scored_input():
    arena, toffs = pack_lines(REVIEW_TAGS)
    return arena, toffs, np.array([0, len(REVIEW_TAGS)], dtype=np.uint64), np.array(REVIEW_SCORES), np.zeros(len(REVIEW_TAGS), np.uint8)


def test_oracle_review_row_known_answer():
    orc = Oracle(shipped_config())
    res = orc.run_scored(*review_scored_input(), audit=review_args(1))
    assert res.review_jsonl[0].decode("utf-8") == REVIEW_JSON
    assert res.review_csv[0].decode("utf-8") == REVIEW_CSV
    # nothing to report (no vocabulary file, so no unknown tags either): no record (Program.cs:845).  "1girl" in the reply would
    # count as multi_person: it matches ^\\d+(girl|girls|boy|boys)$ (Program.cs:837)
    arena, offsets = pack_lines(["girl, blue_hair, blue_eyes, dress, standing, classroom"])
    res = Oracle(shipped_config(vocabulary=None)).run_lines(arena, offsets, audit=review_args(1, vocab_file=None))
    assert res.review_jsonl == [b""] and res.review_csv == [b""]


def test_general_kernel_takes_text_outside_ascii():
    """Audit mode runs every file through the thread-per-file code, which carries the Unicode half of NormalizeTag
    (Program.cs:1150-1154): the audit lists of such lines match the oracle's too."""
    from helpers import NON_ASCII_LINES, assert_same_results, check_non_ascii, check_review_facts, non_ascii_config
    cfg = non_ascii_config()
    sim, orc = HostSim(cfg), Oracle(cfg)
    check_non_ascii(sim.run_lines, orc, sim.histogram_dict)
    arena, offsets = pack_lines(NON_ASCII_LINES)
    want = orc.run_lines(arena, offsets, detail=True, audit=review_args(len(NON_ASCII_LINES)))
    lists, ver = sim.run_audit(arena, offsets)
    assert [l[:3] for l in lists] == oracle_audit_lists(want)
    check_review_facts(lists, sim.review_bits, ver, want)
    # a line of one long tag: LooksBad counts UTF-16 units, fewer than the bytes (Program.cs:1146)
    long_one = "青" * 20
    cfg = shipped_config(vocabulary=(long_one + "\n").encode("utf-8"), vocabulary_is_json=0)
    got, want = HostSim(cfg).run_texts([long_one, "1girl, smile, dress", long_one * 3]), Oracle(cfg).run_texts([long_one, "1girl, smile, dress", long_one * 3])
    assert_same_results(got, want, HostSim(cfg).histogram_dict(got.histogram), "long tag outside ASCII")
    assert not (got.verdicts["flags"] & 0x20).any() and (got.verdicts["flags"][2] & 0x10)


def test_multi_person_digits_are_unicode_digits():
    """^\\d+(girl|girls|boy|boys)$ (Program.cs:837): \\d matches every Unicode decimal digit.  Such a tag fails ValidTagRegex, so only a
    tagger reply can bring it among the raw candidates (line ingest drops it before, Program.cs:600)."""
    from helpers import check_review_facts
    import json
    tag = "\u0662girls"                                       # ARABIC-INDIC DIGIT TWO
    cfg = shipped_config(vocabulary=("1girl\nsmile\ndress\nlong_hair\n%s\nx%s\n" % (tag, tag)).encode("utf-8"), vocabulary_is_json=0)
    sim, orc = HostSim(cfg), Oracle(cfg)
    tags = [tag, "smile", "dress", "long_hair", "x" + tag, "smile", "dress", "long_hair"]
    arena, toffs = pack_lines(tags)
    begin = np.array([0, 4, 8], dtype=np.uint64)
    scores = np.array([0.9, 0.8, 0.7, 0.6] * 2)
    src = np.zeros(8, np.uint8)
    want = orc.run_scored(arena, toffs, begin, scores, src, detail=True, audit=review_args(2))
    lists, ver = sim.run_audit(arena, toffs, tag_begin=begin, scores=scores, sources=src)
    assert not (ver["flags"] & 0x20).any()
    assert [l[:3] for l in lists] == oracle_audit_lists(want)
    check_review_facts(lists, sim.review_bits, ver, want)
    assert "multi_person" in json.loads(want.review_jsonl[0])["Reasons"] and "multi_person" not in json.loads(want.review_jsonl[1])["Reasons"]
    assert lists[0][2] == (tag + ":invalid_tag").encode("utf-8")


def unicode_reply_tags():
    """Tagger-reply tags with characters outside ASCII that are in no table: the scored ingest keeps them as candidates
    (Program.cs:473-487), SelectTags drops them as invalid, and the audit row's dropped list spells them as NormalizeTag left them."""
    files = [
        ["1girl", "CAFÉ AU　LAIT", "smile", "２Girls", "dress", "café au lait。", "long_hair"],
        ["ÜBER TAG", "1girl", "٣boys.", "smile", "dress", "\U0001f600 Face", "Kelvin", "x、"],
        ["smile", "　", "  ", "dress", "1girl", "青", "青い"],
    ]
    tags = [t for f in files for t in f]
    arena, toffs = pack_lines(tags)
    begin = np.cumsum([0] + [len(f) for f in files]).astype(np.uint64)
    scores = np.array([round(0.95 - 0.03 * (i % 9), 2) for i in range(len(tags))])
    return arena, toffs, begin, scores, np.zeros(len(tags), np.uint8)


def test_tagger_reply_tags_outside_ascii():
    from helpers import assert_same_results, check_review_facts, non_ascii_config
    cfg = non_ascii_config()
    sim, orc = HostSim(cfg), Oracle(cfg)
    arena, toffs, begin, scores, src = unicode_reply_tags()
    want = orc.run_scored(arena, toffs, begin, scores, src, detail=True, audit=review_args(len(begin) - 1))
    got = sim.run_scored(arena, toffs, begin, scores, src)
    assert got.counters["unsupported"] == 0
    assert_same_results(got, want, sim.histogram_dict(got.histogram), "reply tags outside ASCII")
    got = sim.run_scored_fast(arena, toffs, begin, scores, src)
    assert_same_results(got, want, sim.histogram_dict(got.histogram), "reply tags outside ASCII, wave kernel first")
    lists, ver = sim.run_audit(arena, toffs, tag_begin=begin, scores=scores, sources=src)
    assert [l[:3] for l in lists] == oracle_audit_lists(want)
    check_review_facts(lists, sim.review_bits, ver, want)
    assert "café_au_lait:invalid_tag".encode("utf-8") in lists[0][2] and "２girls:invalid_tag".encode("utf-8") in lists[0][2]
    # a character that shares its capital with another one: two such spellings would be one candidate to the reference
    arena, toffs = pack_lines(["1girl", "µm", "smile"])
    r = sim.run_scored(arena, toffs, np.array([0, 3], dtype=np.uint64), np.array([0.9, 0.8, 0.7]), np.zeros(3, np.uint8))
    assert r.verdicts["flags"][0] & 0x20


@pytest.mark.parametrize("name,n", [("c3c", 300), ("c2", 300), ("c1", 300), ("c5", 200), ("edge", 0)])
def test_audit_lists_of_the_wave_stages(name, n):
    """Audit mode in the wave stages (the kernels' code on an emulated wavefront): the lists a stage leaves as interned ids - final
    tags, evidence in insertion order, dropped tags with reason, top tagger tags, unknown caption tags - and the review bits equal
    the oracle's for every file a stage takes, and the stages take nearly every file (c3c: the general wave stage lists closures of
    several ExpandImpliedTags rounds; c5: files of more than 128 tags are the long-file stage's)."""
    from helpers import EDGE_LINES, Config, check_review_facts
    from test_pipeline_logic_cpu import fuzz_lines
    if name == "edge":
        cfg = shipped_config()
        arena, offsets = pack_lines(EDGE_LINES + fuzz_lines(17, 400))
    else:
        from imagecaptioner_amd.corpus import make_corpus
        c = make_corpus(name, n, seed=616)
        cfg = Config(**c.config_kwargs)
        arena, offsets = c.arena, c.offsets
    m = len(offsets) - 1
    sim, orc = HostSim(cfg), Oracle(cfg)
    want = orc.run_lines(arena, offsets, detail=True, audit=review_args(m))
    lists, ver, stage = sim.run_audit_wave(arena, offsets)
    exp = oracle_audit_lists(want)
    taken = [i for i in range(m) if stage[i]]
    assert len(taken) >= (m * 9 // 10 if name != "edge" else m // 2), (len(taken), m)
    if name in ("c3c", "c1"):
        assert sum(1 for i in taken if stage[i] >= 2) > (m // 3 if name == "c3c" else 5)      # the general wave stage had work to do
    if name == "c5":
        assert sum(1 for i in taken if stage[i] >= 4) >= 3                                    # ... and the long-file stage (more than 128 tags)
    for i in taken:
        assert lists[i][:3] == exp[i], (i, lists[i][:3], exp[i])
        assert ver[i].tobytes() == want.verdicts[i].tobytes(), i
    sub = lambda seq: [seq[i] for i in taken]
    class Sub:                                                       # the oracle's records of the files a stage took
        review_jsonl = sub(want.review_jsonl)
    check_review_facts(sub(lists), sim.review_bits[taken], ver[taken], Sub)


def sorted_replies(begin, scores, arena, toffs):
    """Each file's entries in ParseJoyTags' order for a reply with scores (score descending, then as they came: Program.cs:487)."""
    tags = [arena[int(toffs[i]):int(toffs[i + 1])].tobytes() for i in range(len(toffs) - 1)]
    perm = []
    for f in range(len(begin) - 1):
        lo, hi = int(begin[f]), int(begin[f + 1])
        perm += sorted(range(lo, hi), key=lambda e: (-scores[e], e))
    enc = [tags[e] for e in perm]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offs[1:] = np.cumsum([len(b) for b in enc])
    return np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8)[:-1].copy(), offs, scores[perm]


@pytest.mark.parametrize("kw", [dict(), dict(strict_known=True, strict_known_explicit=True), dict(denylist=b"smile\nlong_hair\n")])
def test_audit_lists_of_the_scored_wave_kernel(kw):
    """Tagger replies in audit mode in the wave kernel (its code on an emulated wavefront): final tags, evidence, dropped tags with
    reason, the top tagger tags (entries of one string apart in a single reply, ties between the two replies' ranks) and the review
    bits equal the oracle's for every file the kernel takes, whether a single reply comes in score order or not (ParseJoyTags sorts
    it, Program.cs:487)."""
    from helpers import check_review_facts
    from test_merged_cpu import merged_inputs
    cfg = shipped_config(**kw)
    sim, orc = HostSim(cfg), Oracle(cfg)
    arena, toffs, begin, mid, scores = merged_inputs(29, 160)
    m = len(begin) - 1
    sarena, stoffs, sscores = sorted_replies(begin, scores, arena, toffs)
    src = np.zeros(len(scores), np.uint8)
    cases = [("merged15", orc.run_merged(arena, toffs, begin, mid, scores, 15, detail=True, audit=review_args(m)), dict(secondary_begin=mid, secondary_cap=15), (arena, toffs, scores)),
             ("merged2", orc.run_merged(arena, toffs, begin, mid, scores, 2, detail=True, audit=review_args(m)), dict(secondary_begin=mid, secondary_cap=2), (arena, toffs, scores)),
             ("as they come", orc.run_scored(arena, toffs, begin, scores, src, detail=True, audit=review_args(m)), dict(sources=src), (arena, toffs, scores)),
             ("sorted", orc.run_scored(sarena, stoffs, begin, sscores, src, detail=True, audit=review_args(m)), dict(sources=src), (sarena, stoffs, sscores))]
    for what, want, extra, (a, o, s) in cases:
        lists, ver, stage = sim.run_audit_scored_wave(a, o, begin, s, **extra)
        exp = oracle_audit_lists(want)
        taken = [i for i in range(m) if stage[i]]
        assert len(taken) >= m * 6 // 10, (what, len(taken), m)
        for i in taken:
            assert lists[i][:3] == exp[i], (what, i, lists[i][:3], exp[i])
            assert ver[i].tobytes() == want.verdicts[i].tobytes(), (what, i)
        sub = lambda seq: [seq[i] for i in taken]
        class Sub:
            review_jsonl = sub(want.review_jsonl)
        check_review_facts(sub(lists), sim.review_bits[taken], ver[taken], Sub)
