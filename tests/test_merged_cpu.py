"""Two tagger replies per file (primary and secondary threshold) merged as Program.cs:214-221 does: the general kernel body
against the oracle, without a GPU."""
import random

import numpy as np
import pytest

from helpers import DATA, HostSim, Oracle, assert_same_results
from test_pipeline_logic_cpu import CONFIGS


def merged_inputs(seed, n_files):
    """(arena, tag_offsets, tag_begin, secondary_begin, scores).  The secondary reply repeats most of the primary one (a lower
    threshold on the same image) with equal, close (<= 0.03) and distant scores, adds new tags beyond any cap, and both
    replies hold duplicates after normalisation, score ties and blanks."""
    rng = random.Random(seed)
    vocab = [l.strip() for l in open(DATA + "/top_tags.txt") if l.strip()][:500] + ["foot", "fingers", "Pedal Stimulation", "zz_unknown", "x'.", "", " ", "a",
                                                                                      "feet", "penis", "footjob", "soles", "toes", "barefoot"]
    tags, begin, mid, scores = [], [0], [], []
    for _ in range(n_files):
        kp = rng.choice([0, 1, 3, 10, 30, 60])
        prim = [(rng.choice(vocab), rng.choice([round(rng.random(), 2), 0.5, 0.6, 0.28, 0.47])) for _ in range(kp)]
        if prim and rng.random() < 0.3:
            prim.append((prim[0][0].upper(), prim[0][1] + rng.choice([0.0, 0.02, 0.04, -0.02])))   # duplicate after normalisation
        sec = []
        if rng.random() < 0.85:
            for t, s in prim:
                if rng.random() < 0.8:
                    sec.append((t, s + rng.choice([0.0, 0.0, 0.01, -0.01, 0.03, -0.03, 0.031, 0.05, -0.2, 0.3])))
            for _ in range(rng.choice([0, 2, 10, 25])):
                sec.append((rng.choice(vocab), rng.choice([round(0.35 * rng.random(), 2), 0.3, 0.5])))
            rng.shuffle(sec)
        for t, s in prim:
            tags.append(t); scores.append(s)
        mid.append(len(tags))
        for t, s in sec:
            tags.append(t); scores.append(s)
        begin.append(len(tags))
    enc = [t.encode() for t in tags]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offs[1:] = np.cumsum([len(b) for b in enc])
    arena = np.frombuffer(b"".join(enc), dtype=np.uint8).copy() if enc else np.zeros(0, np.uint8)
    return arena, offs, np.array(begin, dtype=np.uint64), np.array(mid, dtype=np.uint64), np.array(scores, dtype=np.float64)


def test_oracle_merge_known_answers():
    """Hand-derived: cap, primary-set pass-through, score/order/source of a merged entry."""
    from helpers import shipped_config
    o = Oracle(shipped_config(safe_mode=False))
    #            primary                  secondary
    tags = ["1girl", "smile", "dress",   "smile", "long_hair", "blush", "hat"]
    scores = [0.9, 0.5, 0.4,             0.52, 0.34, 0.33, 0.32]
    enc = [t.encode() for t in tags]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(b) for b in enc])
    arena = np.frombuffer(b"".join(enc), dtype=np.uint8).copy()
    begin, mid = np.array([0, 7], dtype=np.uint64), np.array([3], dtype=np.uint64)
    cap2 = o.run_merged(arena, offs, begin, mid, scores, 2).caption(0).decode()
    cap0 = o.run_merged(arena, offs, begin, mid, scores, 0).caption(0).decode()
    only_primary = o.run_scored(arena[:int(offs[3])], offs[:4], np.array([0, 3], dtype=np.uint64), scores[:3], np.zeros(3, np.uint8)).caption(0).decode()
    assert "long_hair" in cap2 and "blush" in cap2 and "hat" not in cap2          # two new tags, best scores first
    assert cap0 == only_primary                                                  # cap <= 0: the secondary reply is ignored altogether
    assert "hat" in o.run_merged(arena, offs, begin, mid, scores, 15).caption(0).decode()


@pytest.mark.parametrize("name", ["shipped", "shipped_nonstrict", "builtin"])
@pytest.mark.parametrize("cap", [15, 3, 0])
def test_merged_ingest(name, cap):
    cfg = CONFIGS[name]()
    o, s = Oracle(cfg), HostSim(cfg)
    args = merged_inputs(17 + cap, 400)
    want, got = o.run_merged(*args, cap), s.run_merged(*args, cap)
    assert_same_results(got, want, s.histogram_dict(got.histogram), "%s merged cap %d" % (name, cap))


def clean_replies(seed, n_files, merged):
    """Replies as a tagger sends them: known tags, every tag once per reply, three-digit scores (so ties are rare but present)."""
    rng = random.Random(seed)
    vocab = [l.strip() for l in open(DATA + "/top_tags.txt") if l.strip()][:1500]
    tags, begin, mid, scores = [], [0], [], []
    for _ in range(n_files):
        k = rng.choice([0, 1, 2, 5, 20, 40, 60, 90])
        prim = rng.sample(vocab, k)
        ps = sorted((round(0.4 + 0.6 * rng.random(), 3) for _ in prim), reverse=True)
        for t, s in zip(prim, ps):
            tags.append(t.replace("_", " ") if rng.random() < 0.2 else t); scores.append(s)
        mid.append(len(tags))
        if merged:
            extra = rng.sample(vocab, rng.choice([0, 3, 10, 25]))
            sec = [(t, s) for t, s in zip(prim, ps)] + [(t, round(0.35 + 0.05 * rng.random(), 3)) for t in extra if t not in prim]
            sec.sort(key=lambda x: -x[1])
            for t, s in sec:
                tags.append(t); scores.append(s)
        begin.append(len(tags))
    enc = [t.encode() for t in tags]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offs[1:] = np.cumsum([len(b) for b in enc])
    arena = np.frombuffer(b"".join(enc), dtype=np.uint8).copy() if enc else np.zeros(0, np.uint8)
    return arena, offs, np.array(begin, dtype=np.uint64), np.array(mid, dtype=np.uint64), np.array(scores, dtype=np.float64)


@pytest.mark.parametrize("name", ["shipped", "shipped_nonstrict", "builtin"])
def test_wave_kernel_scored_replies(name):
    """One reply per file through the wave kernel's scored front (general pipeline for what it declines)."""
    from test_pipeline_logic_cpu import scored_inputs
    cfg = CONFIGS[name]()
    o, s = Oracle(cfg), HostSim(cfg)
    arena, offs, begin, _, scores = clean_replies(3, 300, False)
    sources = np.zeros(len(scores), np.uint8)
    want, got = o.run_scored(arena, offs, begin, scores, sources), s.run_scored_fast(arena, offs, begin, scores, sources)
    assert_same_results(got, want, s.histogram_dict(got.histogram), name + " clean replies")
    if name == "shipped":
        assert got.n_general < 30, got.n_general
    args = scored_inputs(5, 300)                                  # unknown tags, duplicates, blanks, mixed sources: mostly declined, still equal
    want, got = o.run_scored(*args), s.run_scored_fast(args[0], args[1], args[2], args[3], args[4])
    assert_same_results(got, want, s.histogram_dict(got.histogram), name + " messy replies")


@pytest.mark.parametrize("name", ["shipped", "builtin"])
@pytest.mark.parametrize("cap", [15, 3, 0])
def test_wave_kernel_merged_replies(name, cap):
    cfg = CONFIGS[name]()
    o, s = Oracle(cfg), HostSim(cfg)
    arena, offs, begin, mid, scores = clean_replies(11 + cap, 300, True)
    want = o.run_merged(arena, offs, begin, mid, scores, cap)
    got = s.run_scored_fast(arena, offs, begin, scores, None, mid, cap)
    assert_same_results(got, want, s.histogram_dict(got.histogram), "%s clean merged cap %d" % (name, cap))
    if name == "shipped":
        assert got.n_general < 100, got.n_general               # two replies of 60 or 90 tags do not fit 128 candidates
    arena, offs, begin, mid, scores = merged_inputs(23 + cap, 300)
    want = o.run_merged(arena, offs, begin, mid, scores, cap)
    got = s.run_scored_fast(arena, offs, begin, scores, None, mid, cap)
    assert_same_results(got, want, s.histogram_dict(got.histogram), "%s messy merged cap %d" % (name, cap))


def test_wave_kernel_joins_repeated_tags():
    """A tag twice in one reply (as written, or as two spellings of one normalised string): one candidate with the best score and
    the earliest order, counted twice in finalCandidates; twice in the secondary reply it takes two places of the cap."""
    from helpers import shipped_config
    cfg = shipped_config()
    o, s = Oracle(cfg), HostSim(cfg)
    rng = random.Random(2)
    vocab = [l.strip() for l in open(DATA + "/top_tags.txt") if l.strip()][:800]
    tags, begin, mid, scores = [], [0], [], []
    for _ in range(300):
        prim = rng.sample(vocab, rng.choice([3, 10, 30]))
        ps = [round(0.4 + 0.6 * rng.random(), 3) for _ in prim]
        for t, sc in list(zip(prim, ps))[:rng.choice([1, 2, 4])]:
            prim.append(rng.choice([t, t.replace("_", " "), t.upper()])); ps.append(round(sc + rng.choice([-0.2, -0.01, 0.0, 0.02, 0.3]), 3))
        tags += prim; scores += ps
        mid.append(len(tags))
        sec = rng.sample(vocab, 12)
        sec += [sec[0], sec[1].replace("_", " "), prim[0]]
        tags += sec; scores += [round(0.3 + 0.2 * rng.random(), 3) for _ in sec]
        begin.append(len(tags))
    enc = [t.encode() for t in tags]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(b) for b in enc])
    arena = np.frombuffer(b"".join(enc), dtype=np.uint8).copy()
    begin, mid, scores = np.array(begin, dtype=np.uint64), np.array(mid, dtype=np.uint64), np.array(scores, dtype=np.float64)
    src = np.zeros(len(scores), np.uint8)
    want, got = o.run_scored(arena, offs, begin, scores, src), s.run_scored_fast(arena, offs, begin, scores, src)
    assert_same_results(got, want, s.histogram_dict(got.histogram), "repeated tags, one reply")
    assert got.n_general < 60, got.n_general
    for cap in (15, 5):
        want, got = o.run_merged(arena, offs, begin, mid, scores, cap), s.run_scored_fast(arena, offs, begin, scores, None, mid, cap)
        assert_same_results(got, want, s.histogram_dict(got.histogram), "repeated tags, merged cap %d" % cap)
        assert got.n_general < 60, got.n_general


def test_wave_kernel_counts_tags_in_no_table():
    """Under strict vocabulary a reply's tags that are in no table are dropped by SelectTags but count as candidates, as dropped
    tags, as evidence and (in the style bucket, score >= 0.60) towards the style cap: the scored front tells them apart by bytes."""
    from helpers import shipped_config
    cfg = shipped_config()
    o, s = Oracle(cfg), HostSim(cfg)
    rng = random.Random(6)
    vocab = [l.strip() for l in open(DATA + "/top_tags.txt") if l.strip()][:800]
    odd = [";)", ";d", "zz_unknown", "ZZ Unknown", "zz_unknown", "best_quality_zz", "masterpiece_zz", "highres_zz", "absurdres_zz", "some thing new", "x!y", "a"]
    tags, begin, mid, scores = [], [0], [], []
    for _ in range(300):
        prim = rng.sample(vocab, rng.choice([3, 10, 30])) + rng.sample(odd, rng.choice([1, 2, 4]))
        rng.shuffle(prim)
        tags += prim; scores += [round(0.2 + 0.8 * rng.random(), 3) for _ in prim]
        mid.append(len(tags))
        sec = rng.sample(vocab, 8) + rng.sample(odd, rng.choice([0, 2, 5])) + prim[:3]
        tags += sec; scores += [round(0.2 + 0.5 * rng.random(), 3) for _ in sec]
        begin.append(len(tags))
    enc = [t.encode() for t in tags]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(b) for b in enc])
    arena = np.frombuffer(b"".join(enc), dtype=np.uint8).copy()
    begin, mid, scores = np.array(begin, dtype=np.uint64), np.array(mid, dtype=np.uint64), np.array(scores, dtype=np.float64)
    src = np.zeros(len(scores), np.uint8)
    want, got = o.run_scored(arena, offs, begin, scores, src), s.run_scored_fast(arena, offs, begin, scores, src)
    assert_same_results(got, want, s.histogram_dict(got.histogram), "tags in no table, one reply")
    assert got.n_general < 30, got.n_general
    assert (want.verdicts["n_dropped"] > 0).mean() > 0.9
    for cap in (15, 2):
        want, got = o.run_merged(arena, offs, begin, mid, scores, cap), s.run_scored_fast(arena, offs, begin, scores, None, mid, cap)
        assert_same_results(got, want, s.histogram_dict(got.histogram), "tags in no table, merged cap %d" % cap)
        assert got.n_general < 30, got.n_general


def test_cached_tagger_replies_decode_like_parse_joy_tags():
    """tp_decode_joytag_replies' decoder (host build) and the scored / merged runs against the oracle reading the same JSON bodies."""
    from helpers import HostSim, Oracle, check_joytag_replies, shipped_config
    for cfg in (shipped_config(), shipped_config(strict_known=False, strict_known_explicit=True, lexicon=b'{"aliases": {"": "smile", "zz_unknown": " "}}')):
        check_joytag_replies(HostSim(cfg), Oracle(cfg))
