"""Existing-caption triage (Program.cs:203-210): the kernel bodies in triage mode against the oracle, without a GPU."""
import numpy as np
import pytest

from helpers import Config, HostSim, Oracle, shipped_config, triage_lines
from test_pipeline_logic_cpu import CONFIGS, fuzz_lines, long_lines


def test_oracle_triage_known_answers():
    o = Oracle(shipped_config())
    texts = ["", "1girl, smile, dress", "sorry, I can't help with that", "1girl",
             "a_tag_name_that_is_longer_than_forty_units_in_total_yes",          # one tag, no comma, 55 units
             "looking_at_viewer, very_long_hair_1234567890123456",               # two tags, 17 + 2 + 33 = 52 units
             "smile, smile, smile"]
    got = list(o.triage_texts(texts))
    # blank -> bad; ordinary -> keep; refusal -> bad; short single tag -> keep
    assert got[:4] == [1, 0, 1, 0]
    # agreement with the two functions called one after the other
    for t, g in zip(texts, got):
        norm = o.normalize_raw_tag_line(t)
        commas = norm.count(",")
        want = (not norm.strip()) or (commas < 2 and len(norm) > 40)
        assert bool(g) == bool(want), (t, norm)


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_triage_matches_oracle(name):
    cfg = CONFIGS[name]()
    o, s = Oracle(cfg), HostSim(cfg)
    lines = triage_lines(5, 700) + fuzz_lines(31, 300)
    want = o.triage_texts(lines)
    got, n_general = s.triage_texts(lines)
    assert not (got == 2).any()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, [(lines[i], int(got[i]), int(want[i])) for i in bad[:5]]
    got_g, _ = s.triage_texts(lines, general_only=True)
    bad = np.nonzero(got_g != want)[0]
    assert bad.size == 0, [(lines[i], int(got_g[i]), int(want[i])) for i in bad[:5]]
    assert 0 < want.sum() < len(lines)


def test_triage_long_files():
    cfg = shipped_config()
    o, s = Oracle(cfg), HostSim(cfg)
    lines = long_lines(9, 30) + [", ".join(["smile"] * 300), ", ".join(["smile"] * 200 + ["looking_at_viewer"] * 200)]
    want = o.triage_texts(lines)
    got, n_general = s.triage_texts(lines)
    assert list(got) == list(want)
    assert n_general < len(lines) // 2


def test_triage_multi_line_texts():
    from test_pipeline_logic_cpu import multi_line_texts
    lines, cfg = multi_line_texts(300, seed=8)
    o, s = Oracle(cfg), HostSim(cfg)
    want = o.triage_texts(lines)
    got, n_general = s.triage_texts(lines)
    assert list(got) == list(want)
    assert n_general == 0 and 0 < want.sum() < len(lines)
