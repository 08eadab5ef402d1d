"""The table compiler (imagecaptioner_amd/csrc/tp_tables.cpp, as built into the host emulation) against the oracle's loaders
on table files in every accepted shape; see loader_shapes.py for the shapes and the reference lines they follow."""
import pytest

from helpers import HostSim, Oracle, assert_same_results
from loader_shapes import REJECTED, SHAPES, shape_lines


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_loader_shapes_match_the_oracle(name):
    cfg = SHAPES[name]()
    o, s = Oracle(cfg), HostSim(cfg)
    lines = shape_lines()
    want = o.run_texts(lines)
    for got in (s.run_texts(lines), s.run_texts_lean(lines)):
        assert_same_results(got, want, s.histogram_dict(got.histogram), name)
    if name.startswith("force_allowed_commas"):
        # the pieces of a cut-up tag are candidates: "ab" alone gives "a" and "_b" (the second is no valid tag)
        i = lines.index("ab")
        assert want.verdicts["n_candidates"][i] == 2 and want.verdicts["n_dropped"][i] >= 1
        i = lines.index("xy, gone")
        assert want.verdicts["n_candidates"][i] == 2                       # "x", "y"; ",," has no pieces
        # existing-caption triage reads the joined line, commas inside a kept tag included (Program.cs:1145-1146)
        texts = lines + ["xy, " + "z" * 40, "x, " + "z" * 40, "gone, " + "z" * 40, "ab" + " " * 50 + ", smile"]
        bad, _ = s.triage_texts(texts)
        assert list(bad) == list(o.triage_texts(texts))


def test_the_shapes_do_what_the_reference_says():
    """Known answers on the oracle, so the parity above is not two loaders ignoring the same file."""
    o = Oracle(SHAPES["lexicon_alt_keys_open"]())
    line = o.select_tags_from_line
    assert "barefoot" in line("bare foot") and "long_hair" in line("overwritten") and "smile" in line("grin")      # aliases, normalization
    assert "looking_at_viewer" in line("eye contact") and "looking_at_viewer" in line("staring") and "dress" in line("frock")   # synonyms
    assert "not_an_array" in line("not_an_array") and "skipped" in line("skipped")
    names = Oracle(SHAPES["actions_category_map"]()).rule_names()["actions"]
    assert names == ["footjob", "pedal_play", "handjob"]
    assert Oracle(SHAPES["actions_single_object"]()).rule_names()["actions"] == ["foot_job"]
    assert Oracle(SHAPES["actions_array_mixed"]()).rule_names()["actions"] == ["footjob", "hand_job"]
    facets = Oracle(SHAPES["facets_array_shape"]()).rule_names()["facets"]
    assert [f["name"] for f in facets] == ["viewpoint", "gaze_dir", "expression", "no_default"]
    o = Oracle(SHAPES["vocabulary_json"]())
    assert o.info.known_found == 1 and o.info.strict_known == 1
    for t in ("1girl", "solo", "long_hair", "hair_ornament", "blue_eyes", "looking_at_viewer", "pov", "from_behind", "feet", "barefoot", "toes", "more", "tags", "groups"):
        assert not o.should_drop_tag(t), t
    assert o.should_drop_tag("blush") and o.should_drop_tag("happy") and o.should_drop_tag("penis")    # a string value is one tag, not a list
    o = Oracle(SHAPES["profile_alwaysadd_pascal"]())
    assert o.trigger == "my_trig" and o.select_tags_from_line("1girl, smile")[:4] == ["my_trig", "1girl", "beret", "bob_cut"]
    assert Oracle(SHAPES["profile_alwaysadd_lower"]()).select_tags_from_line("1girl, dress")[:4] == ["sgrui", "1girl", "smile", "long_hair"]
    o = Oracle(SHAPES["profile_always_add_unbound"]())
    assert o.trigger == "other" and "beret" not in o.select_tags_from_line("1girl, dress")


@pytest.mark.parametrize("name", sorted(REJECTED))
def test_tables_this_build_refuses(name):
    """The reference (and the oracle) load these; the table compiler says so instead of producing captions that could differ."""
    cfg = REJECTED[name]()
    Oracle(cfg)                                   # loads
    with pytest.raises(RuntimeError) as e:
        HostSim(cfg)
    assert "not representable" in str(e.value)


def test_top_tags_file_order_and_round_trip():
    """The formatter the library's tp_format_top_tags calls (top_tags_text), on the host build."""
    from helpers import check_top_tags_file
    check_top_tags_file(HostSim)
