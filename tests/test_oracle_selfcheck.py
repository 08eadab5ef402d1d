"""Pins the oracle to the reference's own test, RunSelfChecks (Program.cs:1951-1983), and to known-answer
vectors read off the reference source (normalisation, score boundaries, load-order and profile-binding quirks).

The reference has no other tests and cannot be built in this image (.NET is absent), so these vectors are what
"parity" is anchored on; everything else compares the HIP path with this oracle.
"""
import json

import pytest

from helpers import DATA, Config, Oracle, shipped_config

SELF_CHECK_LINE = ("1girl, solo, blue_hair, long_hair, blue_eyes, smile, school_uniform, standing, classroom, "
                   "masterpiece, best_quality, highres, watermark")


@pytest.fixture(scope="module")
def builtin():
    # RunSelfChecks runs before any table load: built-in lexicon and action rules, no profile, trigger "trigger_token".
    return Oracle(Config(trigger="trigger_token", minimal_subject_tags="1girl", no_profile=True))


def test_selfcheck_1_to_4_select_tags(builtin):
    tags = builtin.select_tags_from_line(SELF_CHECK_LINE, 1.0)
    assert tags[0] == "trigger_token"                                        # prefixFirst
    assert "1girl" in tags[:3]                                               # hasSubjectEarly
    assert sum(builtin.classify_tag(t) == "StyleQuality" for t in tags) <= 4  # maxQualityRespected
    assert "watermark" not in tags                                           # denyDropped
    # full result, derived by hand from Program.cs:607-700 (styleCap 3; school_uniform is Outfit: Outfit is tested before Background)
    assert tags == ["trigger_token", "1girl", "solo", "blue_hair", "long_hair", "blue_eyes", "smile", "school_uniform", "standing",
                    "classroom", "masterpiece", "best_quality", "highres"]


def test_selfcheck_5_to_8_action_inference(builtin):
    a = builtin.infer_action_tags(["1girl", "feet", "penis"], ["1girl", "feet", "penis"])
    assert "footjob" in a and "penis_between_feet" not in a
    assert a == ["1girl", "feet", "penis", "footjob"]
    b = builtin.infer_action_tags(["1girl", "feet", "penis", "between_feet"], ["1girl", "feet", "penis", "between_feet"])
    assert "penis_between_feet" in b
    assert b == ["1girl", "feet", "penis", "between_feet", "footjob", "penis_between_feet"]
    c = builtin.infer_action_tags(["1girl", "penis", "hand", "grip"], ["1girl", "penis", "hand", "grip"])
    assert "handjob" in c and "hand_wrapping_penis" in c
    assert c == ["1girl", "penis", "hand", "grip", "handjob", "hand_wrapping_penis"]
    d = builtin.infer_action_tags(["1girl", "penis", "vaginal_penetration", "missionary"], ["1girl", "penis", "vaginal_penetration", "missionary"])
    assert "sex" in d and "missionary" in d
    assert d == ["1girl", "penis", "vaginal_penetration", "missionary", "sex"]   # subtypes already present are not re-added (Program.cs:745)


def test_selfcheck_9_alias(builtin):
    assert builtin.normalize_tag("pedal stimulation") == "footjob"


def test_selfcheck_10_caption_format(builtin):
    inferred = builtin.infer_action_tags(["1girl", "feet", "penis"], ["1girl", "feet", "penis"])
    line = builtin.format_caption(inferred)
    assert "# Actions" not in line and "footjob" in line
    assert line == "1girl, feet, penis, footjob"


def test_normalize_tag_known_answers(builtin):
    n = builtin.normalize_tag
    assert n("  Blue   Hair ") == "blue_hair"
    assert n("long__hair") == "long_hair"
    assert n("\"smile\"") == "smile"
    assert n("tag;.;") == "tag"
    assert n("tag。、") == "tag"
    # not idempotent: the quote survives the first pass because '.' was last (Program.cs:1150-1151)
    assert n("x'.") == "x'"
    assert n(n("x'.")) == "x"
    assert n("x.'.'.") == "x.'.'" and n("x.'.'") == "x.'" and n("x.'") == "x"
    # white space exposed by the quote trim is not re-trimmed
    assert n("' x'") == "_x"
    # `$` also matches before a final '\n' (only reachable through quote-wrapped text)
    assert n("\"foo.\n\"") == "foo_"
    assert n("a　b") == "a_b"          # \s covers U+3000
    assert n(" tag ") == "tag"    # Trim() covers the Unicode space separators
    assert n("KIMONO") == "kimono"    # KELVIN SIGN lower-cases to 'k'
    assert n("İx") == "İx"       # U+0130 keeps itself under ToLowerInvariant
    assert n("fellatio") == "blowjob" and n("soles") == "foot_sole" and n("Fingers") == "hand"
    assert n("") == "" and n(" ; ") == ""


def test_is_valid_and_classify(builtin):
    v = builtin.is_valid_tag
    assert v("a") and v("9lives") and v("a_b(c)!'/-")
    assert not v("") and not v(":x") and not v("_a") and not v("A") and not v("a b") and not v("a,b") and not v("café")
    c = builtin.classify_tag
    assert c("1girl") == "Identity" and c("solo") == "Identity"
    assert c("long_sleeves") == "Identity"           # 2-4 alnum segments and no regex hit: "likely character name"
    assert c("simple_background") == "Identity"
    assert c("school_uniform") == "Outfit"           # Outfit before Background
    assert c("blue_hair") == "FaceHairEyes" and c("looking_at_viewer") == "FaceHairEyes"   # "looking_" is in the first regex
    assert c("standing") == "PoseAction" and c("footjob") == "PoseAction"
    assert c("classroom") == "Background" and c("masterpiece") == "StyleQuality"
    assert c("cat") == "Other" and c("a_b_c_d_e") == "Other"


def test_score_boundaries_line_mode():
    """Index 72 scores exactly 0.28 and IS evidence (Program.cs:724 uses <); index 40 scores 0.6 and counts toward
    the style cap (:704 uses >=); with weight 0.7, index 60 is 0.27999999999999997 and is NOT evidence."""
    o = Oracle(Config(trigger="trg", minimal_subject_tags="1girl", no_profile=True))
    tags = ["zq%dx" % i for i in range(100)]      # plain unknown tags, bucket Other
    r = o.run_texts([", ".join(tags)], detail=True)
    ev = r.details[0]["evidence"]
    assert "zq72x" in ev and "zq73x" not in ev and len(ev) == 73
    o7 = Oracle(Config(trigger="trg", minimal_subject_tags="1girl", no_profile=True, line_weight=0.7))
    ev7 = o7.run_texts([", ".join(tags)], detail=True).details[0]["evidence"]
    assert "zq59x" in ev7 and "zq60x" not in ev7 and len(ev7) == 60
    # style cap: four style tags, the fourth at index 40 (score 0.6) still counts -> cap 4; at index 41 it does not -> cap 3
    style = ["masterpiece", "highres", "absurdres", "bokeh"]
    for idx, want in ((40, 4), (41, 3)):
        line = style[:3] + ["zq%dx" % i for i in range(idx - 3)] + [style[3]] + ["monochrome", "detailed"]
        sel = o.select_tags_from_line(", ".join(line), 1.0)
        n_style = sum(o.classify_tag(t) == "StyleQuality" for t in sel[:sel.index("zq0x")])
        assert n_style == min(want, 4), (idx, sel)


def test_load_order_quirk_and_strict_vocabulary():
    """The vocabulary is normalised with the BUILT-IN lexicon (Program.cs:160 runs before :163): `fingers` is stored as
    `hand`, `fellatio` as `blowjob`.  Finding top_tags.txt turns strict mode on (Program.cs:999-1002)."""
    o = Oracle(shipped_config())
    assert o.info.known_found == 1 and o.info.strict_known == 1 and o.info.n_known_tags == 5811
    assert not o.should_drop_tag("hand")
    assert o.should_drop_tag("fingers")       # not a vocabulary entry any more
    assert o.should_drop_tag("zzz_unknown")
    assert not o.should_drop_tag("sgrui")     # force-allowed trigger
    o2 = Oracle(shipped_config(strict_known=False, strict_known_explicit=True))
    assert o2.info.strict_known == 0 and not o2.should_drop_tag("zzz_unknown")


def test_profile_binding_quirk():
    """ProfileConfig binds with PropertyNameCaseInsensitive only: `always_add` (shipped spelling) does not bind AlwaysAdd."""
    o = Oracle(shipped_config())
    assert o.trigger == "sgrui"
    assert o.select_tags_from_line("1girl, smile, dress")[:2] == ["sgrui", "1girl"]
    assert "beret" not in o.select_tags_from_line("1girl, smile, dress")
    prof = json.dumps({"Trigger": "My Trig", "alwaysadd": ["Beret", "bob cut"]}).encode()
    o2 = Oracle(shipped_config(profile=prof))
    assert o2.trigger == "my_trig"
    assert o2.select_tags_from_line("1girl, smile, dress")[:4] == ["my_trig", "1girl", "beret", "bob_cut"]


def test_actions_json_replaces_builtin_rules():
    o = Oracle(shipped_config())
    assert o.rule_names()["actions"] == ["footjob", "handjob", "blowjob"]      # the built-in `sex` rule is gone (Program.cs:1337)
    assert [f["name"] for f in o.rule_names()["facets"]] == ["viewpoint", "gaze", "underwear", "fluids", "ejaculation", "expression", "mouth_shape"]


def test_malformed_json_is_fatal():
    from oracle.oracle_py import OracleError
    with pytest.raises(OracleError):
        Oracle(shipped_config(lexicon=b'{"aliases": {"a": 5}}'))          # GetString on a number throws
    with pytest.raises(OracleError):
        Oracle(shipped_config(actions=b'[{"parent_tag": "x",}]'))         # trailing comma
    Oracle(shipped_config(profile=b'not json'))                           # LoadProfile swallows errors (Program.cs:928-930)


def test_line_ingest_known_answers():
    o = Oracle(shipped_config(strict_known=False, strict_known_explicit=True))
    f = o.normalize_raw_tag_line
    assert f("POSITIVE: 1girl, smile, dress\nNEGATIVE: lowres") == "1girl, smile, dress"
    assert f("```\n1girl, smile, dress\n```") == "1girl, smile, dress"
    assert f("sorry, I can't do that") == ""
    assert f("x" * 41) == "" and f("x" * 40) == "x" * 40
    assert f("a, bb, cc, BB") == "bb, cc"                                  # Length >= 2, Distinct
    assert f("1girl, watermark, rating:safe, 1boy") == "1girl"
