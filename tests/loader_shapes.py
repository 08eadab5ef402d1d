"""Table files in every shape the reference's loaders accept, for the parity tests of the table compiler
(tests/test_loader_shapes_cpu.py on the host emulation, tests/test_gpu_parity.py on the GPU).

Each entry: name -> (Config factory, lines that use what the file defines).  The shapes follow LoadLexicon
(Program.cs:1275-1305), LoadActions / ParseActionRule (Program.cs:1307-1372), LoadRequiredFacets (Program.cs:1374-1406),
ParseStringArray (Program.cs:1408-1415), ExtractKnownTagsFromJson (Program.cs:1036-1053) and LoadProfile (Program.cs:918-931).
"""
import json

from helpers import Config, shipped_config


def _j(obj):
    return json.dumps(obj, ensure_ascii=False).encode("utf-8")


# "aliases", "normalization" and "synonyms" all feed the alias map (later sections overwrite earlier ones key by key);
# "implied" and "implied_tags" both feed the implied map.  Keys and values go through NormalizeTagBasic.
LEXICON_ALT_KEYS = _j({
    "aliases": {"Bare Foot": "barefoot", "overwritten": "smile"},
    "normalization": {"overwritten": "Long Hair", "Grin ": "smile", "ＦＵＬＬ": "dress"},
    "synonyms": {"Looking At Viewer": ["eye contact", "Staring", ""], "not_an_array": "skipped", "dress": ["gown", "frock"], "overwritten": []},
    "implied": {"barefoot": ["feet", "Toes"], "gown": "dress"},
    "implied_tags": {"Smile": ["happy", "happy", "HAPPY", " "], "barefoot": "feet", "blush": []},
})

# a map of categories: every non-blank parent becomes a rule that needs itself, subtypes match themselves
ACTIONS_CATEGORY_MAP = _j({
    "foot play": {"parent": ["Footjob", " ", "pedal play"], "subtypes": ["two-footed footjob", "solejob"]},
    "number": 5,
    "hand": {"parent": "handjob"},
    "no parent": {"subtypes": ["x"]},
})

# one rule object at the top level; parent as an array whose first non-blank entry counts; subtypes as strings and objects
ACTIONS_SINGLE_OBJECT = _j({
    "parent": ["", "Foot Job"], "required_all": "feet", "required_any": ["penis", "Toes", "toes"], "forbidden_any": [],
    "subtypes": ["solejob", {"tag": "Toe Job", "match_any": ["toes"], "match_all": "feet"}, {"match_any": ["no tag"]}, 7, {"tag": " "}],
})

ACTIONS_ARRAY_MIXED = _j([
    {"parent_tag": "footjob", "required_any": ["feet", "soles"]},
    {"parent": "Hand Job", "required_all": ["hand", "penis"], "subtypes": [{"tag": "double handjob", "match_all": ["2girls"]}]},
    {"parent_tag": " ", "parent": ["", " "]},
    "not an object",
    {"required_all": ["no parent"]},
])

# arrays: the first "unknown_..." entry is the default (or unknown_<facet>), every other entry a rule that matches itself
FACETS_ARRAY_SHAPE = _j({
    "viewpoint": ["pov", "from behind", "UNKNOWN_view", "unknown_second"],
    "Gaze Dir": ["looking at viewer", "looking away"],
    "empty": [],
    "number": 3,
    "expression": {"default": "Unknown Expression", "rules": [{"tag": "smile", "match_any": ["smile", "grin"]}, {"tag": ""}, 5,
                                                              {"tag": "Open Mouth", "match_all": ["open mouth"], "match_any": "teeth"}]},
    "no default": {"rules": [{"tag": "blush", "match_any": ["blush"]}]},
})

# every string value and every property name anywhere in the document is a known tag
VOCABULARY_JSON = _j({
    "tags": ["1girl", "Solo", "smile", ["dress", {"Long Hair": "hair ornament"}], 5, None, True],
    "groups": {"Blue Eyes": 3, "looking at viewer": {"pov": ["from behind"]}, "feet": "barefoot", "": "toes"},
    "more": "blush; happy, penis",
})

_SHAPE_LINES = [
    "1girl, solo, Bare Foot, smile, dress",
    "1girl, overwritten, grin, ＦＵＬＬ, eye contact, staring",
    "1girl, gown, frock, not_an_array, skipped, blush",
    "1girl, footjob, feet, penis, toes, solejob, two-footed footjob",
    "1girl, pedal play, handjob, hand, penis, 2girls, double handjob",
    "1girl, foot job, toe job, soles, feet, toes",
    "1girl, pov, from behind, looking at viewer, looking away, open mouth, teeth",
    "1girl, solo, long hair, hair ornament, blue eyes, barefoot, happy",
    "beret, bob cut, 1girl, smile",
    "x",
    "",
    # tokens that an alias turns into a force-allowed tag with a comma (FORCE_ALLOWED_COMMAS): the join and re-split of
    # Program.cs:604, 578 make each piece a candidate, with the scores of the candidates behind them moving along
    "1girl, ab, smile, dress, long hair",
    "xy, 1girl, AB, xy, blue eyes, x",
    "1girl, a, y, xy, standing",
    "gone, 1girl, smile, dress, classroom",
    "ab",
    "xy, gone",
    "1girl, xyz, ab, xy, gone, smile, dress, long hair, blue eyes, standing, classroom, a,_b",
]

# A trigger and always-add entries that hold commas, and aliases whose values are those strings.  NormalizeTag makes the trigger
# "a,_b"; IsValidTag refuses every string with a comma, so none of them is ever added by AddOne (Program.cs:695), but ShouldDropTag
# lets them through the line ingest (Program.cs:1123), whose joined line SplitTags cuts at their commas (Program.cs:604, 578).
FORCE_ALLOWED_COMMAS = dict(profile=_j({"Trigger": "a, b", "AlwaysAdd": ["x,y", "smile", ",,", "p;,q.,r"]}),
                            lexicon=_j({"aliases": {"ab": "A, b", "xy": "x,y", "gone": ",,", "xyz": "p;,q.,r", "bare foot": "barefoot"},
                                        "implied": {"x": ["feet"], "q": ["y"]}}))

SHAPES = {
    "force_allowed_commas": lambda: shipped_config(strict_known=False, strict_known_explicit=True, **FORCE_ALLOWED_COMMAS),
    "force_allowed_commas_strict": lambda: shipped_config(**FORCE_ALLOWED_COMMAS),
    "lexicon_alt_keys": lambda: shipped_config(lexicon=LEXICON_ALT_KEYS),
    "lexicon_alt_keys_open": lambda: shipped_config(lexicon=LEXICON_ALT_KEYS, strict_known=False, strict_known_explicit=True),
    "actions_category_map": lambda: shipped_config(actions=ACTIONS_CATEGORY_MAP, strict_known=False, strict_known_explicit=True),
    "actions_single_object": lambda: shipped_config(actions=ACTIONS_SINGLE_OBJECT, strict_known=False, strict_known_explicit=True),
    "actions_array_mixed": lambda: shipped_config(actions=ACTIONS_ARRAY_MIXED),
    "facets_array_shape": lambda: shipped_config(required_facets=FACETS_ARRAY_SHAPE, strict_known=False, strict_known_explicit=True),
    "facets_array_shape_strict": lambda: shipped_config(required_facets=FACETS_ARRAY_SHAPE),
    "vocabulary_json": lambda: shipped_config(vocabulary=VOCABULARY_JSON, vocabulary_is_json=1),
    "vocabulary_json_builtin_rules": lambda: Config(vocabulary=VOCABULARY_JSON, vocabulary_is_json=1, minimal_subject_tags="1girl", trigger="trig"),
    # AlwaysAdd binds under any capitalisation of the property name, not under the shipped spelling `always_add`
    "profile_alwaysadd_pascal": lambda: shipped_config(profile=_j({"Trigger": "My Trig", "AlwaysAdd": ["Beret", "bob cut", "beret", " "]})),
    "profile_alwaysadd_lower": lambda: shipped_config(profile=_j({"trigger": "sgrui", "alwaysadd": ["smile", "Long Hair"]})),
    "profile_always_add_unbound": lambda: shipped_config(profile=_j({"trigger": "other", "always_add": ["beret"]})),
}


def shape_lines():
    from test_pipeline_logic_cpu import fuzz_lines
    from helpers import EDGE_LINES
    return _SHAPE_LINES + EDGE_LINES + fuzz_lines(37, 250)


# Tables the reference would load and this build refuses when the context is made (TP_E_CONFIG):
REJECTED = {
    # two strings that the reference's ordinal-ignore-case sets take for one tag but that are spelled differently even after
    # ToLowerInvariant (final and medial sigma both fold to the capital): which spelling a caption shows depends on insertion order
    "case_mapping_twins": lambda: shipped_config(lexicon=_j({"implied": {"smile": ["abς"]}}),
                                                 vocabulary="1girl\nsmile\nabσ\n".encode("utf-8"), vocabulary_is_json=0),
    # force-allowed tags with commas are cut up by the line ingest's join and re-split; the tables hold four such tags of up to
    # eight pieces each
    "five_tags_with_commas": lambda: shipped_config(profile=_j({"Trigger": "a,b", "AlwaysAdd": ["c,d", "e,f", "g,h", "i,j"]})),
    "nine_pieces": lambda: shipped_config(profile=_j({"Trigger": "a,b,c,d,e,f,g,h,i"})),
}
