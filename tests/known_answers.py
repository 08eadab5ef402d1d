"""Known answers derived BY HAND from /root/reference/Program.cs (line numbers cited per step), for the parts of the span
the reference's own --self-check (Program.cs:1951-1983) is silent about.  Each vector is an input plus the exact text the
reference would write; the CPU tests check them on the oracle and the host emulation, the GPU tests through the C ABI.
Nothing here was produced by running code: a change of an expected string needs a new derivation, not a regenerated file.

Facts used throughout (shipped tables, tests/golden/reference_data):
  * prefix of every selection: AddOne(trigger "sgrui"), AddOne("1girl") (Program.cs:664-665); the shipped profile's
    "always_add" key does not bind to ProfileConfig.AlwaysAdd (Program.cs:409-413, 923), so nothing else is prefixed;
  * bucket fills in the order Identity 3 / FaceHairEyes 10 / Outfit 8 / PoseAction 6 / Background 4 / StyleQuality, then
    one sweep over the sorted candidates (Program.cs:670-680);
  * ClassifyTag (Program.cs:1103-1111): "solo", "1girl" are SubjectTokens -> Identity; a tag of 2-4 alphanumeric segments
    that none of the five regexes hits is a "likely character name" -> Identity (Program.cs:1113-1119);
  * line ingest scores 1.0, 0.99, ... by candidate index (Program.cs:581); evidence = candidates with !(score < 0.28)
    (Program.cs:724);
  * required_facets.json: seven facets in file order viewpoint, gaze, underwear, fluids, ejaculation, expression,
    mouth_shape; first matching rule wins, else the default tag, which is logged with ":default" (Program.cs:755-775).
"""
import numpy as np

FACET_DEFAULTS = ["unknown_pov", "unknown_gaze", "unknown_underwear", "unknown_fluids", "unknown_ejaculation", "unknown_expression", "unknown_mouth"]
FACET_NAMES = ["viewpoint", "gaze", "underwear", "fluids", "ejaculation", "expression", "mouth_shape"]
ALL_DEFAULT_FIRED = ";".join("facet:%s:%s:default" % (n, d) for n, d in zip(FACET_NAMES, FACET_DEFAULTS))

# ---- KA1: EvaluateCoverage's strings, a facet rule and six defaults (Program.cs:1056-1095, 755-775) -----------------------
# Config: shipped tables, strict vocabulary (top_tags.txt found -> strict on, Program.cs:999-1002).  All six tags are lines of
# top_tags.txt (1, 13, 45, 44, 7, 2), none is an alias key, so hop 1 leaves them alone and all six are candidates.
#   SelectTags: sgrui, 1girl | Identity: "1girl" again fails AddOne (selected.Add false, Program.cs:696) and uses up no share,
#   "solo" | FaceHairEyes by score: blonde_hair (0.99), smile (0.96) | Outfit: school_uniform ("uniform" hits OutfitRegex before
#   "school" could hit BackgroundRegex, Program.cs:1106-1108) | PoseAction: sitting | nothing for Background / Style | sweep adds nothing.
#   EvaluateCoverage on those 7: trigger and 1girl -> identity ok; blonde_hair -> hair ok; no tag holds eyes|eyelashes|eyebrows|
#   pupil|heterochromia -> eyes missing; outfit ok; pose_action ok; no Background bucket tag -> background_scene missing.
#   MissingCategoriesForPrompt = required + enrichment = eyes|background_scene; it is not empty and the OpenAI passes are off,
#   so the stop reason becomes openai_disabled_or_missing_api_key (Program.cs:246-248).
#   Evidence = the six candidates in candidate order (none has implied tags).  "smile" is in evidence: facet "expression" takes
#   its second rule (match_any smile|smiling), logs facet:expression:smile, and adds nothing (smile is in the final set already,
#   Program.cs:768); the other six facets add and log their defaults.  Face and outfit tags present -> not low; no safe-mode tag.
KA1_LINE = "1girl, blonde_hair, sitting, school_uniform, smile, solo"
KA1_FINAL = "sgrui;1girl;solo;blonde_hair;smile;school_uniform;sitting;unknown_pov;unknown_gaze;unknown_underwear;unknown_fluids;unknown_ejaculation;unknown_mouth"
KA1_FIRED = ("facet:viewpoint:unknown_pov:default;facet:gaze:unknown_gaze:default;facet:underwear:unknown_underwear:default;facet:fluids:unknown_fluids:default;"
             "facet:ejaculation:unknown_ejaculation:default;facet:expression:smile;facet:mouth_shape:unknown_mouth:default")
KA1_ROW = ('"a.png","' + KA1_FINAL + '","1girl;blonde_hair;sitting;school_uniform;smile;solo","' + KA1_FIRED + '","","","top_tags.txt","sgrui",'
           '"openai_disabled_or_missing_api_key",'
           '"pass0_joy missing=eyes|background_scene required_missing=eyes details=identity=ok|hair=ok|eyes=missing|outfit=ok|pose_action=ok|background_scene=missing tags=7"')
KA1_CAPTION = KA1_FINAL.replace(";", ", ") + "\n"
KA1_FACET_CHOICE = [0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 1, 0xFFFF]
KA1_AUDIT = {"names": ["a.png"], "vocab_file": "top_tags.txt", "used_profile": "sgrui"}

# ---- KA2: ExpandImpliedTags on the shipped foot / soles / fingers / licking keys (Program.cs:1160-1179, 1155-1156) ------------
# Config: shipped tables without a vocabulary file (nothing is dropped as unknown, GetUnknownTags returns [] - Program.cs:818).
#   Hop 1 (NormalizeRawTagLine, Program.cs:599): all four tokens are ALIAS keys of lexicon.json: foot -> feet, soles -> foot_sole,
#   fingers -> hand, licking -> oral.  They are also IMPLIED keys, but Implied is looked up on the hopped form (Program.cs:1170), so
#   those four entries are never reached; what is reached: foot_sole -> [feet] (present), oral -> [mouth] (added in round 1; round 2
#   adds nothing).  Evidence in insertion order: feet, foot_sole, hand, oral, mouth.
#   SelectTags: sgrui, 1girl | Identity: foot_sole (two alphanumeric segments, no regex hit -> likely character name) | no tag
#   for the other buckets (feet, hand, oral classify Other: "hand" does not hold "hands") | sweep by score: feet, hand, oral.
#   normalizedFinalSet = those six; its own closure appends mouth (oral's implied tag); then the seven facet defaults.
#   Coverage: only identity is met.  No face and no outfit tag -> low (MissingCoreBuckets, Program.cs:1097-1101).
KA2_LINE = "foot, soles, fingers, licking"
KA2_FINAL = "sgrui;1girl;foot_sole;feet;hand;oral;mouth;" + ";".join(FACET_DEFAULTS)
KA2_ROW = ('"b.png","' + KA2_FINAL + '","feet;foot_sole;hand;oral;mouth","' + ALL_DEFAULT_FIRED + '","","","","sgrui","openai_disabled_or_missing_api_key",'
           '"pass0_joy missing=hair|eyes|outfit|pose_action|background_scene required_missing=hair|eyes|outfit|pose_action '
           'details=identity=ok|hair=missing|eyes=missing|outfit=missing|pose_action=missing|background_scene=missing tags=6"')
KA2_CAPTION = KA2_FINAL.replace(";", ", ") + "\n"
KA2_AUDIT = {"names": ["b.png"], "vocab_file": None, "used_profile": "sgrui"}

# ---- KA3: GetUnknownTags' order, "AB" before "A_B" (Program.cs:817-824) ---------------------------------------------------------
# Config: shipped tables, vocabulary found but --no-strict-known-tags (tags outside the vocabulary are kept and reported).
#   OrderBy(StringComparer.OrdinalIgnoreCase) compares upper-cased UTF-16 units: 'B' (0x42) < '_' (0x5F), so "zzab" sorts before
#   "zza_b"; a byte-wise sort of the lower-case strings would put "zza_b" ('_' 0x5F < 'b' 0x62) first.
#   Selection: sgrui, 1girl | Identity: zza_b (likely character name) | blue_hair, blue_eyes | dress | standing | classroom |
#   sweep: zzab -> 9 tags, every coverage category met.  Unknown final tags: sgrui, zza_b, zzab and the seven facet defaults
#   (none of them is a line of top_tags.txt), sorted as above.  Reasons: the raw candidate "1girl" is a SubjectToken and matches
#   ^\d+(girl|girls|boy|boys)$, so the reference reports "multi_person" for it (Program.cs:829-837: a quirk, one girl is not
#   several); nothing male, not low; no banned tokens.  KnownTagDictionaryCount = 5811 = KnownTags.Count (Program.cs:1021-1028 over
#   top_tags.txt's 5813 lines: line 711 "..." normalises to nothing, line 2954 "steins;gate" splits in two at the ';', line 828
#   "licking" becomes "oral" under the built-in lexicon and meets line 556, and line 5025 "gate" meets the second half of 2954).
KA3_LINE = "1girl, zza_b, zzab, blue_hair, blue_eyes, dress, standing, classroom"
KA3_UNKNOWN = ["sgrui", "unknown_ejaculation", "unknown_expression", "unknown_fluids", "unknown_gaze", "unknown_mouth", "unknown_pov", "unknown_underwear",
               "zzab", "zza_b"]
KA3_JSON = ('{"ImagePath":"3.png","CaptionPath":"3.txt","Reasons":["multi_person"],"BannedTokensDetected":[],"DroppedTags":[],"FinalTagCount":9,"ShortCaption":false,'
            '"ThresholdsUsed":"primary=0.35;secondary=0.25","TopJoyTags":["1girl","zza_b","zzab","blue_hair","blue_eyes","dress","standing","classroom"],'
            '"UnknownFinalTags":[' + ",".join('"%s"' % t for t in KA3_UNKNOWN) + '],'
            '"KnownTagDictionaryFound":true,"KnownTagDictionaryCount":5811,"KnownTagDictionaryPath":"top_tags.txt",'
            '"CoverageTrace":["pass0_joy missing= required_missing= details=identity=ok|hair=ok|eyes=ok|outfit=ok|pose_action=ok|background_scene=ok tags=9"],'
            '"CoverageStopReason":"coverage_satisfied"}')
KA3_AUDIT = {"names": ["3.png"], "vocab_file": "top_tags.txt", "used_profile": "sgrui", "image_paths": ["3.png"], "caption_paths": ["3.txt"],
             "thresholds": "primary=0.35;secondary=0.25"}
KA3_CAPTION = "sgrui, 1girl, zza_b, blue_hair, blue_eyes, dress, standing, classroom, zzab, " + ", ".join(FACET_DEFAULTS) + "\n"

# ---- KA4: the 0.03 window of SelectTags' join of equal tags, at 0.03 exactly and just over (Program.cs:637-640) ------------------
# Config: shipped tables without a vocabulary file.  Scored ingest, one reply with per-tag sources: zz (secondary, rank 1),
# hand (OpenAI, rank 2), fingers (primary, rank 0; hop 1 makes it "hand").  Orders 0, 1, 2; the ingest's sort keeps them.
#   a) scores 0.06, 0.06, 0.03.  0.06 is 2 x 0.03 in binary too, so |0.03 - 0.06| == 0.03 and the window applies: the joined
#      "hand" takes the source of lower rank (primary), Score = max = 0.06, Order = min = 1.  Sorted by score desc, rank, order:
#      hand (0.06, rank 0) before zz (0.06, rank 1).  Both classify Other, so the sweep adds them in that order.
#   b) scores 0.53, 0.53, 0.5.  0.53 - 0.5 = 0.030000000000000027 > 0.03 only because of rounding: the window does NOT apply,
#      the higher-scored entry's source stays (rank 2), and zz (rank 1) sorts before hand (rank 2).
KA4_TAGS = ["zz", "hand", "fingers"]
KA4_SOURCES = [1, 2, 0]
KA4A_SCORES = [0.06, 0.06, 0.03]
KA4B_SCORES = [0.53, 0.53, 0.5]
KA4A_CAPTION = "sgrui, 1girl, hand, zz, " + ", ".join(FACET_DEFAULTS) + "\n"
KA4B_CAPTION = "sgrui, 1girl, zz, hand, " + ", ".join(FACET_DEFAULTS) + "\n"

# ---- KA5: MergeCandidateTags never lets the window decide (Program.cs:450-452) ----------------------------------------------------
# PreferSource is handed (old.Score, c.Score) inside the window and (best, best) outside it (Program.cs:452), so its own test
# |scoreA - scoreB| > 0.03 (Program.cs:797) is false either way and the source of lower rank always wins.  Primary reply: aa 0.95,
# hand 0.50; secondary reply: zz 0.90, hand 0.90 (cap 15: zz is new and accepted, hand is in the primary set, Program.cs:801-815).
# Merged "hand": Score 0.90 (the secondary's), Order min(1, 1) = 1, Source primary although the secondary's score is 0.4 higher.
# Sorted: aa 0.95; then the two 0.90s by rank: hand (primary) before zz (secondary, order 0).  A merge that gave "hand" the
# higher-scored entry's source would sort zz first.  All three classify Other.
KA5_TAGS = ["aa", "hand", "zz", "hand"]
KA5_SCORES = [0.95, 0.50, 0.90, 0.90]
KA5_CAPTION = "sgrui, 1girl, aa, hand, zz, " + ", ".join(FACET_DEFAULTS) + "\n"


def pack_tags(tags):
    """(arena, tag_offsets) of a list of tag texts."""
    bs = [t.encode("utf-8") for t in tags]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(b) for b in bs])
    return np.frombuffer(b"".join(bs), dtype=np.uint8).copy(), offs


def check_known_answers(make_engine, shipped_config, scored=True):
    """make_engine(config) -> an object with run_texts / run_scored / run_merged in the TagPipe / Oracle shape.
    Returns the number of vectors checked."""
    n = 0
    # KA1
    e = make_engine(shipped_config())
    r = e.run_texts([KA1_LINE], audit=dict(KA1_AUDIT))
    assert r.caption(0).decode() == KA1_CAPTION
    assert r.audit_rows[0].decode("utf-8").rstrip("\n") == KA1_ROW
    assert [int(x) for x in r.facet_choice[0]] == KA1_FACET_CHOICE
    assert int(r.verdicts["n_selected"][0]) == 7 and not (int(r.verdicts["flags"][0]) & 1)      # not low
    n += 1
    # KA2
    e = make_engine(shipped_config(vocabulary=None))
    r = e.run_texts([KA2_LINE], audit=dict(KA2_AUDIT))
    assert r.caption(0).decode() == KA2_CAPTION
    assert r.audit_rows[0].decode("utf-8").rstrip("\n") == KA2_ROW
    assert int(r.verdicts["flags"][0]) & 1                                                      # TP_F_LOW
    n += 1
    # KA3
    e = make_engine(shipped_config(strict_known=False, strict_known_explicit=True))
    r = e.run_texts([KA3_LINE], audit=dict(KA3_AUDIT))
    assert r.caption(0).decode() == KA3_CAPTION
    assert r.review_jsonl[0].decode("utf-8").rstrip("\n") == KA3_JSON
    n += 1
    if scored:
        e = make_engine(shipped_config(vocabulary=None))
        arena, toffs = pack_tags(KA4_TAGS)
        begin = np.array([0, len(KA4_TAGS)], dtype=np.uint64)
        src = np.array(KA4_SOURCES, dtype=np.uint8)
        assert e.run_scored(arena, toffs, begin, np.array(KA4A_SCORES), src).caption(0).decode() == KA4A_CAPTION
        assert e.run_scored(arena, toffs, begin, np.array(KA4B_SCORES), src).caption(0).decode() == KA4B_CAPTION
        arena, toffs = pack_tags(KA5_TAGS)
        r = e.run_merged(arena, toffs, np.array([0, 4], dtype=np.uint64), np.array([2], dtype=np.uint64), np.array(KA5_SCORES), 15)
        assert r.caption(0).decode() == KA5_CAPTION
        n += 3
    return n
