"""ctypes mirror of include/tagpipe.h (struct layouts and constants only).

This is what a host-language binding of the C ABI looks like: a C# P/Invoke
binding declares the same layouts.
"""
import ctypes as C

TP_OK = 0
TP_E_INVALID_ARG = -1
TP_E_CONFIG = -2
TP_E_DEVICE = -3
TP_E_CAPACITY = -4
TP_E_NOMEM = -5

TP_COV_IDENTITY = 0x01
TP_COV_HAIR = 0x02
TP_COV_EYES = 0x04
TP_COV_OUTFIT = 0x08
TP_COV_POSE_ACTION = 0x10
TP_COV_BACKGROUND = 0x20

TP_F_LOW = 0x01
TP_F_ELIGIBLE = 0x02
TP_F_FOOTJOB = 0x04
TP_F_FALLBACK = 0x08
TP_F_LINE_REJECTED = 0x10
TP_F_UNSUPPORTED = 0x20
TP_F_SAFE_REJECT = 0x40

TP_SRC_PRIMARY_JOY = 0
TP_SRC_SECONDARY_JOY = 1
TP_SRC_OPENAI = 2

# Program.cs:43-45 order; tp_verdict.safe_hit indexes this list.
SAFE_MODE_REJECT_TAGS = (
    "nsfw", "nude", "naked", "nipples", "breasts", "pussy", "cameltoe", "sex", "penis", "cum", "orgasm",
    "masturbation", "fellatio", "anal", "vaginal", "hentai", "explicit", "porn", "bdsm", "futanari",
)


class tp_verdict(C.Structure):
    _fields_ = [
        ("coverage", C.c_uint8),
        ("flags", C.c_uint8),
        ("n_tags", C.c_uint16),
        ("n_candidates", C.c_uint16),
        ("n_selected", C.c_uint16),
        ("safe_hit", C.c_int16),
        ("n_evidence", C.c_uint16),
        ("n_unknown", C.c_uint16),
        ("n_dropped", C.c_uint16),
    ]


assert C.sizeof(tp_verdict) == 16

VERDICT_DTYPE = [
    ("coverage", "u1"), ("flags", "u1"), ("n_tags", "<u2"), ("n_candidates", "<u2"), ("n_selected", "<u2"),
    ("safe_hit", "<i2"), ("n_evidence", "<u2"), ("n_unknown", "<u2"), ("n_dropped", "<u2"),
]


class tp_counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "files", "low", "action_eligible", "action_footjob", "fallback", "line_rejected", "unsupported", "safe_rejected")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class tp_source(C.Structure):
    _fields_ = [("path", C.c_char_p), ("data", C.POINTER(C.c_uint8)), ("len", C.c_uint64)]


class tp_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("lexicon", tp_source),
        ("actions", tp_source),
        ("required_facets", tp_source),
        ("profile", tp_source),
        ("vocabulary", tp_source),
        ("vocabulary_is_json", C.c_int32),
        ("allowlist", tp_source),
        ("denylist", tp_source),
        ("trigger", C.c_char_p),
        ("auto_prefix_tags", C.c_char_p),
        ("minimal_subject_tags", C.c_char_p),
        ("newline", C.c_char_p),
        ("no_profile", C.c_int32),
        ("strict_known", C.c_int32),
        ("strict_known_explicit", C.c_int32),
        ("safe_mode", C.c_int32),
        ("line_weight", C.c_double),
        ("line_source", C.c_int32),
        ("device", C.c_int32),
    ]


class tp_table_info(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "n_strings", "n_action_rules", "n_action_subtypes", "n_facets", "fired_words", "n_known_tags", "strict_known",
        "known_found")]


class tp_result(C.Structure):
    _fields_ = [
        ("n_files", C.c_uint64),
        ("captions", C.POINTER(C.c_uint8)),
        ("caption_offsets", C.POINTER(C.c_uint64)),
        ("verdicts", C.POINTER(tp_verdict)),
        ("counters", tp_counters),
        ("histogram", C.POINTER(C.c_uint64)),
        ("histogram_other", C.c_uint64),
        ("fired", C.POINTER(C.c_uint32)),
        ("facet_choice", C.POINTER(C.c_uint16)),
        ("opaque", C.c_void_p),
    ]


class tp_audit(C.Structure):
    _fields_ = [("text", C.POINTER(C.c_uint8)), ("offsets", C.POINTER(C.c_uint64))]


class tp_device_io(C.Structure):
    _fields_ = [
        ("arena", C.c_void_p),
        ("offsets", C.c_void_p),
        ("n_files", C.c_uint64),
        ("captions", C.c_void_p),
        ("caption_capacity", C.c_uint64),
        ("caption_offsets", C.c_void_p),
        ("verdicts", C.c_void_p),
        ("histogram", C.c_void_p),
        ("counters", C.c_void_p),
        ("fired", C.c_void_p),
        ("facet_choice", C.c_void_p),
        ("stream", C.c_void_p),
    ]


class tp_timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("select_ms", C.c_float), ("general_ms", C.c_float), ("emit_ms", C.c_float),
                ("other_ms", C.c_float), ("general_files", C.c_uint64), ("lean_front_ms", C.c_float), ("lean_back_ms", C.c_float),
                ("fused_ms", C.c_float), ("fused_reruns", C.c_uint64)]


class tp_replies(C.Structure):
    _fields_ = [("n_files", C.c_uint64), ("n_tags", C.c_uint64), ("arena", C.POINTER(C.c_uint8)), ("tag_offsets", C.POINTER(C.c_uint64)),
                ("tag_begin", C.POINTER(C.c_uint64)), ("secondary_begin", C.POINTER(C.c_uint64)), ("scores", C.POINTER(C.c_double)),
                ("sources", C.POINTER(C.c_uint8)), ("status", C.POINTER(C.c_uint8)), ("has_score", C.POINTER(C.c_uint8)), ("opaque", C.c_void_p)]


TP_REPLY_OK, TP_REPLY_NO_TAGS, TP_REPLY_ORDER, TP_REPLY_FAILED = 0, 1, 2, 3


class Config:
    """Keyword-style builder for tp_config that keeps the backing buffers alive.

    Table sources accept a path (str / os.PathLike), raw bytes, or None.
    Defaults follow the reference's option defaults (Program.cs:1515-1545).
    """

    SOURCES = ("lexicon", "actions", "required_facets", "profile", "vocabulary", "allowlist", "denylist")

    def __init__(self, *, lexicon=None, actions=None, required_facets=None, profile=None, vocabulary=None,
                 vocabulary_is_json=-1, allowlist=None, denylist=None, trigger="sgrui", auto_prefix_tags="",
                 minimal_subject_tags="", newline="\n", no_profile=False, strict_known=False,
                 strict_known_explicit=False, safe_mode=True, line_weight=1.0, line_source=TP_SRC_OPENAI, device=0):
        self._keep = []
        c = tp_config()
        c.struct_size = C.sizeof(tp_config)
        local = dict(lexicon=lexicon, actions=actions, required_facets=required_facets, profile=profile,
                     vocabulary=vocabulary, allowlist=allowlist, denylist=denylist)
        for name in self.SOURCES:
            setattr(c, name, self._source(local[name]))
        c.vocabulary_is_json = int(vocabulary_is_json)
        c.trigger = self._cstr(trigger)
        c.auto_prefix_tags = self._cstr(auto_prefix_tags)
        c.minimal_subject_tags = self._cstr(minimal_subject_tags)
        c.newline = self._cstr(newline)
        c.no_profile = int(bool(no_profile))
        c.strict_known = int(bool(strict_known))
        c.strict_known_explicit = int(bool(strict_known_explicit))
        c.safe_mode = int(bool(safe_mode))
        c.line_weight = float(line_weight)
        c.line_source = int(line_source)
        c.device = int(device)
        self.struct = c

    def _cstr(self, s):
        if s is None:
            return None
        b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
        self._keep.append(b)
        return b

    def _source(self, v):
        src = tp_source()
        if v is None:
            return src
        if isinstance(v, (bytes, bytearray, memoryview)):
            buf = (C.c_uint8 * max(1, len(v))).from_buffer_copy(bytes(v) or b"\0")
            self._keep.append(buf)
            src.data = C.cast(buf, C.POINTER(C.c_uint8))
            src.len = len(v)
            return src
        import os
        p = os.fspath(v).encode("utf-8")
        self._keep.append(p)
        src.path = p
        return src
