"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under imagecaptioner_amd/ does.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

from imagecaptioner_amd._abi import VERDICT_DTYPE, Config, tp_config, tp_counters, tp_table_info, tp_verdict

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class orc_result(C.Structure):
    _fields_ = [
        ("n_files", C.c_uint64),
        ("captions", C.POINTER(C.c_uint8)),
        ("caption_offsets", C.POINTER(C.c_uint64)),
        ("verdicts", C.POINTER(tp_verdict)),
        ("counters", tp_counters),
        ("fired", C.POINTER(C.c_uint32)),
        ("facet_choice", C.POINTER(C.c_uint16)),
        ("opaque", C.c_void_p),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_last_error.restype = C.c_char_p
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(tp_config)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_get_table_info.argtypes = [C.c_void_p, C.POINTER(tp_table_info)]
        L.orc_run_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(orc_result)]
        L.orc_run_scored.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                     C.c_int, C.c_int, C.POINTER(orc_result)]
        L.orc_release.argtypes = [C.POINTER(orc_result)]
        L.orc_histogram_text.restype = C.c_void_p
        L.orc_histogram_text.argtypes = [C.POINTER(orc_result), C.POINTER(C.c_uint64)]
        L.orc_detail_json.restype = C.c_char_p
        L.orc_detail_json.argtypes = [C.POINTER(orc_result), C.c_uint64]
        L.orc_audit_row.restype = C.c_char_p
        L.orc_audit_row.argtypes = [C.POINTER(orc_result), C.c_uint64, C.c_char_p, C.c_char_p, C.c_char_p]
        L.orc_review_row.restype = C.c_char_p
        L.orc_review_row.argtypes = [C.c_void_p, C.POINTER(orc_result), C.c_uint64, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
        L.orc_normalize_tag.restype = C.c_char_p
        L.orc_normalize_tag.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int]
        L.orc_classify_tag.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_is_valid_tag.argtypes = [C.c_char_p]
        L.orc_should_drop_tag.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_normalize_raw_tag_line.restype = C.c_char_p
        L.orc_normalize_raw_tag_line.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.orc_run_merged.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                     C.POINTER(orc_result)]
        L.orc_triage_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_infer_action_tags.restype = C.c_char_p
        L.orc_infer_action_tags.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.orc_format_caption.restype = C.c_char_p
        L.orc_format_caption.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_select_tags_from_line.restype = C.c_char_p
        L.orc_select_tags_from_line.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_double]
        L.orc_minimal_fallback.restype = C.c_char_p
        L.orc_minimal_fallback.argtypes = [C.c_void_p]
        L.orc_trigger.restype = C.c_char_p
        L.orc_trigger.argtypes = [C.c_void_p]
        L.orc_rule_names_json.restype = C.c_char_p
        L.orc_rule_names_json.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


BUCKETS = ("Identity", "FaceHairEyes", "Outfit", "PoseAction", "Background", "StyleQuality", "Other")


class OracleError(RuntimeError):
    pass


class OracleResult:
    """Host copy of one oracle batch (numpy arrays + python lists)."""

    def __init__(self, captions, offsets, verdicts, counters, fired, facet_choice, histogram, details):
        self.captions = captions
        self.caption_offsets = offsets
        self.verdicts = verdicts
        self.counters = counters
        self.fired = fired
        self.facet_choice = facet_choice
        self.histogram = histogram  # dict tag(str) -> count
        self.details = details      # list of dicts or None

    def caption(self, i):
        return bytes(self.captions[self.caption_offsets[i]:self.caption_offsets[i + 1]])


class Oracle:
    def __init__(self, config: Config):
        self._cfg = config
        self._L = lib()
        self._h = self._L.orc_create(C.byref(config.struct))
        if not self._h:
            raise OracleError(self._L.orc_last_error().decode("utf-8", "replace"))
        info = tp_table_info()
        self._L.orc_get_table_info(self._h, C.byref(info))
        self.info = info

    def close(self):
        if self._h:
            self._L.orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- single-function probes -------------------------------------------------------------
    def normalize_tag(self, s, basic=False):
        b = s.encode("utf-8") if isinstance(s, str) else s
        return self._L.orc_normalize_tag(self._h, b, len(b), int(basic)).decode("utf-8")

    def classify_tag(self, tag):
        return BUCKETS[self._L.orc_classify_tag(self._h, tag.encode("utf-8"))]

    def is_valid_tag(self, tag):
        return bool(self._L.orc_is_valid_tag(tag.encode("utf-8")))

    def should_drop_tag(self, tag):
        return bool(self._L.orc_should_drop_tag(self._h, tag.encode("utf-8")))

    def normalize_raw_tag_line(self, s):
        b = s.encode("utf-8") if isinstance(s, str) else s
        return self._L.orc_normalize_raw_tag_line(self._h, b, len(b)).decode("utf-8")

    def triage_texts(self, texts):
        """LooksBad(NormalizeRawTagLine(text)) per text (Program.cs:205-207) as a uint8 array."""
        bs = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in texts]
        offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        arena = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8).copy()
        bad = np.zeros(len(bs), dtype=np.uint8)
        self._L.orc_triage_lines(self._h, arena.ctypes.data, offsets.ctypes.data, len(bs), bad.ctypes.data)
        return bad

    def infer_action_tags(self, final_tags, evidence):
        r = self._L.orc_infer_action_tags(self._h, "\n".join(final_tags).encode(), "\n".join(evidence).encode()).decode()
        return r.split("\n") if r else []

    def format_caption(self, final_tags):
        return self._L.orc_format_caption(self._h, "\n".join(final_tags).encode()).decode()

    def select_tags_from_line(self, line, weight=1.0):
        b = line.encode("utf-8") if isinstance(line, str) else line
        r = self._L.orc_select_tags_from_line(self._h, b, len(b), weight).decode("utf-8")
        return r.split(", ") if r else []

    @property
    def minimal_fallback(self):
        return self._L.orc_minimal_fallback(self._h).decode()

    @property
    def trigger(self):
        return self._L.orc_trigger(self._h).decode()

    def rule_names(self):
        return json.loads(self._L.orc_rule_names_json(self._h).decode())

    # ---- batch runs ---------------------------------------------------------------------------
    def _collect(self, res, n, detail, audit=None):
        total = int(res.caption_offsets[n]) if n else 0
        caps = np.ctypeslib.as_array(res.captions, shape=(max(total, 1),))[:total].copy()
        offs = np.ctypeslib.as_array(res.caption_offsets, shape=(n + 1,)).copy()
        ver = np.frombuffer(C.string_at(res.verdicts, 16 * n), dtype=VERDICT_DTYPE).copy() if n else np.zeros(0, VERDICT_DTYPE)
        fw, nf = self.info.fired_words, self.info.n_facets
        fired = np.ctypeslib.as_array(res.fired, shape=(max(n * fw, 1),))[:n * fw].copy().reshape(n, fw)
        fc = np.ctypeslib.as_array(res.facet_choice, shape=(max(n * nf, 1),))[:n * nf].copy().reshape(n, nf)
        ln = C.c_uint64()
        p = self._L.orc_histogram_text(C.byref(res), C.byref(ln))
        hist = {}
        for line in C.string_at(p, ln.value).decode("utf-8").split("\n"):
            if line:
                t, c = line.rsplit("\t", 1)
                hist[t] = int(c)
        details = None
        if detail:
            details = [json.loads(self._L.orc_detail_json(C.byref(res), i).decode("utf-8")) for i in range(n)]
        out = OracleResult(caps, offs, ver, res.counters.as_dict(), fired, fc, hist, details)
        if audit is not None:
            # audit = {"names": [...], "vocab_file": str or None, "used_profile": str}: the audit.csv lines (Program.cs:966), no newline
            vf = audit.get("vocab_file")
            out.audit_rows = [self._L.orc_audit_row(C.byref(res), i, audit["names"][i].encode("utf-8"), None if vf is None else vf.encode("utf-8"),
                                                    audit["used_profile"].encode("utf-8")) for i in range(n)]
            if "image_paths" in audit:
                # review_candidates.jsonl / .csv lines (Program.cs:871-890), no newline; b"" where BuildReviewRecord returns null
                vfb = None if vf is None else vf.encode("utf-8")
                th = audit["thresholds"].encode("utf-8")
                out.review_jsonl, out.review_csv = [
                    [self._L.orc_review_row(self._h, C.byref(res), i, fmt, audit["image_paths"][i].encode("utf-8"), audit["caption_paths"][i].encode("utf-8"), th, vfb)
                     for i in range(n)] for fmt in (0, 1)]
        self._L.orc_release(C.byref(res))
        return out

    def run_lines(self, arena, offsets, threads=1, detail=False, audit=None):
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        res = orc_result()
        rc = self._L.orc_run_lines(self._h, arena.ctypes.data, offsets.ctypes.data, n, threads, int(detail or audit is not None), C.byref(res))
        if rc != 0:
            raise OracleError(self._L.orc_last_error().decode("utf-8", "replace"))
        return self._collect(res, n, detail, audit)

    def run_scored(self, arena, tag_offsets, tag_begin, scores, sources, threads=1, detail=False, audit=None):
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        tag_offsets = np.ascontiguousarray(tag_offsets, dtype=np.uint64)
        tag_begin = np.ascontiguousarray(tag_begin, dtype=np.uint64)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        sources = np.ascontiguousarray(sources, dtype=np.uint8)
        n = len(tag_begin) - 1
        res = orc_result()
        rc = self._L.orc_run_scored(self._h, arena.ctypes.data, tag_offsets.ctypes.data, tag_begin.ctypes.data,
                                    scores.ctypes.data, sources.ctypes.data, n, threads, int(detail or audit is not None), C.byref(res))
        if rc != 0:
            raise OracleError(self._L.orc_last_error().decode("utf-8", "replace"))
        return self._collect(res, n, detail, audit)

    def run_merged(self, arena, tag_offsets, tag_begin, secondary_begin, scores, secondary_cap, threads=1, detail=False, audit=None):
        """Two tagger replies per file, merged as Program.cs:214-221 does, then the pipeline."""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        tag_offsets = np.ascontiguousarray(tag_offsets, dtype=np.uint64)
        tag_begin = np.ascontiguousarray(tag_begin, dtype=np.uint64)
        secondary_begin = np.ascontiguousarray(secondary_begin, dtype=np.uint64)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        n = len(tag_begin) - 1
        res = orc_result()
        rc = self._L.orc_run_merged(self._h, arena.ctypes.data, tag_offsets.ctypes.data, tag_begin.ctypes.data, secondary_begin.ctypes.data,
                                    scores.ctypes.data, int(secondary_cap), n, threads, int(detail or audit is not None), C.byref(res))
        if rc != 0:
            raise OracleError(self._L.orc_last_error().decode("utf-8", "replace"))
        return self._collect(res, n, detail, audit)

    def run_replies(self, primary, secondary=None, secondary_cap=15, threads=1, detail=False, audit=None):
        """Cached tagger reply bodies (lists of bytes), one or two per file: CallJoyTagAsync from the body on, ParseJoyTags, the
        merge of Program.cs:214-221 when there are two, then the pipeline.  The result also carries .reply_status per file:
        0 parsed, 1 no "tags" array, 3 the reference throws on the reply."""
        pa, po = pack_lines(primary)
        n = len(po) - 1
        sa, so = pack_lines(secondary) if secondary is not None else (None, None)
        status = np.zeros(max(n, 1), dtype=np.uint8)
        res = orc_result()
        self._L.orc_run_replies.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_void_p,
                                            C.POINTER(orc_result)]
        rc = self._L.orc_run_replies(self._h, pa.ctypes.data if pa.size else None, po.ctypes.data, sa.ctypes.data if sa is not None and sa.size else None,
                                     so.ctypes.data if so is not None else None, int(secondary_cap), n, threads, int(detail or audit is not None),
                                     status.ctypes.data, C.byref(res))
        if rc != 0:
            raise OracleError(self._L.orc_last_error().decode("utf-8", "replace"))
        out = self._collect(res, n, detail, audit)
        out.reply_status = status[:n]
        return out

    def run_texts(self, texts, **kw):
        arena, offsets = pack_lines(texts)
        return self.run_lines(arena, offsets, **kw)


def pack_lines(texts):
    """Pack an iterable of str/bytes into (uint8 arena, uint64 offsets[n+1])."""
    bs = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in texts]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    arena = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return arena, offsets
