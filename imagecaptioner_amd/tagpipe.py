"""Host-side binding of libtagpipe_hip.so (the C ABI in include/tagpipe.h).

The library runs the tag pipeline on the GPU only.  If the shared object is
missing, or no HIP device is present, construction fails with an error: there is
no CPU path in this package.
"""
import ctypes as C
import os

import numpy as np

from ._abi import (tp_replies, TP_E_CAPACITY, TP_OK, VERDICT_DTYPE, Config, tp_audit, tp_config, tp_counters, tp_device_io, tp_result, tp_table_info,
                   tp_timing)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtagpipe_hip.so")
_LIB = None

EXPORTS = (
    "tp_config_init", "tp_create", "tp_destroy", "tp_last_error", "tp_get_table_info", "tp_get_string", "tp_get_name",
    "tp_run_lines", "tp_run_scored", "tp_release", "tp_run_lines_device", "tp_run_lines_device_noalloc", "tp_format_top_tags", "tp_get_last_timing",
    "tp_triage_lines", "tp_run_merged", "tp_format_fired_rules", "tp_needtag_text", "tp_format_side_files",
    "tp_decode_joytag_replies", "tp_release_replies", "tp_run_replies", "tp_set_audit", "tp_get_audit", "tp_audit_header", "tp_format_audit_rows", "tp_review_csv_header", "tp_format_review_rows",
    "tp_fallback_caption", "tp_error_needtag_text",
)


NOALLOC_MAX_FILES = 16384     # TP_NOALLOC_MAX_FILES


class TagPipeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("tagpipe error %d: %s" % (code, message))
        self.code = code


def load_library():
    """dlopen the in-tree library and declare its prototypes.  Raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("TAGPIPE_LIBRARY") or LIB_PATH     # the override is for diagnostic builds of the same library
    if not os.path.exists(path):
        raise TagPipeError(-3, "HIP extension not built: %s is missing (run `python -m imagecaptioner_amd.build`)" % path)
    try:
        # PyTorch ships its own copy of the HIP runtime.  A process that also uses torch on the GPU must end up with one runtime,
        # not two: with torch's libraries loaded first, this library binds to the same ones.  (Loaded the other way round, torch
        # later reports that no GPU is available.)
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    L.tp_config_init.argtypes = [C.POINTER(tp_config)]
    L.tp_config_init.restype = None
    L.tp_create.argtypes = [C.POINTER(tp_config)]
    L.tp_create.restype = C.c_void_p
    L.tp_destroy.argtypes = [C.c_void_p]
    L.tp_destroy.restype = None
    L.tp_last_error.restype = C.c_char_p
    L.tp_get_table_info.argtypes = [C.c_void_p, C.POINTER(tp_table_info)]
    L.tp_get_string.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32)]
    L.tp_get_name.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32)]
    L.tp_run_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(tp_result)]
    L.tp_run_scored.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(tp_result)]
    L.tp_release.argtypes = [C.c_void_p, C.POINTER(tp_result)]
    L.tp_release.restype = None
    L.tp_run_lines_device.argtypes = [C.c_void_p, C.POINTER(tp_device_io), C.POINTER(C.c_uint64)]
    L.tp_run_lines_device_noalloc.restype = C.c_int
    L.tp_run_lines_device_noalloc.argtypes = [C.c_void_p, C.POINTER(tp_device_io), C.POINTER(C.c_uint64)]
    L.tp_format_top_tags.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.tp_get_last_timing.argtypes = [C.c_void_p, C.POINTER(tp_timing)]
    L.tp_decode_joytag_replies.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(tp_replies)]
    L.tp_release_replies.argtypes = [C.POINTER(tp_replies)]
    L.tp_release_replies.restype = None
    L.tp_run_replies.argtypes = [C.c_void_p, C.POINTER(tp_replies), C.c_int, C.POINTER(tp_result)]
    L.tp_run_merged.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.POINTER(tp_result)]
    L.tp_format_fired_rules.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
    L.tp_needtag_text.restype = C.c_char_p
    L.tp_format_side_files.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
    L.tp_triage_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.tp_set_audit.argtypes = [C.c_void_p, C.c_int]
    L.tp_get_audit.argtypes = [C.POINTER(tp_result), C.POINTER(tp_audit)]
    L.tp_audit_header.restype = C.c_char_p
    L.tp_format_audit_rows.argtypes = [C.c_void_p, C.POINTER(tp_result), C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64,
                                       C.c_void_p, C.POINTER(C.c_uint64)]
    L.tp_review_csv_header.restype = C.c_char_p
    L.tp_format_review_rows.argtypes = [C.c_void_p, C.POINTER(tp_result), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p,
                                        C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
    L.tp_fallback_caption.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.tp_error_needtag_text.restype = C.c_char_p
    _LIB = L
    return L


class BatchResult:
    """Host copy of one batch's outputs."""

    def __init__(self, captions, caption_offsets, verdicts, counters, histogram, histogram_other, fired, facet_choice):
        self.captions = captions                # uint8[total]
        self.caption_offsets = caption_offsets  # uint64[n+1]
        self.verdicts = verdicts                # structured array, VERDICT_DTYPE
        self.counters = counters                # dict
        self.histogram = histogram              # uint64[n_strings]
        self.histogram_other = histogram_other
        self.fired = fired                      # uint32[n, fired_words]
        self.facet_choice = facet_choice        # uint16[n, n_facets]
        self.audit_lists = None                 # audit runs: per file (final_tags, evidence_tags, dropped_tags_with_reason) as bytes, ';'-joined,
                                                #             then (top tagger tags, unknown caption tags), "\n"-separated
        self.review_jsonl = None                # audit runs given image and caption paths: per file the review_candidates.jsonl / .csv line
        self.review_csv = None                  #             (bytes, newline included; b"" where the reference writes no record)
        self.audit_rows = None                  # audit runs: per file the audit.csv line (bytes, newline included)

    @property
    def n_files(self):
        return len(self.caption_offsets) - 1

    def caption(self, i):
        return bytes(self.captions[int(self.caption_offsets[i]):int(self.caption_offsets[i + 1])])


class TagPipe:
    """One GPU context: tables compiled from the reference's configuration files, resident on one device."""

    def __init__(self, config: Config):
        self._L = load_library()
        self._cfg = config
        self._h = self._L.tp_create(C.byref(config.struct))
        if not self._h:
            raise TagPipeError(-2, self._L.tp_last_error().decode("utf-8", "replace"))
        info = tp_table_info()
        self._check(self._L.tp_get_table_info(self._h, C.byref(info)))
        self.info = info
        self._strings = None

    def _check(self, rc):
        if rc != TP_OK:
            raise TagPipeError(rc, self._L.tp_last_error().decode("utf-8", "replace"))

    def close(self):
        if getattr(self, "_h", None):
            self._L.tp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tables -------------------------------------------------------------------------------
    def string(self, i):
        p, n = C.c_char_p(), C.c_uint32()
        self._check(self._L.tp_get_string(self._h, i, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value).decode("utf-8", "replace")

    def strings(self):
        if self._strings is None:
            self._strings = [self.string(i) for i in range(self.info.n_strings)]
        return self._strings

    def name(self, kind, index):
        p, n = C.c_char_p(), C.c_uint32()
        self._check(self._L.tp_get_name(self._h, kind, index, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value).decode("utf-8", "replace")

    def rule_names(self):
        i = self.info
        return {
            "actions": [self.name(0, r) for r in range(i.n_action_rules)],
            "subtypes": [self.name(1, s) for s in range(i.n_action_subtypes)],
            "facets": [{"name": self.name(2, f), "default": self.name(3, f)} for f in range(i.n_facets)],
        }

    def histogram_dict(self, histogram):
        """{tag: count} for the non-zero entries of a histogram array."""
        s = self.strings()
        nz = np.nonzero(np.asarray(histogram)[:len(s)])[0]
        return {s[int(i)]: int(histogram[int(i)]) for i in nz}

    def format_top_tags(self, histogram):
        hist = np.ascontiguousarray(histogram, dtype=np.uint64)
        need = C.c_uint64()
        rc = self._L.tp_format_top_tags(self._h, hist.ctypes.data, None, 0, C.byref(need))
        if rc not in (TP_OK, TP_E_CAPACITY):
            self._check(rc)
        buf = np.zeros(max(need.value, 1), dtype=np.uint8)
        if need.value:
            self._check(self._L.tp_format_top_tags(self._h, hist.ctypes.data, buf.ctypes.data, need.value, C.byref(need)))
        return bytes(buf[:need.value])

    def fired_rules(self, result):
        """tp_format_fired_rules: per file the ';'-joined firedRules string of the audit row (Program.cs:334, 966)."""
        n = len(result.verdicts)
        fired = np.ascontiguousarray(result.fired, dtype=np.uint32)
        fc = np.ascontiguousarray(result.facet_choice, dtype=np.uint16)
        offs = np.zeros(n + 1, dtype=np.uint64)
        need = C.c_uint64()
        rc = self._L.tp_format_fired_rules(self._h, fired.ctypes.data if fired.size else None, fc.ctypes.data if fc.size else None, n, None, 0,
                                           offs.ctypes.data, C.byref(need))
        buf = np.zeros(max(need.value, 1), dtype=np.uint8)
        if need.value:
            assert rc == TP_E_CAPACITY, rc
            self._check(self._L.tp_format_fired_rules(self._h, fired.ctypes.data if fired.size else None, fc.ctypes.data if fc.size else None, n,
                                                      buf.ctypes.data, need.value, offs.ctypes.data, C.byref(need)))
        else:
            self._check(rc)
        raw = buf.tobytes()
        return [raw[int(offs[i]):int(offs[i + 1])].decode("utf-8") for i in range(n)]

    def side_files(self, verdicts, keep_needtag):
        """tp_format_side_files: (needtag action per file, reject text per file) as the loop writes them (Program.cs:319-332)."""
        v = np.ascontiguousarray(verdicts)
        n = len(v)
        act = np.zeros(n, dtype=np.uint8)
        offs = np.zeros(n + 1, dtype=np.uint64)
        need = C.c_uint64()
        rc = self._L.tp_format_side_files(self._h, v.ctypes.data if n else None, n, int(keep_needtag), act.ctypes.data if n else None, None, 0,
                                          offs.ctypes.data, C.byref(need))
        buf = np.zeros(max(need.value, 1), dtype=np.uint8)
        if need.value:
            assert rc == TP_E_CAPACITY, rc
            self._check(self._L.tp_format_side_files(self._h, v.ctypes.data, n, int(keep_needtag), act.ctypes.data, buf.ctypes.data, need.value,
                                                     offs.ctypes.data, C.byref(need)))
        else:
            self._check(rc)
        raw = buf.tobytes()
        return act, [raw[int(offs[i]):int(offs[i + 1])] for i in range(n)]

    def needtag_text(self):
        return self._L.tp_needtag_text()

    def fallback_caption(self):
        """tp_fallback_caption: the caption line the reference writes for an image whose processing failed (Program.cs:344-350)."""
        need = C.c_uint64()
        rc = self._L.tp_fallback_caption(self._h, None, 0, C.byref(need))
        if rc not in (TP_OK, TP_E_CAPACITY):
            self._check(rc)
        buf = np.zeros(max(need.value, 1), dtype=np.uint8)
        self._check(self._L.tp_fallback_caption(self._h, buf.ctypes.data, need.value, C.byref(need)))
        return bytes(buf[:need.value])

    def error_needtag_text(self):
        return self._L.tp_error_needtag_text()

    def last_timing(self):
        t = tp_timing()
        self._check(self._L.tp_get_last_timing(self._h, C.byref(t)))
        return {"total_ms": t.total_ms, "select_ms": t.select_ms, "general_ms": t.general_ms, "emit_ms": t.emit_ms, "other_ms": t.other_ms,
                "general_files": int(t.general_files), "lean_front_ms": t.lean_front_ms, "lean_back_ms": t.lean_back_ms,
                "fused_ms": t.fused_ms, "fused_reruns": int(t.fused_reruns)}

    # ---- host-buffer runs -----------------------------------------------------------------------
    def audit_header(self):
        return self._L.tp_audit_header()

    def _collect_audit(self, res, audit):
        """tp_get_audit + tp_format_audit_rows on a result of an audit run; audit = {"names": [...], "vocab_file": str or None, "used_profile": str}."""
        n = int(res.n_files)
        a = tp_audit()
        self._check(self._L.tp_get_audit(C.byref(res), C.byref(a)))
        offs = np.ctypeslib.as_array(a.offsets, shape=(5 * n + 1,)).copy()
        text = C.string_at(a.text, int(offs[5 * n]))
        lists = [tuple(text[int(offs[5 * i + c]):int(offs[5 * i + c + 1])] for c in range(5)) for i in range(n)]
        names = [x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in audit["names"]]
        assert len(names) == n
        noffs = np.zeros(n + 1, dtype=np.uint64)
        if n:
            noffs[1:] = np.cumsum([len(b) for b in names], dtype=np.uint64)
        narena = np.frombuffer(b"".join(names) + b"\0", dtype=np.uint8).copy()
        vf = audit.get("vocab_file")
        vf = None if vf is None else vf.encode("utf-8")
        prof = audit["used_profile"].encode("utf-8")
        roffs = np.zeros(n + 1, dtype=np.uint64)
        need = C.c_uint64()
        rc = self._L.tp_format_audit_rows(self._h, C.byref(res), narena.ctypes.data, noffs.ctypes.data, vf, prof, None, 0, roffs.ctypes.data, C.byref(need))
        buf = np.zeros(max(need.value, 1), dtype=np.uint8)
        if need.value:
            assert rc == TP_E_CAPACITY, rc
            self._check(self._L.tp_format_audit_rows(self._h, C.byref(res), narena.ctypes.data, noffs.ctypes.data, vf, prof, buf.ctypes.data, need.value,
                                                     roffs.ctypes.data, C.byref(need)))
        else:
            self._check(rc)
        raw = buf.tobytes()
        return lists, [raw[int(roffs[i]):int(roffs[i + 1])] for i in range(n)]

    def review_csv_header(self):
        return self._L.tp_review_csv_header()

    def _collect_review(self, res, audit, fmt):
        """tp_format_review_rows; audit also holds "image_paths", "caption_paths" and "thresholds"."""
        n = int(res.n_files)

        def pack(strs):
            bs = [x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in strs]
            o = np.zeros(n + 1, dtype=np.uint64)
            if n:
                o[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
            return np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8).copy(), o
        ia, io_ = pack(audit["image_paths"])
        ca, co = pack(audit["caption_paths"])
        vf = audit.get("vocab_file")
        vf = None if vf is None else vf.encode("utf-8")
        th = audit["thresholds"].encode("utf-8")
        roffs = np.zeros(n + 1, dtype=np.uint64)
        need = C.c_uint64()
        rc = self._L.tp_format_review_rows(self._h, C.byref(res), fmt, ia.ctypes.data, io_.ctypes.data, ca.ctypes.data, co.ctypes.data, th, vf, None, 0,
                                           roffs.ctypes.data, C.byref(need))
        buf = np.zeros(max(need.value, 1), dtype=np.uint8)
        if need.value:
            assert rc == TP_E_CAPACITY, rc
            self._check(self._L.tp_format_review_rows(self._h, C.byref(res), fmt, ia.ctypes.data, io_.ctypes.data, ca.ctypes.data, co.ctypes.data, th, vf,
                                                      buf.ctypes.data, need.value, roffs.ctypes.data, C.byref(need)))
        else:
            self._check(rc)
        raw = buf.tobytes()
        return [raw[int(roffs[i]):int(roffs[i + 1])] for i in range(n)]

    def _collect(self, res, audit=None):
        n = int(res.n_files)
        total = int(res.caption_offsets[n]) if n else 0
        caps = np.ctypeslib.as_array(res.captions, shape=(max(total, 1),))[:total].copy()
        offs = np.ctypeslib.as_array(res.caption_offsets, shape=(n + 1,)).copy()
        ver = np.frombuffer(C.string_at(res.verdicts, 16 * n), dtype=VERDICT_DTYPE).copy() if n else np.zeros(0, VERDICT_DTYPE)
        ns, fw, nf = self.info.n_strings, self.info.fired_words, self.info.n_facets
        hist = np.ctypeslib.as_array(res.histogram, shape=(ns,)).copy()
        fired = np.ctypeslib.as_array(res.fired, shape=(max(n * fw, 1),))[:n * fw].copy().reshape(n, fw)
        fc = np.ctypeslib.as_array(res.facet_choice, shape=(max(n * nf, 1),))[:n * nf].copy().reshape(n, nf)
        out = BatchResult(caps, offs, ver, res.counters.as_dict(), hist, int(res.histogram_other), fired, fc)
        if audit is not None:
            out.audit_lists, out.audit_rows = self._collect_audit(res, audit)
            if "image_paths" in audit:
                out.review_jsonl, out.review_csv = self._collect_review(res, audit, 0), self._collect_review(res, audit, 1)
        self._L.tp_release(self._h, C.byref(res))
        return out

    def run_lines(self, arena, offsets, audit=None):
        """tp_run_lines: host arena + offsets in, host results out (uploads and downloads included).
        audit = {"names": [...], "vocab_file": str or None, "used_profile": str} switches the context to audit mode for the call
        (tp_set_audit) and fills the result's audit_lists and audit_rows."""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        res = tp_result()
        self._check(self._L.tp_set_audit(self._h, int(audit is not None)))
        try:
            self._check(self._L.tp_run_lines(self._h, arena.ctypes.data if arena.size else None, offsets.ctypes.data, len(offsets) - 1, C.byref(res)))
        finally:
            self._L.tp_set_audit(self._h, 0)
        return self._collect(res, audit)

    def run_texts(self, texts, audit=None):
        bs = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in texts]
        offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        arena = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
        return self.run_lines(arena, offsets, audit=audit)

    def triage_lines(self, arena, offsets):
        """tp_triage_lines: which existing captions the reference would delete and redo (Program.cs:203-210).
        Returns a uint8 array: 1 bad, 0 keep, 2 outside what the device code handles."""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        bad = np.zeros(len(offsets) - 1, dtype=np.uint8)
        self._check(self._L.tp_triage_lines(self._h, arena.ctypes.data if arena.size else None, offsets.ctypes.data, len(offsets) - 1,
                                            bad.ctypes.data if bad.size else None))
        return bad

    def triage_texts(self, texts):
        bs = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in texts]
        offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        arena = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
        return self.triage_lines(arena, offsets)

    def run_scored(self, arena, tag_offsets, tag_begin, scores, sources, audit=None):
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        tag_offsets = np.ascontiguousarray(tag_offsets, dtype=np.uint64)
        tag_begin = np.ascontiguousarray(tag_begin, dtype=np.uint64)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        sources = np.ascontiguousarray(sources, dtype=np.uint8)
        res = tp_result()
        self._check(self._L.tp_set_audit(self._h, int(audit is not None)))
        try:
            self._check(self._L.tp_run_scored(self._h, arena.ctypes.data if arena.size else None, tag_offsets.ctypes.data, tag_begin.ctypes.data,
                                              scores.ctypes.data if scores.size else None, sources.ctypes.data if sources.size else None,
                                              len(tag_begin) - 1, C.byref(res)))
        finally:
            self._L.tp_set_audit(self._h, 0)
        return self._collect(res, audit)

    def run_merged(self, arena, tag_offsets, tag_begin, secondary_begin, scores, secondary_cap=15, audit=None):
        """tp_run_merged: primary and secondary tagger replies per file, merged as Program.cs:214-221 does."""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        tag_offsets = np.ascontiguousarray(tag_offsets, dtype=np.uint64)
        tag_begin = np.ascontiguousarray(tag_begin, dtype=np.uint64)
        secondary_begin = np.ascontiguousarray(secondary_begin, dtype=np.uint64)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        res = tp_result()
        self._check(self._L.tp_set_audit(self._h, int(audit is not None)))
        try:
            self._check(self._L.tp_run_merged(self._h, arena.ctypes.data if arena.size else None, tag_offsets.ctypes.data, tag_begin.ctypes.data,
                                              secondary_begin.ctypes.data if secondary_begin.size else None, scores.ctypes.data if scores.size else None,
                                              int(secondary_cap), len(tag_begin) - 1, C.byref(res)))
        finally:
            self._L.tp_set_audit(self._h, 0)
        return self._collect(res, audit)

    # ---- cached tagger replies ---------------------------------------------------------------------
    def decode_joytag_replies(self, primary, secondary=None):
        """tp_decode_joytag_replies: lists of JSON reply bodies (bytes) -> dict of the arrays run_scored / run_merged take
        (arena, tag_offsets, tag_begin, secondary_begin or None, scores, sources) plus status and has_score per file."""
        def pack(blobs):
            offs = np.zeros(len(blobs) + 1, dtype=np.uint64)
            if blobs:
                offs[1:] = np.cumsum([len(b) for b in blobs])
            data = np.frombuffer(b"".join(blobs), dtype=np.uint8).copy() if offs[-1] else np.zeros(1, np.uint8)
            return data, offs
        n = len(primary)
        pa, po = pack(list(primary))
        sa, so = pack(list(secondary)) if secondary is not None else (None, None)
        if secondary is not None and len(secondary) != n:
            raise ValueError("one secondary reply per primary reply")
        r = tp_replies()
        self._check(self._L.tp_decode_joytag_replies(self._h, pa.ctypes.data, po.ctypes.data, sa.ctypes.data if sa is not None else None,
                                                     so.ctypes.data if so is not None else None, n, C.byref(r)))
        try:
            nt = int(r.n_tags)
            grab = lambda ptr, count, dt: np.ctypeslib.as_array(ptr, shape=(count,)).astype(dt, copy=True) if count else np.zeros(0, dt)
            offs = grab(r.tag_offsets, nt + 1, np.uint64)
            out = {"arena": grab(r.arena, int(offs[-1]), np.uint8), "tag_offsets": offs, "tag_begin": grab(r.tag_begin, n + 1, np.uint64),
                   "secondary_begin": grab(r.secondary_begin, n, np.uint64) if secondary is not None else None,
                   "scores": grab(r.scores, nt, np.float64), "sources": grab(r.sources, nt, np.uint8),
                   "status": grab(r.status, n, np.uint8), "has_score": grab(r.has_score, n, np.uint8)}
        finally:
            self._L.tp_release_replies(C.byref(r))
        return out

    def run_replies(self, decoded, secondary_cap=15, audit=None):
        """tp_run_replies on what decode_joytag_replies returned: one reply per file or two, each kept in the order
        ParseJoyTags leaves it in (sorted only when an element carried a score, Program.cs:487)."""
        keep = {k: np.ascontiguousarray(decoded[k], dtype=dt) for k, dt in (("arena", np.uint8), ("tag_offsets", np.uint64), ("tag_begin", np.uint64),
                                                                             ("scores", np.float64), ("sources", np.uint8), ("has_score", np.uint8))}
        sec = None if decoded.get("secondary_begin") is None else np.ascontiguousarray(decoded["secondary_begin"], dtype=np.uint64)
        ptr = lambda a, ct: C.cast(a.ctypes.data, C.POINTER(ct)) if a is not None and a.size else None
        r = tp_replies()
        r.n_files = len(keep["tag_begin"]) - 1
        r.n_tags = int(keep["tag_begin"][-1])
        r.arena = ptr(keep["arena"], C.c_uint8); r.tag_offsets = ptr(keep["tag_offsets"], C.c_uint64); r.tag_begin = ptr(keep["tag_begin"], C.c_uint64)
        if sec is not None and not sec.size:
            sec = np.zeros(1, np.uint64)                       # never a null pointer: null means "one reply per file"
        r.secondary_begin = None if sec is None else C.cast(sec.ctypes.data, C.POINTER(C.c_uint64))
        r.scores = ptr(keep["scores"], C.c_double); r.sources = ptr(keep["sources"], C.c_uint8)
        r.status = None; r.has_score = ptr(keep["has_score"], C.c_uint8); r.opaque = None
        res = tp_result()
        self._check(self._L.tp_set_audit(self._h, int(audit is not None)))
        try:
            self._check(self._L.tp_run_replies(self._h, C.byref(r), int(secondary_cap), C.byref(res)))
        finally:
            self._L.tp_set_audit(self._h, 0)
        return self._collect(res, audit)

    # ---- device-resident run ---------------------------------------------------------------------
    def run_lines_device(self, io: tp_device_io, noalloc=False):
        """tp_run_lines_device: every pointer in `io` is a device pointer.  Returns the caption byte count;
        raises TagPipeError(code TP_E_CAPACITY) when the caption buffer is too small.  noalloc=True calls
        tp_run_lines_device_noalloc instead (at most NOALLOC_MAX_FILES files, no device allocation during the call)."""
        nbytes = C.c_uint64()
        rc = (self._L.tp_run_lines_device_noalloc if noalloc else self._L.tp_run_lines_device)(self._h, C.byref(io), C.byref(nbytes))
        if rc != TP_OK:
            err = TagPipeError(rc, self._L.tp_last_error().decode("utf-8", "replace"))
            err.required = nbytes.value
            raise err
        return nbytes.value
