"""Shared test utilities: oracle / hostsim bindings, result comparison, edge-case lines."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from imagecaptioner_amd._abi import VERDICT_DTYPE, Config, tp_config, tp_counters, tp_table_info, tp_verdict  # noqa: E402
from oracle.oracle_py import Oracle, pack_lines  # noqa: E402,F401

DATA = os.path.join(ROOT, "tests", "golden", "reference_data")


def shipped_config(**kw):
    """The reference's shipped tables (rules/*.json, profiles/sgrui.json, top_tags.txt)."""
    base = dict(lexicon=os.path.join(DATA, "rules", "lexicon.json"), actions=os.path.join(DATA, "rules", "actions.json"),
                required_facets=os.path.join(DATA, "rules", "required_facets.json"), profile=os.path.join(DATA, "profiles", "sgrui.json"),
                vocabulary=os.path.join(DATA, "top_tags.txt"), minimal_subject_tags="1girl, solo")
    base.update(kw)
    return Config(**base)


# ---- test-only host build of the kernels' per-file pipeline ---------------------------------------
class sim_result(C.Structure):
    _fields_ = [("n_files", C.c_uint64), ("captions", C.POINTER(C.c_uint8)), ("offsets", C.POINTER(C.c_uint64)),
                ("verdicts", C.POINTER(tp_verdict)), ("counters", tp_counters), ("histogram", C.POINTER(C.c_uint64)),
                ("fired", C.POINTER(C.c_uint32)), ("facets", C.POINTER(C.c_uint16)), ("opaque", C.c_void_p)]


_SIM = None


def sim_lib():
    global _SIM
    if _SIM is None:
        d = os.path.join(ROOT, "tests", "hostsim")
        subprocess.check_call(["make", "-C", d, "-s"])
        L = C.CDLL(os.path.join(d, "libtp_hostsim.so"))
        L.sim_last_error.restype = C.c_char_p
        L.sim_create.restype = C.c_void_p
        L.sim_create.argtypes = [C.POINTER(tp_config)]
        L.sim_create_learned.restype = C.c_void_p
        L.sim_create_learned.argtypes = [C.POINTER(tp_config), C.c_char_p, C.c_uint64]
        L.sim_destroy.argtypes = [C.c_void_p]
        L.sim_info.argtypes = [C.c_void_p, C.POINTER(tp_table_info)]
        L.sim_string.restype = C.c_char_p
        L.sim_string.argtypes = [C.c_void_p, C.c_uint32]
        L.sim_run_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(sim_result)]
        L.sim_run_scored.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(sim_result)]
        L.sim_run_lines_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(sim_result), C.POINTER(C.c_uint64)]
        L.sim_run_lines_lean.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(sim_result), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.sim_release.argtypes = [C.POINTER(sim_result)]
        L.sim_run_scored_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64,
                                          C.POINTER(sim_result), C.POINTER(C.c_uint64)]
        L.sim_run_merged.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.POINTER(sim_result)]
        L.sim_triage_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
        L.sim_run_audit.restype = C.c_longlong
        L.sim_run_audit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.sim_run_audit_flags.restype = C.c_longlong
        L.sim_run_audit_flags.argtypes = L.sim_run_audit.argtypes + [C.c_void_p]
        _SIM = L
    return _SIM


class SimResult:
    def __init__(self, captions, offsets, verdicts, counters, histogram, histogram_other, fired, facet_choice):
        self.captions, self.caption_offsets, self.verdicts, self.counters = captions, offsets, verdicts, counters
        self.histogram, self.histogram_other, self.fired, self.facet_choice = histogram, histogram_other, fired, facet_choice

    def caption(self, i):
        return bytes(self.captions[int(self.caption_offsets[i]):int(self.caption_offsets[i + 1])])


class HostSim:
    """The kernel's per-file pipeline source compiled for the host (tests/hostsim) — logic checks only."""

    def __init__(self, config, learned=None):
        """learned: strings interned behind the configuration's own, as the library does with the strings in no table that it
        meets in its input."""
        self._cfg = config
        self._L = sim_lib()
        if learned is None:
            self._h = self._L.sim_create(C.byref(config.struct))
        else:
            blob = "\n".join(learned).encode("utf-8")
            self._h = self._L.sim_create_learned(C.byref(config.struct), blob, len(blob))
        if not self._h:
            raise RuntimeError(self._L.sim_last_error().decode("utf-8", "replace"))
        self.info = tp_table_info()
        self._L.sim_info(self._h, C.byref(self.info))
        self._strings = None

    def strings(self):
        if self._strings is None:
            self._strings = [self._L.sim_string(self._h, i).decode("utf-8", "replace") for i in range(self.info.n_strings)]
        return self._strings

    def histogram_dict(self, histogram):
        s = self.strings()
        nz = np.nonzero(np.asarray(histogram)[:len(s)])[0]
        return {s[int(i)]: int(histogram[int(i)]) for i in nz}

    def decode_joytag_replies(self, primary, secondary=None):
        """The library's reply decoder on the host build, in TagPipe.decode_joytag_replies' output form."""
        self._L.sim_decode_reply.restype = C.c_int64
        self._L.sim_decode_reply.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        n = len(primary)
        arena, offs, scores, sources, begin, mid = [], [0], [], [], [0], []
        status, has = np.zeros(n, np.uint8), np.zeros(n, np.uint8)

        def one(blob, source):
            blob = bytes(blob)
            cap = len(blob) + 16
            a, o, sc = np.zeros(cap, np.uint8), np.zeros(cap + 1, np.uint64), np.zeros(cap, np.float64)
            st, hs = C.c_uint8(), C.c_uint8()
            k = self._L.sim_decode_reply(self._h, blob, len(blob), a.ctypes.data, cap, o.ctypes.data, sc.ctypes.data, cap, C.byref(st), C.byref(hs))
            assert k >= 0
            base = offs[-1]
            arena.append(a[:int(o[k])].copy())
            offs.extend(int(x) + base for x in o[1:k + 1])
            scores.extend(sc[:k])
            sources.extend([source] * k)
            return st.value, hs.value
        for i in range(n):
            status[i], has[i] = one(primary[i], 0)
            if secondary is not None:
                mid.append(len(scores))
                st, hs = one(secondary[i], 1)
                status[i] = max(status[i], st)
                has[i] |= hs << 1
            begin.append(len(scores))
        return {"arena": np.concatenate(arena) if arena else np.zeros(0, np.uint8), "tag_offsets": np.array(offs, dtype=np.uint64),
                "tag_begin": np.array(begin, dtype=np.uint64), "secondary_begin": np.array(mid, dtype=np.uint64) if secondary is not None else None,
                "scores": np.array(scores, dtype=np.float64), "sources": np.array(sources, dtype=np.uint8), "status": status, "has_score": has}

    def format_top_tags(self, histogram):
        hist = np.ascontiguousarray(histogram, dtype=np.uint64)
        assert len(hist) >= self.info.n_strings
        self._L.sim_format_top_tags.restype = C.c_uint64
        self._L.sim_format_top_tags.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        need = self._L.sim_format_top_tags(self._h, hist.ctypes.data, None, 0)
        buf = np.zeros(max(need, 1), dtype=np.uint8)
        self._L.sim_format_top_tags(self._h, hist.ctypes.data, buf.ctypes.data, need)
        return bytes(buf[:need])

    def _collect(self, res, rc):
        if rc != 0:
            raise RuntimeError(self._L.sim_last_error().decode())
        n = int(res.n_files)
        total = int(res.offsets[n]) if n else 0
        caps = np.ctypeslib.as_array(res.captions, shape=(total,)).copy() if total else np.zeros(0, np.uint8)
        offs = np.ctypeslib.as_array(res.offsets, shape=(n + 1,)).copy()
        ver = np.frombuffer(C.string_at(res.verdicts, 16 * n), dtype=VERDICT_DTYPE).copy() if n else np.zeros(0, VERDICT_DTYPE)
        ns, fw, nf = self.info.n_strings, self.info.fired_words, self.info.n_facets
        hist = np.ctypeslib.as_array(res.histogram, shape=(ns + 1,)).copy()
        fired = np.ctypeslib.as_array(res.fired, shape=(max(n * fw, 1),))[:n * fw].copy().reshape(n, fw)
        fc = np.ctypeslib.as_array(res.facets, shape=(max(n * nf, 1),))[:n * nf].copy().reshape(n, nf)
        out = SimResult(caps, offs, ver, res.counters.as_dict(), hist[:ns], int(hist[ns]), fired, fc)
        self._L.sim_release(C.byref(res))
        return out

    def run_lines(self, arena, offsets):
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        pad = np.concatenate([arena, np.zeros(16, np.uint8)])
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        res = sim_result()
        rc = self._L.sim_run_lines(self._h, pad.ctypes.data, offsets.ctypes.data, len(offsets) - 1, C.byref(res))
        return self._collect(res, rc)

    def run_texts(self, texts):
        a, o = pack_lines(texts)
        return self.run_lines(a, o)

    def run_lines_fast(self, arena, offsets):
        """The GPU flow: wave-per-file kernel body on an emulated wavefront, general pipeline for declined files."""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        if arena.size == 0:
            arena = np.zeros(1, np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        res = sim_result()
        ng = C.c_uint64()
        rc = self._L.sim_run_lines_fast(self._h, arena.ctypes.data, offsets.ctypes.data, len(offsets) - 1, C.byref(res), C.byref(ng))
        out = self._collect(res, rc)
        out.n_general = ng.value
        return out

    def run_texts_fast(self, texts):
        a, o = pack_lines(texts)
        return self.run_lines_fast(a, o)

    def run_lines_lean(self, arena, offsets, details=True):
        """The GPU flow with the lean stage in front (as the library runs it).  details=False: the run of a caller that
        takes no detail arrays (the wave stages then do not count the evidence set: n_evidence is 0xFFFF for their files)."""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        if arena.size == 0:
            arena = np.zeros(1, np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        res = sim_result()
        ng, nl = C.c_uint64(), C.c_uint64()
        rc = self._L.sim_run_lines_lean(self._h, arena.ctypes.data, offsets.ctypes.data, len(offsets) - 1, 1 if details else 0, C.byref(res), C.byref(ng), C.byref(nl))
        out = self._collect(res, rc)
        out.n_general, out.n_lean = ng.value, nl.value
        return out

    def run_texts_lean(self, texts, details=True):
        a, o = pack_lines(texts)
        return self.run_lines_lean(a, o, details)

    def run_merged(self, arena, tag_offsets, tag_begin, secondary_begin, scores, secondary_cap):
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        pad = np.concatenate([arena, np.zeros(16, np.uint8)])
        tag_offsets = np.ascontiguousarray(tag_offsets, dtype=np.uint64)
        tag_begin = np.ascontiguousarray(tag_begin, dtype=np.uint64)
        secondary_begin = np.ascontiguousarray(secondary_begin, dtype=np.uint64)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        res = sim_result()
        rc = self._L.sim_run_merged(self._h, pad.ctypes.data, tag_offsets.ctypes.data, tag_begin.ctypes.data, secondary_begin.ctypes.data,
                                    scores.ctypes.data, int(secondary_cap), len(tag_begin) - 1, C.byref(res))
        return self._collect(res, rc)

    def run_scored_fast(self, arena, tag_offsets, tag_begin, scores, sources=None, secondary_begin=None, secondary_cap=0):
        """The scored entry points as the library runs them: wave kernel with the scored front, general pipeline for the rest."""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        tag_offsets = np.ascontiguousarray(tag_offsets, dtype=np.uint64)
        tag_begin = np.ascontiguousarray(tag_begin, dtype=np.uint64)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        sources = np.ascontiguousarray(sources if sources is not None else np.zeros(len(scores), np.uint8), dtype=np.uint8)
        sb = np.ascontiguousarray(secondary_begin, dtype=np.uint64) if secondary_begin is not None else None
        res = sim_result()
        ng = C.c_uint64()
        rc = self._L.sim_run_scored_fast(self._h, arena.ctypes.data, tag_offsets.ctypes.data, tag_begin.ctypes.data, sb.ctypes.data if sb is not None else None,
                                         scores.ctypes.data, sources.ctypes.data, int(secondary_cap), len(tag_begin) - 1, C.byref(res), C.byref(ng))
        out = self._collect(res, rc)
        out.n_general = ng.value
        return out

    def triage_texts(self, texts, general_only=False):
        """(bad, n_general): the triage flow on the emulated wavefront; general_only runs the thread-per-file code alone."""
        bs = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in texts]
        offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        arena = np.frombuffer(b"".join(bs) + b"\0" * 16, dtype=np.uint8).copy()
        bad = np.zeros(len(bs), dtype=np.uint8)
        ng = C.c_uint64()
        rc = self._L.sim_triage_lines(self._h, arena.ctypes.data, offsets.ctypes.data, len(bs), int(general_only), bad.ctypes.data, C.byref(ng))
        if rc != 0:
            raise RuntimeError(self._L.sim_last_error().decode())
        return bad, ng.value

    def run_audit(self, arena, offsets, tag_begin=None, secondary_begin=None, scores=None, sources=None, secondary_cap=0, reply_flags=None):
        """The audit row's tag lists from the general pipeline in audit mode: (lists, verdicts); lists[i] = (final_tags, evidence_tags,
        dropped_tags_with_reason) as bytes, ';'-joined.  Line ingest unless tag_begin is given.  reply_flags: has_score per file
        (tp_run_replies: a reply without scores keeps its order)."""
        arena = np.concatenate([np.ascontiguousarray(arena, dtype=np.uint8), np.zeros(16, np.uint8)])
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ptr = lambda a, dt: None if a is None else np.ascontiguousarray(a, dtype=dt)
        tb, sb, sc, sr = ptr(tag_begin, np.uint64), ptr(secondary_begin, np.uint64), ptr(scores, np.float64), ptr(sources, np.uint8)
        rf = ptr(reply_flags, np.uint8)
        n = (len(tb) if tb is not None else len(offsets)) - 1
        offs3 = np.zeros(5 * n + 1, dtype=np.uint64)
        ver = np.zeros(n, dtype=VERDICT_DTYPE)
        self.review_bits = np.zeros(n, dtype=np.uint32)
        cap = 1 << 20
        while True:
            text = np.zeros(cap, dtype=np.uint8)
            d = lambda a: None if a is None else a.ctypes.data
            rc = self._L.sim_run_audit_flags(self._h, arena.ctypes.data, offsets.ctypes.data, d(tb), d(sb), d(sc), d(sr), int(secondary_cap), n, text.ctypes.data, cap,
                                             offs3.ctypes.data, ver.ctypes.data, self.review_bits.ctypes.data, d(rf))
            if rc == -1:
                cap *= 4
                continue
            assert rc >= 0, "audit list sizes reported by the pipeline differ from the emitted text"
            break
        raw = text.tobytes()
        return [tuple(raw[int(offs3[5 * i + c]):int(offs3[5 * i + c + 1])] for c in range(5)) for i in range(n)], ver

    def run_audit_wave(self, arena, offsets):
        """The audit lists as the wave stages leave them (lean stage, then the general wave stage): (lists, verdicts, stage_of);
        lists[i] = the five columns as bytes (';'-joined, the review record's two newline-separated); stage_of[i] = 0 for a file
        none of them takes (its lists are then empty: the thread-per-file kernel's file)."""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        offs5 = np.zeros(5 * n + 1, dtype=np.uint64)
        ver = np.zeros(n, dtype=VERDICT_DTYPE)
        self.review_bits = np.zeros(n, dtype=np.uint32)
        stage_of = np.zeros(n, dtype=np.uint8)
        cap = 1 << 20
        fn = self._L.sim_run_audit_wave
        fn.restype = C.c_longlong
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        while True:
            text = np.zeros(cap, dtype=np.uint8)
            rc = fn(self._h, arena.ctypes.data, offsets.ctypes.data, n, text.ctypes.data, cap, offs5.ctypes.data, ver.ctypes.data, self.review_bits.ctypes.data,
                    stage_of.ctypes.data)
            if rc == -1:
                cap *= 4
                continue
            assert rc >= 0, "audit list sizes reported by a wave stage differ from the text of its ids: " + self._L.sim_last_error().decode()
            break
        raw = text.tobytes()
        return [tuple(raw[int(offs5[5 * i + c]):int(offs5[5 * i + c + 1])] for c in range(5)) for i in range(n)], ver, stage_of

    def run_audit_scored_wave(self, arena, tag_offsets, tag_begin, scores, sources=None, secondary_begin=None, secondary_cap=0):
        """Tagger replies in audit mode as the wave kernel leaves them: (lists, verdicts, stage_of), as run_audit_wave gives them."""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        tag_offsets = np.ascontiguousarray(tag_offsets, dtype=np.uint64)
        tag_begin = np.ascontiguousarray(tag_begin, dtype=np.uint64)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        sources = np.ascontiguousarray(sources if sources is not None else np.zeros(len(scores), np.uint8), dtype=np.uint8)
        sb = np.ascontiguousarray(secondary_begin, dtype=np.uint64) if secondary_begin is not None else None
        n = len(tag_begin) - 1
        offs5 = np.zeros(5 * n + 1, dtype=np.uint64)
        ver = np.zeros(n, dtype=VERDICT_DTYPE)
        self.review_bits = np.zeros(n, dtype=np.uint32)
        stage_of = np.zeros(n, dtype=np.uint8)
        cap = 1 << 20
        fn = self._L.sim_run_audit_scored_wave
        fn.restype = C.c_longlong
        fn.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        while True:
            text = np.zeros(cap, dtype=np.uint8)
            rc = fn(self._h, arena.ctypes.data, tag_offsets.ctypes.data, tag_begin.ctypes.data, sb.ctypes.data if sb is not None else None, scores.ctypes.data,
                    sources.ctypes.data, int(secondary_cap), n, text.ctypes.data, cap, offs5.ctypes.data, ver.ctypes.data, self.review_bits.ctypes.data, stage_of.ctypes.data)
            if rc == -1:
                cap *= 4
                continue
            assert rc >= 0, "audit list sizes reported by the scored wave kernel differ from the text of its ids: " + self._L.sim_last_error().decode()
            break
        raw = text.tobytes()
        return [tuple(raw[int(offs5[5 * i + c]):int(offs5[5 * i + c + 1])] for c in range(5)) for i in range(n)], ver, stage_of

    def run_scored(self, arena, tag_offsets, tag_begin, scores, sources):
        arena = np.concatenate([np.ascontiguousarray(arena, dtype=np.uint8), np.zeros(16, np.uint8)])
        tag_offsets = np.ascontiguousarray(tag_offsets, dtype=np.uint64)
        tag_begin = np.ascontiguousarray(tag_begin, dtype=np.uint64)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        sources = np.ascontiguousarray(sources, dtype=np.uint8)
        res = sim_result()
        rc = self._L.sim_run_scored(self._h, arena.ctypes.data, tag_offsets.ctypes.data, tag_begin.ctypes.data, scores.ctypes.data,
                                    sources.ctypes.data, len(tag_begin) - 1, C.byref(res))
        return self._collect(res, rc)


# ---- comparison -------------------------------------------------------------------------------------
def oracle_audit_lists(want):
    """(final_tags, evidence_tags, dropped_tags_with_reason) per file, ';'-joined bytes, from an oracle result run with detail."""
    return [tuple(";".join(d[k]).encode("utf-8") for k in ("final_tags", "evidence", "dropped")) for d in want.details]


BANNED_SORTED = ["1boy", "2boys", "3boys", "4boys", "boy", "male", "male_focus", "multiple_boys", "yaoi"]   # Program.cs:33-35 under OrderBy(OrdinalIgnoreCase)


def check_review_facts(lists, review_bits, verdicts, want):
    """The device-side facts behind a review record (banned-token bits, subject-token bits -> reasons; top tagger tags; unknown caption
    tags) against the oracle's record, taken from its JSON line (want.review_jsonl, b"" where the reference writes none)."""
    import json
    for i, line in enumerate(want.review_jsonl):
        rb = int(review_bits[i])
        banned = [BANNED_SORTED[k] for k in range(9) if rb >> k & 1]
        male, girl, multi, low = bool(banned) or bool(rb >> 16 & 1), bool(rb >> 17 & 1), bool(rb >> 18 & 1), bool(int(verdicts["flags"][i]) & 1)
        reasons = (["male_subject_token"] if male else []) + (["mixed_subject_tokens"] if male and girl else []) + (["multi_person"] if multi else []) + \
                  (["short_caption"] if low else [])
        top = lists[i][3].decode("utf-8").split("\n") if lists[i][3] else []
        unknown = lists[i][4].decode("utf-8").split("\n") if lists[i][4] else []
        if not line:
            assert not reasons and not banned and not unknown and int(verdicts["coverage"][i]) == 0x3F, i
            continue
        rec = json.loads(line.decode("utf-8"))
        assert rec["Reasons"] == reasons, (i, rec["Reasons"], reasons)
        assert rec["BannedTokensDetected"] == banned and rec["DroppedTags"] == banned, (i, rec, banned)
        assert rec["TopJoyTags"] == top, (i, rec["TopJoyTags"], top)
        assert sorted(rec["UnknownFinalTags"]) == sorted(unknown), (i, rec["UnknownFinalTags"], unknown)
        assert rec["FinalTagCount"] == int(verdicts["n_selected"][i])


def assert_same_results(got, want, hist_of_got, what="", check_hist=True):
    """got: BatchResult/SimResult of the implementation under test; want: OracleResult.  Bit-exact on everything."""
    n = len(want.caption_offsets) - 1
    assert len(got.caption_offsets) - 1 == n, what
    if not np.array_equal(got.caption_offsets, want.caption_offsets) or not np.array_equal(got.captions, want.captions):
        for i in range(n):
            assert got.caption(i) == want.caption(i), "%s file %d:\n got  %r\n want %r" % (what, i, got.caption(i), want.caption(i))
    for name in got.verdicts.dtype.names:
        if not np.array_equal(got.verdicts[name], want.verdicts[name]):
            i = int(np.nonzero(got.verdicts[name] != want.verdicts[name])[0][0])
            raise AssertionError("%s verdict.%s differs at file %d: got %r want %r (caption %r)" %
                                 (what, name, i, got.verdicts[name][i], want.verdicts[name][i], want.caption(i)))
    assert np.array_equal(got.fired, want.fired), what + " fired bitmap"
    assert np.array_equal(got.facet_choice, want.facet_choice), what + " facet_choice"
    for k, v in want.counters.items():
        assert got.counters[k] == v, "%s counter %s: got %d want %d" % (what, k, got.counters[k], v)
    if check_hist:
        gh = dict(hist_of_got)
        other = got.histogram_other
        wh = dict(want.histogram)
        if other:
            # tags outside the tables are pooled in one counter; the oracle counts them by name
            known = set(gh)
            pooled = sum(c for t, c in wh.items() if t not in known)
            assert pooled == other, "%s histogram_other: got %d want %d" % (what, other, pooled)
            wh = {t: c for t, c in wh.items() if t in known}
        assert gh == wh, "%s histogram differs: %r" % (what, sorted(set(gh.items()) ^ set(wh.items()))[:10])


EDGE_LINES = [
    "",
    "   \t  ",
    "1girl, solo, blue_hair, long_hair, blue_eyes, smile, school_uniform, standing, classroom, masterpiece, best_quality, highres, watermark",
    "1girl",
    "1girl, solo",
    "a, b, c",
    "solo, solo, Solo, SOLO , solo.",
    "1girl, barefoot, penis, long hair, Blue Eyes., smile, dress, sitting, bedroom, toes",
    "sorry, I can't help with that",
    "this is one long sentence without any commas that goes past forty characters",
    "short, line",
    "POSITIVE: 1girl, smile, dress\nNEGATIVE: lowres, bad_hands",
    "Here you go:\npositive:  1girl, long_hair, skirt  \nnegative: text",
    "1girl, smile, dress NEGATIVE: watermark, text",
    "```\n1girl, smile, dress, hat\n```",
    "```json\n1girl, smile, dress, hat```",
    "```1girl, smile, dress",
    "1girl,\r\nsmile,\r\ndress,\r\nhat",
    "1girl,,, ,smile,  ,dress,",
    ",,,",
    "'1girl', \"smile\", `dress`, 'hat'.",
    "x'., y';, z\"., 1girl, smile, dress",
    "' x', 'a b'  , \" q \" , 1girl, smile",
    "long  hair , red__eyes, blue _ ribbon,  1girl",
    "pedal stimulation, feet, penis, 1girl, between_feet",
    "foot, soles, fingers, licking, penis, 1girl",
    "fellatio, penis, 1girl, smile, dress",
    "rating:safe, rating, watermark, signature, 1girl, smile, dress",
    "1boy, male_focus, 1girl, smile, dress",
    ":bad, -bad, _bad, ok_tag, (paren), a!b, a/b, a-b, a'b",
    "nsfw, nude, 1girl, smile, dress",
    "1girl, hand, penis, grip, vaginal_penetration",
    "penis, vaginal, missionary, 1girl",
    "1girl, looking_at_viewer, open_mouth, tongue_out, cum, ejaculation, panties_aside, male_pov",
    "I'm not able to do that, 1girl, smile",
    "as an AI, i cannot, 1girl",
    "tag;, tag2..;, tag3;.;, 1girl",
    "A, B",
    "i\ncan't, 1girl, smile",
    "1girl, " + ", ".join("tag_%d" % i for i in range(120)),
    ", ".join(["blue_hair", "red_hair", "long_hair", "short_hair", "twintails", "ponytail", "braid", "ahoge", "bangs", "blush", "smile", "frown",
               "dress", "skirt", "shirt", "jacket", "coat", "hat", "ribbon", "bow", "boots", "gloves", "standing", "sitting", "lying",
               "kneeling", "walking", "running", "indoors", "outdoors", "bedroom", "street", "city", "park", "lighting", "shadow",
               "bloom", "bokeh", "highres", "absurdres", "masterpiece", "detailed", "cat", "dog", "sword", "flower", "book", "cup", "star", "moon"]),
    "masterpiece, best_quality, highres, absurdres, detailed, monochrome, 1girl",
    "1girl, smile\n",
    "\n\n1girl, smile, dress\n\n",
]


# Text outside ASCII.  The wave-per-file kernels take a token with such bytes when it is, as it stands, a known string that
# normalisation leaves alone, and hand every other such file to the general kernel, which rewrites white space and capitals
# outside ASCII first (Trim / \\s / ToLowerInvariant of Program.cs:1150-1154 over UTF-16) and then works on bytes.
NON_ASCII_VOCAB = ["1girl", "solo", "smile", "dress", "caf\u00e9", "\u9752\u3044\u9aea", "pok\u00e9mon_(creature)", "long_hair", "\u00dcber_tag",
                   "\u00fcber_tag", "\uff46\uff55\uff4c\uff4c", "na\u00efve_girl", "\U0001f600_face", "k_pop"]
NON_ASCII_LINES = [
    "1girl, caf\u00e9, dress, smile",
    "1girl, \u9752\u3044\u9aea, pok\u00e9mon_(creature), long_hair, smile",
    "caf\u00e9,1girl,solo,\u9752\u3044\u9aea,dress",
    "1girl, CAF\u00c9, dress, smile",                          # upper case outside ASCII: lower-casing needed
    "1girl,\u3000caf\u00e9, dress, smile",                    # U+3000 is white space to Trim()
    "1girl, caf\u00e9\u3002, dress, smile",                    # trailing ideographic full stop is stripped
    "1girl, caf\u00e9\u3001\u3002.;\u3001, dress, smile",
    "1girl, caf\u00e9 au lait, dress, smile",                  # blank inside a token with bytes outside ASCII
    "1girl, \u00dcber_tag, dress, smile",                      # in the vocabulary as written, but normalisation lower-cases it
    "1girl, smile, dress, unknown\u00e9",                      # not a known string as it stands
    "1girl, \uff26\uff35\uff2c\uff2c, smile, dress",          # full-width capitals lower-case to a vocabulary entry
    "1girl, NA\u00cfVE\u00a0GIRL, smile, dress",              # no-break space inside a token becomes '_'
    "1girl, na\u00efve\u2003\u2003_\u3000girl, smile, dress",  # runs of white space and underscores collapse
    "\u2028 1girl, smile\u2029, dress \u0085",                 # line and paragraph separators, NEL: white space to Trim()
    "1girl, \U0001f600_face, \U0001f600 FACE, smile, dress",   # beyond the Basic Multilingual Plane: copied as it is
    "1girl, \u212a_pop, smile, dress",                         # KELVIN SIGN lower-cases to an ASCII k
    "1girl, \u9752, \u00e9, x\u00e9, smile, dress",            # one UTF-16 unit: shorter than two (Program.cs:600)
    "smile, i won\u2019t do that, dress, 1girl",               # the refusal phrase with U+2019
    "1girl, I WON\u2019T, dress, smile",
    "i\u00a0can't, 1girl, dress, smile",                       # a no-break space is not the blank of the phrase
    "caf\u00e9 " + "\u9752" * 38,                              # 44 UTF-16 units, under two commas: LooksBad
    "caf\u00e9 " + "\u9752" * 35,                              # 40 units (110 bytes): not too long
    "\U0001f600" * 20 + "," + "\u9752",                       # 42 units (20 pairs + 2), one comma: LooksBad
    "\U0001f600" * 19 + "," + "\u9752",                       # 40 units
    "```\u3000\n1girl, caf\u00e9, smile, dress\n```\u3000",
    "POSITIVE:\u3000caf\u00e9, 1girl, smile, dress\nNEGATIVE: \u9752",
    "caf\u00e9,\r\n1girl,\r\nsmile, dress\r\n",
    "1girl, \u00b5m, smile, dress, \u0131x, \u017fx, \u0130x",   # characters whose case mappings are special
    "1girl, \u03c3\u03c2, \u03a3\u03a3, smile, dress",           # medial and final sigma share their capital (no table string holds either here)
]
# what stays unsupported (TP_F_UNSUPPORTED, empty caption): malformed UTF-8
NON_ASCII_UNSUPPORTED = [b"1girl, caf\xe9, smile, dress", b"1girl, \xc0\xaf, smile", b"1girl, \xed\xa0\x80x, smile", b"1girl, \xf4\x90\x80\x80, smile",
                         b"1girl, smile, \xe9\x9d"]
# ... and, when some table string holds one, lower-case characters that share their capital with another lower-case character
# (medial and final sigma, the micro sign and mu): ordinal-ignore-case comparison takes two such strings for one, byte comparison does not
TWIN_VOCAB = NON_ASCII_VOCAB + ["\u03c3\u03b9\u03b3\u03bc\u03b1"]
TWIN_LINES = [("1girl, \u03c3\u03b9\u03b3\u03bc\u03b1, smile, dress", True), ("1girl, \u03a3\u03b9\u03b3\u03bc\u03b1, smile", True), ("1girl, \u00b5m, smile", True),
              ("1girl, caf\u00e9, smile, dress", False), ("1girl, \u03b1\u03b3, smile, dress", False)]


def non_ascii_config(**kw):
    vocab = ("\n".join(NON_ASCII_VOCAB) + "\n").encode("utf-8")
    return shipped_config(vocabulary=vocab, vocabulary_is_json=0, **kw)


def check_non_ascii(run, oracle, histogram_dict):
    """run(arena, offsets) -> result of the implementation's flow; oracle: Oracle for the same configuration."""
    a, o = pack_lines(NON_ASCII_LINES)
    got, want = run(a, o), oracle.run_lines(a, o)
    assert got.counters["unsupported"] == 0
    assert_same_results(got, want, histogram_dict(got.histogram), "text outside ASCII")
    assert sum(1 for i in range(len(NON_ASCII_LINES)) if want.caption(i) and (want.verdicts["flags"][i] & 0x10) == 0) >= 20   # most lines are captioned, not rejected
    blobs = [x for x in NON_ASCII_UNSUPPORTED]
    offs = np.zeros(len(blobs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(b) for b in blobs])
    got = run(np.frombuffer(b"".join(blobs), dtype=np.uint8).copy(), offs)
    assert got.counters["unsupported"] == len(blobs)
    assert all(got.caption(i) == b"" and got.verdicts["flags"][i] & 0x20 for i in range(len(blobs)))


def check_case_twins(make_engine):
    """With a table string that holds a sigma, lines with characters of that kind are reported, not guessed; others are taken."""
    cfg = shipped_config(vocabulary=("\n".join(TWIN_VOCAB) + "\n").encode("utf-8"), vocabulary_is_json=0)
    eng, o = make_engine(cfg), Oracle(cfg)
    texts = [t for t, _ in TWIN_LINES]
    got, want = eng.run_texts(texts), o.run_texts(texts)
    assert got.verdicts["flags"][1] & 0x20 and got.verdicts["flags"][2] & 0x20
    for i, (text, refused) in enumerate(TWIN_LINES):
        if refused and (got.verdicts["flags"][i] & 0x20):
            assert got.caption(i) == b"", text
        else:        # (the wave kernels take the table string itself when it is spelled exactly as the table spells it)
            assert got.caption(i) == want.caption(i) and got.verdicts[i].tobytes() == want.verdicts[i].tobytes(), text


def triage_lines(seed, n):
    """Existing-caption texts for the triage entry point: well-formed captions, one- and two-tag lines around the 40-unit limit,
    refusals, blanks, repeated tags (which collapse below three tags), fenced and prefixed replies."""
    import random
    rng = random.Random(seed)
    vocab = [l.strip() for l in open(os.path.join(DATA, "top_tags.txt"), encoding="utf-8") if l.strip()]
    longs = [v for v in vocab if len(v) >= 18] or vocab
    out = ["", " ", "\n", ",", ", ,", "1girl", "1girl, smile", "1girl, smile, dress", "sorry, I can't help with that", "POSITIVE: 1girl, smile, dress\nNEGATIVE: blurry",
           "```\n1girl, smile, dress\n```", "smile, smile, smile, smile", "SMILE, smile, Smile, x", "a, b, c", "watermark, text, signature",
           "zz_unknown_tag_that_is_rather_long_indeed, smile", "zz unknown tag that is rather long indeed and more", "1girl," + " " * 60 + "smile"]
    while len(out) < n:
        r = rng.random()
        if r < 0.35:
            k = rng.randint(3, 40)
            toks = [rng.choice(vocab[:2000]) for _ in range(k)]
        elif r < 0.75:
            k = rng.choice([1, 1, 2, 2, 2, 3])
            src = longs if rng.random() < 0.7 else vocab[:500]
            toks = [rng.choice(src) for _ in range(k)]
            if rng.random() < 0.3:
                toks.append(toks[0].upper())
            if rng.random() < 0.2:
                toks.append("zz_" + "x" * rng.randint(0, 30))
        elif r < 0.85:
            toks = [rng.choice(["sorry", "i cannot", "as an ai", "unable to comply", "watermark", "text", "a", "", " "]) for _ in range(rng.randint(1, 4))]
        else:
            toks = [rng.choice(vocab[:300]).replace("_", " ") for _ in range(rng.randint(1, 5))]
        line = rng.choice([", ", ",", " , "]).join(toks)
        if rng.random() < 0.1:
            line = rng.choice(["POSITIVE: ", "```\n", " ", "NEGATIVE: "]) + line + rng.choice(["", "\n```", "\nNEGATIVE: bad", "\r\n"])
        out.append(line)
    return out


def check_top_tags_file(make_engine):
    """tp_format_top_tags: tags by count descending then bytes ascending, zero counts left out, and the text read back as
    the vocabulary file (LoadKnownTagDictionary's line format, Program.cs:1021-1028) makes every counted tag a known tag.

    Parity unpinned: the reference never writes this file and holds no fixture of it; what is checked is the stated
    order, the reader's contract, and captions against the oracle when the written file is the vocabulary."""
    from imagecaptioner_amd.corpus import make_corpus
    c = make_corpus("c1", 400, seed=21)
    cfg = Config(**c.config_kwargs)
    eng = make_engine(cfg)
    r = eng.run_lines(c.arena, c.offsets)
    hist = eng.histogram_dict(r.histogram)
    assert Oracle(cfg).run_lines(c.arena, c.offsets).histogram == hist
    text = eng.format_top_tags(r.histogram)
    assert text.endswith(b"\n") and b"\r" not in text and b"," not in text
    lines = text.decode("utf-8").split("\n")[:-1]
    assert len(lines) == len(set(lines)) and set(lines) == set(hist)          # every counted tag once; none with count zero
    keys = [(-hist[t], t.encode("utf-8")) for t in lines]
    assert keys == sorted(keys) and len({k[0] for k in keys}) < len(keys)      # ties exist and are ordered by bytes
    assert eng.format_top_tags(np.zeros_like(np.asarray(r.histogram))) == b""
    # read back as the vocabulary
    kw = dict(c.config_kwargs)
    kw.update(vocabulary=text, vocabulary_is_json=0)
    o = Oracle(Config(**kw))
    assert o.info.known_found == 1 and o.info.strict_known == 1 and o.info.n_known_tags == len(lines)
    assert not [t for t in lines if o.should_drop_tag(t)]
    eng2 = make_engine(Config(**kw))
    got, want = eng2.run_lines(c.arena, c.offsets), o.run_lines(c.arena, c.offsets)
    assert_same_results(got, want, eng2.histogram_dict(got.histogram), "written top_tags as vocabulary")
    assert set(eng2.histogram_dict(got.histogram)) <= set(lines)               # strict vocabulary: nothing new appears


# ---- cached tagger replies (the JSON bodies of JoyTag's /tag endpoint) -----------------------------------------------------
def joytag_replies(seed, n):
    """(primary, secondary) reply bodies in the shapes ParseJoyTags takes (Program.cs:463-489): objects with tag / name and
    score / confidence, bare strings, elements of other kinds, blank and aliased tags, escapes, and bodies without a tags array."""
    import json
    import random
    rng = random.Random(seed)
    vocab = [l.strip() for l in open(os.path.join(DATA, "top_tags.txt"), encoding="utf-8") if l.strip()][:500]
    extra = ["foot", "fingers", "Pedal Stimulation", "zz_unknown", "x'.", "", " ", "a", "Long Hair", "BLUE EYES", "caf\u00e9", "\u9752\u3044\u9aea", "licking", "soles"]

    def body(kind):
        k = rng.choice([0, 1, 3, 8, 25, 60])
        tags = [rng.choice(vocab + extra) for _ in range(k)]
        if kind == "objects":
            els = []
            for t in tags:
                e = {}
                e[rng.choice(["tag", "tag", "name"])] = t
                r = rng.random()
                if r < 0.7: e["score"] = round(rng.random(), 3)
                elif r < 0.9: e["confidence"] = rng.choice([0.28, 0.6, 0.5, 0.53, 1, 0, 1e-3, 12.5, -0.25])
                if rng.random() < 0.1: e["tag"] = rng.choice([None, 5, ["x"]]); e["name"] = t
                if rng.random() < 0.05: e["score"], e["confidence"] = round(rng.random(), 2), 0.99
                els.append(e)
            return {"tags": els}
        if kind == "strings":
            return {"tags": tags, "model": "joytag"}
        if kind == "objects_no_score":
            return {"tags": [{"tag": t} for t in tags]}
        if kind == "mixed_kinds":
            els = [{"tag": t, "score": round(rng.random(), 2)} for t in tags]
            for junk in (None, 5, True, [1, 2], {}, {"score": 0.9}, {"name": "smile", "confidence": 0.75}):
                els.insert(rng.randint(0, len(els)), junk)
            return {"tags": els}
        if kind == "many_strings":
            return {"tags": [rng.choice(vocab) for _ in range(1100)]}          # past order 990 the score stays 0.01
        if kind == "no_tags":
            return rng.choice([{}, {"tags": None}, {"tags": "smile"}, {"result": ["smile"]}, {"tags": {"tag": "smile"}}])
        raise ValueError(kind)
    kinds = ["objects"] * 6 + ["strings"] * 3 + ["objects_no_score", "mixed_kinds", "no_tags"]
    prim, sec = [], []
    for i in range(n):
        prim.append(json.dumps(body("many_strings" if i == 5 else rng.choice(kinds)), ensure_ascii=rng.random() < 0.5).encode("utf-8"))
        sec.append(json.dumps(body(rng.choice(kinds)), ensure_ascii=rng.random() < 0.5).encode("utf-8"))
    prim[0] = b'\xef\xbb\xbf {"tags": [ {"tag": "smile", "score": 9e-1}, {"name": "Long\\u0020Hair", "confidence": 1E0}, "dress" ] } '
    return prim, sec


# bodies the reference throws on (3), and lists without scores whose default scores rise along the list (2): ParseJoyTags leaves
# those unsorted (Program.cs:487)
BAD_REPLIES = [(b"", 3), (b"not json", 3), (b'["smile"]', 3), (b'{"tags": [{"tag": "smile", "score": "0.9"}]}', 3), (b'{"tags": [{"tag": "smile", "score": null}]}', 3),
               (b'{"tags": [{"tag": "smile", "confidence": [1]}]}', 3), (b'{"tags": ["smile",]}', 3), (b'{"tags": ["smile"]} x', 3),
               (b'{"tags": ["smile", {"tag": "dress"}, "long_hair"]}', 2), (b'{"tags": [{"tag": "dress"}, "smile"]}', 2),
               (b'{"tags": ["smile", {"tag": "dress"}]}', 0), (b'{"tags": [{"tag": "dress", "score": 0.5, "score": "x"}]}', 3)]


def check_joytag_replies(engine, oracle):
    """engine.decode_joytag_replies + run_scored / run_merged against the oracle's own reading of the same bodies."""
    prim, sec = joytag_replies(17, 120)
    d = engine.decode_joytag_replies(prim)
    want = oracle.run_replies(prim)
    assert list(d["status"]) == list(want.reply_status) and set(d["status"]) == {0, 1}
    assert d["secondary_begin"] is None and (d["sources"] == 0).all()
    got = engine.run_scored(d["arena"], d["tag_offsets"], d["tag_begin"], d["scores"], d["sources"])
    assert_same_results(got, want, engine.histogram_dict(got.histogram), "decoded replies")
    assert d["has_score"][0] == 1 and bytes(d["arena"][:int(d["tag_offsets"][3])]) == b"smileLong Hairdress" and list(d["scores"][:3]) == [0.9, 1.0, 0.998]
    for cap in (15, 2, 0):
        d = engine.decode_joytag_replies(prim, sec)
        want = oracle.run_replies(prim, sec, secondary_cap=cap)
        assert list(d["status"]) == list(want.reply_status)
        got = engine.run_merged(d["arena"], d["tag_offsets"], d["tag_begin"], d["secondary_begin"], d["scores"], cap)
        assert_same_results(got, want, engine.histogram_dict(got.histogram), "decoded replies, merged, cap %d" % cap)
    d = engine.decode_joytag_replies([b for b, _ in BAD_REPLIES])
    assert list(d["status"]) == [st for _, st in BAD_REPLIES]
    want = oracle.run_replies([b for b, _ in BAD_REPLIES])
    assert [int(x) for x in want.reply_status] == [st if st != 2 else 0 for _, st in BAD_REPLIES]
    assert all((d["tag_begin"][i + 1] == d["tag_begin"][i]) == (st == 3) for i, (_, st) in enumerate(BAD_REPLIES))
    # an unsorted list gives the same captions and verdicts sorted or not (SelectTags sorts by score itself, Program.cs:651-655)
    got = engine.run_scored(d["arena"], d["tag_offsets"], d["tag_begin"], d["scores"], d["sources"])
    assert_same_results(got, want, engine.histogram_dict(got.histogram), "replies without scores")
    if hasattr(engine, "run_replies"):
        got = engine.run_replies(d)
        assert_same_results(got, want, engine.histogram_dict(got.histogram), "replies without scores, order kept")


# Replies without a score anywhere whose default scores rise along the list: string elements score 1.0 - order * 0.001, an object
# without "score" / "confidence" 0.5 (Program.cs:470, 476).  The evidence set and the dropped tags are listed in the reply's order
# then (Program.cs:627-631, 723-727), not by score.
UNSORTED_REPLIES = [b'{"tags": [{"tag": "dress"}, "smile", {"name": "watermark"}, "long_hair", {"tag": ":bad"}, "blue_eyes", "1girl"]}',
                    b'{"tags": [{"tag": "standing"}, {"tag": "classroom"}, "blue_hair", {"tag": "a\\\"b"}, "text", {"tag": "dress"}, "Smile", "smile"]}',
                    b'{"tags": ["smile", "dress"]}',
                    b'{"tags": [{"tag": "dress", "score": 0.2}, "smile", {"tag": "standing", "score": 0.9}]}']
UNSORTED_SECOND = [b'{"tags": [{"tag": "sitting"}, "smile", "outdoors", {"tag": "dress", "score": 0.4}]}',
                   b'{"tags": ["classroom", {"tag": "red_eyes"}, "blue_hair"]}',
                   b'{"tags": [{"tag": "long_hair"}, "dress", "smile"]}',
                   b'{"tags": []}']


def unsorted_reply_lists(oracle, secondary=False, cap=15):
    """The oracle's audit lists for UNSORTED_REPLIES (with UNSORTED_SECOND as the second replies when asked)."""
    want = oracle.run_replies(UNSORTED_REPLIES, UNSORTED_SECOND if secondary else None, secondary_cap=cap, detail=True)
    return want, oracle_audit_lists(want)


def changed_action_case():
    """A configuration whose action rules name tags NormalizeTag changes (lexicon aliases and the built-in "fellatio"), and
    lines on which the changed tag meets an earlier element of the final list, a later one, another changed tag, or none."""
    import json
    lex = {"aliases": {"bj": "blowjob", "oral_act": "blowjob", "hj": "handjob"}, "implied": {"barefoot": ["feet"]}}
    actions = [
        {"parent_tag": "bj", "required_all": ["penis"], "required_any": ["mouth", "oral"], "forbidden_any": [],
         "subtypes": [{"tag": "oral_act", "match_any": ["oral"], "match_all": []}, {"tag": "deepthroat", "match_any": ["deepthroat"], "match_all": []}]},
        {"parent_tag": "hj", "required_all": ["penis", "hand"], "required_any": [], "forbidden_any": [], "subtypes": []},
        {"parent_tag": "blowjob", "required_all": ["penis", "sucking"], "required_any": [], "forbidden_any": [],
         "subtypes": [{"tag": "fellatio", "match_any": ["sucking"], "match_all": []}]},
    ]
    facets = {"act": {"default": "blowjob", "rules": [{"tag": "handjob", "match_any": ["hand"], "match_all": []}]},
              "view": {"default": "unknown_view", "rules": []}}
    vocab = ("\n".join(["1girl", "solo", "smile", "penis", "mouth", "dress", "oral", "blowjob", "hand", "sucking", "deepthroat", "barefoot", "feet",
                        "long_hair", "standing", "handjob", "unknown_view"]) + "\n").encode()
    cfg = Config(lexicon=json.dumps(lex).encode(), actions=json.dumps(actions).encode(), required_facets=json.dumps(facets).encode(),
                 vocabulary=vocab, vocabulary_is_json=0, minimal_subject_tags="1girl, solo")
    lines = ["1girl, smile, penis, mouth, dress",                       # bj -> blowjob, the facet's default blowjob comes later
             "1girl, smile, penis, oral, dress",                        # bj and oral_act: two changed tags with one new form
             "1girl, smile, penis, oral, blowjob, dress",               # the new form is a selected tag already
             "1girl, smile, penis, hand, mouth, dress",                 # hj -> handjob meets the facet rule's handjob; bj as well
             "1girl, smile, penis, sucking, dress",                     # blowjob, then fellatio -> blowjob
             "1girl, smile, penis, sucking, oral, hand, deepthroat, barefoot, dress",
             "1girl, smile, dress, long_hair, standing"]
    return cfg, lines
