// The per-file tag pipeline: one sequential pass over one file's tags, written
// so that the same source runs as a GPU thread (one file per lane, private
// arrays interleaved across the 64 lanes of a wavefront so that lanes touching
// the same element index hit one contiguous 512-byte row) and, for logic checks
// without a GPU, as a plain host function.
//
// It restates the pure span of the reference's loop body, Program.cs:223-229
// and 292-329 with the OpenAI branch off, fed by ParseTagCandidatesFromLine
// (Program.cs:574-605) or by scored candidates shaped like ParseJoyTags output
// (Program.cs:463-489).  Function-by-function citations are inline.
#pragma once
#include <math.h>

#include "../../include/tagpipe.h"
#include "tp_common.h"

namespace tp {

// Element k of a lane-private array: rows of `stride` elements, one per lane that holds a file (64 when every lane does: the
// lanes' accesses to one element coalesce; 1 when a wavefront holds one file: that file's elements are then neighbours in
// memory, which its many searches through its own lists like much better than one cache line per element).
template <class T>
struct LaneArr {
    T* base;
    uint32_t stride = 64;
    TP_HD T& operator[](uint32_t k) const { return base[(size_t)k * stride]; }
};

// Byte layout of one wavefront's workspace (64 files).  T = the largest token
// count among its files, LB = the largest line length.
struct WsLayout {
    uint64_t o_hnd, o_score, o_order, o_sorted, o_src, o_dl, o_ev, o_snap, o_fin, o_ebits, o_line, o_norm, o_top, o_top_score, o_top_order, total;
    uint32_t cap_t, cap_e, cap_snap, cap_f, cap_norm;
};
TP_HD uint64_t ws_align(uint64_t x) { return (x + 255) & ~(uint64_t)255; }
TP_HD WsLayout ws_layout(uint32_t T, uint32_t LB, uint32_t final_cap, uint32_t evidence_grow, uint32_t lit_words) {
    WsLayout L;
    if (T < 1) T = 1;
    L.cap_t = T;
    L.cap_e = T + evidence_grow;
    L.cap_f = final_cap;
    L.cap_snap = L.cap_e > L.cap_f ? L.cap_e : L.cap_f;
    L.cap_norm = (LB + 16 + 15) & ~15u;
    uint64_t o = 0;
    L.o_hnd = o;    o = ws_align(o + (uint64_t)T * 64 * 8);
    L.o_score = o;  o = ws_align(o + (uint64_t)T * 64 * 8);
    L.o_order = o;  o = ws_align(o + (uint64_t)T * 64 * 4);
    L.o_sorted = o; o = ws_align(o + (uint64_t)T * 64 * 4);
    L.o_src = o;    o = ws_align(o + (uint64_t)T * 64);
    L.o_dl = o;     o = ws_align(o + (uint64_t)T * 64 * 8);
    L.o_ev = o;     o = ws_align(o + (uint64_t)L.cap_e * 64 * 8);
    L.o_snap = o;   o = ws_align(o + (uint64_t)L.cap_snap * 64 * 8);
    L.o_fin = o;    o = ws_align(o + (uint64_t)L.cap_f * 64 * 8);
    L.o_ebits = o;  o = ws_align(o + (uint64_t)(lit_words ? lit_words : 1) * 64 * 4);
    L.o_line = o;   o = ws_align(o + (uint64_t)L.cap_norm * 64);
    L.o_norm = o;   o = ws_align(o + (uint64_t)L.cap_norm * 64);
    L.o_top = o;    o = ws_align(o + (uint64_t)TOP_TAGS * 64 * 8);
    L.o_top_score = o; o = ws_align(o + (uint64_t)TOP_TAGS * 64 * 8);
    L.o_top_order = o; o = ws_align(o + (uint64_t)TOP_TAGS * 64 * 4);
    L.total = o;
    return L;
}

struct FileWs {
    LaneArr<H> hnd, dl, ev, snap, fin, top;
    LaneArr<double> score, top_score;
    LaneArr<uint32_t> order, sorted, ebits, top_order;
    LaneArr<uint8_t> src;
    uint8_t* line;   // thread-contiguous, cap_norm bytes: CRLF-collapsed copy of the raw line when needed
    uint8_t* norm;   // thread-contiguous, cap_norm bytes: bytes of dynamic strings
    uint32_t cap_t, cap_e, cap_snap, cap_f, cap_norm;
};
// stride: files per wavefront (the layout's offsets leave room for 64)
TP_HD FileWs ws_bind(uint8_t* wave_base, const WsLayout& L, uint32_t lane, uint32_t stride = 64) {
    FileWs w;
    w.hnd.stride = w.dl.stride = w.ev.stride = w.snap.stride = w.fin.stride = w.top.stride = stride;
    w.score.stride = w.top_score.stride = stride;
    w.order.stride = w.sorted.stride = w.ebits.stride = w.top_order.stride = stride;
    w.src.stride = stride;
    w.hnd.base = (H*)(wave_base + L.o_hnd) + lane;
    w.score.base = (double*)(wave_base + L.o_score) + lane;
    w.order.base = (uint32_t*)(wave_base + L.o_order) + lane;
    w.sorted.base = (uint32_t*)(wave_base + L.o_sorted) + lane;
    w.src.base = (uint8_t*)(wave_base + L.o_src) + lane;
    w.dl.base = (H*)(wave_base + L.o_dl) + lane;
    w.ev.base = (H*)(wave_base + L.o_ev) + lane;
    w.snap.base = (H*)(wave_base + L.o_snap) + lane;
    w.fin.base = (H*)(wave_base + L.o_fin) + lane;
    w.ebits.base = (uint32_t*)(wave_base + L.o_ebits) + lane;
    w.top.base = (H*)(wave_base + L.o_top) + lane;
    w.top_score.base = (double*)(wave_base + L.o_top_score) + lane;
    w.top_order.base = (uint32_t*)(wave_base + L.o_top_order) + lane;
    w.line = wave_base + L.o_line + (size_t)lane * L.cap_norm;
    w.norm = wave_base + L.o_norm + (size_t)lane * L.cap_norm;
    w.cap_t = L.cap_t; w.cap_e = L.cap_e; w.cap_snap = L.cap_snap; w.cap_f = L.cap_f; w.cap_norm = L.cap_norm;
    return w;
}

struct FileIn {
    // line ingest
    const uint8_t* text;
    uint32_t len;
    // scored ingest (text == nullptr)
    const uint8_t* arena;
    const uint64_t* tag_offsets;
    uint64_t tag_lo, tag_hi;
    const double* scores;
    const uint8_t* sources;
    // two tagger replies to merge (Program.cs:214-221): [tag_lo, tag_mid) primary, [tag_mid, tag_hi) secondary; sources is not read
    bool merged;
    uint64_t tag_mid;
    int secondary_cap;
    // the (primary) reply stays in the order it came in: no element of it carried a score, so ParseJoyTags did not sort it
    // (Program.cs:487).  What follows the list's order downstream: the evidence set's and the dropped tags' insertion order
    // (Program.cs:627-631, 723-727) and which of several entries on one whole sort key comes first.
    bool keep_order;
};
struct FileOut {
    tp_verdict verdict;
    uint32_t n_final;      // handles left in ws.fin: the caption's tags, in order
    uint32_t caption_len;  // bytes of the caption line including the newline
    // mode 3 only: the tag lists of the audit row (AppendAudit, Program.cs:334, 966) and of the review record (BuildReviewRecord,
    // Program.cs:826-868): entries to walk and bytes of the column's text
    uint32_t audit_cnt[5];
    uint32_t audit_len[5];
    uint32_t review_bits;  // REVIEW_* : which banned tags the raw candidates held, and the subject-token tests
};
// review_bits: bit k (k < N_BANNED) = DevTables::banned_ids[k] is among the normalised raw candidates (bannedDetected, Program.cs:828)

// The tag-list columns of the audit row (Program.cs:966): keptTags = finalTags as they stand after EnsureRequiredFacets
// (Program.cs:307), evidenceTags after ExpandImpliedTags (Program.cs:302), finalSelection.DroppedTagsWithReason (Program.cs:682).
// After process_file in mode 3 they sit in the file's workspace: ws.snap, ws.ev and ws.dl.
// Two more columns feed the review record (Program.cs:860-861): the TopJoyTags (the 20 best raw candidates' tags) and the
// caption tags outside the vocabulary (GetUnknownTags before its sort, Program.cs:817-821); their entries are separated by "\n",
// which no normalised tag holds.  The unknown column walks the caption's tags (ws.fin) and keeps those that qualify.
// (AUDIT_* and REVIEW_* are in tp_common.h: the lean stage reports the same lists)

// ================================================================================================
// string helpers
// ================================================================================================
TP_HD const uint8_t* str_ptr(const DevTables& t, uint32_t id) { return t.str_bytes + t.str_off[id]; }
TP_HD uint32_t str_len(const DevTables& t, uint32_t id) { return t.str_off[id + 1] - t.str_off[id]; }

TP_HD uint32_t hflags(const DevTables& t, H h) {
    if (!h_is_dyn(h)) return t.flags[h_id(h)];
    return h_dyn_flags(h) | ((h & H_UNSTABLE) ? (uint32_t)FL_EDGE : 0u);
}
TP_HD uint32_t hbytes(const DevTables& t, H h) { return h_is_dyn(h) ? h_len(h) : str_len(t, h_id(h)); }

TP_HD bool heq(const FileWs& ws, H a, H b) {
    if (!h_is_dyn(a) || !h_is_dyn(b)) return a == b;   // a dynamic string is in no table, so mixed pairs differ
    uint32_t n = h_len(a);
    if (n != h_len(b)) return false;
    const uint8_t* p = ws.norm + h_off(a);
    const uint8_t* q = ws.norm + h_off(b);
    for (uint32_t i = 0; i < n; ++i) if (p[i] != q[i]) return false;
    return true;
}
// Is h in arr[0 .. n)?  Eight elements are fetched at a time so that their loads overlap: a thread walks these lists
// alone, and one load per step would cost a full memory latency each.
TP_HD bool lfind(const FileWs& ws, const LaneArr<H>& arr, uint32_t n, H h) {
    uint32_t i = 0;
    if (!h_is_dyn(h)) {                                            // an interned string equals only its own handle
        for (; i + 8 <= n; i += 8) {
            const H v0 = arr[i], v1 = arr[i + 1], v2 = arr[i + 2], v3 = arr[i + 3], v4 = arr[i + 4], v5 = arr[i + 5], v6 = arr[i + 6], v7 = arr[i + 7];
            if ((v0 == h) | (v1 == h) | (v2 == h) | (v3 == h) | (v4 == h) | (v5 == h) | (v6 == h) | (v7 == h)) return true;
        }
        for (; i < n; ++i) if (arr[i] == h) return true;
        return false;
    }
    for (; i + 8 <= n; i += 8) {
        const H v[8] = {arr[i], arr[i + 1], arr[i + 2], arr[i + 3], arr[i + 4], arr[i + 5], arr[i + 6], arr[i + 7]};
        for (int k = 0; k < 8; ++k) if (h_is_dyn(v[k]) && heq(ws, v[k], h)) return true;
    }
    for (; i < n; ++i) if (heq(ws, arr[i], h)) return true;
    return false;
}

// Index of h in arr[0 .. n), or n: the same walk, eight elements at a time.
TP_HD uint32_t lfind_idx(const FileWs& ws, const LaneArr<H>& arr, uint32_t n, H h) {
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const H v[8] = {arr[i], arr[i + 1], arr[i + 2], arr[i + 3], arr[i + 4], arr[i + 5], arr[i + 6], arr[i + 7]};
        for (int k = 0; k < 8; ++k) if (heq(ws, v[k], h)) return i + (uint32_t)k;
    }
    for (; i < n; ++i) if (heq(ws, arr[i], h)) return i;
    return n;
}

// Attributes of a string that is in no table (already normalised; bytes outside ASCII make it an invalid tag, Program.cs:40).
TP_HD uint32_t classify_dyn(const DevTables& t, const uint8_t* p, uint32_t n) {
    bool valid = n > 0 && is_tag_first8(p[0]);
    uint32_t units = 0;                                // string.Length (Program.cs:600): UTF-16 units
    uint32_t state = 0, mask = 0;
    // CharacterNameRegex ^[a-z0-9]+(_[a-z0-9]+){1,3}$ (Program.cs:47)
    uint32_t segs = 0, seglen = 0;
    bool name_ok = true;
    for (uint32_t i = 0; i < n; ++i) {
        uint8_t c = p[i];
        units += ((c & 0xC0) != 0x80) + (c >= 0xF0);
        if (i > 0 && !is_tag_rest8(c)) valid = false;
        state = t.dfa_next[state * DFA_SYMS + dfa_sym(c)];
        mask |= t.dfa_out[state];
        if (is_tag_first8(c)) { if (seglen == 0) ++segs; ++seglen; }
        else if (c == '_') { if (seglen == 0) name_ok = false; seglen = 0; }
        else name_ok = false;
    }
    if (seglen == 0) name_ok = false;
    bool is_name = name_ok && segs >= 2 && segs <= 4 && !(mask & (CM_FACE | CM_OUTFIT | CM_POSE | CM_BACKGROUND | CM_STYLE));   // IsLikelyCharacterName, Program.cs:1113-1119
    uint32_t bucket = is_name ? B_IDENTITY : (mask & CM_FACE) ? B_FACE : (mask & CM_OUTFIT) ? B_OUTFIT : (mask & CM_POSE) ? B_POSE
                      : (mask & CM_BACKGROUND) ? B_BACKGROUND : (mask & CM_STYLE) ? B_STYLE : B_OTHER;                           // ClassifyTag, Program.cs:1103-1111
    bool rating = n >= 6 && p[0] == 'r' && p[1] == 'a' && p[2] == 't' && p[3] == 'i' && p[4] == 'n' && p[5] == 'g';
    bool drop = !valid || (mask & CM_SORRY) || (t.allow_nonempty && (rating || (mask & CM_WATERMARK))) || t.drop_unknown;    // ShouldDropTag, Program.cs:1121-1133
    return (valid ? (uint32_t)FL_VALID : 0u) | (drop ? (uint32_t)FL_DROP : 0u) | (units >= 2 ? (uint32_t)FL_LEN2 : 0u) | (bucket << FL_BUCKET_SHIFT) |
           ((mask & CM_HAIR) ? (uint32_t)FL_HAIR : 0u) | ((mask & CM_EYES) ? (uint32_t)FL_EYES : 0u);
}

// BuildReviewRecord's subject-token tests (Program.cs:829-837) on a string that is in no table (ASCII, normalised, so lower
// case): such a string is in none of the reference's fixed sets, so it is a subject token only by holding "boys" or "girls".
TP_HD bool has_word(const uint8_t* p, uint32_t n, const char* w, uint32_t wl) {
    for (uint32_t i = 0; i + wl <= n; ++i) {
        uint32_t k = 0;
        while (k < wl && p[i + k] == (uint8_t)w[k]) ++k;
        if (k == wl) return true;
    }
    return false;
}
// Bytes of the decimal digit (Unicode category Nd, which is what \d matches) at p[i], or 0: ASCII digits, and the ten characters
// from each of the Basic Multilingual Plane's other digit zeros.
TP_HD uint32_t nd_len(const uint8_t* p, uint32_t n, uint32_t i) {
    const uint8_t c = p[i];
    if (c >= '0' && c <= '9') return 1;
    if (c < 0xC2 || c > 0xEF) return 0;
    const uint32_t len = c < 0xE0 ? 2u : 3u;
    if (i + len > n) return 0;
    const uint32_t cp = len == 2 ? ((c & 0x1Fu) << 6) | (p[i + 1] & 0x3Fu) : ((c & 0x0Fu) << 12) | ((p[i + 1] & 0x3Fu) << 6) | (p[i + 2] & 0x3Fu);
    const uint16_t zeros[36] = {0x0660, 0x06F0, 0x07C0, 0x0966, 0x09E6, 0x0A66, 0x0AE6, 0x0B66, 0x0BE6, 0x0C66, 0x0CE6, 0x0D66, 0x0DE6, 0x0E50, 0x0ED0, 0x0F20, 0x1040, 0x1090,
                                0x17E0, 0x1810, 0x1946, 0x19D0, 0x1A80, 0x1A90, 0x1B50, 0x1BB0, 0x1C40, 0x1C50, 0xA620, 0xA8D0, 0xA900, 0xA9D0, 0xA9F0, 0xAA50, 0xABF0, 0xFF10};
    for (int k = 0; k < 36; ++k) if (cp >= zeros[k] && cp < zeros[k] + 10u) return len;
    return 0;
}
TP_HD uint32_t subject_bits_dyn(const uint8_t* p, uint32_t n) {
    const bool girls = has_word(p, n, "girls", 5);
    if (!girls && !has_word(p, n, "boys", 4)) return 0;
    uint32_t bits = 0;
    if (girls || has_word(p, n, "girl", 4)) bits |= FL_SUBJ_GIRL;
    bool multi = n >= 9 && has_word(p, 9, "multiple_", 9);
    uint32_t d = 0;
    for (uint32_t l; d < n && (l = nd_len(p, n, d)) != 0;) d += l;          // ^\d+(girl|girls|boy|boys)$, Program.cs:837
    if (d > 0) {
        const uint32_t r = n - d;
        const uint8_t* q = p + d;
        multi |= (r == 4 && has_word(q, 4, "girl", 4)) || (r == 5 && has_word(q, 5, "girls", 5)) || (r == 3 && has_word(q, 3, "boy", 3)) || (r == 4 && has_word(q, 4, "boys", 4));
    }
    if (multi) bits |= FL_SUBJ_MULTI;
    return bits;
}

// ---- eight bytes of a text at a time (texts have eight readable bytes behind their end) -----------------------------------
constexpr uint64_t PL_01 = 0x0101010101010101ull, PL_7F = 0x7F7F7F7F7F7F7F7Full, PL_80 = 0x8080808080808080ull;
TP_HD uint64_t pl_ld8(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
TP_HD uint64_t pl_eq8(uint64_t x, uint8_t c) { const uint64_t y = x ^ (PL_01 * c); return ~(((y & PL_7F) + PL_7F) | y | PL_7F); }   // 0x80 in every byte equal to c
TP_HD uint64_t pl_first(uint32_t n) { return n >= 8 ? PL_80 : (PL_80 & ((1ull << (8 * n)) - 1)); }                                // 0x80 in each of the first n bytes
TP_HD uint32_t pl_low(uint64_t m) { return (uint32_t)__builtin_ctzll(m) >> 3; }                                                   // byte index of the lowest flag, m != 0
TP_HD bool is_letter8(uint8_t c) { c |= 0x20; return c >= 'a' && c <= 'z'; }

// The slot of a normalised token of at most 32 bytes, given as four zero-padded words: the id NormalizeTag gives it, or NO_ID
// when it is in no table (every interned string of up to 32 bytes is a key of the slots).
TP_HD uint32_t slot_lookup(const DevTables& t, uint32_t n, const uint64_t w[4]) {
    const uint64_t hsh = hash_finish(hash_word(hash_word(hash_word(hash_word(HASH_SEED, w[0], 0), w[1], 1), w[2], 2), w[3], 3), n);
    uint32_t slot = (uint32_t)hsh & t.ht_mask;
    const uint32_t h32 = (uint32_t)(hsh >> 32);
    for (;;) {
        const Slot* sl = t.slots + slot;
        const uint32_t slen = sl->slen;
        if (slen == SLOT_EMPTY) return NO_ID;
        if (sl->hash == h32 && slen == n) {
            uint64_t k[4];
            __builtin_memcpy(k, sl->key, 32);
            if (((k[0] ^ w[0]) | (k[1] ^ w[1]) | (k[2] ^ w[2]) | (k[3] ^ w[3])) == 0) return sl->val;
        }
        slot = (slot + 1) & t.ht_mask;
    }
}

// Hash-table probe with the candidate bytes given as plain memory.
TP_HD uint32_t probe_bytes(const DevTables& t, const uint8_t* p, uint32_t n, uint64_t hash) {
    uint32_t slot = (uint32_t)hash & t.ht_mask;
    for (;;) {
        uint64_t hh = t.ht_hash[slot];
        if (hh == 0) return NO_ID;
        if (hh == hash) {
            uint32_t id = t.ht_id[slot];
            if (str_len(t, id) == n) {
                const uint8_t* q = str_ptr(t, id);
                uint32_t i = 0;
                while (i < n && q[i] == p[i]) ++i;
                if (i == n) return id;
            }
        }
        slot = (slot + 1) & t.ht_mask;
    }
}

// Streams NormalizeTagBasic's last three steps (\s+ -> "_", _+ -> "_", ToLowerInvariant; Program.cs:1152-1154)
// over the span left by the trims.
struct NormIt {
    const uint8_t* s;
    uint32_t i, j;
    bool last_us;
    TP_HD NormIt(const uint8_t* s_, uint32_t i_, uint32_t j_) : s(s_), i(i_), j(j_), last_us(false) {}
    TP_HD int next() {
        while (i < j) {
            uint8_t c = s[i++];
            if (is_ws8(c) || c == '_') {
                if (last_us) continue;
                last_us = true;
                return '_';
            }
            last_us = false;
            return lower8(c);
        }
        return -1;
    }
};

// One raw token -> the handle NormalizeTag (Program.cs:1149-1158) gives it.  Returns false when the token is empty
// after trimming (Split's RemoveEmptyEntries, Program.cs:1183-1185).  `final_nl_rule` applies the regex `$` rule of
// Program.cs:1151 (it also matches before a final '\n'): only scored ingest can see a real '\n' there.
// `raw_utf8`: the bytes are as the caller gave them; otherwise utf8_prepare has been over them.
TP_HD bool make_token(const DevTables& t, FileWs& ws, uint32_t& norm_used, const uint8_t* s, uint32_t i, uint32_t j, bool final_nl_rule,
                      H& out, bool& unsupported, bool raw_utf8 = false) {
    while (i < j && is_ws8(s[i])) ++i;                 // TrimEntries / Trim()
    while (j > i && is_ws8(s[j - 1])) --j;
    if (i == j) return false;
    bool high = false;
    for (uint32_t p = i; p < j; ++p) high |= s[p] >= 0x80;
    if (high && raw_utf8) {
        // a token straight from the caller's arena: white space and capitals outside ASCII are rewritten first (the line ingest
        // has done that to the whole line already), into the line buffer, which the scored ingest does not use otherwise
        const uint32_t m = utf8_prepare(t, s + i, j - i, ws.line, ws.cap_norm, false, true);   // tagger replies keep unknown tags (dropped list): always refuse
        if (m == 0xFFFFFFFFu) { unsupported = true; return false; }
        s = ws.line; i = 0; j = m;
        while (i < j && is_ws8(s[i])) ++i;
        while (j > i && is_ws8(s[j - 1])) --j;
        if (i == j) return false;
    }
    while (i < j && is_quote8(s[i])) ++i;              // Trim('"', '\'', '`')
    while (j > i && is_quote8(s[j - 1])) --j;
    uint32_t tail = j;                                 // [;。、\.]+$
    if (final_nl_rule && j > i && s[j - 1] == '\n') {
        const uint32_t q = strip_tail_punct(s, i, j - 1);
        // the punctuation run before the final newline goes, the newline stays (and becomes '_')
        if (q < j - 1) {
            // normalise [i,q) followed by the newline: handled by hashing the two pieces in sequence below
            tail = q;
        } else {
            tail = j;
        }
        // pass 1: hash + length
        Hasher hs;
        bool last_us = false;
        {
            NormIt it(s, i, tail);
            for (int c; (c = it.next()) >= 0;) hs.add((uint8_t)c);
            last_us = it.last_us;
        }
        bool add_us = (tail < j) && !last_us;          // the kept '\n' -> '_' unless it merges with a preceding '_'
        if (tail == j) add_us = false;                 // newline already inside [i,tail)
        if (add_us) hs.add('_');
        const uint32_t n = hs.n;
        const uint64_t hsh = hs.done();
        // materialise (rare path) and look it up
        if (n > MAX_DYN_LEN || norm_used + n > ws.cap_norm) { unsupported = true; return false; }
        uint8_t* dst = ws.norm + norm_used;
        {
            NormIt it(s, i, tail);
            uint32_t k = 0;
            for (int c; (c = it.next()) >= 0;) dst[k++] = (uint8_t)c;
            if (add_us) dst[k++] = '_';
        }
        uint32_t id = probe_bytes(t, dst, n, hsh);
        if (id != NO_ID) { out = t.alias1[id]; return true; }
        out = h_make_dyn(norm_used, n, classify_dyn(t, dst, n), n > 0 && is_quote8(dst[n - 1]));
        norm_used += n;
        return true;
    }
    tail = strip_tail_punct(s, i, tail);
    j = tail;
    // The usual token holds nothing the steps above and below change (no capital, white space, doubled underscore, and by now
    // no quote or punctuation at its ends): its bytes are the normalised string, looked up as four words in one probe.
    if (j - i <= 32 && !high && !is_quote8(s[i])) {              // (the word-wise tests below are for bytes under 0x80)
        const uint32_t n = j - i;
        uint64_t w[4], changed = 0, prev_us = 0;
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t r = n > 8 * k ? n - 8 * k : 0;
            const uint64_t x = r == 0 ? 0 : pl_ld8(s + i + 8 * k) & (r >= 8 ? ~0ull : ((1ull << (8 * r)) - 1));
            w[k] = x;
            const uint64_t up = (x + PL_01 * 0x3F) & ~(x + PL_01 * 0x25) & PL_80;                         // 'A' .. 'Z' (bytes below 0x80)
            const uint64_t wsm = pl_eq8(x, ' ') | ((x + PL_01 * (0x80 - 9)) & ~(x + PL_01 * (0x80 - 14)) & PL_80);   // white space
            const uint64_t us = pl_eq8(x, '_');
            changed |= up | wsm | (us & ((us << 8) | (prev_us >> 56)));
            prev_us = us;
        }
        if (!changed) {
            const uint32_t id = slot_lookup(t, n, w);
            if (id != NO_ID) { out = id; return true; }
        }
    }
    // pass 1: hash and length of the normalised form
    Hasher hs;
    {
        NormIt it(s, i, j);
        for (int c; (c = it.next()) >= 0;) hs.add((uint8_t)c);
    }
    const uint32_t n = hs.n;
    const uint64_t hsh = hs.done();
    // probe, verifying the bytes against the stream
    uint32_t slot = (uint32_t)hsh & t.ht_mask;
    for (;;) {
        uint64_t hh = t.ht_hash[slot];
        if (hh == 0) break;
        if (hh == hsh) {
            uint32_t id = t.ht_id[slot];
            if (str_len(t, id) == n) {
                const uint8_t* q = str_ptr(t, id);
                NormIt it(s, i, j);
                uint32_t k = 0;
                for (; k < n; ++k) if ((int)q[k] != it.next()) break;
                if (k == n) { out = t.alias1[id]; return true; }
            }
        }
        slot = (slot + 1) & t.ht_mask;
    }
    // in no table: keep the bytes
    if (n > MAX_DYN_LEN || norm_used + n > ws.cap_norm) { unsupported = true; return false; }
    uint8_t* dst = ws.norm + norm_used;
    {
        NormIt it(s, i, j);
        uint32_t k = 0;
        for (int c; (c = it.next()) >= 0;) dst[k++] = (uint8_t)c;
    }
    out = h_make_dyn(norm_used, n, classify_dyn(t, dst, n), n > 0 && is_quote8(dst[n - 1]));
    norm_used += n;
    return true;
}

// NormalizeTag applied to a string the pipeline already holds (Program.cs:1149-1158).
TP_HD H norm_hop(const DevTables& t, const FileWs& ws, H h) {
    if (!h_is_dyn(h)) return t.next[h_id(h)];
    if (!(h & H_UNSTABLE)) return h;
    // An already-normalised string that ends in a quote: the quote trim and then the punctuation strip shorten it.
    const uint8_t* p = ws.norm + h_off(h);
    uint32_t n = h_len(h);
    while (n > 0 && is_quote8(p[n - 1])) --n;
    n = strip_tail_punct(p, 0, n);
    if (n == 0) return t.alias1[EMPTY_ID];
    Hasher hs;
    for (uint32_t k = 0; k < n; ++k) hs.add(p[k]);
    uint32_t id = probe_bytes(t, p, n, hs.done());
    if (id != NO_ID) return t.alias1[id];
    return h_make_dyn(h_off(h), n, classify_dyn(t, p, n), is_quote8(p[n - 1]));
}

// ExpandImpliedTags (Program.cs:1160-1179) on an insertion-ordered set.
// The reference repeats whole rounds (normalise every tag, add each form that is missing, then its implied tags) until nothing
// changes.  A tag that was there when an earlier round began has had its form and that form's implied tags added by that round,
// and the set only grows, so visiting it again adds nothing: each round needs to visit only the tags the round before appended,
// in their order, which appends the same tags in the same order.
TP_HD void expand_implied(const DevTables& t, FileWs& ws, const LaneArr<H>& L, uint32_t& n, uint32_t cap, bool& overflow) {
    bool any = false;
    for (uint32_t i = 0; i < n; ++i) if (hflags(t, L[i]) & FL_EDGE) { any = true; break; }
    if (!any) return;   // every NormalizeTag is the identity and nothing has implied tags: the loop body cannot change the set
    uint32_t done = 0;
    while (done < n) {
        const uint32_t round_end = n;                  // tags appended from here on belong to the next round's snapshot
        for (uint32_t i = done; i < round_end; ++i) {
            const H cur = L[i];
            if (!(hflags(t, cur) & FL_EDGE)) continue;  // its form is itself and has no implied tags
            const H tag = norm_hop(t, ws, cur);
            if (hflags(t, tag) & FL_BLANK) continue;
            if (tag != cur && !lfind(ws, L, n, tag)) {
                if (n >= cap) { overflow = true; return; }
                L[n++] = tag;
            }
            if (!h_is_dyn(tag)) {
                const uint32_t a = t.imp_off[h_id(tag)], b = t.imp_off[h_id(tag) + 1];
                for (uint32_t m = a; m < b; ++m) {
                    const H imp = t.imp_ids[m];
                    if (lfind(ws, L, n, imp)) continue;
                    if (n >= cap) { overflow = true; return; }
                    L[n++] = imp;
                }
            }
        }
        done = round_end;
    }
}

TP_HD bool ebit(const FileWs& ws, uint32_t idx) { return (ws.ebits[idx >> 5] >> (idx & 31)) & 1u; }
TP_HD bool all_set(const DevTables& t, const FileWs& ws, uint32_t off, uint32_t cnt) {
    for (uint32_t k = 0; k < cnt; ++k) if (!ebit(ws, t.pool[off + k])) return false;
    return true;
}
TP_HD bool any_set(const DevTables& t, const FileWs& ws, uint32_t off, uint32_t cnt) {
    for (uint32_t k = 0; k < cnt; ++k) if (ebit(ws, t.pool[off + k])) return true;
    return false;
}
// MatchesSubtypeRule / MatchesFacetRule (Program.cs:784-794)
TP_HD bool match_any_all(const DevTables& t, const FileWs& ws, const DSub& r) {
    if (r.any_cnt && !any_set(t, ws, r.any_off, r.any_cnt)) return false;
    if (r.all_cnt && !all_set(t, ws, r.all_off, r.all_cnt)) return false;
    return true;
}

TP_HD bool match_ci(const uint8_t* s, uint32_t pos, uint32_t end, const char* pat, uint32_t plen, bool nl_is_space) {
    if (pos + plen > end) return false;
    for (uint32_t k = 0; k < plen; ++k) {
        uint8_t c = s[pos + k];
        if (nl_is_space && c == '\n') c = ' ';
        if (lower8(c) != (uint8_t)pat[k]) return false;
    }
    return true;
}
TP_HD void trim_span(const uint8_t* s, uint32_t& a, uint32_t& b) {
    while (a < b && is_ws8(s[a])) ++a;
    while (b > a && is_ws8(s[b - 1])) --b;
}
// DropTagsContains (Program.cs:56-58) over a line whose '\n' count as ' ' (Program.cs:595); the entry with U+2019 is
// matched on its UTF-8 bytes.
TP_HD bool has_refusal_phrase(const uint8_t* s, uint32_t a, uint32_t b) {
    for (uint32_t p = a; p < b; ++p) {
        switch (lower8(s[p])) {
            case 'n': if (match_ci(s, p, b, "negative:", 9, true)) return true; break;
            case 'p': if (match_ci(s, p, b, "positive:", 9, true)) return true; break;
            case 's': if (match_ci(s, p, b, "sorry", 5, true)) return true; break;
            case 'i':
                if (p + 1 < b && (s[p + 1] == ' ' || s[p + 1] == '\n' || s[p + 1] == '\'')) {
                    if (match_ci(s, p, b, "i can't", 7, true) || match_ci(s, p, b, "i cannot", 8, true) ||
                        match_ci(s, p, b, "i'm not able", 12, true) || match_ci(s, p, b, "i won't", 7, true) ||
                        match_ci(s, p, b, "i won\xE2\x80\x99t", 9, true)) return true;
                }
                break;
            case 'c': if (match_ci(s, p, b, "cannot provide", 14, true)) return true; break;
            case 'u': if (match_ci(s, p, b, "unable to", 9, true)) return true; break;
            case 'a': if (match_ci(s, p, b, "as an ai", 8, true)) return true; break;
            default: break;
        }
    }
    return false;
}

// Entry k of an audit column: where its bytes are, and for the dropped column the reason the reference appends.
struct AuditElem { const uint8_t* p; uint32_t len; const char* reason; bool keep; };
TP_HD bool tag_unknown(const DevTables& t, H h) { return t.known_active && (h_is_dyn(h) || !(t.flags[h_id(h)] & FL_KNOWN)); }   // Program.cs:818-820
TP_HD AuditElem audit_elem(const DevTables& t, const FileWs& ws, int col, uint32_t k) {
    const H h = col == AUDIT_KEPT ? ws.snap[k] : col == AUDIT_EVIDENCE ? ws.ev[k] : col == AUDIT_DROPPED ? ws.dl[k] : col == AUDIT_TOP ? ws.top[k] : ws.fin[k];
    AuditElem e;
    e.keep = col != AUDIT_UNKNOWN || tag_unknown(t, h);
    if (h_is_dyn(h)) { e.p = ws.norm + h_off(h); e.len = h_len(h); }
    else { e.p = str_ptr(t, h_id(h)); e.len = str_len(t, h_id(h)); }
    e.reason = col != AUDIT_DROPPED ? nullptr : !(hflags(t, h) & FL_VALID) ? ":invalid_tag" : ":policy_drop";   // Program.cs:630-631
    return e;
}

// has_refusal_phrase for a text without ':' (so without the two entries that hold one).  Every other entry but "sorry" has a
// blank that follows a letter: the text is searched for such blanks (a line feed counts as the blank it becomes) and for
// 's', and the entries are tried only there.
TP_HD bool has_refusal_phrase_nocolon(const uint8_t* s, uint32_t a, uint32_t b) {
    for (uint32_t p = a; p < b; p += 8) {
        const uint64_t x = pl_ld8(s + p), in = pl_first(b - p);
        uint64_t ss = pl_eq8(x | (PL_01 * 0x20), 's') & in;
        uint64_t sp = (pl_eq8(x, ' ') | pl_eq8(x, '\n')) & in & ~(pl_eq8(x, ',') << 8);   // a blank right after a comma starts no entry
        while (ss) {
            const uint32_t q = p + pl_low(ss);
            ss &= ss - 1;
            if (match_ci(s, q, b, "sorry", 5, true)) return true;
        }
        while (sp) {
            const uint32_t q = p + pl_low(sp);
            sp &= sp - 1;
            if (q <= a || !is_letter8(s[q - 1])) continue;
            if (match_ci(s, q - 1, b, "i can't", 7, true) || match_ci(s, q - 1, b, "i cannot", 8, true) || match_ci(s, q - 1, b, "i won't", 7, true) ||
                match_ci(s, q - 1, b, "i won\xE2\x80\x99t", 9, true)) return true;
            if (q >= a + 2 && match_ci(s, q - 2, b, "as an ai", 8, true)) return true;
            if (q >= a + 3 && match_ci(s, q - 3, b, "i'm not able", 12, true)) return true;
            if (q >= a + 6 && (match_ci(s, q - 6, b, "cannot provide", 14, true) || match_ci(s, q - 6, b, "unable to", 9, true))) return true;
        }
    }
    return false;
}

// ================================================================================================
// The pipeline.  mode 0: whole span.  mode 2: existing-caption triage (line ingest only; stops after the candidates).
// mode 1: stop after SelectTags and leave its result in ws.fin
// (used once at load time to evaluate the minimal fallback line, Program.cs:298).
// mode 3: whole span, and the lists of the audit row are kept and sized (see FileOut).
// ================================================================================================
#if defined(TP_PIPE_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define TP_PIPE_STAMP(k) pipe_stamp[k] = wall_clock64();
#define TP_PIPE_STAMP_DECL unsigned long long pipe_stamp[16] = {0}; pipe_stamp[0] = wall_clock64();
#define TP_PIPE_STAMP_PRINT if (blockIdx.x == 0 && threadIdx.x == 0) { printf("pipe stamps (100 MHz ticks):"); for (int q = 1; q < 16; ++q) printf(" %d:%llu", q, pipe_stamp[q] ? pipe_stamp[q] - pipe_stamp[0] : 0ull); printf("\n"); }
#else
#define TP_PIPE_STAMP(k)
#define TP_PIPE_STAMP_DECL
#define TP_PIPE_STAMP_PRINT
#endif
TP_HD void process_file(const DevTables& t, const FileIn& in, FileWs& ws, FileOut& out, uint32_t* fired, uint16_t* facet_choice, int mode) {
    TP_PIPE_STAMP_DECL
    tp_verdict& v = out.verdict;
    v.coverage = 0; v.flags = 0; v.n_tags = 0; v.n_candidates = 0; v.n_selected = 0; v.safe_hit = -1; v.n_evidence = 0; v.n_unknown = 0; v.n_dropped = 0;
    out.n_final = 0;
    out.caption_len = 0;
    for (int c = 0; c < AUDIT_COLS; ++c) { out.audit_cnt[c] = 0; out.audit_len[c] = 0; }
    out.review_bits = 0;
    if (fired) for (uint32_t w = 0; w < t.fired_words; ++w) fired[w] = 0;
    if (facet_choice) for (uint32_t f = 0; f < t.n_facets; ++f) facet_choice[f] = 0xFFFF;

    bool unsupported = false, overflow = false, rejected = false;
    uint32_t n = 0;          // candidates
    uint32_t norm_used = 0;

    if (in.text) {
        // ---------------- NormalizeRawTagLine (Program.cs:589-605) ----------------
        const uint8_t* s = in.text;
        uint32_t a = 0, b = in.len;
        bool has_colon = false, has_tick = false, has_crlf = false, has_high = false;
        {
            bool prev_cr = false;                               // the word before ended in a carriage return
            for (uint32_t p = 0; p < b; p += 8) {
                const uint64_t x = pl_ld8(s + p), in = pl_first(b - p);
                has_high |= (x & in) != 0;                      // such tokens are taken only as known strings, see make_token
                has_colon |= (pl_eq8(x, ':') & in) != 0;
                has_tick |= (pl_eq8(x, '`') & in) != 0;
                const uint64_t cr = pl_eq8(x, '\r') & in, lf = pl_eq8(x, '\n') & in;
                if ((cr & (lf >> 8)) || (prev_cr && (lf & 0x80))) has_crlf = true;
                prev_cr = (cr >> 63) != 0;
            }
        }
        if (has_high) {
            // white space and capitals outside ASCII are rewritten once, here (and "\r\n" collapsed in the same pass): every
            // step below then works on bytes, with ASCII white space only and nothing left for ToLowerInvariant to change
            const uint32_t k = utf8_prepare(t, s, b, ws.line, ws.cap_norm, true, t.tables_hold_twins != 0);   // unknown tags outside ASCII are dropped here
            if (k == 0xFFFFFFFFu) { v.flags = TP_F_UNSUPPORTED; return; }
            s = ws.line;
            b = k;
        } else if (has_crlf) {                                  // raw.Replace("\r\n", "\n")
            uint32_t k = 0;
            for (uint32_t p = 0; p < b; ++p) {
                if (s[p] == '\r' && p + 1 < b && s[p + 1] == '\n') continue;
                ws.line[k++] = s[p];
            }
            s = ws.line;
            b = k;
        }
        trim_span(s, a, b);                                     // .Trim()  (a blank raw line ends here too, Program.cs:590)
        if (a < b && has_colon) {
            // ExtractAfterPrefixLine(s, "POSITIVE:"), Program.cs:1199-1205
            uint32_t p = a;
            for (;;) {
                uint32_t e = p;
                while (e < b && s[e] != '\n') ++e;
                uint32_t la = p, lb = e;
                trim_span(s, la, lb);
                if (match_ci(s, la, lb, "positive:", 9, false)) {
                    la += 9;
                    trim_span(s, la, lb);
                    if (la < lb) { a = la; b = lb; }            // only a non-blank remainder replaces s (Program.cs:593)
                    break;
                }
                if (e >= b) break;
                p = e + 1;
            }
            // CutAfterPrefix(s, "NEGATIVE:"), Program.cs:1207-1210
            for (uint32_t q = a; q + 9 <= b; ++q) {
                if (match_ci(s, q, b, "negative:", 9, false)) { b = q; trim_span(s, a, b); break; }
            }
        }
        if (a < b && has_tick) {
            // StripCodeFences, Program.cs:1189-1197
            trim_span(s, a, b);
            if (b - a >= 3 && s[a] == '`' && s[a + 1] == '`' && s[a + 2] == '`') {
                uint32_t q = a;
                while (q < b && s[q] != '\n') ++q;
                if (q < b) a = q + 1;
                if (b - a >= 3) {
                    for (uint32_t r = b - 3 + 1; r-- > a;) {
                        if (s[r] == '`' && s[r + 1] == '`' && s[r + 2] == '`') { b = r; break; }
                    }
                }
                trim_span(s, a, b);
            }
        }
        trim_span(s, a, b);                                     // .Replace("\n", " ").Trim()
        // LooksBad, Program.cs:1141-1147
        if (a >= b) rejected = true;
        else {
            uint32_t commas = 0;
            for (uint32_t p = a; p < b; p += 8) commas += (uint32_t)__builtin_popcountll(pl_eq8(pl_ld8(s + p), ',') & pl_first(b - p));
            if (has_colon ? has_refusal_phrase(s, a, b) : has_refusal_phrase_nocolon(s, a, b)) rejected = true;
            else if (commas < 2 && (has_high ? utf16_units(s + a, b - a) : b - a) > 40) rejected = true;   // s.Length: UTF-16 units (Program.cs:1146)
        }
        TP_PIPE_STAMP(12)
        if (!rejected) {
            // SplitTags -> NormalizeTag -> Length >= 2 -> !ShouldDropTag -> Distinct (Program.cs:598-602)
            uint32_t i = a;
            for (;;) {
                uint32_t j = i;
                while (j < b) {                                  // the next comma
                    const uint64_t m = pl_eq8(pl_ld8(s + j), ',') & pl_first(b - j);
                    if (m) { j += pl_low(m); break; }
                    j += 8;
                }
                if (j > b) j = b;
                H h;
                if (make_token(t, ws, norm_used, s, i, j, false, h, unsupported)) {
                    uint32_t f = hflags(t, h);
                    if ((f & FL_LEN2) && !(f & FL_DROP) && !lfind(ws, ws.hnd, n, h)) {
                        if (n >= ws.cap_t) { unsupported = true; }
                        else ws.hnd[n++] = h;
                    }
                }
                if (unsupported) break;
                if (j >= b) break;
                i = j + 1;
            }
        }
        TP_PIPE_STAMP(13)
        if (unsupported) { v.flags = TP_F_UNSUPPORTED; return; }
        // which of the configuration's FL_SPLIT strings a kept tag is (t.n_split otherwise)
        auto split_index = [&](H h) -> uint32_t {
            if (h_is_dyn(h) || !(t.flags[h_id(h)] & FL_SPLIT)) return t.n_split;
            for (uint32_t k = 0; k < t.n_split; ++k) if (t.split_src[k] == h_id(h)) return k;
            return t.n_split;
        };
        if (mode == 2) {
            // Existing-caption triage stops here: LooksBad of the kept tags joined with ", " (Program.cs:206-207, 604, 1141-1147)
            bool bad = rejected || n == 0;
            uint32_t units = 0, commas = n ? n - 1 : 0;
            for (uint32_t i = 0; i < n && !bad; ++i) {
                H h = ws.hnd[i];
                if (h_is_dyn(h)) units += h_len(h);                // ASCII: one UTF-16 unit per byte (a phrase would have dropped it)
                else { if (t.flags[h_id(h)] & FL_PHRASE) bad = true; units += t.rec[h_id(h)].len16; }
                if (t.n_split) { const uint32_t k = split_index(h); if (k < t.n_split) commas += t.split_commas[k]; }   // a kept tag's own commas count too (Program.cs:1145)
            }
            if (!bad && commas < 2 && units + 2 * (n - 1) > 40) bad = true;
            v.flags = bad ? (uint8_t)TP_F_TRIAGE_BAD : (uint8_t)0;
            v.n_candidates = (uint16_t)(n > 0xFFFF ? 0xFFFF : n);
            return;
        }
        // The join and the second SplitTags (Program.cs:604, 578) return the kept tags as they are, except a force-allowed tag
        // that holds ',': its pieces take its place, each a candidate of its own.  In place: tags without pieces go first
        // (the list shrinks), then the list grows from its end.
        if (t.n_split) {
            uint32_t m = 0, grown = 0;
            for (uint32_t i = 0; i < n; ++i) {
                const H h = ws.hnd[i];
                const uint32_t k = split_index(h);
                if (k < t.n_split && t.split_cnt[k] == 0) continue;
                grown += k < t.n_split ? t.split_cnt[k] : 1u;
                ws.hnd[m++] = h;
            }
            if (grown > ws.cap_t) { v.flags = TP_F_UNSUPPORTED; return; }
            for (uint32_t i = m, dst = grown; i-- > 0;) {
                const H h = ws.hnd[i];
                const uint32_t k = split_index(h);
                if (k == t.n_split) { ws.hnd[--dst] = h; continue; }
                for (uint32_t q = t.split_cnt[k]; q-- > 0;) ws.hnd[--dst] = (H)t.split_ids[k][q];
            }
            n = grown;
        }
        // ParseTagCandidatesFromLine scores, Program.cs:581-583
        for (uint32_t i = 0; i < n; ++i) {
            double base = fmax(0.05, 1.0 - (double)(int)i * 0.01);
            ws.score[i] = base * t.line_weight;
            ws.order[i] = i;
            ws.src[i] = (uint8_t)t.line_source;
        }
    } else {
        // ---------------- scored ingest, shaped like ParseJoyTags (Program.cs:473-487) ----------------
        // One tagger reply: tags [lo, hi) become entries g0 .. of the candidate arrays, ordered by (score descending, order ascending).
        auto ingest = [&](uint64_t lo, uint64_t hi, uint32_t g0, int fixed_source, bool keep) -> uint32_t {
            uint32_t order = 0, m = 0;
            for (uint64_t k = lo; k < hi; ++k) {
                const uint8_t* s = in.arena + in.tag_offsets[k];
                uint32_t len = (uint32_t)(in.tag_offsets[k + 1] - in.tag_offsets[k]);
                H h;
                bool nonempty = make_token(t, ws, norm_used, s, 0, len, true, h, unsupported, true);
                if (unsupported) break;
                if (!nonempty) h = t.alias1[EMPTY_ID];                // NormalizeTag("") still goes through the alias lookup
                if (hflags(t, h) & FL_BLANK) continue;
                if (g0 + m >= ws.cap_t) { unsupported = true; break; }
                // insertion by (score descending, order ascending): OrderByDescending(Score).ThenBy(Order), Program.cs:487
                double sc = in.scores[k];
                uint32_t pos = g0 + m;
                while (!keep && pos > g0 && ws.score[pos - 1] < sc) {
                    ws.hnd[pos] = ws.hnd[pos - 1]; ws.score[pos] = ws.score[pos - 1]; ws.order[pos] = ws.order[pos - 1]; ws.src[pos] = ws.src[pos - 1];
                    --pos;
                }
                ws.hnd[pos] = h; ws.score[pos] = sc; ws.order[pos] = order++; ws.src[pos] = fixed_source >= 0 ? (uint8_t)fixed_source : in.sources[k];
                ++m;
            }
            return m;
        };
        if (!in.merged) n = ingest(in.tag_lo, in.tag_hi, 0, -1, in.keep_order);
        else {
            const uint32_t np = ingest(in.tag_lo, in.tag_mid, 0, TP_SRC_PRIMARY_JOY, in.keep_order);
            const uint32_t ns = unsupported ? 0 : ingest(in.tag_mid, in.tag_hi, np, TP_SRC_SECONDARY_JOY, false);   // LimitSecondaryTags sorts it in any case (Program.cs:805)
            if (!unsupported) {
                // MergeCandidateTags (Program.cs:442-461) over the primary reply followed by what LimitSecondaryTags
                // (Program.cs:801-815) lets through of the secondary one.  The map's values sit in insertion order in
                // entries 0 .. nm of the same arrays: entry nm is written only after entry i >= nm has been read.
                uint32_t nm = 0, nm_primary = 0;
                auto feed = [&](uint32_t i, uint32_t limit_to) -> bool {          // limit_to: entries searched for the tag
                    H h = ws.hnd[i]; double sc = ws.score[i]; uint32_t od = ws.order[i]; uint32_t sr = ws.src[i];
                    const uint32_t j = lfind_idx(ws, ws.hnd, limit_to, h);
                    if (j == limit_to) return false;
                    double osc = ws.score[j];
                    uint32_t osr = ws.src[j];
                    // PreferSource (Program.cs:796-799) on (old, new): far-apart scores are both replaced by the better one
                    // before the call (Program.cs:452), so its own distance test sees them equal or close either way
                    bool close = fabs(osc - sc) <= 0.03;
                    double best = sc > osc ? sc : osc;
                    double sa = close ? osc : best, sb = close ? sc : best;
                    uint32_t source = fabs(sa - sb) > 0.03 ? (sa >= sb ? osr : sr) : (source_rank(osr) <= source_rank(sr) ? osr : sr);
                    ws.score[j] = best;                                   // Math.Max
                    if (od < ws.order[j]) ws.order[j] = od;               // Math.Min
                    ws.src[j] = (uint8_t)source;
                    return true;
                };
                auto append = [&](uint32_t i) {
                    ws.hnd[nm] = ws.hnd[i]; ws.score[nm] = ws.score[i]; ws.order[nm] = ws.order[i]; ws.src[nm] = ws.src[i];
                    ++nm;
                };
                for (uint32_t i = 0; i < np; ++i) if (!feed(i, nm)) append(i);
                nm_primary = nm;
                int cap = in.secondary_cap;
                if (cap > 0) {
                    for (uint32_t i = np; i < np + ns; ++i) {
                        const bool in_primary = lfind_idx(ws, ws.hnd, nm_primary, ws.hnd[i]) < nm_primary;
                        if (!in_primary) { if (cap == 0) continue; --cap; }
                        if (!feed(i, nm)) append(i);
                    }
                }
                // OrderByDescending(Score).ThenBy(SourceRank).ThenBy(Order), stable (Program.cs:460)
                for (uint32_t k = 1; k < nm; ++k) {
                    H h = ws.hnd[k]; double sc = ws.score[k]; uint32_t od = ws.order[k]; uint32_t sr = ws.src[k];
                    int rk = source_rank(sr);
                    uint32_t pos = k;
                    while (pos > 0) {
                        double qs = ws.score[pos - 1];
                        int qr = source_rank(ws.src[pos - 1]);
                        bool q_first = (qs > sc) || (qs == sc && (qr < rk || (qr == rk && ws.order[pos - 1] <= od)));
                        if (q_first) break;
                        ws.hnd[pos] = ws.hnd[pos - 1]; ws.score[pos] = ws.score[pos - 1]; ws.order[pos] = ws.order[pos - 1]; ws.src[pos] = ws.src[pos - 1];
                        --pos;
                    }
                    ws.hnd[pos] = h; ws.score[pos] = sc; ws.order[pos] = od; ws.src[pos] = (uint8_t)sr;
                }
                n = nm;
            }
        }
        if (unsupported) { v.flags = TP_F_UNSUPPORTED; return; }
    }
    v.n_candidates = (uint16_t)(n > 0xFFFF ? 0xFFFF : n);

    TP_PIPE_STAMP(1)
    // ---------------- DetermineStyleQualityCap (Program.cs:702-710) ----------------
    int style_cap;
    {
        H strong[4];
        int ns = 0;
        for (uint32_t i = 0; i < n && ns < 4; ++i) {
            H h = ws.hnd[i];
            if (fl_bucket(hflags(t, h)) != B_STYLE || !(ws.score[i] >= 0.60)) continue;
            bool seen = false;
            for (int k = 0; k < ns; ++k) if (heq(ws, strong[k], h)) seen = true;
            if (!seen) strong[ns++] = h;
        }
        style_cap = ns < 2 ? 2 : ns;                             // Clamp(n, 2, 4)
    }

    TP_PIPE_STAMP(2)
    // ---------------- SelectTags, first half (Program.cs:625-655) + BuildEvidenceTags (Program.cs:721-729) ----------------
    // Both walk the candidates in order applying NormalizeTag once, so they share the walk.  The unique list is compacted
    // in place: entry nu is written only after candidate i >= nu has been read.
    uint32_t nu = 0, ndl = 0, ne = 0, ntop = 0, rbits = 0;
    for (uint32_t i = 0; i < n; ++i) {
        H h2 = norm_hop(t, ws, ws.hnd[i]);
        double sc = ws.score[i];
        uint32_t od = ws.order[i];
        uint32_t sr = ws.src[i];
        if (sr > 3) sr = 3;                                       // every other source ranks alike (Program.cs:712-717); the byte's upper bits are used below
        uint32_t f2 = hflags(t, h2);
        if (mode == 3) {
            // TopJoyTags: rawCandidates.OrderByDescending(Score).ThenBy(Order).Take(20).Select(c => c.Tag), a stable sort (Program.cs:860)
            uint32_t pos = ntop;
            while (pos > 0 && (ws.top_score[pos - 1] < sc || (ws.top_score[pos - 1] == sc && ws.top_order[pos - 1] > od))) --pos;
            if (pos < (uint32_t)TOP_TAGS) {
                for (uint32_t q = ntop < (uint32_t)TOP_TAGS ? ntop : (uint32_t)TOP_TAGS - 1; q > pos; --q) {
                    ws.top[q] = ws.top[q - 1]; ws.top_score[q] = ws.top_score[q - 1]; ws.top_order[q] = ws.top_order[q - 1];
                }
                ws.top[pos] = ws.hnd[i]; ws.top_score[pos] = sc; ws.top_order[pos] = od;
                if (ntop < (uint32_t)TOP_TAGS) ++ntop;
            }
        }
        if (f2 & FL_BLANK) continue;
        if (mode == 3) {                                          // rawTokens of BuildReviewRecord (Program.cs:827-837)
            if (h_is_dyn(h2)) rbits |= subject_bits_dyn(ws.norm + h_off(h2), h_len(h2)) >> 10;
            else {
                const uint32_t fs = t.flags[h_id(h2)];
                if (fs & FL_BANNED) for (int k = 0; k < N_BANNED; ++k) if (t.banned_ids[k] == h_id(h2)) rbits |= 1u << k;
                rbits |= (fs & (FL_SUBJ_MALE | FL_SUBJ_GIRL | FL_SUBJ_MULTI)) >> 10;
            }
        }
        if (!(sc < 0.28) && (mode == 0 || mode == 3)) {                          // evidence keeps invalid and dropped tags too
            if (!lfind(ws, ws.ev, ne, h2)) {
                if (ne >= ws.cap_e) overflow = true; else ws.ev[ne++] = h2;
            }
        }
        if (!(f2 & FL_VALID) || (f2 & FL_DROP)) {                 // "{n}:invalid_tag" / "{n}:policy_drop", made distinct at Program.cs:682
            if (!lfind(ws, ws.dl, ndl, h2)) ws.dl[ndl++] = h2;
            continue;
        }
        const uint32_t j = lfind_idx(ws, ws.hnd, nu, h2);
        if (j == nu) {
            ws.hnd[nu] = h2; ws.score[nu] = sc; ws.order[nu] = od; ws.src[nu] = (uint8_t)(sr | (fl_bucket(f2) << 4));
            ++nu;
            continue;
        }
        double osc = ws.score[j];
        uint32_t osr = ws.src[j] & 0xF;
        bool prefer_new = sc > osc;
        if (fabs(sc - osc) <= 0.03) prefer_new = source_rank(sr) < source_rank(osr);
        ws.score[j] = sc > osc ? sc : osc;                        // Math.Max
        if (od < ws.order[j]) ws.order[j] = od;                   // Math.Min
        ws.src[j] = (uint8_t)((prefer_new ? sr : osr) | (ws.src[j] & 0xF0));
    }
    v.n_dropped = (uint16_t)(ndl > 0xFFFF ? 0xFFFF : ndl);

    // OrderByDescending(Score).ThenBy(SourceRank).ThenBy(Order), Program.cs:651-655, stable: merged replies can tie on the whole key.
    for (uint32_t k = 0; k < nu; ++k) {
        double sc = ws.score[k];
        int rk = source_rank(ws.src[k] & 0xF);
        uint32_t od = ws.order[k];
        uint32_t pos = k;
        while (pos > 0) {
            uint32_t q = ws.sorted[pos - 1];
            double qs = ws.score[q];
            int qr = source_rank(ws.src[q] & 0xF);
            bool q_first = (qs > sc) || (qs == sc && (qr < rk || (qr == rk && ws.order[q] <= od)));
            if (q_first) break;
            ws.sorted[pos] = q;
            --pos;
        }
        ws.sorted[pos] = k;
    }

    TP_PIPE_STAMP(3)
    // ---------------- SelectTags, second half (Program.cs:661-699) ----------------
    uint32_t nr = 0;   // result list, built in ws.fin
    auto add_one = [&](H tag) -> bool {                           // AddOne, Program.cs:692-699
        H nn = norm_hop(t, ws, tag);
        uint32_t f = hflags(t, nn);
        if ((f & FL_BLANK) || nr >= (uint32_t)MAX_TOTAL_TAGS) return false;
        if (!(f & FL_VALID)) return false;
        if ((f & FL_DROP) || lfind(ws, ws.fin, nr, nn)) return false;
        ws.fin[nr++] = nn;
        return true;
    };
    // AddOne normalises its argument; trigger_id / girl_add_id / always_ids already are those normalised forms, so they
    // are added through the tail of AddOne only.
    auto add_normalised = [&](uint32_t id) {
        uint32_t f = t.flags[id];
        if ((f & FL_BLANK) || nr >= (uint32_t)MAX_TOTAL_TAGS) return;
        if (!(f & FL_VALID)) return;
        if ((f & FL_DROP) || lfind(ws, ws.fin, nr, (H)id)) return;
        ws.fin[nr++] = (H)id;
    };
    add_normalised(t.trigger_id);
    add_normalised(t.girl_add_id);
    for (uint32_t k = 0; k < t.n_always; ++k) add_normalised(t.always_ids[k]);
    {
        const int caps[6] = {3, 10, 8, 6, 4, style_cap};          // Program.cs:609-616
        for (uint32_t bkt = 0; bkt < 6; ++bkt) {                  // FillBucket, Program.cs:684-690
            int used = 0;
            for (uint32_t k = 0; k < nu; ++k) {
                uint32_t q = ws.sorted[k];
                const uint8_t sb = ws.src[q];
                if ((uint32_t)((sb >> 4) & 7) != bkt) continue;
                if (nr >= (uint32_t)MAX_TOTAL_TAGS || used >= caps[bkt]) break;
                if (add_one(ws.hnd[q])) { ++used; ws.src[q] = (uint8_t)(sb | 0x80); }   // bit 7: its tag is in the result
            }
        }
        for (uint32_t k = 0; k < nu; ++k) {                       // final sweep, Program.cs:677-680
            if (nr >= (uint32_t)MAX_TOTAL_TAGS) break;
            const uint32_t q = ws.sorted[k];
            if (ws.src[q] & 0x80) continue;                       // AddOne would find its tag selected already
            add_one(ws.hnd[q]);
        }
    }
    v.n_selected = (uint16_t)nr;
    if (mode == 1) { out.n_final = nr; return; }

    TP_PIPE_STAMP(4)
    // ---------------- EvaluateCoverage (Program.cs:1056-1095) and normalizedFinalSet (Program.cs:303) ----------------
    // Both start from Select(NormalizeTag).Where(non-blank).Distinct() of the selected tags: computed once, in place.
    uint32_t nf = 0;
    for (uint32_t i = 0; i < nr; ++i) {
        H h4 = norm_hop(t, ws, ws.fin[i]);
        if (hflags(t, h4) & FL_BLANK) continue;
        if (lfind(ws, ws.fin, nf, h4)) continue;
        ws.fin[nf++] = h4;
    }
    {
        bool has_trigger = false, has_girl = false, hair = false, eyes = false, outfit = false, pose = false, bg = false;
        for (uint32_t i = 0; i < nf; ++i) {
            H h = ws.fin[i];
            uint32_t f = hflags(t, h);
            if (h == (H)t.trigger_id) has_trigger = true;
            if (h == (H)t.girl_lit_id) has_girl = true;
            hair |= (f & FL_HAIR) != 0;
            eyes |= (f & FL_EYES) != 0;
            uint32_t bk = fl_bucket(f);
            outfit |= bk == B_OUTFIT;
            pose |= bk == B_POSE;
            bg |= bk == B_BACKGROUND;
        }
        v.coverage = (uint8_t)(((has_trigger && has_girl) ? TP_COV_IDENTITY : 0) | (hair ? TP_COV_HAIR : 0) | (eyes ? TP_COV_EYES : 0) |
                               (outfit ? TP_COV_OUTFIT : 0) | (pose ? TP_COV_POSE_ACTION : 0) | (bg ? TP_COV_BACKGROUND : 0));
    }
    uint32_t flags = rejected ? TP_F_LINE_REJECTED : 0;
    if (nr < 3) {                                                 // Program.cs:297-298
        flags |= TP_F_FALLBACK;
        nf = 0;
        for (uint32_t i = 0; i < t.n_fallback; ++i) {
            H h4 = t.next[t.fallback_ids[i]];
            if (t.flags[h_id(h4)] & FL_BLANK) continue;
            if (lfind(ws, ws.fin, nf, h4)) continue;
            ws.fin[nf++] = h4;
        }
    }

    TP_PIPE_STAMP(5)
    // ---------------- evidence closure (Program.cs:301-302) ----------------
    expand_implied(t, ws, ws.ev, ne, ws.cap_e, overflow);
    v.n_evidence = (uint16_t)((ne > 0xFFFF || !fired) ? 0xFFFF : ne);   // counted for callers that take the detail arrays
    for (uint32_t w = 0; w < t.lit_words; ++w) ws.ebits[w] = 0;
    bool ev_feet = false, ev_penis = false;
    for (uint32_t i = 0; i < ne; ++i) {
        H e = ws.ev[i];
        if (!h_is_dyn(e)) {
            uint32_t id = h_id(e);
            uint32_t li = t.lit[id];
            if (li != NO_ID) ws.ebits[li >> 5] |= 1u << (li & 31);
            uint32_t f = t.flags[id];
            ev_feet |= (f & FL_NFEET) != 0;                       // HasTag(evidenceTags, "feet"), Program.cs:309, 719
            ev_penis |= (f & FL_NPENIS) != 0;
        } else if (e & H_UNSTABLE) {
            H ne2 = norm_hop(t, ws, e);
            ev_feet |= ne2 == (H)t.feet_n;
            ev_penis |= ne2 == (H)t.penis_n;
        }
    }

    TP_PIPE_STAMP(6)
    // ---------------- final set closure (Program.cs:304-305) ----------------
    expand_implied(t, ws, ws.fin, nf, ws.cap_f, overflow);

    TP_PIPE_STAMP(7)
    // ---------------- InferActionTags (Program.cs:731-753) ----------------
    {
        uint32_t ni = 0;
        for (uint32_t i = 0; i < nf; ++i) {                       // Select(NormalizeTag).Where(non-blank).Distinct()
            H h5 = norm_hop(t, ws, ws.fin[i]);
            if (hflags(t, h5) & FL_BLANK) continue;
            if (lfind(ws, ws.fin, ni, h5)) continue;
            ws.fin[ni++] = h5;
        }
        nf = ni;
        for (uint32_t r = 0; r < t.n_rules; ++r) {
            const DRule& rule = t.rules[r];
            // MatchesActionRule, Program.cs:777-782
            if (rule.all_cnt && !all_set(t, ws, rule.all_off, rule.all_cnt)) continue;
            if (rule.any_cnt && !any_set(t, ws, rule.any_off, rule.any_cnt)) continue;
            if (any_set(t, ws, rule.forb_off, rule.forb_cnt)) continue;
            if (!lfind(ws, ws.fin, nf, (H)rule.parent_id)) { if (nf >= ws.cap_f) overflow = true; else ws.fin[nf++] = rule.parent_id; }
            if (fired) fired[r >> 5] |= 1u << (r & 31);
            for (uint32_t sidx = rule.sub_off; sidx < rule.sub_off + rule.sub_cnt; ++sidx) {
                const DSub& st = t.subs[sidx];
                if (!match_any_all(t, ws, st)) continue;
                if (!lfind(ws, ws.fin, nf, (H)st.tag_id)) { if (nf >= ws.cap_f) overflow = true; else ws.fin[nf++] = st.tag_id; }
                uint32_t bit = t.n_rules + sidx;
                if (fired) fired[bit >> 5] |= 1u << (bit & 31);
            }
        }
    }
    TP_PIPE_STAMP(8)
    // ---------------- EnsureRequiredFacets (Program.cs:755-775) ----------------
    for (uint32_t f = 0; f < t.n_facets; ++f) {
        const DFacet& facet = t.facets[f];
        uint32_t selected = facet.def_id;
        for (uint32_t r = 0; r < facet.rule_cnt; ++r) {
            const DSub& fr = t.facet_rules[facet.rule_off + r];
            if (!match_any_all(t, ws, fr)) continue;
            selected = fr.tag_id;
            if (facet_choice) facet_choice[f] = (uint16_t)r;
            break;
        }
        if (!lfind(ws, ws.fin, nf, (H)selected)) { if (nf >= ws.cap_f) overflow = true; else ws.fin[nf++] = selected; }
    }

    TP_PIPE_STAMP(9)
    // ---------------- verdicts on the final list (Program.cs:319, 329) ----------------
    {
        bool face = false, outfit = false;
        int safe = -1;
        for (uint32_t i = 0; i < nf; ++i) {
            uint32_t f = hflags(t, ws.fin[i]);
            uint32_t bk = fl_bucket(f);
            face |= bk == B_FACE;                                 // MissingCoreBuckets, Program.cs:1097-1101
            outfit |= bk == B_OUTFIT;
            if (safe < 0 && !h_is_dyn(ws.fin[i]) && (f & FL_SAFE)) safe = (int)((f >> FL_SAFE_IDX_SHIFT) & 31);   // DetectSafeModeRejectReason, Program.cs:942-945
        }
        if (!face || !outfit) flags |= TP_F_LOW;
        if (t.safe_mode && safe >= 0) { flags |= TP_F_SAFE_REJECT; v.safe_hit = (int16_t)safe; }
    }

    TP_PIPE_STAMP(10)
    // ---------------- the audit row's lists (Program.cs:334, 966) ----------------
    // keptTags is the final list as it stands here, before FormatCaptionSingleLine normalises it again: a copy goes to
    // ws.snap, which the closures no longer need (cap_snap >= cap_f >= nf).
    uint32_t akept = 0, alen[AUDIT_COLS] = {0, 0, 0, 0, 0};
    if (mode == 3) {
        akept = nf;
        for (uint32_t i = 0; i < nf; ++i) { H h = ws.fin[i]; ws.snap[i] = h; alen[AUDIT_KEPT] += hbytes(t, h) + (i ? 1u : 0u); }
        for (uint32_t i = 0; i < ne; ++i) alen[AUDIT_EVIDENCE] += hbytes(t, ws.ev[i]) + (i ? 1u : 0u);
        for (uint32_t i = 0; i < ndl; ++i) alen[AUDIT_DROPPED] += hbytes(t, ws.dl[i]) + AUDIT_REASON_LEN + (i ? 1u : 0u);
        for (uint32_t i = 0; i < ntop; ++i) alen[AUDIT_TOP] += hbytes(t, ws.top[i]) + (i ? 1u : 0u);
    }

    TP_PIPE_STAMP(11)
    // ---------------- FormatCaptionSingleLine (Program.cs:971-974) and GetUnknownTags (Program.cs:817-824) ----------------
    // Both take Select(NormalizeTag).Where(non-blank).Distinct() of the final list.
    uint32_t nc = 0, bytes = 0, unknown = 0;
    bool footjob = false;
    for (uint32_t i = 0; i < nf; ++i) {
        H h6 = norm_hop(t, ws, ws.fin[i]);
        uint32_t f = hflags(t, h6);
        if (f & FL_BLANK) continue;
        if (lfind(ws, ws.fin, nc, h6)) continue;
        ws.fin[nc++] = h6;
        bytes += hbytes(t, h6);
        if (t.known_active && (h_is_dyn(h6) || !(f & FL_KNOWN))) { alen[AUDIT_UNKNOWN] += hbytes(t, h6) + (unknown ? 1u : 0u); ++unknown; }
        if (h6 == (H)t.footjob_n) footjob = true;                 // HasTag(finalTags, "footjob"), Program.cs:311
    }
    if (overflow) { v.flags = TP_F_UNSUPPORTED; out.n_final = 0; out.caption_len = 0; return; }
    if (ev_feet && ev_penis) { flags |= TP_F_ELIGIBLE; if (footjob) flags |= TP_F_FOOTJOB; }
    v.flags = (uint8_t)flags;
    v.n_tags = (uint16_t)(nc > 0xFFFF ? 0xFFFF : nc);
    v.n_unknown = (uint16_t)(unknown > 0xFFFF ? 0xFFFF : unknown);
    out.n_final = nc;
    if (mode == 3) {
        out.audit_cnt[AUDIT_KEPT] = akept; out.audit_cnt[AUDIT_EVIDENCE] = ne; out.audit_cnt[AUDIT_DROPPED] = ndl;
        out.audit_cnt[AUDIT_TOP] = ntop; out.audit_cnt[AUDIT_UNKNOWN] = nc;
        out.review_bits = rbits;
        for (int c = 0; c < AUDIT_COLS; ++c) out.audit_len[c] = alen[c];
    }
    TP_PIPE_STAMP(14)
    TP_PIPE_STAMP_PRINT
    out.caption_len = bytes + (nc ? 2 * (nc - 1) : 0) + t.newline_len;    // string.Join(", ", …) + Environment.NewLine, Program.cs:316-317
}

}  // namespace tp
