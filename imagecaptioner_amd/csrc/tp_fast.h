// Wave-per-file form of the tag pipeline: the 64 lanes of one wavefront cooperate on one file whose state lives in
// LDS and registers.  Lanes own bytes of the line, then tokens, then candidates, then final tags; set operations are
// ballots or a small LDS hash set; sequential parts of the reference (the bucket fills of SelectTags, the ordered
// growth of the final set) are turned into prefix-popcount arithmetic or short wave-uniform loops.
//
// It takes the COMMON shape of a file and hands anything else to the general one-thread-per-file kernel
// (tp_pipeline.h) by returning false (whatever it wrote by then is overwritten by that kernel):
//   * a line of at most FAST_LINE_MAX bytes (LONG_LINE_MAX in the long-file stage) without "positive:" / "negative:" and
//     without '`' (so no POSITIVE:/NEGATIVE:/code-fence handling, Program.cs:592-595) and without a carriage return of its
//     own inside the trimmed text (line feeds there are the blanks they become, Program.cs:595; "\r\n" pairs there are taken
//     out by the stages behind the first, Program.cs:591);
//   * at most FAST_TOK_MAX tokens, every kept token an interned string;
//   * no two different candidates that NormalizeTag maps to one selected tag (their AddOne order would matter);
//   * lists that fit their LDS capacities.
// Both kernels restate the same reference span (Program.cs:223-229, 292-329, 574-605) and must give identical
// results; citations below name the reference lines each step stands for.
//
// All wave collectives (ballot / bcast / sum / sync) are executed in wave-uniform control flow: every branch around one
// depends only on ballot results or kernel arguments.  `W` supplies the collectives: hardware intrinsics on the GPU,
// a lock-step emulation in the host-side logic test.
#pragma once
#include <math.h>

#include "../../include/tagpipe.h"
#include "tp_common.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define TP_UNROLL _Pragma("unroll")
#else
#define TP_UNROLL
#endif

// Diagnostic builds stop the pipeline after section TP_STOP_AFTER (the value handed to TP_STOP keeps the section's results
// alive) so that the kernel's time up to each section can be measured.
#ifndef TP_STOP_AFTER
#define TP_STOP_AFTER (-1)
#endif
#define TP_STOP(k, expr)                                                                                     \
    if (TP_STOP_AFTER == (k)) {                                                                              \
        fin32[lane] = (uint32_t)(expr);                                                                      \
        out.verdict = tp_verdict{};                                                                          \
        out.n_final = 0;                                                                                     \
        out.caption_len = 0;                                                                                 \
        return true;                                                                                         \
    }

namespace tp {

constexpr uint32_t FAST_LINE_MAX = 2048;   // bytes of one line held in LDS
constexpr uint32_t FAST_TOK_MAX = 128;     // tokens per line (two per lane)
constexpr uint32_t FAST_FIN_MAX = 128;     // final tags (two per lane)
constexpr uint32_t FAST_SET_SLOTS = 256;   // LDS hash set
constexpr uint32_t FAST_LIT_WORDS = 128;   // evidence bitset words (4096 rule literals)
constexpr uint32_t FAST_FIRED_WORDS = 64;  // fired-rule bitmap words (2048 rules + subtypes)
constexpr uint32_t FAST_CAPTION_MAX = 2048; // caption bytes (the emit kernel puts a caption together in LDS)

// The part of a file's LDS state that the back half of the pipeline (coverage, closures, rules, facets, verdict) works on.
struct alignas(16) BackLds {
    uint32_t fin[FAST_FIN_MAX];
    uint32_t fattr[FAST_FIN_MAX];          // flags of fin[k] (files on the settled-tags path)
    uint32_t flen[FAST_FIN_MAX];           // byte length of fin[k]
    uint32_t set_key[FAST_SET_SLOTS];
    uint32_t set_min[FAST_SET_SLOTS];
    uint32_t ebits[FAST_LIT_WORDS];
    uint32_t firedw[FAST_FIRED_WORDS];
    uint32_t fhit[FAST_FRULES_MAX / 32];   // bit p: facet rule p matches the evidence
};
// ... plus the line and its tokens, sized for LINE bytes and TOK tokens.
template <uint32_t LINE, uint32_t TOK>
struct alignas(16) LineLds : BackLds {
    static constexpr uint32_t LINE_CAP = LINE, TOK_CAP = TOK;
    uint8_t line[LINE + 64];     // 16 spare bytes in front (the byte before the line is read), the line at its 16-byte phase, tail padding
    uint16_t cpos[TOK];
    uint32_t tspec[TOK / 32];    // bit k: token k holds a byte that needs the byte-wise normaliser
    uint32_t thigh[TOK / 32];    // bit k: token k holds a byte >= 0x80
};
using FastLds = LineLds<FAST_LINE_MAX, FAST_TOK_MAX>;

struct FastOut {
    tp_verdict verdict;
    uint32_t n_final;
    uint32_t caption_len;
};
// Audit mode: where a wave stage leaves a file's other lists as interned ids (row: AUD_STRIDE words laid out by the AUD_*
// constants of tp_common.h - the evidence set in insertion order, the dropped candidates with their reason, the first TOP_TAGS
// candidates, the final list when the caption's list differs from it), their entry counts and text lengths per column (cnt,
// len: AUDIT_COLS entries each) and the review record's bits; what the general kernel reports in the same arrays for its files.
struct LeanAudit { uint32_t* row; uint32_t* cnt; uint32_t* len; uint32_t* rbits; };

TP_HD uint32_t popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(x);
#else
    return (uint32_t)__builtin_popcountll(x);
#endif
}
TP_HD uint32_t ctz64(uint64_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(__ffsll((long long)x) - 1);
#else
    return (uint32_t)__builtin_ctzll(x);
#endif
}
TP_HD uint32_t clz64(uint64_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clzll((long long)x);
#else
    return (uint32_t)__builtin_clzll(x);
#endif
}

// Branch-free byte predicates on values < 0x80 (results are 0 / 1).
TP_HD uint32_t ws32(uint32_t c) { return (uint32_t)(c == 0x20u) | (uint32_t)(c - 9u <= 4u); }      // char.IsWhiteSpace below U+0080
TP_HD uint32_t upper32(uint32_t c) { return (uint32_t)(c - 'A' <= 25u); }
TP_HD uint32_t alnum32(uint32_t c) { return (uint32_t)(c - 'a' <= 25u) | (uint32_t)(c - '0' <= 9u); }   // lower-case letters and digits
// the other characters ValidTagRegex allows after the first one: _ ( ) ! ' / -   (Program.cs:40)
TP_HD uint32_t tagextra32(uint32_t c) {
    constexpr uint64_t M = (1ull << ('!' - 32)) | (1ull << ('\'' - 32)) | (1ull << ('(' - 32)) | (1ull << (')' - 32)) | (1ull << ('-' - 32)) | (1ull << ('/' - 32)) | (1ull << ('_' - 32));
    return (uint32_t)(M >> ((c - 32u) & 63u)) & (uint32_t)(c - 32u < 64u);
}
TP_HD uint32_t ld32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
TP_HD uint64_t ld64u(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
// Lower-cases four ASCII bytes at once; a byte >= 0x80 only disturbs the bytes above it.
TP_HD uint32_t lower4(uint32_t x) {
    uint32_t up = ((x + 0x3F3F3F3Fu) & ~(x + 0x25252525u)) & 0x80808080u;
    return x | (up >> 2);
}
// Four bytes at a time: the result holds 0x80 in every byte position where the test is true.
constexpr uint32_t rep4(char c) { return 0x01010101u * (uint32_t)(uint8_t)c; }
TP_HD uint32_t eq4(uint32_t x, uint32_t pat) { uint32_t y = x ^ pat; return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu); }   // exact for any bytes
TP_HD uint32_t ge4(uint32_t x, uint32_t k) { return (x + 0x01010101u * (0x80u - k)) & 0x80808080u; }                             // bytes of x below 0x80 only
TP_HD uint32_t ws4(uint32_t x) { return eq4(x, 0x20202020u) | (ge4(x, 9) & ~ge4(x, 14)); }
TP_HD uint32_t win4(uint32_t lo, uint32_t hi, uint32_t k) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * k)); }        // bytes k .. k+3 of hi:lo
// Byte mask of the positions p .. p+3 that lie below `limit`.
TP_HD uint32_t keep4(uint32_t p, uint32_t limit) {
    int nv = (int)limit - (int)p;
    nv = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
    return nv == 4 ? 0xFFFFFFFFu : ((1u << (8 * nv)) - 1u);
}
// The same on eight bytes.
constexpr uint64_t REP8_01 = 0x0101010101010101ull, REP8_7F = 0x7F7F7F7F7F7F7F7Full, REP8_80 = 0x8080808080808080ull, REP8_US = REP8_01 * (uint64_t)'_';
TP_HD uint64_t eq8(uint64_t x, uint64_t pat) { uint64_t y = x ^ pat; return ~(((y & REP8_7F) + REP8_7F) | y | REP8_7F); }
TP_HD uint64_t ge8(uint64_t x, uint32_t k) { return (x + REP8_01 * (0x80u - k)) & REP8_80; }                                    // bytes of x below 0x80 only
TP_HD uint64_t ws8(uint64_t x) { return eq8(x, REP8_01 * 0x20u) | (ge8(x, 9) & ~ge8(x, 14)); }
TP_HD uint64_t lower8w(uint64_t x) { uint64_t up = ((x + REP8_01 * 0x3Fu) & ~(x + REP8_01 * 0x25u)) & REP8_80; return x | (up >> 2); }   // bytes below 0x80 only
// '_' in the bytes that flags (0x80 per byte) marks
TP_HD uint64_t to_us8(uint64_t x, uint64_t flags) { uint64_t m = (flags - (flags >> 7)) | flags; return (x & ~m) | (REP8_US & m); }
TP_HD uint32_t ctz32(uint32_t x) { return (uint32_t)__builtin_ctz(x); }   // x != 0
TP_HD uint32_t clz32(uint32_t x) { return (uint32_t)__builtin_clz(x); }   // x != 0
// Byte classes of the line scan, one table byte per input byte (the table sits in LDS: DevTables-independent).
constexpr uint32_t CL_WS = 1, CL_COMMA = 2, CL_SPEC = 4, CL_US = 8, CL_COLON = 16, CL_BAD = 32, CL_S = 64, CL_O = 128;
TP_HD uint8_t byte_class(uint32_t c) {
    uint32_t k = 0;
    if (c == 0x20 || (c >= 9 && c <= 13)) k |= CL_WS;
    if (c == ',') k |= CL_COMMA;
    if ((c >= 'A' && c <= 'Z') || c == '"' || c == '\'' || c == ';' || c == '.') k |= CL_SPEC;   // what NormalizeTag changes or strips (Program.cs:1150-1154)
    if (c == '_') k |= CL_US;
    if (c >= 0x80) k |= CL_SPEC | CL_US;                                // both at once: a byte of a character outside ASCII
    if (c == ':') k |= CL_COLON;
    if (c == '`' || c == '\r') k |= CL_BAD;                              // a code fence; a carriage return ("\r\n" shrinks to one character, Program.cs:591): scan_line takes those out first
    if (c == 's' || c == 'S') k |= CL_S;
    if (c == 'o' || c == 'O') k |= CL_O;
    return (uint8_t)k;
}
// The classes of the four bytes of x, in the same byte positions.
TP_HD uint32_t class4(const uint8_t* cls, uint32_t x) {
    return (uint32_t)cls[x & 0xFFu] | ((uint32_t)cls[(x >> 8) & 0xFFu] << 8) | ((uint32_t)cls[(x >> 16) & 0xFFu] << 16) | ((uint32_t)cls[x >> 24] << 24);
}
constexpr uint32_t c4(char a, char b, char c, char d) { return (uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) | ((uint32_t)(uint8_t)d << 24); }
// The low min(max(n - 8k, 0), 8) bytes of word k of an n-byte string.
TP_HD uint64_t word_mask(uint32_t n, uint32_t k) {
    int rem = (int)n - (int)(8 * k);
    rem = rem < 0 ? 0 : (rem > 8 ? 8 : rem);
    return rem == 8 ? ~0ull : ((1ull << (8 * rem)) - 1);
}


// ---- LDS hash set: key -> smallest order value seen ---------------------------------------------------------------
template <class W>
TP_HD void set_clear(W& w, BackLds& L) {
    static_assert(FAST_SET_SLOTS == 256 && NO_ID == 0xFFFFFFFFu, "one 16-byte store per lane and array");
    w.fill16(&L.set_key[4 * w.lane()], 0xFFFFFFFFu);
    w.fill16(&L.set_min[4 * w.lane()], 0xFFFFFFFFu);
    w.sync();
}
// Returns the slot, or NO_ID when the table is full.  `created` is true for exactly one caller per key.
template <class W>
TP_HD uint32_t set_put(W& w, BackLds& L, uint32_t key, uint32_t order, bool& created) {
    // Double hashing: the wavefront stays in this loop until its slowest lane is done, and an odd step that differs from
    // key to key keeps the longest probe sequence short where a step of one would let keys pile up behind one another.
    const uint32_t hsh = key * 2654435761u;
    uint32_t s = hsh >> 24;
    const uint32_t step = ((hsh >> 15) & 0xFEu) | 1u;
    created = false;
    for (uint32_t probe = 0; probe < FAST_SET_SLOTS; ++probe) {
        uint32_t old = w.lds_cas(&L.set_key[s], NO_ID, key);
        if (old == NO_ID || old == key) {
            created = old == NO_ID;
            w.lds_min(&L.set_min[s], order);
            return s;
        }
        s = (s + step) & (FAST_SET_SLOTS - 1);
    }
    return NO_ID;
}

// The slot of a key that was put, NO_ID when it was not (no writes: call it only between syncs that separate it from puts).
TP_HD uint32_t set_find(const BackLds& L, uint32_t key) {
    const uint32_t hsh = key * 2654435761u;
    uint32_t s = hsh >> 24;
    const uint32_t step = ((hsh >> 15) & 0xFEu) | 1u;
    for (uint32_t probe = 0; probe < FAST_SET_SLOTS; ++probe) {
        const uint32_t k = L.set_key[s];
        if (k == key) return s;
        if (k == NO_ID) return NO_ID;
        s = (s + step) & (FAST_SET_SLOTS - 1);
    }
    return NO_ID;
}

// First-occurrence filter over up to 128 (key, valid) pairs held two per lane (element index = blk * 64 + lane); only the
// first nblk blocks are looked at.  On return first[blk] marks the elements that are the first with their key; returns
// false if the set overflowed.
template <class W>
TP_HD bool dedupe_first(W& w, BackLds& L, const uint32_t key[2], const bool valid[2], bool first[2], int nblk = 2) {
    set_clear(w, L);
    uint32_t slot[2] = {NO_ID, NO_ID};
    bool full = false;
    TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
        if (valid[blk]) {
            bool created;
            slot[blk] = set_put(w, L, key[blk], (uint32_t)blk * 64 + w.lane(), created);
            if (slot[blk] == NO_ID) full = true;
        }
    }
    w.sync();
    first[0] = first[1] = false;
    TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) first[blk] = valid[blk] && slot[blk] != NO_ID && L.set_min[slot[blk]] == (uint32_t)blk * 64 + w.lane();
    bool any_full = w.ballot(full) != 0;
    w.sync();
    return !any_full;
}

// The same with an order of the caller's choosing: first[blk] marks the element whose val is the lowest among those with
// its key (vals are distinct).
template <class W>
TP_HD bool dedupe_lowest(W& w, BackLds& L, const uint32_t key[2], const bool valid[2], const uint32_t val[2], bool first[2], int nblk = 2) {
    set_clear(w, L);
    uint32_t slot[2] = {NO_ID, NO_ID};
    bool full = false;
    TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
        if (valid[blk]) {
            bool created;
            slot[blk] = set_put(w, L, key[blk], val[blk], created);
            if (slot[blk] == NO_ID) full = true;
        }
    }
    w.sync();
    first[0] = first[1] = false;
    TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) first[blk] = valid[blk] && slot[blk] != NO_ID && L.set_min[slot[blk]] == val[blk];
    bool any_full = w.ballot(full) != 0;
    w.sync();
    return !any_full;
}

TP_HD bool lds_match(const uint8_t* s, uint32_t pos, uint32_t end, const char* pat, uint32_t plen) {
    if (pos + plen > end) return false;
    for (uint32_t k = 0; k < plen; ++k) {
        uint8_t c = s[pos + k];
        if (c == '\n') c = ' ';                                    // LooksBad sees the line after Replace("\n", " "), Program.cs:595
        if (lower8(c) != (uint8_t)pat[k]) return false;
    }
    return true;
}

// Lookup of a normalised string of n <= 32 bytes, given as zero-padded little-endian words, in the slot table: both cache
// lines of the first slot are fetched at once, so a first-probe hit costs one memory round trip and brings the record along.
TP_HD uint32_t slot_find32(const DevTables& t, uint64_t hsh, uint32_t n, uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, Rec& rec, uint32_t& kflags) {
    uint32_t slot = (uint32_t)hsh & t.ht_mask;
    const uint32_t h32 = (uint32_t)(hsh >> 32);
    for (;;) {
        const Slot* sl = t.slots + slot;
        const uint32_t hash = sl->hash, slen = sl->slen, val = sl->val;
        kflags = sl->kflags;
        uint64_t k[4];
        __builtin_memcpy(k, sl->key, 32);
        rec = sl->rec;
        if (slen == SLOT_EMPTY) return NO_ID;
        if (hash == h32 && slen == n && ((k[0] ^ w0) | (k[1] ^ w1) | (k[2] ^ w2) | (k[3] ^ w3)) == 0) return val;
        slot = (slot + 1) & t.ht_mask;
    }
}

// One token of the line (LDS bytes [i, j)) -> what NormalizeTag makes of it (Program.cs:1149-1158).
//   returns 0: empty after trimming, skipped (Program.cs:1183-1185)
//           1: interned, id in `out`
//           2: not in any table
//           3: the file is not for this kernel
//           4: (only with miss_words) in no table and dropped under strict vocabulary; its normalised bytes, at most 32, in miss_words
// On a hit `rec` receives the record of `out`.
// `high`: the token holds bytes of characters outside ASCII.  This kernel does not decode them; it accepts the token only
// when its bytes as they stand are a known string that the normalisation before the alias lookup leaves alone: then that
// normalisation of the token yields this very string.
// `special` is false when the token is known to hold nothing for the normaliser to change apart from one leading
// white-space byte: then its bytes are the normalised string and are hashed straight from the line.  Otherwise the
// bytes go through the byte-wise normaliser; they are packed into 8-byte words as they are produced (the first 32 bytes
// stay in registers), so hashing costs one multiply per word and a probe hit is verified word by word against
// DevTables::str_words.
TP_HD int fast_token(const DevTables& t, const uint8_t* s, uint32_t i, uint32_t j, bool special, bool high, uint32_t& out, Rec& rec, bool& valid_dyn,
                     uint32_t& dyn_len, uint64_t* miss_words = nullptr) {
    if (high && special) return 3;
    // Up to 32 bytes that are, or with a few byte-wise changes become, the normalised string: hashed straight from the line.
    uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    uint32_t n = 0;
    bool have_words = false;
    if (!special) {
        const uint32_t i1 = i + ((uint32_t)(i < j) & ws32(s[i]));
        n = j - i1;
        if (n == 0) return 0;
        if (n <= 32) {
            w0 = ld64u(s + i1) & word_mask(n, 0); w1 = ld64u(s + i1 + 8) & word_mask(n, 1); w2 = ld64u(s + i1 + 16) & word_mask(n, 2);
            w3 = ld64u(s + i1 + 24) & word_mask(n, 3);
            have_words = true;
        }
    } else {
        // Capital letters and single white-space characters inside the tag are the common case ("Long hair"): without quotes at
        // the ends, trailing ';' or '.', and runs of white space and underscores, NormalizeTag (Program.cs:1150-1154) maps byte
        // to byte: lower case, white space to '_'.
        while (i < j && ws32(s[i])) ++i;
        while (j > i && ws32(s[j - 1])) --j;
        if (i == j) return 0;
        n = j - i;
        if (n <= 32 && !is_quote8(s[i]) && !is_quote8(s[j - 1]) && !is_punct8(s[j - 1])) {
            w0 = ld64u(s + i) & word_mask(n, 0); w1 = ld64u(s + i + 8) & word_mask(n, 1); w2 = ld64u(s + i + 16) & word_mask(n, 2);
            w3 = ld64u(s + i + 24) & word_mask(n, 3);
            const uint64_t f0 = ws8(w0), f1 = ws8(w1), f2 = ws8(w2), f3 = ws8(w3);                     // 0x80 in the white-space bytes
            const uint64_t u0 = f0 | eq8(w0, REP8_US), u1 = f1 | eq8(w1, REP8_US), u2 = f2 | eq8(w2, REP8_US), u3 = f3 | eq8(w3, REP8_US);
            const uint64_t runs = (u0 & (u0 << 8)) | (u1 & ((u1 << 8) | (u0 >> 56))) | (u2 & ((u2 << 8) | (u1 >> 56))) | (u3 & ((u3 << 8) | (u2 >> 56)));
            if (runs == 0) {
                w0 = to_us8(lower8w(w0), f0); w1 = to_us8(lower8w(w1), f1); w2 = to_us8(lower8w(w2), f2); w3 = to_us8(lower8w(w3), f3);
                have_words = true;
            }
        }
    }
    if (have_words) {
        const uint64_t h = hash_word(hash_word(hash_word(hash_word(HASH_SEED, w0, 0), w1, 1), w2, 2), w3, 3);   // absent words are zero
        uint32_t kflags = 0;
        uint32_t id = slot_find32(t, hash_finish(h, n), n, w0, w1, w2, w3, rec, kflags);
        if (id != NO_ID && (!high || (kflags & SLOT_KEY_BASIC))) { out = id; return 1; }
        // These bytes are the normalised string, and every interned string of up to 32 bytes is a key of the slots: a miss
        // means the tag is in no table.  Under strict vocabulary it is dropped whatever it looks like (Program.cs:1121-1133).
        if (id == NO_ID && !high && t.drop_unknown) {
            valid_dyn = false; dyn_len = n;
            if (miss_words) { miss_words[0] = w0; miss_words[1] = w1; miss_words[2] = w2; miss_words[3] = w3; return 4; }   // ... for a caller that tells such strings apart
            return 2;
        }
    }
    if (high) return 3;
    while (i < j && ws32(s[i])) ++i;
    while (j > i && ws32(s[j - 1])) --j;
    if (i == j) return 0;
    while (i < j && is_quote8(s[i])) ++i;
    while (j > i && is_quote8(s[j - 1])) --j;
    while (j > i && is_punct8(s[j - 1])) --j;
    w0 = w1 = w2 = w3 = 0;
    n = 0;
    uint64_t cur = 0, h = HASH_SEED;
    uint32_t last_us = 0, valid = 1;
    for (uint32_t p = i; p < j; ++p) {                       // \s+ -> "_", _+ -> "_", lower case (Program.cs:1152-1154)
        uint32_t c = s[p];
        const uint32_t us = ws32(c) | (uint32_t)(c == '_');
        const uint32_t skip = us & last_us;
        last_us = us;
        c = us ? (uint32_t)'_' : c + (upper32(c) << 5);
        valid &= skip | alnum32(c) | ((uint32_t)(n != 0) & tagextra32(c));
        cur |= skip ? 0ull : (uint64_t)c << (8 * (n & 7));
        n += 1u - skip;
        if (!skip && (n & 7) == 0) {
            h = hash_word(h, cur, (n >> 3) - 1);
            const uint32_t wi = (n >> 3) - 1;
            w0 = wi == 0 ? cur : w0; w1 = wi == 1 ? cur : w1; w2 = wi == 2 ? cur : w2; w3 = wi == 3 ? cur : w3;
            cur = 0;
        }
    }
    if (n & 7) {
        h = hash_word(h, cur, n >> 3);
        const uint32_t wi = n >> 3;
        w0 = wi == 0 ? cur : w0; w1 = wi == 1 ? cur : w1; w2 = wi == 2 ? cur : w2; w3 = wi == 3 ? cur : w3;
    }
    const uint64_t hsh = hash_finish(h, n);
    const uint32_t nwords = (n + 7) >> 3;
    uint32_t slot = (uint32_t)hsh & t.ht_mask;
    for (;;) {
        uint64_t hh = t.ht_hash[slot];
        if (hh == 0) break;
        if (hh == hsh) {
            HtInfo info = t.ht_info[slot];
            if (info.slen == n) {
                const uint64_t* q = t.str_words + info.woff;
                bool same = true;
                if (nwords > 0 && q[0] != w0) same = false;
                if (nwords > 1 && q[1] != w1) same = false;
                if (nwords > 2 && q[2] != w2) same = false;
                if (nwords > 3 && q[3] != w3) same = false;
                if (same && nwords > 4) {                        // a long token: re-normalise the tail against the stored words
                    uint32_t k = 0;
                    bool lu = false;
                    uint64_t acc = 0;
                    for (uint32_t p = i; p < j && same; ++p) {
                        uint8_t c = s[p];
                        if (is_ws8(c) || c == '_') { if (lu) continue; lu = true; c = '_'; }
                        else { lu = false; c = lower8(c); }
                        acc |= (uint64_t)c << (8 * (k & 7));
                        if ((++k & 7) == 0) { if (k > 32 && q[(k >> 3) - 1] != acc) same = false; acc = 0; }
                    }
                    if (same && (k & 7) && q[k >> 3] != acc) same = false;
                }
                if (same) { out = info.val; rec = t.rec[info.val]; return 1; }
            }
        }
        slot = (slot + 1) & t.ht_mask;
    }
    valid_dyn = valid != 0 && n > 0;
    dyn_len = n;
    return 2;
}

// MatchesActionRule / MatchesSubtypeRule / MatchesFacetRule on the LDS evidence bitset (Program.cs:777-794).
TP_HD bool fbit(const BackLds& L, uint32_t idx) { return (L.ebits[idx >> 5] >> (idx & 31)) & 1u; }
TP_HD bool f_all(const DevTables& t, const BackLds& L, uint32_t off, uint32_t cnt) {
    for (uint32_t k = 0; k < cnt; ++k) if (!fbit(L, t.pool[off + k])) return false;
    return true;
}
TP_HD bool f_any(const DevTables& t, const BackLds& L, uint32_t off, uint32_t cnt) {
    for (uint32_t k = 0; k < cnt; ++k) if (fbit(L, t.pool[off + k])) return true;
    return false;
}
TP_HD bool f_match_sub(const DevTables& t, const BackLds& L, const DSub& r) {
    if (r.any_cnt && !f_any(t, L, r.any_off, r.any_cnt)) return false;
    if (r.all_cnt && !f_all(t, L, r.all_off, r.all_cnt)) return false;
    return true;
}

// One record of the rule cache against the evidence bitset: every `all` literal present, some `any` literal present (when
// there are any), no forbidden literal present (MatchesActionRule / MatchesSubtypeRule / MatchesFacetRule, Program.cs:777-794).
TP_HD bool rc_match(const BackLds& L, const uint32_t* rc, uint32_t rec) {
    const uint32_t* r = rc + rec;
    const uint32_t cnts = r[2], n_all = cnts & 255u, n_any = (cnts >> 8) & 255u, n_forb = (cnts >> 16) & 255u, tot = n_all + n_any + n_forb;
    const uint32_t* lits = tot <= 4 ? r + 4 : rc + (r[3] & 0x7FFFu);
    bool ok_all = true, ok_any = n_any == 0, ok_forb = true;
    for (uint32_t i = 0; i < tot; ++i) {
        const bool bit = fbit(L, lits[i]);
        if (i < n_all) ok_all = ok_all && bit;
        else if (i < n_all + n_any) ok_any = ok_any || bit;
        else ok_forb = ok_forb && !bit;
    }
    return ok_all && ok_any && ok_forb;
}

// Appends `id` to the final list unless present.  own0/own1 mirror L.fin[lane], L.fin[lane + 64] in registers.
// Wave-uniform: `id`, `nf` and the return value are the same in every lane.  Returns false on overflow.
// With in_set the LDS set (which then holds every element of the list with order 0) learns the new element too.
template <class W>
TP_HD bool fin_append(W& w, BackLds& L, uint32_t& own0, uint32_t& own1, uint32_t& nf, uint32_t id, bool in_set, uint32_t fin_max, uint32_t attr = 0,
                      uint32_t len = 0) {
    if (w.ballot(own0 == id || own1 == id) != 0) return true;
    if (nf >= fin_max) return false;
    if (w.lane() == (nf & 63)) { if (nf < 64) own0 = id; else own1 = id; }
    bool full = false;
    if (w.lane() == 0) {
        L.fin[nf] = id;
        L.fattr[nf] = attr;
        L.flen[nf] = len;
        if (in_set) { bool created; full = set_put(w, L, id, 0, created) == NO_ID; }
    }
    ++nf;
    return in_set ? w.ballot(full) == 0 : true;
}

// What the line scan leaves behind: the line in LDS (s), its trimmed extent [a, b), the comma count (comma positions are in
// L.cpos, per-token marks in L.tspec / L.thigh) and whether NormalizeRawTagLine rejects the line.
struct Scan { const uint8_t* s; uint32_t a, b, ncomma; bool rejected; };

// S1 + S2: the line goes to LDS, is trimmed and split at its commas (NormalizeRawTagLine, LooksBad, SplitTags:
// Program.cs:589-596, 1141-1147, 1181-1187).  Returns false when the file needs the general kernel.
// CRLF: take "\r\n" pairs inside the text out of the LDS copy and scan again (the first wave stage leaves such files to the second).
template <bool CRLF, bool WIDE, class W, class LDS>
TP_HD __attribute__((always_inline)) bool scan_line(W& w, LDS& L, const uint8_t* cls, const uint8_t* text, uint32_t len, Scan& sc) {
    const uint32_t lane = w.lane();
    if (len > LDS::LINE_CAP) return false;
    w.tick(0);
    // ---- S1: line -> LDS in aligned 16-byte pieces -------------------------------------------------------------------
    const uint32_t shift = (uint32_t)((uintptr_t)text & 15);
    {
        const uint8_t* abase = text - shift;
        uint32_t pieces = (shift + len + 15) >> 4;
        for (uint32_t c = lane; c < pieces; c += 64) w.copy16(L.line + 16 + 16 * c, abase + 16 * c);
        if (lane < LDS::TOK_CAP / 32) { L.tspec[lane] = 0; L.thigh[lane] = 0; }
    }
    w.sync();
    w.tick(1);
    const uint8_t* s = L.line + 16 + shift;

    // ---- S2: trim (NormalizeRawTagLine, Program.cs:589-596): first non-blank byte from the front, last from the back ----
    constexpr uint32_t B4 = 0x01010101u;                            // bit 0 of each of four bytes
    uint32_t a = 0, b = 0;
    bool blank_line = true;
    for (uint32_t base = 0; base < len; base += 256) {
        const uint32_t p = base + 4 * lane;
        const uint32_t keep = keep4(p, len) & B4;
        const uint32_t nonws = ~class4(cls, ld32u(s + (p < len ? p : len))) & keep;      // CL_WS is bit 0
        const uint64_t m = w.ballot(nonws != 0);
        if (m) { const uint32_t q = ctz64(m); a = base + 4 * q + (ctz32(w.bcast(nonws, q)) >> 3); blank_line = false; break; }
    }
    if (!blank_line) {
        for (uint32_t base = ((len - 1) >> 8) << 8;; base -= 256) {
            const uint32_t p = base + 4 * lane;
            const uint32_t keep = keep4(p, len) & B4;
            const uint32_t nonws = ~class4(cls, ld32u(s + (p < len ? p : len))) & keep;
            const uint64_t m = w.ballot(nonws != 0);
            if (m) { const uint32_t q = 63 - clz64(m); b = base + 4 * q + ((31 - clz32(w.bcast(nonws, q))) >> 3) + 1; break; }
            if (base == 0) break;                                  // not reached: a non-blank byte exists
        }
    }

    // commas, tokens that need the byte-wise normaliser, bytes this kernel leaves to the general one, LooksBad (Program.cs:1141-1147)
    uint32_t ncomma = 0;
    bool refusal = false, keyword = false;
    uint64_t any_blank = 0, any_bad = 0, any_high = 0;
    // WIDE: 16 bytes a lane and step (a kilobyte a step) - the classes of a lane's four words are looked up side by side, the commas
    // are numbered by one wave prefix sum over the lanes' counts, and the rarer marks go by the set bits of a lane's 16-bit masks -
    // for the stages whose lines are long (measured on files of more than 128 tags: the scan was a third of a file's cycles at 256
    // bytes a step; c5 0.68 -> 0.61 ms for the later stages).  Else 4 bytes a lane and step, which is less work for a line of a
    // few hundred bytes (c3c: 1.37 against 1.40 ms).
    auto scan_pass = [&]() __attribute__((always_inline)) {
    ncomma = 0; refusal = false; keyword = false; any_blank = 0; any_bad = 0; any_high = 0;
    if constexpr (WIDE) {
    auto nibble = [](uint32_t m4) { return ((m4 & B4) * 0x10204080u) >> 28; };   // bit 0 of each of four bytes -> bits 0 .. 3
    for (uint32_t base = a; base < b; base += 1024) {
        const uint32_t p0 = base + 16 * lane;
        uint32_t cw[4], inm[4];
        TP_UNROLL for (int q = 0; q < 4; ++q) {
            const uint32_t p = p0 + 4 * q, pp = p < b ? p : b;
            inm[q] = keep4(p, b) & B4;
            cw[q] = class4(cls, ld32u(s + pp));
        }
        const uint32_t pp0 = p0 < b ? p0 : b, pp3 = p0 + 12 < b ? p0 + 12 : b;
        const uint32_t before = p0 == a ? CL_COMMA : (uint32_t)cls[s[pp0 - 1]], after = (uint32_t)cls[s[pp3 + 4]];
        uint32_t commas16 = 0, blank16 = 0, high16 = 0, sp16 = 0, bad = 0, colon16 = 0, so16 = 0;
        TP_UNROLL for (int q = 0; q < 4; ++q) {
            const uint32_t c = cw[q], in = inm[q];
            const uint32_t pcw = (c << 8) | (q == 0 ? before : cw[q - 1] >> 24);    // the classes of the bytes before them
            const uint32_t ncw = (c >> 8) | ((q == 3 ? after : cw[q + 1] & 0xFFu) << 24);   // ... and after them
            // A byte the normaliser would change or drop: upper case, quotes, trailing-punctuation characters, white space
            // other than a single byte right after a comma, the second of two underscores.
            const uint32_t blank = c & ~(pcw >> 1) & in;
            const uint32_t high = (c >> 2) & (c >> 3) & in;            // bytes of characters outside ASCII: the token must be a known string as it stands
            const uint32_t sp = ((((c >> 2) & ~high) | ((c >> 3) & ~(c >> 2) & (pcw >> 3) & ~(pcw >> 2))) & in) | blank;
            commas16 |= nibble((c >> 1) & in) << (4 * q);
            blank16 |= nibble(blank) << (4 * q);
            high16 |= nibble(high) << (4 * q);
            sp16 |= nibble(sp) << (4 * q);
            bad |= (c >> 5) & in;
            colon16 |= nibble((c >> 4) & in) << (4 * q);
            so16 |= nibble((c >> 6) & (ncw >> 7) & in) << (4 * q);   // DropTagsContains (Program.cs:56-58): "sorry" is the one entry without a blank inside; look closer after "so"
        }
        const uint32_t cnt = popc64(commas16);
        const uint32_t incl = w.scan_incl(cnt);
        const uint32_t k0 = ncomma + incl - cnt;                   // commas before the lane's first byte = index of the token that byte belongs to
        auto token_of = [&](uint32_t i) { return k0 + popc64(commas16 & ((1u << i) - 1u)); };
        for (uint32_t m = commas16; w.ballot(m != 0);) {
            if (m) { const uint32_t i = ctz32(m), k = token_of(i); if (k < LDS::TOK_CAP) L.cpos[k] = (uint16_t)(p0 + i); m &= m - 1; }
        }
        ncomma += w.bcast(incl, 63);
        any_blank |= w.ballot(blank16 != 0);
        any_bad |= w.ballot(bad != 0);
        any_high |= w.ballot(high16 != 0);
        for (uint32_t m = sp16 | high16; w.ballot(m != 0);) {
            if (m) {
                const uint32_t i = ctz32(m), k = token_of(i);
                if (k < LDS::TOK_CAP) {
                    if ((sp16 >> i) & 1u) w.lds_or(&L.tspec[k >> 5], 1u << (k & 31));
                    if ((high16 >> i) & 1u) w.lds_or(&L.thigh[k >> 5], 1u << (k & 31));
                }
                m &= m - 1;
            }
        }
        if (w.ballot(colon16 != 0)) {   // only "positive:" / "negative:" matter (Program.cs:57, 592-594); other colons are ordinary tag text
            bool key = false;
            for (uint32_t m = colon16; m; m &= m - 1) {
                const uint32_t pj = p0 + ctz32(m);
                if (pj >= 8 && (lds_match(s, pj - 8, len, "positive", 8) || lds_match(s, pj - 8, len, "negative", 8))) key = true;
            }
            if (w.ballot(key)) keyword = true;
        }
        if (w.ballot(so16 != 0)) {
            bool hit = false;
            for (uint32_t m = so16; m; m &= m - 1)
                if (lds_match(s, p0 + ctz32(m), b, "sorry", 5)) hit = true;
            if (w.ballot(hit)) refusal = true;
        }
    }
    } else {
    for (uint32_t base = a; base < b; base += 256) {
        const uint32_t p = base + 4 * lane;
        const uint32_t pp = p < b ? p : b;
        const uint32_t inm = keep4(p, b) & B4;
        const uint32_t x = ld32u(s + pp);
        const uint32_t cw = class4(cls, x);
        uint32_t pcw = (cw << 8) | (p == a ? CL_COMMA : (uint32_t)cls[s[pp - 1]]);   // the classes of the bytes before them
        const uint32_t ncw = (cw >> 8) | ((uint32_t)cls[s[pp + 4]] << 24);            // ... and after them
        const uint32_t cm = (cw >> 1) & inm;                        // commas
        const uint64_t mc[4] = {w.ballot(cm & 1u), w.ballot(cm & 0x100u), w.ballot(cm & 0x10000u), w.ballot(cm & 0x1000000u)};
        uint32_t k[4];                                             // commas before byte j = index of the token it belongs to
        k[0] = ncomma + w.rank(mc[0]) + w.rank(mc[1]) + w.rank(mc[2]) + w.rank(mc[3]);
        k[1] = k[0] + (cm & 1u);
        k[2] = k[1] + ((cm >> 8) & 1u);
        k[3] = k[2] + ((cm >> 16) & 1u);
        if (cm) {
            for (uint32_t j = 0; j < 4; ++j)
                if ((cm & (1u << (8 * j))) && k[j] < LDS::TOK_CAP) L.cpos[k[j]] = (uint16_t)(p + j);
        }
        ncomma += popc64(mc[0]) + popc64(mc[1]) + popc64(mc[2]) + popc64(mc[3]);
        // A byte the normaliser would change or drop: upper case, quotes, trailing-punctuation characters, white space
        // other than a single byte right after a comma, the second of two underscores.
        const uint32_t blank = cw & ~(pcw >> 1) & inm;
        const uint32_t high = (cw >> 2) & (cw >> 3) & inm;           // bytes of characters outside ASCII: the token must be a known string as it stands
        const uint32_t sp = ((((cw >> 2) & ~high) | ((cw >> 3) & ~(cw >> 2) & (pcw >> 3) & ~(pcw >> 2))) & inm) | blank;
        any_blank |= w.ballot(blank != 0);
        any_bad |= w.ballot(((cw >> 5) & inm) != 0);
        any_high |= w.ballot(high != 0);
        if (sp | high) {
            for (uint32_t j = 0; j < 4; ++j) {
                if ((sp & (1u << (8 * j))) && k[j] < LDS::TOK_CAP) w.lds_or(&L.tspec[k[j] >> 5], 1u << (k[j] & 31));
                if ((high & (1u << (8 * j))) && k[j] < LDS::TOK_CAP) w.lds_or(&L.thigh[k[j] >> 5], 1u << (k[j] & 31));
            }
        }
        const uint32_t colon = (cw >> 4) & inm;
        if (w.ballot(colon != 0)) {   // only "positive:" / "negative:" matter (Program.cs:57, 592-594); other colons are ordinary tag text
            bool key = false;
            for (uint32_t j = 0; j < 4; ++j)
                if ((colon & (1u << (8 * j))) && p + j >= 8 && (lds_match(s, p + j - 8, len, "positive", 8) || lds_match(s, p + j - 8, len, "negative", 8))) key = true;
            if (w.ballot(key)) keyword = true;
        }
        // DropTagsContains (Program.cs:56-58): "sorry" is the one entry without a blank inside; look closer after "so"
        const uint32_t so = (cw >> 6) & (ncw >> 7) & inm;
        if (w.ballot(so != 0)) {
            bool hit = false;
            for (uint32_t j = 0; j < 4; ++j)
                if ((so & (1u << (8 * j))) && lds_match(s, p + j, b, "sorry", 5)) hit = true;
            if (w.ballot(hit)) refusal = true;
        }
    }
    }
    };
    scan_pass();
    if constexpr (!CRLF) { if (any_bad) return false; }
    else if (any_bad) {
    // "\r\n" inside the text: the reference starts with raw.Replace("\r\n", "\n") (Program.cs:591).  Those carriage returns are
    // taken out of the LDS copy in place and the scan runs once more; a carriage return of its own, or a '`', ends it here.
    {
        uint8_t* sw = const_cast<uint8_t*>(s);
        uint32_t written = a;
        bool other = false;
        for (uint32_t base = a; base < b; base += 256) {
            const uint32_t p = base + 4 * lane;
            const uint32_t pp = p < b ? p : b;
            const uint32_t x = ld32u(s + pp), nx = s[pp + 4];
            uint32_t keep = 0, cnt = 0;
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t c = (x >> (8 * j)) & 0xFFu, c1 = j < 3 ? (x >> (8 * j + 8)) & 0xFFu : nx;
                const bool in = p + j < b;
                const bool gone = in && c == '\r' && c1 == '\n' && p + j + 1 < b;
                if (in && !gone && (cls[c] & CL_BAD)) other = true;
                if (in && !gone) { keep |= 1u << j; ++cnt; }
            }
            const uint32_t incl = w.scan_incl(cnt);
            w.sync();                                              // every lane holds its bytes before any is overwritten
            uint32_t d = written + incl - cnt;
            for (uint32_t j = 0; j < 4; ++j) if (keep & (1u << j)) sw[d++] = (uint8_t)(x >> (8 * j));
            written += w.bcast(incl, 63);
            w.sync();
        }
        if (w.ballot(other)) return false;
        b = written;
        if (lane < LDS::TOK_CAP / 32) { L.tspec[lane] = 0; L.thigh[lane] = 0; }
        w.sync();
    }
    scan_pass();
    if (any_bad) return false;
    }
    if (any_high && ncomma < 2) return false;   // LooksBad measures such a line in UTF-16 units (Program.cs:1145)
    if (any_blank && !refusal) {
        // The other entries (without the two that hold ':') all have a blank that follows a letter; such a blank was seen.
        for (uint32_t base = a; base < b; base += 64) {
            const uint32_t p = base + lane;
            const bool inb = p < b;
            uint32_t q = lower4(ld32u(s + (inb ? p : b)));
            const uint32_t nl = eq4(q, rep4('\n'));                  // a line feed counts as the blank it becomes (Program.cs:595): 0x0A ^ 0x2A = 0x20
            q ^= (nl >> 2) | (nl >> 4) | (nl >> 6);
            const bool pre = inb & ((q == c4('i', ' ', 'c', 'a')) | (q == c4('i', '\'', 'm', ' ')) | (q == c4('i', ' ', 'w', 'o')) | (q == c4('c', 'a', 'n', 'n')) |
                                    (q == c4('u', 'n', 'a', 'b')) | (q == c4('a', 's', ' ', 'a')));
            if (w.ballot(pre)) {
                bool hit = false;
                if (pre) {
                    switch (q & 0xFFu) {
                        case 'i': hit = lds_match(s, p, b, "i can't", 7) || lds_match(s, p, b, "i cannot", 8) || lds_match(s, p, b, "i'm not able", 12) || lds_match(s, p, b, "i won't", 7); break;
                        case 'c': hit = lds_match(s, p, b, "cannot provide", 14); break;
                        case 'u': hit = lds_match(s, p, b, "unable to", 9); break;
                        case 'a': hit = lds_match(s, p, b, "as an ai", 8); break;
                        default: break;
                    }
                }
                if (w.ballot(hit)) refusal = true;
            }
        }
    }
    if (keyword) return false;
    sc.s = s; sc.a = a; sc.b = b; sc.ncomma = ncomma;
    sc.rejected = blank_line || refusal || (ncomma < 2 && (b - a) > 40);
    w.sync();
    return true;
}


// The back half of the pipeline for one file, on the selected tags in L.fin[0 .. nr) (with their flags and lengths in
// L.fattr / L.flen when `settled`): coverage, final-set closure, action rules, required facets, verdict, caption tag ids.
// kid: the per-position columns of the selected tags' implied tags (settled files only).
// evidence(n_evidence, ev_feet, ev_penis): the caller's evidence step (it owns the candidates); false = decline the file.
struct TailIn {
    uint32_t nr, n, n_selected, n_dropped;
    bool settled, rejected;
    uint32_t (*kid)[64];
};
template <int NF, bool AUD = false, class W, class Ev>
TP_HD __attribute__((always_inline)) bool fast_tail(W& w, const DevTables& t, BackLds& L, const uint32_t* rc, const TailIn& in_, Ev& evidence, FastOut& out, uint32_t* fin32,
                                                   uint32_t* fired, uint16_t* facet_choice, const LeanAudit aud = LeanAudit{nullptr, nullptr, nullptr, nullptr}) {
    constexpr uint32_t FIN_MAX = 64u * NF;
    const uint32_t lane = w.lane();
    const uint32_t nr = in_.nr, n = in_.n, n_selected = in_.n_selected, n_dropped = in_.n_dropped;
    bool settled = in_.settled;
    const bool rejected = in_.rejected;
    uint32_t(*kid)[64] = in_.kid;
    uint32_t(*kid_attr)[64] = kid + 2;
    uint32_t(*kid_len)[64] = kid + 4;

    w.tick(5);
    TP_STOP(5, L.fin[lane] ^ nr ^ n_selected)
    // ---- S7: EvaluateCoverage (Program.cs:1056-1095) and normalizedFinalSet (Program.cs:303) -------------------------------
    uint32_t own0 = NO_ID, own1 = NO_ID;   // the final list, entry k in lane k & 63
    uint32_t nf = 0;
    uint32_t coverage = 0;
    {
        const bool in = lane < nr;
        const uint32_t r = in ? L.fin[lane] : EMPTY_ID;
        uint32_t t4 = r, f4 = in ? L.fattr[lane] : (uint32_t)FL_BLANK;
        uint64_t m_hop4 = 0;
        if (!settled) {
            Rec rr = t.rec[r];
            t4 = rr.next;
            m_hop4 = w.ballot(in && t4 != r);
            f4 = rr.flags;
            if (m_hop4 && in && t4 != r) f4 = t.rec[t4].flags;
        }
        const bool nb4 = in && !(f4 & FL_BLANK);
        const bool has_trigger = w.ballot(nb4 && t4 == t.trigger_id) != 0, has_girl = w.ballot(nb4 && t4 == t.girl_lit_id) != 0;
        coverage = ((has_trigger && has_girl) ? TP_COV_IDENTITY : 0) | (w.ballot(nb4 && (f4 & FL_HAIR)) ? TP_COV_HAIR : 0) |
                   (w.ballot(nb4 && (f4 & FL_EYES)) ? TP_COV_EYES : 0) | (w.ballot(nb4 && fl_bucket(f4) == B_OUTFIT) ? TP_COV_OUTFIT : 0) |
                   (w.ballot(nb4 && fl_bucket(f4) == B_POSE) ? TP_COV_POSE_ACTION : 0) | (w.ballot(nb4 && fl_bucket(f4) == B_BACKGROUND) ? TP_COV_BACKGROUND : 0);
        if (nr >= 3) {
            bool keep4 = nb4;
            if (m_hop4) {
                uint32_t key[2] = {t4, NO_ID};
                bool valid[2] = {nb4, false};
                bool fst[2];
                if (!dedupe_first(w, L, key, valid, fst, 1)) return false;
                keep4 = fst[0];
            }
            uint64_t m = w.ballot(keep4);
            nf = popc64(m);
            w.sync();
            if (keep4) L.fin[w.rank(m)] = t4;
        } else {                                                   // fewer than 3 tags: the minimal fallback selection (Program.cs:297-298)
            nf = t.n_fb_final;
            w.sync();
            if (lane < nf) L.fin[lane] = t.fb_final[lane];
        }
        w.sync();
        own0 = lane < nf ? L.fin[lane] : NO_ID;
    }
    uint32_t flags = (rejected ? TP_F_LINE_REJECTED : 0u) | (nr < 3 ? TP_F_FALLBACK : 0u);

    w.tick(6);
    TP_STOP(6, own0 ^ nf ^ coverage ^ flags)
    uint32_t n_evidence = 0;
    bool ev_feet = false, ev_penis = false;
    if (!evidence(n_evidence, ev_feet, ev_penis)) return false;

    w.tick(7);
    TP_STOP(7, own0 ^ nf ^ n_evidence ^ (uint32_t)ev_feet ^ (uint32_t)ev_penis ^ L.ebits[lane])
    // ---- S9: final-set closure, in the reference's insertion order (ExpandImpliedTags, Program.cs:304, 1160-1179) ------------
    // The set grows breadth-first: a round visits every element in order and appends NormalizeTag(it) and its implied tags
    // (its "steps").  A round whose elements have at most two steps each runs in parallel: every step is offered to the LDS
    // set under the key (element index, step index); the smallest key of an id new to the set wins, and a winner's place
    // in the list is the number of winners with smaller keys.  From here to S11 the set holds the list's elements.
    uint32_t gnext[2] = {EMPTY_ID, EMPTY_ID}, gflags[2] = {0, 0};   // NormalizeTag and flags of the entries this lane owns, read once here
    if (settled) {
        // One round settles it: the implied tags of the selected tags (read from the per-position columns written in S6) are
        // offered to the set under the key (position, first or second); winners are appended in key order with their flags
        // and lengths; whatever an appended tag's own visit would append is already among the tags offered here.
        set_clear(w, L);
        if (lane < nf) { bool created; set_put(w, L, own0, 0, created); }
        w.sync();
        uint32_t k_id[2] = {NO_ID, NO_ID}, slot[2] = {NO_ID, NO_ID};
        bool full = false;
        if (lane < nf) { k_id[0] = kid[0][lane]; k_id[1] = kid[1][lane]; }
        for (uint32_t j = 0; j < 2; ++j)
            if (k_id[j] != NO_ID) { bool created; slot[j] = set_put(w, L, k_id[j], 1 + ((lane << 1) | j), created); if (slot[j] == NO_ID) full = true; }
        w.sync();
        if (w.ballot(full)) return false;
        bool win[2];
        for (uint32_t j = 0; j < 2; ++j) win[j] = slot[j] != NO_ID && L.set_min[slot[j]] == 1 + ((lane << 1) | j);
        const uint64_t mw0 = w.ballot(win[0]), mw1 = w.ballot(win[1]);
        const uint32_t grown = popc64(mw0) + popc64(mw1);
        if (nf + grown > FIN_MAX) return false;
        uint32_t p = nf + w.rank(mw0) + w.rank(mw1);
        if (win[0]) { L.fin[p] = k_id[0]; L.fattr[p] = kid_attr[0][lane]; L.flen[p] = kid_len[0][lane]; L.set_min[slot[0]] = 0; ++p; }
        if (win[1]) { L.fin[p] = k_id[1]; L.fattr[p] = kid_attr[1][lane]; L.flen[p] = kid_len[1][lane]; L.set_min[slot[1]] = 0; }
        nf += grown;
        w.sync();
        own0 = lane < nf ? L.fin[lane] : NO_ID;
        own1 = lane + 64 < nf ? L.fin[lane + 64] : NO_ID;
        gnext[0] = own0; gnext[1] = own1;                          // NormalizeTag leaves every element alone
    } else if (t.closure_tables && t.reach_levels_ok) {
        // Every element brings the list of its own closure in the order ExpandImpliedTags appends it, with the round of each
        // entry (DevTables::reach_ent, or the two steps of its record); a round's additions over all elements are ordered by
        // element, then by place in the element's list, and a string counts where it comes first.  So every entry goes into
        // the set under the key (round, element, place), and the lowest key of a string stays in its slot; afterwards the
        // slots that are not the elements' own hold exactly the strings to append, and the keys' order is their order:
        // one trip to the tables for all rounds together instead of one per round.
        set_clear(w, L);
        if (lane < nf) { bool created; set_put(w, L, own0, 0, created); }   // nf <= 64 here, and 256 slots cannot fill up
        const uint32_t nf_seeds = nf;
        Rec rs;
        rs.flags = 0; rs.next = EMPTY_ID; rs.step_cnt = 0; rs.step0 = rs.step1 = NO_ID;
        uint32_t e0 = 0, e1 = 0;
        if (lane < nf) { rs = t.rec[own0]; e0 = t.reach_off[own0]; e1 = t.reach_off[own0 + 1]; }
        w.sync();
        gnext[0] = rs.next; gflags[0] = rs.flags;
        const bool edge = lane < nf && (rs.flags & FL_EDGE) != 0;
        if (w.ballot(edge)) {
            const bool own_steps = (rs.flags & FL_REACH2) != 0;
            const uint32_t ne = !edge ? 0u : own_steps ? (rs.step_cnt < 2 ? rs.step_cnt : 2u) : e1 - e0;
            bool full = ne > 0xFFFFu;
            for (uint32_t k0 = 0; k0 < ne && !full; k0 += 4) {          // four entries' loads at a time
                uint32_t id[4], round[4];
                for (uint32_t j = 0; j < 4; ++j)
                    if (k0 + j < ne) {
                        if (own_steps) { id[j] = k0 + j ? rs.step1 : rs.step0; round[j] = 1; }
                        else { const uint64_t e = *(const uint64_t*)(t.reach_ent + 2 * (size_t)(e0 + k0 + j)); id[j] = (uint32_t)e; round[j] = (uint32_t)(e >> 58); }
                    }
                for (uint32_t j = 0; j < 4; ++j)
                    if (k0 + j < ne) { bool created; if (set_put(w, L, id[j], round[j] << 24 | lane << 16 | (k0 + j), created) == NO_ID) full = true; }
            }
            if (w.ballot(full)) return false;
            w.sync();
            // the new strings: four slots per lane; their keys and ids side by side in the (unused on this path) flag and
            // length columns, then every new string's place is the number of new strings with a lower key
            uint32_t wid[4], wkey[4], mine = 0;
            for (uint32_t j = 0; j < 4; ++j) {
                wid[j] = L.set_key[4 * lane + j]; wkey[j] = L.set_min[4 * lane + j];
                if (wid[j] == NO_ID || wkey[j] == 0) wkey[j] = 0xFFFFFFFFu; else ++mine;
            }
            const uint32_t incl = w.scan_incl(mine), grown = w.bcast(incl, 63);
            if (nf + grown > FIN_MAX) return false;
            uint32_t at = incl - mine;
            for (uint32_t j = 0; j < 4; ++j) if (wkey[j] != 0xFFFFFFFFu) L.fattr[at++] = wkey[j];
            w.sync();
            uint32_t rank[4] = {0, 0, 0, 0};
            for (uint32_t i = 0; i < grown; ++i) {
                const uint32_t k = L.fattr[i];
                for (uint32_t j = 0; j < 4; ++j) rank[j] += (uint32_t)(k < wkey[j]);
            }
            for (uint32_t j = 0; j < 4; ++j) if (wkey[j] != 0xFFFFFFFFu) { L.fin[nf + rank[j]] = wid[j]; L.set_min[4 * lane + j] = 0; }
            nf += grown;
            w.sync();
            own0 = lane < nf ? L.fin[lane] : NO_ID;
            own1 = lane + 64 < nf ? L.fin[lane + 64] : NO_ID;
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < NF) {          // NormalizeTag and flags of what the closure added
                const uint32_t idx = (uint32_t)blk * 64 + lane;
                if (idx >= nf_seeds && idx < nf) { const Rec ra = t.rec[blk == 0 ? own0 : own1]; gnext[blk] = ra.next; gflags[blk] = ra.flags; }
            }
        }
    } else {
        set_clear(w, L);
        if (lane < nf) { bool created; set_put(w, L, own0, 0, created); }   // nf <= 64 here, and 256 slots cannot fill up
        w.sync();
        uint32_t lo = 0, hi = nf;
        while (lo < hi) {
            const int nblk = NF == 1 ? 1 : (hi > 64 ? 2 : 1);
            uint32_t so[2] = {0, 0}, sc[2] = {0, 0}, sid[2][2] = {{NO_ID, NO_ID}, {NO_ID, NO_ID}};
            bool edge[2] = {false, false};
            uint64_t m_edge[2] = {0, 0}, m_long = 0;
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
                const uint32_t idx = (uint32_t)blk * 64 + lane;
                const bool in = idx >= lo && idx < hi;
                if (in) {
                    const Rec rr = t.rec[blk == 0 ? own0 : own1];
                    gnext[blk] = rr.next;
                    gflags[blk] = rr.flags;
                    edge[blk] = (rr.flags & FL_EDGE) != 0;
                    so[blk] = rr.step_off; sc[blk] = rr.step_cnt; sid[blk][0] = rr.step0; sid[blk][1] = rr.step1;
                }
                m_edge[blk] = w.ballot(edge[blk]);
                m_long |= w.ballot(edge[blk] && sc[blk] > 2);
            }
            const uint32_t nf0 = nf;
            if (m_long) {                                          // long step lists: one element at a time
                TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
                    uint64_t m = m_edge[blk];
                    while (m) {
                        const uint32_t q = ctz64(m);
                        m &= m - 1;
                        const uint32_t cnt = w.bcast(sc[blk], q), off = w.bcast(so[blk], q), a0 = w.bcast(sid[blk][0], q), a1 = w.bcast(sid[blk][1], q);
                        for (uint32_t k = 0; k < cnt; ++k) {
                            const uint32_t x = k == 0 ? a0 : (k == 1 ? a1 : t.step_ids[off + k]);
                            if (!fin_append(w, L, own0, own1, nf, x, true, FIN_MAX)) return false;
                        }
                    }
                }
                w.sync();
            } else if (m_edge[0] | m_edge[1]) {
                uint32_t slot[2][2] = {{NO_ID, NO_ID}, {NO_ID, NO_ID}};
                bool full = false;
                TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk)
                    for (uint32_t j = 0; j < 2; ++j)
                        if (edge[blk] && j < sc[blk]) {
                            bool created;
                            slot[blk][j] = set_put(w, L, sid[blk][j], 1 + ((((uint32_t)blk * 64 + lane) << 1) | j), created);
                            if (slot[blk][j] == NO_ID) full = true;
                        }
                w.sync();
                if (w.ballot(full)) return false;
                bool win[2][2] = {{false, false}, {false, false}};
                uint64_t mw[2][2] = {{0, 0}, {0, 0}};
                TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk)
                    for (uint32_t j = 0; j < 2; ++j) {
                        win[blk][j] = slot[blk][j] != NO_ID && L.set_min[slot[blk][j]] == 1 + ((((uint32_t)blk * 64 + lane) << 1) | j);
                        mw[blk][j] = w.ballot(win[blk][j]);
                    }
                const uint32_t c0 = popc64(mw[0][0]) + popc64(mw[0][1]), c1 = popc64(mw[1][0]) + popc64(mw[1][1]);
                if (nf + c0 + c1 > FIN_MAX) return false;
                TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
                    uint32_t pos = nf + (blk ? c0 : 0) + w.rank(mw[blk][0]) + w.rank(mw[blk][1]);
                    if (win[blk][0]) { L.fin[pos] = sid[blk][0]; L.set_min[slot[blk][0]] = 0; ++pos; }
                    if (win[blk][1]) { L.fin[pos] = sid[blk][1]; L.set_min[slot[blk][1]] = 0; }
                }
                nf += c0 + c1;
                w.sync();
                own0 = lane < nf ? L.fin[lane] : NO_ID;
                own1 = lane + 64 < nf ? L.fin[lane + 64] : NO_ID;
            }
            (void)nf0;
            lo = hi;
            hi = nf;
        }
        w.sync();
    }

    w.tick(8);
    TP_STOP(8, own0 ^ own1 ^ nf ^ gnext[0] ^ gflags[0] ^ gnext[1] ^ gflags[1])
    // ---- S10: InferActionTags (Program.cs:731-753) ---------------------------------------------------------------------------
    bool rules_possible = true, frules_possible = true;
    {
        // inferred = Select(NormalizeTag).Where(non-blank).Distinct()
        uint32_t g[2] = {EMPTY_ID, EMPTY_ID};
        bool nb[2] = {false, false};
        bool changed = false;
        if (!settled)
            for (int blk = 0; blk < NF; ++blk) {
                uint32_t idx = (uint32_t)blk * 64 + lane;
                uint32_t id = blk == 0 ? own0 : own1;
                bool in = idx < nf;
                g[blk] = gnext[blk];
                nb[blk] = in && !(gflags[blk] & FL_NEXT_BLANK);
                if (in && (g[blk] != id || !nb[blk])) changed = true;
            }
        if (!settled && w.ballot(changed)) {
            bool fst[2];
            if (!dedupe_first(w, L, g, nb, fst, NF)) return false;
            uint64_t m0 = w.ballot(fst[0]), m1 = NF > 1 ? w.ballot(fst[1]) : 0;
            nf = popc64(m0) + popc64(m1);
            if (fst[0]) L.fin[w.rank(m0)] = g[0];
            if (fst[1]) L.fin[popc64(m0) + w.rank(m1)] = g[1];
            w.sync();
            own0 = lane < nf ? L.fin[lane] : NO_ID;
            own1 = lane + 64 < nf ? L.fin[lane + 64] : NO_ID;
            set_clear(w, L);                                       // the set follows the rewritten list
            bool created;
            if (lane < nf) set_put(w, L, own0, 0, created);
            if (lane + 64 < nf) set_put(w, L, own1, 0, created);
            w.sync();
        }
        if (lane < t.fired_words) L.firedw[lane] = 0;
        w.sync();
        // No rule can fire unless the evidence holds one of the literals some rule tests positively; likewise for facet rules.
        bool lit_r = false, lit_f = false;
        for (uint32_t k = lane; k < t.lit_words; k += 64) {
            const uint32_t e = L.ebits[k];
            lit_r = lit_r || (e & rc[t.rc_rule_mask + k]) != 0;
            lit_f = lit_f || (e & rc[t.rc_frule_mask + k]) != 0;
        }
        rules_possible = t.rules_vacuous || w.ballot(lit_r) != 0;
        frules_possible = t.frules_vacuous || w.ballot(lit_f) != 0;
        if (rules_possible)
        for (uint32_t r0 = 0; r0 < t.n_rules; r0 += 64) {
            const uint32_t r = r0 + lane;
            const bool hit = r < t.n_rules && rc_match(L, rc, r * RC_REC_WORDS);                  // MatchesActionRule, Program.cs:777-782
            uint64_t m = w.ballot(hit);
            if (hit) w.lds_or(&L.firedw[r >> 5], 1u << (r & 31));      // "action:{parent}" (Program.cs:741)
            while (m) {
                const uint32_t* rr = rc + (r0 + ctz64(m)) * RC_REC_WORDS;
                m &= m - 1;
                const uint32_t parent = rr[0], sub_off = rr[1], sub_cnt = rr[3] >> 16;
                if (rr[3] & 0x8000u) settled = false;              // a tag NormalizeTag changes joins the list: S12 looks every tag up again
                if (!fin_append(w, L, own0, own1, nf, parent, true, FIN_MAX, rr[8], rr[9])) return false;
                for (uint32_t s0 = 0; s0 < sub_cnt; s0 += 64) {
                    const uint32_t si = s0 + lane;
                    const uint32_t srec = t.rc_subs + (sub_off + (si < sub_cnt ? si : 0)) * RC_REC_WORDS;
                    const bool shit = si < sub_cnt && rc_match(L, rc, srec);
                    uint64_t ms = w.ballot(shit);
                    if (shit) {                                    // "subtype:{parent}:{tag}" (Program.cs:748)
                        uint32_t bit = t.n_rules + sub_off + si;
                        w.lds_or(&L.firedw[bit >> 5], 1u << (bit & 31));
                    }
                    while (ms) {
                        const uint32_t qs = ctz64(ms);
                        ms &= ms - 1;
                        const uint32_t* sr = rc + t.rc_subs + (sub_off + s0 + qs) * RC_REC_WORDS;
                        if (sr[3] & 0x8000u) settled = false;
                        if (!fin_append(w, L, own0, own1, nf, sr[0], true, FIN_MAX, sr[8], sr[9])) return false;
                    }
                }
            }
        }
    }
    w.tick(9);
    TP_STOP(9, own0 ^ own1 ^ nf ^ L.firedw[lane])
    // ---- S11: EnsureRequiredFacets (Program.cs:755-775) -----------------------------------------------------------------------
    // every facet rule on its own lane first, then every facet picks its first matching rule
    for (uint32_t p0 = 0; p0 < t.n_frules; p0 += 64) {
        const uint32_t p = p0 + lane;
        const uint64_t m = frules_possible ? w.ballot(p < t.n_frules && rc_match(L, rc, t.rc_frules + p * RC_REC_WORDS)) : 0;
        if (lane == 0) { L.fhit[p0 >> 5] = (uint32_t)m; L.fhit[(p0 >> 5) + 1] = (uint32_t)(m >> 32); }
    }
    w.sync();
    for (uint32_t f0 = 0; f0 < t.n_facets; f0 += 64) {
        const uint32_t f = f0 + lane;
        uint32_t selected = NO_ID, sel_attr = 0, sel_len = 0;
        bool sel_changes = false;                                  // NormalizeTag does not leave the chosen tag alone
        if (f < t.n_facets) {
            const uint32_t* fr = rc + t.rc_facets + RC_FACET_WORDS * f;
            const uint32_t rule_off = fr[1], rule_cnt = fr[2];
            selected = fr[0]; sel_attr = fr[3]; sel_len = fr[4]; sel_changes = fr[5] != 0;
            uint16_t choice = 0xFFFF;
            for (uint32_t r = 0; r < rule_cnt; ++r) {
                const uint32_t p = rule_off + r;
                if ((L.fhit[p >> 5] >> (p & 31)) & 1u) {
                    const uint32_t* pr = rc + t.rc_frules + p * RC_REC_WORDS;
                    selected = pr[0]; sel_attr = pr[8]; sel_len = pr[9]; sel_changes = (pr[3] & 0x8000u) != 0;
                    choice = (uint16_t)r;
                    break;
                }
            }
            if (facet_choice) facet_choice[f] = choice;
        }
        // append the chosen tags in facet order, skipping those already in the list: the first facet to offer a new id wins
        w.sync();
        uint32_t slot = NO_ID;
        bool full = false;
        if (f < t.n_facets) { bool created; slot = set_put(w, L, selected, 1 + lane, created); full = slot == NO_ID; }
        w.sync();
        if (w.ballot(full)) return false;
        const bool win = slot != NO_ID && L.set_min[slot] == 1 + lane;
        const uint64_t mw = w.ballot(win);
        if (w.ballot(win && sel_changes)) settled = false;
        if (nf + popc64(mw) > FIN_MAX) return false;
        if (win) { const uint32_t p = nf + w.rank(mw); L.fin[p] = selected; L.fattr[p] = sel_attr; L.flen[p] = sel_len; L.set_min[slot] = 0; }
        nf += popc64(mw);
        w.sync();
        own0 = lane < nf ? L.fin[lane] : NO_ID;
        own1 = lane + 64 < nf ? L.fin[lane + 64] : NO_ID;
    }
    w.sync();

    w.tick(10);
    TP_STOP(10, own0 ^ own1 ^ nf)
    // ---- S12: verdicts on the final list (Program.cs:319, 329), GetUnknownTags (:817-824), FormatCaptionSingleLine (:971-974) ---
    uint32_t c[2] = {EMPTY_ID, EMPTY_ID}, clen[2] = {0, 0}, idlen[2] = {0, 0};
    bool cnb[2] = {false, false}, cknown[2] = {false, false};
    bool face = false, outfit = false, changed = false;
    uint32_t safe_idx[2] = {0, 0};
    bool safe[2] = {false, false};
    for (int blk = 0; blk < NF; ++blk) {
        uint32_t idx = (uint32_t)blk * 64 + lane;
        uint32_t id = blk == 0 ? own0 : own1;
        bool in = idx < nf;
        uint32_t fl, cfl;                                          // flags of the tag and of what NormalizeTag makes of it
        if (settled) {                                             // ... which is the tag itself; flags and length sit in the list's columns
            fl = cfl = in ? L.fattr[idx] : (uint32_t)FL_BLANK;
            c[blk] = in ? id : EMPTY_ID;
            clen[blk] = in ? L.flen[idx] : 0;
            idlen[blk] = clen[blk];
        } else {
            Rec rr = t.rec[in ? id : EMPTY_ID];
            fl = rr.flags;
            idlen[blk] = in ? rr.len : 0;
            c[blk] = rr.next;
            Rec rn = rr;
            if (in && c[blk] != id) { rn = t.rec[c[blk]]; changed = true; }
            cfl = rn.flags;
            clen[blk] = rn.len;
        }
        if (in) {
            uint32_t bk = fl_bucket(fl);
            face |= bk == B_FACE;                                  // MissingCoreBuckets, Program.cs:1097-1101
            outfit |= bk == B_OUTFIT;
            if (fl & FL_SAFE) { safe[blk] = true; safe_idx[blk] = (fl >> FL_SAFE_IDX_SHIFT) & 31; }
        }
        cnb[blk] = in && !(cfl & FL_BLANK);
        if (in && !cnb[blk]) changed = true;
        cknown[blk] = (cfl & FL_KNOWN) != 0;
    }
    if (!(w.ballot(face) && w.ballot(outfit))) flags |= TP_F_LOW;
    int safe_hit = -1;
    {
        uint64_t ms0 = w.ballot(safe[0]), ms1 = NF > 1 ? w.ballot(safe[1]) : 0;
        if (t.safe_mode && (ms0 | ms1)) {                          // first final tag in SafeModeRejectTags (Program.cs:942-945)
            uint32_t v = ms0 ? w.bcast(safe_idx[0], ctz64(ms0)) : w.bcast(safe_idx[1], ctz64(ms1));
            safe_hit = (int)v;
            flags |= TP_F_SAFE_REJECT;
        }
    }
    bool keepc[2] = {cnb[0], cnb[1]};
    if (w.ballot(changed)) {
        if (!dedupe_first(w, L, c, cnb, keepc, NF)) return false;
    }
    const uint64_t mc0 = w.ballot(keepc[0]), mc1 = NF > 1 ? w.ballot(keepc[1]) : 0;
    const uint32_t nc = popc64(mc0) + popc64(mc1);
    uint32_t bytes = (keepc[0] ? clen[0] : 0) + (keepc[1] ? clen[1] : 0);
    bytes = w.sum(bytes);
    const uint32_t unknown = t.known_active ? popc64(w.ballot(keepc[0] && !cknown[0])) + (NF > 1 ? popc64(w.ballot(keepc[1] && !cknown[1])) : 0) : 0;
    const bool footjob = w.ballot((keepc[0] && c[0] == t.footjob_n) || (keepc[1] && c[1] == t.footjob_n)) != 0;   // HasTag(finalTags, "footjob"), Program.cs:311
    w.sync();
    if (fired && lane < t.fired_words) fired[lane] = L.firedw[lane];
    if (keepc[0]) fin32[w.rank(mc0)] = c[0];
    if (keepc[1]) fin32[popc64(mc0) + w.rank(mc1)] = c[1];
    if (ev_feet && ev_penis) { flags |= TP_F_ELIGIBLE; if (footjob) flags |= TP_F_FOOTJOB; }
    if constexpr (AUD) {
        // keptTags = the final list as it stands (Program.cs:334); the caption's list (fin32) differs from it when NormalizeTag changes
        // a tag or makes one blank (Program.cs:972); GetUnknownTags walks the caption's list (Program.cs:817-821)
        const bool differs = w.ballot(changed) != 0;
        const uint32_t kbytes = w.sum(idlen[0] + idlen[1]);
        const uint32_t ubytes = t.known_active ? w.sum((keepc[0] && !cknown[0] ? clen[0] : 0u) + (keepc[1] && !cknown[1] ? clen[1] : 0u)) : 0u;
        if (differs) {
            if (lane < nf) aud.row[AUD_KEPT + lane] = own0;
            if (lane + 64 < nf) aud.row[AUD_KEPT + lane + 64] = own1;
        }
        if (lane == 0) {
            aud.row[AUD_FLAGS] = differs ? 1u : 0u;
            aud.cnt[AUDIT_KEPT] = nf; aud.len[AUDIT_KEPT] = kbytes + (nf ? nf - 1 : 0u);
            aud.cnt[AUDIT_UNKNOWN] = nc; aud.len[AUDIT_UNKNOWN] = ubytes + (unknown ? unknown - 1 : 0u);
        }
    }

    w.tick(11);
    tp_verdict& v = out.verdict;
    v.coverage = (uint8_t)coverage;
    v.flags = (uint8_t)flags;
    v.n_tags = (uint16_t)nc;
    v.n_candidates = (uint16_t)n;
    v.n_selected = (uint16_t)n_selected;
    v.safe_hit = (int16_t)safe_hit;
    v.n_evidence = (uint16_t)n_evidence;
    v.n_unknown = (uint16_t)unknown;
    v.n_dropped = (uint16_t)n_dropped;
    out.n_final = nc;
    out.caption_len = bytes + (nc ? 2 * (nc - 1) : 0) + t.newline_len;
    return out.caption_len <= FAST_CAPTION_MAX;
}

// The pipeline for one file.  Returns false when the file needs the general kernel.
// NT = 1: at most 64 tokens (so candidates), one per lane; NT = 2: 128, two per lane.  NF likewise for the final tags.
// Fewer per lane means fewer registers and less work per wavefront.
// rc: the rule cache (DevTables::rc, or a copy of it in faster memory); cls: byte_class() of 0..255 as a table.
// fin32: where the caption's tag ids go (FAST_FIN_MAX entries); fired / facet_choice may be null.
// Front: LineFront (the candidates come from the line `text`, as above) or a type with `scored == true` whose run() puts
// scored candidates on the lanes in the order SelectTags sorts them into (ScoredFront below); text / len / cls / triage
// are not used then.
struct LineFront { static constexpr bool scored = false; };
template <int NT, int NF, class W, class Front = LineFront, bool AUD = false>
TP_HD bool fast_file(W& w, const DevTables& t, FastLds& L, const uint32_t* rc, const uint8_t* cls, const uint8_t* text, uint32_t len, FastOut& out, uint32_t* fin32,
                     uint32_t* fired, uint16_t* facet_choice, bool triage = false, Front* front = nullptr, const LeanAudit aud = LeanAudit{nullptr, nullptr, nullptr, nullptr}) {
    static_assert((NT == 1 || NT == 2) && (NF == 1 || NF == 2), "one or two elements per lane");
    const uint32_t lane = w.lane();
    int nblk = 1;
    uint32_t n = 0;
    bool act[2] = {false, false};
    uint32_t h1[2] = {EMPTY_ID, EMPTY_ID}, ci[2] = {0, 0};
    Rec r1[2];
    bool rejected = false;
    bool seed_in[2] = {false, false}, strong_in[2] = {false, false};   // scored candidates: !(score < 0.28), score >= 0.60
    if constexpr (Front::scored) {
        r1[0] = r1[1] = Rec{FL_BLANK, EMPTY_ID, NO_ID, 0, 0, 0, NO_ID, NO_ID};
        if (!front->run(w, t, L, nblk, n, act, h1, r1, seed_in, strong_in)) return false;
        ci[0] = lane; ci[1] = 64 + lane;                           // the candidates sit on the lanes in their order
        if (AUD && front->audit_unfit) return false;              // lists this stage cannot write as interned ids in the reference's order
    } else {
    Scan sc;
    if (!scan_line<NT == 2, NT == 2>(w, L, cls, text, len, sc)) return false;
    const uint8_t* s = sc.s;
    const uint32_t a = sc.a, b = sc.b, ncomma = sc.ncomma;
    rejected = sc.rejected;
    const uint32_t T = ncomma + 1;
    if (T > 64u * NT) return false;
    w.tick(2);
    TP_STOP(2, a ^ b ^ ncomma ^ T ^ (uint32_t)rejected ^ L.cpos[lane] ^ L.tspec[lane & 3])

    // ---- S4: tokens -> interned ids; Length >= 2, !ShouldDropTag, Distinct (Program.cs:598-602) ---------------------------
    // A token's lane stays its candidate's lane: candidate index = number of kept tokens before it.
    nblk = NT == 1 ? 1 : (T > 64 ? 2 : 1);                        // tokens (and so candidates) sit two per lane only in long lines
    r1[0] = r1[1] = Rec{FL_BLANK, EMPTY_ID, NO_ID, 0, 0, 0, NO_ID, NO_ID};
    {
        bool keep[2] = {false, false};
        bool dyn_survivor = false;
        if (!rejected) {
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
                uint32_t tix = (uint32_t)blk * 64 + lane;
                if (tix < T) {
                    uint32_t i = tix == 0 ? a : (uint32_t)L.cpos[tix - 1] + 1;
                    uint32_t j = tix == T - 1 ? b : (uint32_t)L.cpos[tix];
                    uint32_t id = NO_ID, dlen = 0;
                    bool vdyn = false;
                    const bool special = (L.tspec[tix >> 5] >> (tix & 31)) & 1u, high = (L.thigh[tix >> 5] >> (tix & 31)) & 1u;
                    Rec rr;
                    int kind = fast_token(t, s, i, j, special, high, id, rr, vdyn, dlen);
                    if (kind == 1) {
                        if ((rr.flags & FL_LEN2) && !(rr.flags & FL_DROP)) { h1[blk] = id; r1[blk] = rr; keep[blk] = true; }
                        if (rr.flags & FL_SPLIT) dyn_survivor = true;   // the join + re-split cuts it up: general kernel
                    } else if (kind == 3) {
                        dyn_survivor = true;                       // declines the file below
                    } else if (kind == 2) {
                        // a string outside the tables: gone under strict vocabulary, too short, or invalid; anything else may survive
                        if (!t.drop_unknown && dlen >= 2 && vdyn) dyn_survivor = true;
                    }
                }
            }
        }
        if (w.ballot(dyn_survivor)) return false;
        if (!dedupe_first(w, L, h1, keep, act, nblk)) return false;
        const uint64_t m0 = w.ballot(act[0]), m1 = nblk > 1 ? w.ballot(act[1]) : 0;
        n = popc64(m0) + popc64(m1);
        ci[0] = w.rank(m0);
        ci[1] = popc64(m0) + w.rank(m1);
        if (!act[0]) { h1[0] = EMPTY_ID; r1[0] = Rec{FL_BLANK, EMPTY_ID, NO_ID, 0, 0, 0, NO_ID, NO_ID}; }
        if (!act[1]) { h1[1] = EMPTY_ID; r1[1] = Rec{FL_BLANK, EMPTY_ID, NO_ID, 0, 0, 0, NO_ID, NO_ID}; }
    }

    if (triage) {
        // Existing-caption triage stops here: LooksBad of the kept tags joined with ", " (Program.cs:206-207, 604, 1141-1147)
        bool phrase = false;
        uint32_t units = 0;
        TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk)
            if (act[blk]) { phrase = phrase || (r1[blk].flags & FL_PHRASE) != 0; units += r1[blk].len16; }
        bool bad = rejected || n == 0 || w.ballot(phrase) != 0;
        if (!bad && n <= 2) bad = w.sum(units) + 2 * (n - 1) > 40;   // fewer than two commas and more than 40 UTF-16 units
        out.verdict = tp_verdict{};
        out.verdict.flags = bad ? (uint8_t)TP_F_TRIAGE_BAD : (uint8_t)0;
        out.verdict.n_candidates = (uint16_t)n;
        out.verdict.safe_hit = -1;
        out.n_final = 0;
        out.caption_len = 0;
        return true;
    }
    }
    w.tick(3);
    TP_STOP(3, h1[0] ^ h1[1] ^ r1[0].flags ^ r1[0].next ^ r1[1].flags ^ r1[0].lit ^ r1[0].step0 ^ ci[0] ^ ci[1] ^ n ^ (uint32_t)act[0] ^ (uint32_t)act[1])
    // ---- S5: SelectTags first half, candidates on lanes (Program.cs:625-655); evidence seeds (Program.cs:721-729) ----------
    // In line ingest the scores fall with the candidate index, so "sorted" order is candidate order and the first of
    // two candidates that NormalizeTag merges carries the merged score, order and source.
    bool nonblank2[2] = {false, false}, first2[2] = {false, false}, bad2[2] = {false, false}, uniq[2] = {false, false}, ev_seed[2] = {false, false};
    uint32_t t2[2] = {EMPTY_ID, EMPTY_ID};
    Rec r2[2];
    r2[0] = r1[0]; r2[1] = r1[1];
    uint32_t style_n = 0;
    uint64_t m_hop2 = 0;
    TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
        t2[blk] = r1[blk].next;
        bool hop = act[blk] && t2[blk] != h1[blk];
        uint64_t mh = w.ballot(hop);
        m_hop2 |= mh;
        if (mh && hop) r2[blk] = t.rec[t2[blk]];
        style_n += popc64(w.ballot(act[blk] && fl_bucket(r1[blk].flags) == B_STYLE && (Front::scored ? strong_in[blk] : ci[blk] < t.style_ci_limit)));   // DetermineStyleQualityCap, Program.cs:702-710
        nonblank2[blk] = act[blk] && !(r2[blk].flags & FL_BLANK);
        first2[blk] = nonblank2[blk];
        bad2[blk] = nonblank2[blk] && (!(r2[blk].flags & FL_VALID) || (r2[blk].flags & FL_DROP));
        ev_seed[blk] = Front::scored ? seed_in[blk] : ci[blk] < t.seed_ci_limit;   // !(score < 0.28), Program.cs:724
    }
    if constexpr (Front::scored) style_n += front->extra_style;
    const int style_cap = style_n < 2 ? 2 : (style_n > 4 ? 4 : (int)style_n);
    if (m_hop2) {   // a hop may have merged two candidates: keep first occurrences
        if (!dedupe_first(w, L, t2, nonblank2, first2, nblk)) return false;
        // With scores of their own the merged score, order and source may move the tag in the sorted order (Program.cs:636-655): not followed here
        if (Front::scored && w.ballot((nonblank2[0] && !first2[0]) || (nonblank2[1] && !first2[1]))) return false;
    }
    uint32_t n_dropped = 0;
    TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
        uniq[blk] = first2[blk] && !bad2[blk];
        n_dropped += popc64(w.ballot(first2[blk] && bad2[blk]));                            // distinct "{n}:invalid_tag" / "{n}:policy_drop"
        ev_seed[blk] = ev_seed[blk] && first2[blk];
    }

    if constexpr (Front::scored) n_dropped += front->extra_dropped;
    if constexpr (AUD) {
        // DroppedTagsWithReason in candidate order (Program.cs:630-631, 682), the first TOP_TAGS raw candidates (Program.cs:860), and
        // what BuildReviewRecord finds among the normalised raw candidates (Program.cs:827-837)
        if (n_dropped > AUD_TOP - AUD_DROPPED) return false;        // no room in the row
        const uint64_t mb0 = w.ballot(first2[0] && bad2[0]), mb1 = nblk > 1 ? w.ballot(first2[1] && bad2[1]) : 0;
        uint32_t dbytes = 0, tbytes = 0, rb = 0;
        bool banned = false;
        TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
            if (first2[blk] && bad2[blk]) {
                const uint32_t pos = blk ? popc64(mb0) + w.rank(mb1) : w.rank(mb0);
                aud.row[AUD_DROPPED + pos] = t2[blk] | ((r2[blk].flags & FL_VALID) ? 0x80000000u : 0u);
                dbytes += r2[blk].len + AUDIT_REASON_LEN;
            }
            if (!Front::scored && act[blk] && ci[blk] < (uint32_t)TOP_TAGS) { aud.row[AUD_TOP + ci[blk]] = h1[blk]; tbytes += r1[blk].len; }   // (the scored front lists its own)
            if (nonblank2[blk]) { rb |= (r2[blk].flags & (FL_SUBJ_MALE | FL_SUBJ_GIRL | FL_SUBJ_MULTI)) >> 10; banned = banned || (r2[blk].flags & FL_BANNED) != 0; }
        }
        uint32_t rbits = w.or_all(rb);
        if (w.ballot(banned) != 0)
            for (int k = 0; k < N_BANNED; ++k)
                if (w.ballot((nonblank2[0] && t2[0] == t.banned_ids[k]) || (nonblank2[1] && t2[1] == t.banned_ids[k])) != 0) rbits |= 1u << k;
        dbytes = w.sum(dbytes); tbytes = w.sum(tbytes);
        const uint32_t ntop = n < (uint32_t)TOP_TAGS ? n : (uint32_t)TOP_TAGS;
        if (lane == 0) {
            aud.cnt[AUDIT_DROPPED] = n_dropped; aud.len[AUDIT_DROPPED] = dbytes + (n_dropped ? n_dropped - 1 : 0u);
            if (!Front::scored) { aud.cnt[AUDIT_TOP] = ntop; aud.len[AUDIT_TOP] = tbytes + (ntop ? ntop - 1 : 0u); }
            *aud.rbits = rbits;
        }
    }
    w.tick(4);
    TP_STOP(4, t2[0] ^ t2[1] ^ r2[0].flags ^ r2[1].flags ^ r2[0].next ^ (uint32_t)uniq[0] ^ (uint32_t)uniq[1] ^ (uint32_t)ev_seed[0] ^ (uint32_t)ev_seed[1] ^ (uint32_t)style_cap ^ n_dropped)
    // ---- S6: SelectTags second half (Program.cs:661-699) -----------------------------------------------------------------
    uint32_t t3[2] = {EMPTY_ID, EMPTY_ID};
    bool ok3[2] = {false, false};
    uint64_t m_hop3 = 0;
    TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
        t3[blk] = r2[blk].next;
        bool hop = uniq[blk] && t3[blk] != t2[blk];
        uint64_t mh = w.ballot(hop);
        m_hop3 |= mh;
        uint32_t f3 = r2[blk].flags;
        if (mh && hop) f3 = t.rec[t3[blk]].flags;
        ok3[blk] = uniq[blk] && !(f3 & FL_BLANK) && (f3 & FL_VALID) && !(f3 & FL_DROP) && !(f3 & FL_PREFIX);   // AddOne's tests; prefix tags are already in
    }
    uint32_t nr = t.n_prefix;
    bool settled = false;
    // per-position columns of the selected tags' implied tags; they reuse the line buffer, which is not read after S4
    uint32_t(*kid)[64] = (uint32_t(*)[64])(L.line);
    uint32_t(*kid_attr)[64] = kid + 2;
    uint32_t(*kid_len)[64] = kid + 4;
    static_assert(sizeof(L.line) >= 6 * 64 * sizeof(uint32_t), "six columns of 64 words");
    {
        const int caps[6] = {3, 10, 8, 6, 4, style_cap};
        bool taken[2] = {false, false};
        uint32_t pos[2] = {0, 0};
        for (uint32_t pass = 0;; ++pass) {
            nr = t.n_prefix;
            taken[0] = taken[1] = false;
            for (uint32_t bk = 0; bk <= 6; ++bk) {                // FillBucket for the six capped buckets in order, then (bk == 6) the final sweep (Program.cs:670-680)
                bool el0 = ok3[0] && !taken[0] && (bk == 6 || fl_bucket(r2[0].flags) == bk);
                bool el1 = ok3[1] && !taken[1] && (bk == 6 || fl_bucket(r2[1].flags) == bk);
                uint64_t m0 = w.ballot(el0), m1 = nblk > 1 ? w.ballot(el1) : 0;
                uint32_t room = nr < (uint32_t)MAX_TOTAL_TAGS ? (uint32_t)MAX_TOTAL_TAGS - nr : 0;
                uint32_t take = popc64(m0) + popc64(m1);
                if (bk < 6 && take > (uint32_t)caps[bk]) take = (uint32_t)caps[bk];
                if (take > room) take = room;
                uint32_t rank0 = w.rank(m0), rank1 = popc64(m0) + w.rank(m1);
                if (el0 && rank0 < take) { taken[0] = true; pos[0] = nr + rank0; }
                if (el1 && rank1 < take) { taken[1] = true; pos[1] = nr + rank1; }
                nr += take;
            }
            if (!m_hop3) break;
            // AddOne's own NormalizeTag (Program.cs:693) moved a tag: two candidates may now name one tag, and the one whose
            // AddOne comes second fails (selected.Add, Program.cs:696) without using up its bucket's share (:688).  Positions are
            // the order of the successful AddOne calls, and up to the first such failure the fills above are what the
            // reference does: the candidate that fails there leaves, and the fills are done again without it.
            bool fst[2];
            if (!dedupe_lowest(w, L, t3, taken, pos, fst, nblk)) return false;
            const bool lose0 = taken[0] && !fst[0], lose1 = taken[1] && !fst[1];
            const uint32_t lo = w.or_all((lose0 && pos[0] < 32 ? 1u << pos[0] : 0u) | (lose1 && pos[1] < 32 ? 1u << pos[1] : 0u));
            const uint32_t hi = w.or_all((lose0 && pos[0] >= 32 ? 1u << (pos[0] - 32) : 0u) | (lose1 && pos[1] >= 32 ? 1u << (pos[1] - 32) : 0u));
            if ((lo | hi) == 0) break;
            if (pass >= 16) return false;
            const uint32_t first_fail = lo ? ctz32(lo) : 32u + ctz32(hi);
            if (lose0 && pos[0] == first_fail) ok3[0] = false;
            if (lose1 && pos[1] == first_fail) ok3[1] = false;
        }
        // A file is on the settled-tags path when NormalizeTag leaves every selected tag alone and every selected tag's
        // implied tags are settled leaves (FL_LEAF2): then flags and lengths of everything the final list can hold are known
        // from records already loaded and from the rule cache, and the rest of the pipeline needs no further table lookups.
        bool unsettled = false;
        TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk)
            if (taken[blk] && (r2[blk].next != t2[blk] || ((r2[blk].flags & FL_EDGE) && !(r2[blk].flags & FL_LEAF2)))) unsettled = true;
        settled = t.prefix_simple && nr >= 3 && nr <= 64 && !w.ballot(unsettled);
        if (settled) { kid[0][lane] = NO_ID; kid[1][lane] = NO_ID; }
        w.sync();
        if (lane < t.n_prefix) { L.fin[lane] = t.prefix_ids[lane]; L.fattr[lane] = t.prefix_flags[lane]; L.flen[lane] = t.prefix_len[lane]; }
        TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
            if (taken[blk]) {
                const uint32_t p = pos[blk];
                L.fin[p] = t3[blk];
                L.fattr[p] = r2[blk].flags;
                L.flen[p] = r2[blk].len;
                if (settled && (r2[blk].flags & FL_EDGE)) {          // its implied tags, for the one closure round of S9
                    if (r2[blk].step_cnt > 0) { kid[0][p] = r2[blk].step0; kid_attr[0][p] = r2[blk].s0_flags; kid_len[0][p] = r2[blk].s0_len; }
                    if (r2[blk].step_cnt > 1) { kid[1][p] = r2[blk].step1; kid_attr[1][p] = r2[blk].s1_flags; kid_len[1][p] = r2[blk].s1_len; }
                }
            }
        }
        w.sync();
    }
    const uint32_t n_selected = nr;

    // ---- S8: evidence closure (Program.cs:301-302) -> literal bitset, count, feet/penis; the back half runs it after the
    // coverage step, as the reference does ---------------------------------------------------------------------------------
    auto evidence = [&](uint32_t& n_evidence, bool& ev_feet, bool& ev_penis) -> bool {
    for (uint32_t k = lane; k < t.lit_words; k += 64) L.ebits[k] = 0;
    w.sync();
    n_evidence = 0;
    ev_feet = false; ev_penis = false;
    {
        uint64_t m_edge = 0;
        TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
            m_edge |= w.ballot(ev_seed[blk] && (r2[blk].flags & FL_EDGE));
            if (ev_seed[blk]) {
                uint32_t li = r2[blk].lit;
                if (li != NO_ID) w.lds_or(&L.ebits[li >> 5], 1u << (li & 31));
                ev_feet |= (r2[blk].flags & FL_NFEET) != 0;         // HasTag(evidenceTags, ...), Program.cs:309, 719
                ev_penis |= (r2[blk].flags & FL_NPENIS) != 0;
            }
        }
        uint64_t m_far = 0;                                        // seeds whose closure reaches beyond their own two steps
        TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) m_far |= w.ballot(ev_seed[blk] && (r2[blk].flags & FL_EDGE) && !(r2[blk].flags & FL_REACH2));
        const bool count = fired != nullptr;                       // the size of the evidence set is a detail output: it takes the set to count
        if constexpr (AUD) {
            // The evidence set as a list in the reference's insertion order (BuildEvidenceTags, then ExpandImpliedTags' rounds:
            // Program.cs:721-729, 1160-1179): the seeds in candidate order; then, round by round, what the round appends - for
            // every member the round before appended, in their order, what a visit of it appends.  Every seed brings the list of
            // its own closure in that order with the round of each entry (DevTables::reach_ent, or its record's two steps); a
            // round's additions over all seeds are ordered by seed, then by place in the seed's list; a string counts where it
            // comes first.  So every entry goes into the set under the key (round, seed, place), the lowest key of a string
            // wins, and a winner's position is the number of winners with lower keys.
            constexpr uint32_t EV_CAP = AUD_DROPPED - AUD_EVIDENCE;
            if (m_far && (!t.closure_tables || !t.reach_levels_ok)) return false;
            set_clear(w, L);
            const uint64_t ms0 = w.ballot(ev_seed[0]), ms1 = nblk > 1 ? w.ballot(ev_seed[1]) : 0;
            uint32_t total = popc64(ms0) + popc64(ms1);
            if (total > EV_CAP) return false;
            uint32_t ebytes = 0;
            bool full = false;
            uint32_t ne[2] = {0, 0}, e0[2] = {0, 0};                // entries of the seed's list, and where the list starts
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
                if (ev_seed[blk]) {
                    bool created;
                    const uint32_t pos = blk ? popc64(ms0) + w.rank(ms1) : w.rank(ms0);
                    if (set_put(w, L, t2[blk], pos, created) == NO_ID) full = true;
                    aud.row[AUD_EVIDENCE + pos] = t2[blk];
                    ebytes += r2[blk].len;
                    if (r2[blk].flags & FL_EDGE) {
                        if (r2[blk].flags & FL_REACH2) ne[blk] = r2[blk].step_cnt;
                        else { e0[blk] = t.reach_off[t2[blk]]; ne[blk] = t.reach_off[t2[blk] + 1] - e0[blk]; }
                        if (ne[blk] > 0xFFFFu) full = true;
                    }
                }
            }
            w.sync();
            auto entry = [&](int blk, uint32_t k, uint32_t& id, uint32_t& info) {
                if (r2[blk].flags & FL_REACH2) { id = k ? r2[blk].step1 : r2[blk].step0; info = ((k ? r2[blk].s1_info : r2[blk].s0_info) & 0x3FFFFFFu) | 1u << 26; }
                else { id = t.reach_ent[2 * (size_t)(e0[blk] + k)]; info = t.reach_ent[2 * (size_t)(e0[blk] + k) + 1]; }
            };
            auto key_of = [&](int blk, uint32_t k, uint32_t info) { return (info >> 26) << 24 | ci[blk] << 16 | k; };
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
                for (uint32_t k = 0; k < ne[blk] && !full; ++k) {
                    uint32_t id, info;
                    entry(blk, k, id, info);
                    bool created;
                    if (set_put(w, L, id, key_of(blk, k, info), created) == NO_ID) full = true;
                    const uint32_t li = info & 0xFFFFFFu;
                    if (li != 0xFFFFFFu) w.lds_or(&L.ebits[li >> 5], 1u << (li & 31));
                    ev_feet |= (info >> 24) & 1u;
                    ev_penis |= (info >> 25) & 1u;
                }
            }
            if (w.ballot(full)) return false;
            w.sync();
            uint32_t cur[2] = {0, 0};
            for (uint32_t lev = 1; lev < 64; ++lev) {
                if (!w.ballot(cur[0] < ne[0] || cur[1] < ne[1])) break;
                uint32_t wins[2] = {0, 0};
                TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
                    for (uint32_t k = cur[blk]; k < ne[blk]; ++k) {
                        uint32_t id, info;
                        entry(blk, k, id, info);
                        if ((info >> 26) != lev) break;
                        const uint32_t slot = set_find(L, id);
                        wins[blk] += (uint32_t)(slot != NO_ID && L.set_min[slot] == key_of(blk, k, info));
                    }
                }
                const uint32_t incl0 = w.scan_incl(wins[0]), tot0 = w.bcast(incl0, 63);
                uint32_t incl1 = 0, tot1 = 0;
                if (nblk > 1) { incl1 = w.scan_incl(wins[1]); tot1 = w.bcast(incl1, 63); }
                if (total + tot0 + tot1 > EV_CAP) return false;
                uint32_t at[2] = {total + incl0 - wins[0], total + tot0 + incl1 - wins[1]};
                TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
                    uint32_t k = cur[blk];
                    for (; k < ne[blk]; ++k) {
                        uint32_t id, info;
                        entry(blk, k, id, info);
                        if ((info >> 26) != lev) break;
                        const uint32_t slot = set_find(L, id);
                        if (slot != NO_ID && L.set_min[slot] == key_of(blk, k, info)) { aud.row[AUD_EVIDENCE + at[blk]++] = id; ebytes += t.attr2[2 * (size_t)id + 1]; }
                    }
                    cur[blk] = k;
                }
                total += tot0 + tot1;
            }
            if (w.ballot(cur[0] < ne[0] || cur[1] < ne[1])) return false;
            n_evidence = total;
            ebytes = w.sum(ebytes);
            if (lane == 0) { aud.cnt[AUDIT_EVIDENCE] = total; aud.len[AUDIT_EVIDENCE] = ebytes + (total ? total - 1 : 0u); }
        } else
        if (!m_edge) {
            n_evidence = popc64(w.ballot(ev_seed[0])) + (nblk > 1 ? popc64(w.ballot(ev_seed[1])) : 0);
        } else if (!count && !m_far) {
            // literals and the two probes of the seeds' implied tags straight from the seeds' records
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
                if (ev_seed[blk] && (r2[blk].flags & FL_EDGE)) {
                    const uint32_t i0 = r2[blk].step_cnt > 0 ? r2[blk].s0_info : 0xFFFFFFu, i1 = r2[blk].step_cnt > 1 ? r2[blk].s1_info : 0xFFFFFFu;
                    if ((i0 & 0xFFFFFFu) != 0xFFFFFFu) w.lds_or(&L.ebits[(i0 & 0xFFFFFFu) >> 5], 1u << (i0 & 31));
                    if ((i1 & 0xFFFFFFu) != 0xFFFFFFu) w.lds_or(&L.ebits[(i1 & 0xFFFFFFu) >> 5], 1u << (i1 & 31));
                    ev_feet |= ((i0 | i1) >> 24) & 1u;
                    ev_penis |= ((i0 | i1) >> 25) & 1u;
                }
            }
        } else if (!count && t.closure_tables) {
            // nobody asks how many: the literals and the feet / penis facts of every closure member, no set
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < nblk) {
                if (ev_seed[blk] && (r2[blk].flags & FL_EDGE)) {
                    auto note = [&](uint32_t info) {
                        const uint32_t li = info & 0xFFFFFFu;
                        if (li != 0xFFFFFFu) w.lds_or(&L.ebits[li >> 5], 1u << (li & 31));
                        ev_feet |= (info >> 24) & 1u;
                        ev_penis |= (info >> 25) & 1u;
                    };
                    if (r2[blk].flags & FL_REACH2) {
                        if (r2[blk].step_cnt > 0) note(r2[blk].s0_info);
                        if (r2[blk].step_cnt > 1) note(r2[blk].s1_info);
                    } else {
                        const uint32_t m1 = t.reach_off[t2[blk] + 1];
                        for (uint32_t m = t.reach_off[t2[blk]]; m < m1; m += 4) {              // four entries' loads at a time
                            uint32_t info[4];
                            for (uint32_t j = 0; j < 4; ++j) if (m + j < m1) info[j] = t.reach_ent[2 * (size_t)(m + j) + 1];
                            for (uint32_t j = 0; j < 4; ++j) if (m + j < m1) note(info[j]);
                        }
                    }
                }
            }
        } else {
            if (m_far && !t.closure_tables) return false;
            set_clear(w, L);
            uint32_t made = 0;
            bool full = false;
            for (int blk = 0; blk < 2; ++blk) {
                if (ev_seed[blk]) {
                    bool created;
                    if (set_put(w, L, t2[blk], 0, created) == NO_ID) full = true;
                    made += created;
                }
            }
            w.sync();
            auto add = [&](uint32_t id, uint32_t info) {            // one member of a seed's closure
                bool created;
                if (set_put(w, L, id, 0, created) == NO_ID) { full = true; return; }
                if (created) {
                    ++made;
                    const uint32_t li = info & 0xFFFFFFu;
                    if (li != 0xFFFFFFu) w.lds_or(&L.ebits[li >> 5], 1u << (li & 31));
                }
                ev_feet |= (info >> 24) & 1u;
                ev_penis |= (info >> 25) & 1u;
            };
            for (int blk = 0; blk < 2; ++blk) {
                if (ev_seed[blk] && (r2[blk].flags & FL_EDGE)) {
                    if (r2[blk].flags & FL_REACH2) {                // the usual case: the record itself lists the closure
                        if (r2[blk].step_cnt > 0) add(r2[blk].step0, r2[blk].s0_info);
                        if (r2[blk].step_cnt > 1) add(r2[blk].step1, r2[blk].s1_info);
                    } else {
                        const uint32_t m1 = t.reach_off[t2[blk] + 1];
                        for (uint32_t m = t.reach_off[t2[blk]]; m < m1 && !full; m += 4) {      // four entries' loads at a time
                            uint64_t e[4];
                            for (uint32_t j = 0; j < 4; ++j) if (m + j < m1) e[j] = *(const uint64_t*)(t.reach_ent + 2 * (size_t)(m + j));
                            for (uint32_t j = 0; j < 4; ++j) if (m + j < m1) add((uint32_t)e[j], (uint32_t)(e[j] >> 32));
                        }
                    }
                }
            }
            if (w.ballot(full)) return false;
            n_evidence = w.sum(made);
            if (n_evidence > FAST_SET_SLOTS * 3 / 4) return false;
        }
        if constexpr (Front::scored) n_evidence += front->extra_evidence;
        if (!count && !AUD) n_evidence = 0xFFFF;
        ev_feet = w.ballot(ev_feet) != 0;
        ev_penis = w.ballot(ev_penis) != 0;
        w.sync();
    }
    return true;
    };

    TailIn ti;
    ti.nr = nr; ti.n = n; ti.n_selected = n_selected; ti.n_dropped = n_dropped;
    if constexpr (Front::scored) ti.n = front->n_candidates;
    ti.settled = settled; ti.rejected = rejected; ti.kid = kid;
    return fast_tail<NF, AUD>(w, t, L, rc, ti, evidence, out, fin32, fired, facet_choice, aud);
}

// ================================================================================================================
// Scored candidates: the tagger replies of one file as (text, score) pairs (ParseJoyTags, Program.cs:463-489), one reply or
// the primary and the secondary one merged as the per-image loop does (Program.cs:214-221, 442-461, 801-815).  run() leaves
// the candidates on the lanes in the order SelectTags sorts them into.  Files it does not take (more than 128 tags or
// FAST_LINE_MAX bytes of text, bytes outside ASCII, a line feed, a tag in no table, two tags of one reply that NormalizeTag
// maps to one string, candidates that tie in score, source and order) go to the general kernel.
// ================================================================================================================
struct ScoredIn {
    const uint8_t* arena;
    const uint64_t* tag_offsets;
    uint64_t lo, mid, hi;          // tags [lo, hi); merged: [lo, mid) primary reply, [mid, hi) secondary reply
    const double* scores;
    const uint8_t* sources;        // per tag (not read when merged)
    bool merged;
    int cap;                       // LimitSecondaryTags' cap on tags the primary reply does not hold
};

TP_HD uint32_t classify_dyn(const DevTables& t, const uint8_t* p, uint32_t n);   // tp_pipeline.h: attributes of a string in no table

template <class W>
struct ScoredFront {
    static constexpr bool scored = true;
    ScoredIn in;
    // Candidates in no table (strict vocabulary drops them in SelectTags, Program.cs:631-634, but they count): distinct strings
    // among the candidates, those of them with !(score < 0.28) (evidence, Program.cs:724), those in the style bucket with score >= 0.60
    uint32_t extra_dropped = 0, extra_evidence = 0, extra_style = 0;
    // Audit mode (aud.row set) lists the candidates by name.  The first 20 by (score, order) - TopJoyTags, Program.cs:860: every
    // entry of a single reply on its own, the merged candidates of two - are written here, where the entries' own scores and
    // orders still are (the lanes end up in SelectTags' order, which looks at the source rank before the order and joins the
    // entries of one string).  audit_unfit: a candidate in no table, which has no id to be listed by, or a single reply whose
    // strings do not stand in the lanes' order (the dropped tags are listed in the order of ParseJoyTags' list, Program.cs:487,
    // 627-631: equal scores between strings of different source ranks, or a joined string whose best entry is not its earliest).
    LeanAudit aud{nullptr, nullptr, nullptr, nullptr};
    bool audit_unfit = false;
    // entry e's place among the live entries by (score descending, order, source rank), when below TOP_TAGS: its id into the row
    TP_HD void top_tags(W& w, const DevTables& t, FastLds& L, uint32_t M, const bool (&live)[2], const uint32_t (&id)[2], const uint64_t (&K)[2],
                        const uint32_t (&order)[2], const uint32_t (&src)[2]) {
        uint64_t* Kq = (uint64_t*)L.line;                          // the text is not read any more
        uint32_t* oq = (uint32_t*)(L.line + 1024);
        const uint32_t lane = w.lane();
        uint32_t key2[2];
        w.sync();
        TP_UNROLL for (int blk = 0; blk < 2; ++blk) {
            key2[blk] = live[blk] ? (order[blk] << 8 | (uint32_t)source_rank(src[blk])) : 0xFFFFFFFFu;
            const uint32_t e = (uint32_t)blk * 64 + lane;
            if (e < M) { Kq[e] = K[blk]; oq[e] = key2[blk]; }
        }
        w.sync();
        uint32_t before[2] = {0, 0};
        for (uint32_t f = 0; f < M; ++f) {
            const uint32_t of = oq[f];
            if (of == 0xFFFFFFFFu) continue;
            const uint64_t kf = Kq[f];
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) before[blk] += (uint32_t)(kf < K[blk] || (kf == K[blk] && of < key2[blk]));
        }
        uint32_t tbytes = 0;
        TP_UNROLL for (int blk = 0; blk < 2; ++blk)
            if (live[blk] && before[blk] < (uint32_t)TOP_TAGS) { aud.row[AUD_TOP + before[blk]] = id[blk]; tbytes += t.attr2[2 * (size_t)id[blk] + 1]; }
        tbytes = w.sum(tbytes);
        const uint32_t n_live = popc64(w.ballot(live[0])) + popc64(w.ballot(live[1]));
        const uint32_t ntop = n_live < (uint32_t)TOP_TAGS ? n_live : (uint32_t)TOP_TAGS;
        if (lane == 0) { aud.cnt[AUDIT_TOP] = ntop; aud.len[AUDIT_TOP] = tbytes + (ntop ? ntop - 1 : 0u); }
        w.sync();
    }
    uint32_t n_candidates = 0;     // finalCandidates.Count (Program.cs:223): one reply keeps entries with one string apart, the merge of two joins them

    // double -> key that sorts ascending where the score sorts descending (finite scores; -0.0 counts as 0.0)
    static TP_HD uint64_t desc_key(double sc) {
        sc = sc + 0.0;
        uint64_t b;
        __builtin_memcpy(&b, &sc, 8);
        const uint64_t asc = b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);
        return ~asc;
    }
    static TP_HD double key_score(uint64_t key) {
        const uint64_t asc = ~key;
        const uint64_t b = asc ^ ((asc >> 63) ? 0x8000000000000000ull : ~0ull);
        double sc;
        __builtin_memcpy(&sc, &b, 8);
        return sc;
    }

    TP_HD bool run(W& w, const DevTables& t, FastLds& L, int& nblk, uint32_t& n, bool (&act)[2], uint32_t (&h1)[2], Rec (&r1)[2], bool (&seed)[2], bool (&strong)[2]) {
        const uint32_t lane = w.lane();
        const uint64_t M64 = in.hi - in.lo;
        if (M64 > 128) return false;
        const uint32_t M = (uint32_t)M64, P = in.merged ? (uint32_t)(in.mid - in.lo) : M;
        nblk = 1; n = 0;
        if (M == 0) {
            if (aud.row && lane == 0) { aud.cnt[AUDIT_TOP] = 0; aud.len[AUDIT_TOP] = 0; }
            return true;
        }
        const uint64_t base = in.tag_offsets[in.lo];
        const uint64_t bytes64 = in.tag_offsets[in.hi] - base;
        if (bytes64 > FAST_LINE_MAX) return false;
        // ---- the tags' text -> LDS, as S1 of scan_line does with a line ----
        const uint8_t* text = in.arena + base;
        const uint32_t shift = (uint32_t)((uintptr_t)text & 15);
        const uint32_t pieces = (shift + (uint32_t)bytes64 + 15) >> 4;
        for (uint32_t c = lane; c < pieces; c += 64) w.copy16(L.line + 16 + 16 * c, text - shift + 16 * c);
        w.sync();
        bool odd = false;                                          // a byte outside ASCII, a line feed (Program.cs:1151's `$` also matches before one)
        for (uint32_t c = lane; c < pieces; c += 64)
            for (uint32_t k = 0; k < 4; ++k) { const uint32_t x = ld32u(L.line + 16 + 16 * c + 4 * k); odd = odd || (x & 0x80808080u) != 0 || eq4(x, rep4('\n')) != 0; }
        if (w.ballot(odd)) return false;
        const uint8_t* s = L.line + 16 + shift;
        const int mblk = M > 64 ? 2 : 1;
        // ---- ParseJoyTags: NormalizeTag of every tag, blank ones skipped (Program.cs:473-487) ----
        uint32_t id[2] = {NO_ID, NO_ID}, src[2] = {0, 0};
        uint64_t K[2] = {~0ull, ~0ull};
        bool valid[2] = {false, false}, decline = false;
        bool unk[2] = {false, false}, ustyle[2] = {false, false};  // a string in no table: told apart by its bytes
        uint64_t uw[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        uint32_t un[2] = {0, 0};
        TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < mblk) {
            const uint32_t e = (uint32_t)blk * 64 + lane;
            if (e < M) {
                const uint32_t i = (uint32_t)(in.tag_offsets[in.lo + e] - base), j = (uint32_t)(in.tag_offsets[in.lo + e + 1] - base);
                K[blk] = desc_key(in.scores[in.lo + e]);
                src[blk] = in.merged ? (e < P ? (uint32_t)TP_SRC_PRIMARY_JOY : (uint32_t)TP_SRC_SECONDARY_JOY) : (uint32_t)in.sources[in.lo + e];
                uint32_t dlen = 0;
                bool vdyn = false;
                Rec rr;
                const int kind = j < i ? 3 : fast_token(t, s, i, j, true, false, id[blk], rr, vdyn, dlen, uw[blk]);
                if (kind == 0) { id[blk] = t.alias1[EMPTY_ID]; rr = t.rec[id[blk]]; }      // NormalizeTag("") still goes through the alias lookup
                if (kind == 4) { unk[blk] = true; un[blk] = dlen; valid[blk] = true; }      // stays a candidate: it counts as dropped and as evidence
                else if (kind >= 2) decline = true;               // a string in no table that may survive, or that took the byte-wise normaliser: general kernel
                else valid[blk] = !(rr.flags & FL_BLANK);
            }
        }
        if (w.ballot(decline)) return false;
        {
            // strings in no table get the element index of their first occurrence as a stand-in id (above every real id)
            const uint64_t mu[2] = {w.ballot(unk[0]), mblk > 1 ? w.ballot(unk[1]) : 0};
            for (int b2 = 0; b2 < 2; ++b2) {
                uint64_t m = mu[b2];
                while (m) {
                    const uint32_t q = ctz64(m);
                    m &= m - 1;
                    const uint32_t qn = w.bcast(un[b2], q);
                    uint64_t qw[4];
                    for (int k = 0; k < 4; ++k) qw[k] = (uint64_t)w.bcast((uint32_t)uw[b2][k], q) | ((uint64_t)w.bcast((uint32_t)(uw[b2][k] >> 32), q) << 32);
                    TP_UNROLL for (int blk = 0; blk < 2; ++blk)
                        if (unk[blk] && id[blk] == NO_ID && un[blk] == qn && uw[blk][0] == qw[0] && uw[blk][1] == qw[1] && uw[blk][2] == qw[2] && uw[blk][3] == qw[3])
                            id[blk] = 0xF0000000u | ((uint32_t)b2 * 64 + q);
                }
            }
            TP_UNROLL for (int blk = 0; blk < 2; ++blk)
                if (unk[blk]) {
                    uint8_t bytes[32];
                    for (int k = 0; k < 32; ++k) bytes[k] = (uint8_t)(uw[blk][k >> 3] >> (8 * (k & 7)));
                    ustyle[blk] = fl_bucket(classify_dyn(t, bytes, un[blk])) == B_STYLE;   // DetermineStyleQualityCap classifies whatever the tagger sent (Program.cs:704-706)
                }
        }
        // order: the running index among the reply's kept tags (Program.cs:474, 486)
        uint32_t order[2] = {0, 0};
        {
            const uint64_t lowP = P >= 64 ? ~0ull : ((1ull << P) - 1), lowP1 = P <= 64 ? 0ull : (P >= 128 ? ~0ull : ((1ull << (P - 64)) - 1));
            const uint64_t v0 = w.ballot(valid[0]), v1 = mblk > 1 ? w.ballot(valid[1]) : 0;
            const uint64_t p0 = v0 & lowP, p1 = v1 & lowP1, s0 = v0 & ~lowP, s1 = v1 & ~lowP1;   // kept tags of the primary / secondary reply
            const uint32_t np0 = popc64(p0), ns0 = popc64(s0);
            order[0] = lane < P ? w.rank(p0) : w.rank(s0);
            order[1] = lane + 64 < P ? np0 + w.rank(p1) : ns0 + w.rank(s1);
        }
        const bool isP[2] = {lane < P, lane + 64 < P};
        bool accept[2] = {valid[0], valid[1]};
        // TopJoyTags of one reply: its entries as they are, before entries of one string are joined
        if (aud.row && !in.merged && !w.ballot((valid[0] && unk[0]) || (valid[1] && unk[1]))) top_tags(w, t, L, M, valid, id, K, order, src);
        if (in.merged) {
            // LimitSecondaryTags (Program.cs:801-815), entry by entry: what the primary reply holds passes; of the other entries
            // the `cap` best by (score, order) pass
            set_clear(w, L);
            bool full = false;
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < mblk)
                if (valid[blk] && isP[blk]) { bool created; full = full || set_put(w, L, id[blk], 0, created) == NO_ID; }
            w.sync();
            if (w.ballot(full)) return false;
            bool sonly[2] = {false, false};
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < mblk) sonly[blk] = valid[blk] && !isP[blk] && set_find(L, id[blk]) == NO_ID;
            const uint32_t n_sonly = popc64(w.ballot(sonly[0])) + (mblk > 1 ? popc64(w.ballot(sonly[1])) : 0);
            if (in.cap <= 0) { if (!isP[0]) accept[0] = false; if (!isP[1]) accept[1] = false; }   // "if (cap <= 0 ...) return []": nothing of the secondary reply passes
            else if (n_sonly > (uint32_t)in.cap) {
                uint64_t* rk = (uint64_t*)L.line;                  // the text is not read any more
                uint32_t* ro = (uint32_t*)(L.line + 1024);
                w.sync();
                TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < mblk) { const uint32_t e = (uint32_t)blk * 64 + lane; if (e < M) { rk[e] = K[blk]; ro[e] = sonly[blk] ? order[blk] : 0xFFFFFFFFu; } }
                w.sync();
                uint32_t before[2] = {0, 0};
                for (uint32_t f = P; f < M; ++f) {
                    const uint32_t of = ro[f];
                    if (of == 0xFFFFFFFFu) continue;
                    const uint64_t kf = rk[f];
                    TP_UNROLL for (int blk = 0; blk < 2; ++blk) before[blk] += (uint32_t)(kf < K[blk] || (kf == K[blk] && of < order[blk]));
                }
                TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (sonly[blk] && before[blk] >= (uint32_t)in.cap) accept[blk] = false;
            }
            w.sync();
        }
        // Entries with one normalised string become one candidate with the best score and the earliest order (ParseJoyTags keeps
        // them apart, MergeCandidateTags and SelectTags join them: Program.cs:444-458, 636-649); its source is the one of lower
        // rank wherever PreferSource is asked with the scores it gets there, which between the two replies is the primary one.
        bool alive[2] = {false, false}, joined = false;            // joined (audit mode): a single reply held two entries of one string,
        uint32_t list_order[2] = {0, 0};                           //   which stands in the reply's list at the order of its best entry
        {
            uint64_t* gK = (uint64_t*)L.line;                      // per set slot: the best key of the entries with that string
            static_assert(FAST_SET_SLOTS * 8 <= FAST_LINE_MAX && FAST_SET_SLOTS / 32 <= FAST_FRULES_MAX / 32, "per-slot columns fit the line buffer and the facet-hit words");
            set_clear(w, L);
            for (uint32_t k = lane; k < FAST_SET_SLOTS; k += 64) gK[k] = ~0ull;
            if (lane < FAST_SET_SLOTS / 32) L.fhit[lane] = 0;      // bit per slot: an entry of the primary reply is among them
            w.sync();
            uint32_t slot[2] = {NO_ID, NO_ID};
            bool rep[2] = {false, false}, full = false;
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < mblk)
                if (accept[blk]) {
                    slot[blk] = set_put(w, L, id[blk], order[blk], rep[blk]);
                    if (slot[blk] == NO_ID) full = true;
                    else {
                        w.lds_min64(&gK[slot[blk]], K[blk]);
                        if (in.merged && isP[blk]) w.lds_or(&L.fhit[slot[blk] >> 5], 1u << (slot[blk] & 31));
                    }
                }
            w.sync();
            if (w.ballot(full)) return false;
            if (!in.merged && w.ballot((accept[0] && !rep[0]) || (accept[1] && !rep[1]))) {
                if (aud.row) {
                    // where a joined string stands in ParseJoyTags' list: at its best entry, the earliest of them when scores tie
                    auto best_order = [&](uint32_t k) -> uint32_t& { return k < FAST_FIN_MAX ? L.fin[k] : L.fattr[k - FAST_FIN_MAX]; };
                    static_assert(FAST_SET_SLOTS <= 2 * FAST_FIN_MAX, "a word per set slot in the two arrays");
                    for (uint32_t k = lane; k < FAST_SET_SLOTS; k += 64) best_order(k) = 0xFFFFFFFFu;
                    w.sync();
                    TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < mblk)
                        if (accept[blk] && K[blk] == gK[slot[blk]]) w.lds_min(&best_order(slot[blk]), order[blk]);
                    w.sync();
                    TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < mblk) if (accept[blk] && rep[blk]) list_order[blk] = best_order(slot[blk]);
                    joined = true;
                }
                // one reply with sources of its own per tag: joined entries of different sources would need PreferSource followed
                const uint64_t m0 = w.ballot(accept[0]);
                const uint32_t ref = m0 ? w.bcast(src[0], ctz64(m0)) : w.bcast(src[1], ctz64(w.ballot(accept[1])));
                if (w.ballot((accept[0] && src[0] != ref) || (accept[1] && src[1] != ref))) return false;
            }
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < mblk) {
                alive[blk] = accept[blk] && rep[blk];
                if (alive[blk]) {
                    K[blk] = gK[slot[blk]];
                    order[blk] = L.set_min[slot[blk]];
                    if (in.merged) src[blk] = ((L.fhit[slot[blk] >> 5] >> (slot[blk] & 31)) & 1u) ? (uint32_t)TP_SRC_PRIMARY_JOY : (uint32_t)TP_SRC_SECONDARY_JOY;
                }
            }
            w.sync();
        }
        // ---- the order SelectTags works in: score descending, source rank, order (Program.cs:460, 651-655) ----
        uint64_t* Kq = (uint64_t*)L.line;
        uint32_t* k2q = (uint32_t*)(L.line + 1024);
        uint32_t* idq = (uint32_t*)(L.line + 1536);
        {
            // the candidates in no table are counted here and take no lane: SelectTags drops every one of them
            bool useed[2], ustrong[2];
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) { const double sc = key_score(K[blk]); useed[blk] = alive[blk] && unk[blk] && !(sc < 0.28); ustrong[blk] = alive[blk] && unk[blk] && ustyle[blk] && sc >= 0.60; }
            n_candidates = popc64(w.ballot(in.merged ? alive[0] : valid[0])) + (mblk > 1 ? popc64(w.ballot(in.merged ? alive[1] : valid[1])) : 0);
            extra_dropped = popc64(w.ballot(alive[0] && unk[0])) + (mblk > 1 ? popc64(w.ballot(alive[1] && unk[1])) : 0);
            extra_evidence = popc64(w.ballot(useed[0])) + (mblk > 1 ? popc64(w.ballot(useed[1])) : 0);
            extra_style = popc64(w.ballot(ustrong[0])) + (mblk > 1 ? popc64(w.ballot(ustrong[1])) : 0);
            if (aud.row) {
                if (w.ballot((valid[0] && unk[0]) || (valid[1] && unk[1]))) audit_unfit = true;
                // TopJoyTags of two replies: the merged candidates by their merged scores and orders (one reply's entries were listed above)
                else if (in.merged) top_tags(w, t, L, M, alive, id, K, order, src);
            }
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (unk[blk]) alive[blk] = false;
        }
        uint32_t k2[2];
        TP_UNROLL for (int blk = 0; blk < 2; ++blk) {
            k2[blk] = alive[blk] ? ((uint32_t)source_rank(src[blk]) << 16) | order[blk] : 0xFFFFFFFFu;
            const uint32_t e = (uint32_t)blk * 64 + lane;
            if (blk < mblk && e < M) { Kq[e] = K[blk]; k2q[e] = k2[blk]; }
        }
        w.sync();
        uint32_t before[2] = {0, 0};
        bool tie = false;
        for (uint32_t f = 0; f < M; ++f) {
            const uint32_t kf2 = k2q[f];
            if (kf2 == 0xFFFFFFFFu) continue;
            const uint64_t kf = Kq[f];
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) {
                before[blk] += (uint32_t)(kf < K[blk] || (kf == K[blk] && kf2 < k2[blk]));
                tie = tie || (alive[blk] && kf == K[blk] && kf2 == k2[blk] && f != (uint32_t)blk * 64 + lane);
            }
        }
        if (aud.row && !in.merged) {
            // A single reply is walked in ParseJoyTags' order, (score descending, order) over its entries (Program.cs:487, 627), each
            // string counting where its first - best - entry stands; the lanes are in (score, source rank, order) of the joined
            // entries.  The two agree unless scores tie: between strings of different ranks, or where a joined string's earliest
            // order is not its best entry's.
            uint32_t* loq = idq;                                   // (written only after this)
            TP_UNROLL for (int blk = 0; blk < 2; ++blk) {
                if (!joined) list_order[blk] = order[blk];
                const uint32_t e = (uint32_t)blk * 64 + lane;
                if (blk < mblk && e < M) loq[e] = list_order[blk];
            }
            w.sync();
            uint32_t earlier[2] = {0, 0};
            for (uint32_t f = 0; f < M; ++f) {
                if (k2q[f] == 0xFFFFFFFFu) continue;
                const uint64_t kf = Kq[f];
                const uint32_t lf = loq[f];
                TP_UNROLL for (int blk = 0; blk < 2; ++blk) earlier[blk] += (uint32_t)(kf < K[blk] || (kf == K[blk] && lf < list_order[blk]));
            }
            if (w.ballot((alive[0] && earlier[0] != before[0]) || (alive[1] && earlier[1] != before[1]))) audit_unfit = true;
        }
        if (w.ballot(tie)) return false;                           // the insertion order of the merge decides between them: general kernel
        n = popc64(w.ballot(alive[0])) + (mblk > 1 ? popc64(w.ballot(alive[1])) : 0);
        w.sync();
        TP_UNROLL for (int blk = 0; blk < 2; ++blk) if (blk < mblk)
            if (alive[blk]) {
                idq[before[blk]] = id[blk];
                const double merged_score = key_score(K[blk]);   // !(score < 0.28) and score >= 0.60 of the merged score
                k2q[before[blk]] = (uint32_t)(!(merged_score < 0.28)) | ((uint32_t)(merged_score >= 0.60) << 1);
            }
        w.sync();
        nblk = n > 64 ? 2 : 1;
        TP_UNROLL for (int blk = 0; blk < 2; ++blk) {
            const uint32_t p = (uint32_t)blk * 64 + lane;
            act[blk] = blk < nblk && p < n;
            if (act[blk]) {
                h1[blk] = idq[p];
                const uint32_t fl = k2q[p];
                seed[blk] = (fl & 1u) != 0;
                strong[blk] = (fl & 2u) != 0;
                r1[blk] = t.rec[h1[blk]];
            }
        }
        w.sync();
        return true;
    }
};

// ================================================================================================================
// Long files: up to LONG_TOK_MAX tokens in LONG_LINE_MAX bytes.  The candidates live in LDS arrays indexed by token and
// are walked 64 at a time; the back half is the same as for short files.  Taken only when NormalizeTag leaves every
// candidate alone (otherwise two candidates may merge, which this path does not follow): everything else goes to the
// general kernel.
// ================================================================================================================
// The state is sized three ways (tokens, line bytes): a wavefront's latencies are hidden by the other wavefronts of its SIMD,
// and how many of those there are is a matter of how many files' state fits the CU's LDS (44 KB each for the largest size:
// three to a CU; 16 KB for the smallest: ten).
constexpr uint32_t LONG_LINE_MAX = 8192;
constexpr uint32_t LONG_TOK_MAX = 1024;

template <uint32_t LINE, uint32_t TOK>
struct alignas(16) LongLdsT : LineLds<LINE, TOK> {
    static constexpr uint32_t SET_SLOTS = 2 * TOK;
    uint32_t big_key[SET_SLOTS];         // token ids, for the first-occurrence filter (Distinct, Program.cs:602)
    uint32_t big_min[SET_SLOTS];         // ... smallest token index per id
    uint32_t c_id[TOK];                  // per token: the id NormalizeTag gives it, NO_ID when it is no candidate (any more)
    uint32_t c_fl[TOK];                  // that string's flags
    uint16_t c_len[TOK];                 // its byte length
    uint16_t c_slot[TOK];                // its slot in the big set
};
typedef LongLdsT<LONG_LINE_MAX, LONG_TOK_MAX> LongLds;           // the largest size
typedef LongLdsT<LONG_LINE_MAX, 512> LongLds512;
typedef LongLdsT<4096, 256> LongLds256;

template <class W, class LL>
TP_HD uint32_t big_put(W& w, LL& L, uint32_t key, uint32_t order) {
    static_assert((LL::SET_SLOTS & (LL::SET_SLOTS - 1)) == 0, "a power of two");
    uint32_t s = ((key * 2654435761u) >> 16) & (LL::SET_SLOTS - 1);
    for (uint32_t probe = 0; probe < LL::SET_SLOTS; ++probe) {
        uint32_t old = w.lds_cas(&L.big_key[s], NO_ID, key);
        if (old == NO_ID || old == key) { w.lds_min(&L.big_min[s], order); return s; }
        s = (s + 1) & (LL::SET_SLOTS - 1);
    }
    return NO_ID;
}

// AUD: audit mode - the file's other lists go to its audit row (LeanAudit), as fast_file leaves them.
template <bool AUD = false, class W, class LL>
TP_HD bool long_file(W& w, const DevTables& t, LL& L, const uint32_t* rc, const uint8_t* cls, const uint8_t* text, uint32_t len, FastOut& out, uint32_t* fin32,
                     uint32_t* fired, uint16_t* facet_choice, bool triage = false, const LeanAudit aud = LeanAudit{nullptr, nullptr, nullptr, nullptr}) {
    constexpr uint32_t TOK_MAX = LL::TOK_CAP;
    const uint32_t lane = w.lane();
    w.tick(0);
    Scan sc;
    if (!scan_line<true, true>(w, L, cls, text, len, sc)) return false;
    const uint8_t* s = sc.s;
    const uint32_t T = sc.ncomma + 1;
    if (T > TOK_MAX) return false;
    w.tick(25);
    static_assert(LL::SET_SLOTS % 256 == 0, "four words per lane and step");
    for (uint32_t k = 0; k < LL::SET_SLOTS / 256; ++k) { w.fill16(&L.big_key[256 * k + 4 * lane], 0xFFFFFFFFu); w.fill16(&L.big_min[256 * k + 4 * lane], 0xFFFFFFFFu); }
    w.sync();

    // ---- tokens -> ids (Program.cs:598-602), 64 at a time -------------------------------------------------------------------
    bool decline = false;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t tix = base + lane;
        uint32_t cid = NO_ID;
        if (tix < T && !sc.rejected) {
            const uint32_t i = tix == 0 ? sc.a : (uint32_t)L.cpos[tix - 1] + 1;
            const uint32_t j = tix == T - 1 ? sc.b : (uint32_t)L.cpos[tix];
            uint32_t id = NO_ID, dlen = 0;
            bool vdyn = false;
            const bool special = (L.tspec[tix >> 5] >> (tix & 31)) & 1u, high = (L.thigh[tix >> 5] >> (tix & 31)) & 1u;
            Rec rr;
            const int kind = fast_token(t, s, i, j, special, high, id, rr, vdyn, dlen);
            if (kind == 1) {
                if ((rr.flags & FL_LEN2) && !(rr.flags & FL_DROP)) {
                    if ((rr.next != id && !triage) || (rr.flags & FL_SPLIT)) decline = true;  // NormalizeTag would change it again; the join + re-split would cut it up
                    const uint32_t slot = big_put(w, L, id, tix);
                    if (slot == NO_ID) decline = true;
                    cid = id;
                    L.c_fl[tix] = rr.flags;
                    L.c_len[tix] = (uint16_t)rr.len;
                    L.c_slot[tix] = (uint16_t)slot;
                    if (rr.len > 0xFFFFu) decline = true;
                }
            } else if (kind == 3) {
                decline = true;
            } else if (kind == 2) {
                if (!t.drop_unknown && dlen >= 2 && vdyn) decline = true;   // a string outside the tables would survive
            }
        }
        if (tix < TOK_MAX) L.c_id[tix] = cid;
    }
    if (w.ballot(decline)) return false;
    w.sync();
    w.tick(26);
    if (triage) {
        // Existing-caption triage: LooksBad of the kept tags joined with ", " (Program.cs:206-207, 604, 1141-1147)
        uint32_t k = 0, units = 0;
        bool phrase = false;
        for (uint32_t base = 0; base < T; base += 64) {
            const uint32_t tix = base + lane;
            const uint32_t id = tix < T ? L.c_id[tix] : NO_ID;
            const bool act = id != NO_ID && L.big_min[L.c_slot[tix]] == tix;
            k += popc64(w.ballot(act));
            if (act) phrase = phrase || (L.c_fl[tix] & FL_PHRASE) != 0;
            if (act && k <= 2) units += t.rec[id].len16;           // only matters for at most two tags (k never falls again)
        }
        bool bad = sc.rejected || k == 0 || w.ballot(phrase) != 0;
        if (!bad && k <= 2) bad = w.sum(units) + 2 * (k - 1) > 40;
        out.verdict = tp_verdict{};
        out.verdict.flags = bad ? (uint8_t)TP_F_TRIAGE_BAD : (uint8_t)0;
        out.verdict.n_candidates = (uint16_t)k;
        out.verdict.safe_hit = -1;
        out.n_final = 0;
        out.caption_len = 0;
        return true;
    }

    // ---- candidates in order: first occurrences; dropped-tag count, style cap, evidence (Program.cs:625-655, 702-710, 721-729) ---
    const bool count = fired != nullptr;
    for (uint32_t k = lane; k < t.lit_words; k += 64) L.ebits[k] = 0;
    set_clear(w, L);                                               // the evidence set, when it is to be counted
    uint32_t n = 0, n_dropped = 0, style_n = 0, made = 0;
    bool ev_feet = false, ev_penis = false, full = false, far_no_table = false;
    uint64_t any_edge = 0;
    // audit mode: bytes of the dropped tags' and the first tagger tags' text, review bits, seeds so far and their bytes
    uint32_t dbytes = 0, tbytes = 0, rb = 0, n_seed = 0, ebytes = 0;
    if (AUD && (!t.reach_levels_ok || !count)) return false;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t tix = base + lane;
        const uint32_t id = tix < T ? L.c_id[tix] : NO_ID;
        const bool act = id != NO_ID && L.big_min[L.c_slot[tix]] == tix;
        const uint32_t f = act ? L.c_fl[tix] : (uint32_t)FL_BLANK;
        const uint64_t m = w.ballot(act);
        const uint32_t ci = n + w.rank(m);
        n += popc64(m);
        const bool nonblank = act && !(f & FL_BLANK);
        const bool bad = nonblank && (!(f & FL_VALID) || (f & FL_DROP));
        if constexpr (AUD) {
            // DroppedTagsWithReason in candidate order (Program.cs:630-631, 682), the first TOP_TAGS raw candidates (Program.cs:860),
            // what BuildReviewRecord finds among the normalised raw candidates (Program.cs:827-837)
            const uint64_t mb = w.ballot(bad);
            if (n_dropped + popc64(mb) > AUD_TOP - AUD_DROPPED) return false;   // no room in the row
            if (bad) { aud.row[AUD_DROPPED + n_dropped + w.rank(mb)] = id | ((f & FL_VALID) ? 0x80000000u : 0u); dbytes += (uint32_t)L.c_len[tix] + AUDIT_REASON_LEN; }
            if (act && ci < (uint32_t)TOP_TAGS) { aud.row[AUD_TOP + ci] = id; tbytes += L.c_len[tix]; }
            if (nonblank) {
                rb |= (f & (FL_SUBJ_MALE | FL_SUBJ_GIRL | FL_SUBJ_MULTI)) >> 10;
                if (f & FL_BANNED) for (int k = 0; k < N_BANNED; ++k) if (id == t.banned_ids[k]) rb |= 1u << k;
            }
        }
        n_dropped += popc64(w.ballot(bad));
        style_n += popc64(w.ballot(act && fl_bucket(f) == B_STYLE && ci < t.style_ci_limit));
        const bool ok = nonblank && !bad && !(f & FL_PREFIX);      // AddOne's tests; prefix tags are already in
        if (tix < T && !ok) L.c_id[tix] = NO_ID;                  // from here on c_id holds what the bucket fills may take
        const bool seed = nonblank && ci < t.seed_ci_limit;        // !(score < 0.28), Program.cs:724
        const uint64_t m_seed = w.ballot(seed);
        if (m_seed) {
            Rec r;
            r.flags = 0; r.lit = NO_ID; r.step_cnt = 0;
            if (seed) r = t.rec[id];
            if (seed) {
                if (r.lit != NO_ID) w.lds_or(&L.ebits[r.lit >> 5], 1u << (r.lit & 31));
                ev_feet |= (r.flags & FL_NFEET) != 0;
                ev_penis |= (r.flags & FL_NPENIS) != 0;
            }
            const bool edge = seed && (r.flags & FL_EDGE);
            any_edge |= w.ballot(edge);
            // audit mode: the set's order values make the evidence list (see fast_file): a seed's is its place among the seeds,
            // a closure member's (round, seed, place in the seed's list)
            const uint32_t spos = n_seed + w.rank(m_seed);
            n_seed += popc64(m_seed);
            uint32_t place = 0;
            auto add = [&](uint32_t x, uint32_t info) {            // one member of a seed's closure
                bool created = true;
                const uint32_t order = AUD ? ((info >> 26) ? (info >> 26) : 1u) << 24 | spos << 16 | place : 0u;
                if (AUD && place > 0xFFFFu) { full = true; return; }
                ++place;
                if (count) { if (set_put(w, L, x, order, created) == NO_ID) { full = true; return; } }
                if (created) {
                    ++made;
                    const uint32_t li = info & 0xFFFFFFu;
                    if (li != 0xFFFFFFu) w.lds_or(&L.ebits[li >> 5], 1u << (li & 31));
                }
                ev_feet |= (info >> 24) & 1u;
                ev_penis |= (info >> 25) & 1u;
            };
            if (count && seed) { bool created; if (set_put(w, L, id, AUD ? spos : 0u, created) == NO_ID) full = true; made += created; }
            if constexpr (AUD) { if (seed) { aud.row[AUD_EVIDENCE + spos] = id; ebytes += L.c_len[tix]; } }
            w.sync();
            if (edge) {
                if (r.flags & FL_REACH2) {                          // (its two steps belong to the first round)
                    if (r.step_cnt > 0) add(r.step0, r.s0_info & 0x3FFFFFFu);
                    if (r.step_cnt > 1) add(r.step1, r.s1_info & 0x3FFFFFFu);
                } else if (!t.closure_tables) {
                    far_no_table = true;
                } else {
                    for (uint32_t q = t.reach_off[id]; q < t.reach_off[id + 1] && !full; ++q) add(t.reach_ent[2 * q], t.reach_ent[2 * q + 1]);
                }
            }
        }
    }
    if (w.ballot(full || far_no_table)) return false;
    uint32_t n_evidence = 0xFFFF;
    if (count) {
        n_evidence = w.sum(made);
        if (n_evidence > FAST_SET_SLOTS * 3 / 4) return false;
    }
    ev_feet = w.ballot(ev_feet) != 0;
    ev_penis = w.ballot(ev_penis) != 0;
    w.sync();
    if constexpr (AUD) {
        // the evidence list behind the seeds: the set's slots whose order value is a closure member's hold exactly the strings
        // the rounds append, and the order values' order is theirs (four slots per lane; the keys side by side in the big
        // set's array, which nothing reads any more)
        if (n_evidence > AUD_DROPPED - AUD_EVIDENCE) return false;
        uint32_t wid[4], wkey[4], mine = 0;
        for (uint32_t j = 0; j < 4; ++j) {
            wid[j] = L.set_key[4 * lane + j]; wkey[j] = L.set_min[4 * lane + j];
            if (wid[j] == NO_ID || wkey[j] < (1u << 24)) wkey[j] = 0xFFFFFFFFu; else ++mine;
        }
        const uint32_t incl = w.scan_incl(mine), grown = w.bcast(incl, 63);
        uint32_t at = incl - mine;
        for (uint32_t j = 0; j < 4; ++j) if (wkey[j] != 0xFFFFFFFFu) L.big_key[at++] = wkey[j];
        w.sync();
        uint32_t rank[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < grown; ++i) {
            const uint32_t k = L.big_key[i];
            for (uint32_t j = 0; j < 4; ++j) rank[j] += (uint32_t)(k < wkey[j]);
        }
        for (uint32_t j = 0; j < 4; ++j) if (wkey[j] != 0xFFFFFFFFu) { aud.row[AUD_EVIDENCE + n_seed + rank[j]] = wid[j]; ebytes += t.attr2[2 * (size_t)wid[j] + 1]; }
        const uint32_t total = n_seed + grown;
        dbytes = w.sum(dbytes); tbytes = w.sum(tbytes); ebytes = w.sum(ebytes);
        const uint32_t rbits = w.or_all(rb), ntop = n < (uint32_t)TOP_TAGS ? n : (uint32_t)TOP_TAGS;
        if (lane == 0) {
            aud.cnt[AUDIT_EVIDENCE] = total; aud.len[AUDIT_EVIDENCE] = ebytes + (total ? total - 1 : 0u);
            aud.cnt[AUDIT_DROPPED] = n_dropped; aud.len[AUDIT_DROPPED] = dbytes + (n_dropped ? n_dropped - 1 : 0u);
            aud.cnt[AUDIT_TOP] = ntop; aud.len[AUDIT_TOP] = tbytes + (ntop ? ntop - 1 : 0u);
            *aud.rbits = rbits;
        }
        w.sync();
    }
    w.tick(27);

    // ---- bucket fills (Program.cs:661-699): every bucket walks the candidates in order -------------------------------------------
    const int style_cap = style_n < 2 ? 2 : (style_n > 4 ? 4 : (int)style_n);
    const int caps[6] = {3, 10, 8, 6, 4, style_cap};
    uint32_t nr = t.n_prefix;
    if (lane < t.n_prefix) { L.fin[lane] = t.prefix_ids[lane]; L.fattr[lane] = t.prefix_flags[lane]; L.flen[lane] = t.prefix_len[lane]; }
    for (uint32_t bk = 0; bk <= 6; ++bk) {
        uint32_t room = nr < (uint32_t)MAX_TOTAL_TAGS ? (uint32_t)MAX_TOTAL_TAGS - nr : 0;
        uint32_t rem = bk < 6 ? ((uint32_t)caps[bk] < room ? (uint32_t)caps[bk] : room) : room;
        for (uint32_t base = 0; base < T && rem; base += 64) {
            const uint32_t tix = base + lane;
            const uint32_t id = tix < T ? L.c_id[tix] : NO_ID;
            const uint32_t f = id != NO_ID ? L.c_fl[tix] : 0;
            const bool el = id != NO_ID && (bk == 6 || fl_bucket(f) == bk);
            const uint64_t m = w.ballot(el);
            const uint32_t avail = popc64(m), take = avail < rem ? avail : rem;
            const uint32_t rank = w.rank(m);
            if (el && rank < take) {
                const uint32_t p = nr + rank;
                L.fin[p] = id; L.fattr[p] = f; L.flen[p] = L.c_len[tix];
                L.c_id[tix] = NO_ID;                               // taken
            }
            nr += take;
            rem -= take;
        }
    }
    w.sync();

    w.tick(28);
    auto evidence = [&](uint32_t& ne, bool& feet, bool& penis) -> bool { ne = n_evidence; feet = ev_feet; penis = ev_penis; return true; };
    TailIn ti;
    ti.nr = nr; ti.n = n; ti.n_selected = nr; ti.n_dropped = n_dropped;
    ti.settled = false; ti.rejected = sc.rejected; ti.kid = nullptr;
    return fast_tail<2, AUD>(w, t, L, rc, ti, evidence, out, fin32, fired, facet_choice, aud);
}

}  // namespace tp
