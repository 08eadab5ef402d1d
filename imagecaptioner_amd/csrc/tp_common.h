// Shared host/device definitions of the batch tag pipeline: string handles,
// per-string attribute bits, the device table image, hashing.
//
// Everything the per-file pipeline compares is an INTERNED string (an id into a
// table compiled on the host from the rule/lexicon/vocabulary files) or a
// DYNAMIC string (normalised bytes of an input token that is in no table).
// The reference re-applies NormalizeTag (Program.cs:1149-1158) at about eight
// sites per tag; for an interned id that is one table read (next[id]), for a
// dynamic string it is the identity unless the string ends in a quote character.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#ifdef TP_HD_ALL_FLAT                                               // experiment: every helper always inlined
#define TP_HD __host__ __device__ __attribute__((always_inline)) inline
#else
#define TP_HD __host__ __device__ inline
#endif
// Always inlined, whatever the number of call sites: a function body that two kernels of one file share is otherwise inlined
// at a later stage, and the register allocation of both kernels gets worse for it (spills in the lean stage's back half).
#define TP_HD_FLAT __host__ __device__ __attribute__((always_inline)) inline
#else
#define TP_HD inline
#define TP_HD_FLAT inline
#endif

namespace tp {

// ---- string handle -----------------------------------------------------------------------------
// interned: the id (upper 32 bits zero).
// dynamic : bit63 | bit62 = ends in a quote (one more NormalizeTag changes it) | attr bits 48..55 |
//           length 32..47 | byte offset into the file's normalised-token buffer 0..31.
typedef uint64_t H;
constexpr H H_DYN = 1ull << 63;
constexpr H H_UNSTABLE = 1ull << 62;
TP_HD bool h_is_dyn(H h) { return (h & H_DYN) != 0; }
TP_HD uint32_t h_id(H h) { return (uint32_t)h; }
TP_HD uint32_t h_off(H h) { return (uint32_t)h; }
TP_HD uint32_t h_len(H h) { return (uint32_t)(h >> 32) & 0xFFFFu; }
constexpr uint32_t MAX_DYN_LEN = 0xFFFFu;

// ---- attribute bits (same positions for interned flags[] words and dynamic handles) ---------------
enum : uint32_t {
    FL_BLANK = 1u << 0,        // string.IsNullOrWhiteSpace
    FL_VALID = 1u << 1,        // IsValidTag, Program.cs:1135-1139
    FL_DROP = 1u << 2,         // ShouldDropTag under the loaded configuration, Program.cs:1121-1133
    FL_LEN2 = 1u << 3,         // at least two UTF-16 code units, Program.cs:600
    FL_BUCKET_SHIFT = 4,       // ClassifyTag, Program.cs:1103-1111 (3 bits)
    FL_BUCKET_MASK = 7u << 4,
    FL_HAIR = 1u << 7,         // HairCoverageRegex, Program.cs:53
    FL_EYES = 1u << 8,         // EyeCoverageRegex, Program.cs:54
    FL_DYN_BITS = 0x1FEu,      // bits 1..8 travel inside a dynamic handle
    FL_BANNED = 1u << 9,       // BannedFinalTags, Program.cs:33-35
    FL_SAFE = 1u << 10,        // SafeModeRejectTags member, Program.cs:43-45
    FL_SAFE_IDX_SHIFT = 11,    // index in that list (5 bits)
    FL_KNOWN = 1u << 16,       // KnownTags member
    FL_EDGE = 1u << 17,        // NormalizeTag changes it, or it has implied tags: ExpandImpliedTags can act on it
    FL_NFEET = 1u << 18,       // NormalizeTag(it) == NormalizeTag("feet"), Program.cs:309, 719
    FL_NPENIS = 1u << 19,
    FL_NFOOTJOB = 1u << 20,
    FL_NEXT_BLANK = 1u << 21,  // NormalizeTag(it) is blank
    FL_PREFIX = 1u << 22,      // it is one of the tags SelectTags adds before the candidates (trigger, 1girl, always-add)
    FL_REACH2 = 1u << 23,      // the closure of {it} holds nothing beyond its at most two steps; Rec::s0_info / s1_info describe them
    FL_PHRASE = 1u << 25,      // holds one of the LooksBad phrases (Program.cs:56-58, 1144); only a force-allowed tag can and still be kept
    // BuildReviewRecord's tests on a raw token that counts as a subject token (in SubjectTokens, or holding "boys" / "girls"; Program.cs:829-837):
    FL_SUBJ_MALE = 1u << 26,   //   it is in MaleSubjectTokens (Program.cs:30-32, 835)
    FL_SUBJ_GIRL = 1u << 27,   //   it holds "girl" (Program.cs:836)
    FL_SUBJ_MULTI = 1u << 28,  //   it starts with "multiple_" or matches ^\d+(girl|girls|boy|boys)$ (Program.cs:837; \d is the Unicode category Nd)
    FL_SPLIT = 1u << 31,       // a force-allowed tag that holds ',' (or white space at a piece's end): the line ingest keeps it (Program.cs:1123) and its
                               //   join + re-split (Program.cs:604, 578) cuts it into the pieces DevTables::split_ids lists
    FL_FIX = 1u << 30,         // NormalizeTag leaves it alone and it is not blank
    FL_FENCE = 1u << 29,       // holds '`': StripCodeFences may act on a line with it (Program.cs:595, 1189-1210)
    FL_LEAF2 = 1u << 24,       // FL_REACH2, and each step is a non-blank string that NormalizeTag leaves alone (whatever a step's own
                               // visit appends is among these steps): one ExpandImpliedTags round settles it; Rec::s0_flags .. s1_len
                               // describe the steps
};
enum Bucket : uint32_t { B_IDENTITY = 0, B_FACE = 1, B_OUTFIT = 2, B_POSE = 3, B_BACKGROUND = 4, B_STYLE = 5, B_OTHER = 6 };
TP_HD uint32_t fl_bucket(uint32_t f) { return (f & FL_BUCKET_MASK) >> FL_BUCKET_SHIFT; }

TP_HD H h_make_dyn(uint32_t off, uint32_t len, uint32_t attr_bits, bool unstable) {
    return H_DYN | (unstable ? H_UNSTABLE : 0) | ((uint64_t)((attr_bits & FL_DYN_BITS) >> 1) << 48) | ((uint64_t)len << 32) | off;
}
TP_HD uint32_t h_dyn_flags(H h) { return ((uint32_t)(h >> 48) & 0xFFu) << 1; }

// ---- substring classifier (Aho-Corasick automaton over the literals of Program.cs:48-58) -------------
enum : uint32_t {
    CM_FACE = 1u << 0, CM_OUTFIT = 1u << 1, CM_POSE = 1u << 2, CM_BACKGROUND = 1u << 3, CM_STYLE = 1u << 4,
    CM_HAIR = 1u << 5, CM_EYES = 1u << 6,
    CM_SORRY = 1u << 7,      // the one DropTagsContains entry a normalised tag can hold
    CM_WATERMARK = 1u << 8,  // allowlist clause, Program.cs:1130
};
constexpr int DFA_SYMS = 28;  // a..z, '_', everything else
TP_HD uint32_t dfa_sym(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint32_t)(c - 'a') : (c == '_' ? 26u : 27u); }

// ---- byte classes (ASCII) ----------------------------------------------------------------------------
TP_HD bool is_ws8(uint8_t c) { return c == 0x20 || (c >= 0x09 && c <= 0x0D); }       // char.IsWhiteSpace below U+0080
TP_HD bool is_quote8(uint8_t c) { return c == '"' || c == '\'' || c == '`'; }         // Program.cs:1150
TP_HD bool is_punct8(uint8_t c) { return c == ';' || c == '.'; }                      // ASCII part of [;。、\.], Program.cs:1151
TP_HD uint8_t lower8(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }
TP_HD bool is_tag_first8(uint8_t c) { return (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9'); }   // ValidTagRegex, Program.cs:40
TP_HD bool is_tag_rest8(uint8_t c) {
    return is_tag_first8(c) || c == '_' || c == '(' || c == ')' || c == '!' || c == '\'' || c == '/' || c == '-';
}

// ---- text outside ASCII ---------------------------------------------------------------------------------
// char.IsWhiteSpace for a character of the Basic Multilingual Plane (also what the regex class \s matches)
TP_HD bool is_space16(uint32_t c) {
    return c == 0x20 || (c >= 0x09 && c <= 0x0D) || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 ||
           c == 0x202F || c == 0x205F || c == 0x3000;
}
constexpr uint8_t WS_STAND_IN = 0x0C;     // the ASCII white-space byte a white-space character outside ASCII is rewritten to (not ' ', which phrases hold)

// UTF-16 code units of valid UTF-8 text: every byte but the continuation bytes, and one more for each four-byte sequence.
TP_HD uint32_t utf16_units(const uint8_t* p, uint32_t n) {
    uint32_t u = 0;
    for (uint32_t i = 0; i < n; ++i) u += ((p[i] & 0xC0) != 0x80) + (p[i] >= 0xF0);
    return u;
}
// The end of text[i .. j) without its trailing run of [;。、\.] (Program.cs:1151; U+3002 and U+3001 are three bytes each).
TP_HD uint32_t strip_tail_punct(const uint8_t* s, uint32_t i, uint32_t j) {
    for (;;) {
        if (j > i && (s[j - 1] == ';' || s[j - 1] == '.')) --j;
        else if (j >= i + 3 && s[j - 3] == 0xE3 && s[j - 2] == 0x80 && (s[j - 1] == 0x82 || s[j - 1] == 0x81)) j -= 3;
        else return j;
    }
}

// Text with bytes outside ASCII, made ready for the byte-wise steps of the pipeline: src[0 .. n) must be well-formed UTF-8
// (shortest forms, no surrogates, at most U+10FFFF).  Every white-space character outside ASCII becomes WS_STAND_IN (one
// UTF-16 unit either way), every other character of the Basic Multilingual Plane becomes its ToLowerInvariant image,
// characters beyond it are copied (char-wise lower-casing leaves surrogates alone), "\r\n" becomes "\n" when drop_crlf.
// Returns the bytes written, or 0xFFFFFFFF when the text is not well-formed, does not fit `cap`, or - with refuse_twins -
// holds a character whose lower-case form shares its capital with another (the reference's ordinal-ignore-case sets would
// take two strings that differ in such characters for one; this code compares bytes).
template <class TT>
TP_HD uint32_t utf8_prepare(const TT& t, const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, bool drop_crlf, bool refuse_twins) {
    uint32_t i = 0, m = 0;
    while (i < n) {
        const uint8_t c = src[i];
        if (c < 0x80) {
            if (drop_crlf && c == '\r' && i + 1 < n && src[i + 1] == '\n') { ++i; continue; }
            if (m >= cap) return 0xFFFFFFFFu;
            dst[m++] = c; ++i;
            continue;
        }
        uint32_t len, cp;
        if (c >= 0xC2 && c <= 0xDF) { len = 2; cp = c & 0x1Fu; }
        else if (c >= 0xE0 && c <= 0xEF) { len = 3; cp = c & 0x0Fu; }
        else if (c >= 0xF0 && c <= 0xF4) { len = 4; cp = c & 0x07u; }
        else return 0xFFFFFFFFu;
        if (i + len > n) return 0xFFFFFFFFu;
        for (uint32_t k = 1; k < len; ++k) {
            const uint8_t d = src[i + k];
            if ((d & 0xC0) != 0x80) return 0xFFFFFFFFu;
            cp = (cp << 6) | (d & 0x3Fu);
        }
        if ((len == 3 && cp < 0x800) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF)) return 0xFFFFFFFFu;
        uint32_t out = cp;
        if (len < 4) {
            if (is_space16(cp)) out = WS_STAND_IN;
            else {
                uint32_t lo = 0, hi = t.n_case_map;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((t.case_map[mid] >> 16) < cp) lo = mid + 1; else hi = mid; }
                if (lo < t.n_case_map && (t.case_map[lo] >> 16) == cp) out = t.case_map[lo] & 0xFFFFu;
                if (refuse_twins && out >= 0x80) {
                    const uint32_t* tw = t.case_map + t.n_case_map;
                    lo = 0; hi = t.n_case_twin;
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (tw[mid] < out) lo = mid + 1; else hi = mid; }
                    if (lo < t.n_case_twin && tw[lo] == out) return 0xFFFFFFFFu;
                }
            }
        }
        const uint32_t olen = out < 0x80 ? 1u : out < 0x800 ? 2u : out < 0x10000 ? 3u : 4u;
        if (m + olen > cap) return 0xFFFFFFFFu;
        if (olen == 1) dst[m] = (uint8_t)out;
        else if (olen == 2) { dst[m] = (uint8_t)(0xC0 | (out >> 6)); dst[m + 1] = (uint8_t)(0x80 | (out & 0x3F)); }
        else if (olen == 3) { dst[m] = (uint8_t)(0xE0 | (out >> 12)); dst[m + 1] = (uint8_t)(0x80 | ((out >> 6) & 0x3F)); dst[m + 2] = (uint8_t)(0x80 | (out & 0x3F)); }
        else { dst[m] = src[i]; dst[m + 1] = src[i + 1]; dst[m + 2] = src[i + 2]; dst[m + 3] = src[i + 3]; }
        m += olen; i += len;
    }
    return m;
}

// ---- hashing (equality is always verified on bytes; the hash only places and pre-filters) ----------------
// The string is taken as little-endian 8-byte words (the last one zero-padded).  Word k is rotated by 13k bits and
// folded in with xor, so absent (zero) words cost nothing and the words can be folded in any order; two multiply /
// shift-xor rounds at the end spread the bits.
constexpr uint64_t HASH_SEED = 0xcbf29ce484222325ull;
TP_HD uint64_t rotl64(uint64_t x, uint32_t r) { r &= 63u; return r ? (x << r) | (x >> (64u - r)) : x; }
TP_HD uint64_t hash_word(uint64_t h, uint64_t w, uint32_t k) { return h ^ rotl64(w, 13u * k); }
TP_HD uint64_t hash_finish(uint64_t h, uint32_t n_bytes) {
    h ^= n_bytes;
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h ? h : 1;          // 0 marks an empty slot
}
struct Hasher {
    uint64_t h = HASH_SEED, w = 0;
    uint32_t n = 0;
    TP_HD void add(uint8_t c) {
        w |= (uint64_t)c << (8 * (n & 7));
        if ((++n & 7) == 0) { h = hash_word(h, w, (n >> 3) - 1); w = 0; }
    }
    TP_HD uint64_t done() const { return hash_finish((n & 7) ? hash_word(h, w, n >> 3) : h, n); }
};

// ---- compiled rules --------------------------------------------------------------------------------------
// Literal lists are [off, off+cnt) ranges of DevTables::pool holding indices into the per-file evidence bitset.
struct DRule { uint32_t parent_id, all_off, all_cnt, any_off, any_cnt, forb_off, forb_cnt, sub_off, sub_cnt; };  // ActionRule, Program.cs:129
struct DSub { uint32_t tag_id, any_off, any_cnt, all_off, all_cnt; };                                            // ActionSubtypeRule / FacetRule, :130-131
struct DFacet { uint32_t def_id, rule_off, rule_cnt; };                                                          // RequiredFacet, :132

// Per-string record for the wave-per-file kernel: everything one NormalizeTag hop needs, one cache line.
struct alignas(64) Rec {
    uint32_t flags, next, lit, len;
    uint32_t step_off, step_cnt, step0, step1;   // its ExpandImpliedTags step list (DevTables::step_ids), the first two entries inline
    uint32_t s0_info, s1_info;                   // with FL_REACH2: literal index (24 bits, 0xFFFFFF = none) | FL_NFEET-ness << 24 | FL_NPENIS-ness << 25 of step0 / step1
    uint32_t s0_flags, s0_len, s1_flags, s1_len; // with FL_LEAF2: flags and byte length of step0 / step1
    uint32_t len16;                              // length in UTF-16 code units (what string.Length counts)
    uint32_t pad[1];
};
// One slot of the wave-per-file kernel's lookup table, two cache lines: a normalised token (key, up to 32 bytes inline) ->
// the string NormalizeTag makes of it (val) and that string's record, so that a hit needs no second dependent load.
// The table has the slots of DevTables::ht_hash (same index = same entry).
struct alignas(128) Slot {
    uint32_t hash;       // high half of the entry's 64-bit hash
    uint32_t slen;       // key length in bytes; SLOT_EMPTY in an unused slot
    uint32_t val;        // id of the alias lookup's result
    uint32_t woff;       // the key's words in DevTables::str_words (keys longer than 32 bytes are compared there)
    uint32_t key[8];     // the key's first 32 bytes, zero padded
    uint32_t kflags;     // SLOT_KEY_BASIC
    uint32_t pad[3];
    Rec rec;             // DevTables::rec[val]
};
constexpr uint32_t SLOT_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t SLOT_KEY_BASIC = 1u;   // the key is what NormalizeTag's steps before the alias lookup make of itself (Program.cs:1150-1154)
// Hash-table payload: what the alias lookup makes of the string in this slot, and where its bytes are (to verify a probe).
struct HtInfo { uint32_t val, woff, slen, pad; };   // woff: first 8-byte word of the string in DevTables::str_words

// One slot of the lean stage's lookup table (tp_lean.h), half a cache line: a string of at most 32 bytes that the normalisation
// before the alias lookup leaves alone (key) -> what the stage needs of the string NormalizeTag makes of it (val).  Every key
// sits in one of the two slots its hash names (lean_slots()), so a lookup is one or two independent 16-byte loads, then the
// payload of the slot whose first half matched (and, from DevTables::lkey_hi, the second half of a long key).  The table is kept
// small (at most half full, 32 bytes a slot) so that it stays in the L2 cache of every XCD.
struct alignas(32) LeanSlot {
    uint32_t key_lo[4];  // the key's first 16 bytes, zero padded: most keys whole, so most lookups compare these alone
    uint32_t head;       // val | key length << 20 (0 in an unused slot) | min(step count, 3) << 26 | (NormalizeTag changes val) << 28
    uint32_t flags;      // of val
    uint32_t hi;         // keys longer than 16 bytes: which four words of DevTables::lkey_hi hold bytes 16 .. 31 (zero padded)
    uint32_t pad;
};
constexpr uint32_t LEAN_KEY_MAX = 32;
constexpr uint32_t LEAN_HEAD_ID_MASK = 0xFFFFFu;
constexpr uint32_t CREC_LIT_NONE = 0xFFFu;         // no literal, in DevTables::crec
constexpr uint32_t CREC_TOO_LONG = 1u << 31;       // in a record's third word: a length that a final-list word has no room for
// What a tag adds to the sums and tests over a file's final and selected lists, as one word per string (DevTables::fw_tix),
// so that a wavefront accumulates them with an OR and an add per list element instead of testing the flags one by one.
enum : uint32_t {
    FW_LEN_MASK = 0x3FFu,      // byte length (below 1023; the record carries CREC_TOO_LONG otherwise)
    FW_FACE = 1u << 10,        // ClassifyTag == "face"   (MissingCoreBuckets, Program.cs:1097-1101)
    FW_OUTFIT = 1u << 11,      // ClassifyTag == "outfit" (the same, and EvaluateCoverage, Program.cs:1056-1095)
    FW_SAFE = 1u << 12,        // SafeModeRejectTags member (Program.cs:942-945), its index in bits 13..17
    FW_SAFE_IDX_SHIFT = 13,
    FW_FOOTJOB = 1u << 18,     // the string NormalizeTag makes of "footjob" (Program.cs:309-311)
    FW_TRIGGER = 1u << 19,     // the trigger tag          (EvaluateCoverage's identity test)
    FW_UNKNOWN = 1u << 20,     // not in KnownTags (GetUnknownTags, Program.cs:817-824); the sum of a list's words under FW_SUM_MASK holds
    FW_SUM_MASK = FW_LEN_MASK | FW_UNKNOWN,   //   its caption bytes in bits 0..19 and its unknown tags from bit 20 (at most 128 tags below 1023 bytes)
    FW_GIRL = 1u << 21,        // "1girl" as normalised
    FW_HAIR = 1u << 22, FW_EYES = 1u << 23, FW_POSE = 1u << 24, FW_BACKGROUND = 1u << 25,
    FW_BLANK = 1u << 26,       // FL_BLANK
    FW_FIX = 1u << 27,         // FL_FIX
    FW_EDGE = 1u << 28,        // FL_EDGE
};
// Columns of list text in audit mode (see tp_pipeline.h) and the review record's bits (bit k < N_BANNED: banned tag k is among
// the normalised raw candidates, Program.cs:828; then the subject-token tests of Program.cs:835-837)
constexpr int AUDIT_KEPT = 0, AUDIT_EVIDENCE = 1, AUDIT_DROPPED = 2, AUDIT_TOP = 3, AUDIT_UNKNOWN = 4, AUDIT_COLS = 5;
constexpr uint32_t AUDIT_REASON_LEN = 12;   // ":invalid_tag" and ":policy_drop" have the same length (Program.cs:630-631)
constexpr uint32_t REVIEW_MALE = 1u << 16, REVIEW_GIRL = 1u << 17, REVIEW_MULTI = 1u << 18;
// One file's list entries of the audit row and the review record as the lean stage leaves them (interned ids): the evidence
// set in insertion order, the dropped candidates (bit 31: dropped by policy, else invalid), the first TOP_TAGS candidates.
// AUD_FLAGS: 1 when the final list is not the caption's list (a tag NormalizeTag changes); the final list is then at AUD_KEPT.
constexpr uint32_t AUD_EVIDENCE = 0, AUD_DROPPED = 192, AUD_TOP = 256, AUD_FLAGS = 287, AUD_KEPT = 288, AUD_STRIDE = 288 + 128;
constexpr uint32_t LEAN_TIX_MAX = 512;             // target indices the lean stage's direct first-occurrence table has room for
constexpr uint32_t TIX_NONE = 0xFFFu;
TP_HD uint32_t rotl32(uint32_t x, uint32_t r) { return (x << r) | (x >> (32u - r)); }   // 0 < r < 32
// The two slots of a key given as eight zero-padded little-endian words; the table has 1 << (32 - shift) slots.  Two
// accumulators with different rotations, so that keys whose words merely trade places do not share both slots; two multiply
// rounds each, and the slot from the top bits of the product (the low bits of a product see only the low bits of its
// factors: tag names that differ in a digit or two would crowd into few slots).
// `seed` is the table's (DevTables::lslot_seed): the builder takes the first one under which every key finds a place.
TP_HD void lean_slots(const uint32_t k[8], uint32_t n, uint32_t shift, uint32_t seed, uint32_t& s1, uint32_t& s2) {
    const uint32_t x = seed ^ k[0] ^ rotl32(k[1], 5) ^ rotl32(k[2], 10) ^ rotl32(k[3], 15) ^ rotl32(k[4], 20) ^ rotl32(k[5], 25) ^ rotl32(k[6], 30) ^ rotl32(k[7], 3) ^ (n << 26);
    const uint32_t y = rotl32(seed, 16) ^ k[0] ^ rotl32(k[1], 11) ^ rotl32(k[2], 22) ^ rotl32(k[3], 1) ^ rotl32(k[4], 12) ^ rotl32(k[5], 23) ^ rotl32(k[6], 2) ^ rotl32(k[7], 13) ^ n;
    uint32_t a = (x ^ (x >> 15)) * 2654435761u, b = (y ^ (y >> 13)) * 0x85EBCA77u;
    a = (a ^ (a >> 13)) * 0xC2B2AE35u;
    b = (b ^ (b >> 16)) * 0x27D4EB2Fu;
    s1 = a >> shift;
    s2 = b >> shift;
    if (s2 == s1) s2 = (s1 + 1) & (0xFFFFFFFFu >> shift);
}

constexpr uint32_t NO_ID = 0xFFFFFFFFu;
constexpr uint32_t EMPTY_ID = 0;          // the empty string is always id 0
constexpr int MAX_TOTAL_TAGS = 42;        // Program.cs:38
constexpr int N_BANNED = 9;               // BannedFinalTags, Program.cs:33-35
constexpr int MAX_SPLIT = 4;              // force-allowed tags the line ingest cuts into pieces (FL_SPLIT), and pieces of one
constexpr int MAX_SPLIT_PIECES = 8;
constexpr int TOP_TAGS = 20;              // TopJoyTags of a review record, Program.cs:860
static_assert(AUD_TOP + TOP_TAGS <= AUD_FLAGS, "the flag word lies behind the top tags");
constexpr int MAX_ALWAYS_ADD = 64;
constexpr int MAX_FALLBACK = MAX_TOTAL_TAGS;
constexpr uint32_t RC_REC_WORDS = 12;
constexpr uint32_t RC_FACET_WORDS = 8;
constexpr uint32_t FAST_RC_MAX_WORDS = 8192;   // 32 KB of LDS per block at most
constexpr uint32_t FAST_FRULES_MAX = 512;      // facet rules the wave-per-file kernel keeps one hit bit each for

// The table image as the kernels see it (all pointers are device pointers; on the host-side
// logic check they are host pointers).
struct DevTables {
    // interned strings
    uint32_t n_strings;
    uint32_t n_public;           // ids below this are the configuration's strings (what the tag histogram has a count for); ids from here on are
                                 // strings learned from earlier input (tp_api.cpp): counted with the tags in no table
    // Tokens in no table that the lean stage meets while such tags are kept (no strict vocabulary) are noted here, once per
    // string: a set of 64-bit hashes (0 = free) with the string's length (unk_ent) and a copy of its bytes (unk_text, 32 per slot)
    // beside each, so that nothing noted points into the caller's input once the run is over.  Null when nothing is to be noted.
    unsigned long long* unk_keys;
    unsigned long long* unk_ent;
    uint8_t* unk_text;
    uint32_t* unk_count;
    uint32_t unk_mask;           // slots - 1
    const uint8_t* str_bytes;
    const uint32_t* str_off;     // n_strings + 1
    const uint32_t* flags;       // FL_*
    const uint32_t* next;        // id of NormalizeTag(string)
    const uint32_t* alias1;      // id of the alias lookup alone (Program.cs:1155-1156) on the string as it stands
    const uint32_t* lit;         // evidence-bitset index of the string when a rule names it, else NO_ID
    const uint32_t* imp_off;     // n_strings + 1, ranges of imp_ids
    const uint32_t* imp_ids;     // NormalizeTag(implied) ids, blank ones removed (Program.cs:1171-1175)
    // open-addressing hash table over the strings' bytes
    const uint64_t* ht_hash;     // 0 = empty slot (a real hash of 0 is stored as 1)
    const uint32_t* ht_id;
    uint32_t ht_mask;
    // classifier automaton
    const uint16_t* dfa_next;    // [state * DFA_SYMS + sym]
    const uint16_t* dfa_out;     // CM_* reached in state
    // rules
    uint32_t n_lits, lit_words;
    const uint32_t* pool;
    const DRule* rules; uint32_t n_rules;
    const DSub* subs; uint32_t n_subs;
    const DFacet* facets; uint32_t n_facets;
    const DSub* facet_rules;
    uint32_t fired_words;
    // configuration
    uint32_t trigger_id;         // NormalizeTag(options.Trigger): what AddOne adds first (Program.cs:664, 693) and EvaluateCoverage looks for (:1058)
    uint32_t girl_add_id;        // NormalizeTag("1girl") (Program.cs:665)
    uint32_t girl_lit_id;        // "1girl" as EvaluateCoverage compares it (Program.cs:1059)
    uint32_t n_always; uint32_t always_ids[MAX_ALWAYS_ADD];   // what AddOne makes of each profile always-add tag (Program.cs:666-668)
    uint32_t n_fallback; uint32_t fallback_ids[MAX_FALLBACK]; // SelectTags of the minimal fallback line (Program.cs:298), computed at load
    uint32_t feet_n, penis_n, footjob_n;                       // NormalizeTag of the three probe tags (Program.cs:309-311)
    uint32_t drop_unknown;       // KnownTagDictionaryFound && ActiveStrictKnownTags: every string outside the tables is dropped (Program.cs:1131)
    uint32_t allow_nonempty;     // AllowlistTags.Count > 0 (Program.cs:1130)
    uint32_t known_active;       // KnownTagDictionaryFound && KnownTags.Count > 0 (Program.cs:818)
    uint32_t safe_mode;
    double line_weight;
    uint32_t line_source;
    uint32_t newline_len; uint8_t newline[4];
    // extra tables of the wave-per-file kernel
    const Rec* rec;              // n_strings
    const Slot* slots;           // ht_mask + 1
    const HtInfo* ht_info;       // per hash slot
    const uint64_t* str_words;   // every string again as zero-padded little-endian 8-byte words, for word-wise probe verification
    const LeanSlot* lslots;      // the lean stage's lookup table (null when it could not be built)
    uint32_t lslot_shift, lslot_seed;   // 1 << (32 - lslot_shift) slots
    const uint32_t* lkey_hi;     // second halves of the lean table's long keys, four words each (LeanSlot::hi)
    const uint32_t* case_map;    // n_case_map entries sorted by character: character << 16 | what ToLowerInvariant makes of it, for every character outside
    uint32_t n_case_map;         //   ASCII that it changes; then n_case_twin lower-case characters, sorted, that share their capital with another
    uint32_t n_case_twin;        //   lower-case character (ordinal-ignore-case comparison takes the two for one, byte comparison does not)
    uint32_t tables_hold_twins;  // some table string holds such a character
    const uint32_t* emit_rec;    // per id, four words, for the caption writer: a string of up to 15 bytes itself (zero-padded) with its length in byte 15;
                                 //   a longer one as {offset into str_bytes, byte length, 0, 0xFF << 24}
    const uint32_t* emit_rec32;  // per id, eight words, for the caption writer: a string of up to 31 bytes itself (zero-padded) with its length in byte 31;
                                 //   a longer one as {offset into str_bytes, byte length, 0 ..., 0xFF << 24}
    const uint32_t* attr2;       // per id: flags, byte length (two words)
    const uint32_t* crec;        // per id, eight words, for the lean stage's back half: flags; its final-list word (FW_*); Rec::lit (12 bits,
                                 // CREC_LIT_NONE = none) | its target index << 12 (12 bits, TIX_NONE = none) | min(step count, 3) << 24 | CREC_TOO_LONG;
                                 // Rec::s0_info | s1_info << 16, each as literal index (12 bits), feet-ness << 12, penis-ness << 13; step0 and step1,
                                 // each as id | target index << 20, or NO_ID; the final-list words of step0 and step1 (0 when absent)
                                 // (steps: the first two entries of the string's ExpandImpliedTags step list; absent without FL_EDGE)
    const uint32_t* fw_tix;      // per id, two words: the final-list word (FW_*), the target index or TIX_NONE
    uint32_t n_tix;              // target indices in use: one per string that is a step of a string NormalizeTag leaves alone, or a facet's default
                                 // or rule tag (what can occur twice in the final list of a file of the lean stage); above LEAN_TIX_MAX none are given out
    const uint32_t* step_off;    // n_strings + 1: what one ExpandImpliedTags visit of the string appends, in order:
    const uint32_t* step_ids;    //   NormalizeTag(it), then the implied tags of that (Program.cs:1164-1175); empty when nothing can change
    const uint32_t* reach_off;   // n_strings + 1: every other string the closure of {it} holds, as pairs
    const uint32_t* reach_ent;   //   {id, literal index (24 bits, 0xFFFFFF = none) | FL_NFEET-ness << 24 | FL_NPENIS-ness << 25 | round << 26}, in the order
                                 //   ExpandImpliedTags appends them to {it} (round = the loop round that does, from 1)
    uint32_t closure_tables;     // 0 when the reach table would be too large: files that need a closure take the general kernel
    uint32_t reach_levels_ok;    // every reach entry's round fits the six bits its info word has for it
    uint32_t fast_ok;            // the configuration fits the wave-per-file kernel at all
    uint32_t n_prefix; uint32_t prefix_ids[MAX_TOTAL_TAGS];      // result list after the AddOne calls of Program.cs:664-668
    uint32_t n_fb_final; uint32_t fb_final[MAX_FALLBACK];        // normalizedFinalSet of the fallback selection (Program.cs:298, 303)
    uint32_t final_cap;          // capacity of the per-file final-tag list
    uint32_t evidence_grow;      // evidence list capacity = candidates + this
    // Line ingest scores fall with the candidate index (Program.cs:581), so score tests are index tests:
    // Rule cache of the wave-per-file kernel (copied to LDS by every block): action rules, subtypes and facet rules as
    // RC_REC_WORDS-word records {tag id, first subtype (rules), all_cnt | any_cnt << 8 | forbidden_cnt << 16,
    // literal list offset in the cache (15 bits) | 0x8000 when NormalizeTag does not leave the tag alone | subtype count << 16,
    // four literals inline (all, any, forbidden in this order; the offset is used when there are more), the tag's flags, its
    // byte length, its final-list word, its target index}, then facets as RC_FACET_WORDS-word records {default id, first facet
    // rule, rule count, the default's flags, its byte length, 1 when NormalizeTag does not leave the default alone, its final-list
    // word, its target index}, then the longer literal lists.
    const uint32_t* rc;
    uint32_t rc_words, rc_subs, rc_frules, rc_facets, n_frules;
    // Literal masks in the cache (lit_words words each): the literals action rules (facet rules) test positively.  A file whose
    // evidence holds none of them cannot fire a rule, unless some rule has no positive literal at all (the *_vacuous flags).
    uint32_t rc_rule_mask, rc_frule_mask, rules_vacuous, frules_vacuous;
    uint32_t prefix_simple;      // the same for the prefix tags, and none of them has steps
    uint32_t prefix_flags[MAX_TOTAL_TAGS], prefix_len[MAX_TOTAL_TAGS];   // flags and byte length per prefix tag
    uint32_t banned_ids[N_BANNED];   // BannedFinalTags (Program.cs:33-35) in ordinal-ignore-case order, the order BuildReviewRecord lists them in (Program.cs:828)
    // FL_SPLIT strings: split_src[k] is cut into the strings split_ids[k][0 .. split_cnt[k]) (SplitTags of it, as they stand: the
    // candidates' NormalizeTag sites see them next); split_commas[k] = the ',' it holds (LooksBad of the joined line counts them);
    // split_grow = how many candidates one token can become at most (1 without such strings)
    uint32_t n_split, split_grow;
    uint32_t split_src[MAX_SPLIT], split_cnt[MAX_SPLIT], split_commas[MAX_SPLIT], split_ids[MAX_SPLIT][MAX_SPLIT_PIECES];
    uint32_t seed_ci_limit;      // candidates below this index have !(score < 0.28)  (BuildEvidenceTags, Program.cs:724)
    uint32_t style_ci_limit;     // candidates below this index have score >= 0.60     (DetermineStyleQualityCap, Program.cs:706)
};

TP_HD int source_rank(uint32_t s) { return s == 0 ? 0 : s == 1 ? 1 : s == 2 ? 2 : 3; }   // Program.cs:712-717

}  // namespace tp
