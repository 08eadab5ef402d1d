// Lean wave-per-file stage of the tag pipeline: the same reference span as tp_fast.h (Program.cs:223-229, 292-329,
// 574-605), restricted to the shape most tag files have, and written for a low instruction count per file (the wave
// kernels are bound by vector-instruction issue, not by memory).  A file it does not take is declined (return false)
// and goes through the stages of tp_fast.h, which restate the same span in general.
//
// What it takes:
//   * a line of at most LEAN_LINE_MAX bytes, all below 0x80, with at least two commas and at most 64 tokens of at most
//     32 bytes each, every token with at most one white-space byte at either end;
//   * every token, as written, either an interned string that the normalisation before the alias lookup leaves alone
//     (Program.cs:1150-1154; then NormalizeTag of the token is the alias lookup's result, which the lookup slot holds) and
//     that holds neither a LooksBad phrase nor '`', or - under strict vocabulary only - a string in no table made of
//     [a-z0-9_()!/-] without "__" and without "sorry" (dropped whatever else it is, Program.cs:1131);
//   * every kept tag one that NormalizeTag leaves alone, every selected tag's implied tags settled leaves, every
//     evidence seed's closure held by its record (the "settled" case of tp_fast.h): nothing is looked up again;
//   * no action rule that fires, at least 3 selected tags, no
//     prefix tag or facet tag that NormalizeTag changes.
//
// How it differs from tp_fast.h in method:
//   * the line is taken 16 bytes per lane straight from the global load (one pass for a kilobyte), commas are found by
//     byte-parallel arithmetic on the four words, tokens are numbered by a wave prefix sum of the per-lane comma counts;
//   * only separators are classified: a token's bytes are hashed as they stand, and the table tells whether they are a
//     normalised string;
//   * "first occurrence of an id" (Distinct, the ordered growth of the final set) is decided by min-order tables: every
//     element offers order << 20 | id to the table slots its id hashes to with an LDS minimum; an element whose id owns
//     a slot learns the lowest order of its id from it; elements that lost both slots go another round with another hash;
//   * the final set's three growth steps (implied tags, action tags, facet tags: Program.cs:304-307) are decided in ONE
//     such pass, since the facet choices depend on the evidence only; tag ids go straight to their final positions in
//     the output row, no list is kept in LDS.
#pragma once
#include "tp_fast.h"

// Host-side tests may count why files are declined (reason = position of the test in lean_file).
#if defined(TP_LEAN_STATS) && !defined(__HIP_DEVICE_COMPILE__)
extern "C" unsigned long long g_lean_declines[32];
#define LEAN_NO(k) ((void)(w.lane() == 0 ? ++g_lean_declines[k] : 0ull), false)
#else
#define LEAN_NO(k) false
#endif

// Diagnostic builds stop the lean stage after section TP_LEAN_STOP (the value handed over keeps the section's results alive).
#ifndef TP_LEAN_STOP
#define TP_LEAN_STOP (-1)
#endif
#define LEAN_STOP(k, expr)                                                                                   \
    if (TP_LEAN_STOP == (k)) {                                                                               \
        fin32[lane] = (uint32_t)(expr);                                                                      \
        out.verdict = tp_verdict{};                                                                          \
        out.n_final = 0;                                                                                     \
        out.caption_len = 0;                                                                                 \
        return true;                                                                                         \
    }

namespace tp {

constexpr uint32_t LEAN_LINE_MAX = 1008;      // with the start's 16-byte phase at most 64 pieces of 16 bytes: one per lane
#ifndef LEAN_BACK_TAB_BITS
#define LEAN_BACK_TAB_BITS 8
#endif
constexpr uint32_t LEAN_ID_BITS = 20;         // table values are order << 20 | id
constexpr uint32_t LEAN_ID_MASK = (1u << LEAN_ID_BITS) - 1;
constexpr uint32_t LEAN_DEAD = 0xFFFFFFFFu;   // no element
constexpr uint32_t LEAN_ROUNDS = 5;
constexpr uint32_t LEAN_FACETS_MAX = 64;      // one facet per lane
constexpr uint32_t LEAN_LIT_WORDS = 64;       // evidence bitset words (2048 rule literals)

// What the back half needs to know about a candidate is its string's record in DevTables::crec (eight words).  The records of
// a file's candidates are in LDS as two planes of 16 bytes per lane (LeanLds::cr), put there by whoever calls lean_back -
// the GPU kernel by loads that write LDS directly, one file ahead - and stay there until the selected tags are on the
// lanes by position.
struct LeanCr { uint32_t w[4]; };
// Tags one level further down (steps of a selected tag's steps), per position: two per lane.
struct LeanDeep { uint32_t val, fw, pad0, pad1; };

// The front half's LDS state per wavefront (line scan, token lookup, Distinct) ...
constexpr uint32_t LEAN_CPOS_NOTE = 67;
struct alignas(16) LeanFrontLds {
    uint8_t line[16 + 1024 + 48];             // 16 spare bytes in front, 64 pieces, slack for 32-byte token reads
    uint32_t cpos[68];                        // cpos[1 + k]: position of comma k in the text; cpos[0] = -1, cpos[T] = len;
                                              // cpos[LEAN_CPOS_NOTE]: set when the file was declined over plain tokens in no table
    static constexpr uint32_t TAB_BITS = 8, TAB = 1u << TAB_BITS;   // slots per min-order table
    uint32_t tab[2 * TAB];                    // two min-order tables
};
// ... and the back half's (everything from the candidates on).
struct alignas(16) LeanLds {
    union {
        LeanCr cr[2][64];                     // cr[h][l]: words 4h .. 4h + 3 of candidate l's record, until the selected tags are on the lanes by position
        LeanDeep deep[2][64];                 // after that
    };
    uint32_t ids[64];                         // the candidates' ids
    static constexpr uint32_t TAB_BITS = LEAN_BACK_TAB_BITS, TAB = 1u << TAB_BITS;   // slots per min-order table
    uint32_t tab[2 * TAB];                    // two min-order tables
    uint32_t ebits[LEAN_LIT_WORDS];           // evidence bitset over the rule literals
    uint32_t fhit[FAST_FRULES_MAX / 32];      // bit p: facet rule p matches the evidence
    uint32_t inv[64];                         // selected position -> candidate lane
    uint32_t act[64][4];                      // the tags the action rules offer, in rule order: val, final-list word
    uint32_t scratch[4];
};
// Per block: masks that cut a 32-byte read down to a token's length; the tags SelectTags adds before the candidates.
struct alignas(16) LeanShared {
    uint64_t tail_mask[33][4];
    uint32_t prefix_ids[MAX_TOTAL_TAGS], prefix_flags[MAX_TOTAL_TAGS], prefix_fw[MAX_TOTAL_TAGS], prefix_tix[MAX_TOTAL_TAGS];
};
template <class TT>
TP_HD_FLAT void lean_shared_init(const TT& t, LeanShared& S, uint32_t tid, uint32_t nthreads) {
    for (uint32_t k = tid; k < 33 * 4; k += nthreads) S.tail_mask[k >> 2][k & 3] = word_mask(k >> 2, k & 3);
    for (uint32_t k = tid; k < (uint32_t)MAX_TOTAL_TAGS; k += nthreads) {
        S.prefix_ids[k] = k < t.n_prefix ? t.prefix_ids[k] : NO_ID;
        S.prefix_flags[k] = k < t.n_prefix ? t.prefix_flags[k] : 0;
        S.prefix_fw[k] = k < t.n_prefix ? t.fw_tix[2 * (size_t)t.prefix_ids[k]] : 0;
        S.prefix_tix[k] = k < t.n_prefix ? t.fw_tix[2 * (size_t)t.prefix_ids[k] + 1] : TIX_NONE;
    }
}
// The tables fit this stage at all (checked once on the host).
inline bool lean_tables_ok(const DevTables& t) {
    for (uint32_t k = 0; k < t.n_prefix && k < (uint32_t)MAX_TOTAL_TAGS; ++k) if (t.prefix_len[k] >= FW_LEN_MASK) return false;   // no room in a final-list word
    return t.fast_ok && t.prefix_simple && t.lslots != nullptr && t.n_strings <= LEAN_ID_MASK && t.n_facets <= LEAN_FACETS_MAX && t.n_prefix <= 8 && !t.rules_vacuous &&
           t.lit_words <= LEAN_LIT_WORDS;
}

// Bit k of the result: byte k of x differs from the byte pat repeats (exact for any bytes).
TP_HD_FLAT uint32_t ne_nibble(uint32_t x, uint32_t pat) {
    const uint32_t z = x ^ pat;
    uint32_t t = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;   // 0x80 where the byte differs
    t |= t << 7;
    t |= t << 14;
    return t >> 28;
}
TP_HD_FLAT uint32_t ws1(uint32_t c) { return (uint32_t)(c == 0x20u) | (uint32_t)(c - 9u <= 4u); }   // char.IsWhiteSpace below U+0080
TP_HD_FLAT uint32_t lean_val(uint32_t order, uint32_t id, bool live) { return live ? (order << LEAN_ID_BITS) | id : LEAN_DEAD; }

// First occurrences by min-order tables.  val[e] = order << 20 | id describes element e of this lane (orders are distinct
// and below 2^12, ids below 2^20), LEAN_DEAD means no element.  Bit e of the result is set for the element with the lowest
// order among those with its id; bit 31 is set when elements are left undecided after LEAN_ROUNDS rounds (wave-uniform).
template <int NE, class W, class LDS>
TP_HD_FLAT __attribute__((always_inline)) uint32_t lean_first(W& w, LDS& L, const uint32_t (&val)[NE]) {
    uint32_t pend = 0, first = 0;
    TP_UNROLL for (int e = 0; e < NE; ++e) pend |= (uint32_t)(val[e] != LEAN_DEAD) << e;
    for (uint32_t round = 0; round < LEAN_ROUNDS; ++round) {
        static_assert((2 * LDS::TAB) % 256 == 0, "whole 16-byte stores per lane");
        TP_UNROLL for (uint32_t k = 0; k < 2 * LDS::TAB / 256; ++k) w.fill16(&L.tab[256 * k + 4 * w.lane()], 0xFFFFFFFFu);
        w.sync();
        const uint32_t K = round == 0 ? 2654435761u : (round == 1 ? 0x85EBCA77u : (round == 2 ? 0xC2B2AE3Du : (round == 3 ? 0x27D4EB2Fu : 0x165667B1u)));
        uint32_t p[NE];
        TP_UNROLL for (int e = 0; e < NE; ++e) {
            p[e] = (val[e] & LEAN_ID_MASK) * K;
            if ((pend >> e) & 1u) {
                w.lds_min(&L.tab[p[e] >> (32 - LDS::TAB_BITS)], val[e]);
                w.lds_min(&L.tab[LDS::TAB + ((p[e] >> (32 - 2 * LDS::TAB_BITS)) & (LDS::TAB - 1))], val[e]);
            }
        }
        w.sync();
        TP_UNROLL for (int e = 0; e < NE; ++e) {
            if ((pend >> e) & 1u) {
                const uint32_t va = L.tab[p[e] >> (32 - LDS::TAB_BITS)], vb = L.tab[LDS::TAB + ((p[e] >> (32 - 2 * LDS::TAB_BITS)) & (LDS::TAB - 1))];
                const bool ma = ((va ^ val[e]) & LEAN_ID_MASK) == 0, mb = ((vb ^ val[e]) & LEAN_ID_MASK) == 0;
                if (ma || mb) { first |= (uint32_t)((ma ? va : vb) == val[e]) << e; pend &= ~(1u << e); }
            }
        }
        if (!w.ballot(pend != 0)) return first;
        w.sync();                                                  // every lane has read the tables before they are cleared again
    }
    return first | 0x80000000u;
}

// A token in no table, dropped under strict vocabulary whatever else it is (Program.cs:1131), provided normalisation
// leaves its bytes alone and the line scan's phrase test (LooksBad, Program.cs:1141-1147) cannot hit inside it.
TP_HD_FLAT bool lean_plain_unknown(const uint8_t* p, uint32_t n) {
    uint32_t prev = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t c = p[k];
        const bool ok = alnum32(c) || c == '_' || c == '(' || c == ')' || c == '!' || c == '/' || c == '-';
        if (!ok || (c == '_' && prev == '_')) return false;
        if (c == 's' && k + 5 <= n && p[k + 1] == 'o' && p[k + 2] == 'r' && p[k + 3] == 'r' && p[k + 4] == 'y') return false;
        prev = c;
    }
    return true;
}

// A token in no table that this stage has to decline because such tags are kept (no strict vocabulary): noted once per
// string in DevTables::unk_keys (64-bit hash, 0 = free; linear probing) with a copy of its bytes, so that the host can add it
// to the tables for later runs.  p: the token in the input, n <= 32 bytes; the input is not looked at again after the run.
template <class TT>
TP_HD_FLAT void lean_note_unknown(const TT& t, const uint8_t* p, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t h1 = 2166136261u, h2 = 0x9E3779B9u ^ n;               // two byte-wise 32-bit hashes
    for (uint32_t k = 0; k < n; ++k) { const uint32_t c = p[k]; h1 = (h1 ^ c) * 16777619u; h2 = (h2 + c) * 0x85EBCA77u; h2 ^= h2 >> 15; }
    const unsigned long long h = ((unsigned long long)h1 << 32 | h2) | 1ull;
    uint32_t slot = (h1 ^ (h2 >> 7)) & t.unk_mask;
    for (uint32_t probe = 0; probe < 32; ++probe) {
        const unsigned long long old = atomicCAS(&t.unk_keys[slot], 0ull, h);
        if (old == 0ull) {
            uint8_t* o = t.unk_text + 32 * (size_t)slot;
            for (uint32_t k = 0; k < n; ++k) o[k] = p[k];
            t.unk_ent[slot] = (unsigned long long)n;
            atomicAdd(t.unk_count, 1u);
            return;
        }
        if (old == h) return;
        slot = (slot + 1) & t.unk_mask;
    }
#else
    (void)t; (void)p; (void)n;
#endif
}
// The plain tokens of a file that lean_front has just declined over such tokens (L still holds its line and commas),
// noted whether a table holds them or not: the host leaves out what it knows.  A pass of its own, for declined files
// only, so that the front half itself carries none of this.
template <class W, class TT>
TP_HD_FLAT void lean_note_tokens(W& w, const TT& t, LeanFrontLds& L, const uint8_t* text, uint32_t len) {
    const uint32_t lane = w.lane();
    const uint32_t shift = (uint32_t)((uintptr_t)text & 15);
    const uint8_t* s = L.line + 16 + shift;
    const uint64_t at = w.ballot(L.cpos[lane + 1] == len);         // cpos[T] = len (commas lie below len), 1 <= T <= 65
    const uint32_t T = at ? ctz64(at) + 1 : (L.cpos[65] == len ? 65u : 0u);
    if (lane < T) {
        uint32_t i = L.cpos[lane] + 1, j = L.cpos[lane + 1];
        if (i < j && ws1(s[i])) ++i;
        if (j > i && ws1(s[j - 1])) --j;
        const uint32_t n = j - i;
        if (n >= 1 && n <= 32 && lean_plain_unknown(s + i, n)) lean_note_unknown(t, text + i, n);
    }
}

// One record of the rule cache against the evidence bitset (MatchesFacetRule and its kin, Program.cs:777-794).
TP_HD_FLAT bool lean_rc_match(const uint32_t* ebits, const uint32_t* rc, uint32_t rec) {
    const uint32_t* r = rc + rec;
    const uint32_t cnts = r[2], n_all = cnts & 255u, n_any = (cnts >> 8) & 255u, n_forb = (cnts >> 16) & 255u, tot = n_all + n_any + n_forb;
    const uint32_t* lits = tot <= 4 ? r + 4 : rc + (r[3] & 0x7FFFu);
    bool ok_all = true, ok_any = n_any == 0, ok_forb = true;
    for (uint32_t i = 0; i < tot; ++i) {
        const uint32_t li = lits[i];
        const bool bit = (ebits[li >> 5] >> (li & 31)) & 1u;
        if (i < n_all) ok_all = ok_all && bit;
        else if (i < n_all + n_any) ok_any = ok_any || bit;
        else ok_forb = ok_forb && !bit;
    }
    return ok_all && ok_any && ok_forb;
}

// The rare parts of the pipeline are functions of their own (measured: as real calls they cost the usual path more, in
// saved and restored registers, than they take off its register count; LEAN_X_NOINLINE builds them as calls).
#if defined(__HIP_DEVICE_COMPILE__) && defined(LEAN_X_NOINLINE)
#define TP_NOINLINE __attribute__((noinline))
#else
#define TP_NOINLINE
#endif

// What the final list's elements add up to: the OR of their final-list words (FW_*: bucket bits of MissingCoreBuckets,
// Program.cs:1097-1101, safe-mode and footjob membership), caption bytes (low 20 bits of sum) and tags outside the
// vocabulary (from bit 20).
struct LeanAcc { uint32_t sum, bits; };
constexpr uint32_t LA_UNKNOWN_SHIFT = 20;
static_assert(FW_UNKNOWN == 1u << LA_UNKNOWN_SHIFT, "a word's unknown bit sits where the sum counts unknown tags");
TP_HD_FLAT void lean_emit(LeanAcc& a, uint32_t* fin32, bool first, uint32_t val, uint32_t fw, uint32_t at) {
    if (first) {
        a.bits |= fw;
        a.sum += fw & FW_SUM_MASK;
        fin32[at] = val & LEAN_ID_MASK;
    }
}
// The first safe-mode hit of the final list (Program.cs:942-945) as position << 8 | index in SafeModeRejectTags; looked for
// only in a file whose list holds such a tag at all.
TP_HD_FLAT uint32_t lean_safe_key(bool first, uint32_t fw, uint32_t at) { return first && (fw & FW_SAFE) ? (at << 8) | ((fw >> FW_SAFE_IDX_SHIFT) & 31u) : 0xFFFFFFFFu; }
TP_HD_FLAT uint32_t min32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// InferActionTags (Program.cs:731-753) for a file whose evidence holds a literal some rule tests positively: the tags of
// the rules and subtypes that match go to L.act in rule order, their bits to `fired` (when asked for).  Returns how many
// tags there are, LEAN_DEAD when the file is not for this stage.  A tag NormalizeTag changes stays in the final list as it is
// (Program.cs:738, 745) and is normalised when the caption is put together (Program.cs:972): with CHANGING such tags are
// allowed and bit q of `chg` says that tag q is one (lean_caption_fix deals with them); without, the file is declined.
template <bool CHANGING, class W, class TT>
TP_HD_FLAT TP_NOINLINE uint32_t lean_actions(W& w, const TT& t, LeanLds& L, const uint32_t* rc, uint32_t* fired, uint64_t& chg) {
    const uint32_t lane = w.lane();
    uint32_t* firedw = L.tab;                                       // the table area is idle here
    static_assert(FAST_FIRED_WORDS <= 2 * LeanLds::TAB, "the fired bitmap fits the table area");
    if (lane < t.fired_words) firedw[lane] = 0;
    w.sync();
    uint32_t n_act = 0;
    bool changes = false;
    for (uint32_t r0 = 0; r0 < t.n_rules; r0 += 64) {
        const uint32_t ri = r0 + lane;
        const bool hit = ri < t.n_rules && lean_rc_match(L.ebits, rc, ri * RC_REC_WORDS);   // MatchesActionRule, Program.cs:777-782
        uint64_t m = w.ballot(hit);
        if (hit) w.lds_or(&firedw[ri >> 5], 1u << (ri & 31));                              // "action:{parent}" (Program.cs:741)
        while (m) {
            const uint32_t* rr = rc + (r0 + ctz64(m)) * RC_REC_WORDS;
            m &= m - 1;
            const uint32_t sub_off = rr[1], sub_cnt = rr[3] >> 16;
            if (n_act >= 64) return LEAN_DEAD;
            if ((rr[10] & FW_LEN_MASK) == FW_LEN_MASK || (rr[8] & FL_BLANK)) changes = true;   // too long for its final-list word, or blank
            else if (rr[3] & 0x8000u) { if (CHANGING) chg |= 1ull << n_act; else changes = true; }   // a tag NormalizeTag changes
            if (lane == 0) { L.act[n_act][0] = lean_val(640 + n_act, rr[0], true); L.act[n_act][1] = rr[10]; }
            ++n_act;
            for (uint32_t s0 = 0; s0 < sub_cnt; s0 += 64) {
                const uint32_t si = s0 + lane;
                const uint32_t srec = t.rc_subs + (sub_off + (si < sub_cnt ? si : 0)) * RC_REC_WORDS;
                const bool shit = si < sub_cnt && lean_rc_match(L.ebits, rc, srec);        // MatchesSubtypeRule, Program.cs:784-788
                const uint64_t ms = w.ballot(shit);
                if (n_act + popc64(ms) > 64) return LEAN_DEAD;
                if (shit) {
                    const uint32_t bit = t.n_rules + sub_off + si;                           // "subtype:{parent}:{tag}" (Program.cs:748)
                    w.lds_or(&firedw[bit >> 5], 1u << (bit & 31));
                    const uint32_t* sr = rc + srec;
                    if ((sr[10] & FW_LEN_MASK) == FW_LEN_MASK || (sr[8] & FL_BLANK) || (!CHANGING && (sr[3] & 0x8000u))) changes = true;
                    const uint32_t q = n_act + w.rank(ms);
                    L.act[q][0] = lean_val(640 + q, sr[0], true); L.act[q][1] = sr[10];
                }
                if (CHANGING) {
                    uint64_t mc = w.ballot(shit && (rc[srec + 3] & 0x8000u) != 0);
                    while (mc) { const uint32_t l = ctz64(mc); mc &= mc - 1; chg |= 1ull << (n_act + popc64(ms & ((1ull << l) - 1))); }
                }
                n_act += popc64(ms);
            }
        }
    }
    if (w.ballot(changes) != 0) return LEAN_DEAD;
    w.sync();
    if (fired && lane < t.fired_words) fired[lane] = firedw[lane];
    w.sync();                                                       // the table area is read before its next use clears it
    return n_act;
}

// A selected tag whose steps lead further (no FL_LEAF2): a step that is itself settled (FL_LEAF2) names the tags its visit
// appends in the next round of ExpandImpliedTags (Program.cs:1160-1179); at most two steps, at most two of those tags per
// selected tag, and nothing beyond them, or the file is not for this stage (returns true).  The tags go to L.deep.
template <class W, class TT>
TP_HD_FLAT TP_NOINLINE bool lean_deep(W& w, const TT& t, LeanLds& L, bool deep, uint32_t k_cnt, uint32_t kv0, uint32_t kv1, uint32_t kw0, uint32_t kw1) {
    const uint32_t lane = w.lane();
    w.sync();                                                       // the candidates' records have been read: their place is taken by the deeper tags
    bool bad_deep = deep && k_cnt > 2;
    LeanDeep g0 = {LEAN_DEAD, 0, 0, 0}, g1 = {LEAN_DEAD, 0, 0, 0};
    if (deep && !bad_deep) {
        uint32_t ng = 0;
        for (uint32_t c = 0; c < 2; ++c) {
            const uint32_t kv = c ? kv1 : kv0, kw = c ? kw1 : kw0;
            if (kv == LEAN_DEAD) continue;
            if (!(kw & FW_FIX)) { bad_deep = true; continue; }     // a step NormalizeTag changes
            if (!(kw & FW_EDGE)) continue;
            const Rec rk = t.rec[kv & LEAN_ID_MASK];
            if (!(rk.flags & FL_LEAF2) || ng + rk.step_cnt > 2) { bad_deep = true; continue; }
            for (uint32_t j = 0; j < rk.step_cnt; ++j) {
                LeanDeep g;
                const uint32_t gid = j == 0 ? rk.step0 : rk.step1;
                g.val = lean_val(320 + 4 * lane + 2 * c + j, gid, true);
                g.fw = t.fw_tix[2 * (size_t)gid]; g.pad0 = g.pad1 = 0;
                if ((g.fw & FW_LEN_MASK) == FW_LEN_MASK) bad_deep = true;   // a length the word has no room for
                if (ng == 0) g0 = g; else g1 = g;
                ++ng;
            }
        }
    }
    L.deep[0][lane] = g0; L.deep[1][lane] = g1;
    w.sync();
    return w.ballot(bad_deep) != 0;
}

// The first-occurrence pass of a file with deeper tags or action tags: seven elements per lane.
struct LeanWide { uint32_t fb, grown2, nact; };
template <class W>
TP_HD_FLAT TP_NOINLINE LeanWide lean_first_wide(W& w, LeanLds& L, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, bool any_deep, uint32_t n_act) {
    const uint32_t lane = w.lane();
    const uint32_t v7[7] = {v0, v1, v2, v3, any_deep ? L.deep[0][lane].val : LEAN_DEAD, any_deep ? L.deep[1][lane].val : LEAN_DEAD, lane < n_act ? L.act[lane][0] : LEAN_DEAD};
    LeanWide r;
    r.fb = lean_first<7>(w, L, v7);
    r.grown2 = popc64(w.ballot(r.fb & 16u)) + popc64(w.ballot(r.fb & 32u));
    r.nact = popc64(w.ballot(r.fb & 64u));
    return r;
}
// ... and what its deeper tags (from position base_g on) and action tags (from base_a on) add to the final list.
template <class W, class TT>
TP_HD_FLAT TP_NOINLINE LeanAcc lean_emit_wide(W& w, const TT& t, LeanLds& L, uint32_t* fin32, uint32_t fb, uint32_t base_g, uint32_t base_a, bool any_deep, uint32_t n_act,
                                         uint32_t& safe_key) {
    const uint32_t lane = w.lane();
    LeanAcc a = {0, 0};
    safe_key = 0xFFFFFFFFu;
    const uint64_t mg0 = w.ballot(fb & 16u), mg1 = w.ballot(fb & 32u), ma = w.ballot(fb & 64u);
    if (any_deep) {
        const LeanDeep g0 = L.deep[0][lane], g1 = L.deep[1][lane];
        const uint32_t pos_g0 = base_g + w.rank(mg0) + w.rank(mg1);
        lean_emit(a, fin32, (fb & 16u) != 0, g0.val, g0.fw, pos_g0);
        lean_emit(a, fin32, (fb & 32u) != 0, g1.val, g1.fw, pos_g0 + ((fb >> 4) & 1u));
        safe_key = min32(lean_safe_key((fb & 16u) != 0, g0.fw, pos_g0), lean_safe_key((fb & 32u) != 0, g1.fw, pos_g0 + ((fb >> 4) & 1u)));
    }
    if (lane < n_act) {
        lean_emit(a, fin32, (fb & 64u) != 0, L.act[lane][0], L.act[lane][1], base_a + w.rank(ma));
        safe_key = min32(safe_key, lean_safe_key((fb & 64u) != 0, L.act[lane][1], base_a + w.rank(ma)));
    }
    return a;
}

// A final list with action tags NormalizeTag changes (bits of chg; fb bit 6: the tag is in the list; its position is base_a +
// its rank among those).  The list in fin32[0 .. nf) stays what the verdict and the audit row speak of; the caption's list
// is that list with NormalizeTag applied and first occurrences kept (FormatCaptionSingleLine, Program.cs:971-974): only a
// changed tag's new form can meet another element.  The caption's list replaces the final list in fin32; the final list goes
// to aud_row (AUD_KEPT) when there is one.  GetUnknownTags (Program.cs:817-824) normalises the same way, so the tags outside
// the vocabulary are counted in the caption's list (unknown, their bytes in ubytes).  n = LEAN_DEAD: not for this stage.
struct LeanCap { uint32_t n, bytes, unknown, ubytes; };
template <class W, class TT>
TP_HD_FLAT TP_NOINLINE LeanCap lean_caption_fix(W& w, const TT& t, LeanLds& L, uint32_t* fin32, uint32_t fb, uint32_t base_a, uint32_t n_act, uint64_t chg, uint32_t nf,
                                                uint32_t* aud_row) {
    const uint32_t lane = w.lane();
    w.sync();                                                       // the final list is in fin32
    uint32_t* cN = L.tab;                                           // the changed tags' new forms and positions (the table area is idle here)
    uint32_t* cP = L.tab + 64;
    const uint64_t ma = w.ballot(fb & 64u);
    const bool mine = lane < n_act && (fb & 64u) != 0 && ((chg >> lane) & 1ull) != 0;
    const uint64_t mc = w.ballot(mine);
    bool bad = false;
    if (mine) {
        const uint32_t N = t.next[L.act[lane][0] & LEAN_ID_MASK];
        const uint32_t fwN = t.fw_tix[2 * (size_t)N];
        bad = (fwN & FW_BLANK) != 0 || (fwN & FW_LEN_MASK) == FW_LEN_MASK;
        cN[w.rank(mc)] = N; cP[w.rank(mc)] = base_a + w.rank(ma);
    }
    if (w.ballot(bad) != 0) return LeanCap{LEAN_DEAD, 0, 0, 0};
    w.sync();
    const uint32_t nc = popc64(mc);
    uint32_t id[2], cap[2];
    bool have[2], keep[2];
    for (uint32_t e = 0; e < 2; ++e) {
        const uint32_t j = lane + 64 * e;
        have[e] = j < nf;
        id[e] = have[e] ? fin32[j] : NO_ID;
        cap[e] = id[e];
        keep[e] = have[e];
    }
    for (uint32_t c = 0; c < nc; ++c) for (uint32_t e = 0; e < 2; ++e) if (lane + 64 * e == cP[c]) cap[e] = cN[c];
    for (uint32_t c = 0; c < nc; ++c) {
        const uint32_t N = cN[c], P = cP[c];
        bool earlier = false;
        for (uint32_t e = 0; e < 2; ++e) {
            const uint32_t j = lane + 64 * e;
            if (have[e] && j != P && cap[e] == N) { if (j < P) earlier = true; else keep[e] = false; }
        }
        if (w.ballot(earlier) != 0) for (uint32_t e = 0; e < 2; ++e) if (lane + 64 * e == P) keep[e] = false;
    }
    uint32_t len = 0, unk = 0, ulen = 0;
    for (uint32_t e = 0; e < 2; ++e) if (keep[e]) {
        const uint32_t fw = t.fw_tix[2 * (size_t)cap[e]];
        len += fw & FW_LEN_MASK;
        if (t.known_active && (fw & FW_UNKNOWN)) { ++unk; ulen += fw & FW_LEN_MASK; }
    }
    const uint64_t k0 = w.ballot(keep[0]), k1 = w.ballot(keep[1]);
    if (aud_row) for (uint32_t e = 0; e < 2; ++e) if (have[e]) aud_row[AUD_KEPT + lane + 64 * e] = id[e];
    w.sync();                                                       // every lane has read its part of the list
    if (keep[0]) fin32[w.rank(k0)] = cap[0];
    if (keep[1]) fin32[popc64(k0) + w.rank(k1)] = cap[1];
    return LeanCap{popc64(k0) + popc64(k1), w.sum(len), w.sum(unk), w.sum(ulen)};
}

// The piece of the line this lane scans: the aligned 16 bytes number `lane` from the 16-byte boundary at or before the
// line's start (zeros behind the line's last piece).  A caller may fetch it well before lean_file needs it.
template <class W>
TP_HD_FLAT void lean_fetch(W& w, const uint8_t* text, uint32_t len, uint32_t (&x)[4]) {
    const uint32_t shift = (uint32_t)((uintptr_t)text & 15);
    x[0] = x[1] = x[2] = x[3] = 0;
    if (len <= LEAN_LINE_MAX && 16 * w.lane() < shift + len) w.load16(text - shift + 16 * w.lane(), x);
}

// The front half for one file: line scan, token lookup, Distinct (Program.cs:589-602).  x: what lean_fetch(w, text, len, x)
// gave.  On success the file's candidates - interned ids in candidate order - are in row[0 .. n_cand); false = declined.
// TT: DevTables, or DevTables in the constant address space (the GPU kernels read the table scalars with scalar loads
// where they are used).
template <class W, class TT>
TP_HD_FLAT bool lean_front(W& w, const TT& t, LeanFrontLds& L, const LeanShared& S, const uint8_t* text, uint32_t len, const uint32_t (&x)[4], uint32_t* row, uint32_t& n_cand) {
    const uint32_t lane = w.lane();
    if (len > LEAN_LINE_MAX || len == 0) return LEAN_NO(1);
    w.tick(16);
    // ---- the line, 16 bytes per lane: to LDS for the token reads, and scanned for commas in registers -------------------
    const uint32_t shift = (uint32_t)((uintptr_t)text & 15);
    uint32_t T;
    {
        const uint32_t end = shift + len, pieces = (end + 15) >> 4;   // at most 64
        w.store16(L.line + 16 + 16 * lane, x);
        const uint32_t ne = ne_nibble(x[0], rep4(',')) | (ne_nibble(x[1], rep4(',')) << 4) | (ne_nibble(x[2], rep4(',')) << 8) | (ne_nibble(x[3], rep4(',')) << 12);
        const int lo16 = (int)shift - 16 * (int)lane, hi16 = (int)end - 16 * (int)lane;
        const uint32_t lo = lo16 < 0 ? 0u : (uint32_t)lo16, hi = hi16 < 0 ? 0u : (hi16 > 16 ? 16u : (uint32_t)hi16);   // the piece's bytes [lo, hi) belong to the text
        const uint32_t valid = hi > lo ? (((1u << (hi - lo)) - 1u) << lo) : 0u;
        uint32_t commas = ~ne & valid;
        const uint32_t high = (x[0] | x[1] | x[2] | x[3]) & 0x80808080u;   // a byte of a character outside ASCII (the neighbours' bytes count too: declining is safe)
        const uint32_t cnt = popc64(commas);
        const uint32_t incl = w.scan_incl(cnt);
        const uint32_t ncomma = w.bcast(incl, 63);
        if (w.ballot(high != 0 && lane < pieces) != 0) return LEAN_NO(2);
        T = ncomma + 1;
        if (ncomma < 2 || T > 64) return LEAN_NO(3);                // fewer than two commas: LooksBad measures the line (Program.cs:1145)
        uint32_t k = incl - cnt;
        while (commas) { L.cpos[1 + k] = 16 * lane + ctz32(commas) - shift; ++k; commas &= commas - 1; }
        if (lane == 0) { L.cpos[0] = 0xFFFFFFFFu; L.cpos[T] = len; }
    }
    w.sync();
    w.tick(17);

    // ---- tokens -> interned ids (NormalizeRawTagLine's per-token steps and Distinct, Program.cs:598-602) -----------------
    uint32_t id = NO_ID, f = FL_BLANK;                              // what NormalizeTag makes of this lane's token, and its flags
    bool kept = false;
    {
        const uint8_t* s = L.line + 16 + shift;
        const bool tok = lane < T;
        uint32_t n = 0, i2 = 0;
        bool decline = false, miss = false;
        {
            uint32_t i = 0, j = 0;
            if (tok) { i = L.cpos[lane] + 1; j = L.cpos[lane + 1]; }
            const uint32_t head = ld32u(s + i), tail = ld32u(s + j - 4);
            const uint32_t lead = (uint32_t)(i < j) & ws1(head & 0xFFu);
            i2 = i + lead;
            const uint32_t trail = (uint32_t)(j > i2) & ws1(tail >> 24);
            const uint32_t j2 = j - trail;
            const uint32_t inner_first = lead ? (head >> 8) & 0xFFu : head & 0xFFu, inner_last = trail ? (tail >> 16) & 0xFFu : tail >> 24;
            n = j2 - i2;
            if (n > 0 && (ws1(inner_first) | ws1(inner_last))) decline = true;   // more white space at an end than this stage trims
            if (n > 32) decline = true;
        }
        if (w.ballot(tok && decline) != 0) return LEAN_NO(4);
        bool hop = false;
        if (tok && n > 0) {
            // the token's bytes as eight zero-padded words and its two slots
            const uint64_t* mk = S.tail_mask[n];
            const uint64_t w0 = ld64u(s + i2) & mk[0], w1 = ld64u(s + i2 + 8) & mk[1], w2 = ld64u(s + i2 + 16) & mk[2], w3 = ld64u(s + i2 + 24) & mk[3];
            const uint32_t kw[8] = {(uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32), (uint32_t)w2, (uint32_t)(w2 >> 32), (uint32_t)w3, (uint32_t)(w3 >> 32)};
            uint32_t s1, s2;
            lean_slots(kw, n, t.lslot_shift, t.lslot_seed, s1, s2);
            const LeanSlot* A = t.lslots + s1;
            const LeanSlot* B = t.lslots + s2;
            // the table builder fills a key's first slot when it can, so most keys are found there: the second slot is read
            // only by the lanes whose first slot holds something else
            uint64_t la[2];
            __builtin_memcpy(la, A->key_lo, 16);
            const bool lo_a = ((la[0] ^ w0) | (la[1] ^ w1)) == 0;
            auto first_half_in_b = [&]() {
                uint64_t lb[2];
                __builtin_memcpy(lb, B->key_lo, 16);
                return ((lb[0] ^ w0) | (lb[1] ^ w1)) == 0;
            };
            const bool lo_b = !lo_a && first_half_in_b();
            miss = true;
            // the slot whose first half matches holds the token when the lengths agree and, for a token longer than 16 bytes,
            // the second half matches too (two long keys with one first half may sit in both slots: then the other is tried)
            auto try_slot = [&](const LeanSlot* P) {
                uint32_t pay[4];
                __builtin_memcpy(pay, &P->head, 16);
                if (((pay[0] >> 20) & 63u) != n) return;
                if (n > 16) {
                    uint64_t kh[2];
                    __builtin_memcpy(kh, t.lkey_hi + 4 * (size_t)pay[2], 16);
                    if (((kh[0] ^ w2) | (kh[1] ^ w3)) != 0) return;
                }
                miss = false;
                id = pay[0] & LEAN_HEAD_ID_MASK; f = pay[1];
                hop = ((pay[0] >> 28) & 1u) != 0;
            };
            if (lo_a || lo_b) try_slot(lo_a ? A : B);
            if (miss && lo_a && first_half_in_b()) try_slot(B);        // the first slot held another key with this first half
        }
        if (w.ballot(miss) != 0) {
            // in no table as it stands: dropped under strict vocabulary, provided it is plain (then normalisation leaves it
            // alone and it is in no table at all); otherwise the general stages normalise it
            bool keep_it = false;                                   // a plain token in no table, and such tags are kept: declined, and worth noting
            if (miss) {
                const bool plain = lean_plain_unknown(s + i2, n);
                if (!(t.drop_unknown && plain)) decline = true;
                keep_it = plain && !t.drop_unknown;
            }
            if (w.ballot(keep_it) != 0 && lane == 0) L.cpos[LEAN_CPOS_NOTE] = 1;
        }
        // LooksBad's phrases and code fences are looked for in the raw line (Program.cs:592-596): any token holding one declines
        if (id != NO_ID && (f & (FL_PHRASE | FL_FENCE | FL_SPLIT))) decline = true;
        kept = id != NO_ID && (f & FL_LEN2) && !(f & FL_DROP);     // Length >= 2, !ShouldDropTag (Program.cs:600-601)
        if (kept && hop) decline = true;                           // NormalizeTag would change it again at the next site: tp_fast.h follows such hops
        if (w.ballot(decline) != 0) return LEAN_NO(5);
    }
    w.tick(18);
    // Distinct (Program.cs:602): first occurrences, handed on in candidate order
    const uint32_t v1[1] = {lean_val(lane, id, kept)};
    const uint32_t fb = lean_first<1>(w, L, v1);
    if (fb >> 31) return LEAN_NO(6);
    const bool act = (fb & 1u) != 0;
    const uint64_t m_act = w.ballot(act);
    n_cand = popc64(m_act);
    // fewer than three selected tags mean the minimal fallback line (Program.cs:297-298): every candidate AddOne accepts ends up
    // selected (the last sweep takes what the buckets left, Program.cs:677-680), so their number says it here already
    if (t.n_prefix + popc64(w.ballot(act && !(f & FL_BLANK) && (f & FL_VALID) && !(f & FL_DROP) && !(f & FL_PREFIX))) < 3u) return LEAN_NO(7);
    if (act) row[w.rank(m_act)] = id;
    w.tick(19);
    return true;
}

// The back half for one file: everything from the candidates on (Program.cs:223-229, 292-329).  id: candidate number `lane`
// of the file (lane < n_cand; NO_ID otherwise), as lean_front left it; L.cr[0][l], L.cr[1][l]: the record of candidate l's
// string (DevTables::crec), for every l < n_cand, put there by the caller.  records_free() is called at most once, at the
// point from which L.cr is no longer read (a caller that fills it for the next file ahead of time may start then; it is
// not called on every path).  rc: the rule cache; fin32: the caption's tag ids (FAST_FIN_MAX entries); fired (fired_words
// words) and facet_choice may be null; with fired the evidence set is counted (tp_verdict::n_evidence).  false = declined.
// aud (audit mode; row null otherwise): where the file's other lists go - the evidence set in insertion order, the dropped
// candidates with their reason, the first TOP_TAGS candidates (AUD_* in tp_common.h) - with their entry counts, text
// lengths and the review record's bits, as the general kernel reports them in audit mode (tp_pipeline.h); the kept tags
// and the caption tags outside the vocabulary are read from fin32 by whoever writes the text.
// DETAIL false (the plain run's build of the kernel): aud is taken as null whatever is passed, and with fired a file with a
// seed whose closure has more than two members is declined (the evidence set is counted, not listed).
// FIX: files whose action tags NormalizeTag changes are taken (lean_caption_fix); without it they are declined.
template <bool DETAIL = true, bool FIX = DETAIL, class W, class TT, class RF>
TP_HD_FLAT bool lean_back(W& w, const TT& t, LeanLds& L, const LeanShared& S, const uint32_t* rc, uint32_t id, uint32_t n_cand, FastOut& out, uint32_t* fin32,
                     uint32_t* fired_or_null, uint16_t* facet_choice_or_null, RF&& records_free, const LeanAudit aud_or_null = LeanAudit{nullptr, nullptr, nullptr, nullptr}) {
    uint32_t* const fired = fired_or_null;
    uint16_t* const facet_choice = facet_choice_or_null;
    const LeanAudit aud = DETAIL ? aud_or_null : LeanAudit{nullptr, nullptr, nullptr, nullptr};
    const uint32_t lane = w.lane();
    const bool act = lane < n_cand;
    const uint32_t ci = lane;
    // the candidate's record: what NormalizeTag, ClassifyTag, the rules and ExpandImpliedTags need to know of the string
    uint32_t f = FL_BLANK, lit = CREC_LIT_NONE, s_info = CREC_LIT_NONE | CREC_LIT_NONE << 16, ev_step[2] = {NO_ID, NO_ID};
    uint32_t own_len = 0, step_fw[2] = {0, 0}, own_cnt = 0;         // (for the evidence set as a list: lengths, the steps' words, the step count)
    {
        bool too_long = false;
        if (act) {
            uint32_t c0[4], c1[4];
            w.load16((const uint8_t*)&L.cr[0][lane], c0);
            w.load16((const uint8_t*)&L.cr[1][lane], c1);
            f = c0[0]; lit = c0[2] & CREC_LIT_NONE; s_info = c0[3];
            if (DETAIL && (fired || aud.row)) { own_len = c0[1] & FW_LEN_MASK; step_fw[0] = c1[2]; step_fw[1] = c1[3]; own_cnt = (c0[2] >> 24) & 3u; }
            too_long = (c0[2] & CREC_TOO_LONG) != 0;
            ev_step[0] = c1[0] == NO_ID ? NO_ID : c1[0] & LEAN_ID_MASK; ev_step[1] = c1[1] == NO_ID ? NO_ID : c1[1] & LEAN_ID_MASK;
            L.ids[lane] = id;
        }
        if (w.ballot(too_long) != 0) return LEAN_NO(8);           // a length the record has no room for
    }
    w.tick(20);
    LEAN_STOP(1, f ^ lit ^ s_info ^ ev_step[0] ^ ev_step[1])

    // ---- candidates (Program.cs:625-655): scores fall with the candidate index, so sorted order is candidate order ----------
    const bool nonblank = act && !(f & FL_BLANK);
    const bool bad = nonblank && (!(f & FL_VALID) || (f & FL_DROP));
    const uint32_t n_dropped = popc64(w.ballot(bad));               // "{n}:invalid_tag" / "{n}:policy_drop"
    const uint32_t style_n = popc64(w.ballot(act && fl_bucket(f) == B_STYLE && ci < t.style_ci_limit));   // DetermineStyleQualityCap, Program.cs:702-710
    const bool seed = nonblank && ci < t.seed_ci_limit;             // !(score < 0.28), Program.cs:724

    // ---- evidence (BuildEvidenceTags + ExpandImpliedTags, Program.cs:301-302): literal bitset, feet / penis -----------------
    for (uint32_t k = lane; k < t.lit_words; k += 64) L.ebits[k] = 0;
    w.sync();
    const bool seed_edge = seed && (f & FL_EDGE);
    // a closure beyond the record's two steps is read from the reach lists
    if (w.ballot(seed_edge && !(f & FL_REACH2)) != 0 && !t.closure_tables) return LEAN_NO(9);
    bool ev_feet = false, ev_penis = false;
    uint32_t n_evidence = 0xFFFF;
    {
        bool ev_far = false;
        uint32_t ev_id[3] = {id, NO_ID, NO_ID};
        if (seed) {
            if (lit != CREC_LIT_NONE) w.lds_or(&L.ebits[lit >> 5], 1u << (lit & 31));
            ev_feet = (f & FL_NFEET) != 0; ev_penis = (f & FL_NPENIS) != 0;
            if ((f & FL_EDGE) && !(f & FL_REACH2)) {
                const uint32_t m0 = t.reach_off[id], m1 = t.reach_off[id + 1];
                if (!DETAIL && fired && m1 - m0 > 2) ev_far = true;   // (the plain run's build) more members than this lane can hand to the count
                for (uint32_t m = m0; m < m1; ++m) {
                    const uint32_t info = t.reach_ent[2 * m + 1], li = info & 0xFFFFFFu;
                    if (li != 0xFFFFFFu) w.lds_or(&L.ebits[li >> 5], 1u << (li & 31));
                    ev_feet |= ((info >> 24) & 1u) != 0;
                    ev_penis |= ((info >> 25) & 1u) != 0;
                    if (!DETAIL && m - m0 < 2) ev_id[1 + (m - m0)] = t.reach_ent[2 * m];
                }
                if (DETAIL) { ev_id[1] = ev_step[0]; ev_id[2] = ev_step[1]; }
            } else if (f & FL_EDGE) {
                const uint32_t l0 = s_info & CREC_LIT_NONE, l1 = (s_info >> 16) & CREC_LIT_NONE;   // absent steps have no literal and no bits
                if (l0 != CREC_LIT_NONE) w.lds_or(&L.ebits[l0 >> 5], 1u << (l0 & 31));
                if (l1 != CREC_LIT_NONE) w.lds_or(&L.ebits[l1 >> 5], 1u << (l1 & 31));
                ev_feet |= (s_info & 0x10001000u) != 0;
                ev_penis |= (s_info & 0x20002000u) != 0;
                ev_id[1] = ev_step[0]; ev_id[2] = ev_step[1];
            }
        }
        ev_feet = w.ballot(ev_feet) != 0;
        ev_penis = w.ballot(ev_penis) != 0;
        if constexpr (!DETAIL) {
            if (fired) {                                            // the evidence set's size only: distinct ids among seeds and their closures
                if (w.ballot(ev_far) != 0) return LEAN_NO(9);
                const uint32_t v3[3] = {lean_val(3 * lane, ev_id[0], seed), lean_val(3 * lane + 1, ev_id[1], seed && ev_id[1] != NO_ID),
                                        lean_val(3 * lane + 2, ev_id[2], seed && ev_id[2] != NO_ID)};
                const uint32_t fb = lean_first<3>(w, L, v3);
                if (fb >> 31) return LEAN_NO(10);
                n_evidence = popc64(w.ballot(fb & 1u)) + popc64(w.ballot(fb & 2u)) + popc64(w.ballot(fb & 4u));
            }
        } else if (fired || aud.row) {
            // The evidence set as a list, in insertion order (BuildEvidenceTags, then ExpandImpliedTags' rounds: Program.cs:721-729,
            // 1160-1179): the seeds in candidate order; then what the first round appends, the seeds' steps by seed; then what the
            // second round appends, the steps of those steps that lead further, by the step they belong to.  A seed whose closure
            // goes beyond that (more than two steps, a step's step that leads further still) is not for this stage.  Its size is
            // a detail output (n_evidence), the list itself goes into the audit row.
            const bool two_levels = seed && (f & FL_EDGE) && !(f & FL_REACH2);
            const bool any_two = w.ballot(two_levels) != 0;
            uint32_t l2[4] = {NO_ID, NO_ID, NO_ID, NO_ID}, l2_len[4] = {0, 0, 0, 0};
            if (two_levels) {
                if (own_cnt > 2) ev_far = true;
                for (uint32_t j = 0; j < 2; ++j) {
                    const uint32_t e = ev_step[j];
                    if (e == NO_ID || !(step_fw[j] & FW_EDGE)) continue;
                    const Rec re = t.rec[e];
                    if (!(re.flags & FL_REACH2) || re.step_cnt > 2) { ev_far = true; continue; }
                    if (re.step_cnt > 0) l2[2 * j] = re.step0;
                    if (re.step_cnt > 1) l2[2 * j + 1] = re.step1;
                }
                if (aud.row) for (uint32_t q = 0; q < 4; ++q) if (l2[q] != NO_ID) {
                    l2_len[q] = t.fw_tix[2 * (size_t)l2[q]] & FW_LEN_MASK;
                    if (l2_len[q] == FW_LEN_MASK) ev_far = true;
                }
            }
            if (w.ballot(ev_far) != 0) return LEAN_NO(9);
            const uint32_t v7[7] = {lean_val(lane, ev_id[0], seed), lean_val(64 + 2 * lane, ev_id[1], seed && ev_id[1] != NO_ID),
                                    lean_val(65 + 2 * lane, ev_id[2], seed && ev_id[2] != NO_ID),
                                    lean_val(192 + 4 * lane, l2[0], l2[0] != NO_ID), lean_val(193 + 4 * lane, l2[1], l2[1] != NO_ID),
                                    lean_val(194 + 4 * lane, l2[2], l2[2] != NO_ID), lean_val(195 + 4 * lane, l2[3], l2[3] != NO_ID)};
            uint32_t fb;
            if (any_two) fb = lean_first<7>(w, L, v7);
            else { const uint32_t v3[3] = {v7[0], v7[1], v7[2]}; fb = lean_first<3>(w, L, v3); }
            if (fb >> 31) return LEAN_NO(10);
            const uint64_t ms = w.ballot(fb & 1u), m0 = w.ballot(fb & 2u), m1 = w.ballot(fb & 4u);
            uint64_t mq[4] = {0, 0, 0, 0};
            if (any_two) for (uint32_t q = 0; q < 4; ++q) mq[q] = w.ballot(fb & (8u << q));
            const uint32_t n0 = popc64(ms), n1 = popc64(m0) + popc64(m1), n2 = popc64(mq[0]) + popc64(mq[1]) + popc64(mq[2]) + popc64(mq[3]);
            n_evidence = n0 + n1 + n2;
            if (aud.row && n_evidence > AUD_DROPPED - AUD_EVIDENCE) return LEAN_NO(9);   // no room in the row
            if (aud.row) {
                const uint32_t at1 = n0 + w.rank(m0) + w.rank(m1);
                if (fb & 1u) aud.row[AUD_EVIDENCE + w.rank(ms)] = ev_id[0];
                if (fb & 2u) aud.row[AUD_EVIDENCE + at1] = ev_id[1];
                if (fb & 4u) aud.row[AUD_EVIDENCE + at1 + ((fb >> 1) & 1u)] = ev_id[2];
                uint32_t mine = ((fb & 1u) ? own_len : 0u) + ((fb & 2u) ? step_fw[0] & FW_LEN_MASK : 0u) + ((fb & 4u) ? step_fw[1] & FW_LEN_MASK : 0u);
                if (any_two) {
                    uint32_t at2 = n0 + n1 + w.rank(mq[0]) + w.rank(mq[1]) + w.rank(mq[2]) + w.rank(mq[3]);
                    for (uint32_t q = 0; q < 4; ++q) if (fb & (8u << q)) { aud.row[AUD_EVIDENCE + at2++] = l2[q]; mine += l2_len[q]; }
                }
                const uint32_t bytes = w.sum(mine);
                if (lane == 0) { aud.cnt[AUDIT_EVIDENCE] = n_evidence; aud.len[AUDIT_EVIDENCE] = bytes + (n_evidence ? n_evidence - 1 : 0u); }
            }
        }
        if (aud.row) {
            // DroppedTagsWithReason (Program.cs:630-631, 682), the first TOP_TAGS raw candidates (Program.cs:860), the review record's bits
            const uint64_t mb = w.ballot(bad);
            if (bad) aud.row[AUD_DROPPED + w.rank(mb)] = id | ((f & FL_VALID) ? 0x80000000u : 0u);
            const uint32_t ntop = n_cand < (uint32_t)TOP_TAGS ? n_cand : (uint32_t)TOP_TAGS;
            if (lane < ntop) aud.row[AUD_TOP + lane] = id;
            const uint32_t dbytes = w.sum(bad ? own_len + AUDIT_REASON_LEN : 0u), tbytes = w.sum(lane < ntop ? own_len : 0u);
            uint32_t rbits = w.or_all(nonblank ? (f & (FL_SUBJ_MALE | FL_SUBJ_GIRL | FL_SUBJ_MULTI)) >> 10 : 0u);
            if (w.ballot(nonblank && (f & FL_BANNED)) != 0)
                for (int k = 0; k < N_BANNED; ++k) if (w.ballot(nonblank && id == t.banned_ids[k]) != 0) rbits |= 1u << k;
            if (lane == 0) {
                aud.cnt[AUDIT_DROPPED] = n_dropped; aud.len[AUDIT_DROPPED] = dbytes + (n_dropped ? n_dropped - 1 : 0u);
                aud.cnt[AUDIT_TOP] = ntop; aud.len[AUDIT_TOP] = tbytes + (ntop ? ntop - 1 : 0u);
                *aud.rbits = rbits;
            }
        }
    }
    w.sync();
    w.tick(21);
    LEAN_STOP(2, n_dropped + style_n + (uint32_t)seed + (uint32_t)ev_feet + (uint32_t)ev_penis + L.ebits[lane] + n_evidence)

    // ---- InferActionTags / EnsureRequiredFacets (Program.cs:731-775): both read the evidence only -------------------------
    bool frules_possible;
    uint32_t n_act = 0;                                             // tags the action rules offer (in L.act)
    uint64_t chg = 0;                                               // those of them NormalizeTag changes (the detail build takes such files)
    {
        bool lit_r = false, lit_f = false;
        for (uint32_t k = lane; k < t.lit_words; k += 64) {
            const uint32_t e = L.ebits[k];
            lit_r = lit_r || (e & rc[t.rc_rule_mask + k]) != 0;
            lit_f = lit_f || (e & rc[t.rc_frule_mask + k]) != 0;
        }
        frules_possible = t.frules_vacuous || w.ballot(lit_f) != 0;
        const bool rules_possible = w.ballot(lit_r) != 0;           // the evidence holds a literal some action rule tests positively: look closer
        if (rules_possible) {
            n_act = lean_actions<FIX>(w, t, L, rc, fired, chg);
            if (n_act == LEAN_DEAD) return LEAN_NO(11);
        } else if (fired && lane < t.fired_words) fired[lane] = 0;
    }
    uint32_t f_val = LEAN_DEAD, f_fw = 0, f_tix = TIX_NONE;        // lane q: the tag facet q asks for
    {
        if (frules_possible) {
            for (uint32_t p0 = 0; p0 < t.n_frules; p0 += 64) {
                const uint32_t p = p0 + lane;
                const uint64_t m = w.ballot(p < t.n_frules && lean_rc_match(L.ebits, rc, t.rc_frules + p * RC_REC_WORDS));
                if (lane == 0) { L.fhit[p0 >> 5] = (uint32_t)m; L.fhit[(p0 >> 5) + 1] = (uint32_t)(m >> 32); }
            }
            w.sync();
        }
        bool changes = false;
        if (lane < t.n_facets) {
            const uint32_t* fr = rc + t.rc_facets + RC_FACET_WORDS * lane;
            uint32_t fsel = fr[0];
            f_fw = fr[6]; f_tix = fr[7]; changes = fr[5] != 0;
            uint16_t choice = 0xFFFF;
            if (frules_possible) {
                const uint32_t rule_off = fr[1], rule_cnt = fr[2];
                for (uint32_t q = 0; q < rule_cnt; ++q) {           // first matching rule wins (Program.cs:760-766)
                    const uint32_t p = rule_off + q;
                    if ((L.fhit[p >> 5] >> (p & 31)) & 1u) {
                        const uint32_t* pr = rc + t.rc_frules + p * RC_REC_WORDS;
                        fsel = pr[0]; f_fw = pr[10]; f_tix = pr[11]; changes = (pr[3] & 0x8000u) != 0;
                        choice = (uint16_t)q;
                        break;
                    }
                }
            }
            f_val = lean_val(1024 + lane, fsel, true);
            if ((f_fw & FW_LEN_MASK) == FW_LEN_MASK) changes = true;   // a length the word has no room for
            if (facet_choice) facet_choice[lane] = choice;
        }
        if (w.ballot(changes) != 0) return LEAN_NO(12);             // a facet tag NormalizeTag changes
    }
    w.tick(22);
    LEAN_STOP(3, f_val ^ f_fw ^ f_tix ^ n_act ^ n_dropped ^ style_n ^ (uint32_t)ev_feet)

    // ---- SelectTags' bucket fills (Program.cs:661-699) ------------------------------------------------------------------------
    uint32_t nr = t.n_prefix;
    {
        const uint32_t style_cap = style_n < 2 ? 2u : (style_n > 4 ? 4u : style_n);
        const bool ok = nonblank && !bad && !(f & FL_PREFIX);       // AddOne's tests; prefix tags are already in
        // FillBucket for the six capped buckets (Program.cs:670-678): a candidate belongs to one of them at most, so each lane
        // keeps its rank among its own bucket's candidates and that bucket's place and share (place << 8 | share); what a
        // bucket takes depends on the room the buckets before it left, which is counted in scalar registers
        const uint32_t key = ok ? fl_bucket(f) : 7u;
        uint32_t rk = 0, share = 0;
        TP_UNROLL for (uint32_t bk = 0; bk < 6; ++bk) {
            const bool mine = key == bk;
            const uint64_t m = w.ballot(mine);
            const uint32_t room = nr < (uint32_t)MAX_TOTAL_TAGS ? (uint32_t)MAX_TOTAL_TAGS - nr : 0u;
            const uint32_t cap = bk == 0 ? 3u : bk == 1 ? 10u : bk == 2 ? 8u : bk == 3 ? 6u : bk == 4 ? 4u : style_cap;
            uint32_t take = popc64(m);
            take = take < cap ? take : cap;
            take = take < room ? take : room;
            rk = mine ? w.rank(m) : rk;
            share = mine ? (nr << 8 | take) : share;
            nr += take;
        }
        bool taken = key < 6u && rk < (share & 255u);
        uint32_t pos = (share >> 8) + rk;
        {                                                           // the final sweep over what is left, in candidate order (Program.cs:680)
            const bool el = ok && !taken;
            const uint64_t m = w.ballot(el);
            const uint32_t room = nr < (uint32_t)MAX_TOTAL_TAGS ? (uint32_t)MAX_TOTAL_TAGS - nr : 0u;
            uint32_t take = popc64(m);
            take = take < room ? take : room;
            const uint32_t r6 = w.rank(m);
            if (el && r6 < take) { taken = true; pos = nr + r6; }
            nr += take;
        }
        if (nr < 3) return LEAN_NO(7);                              // the minimal fallback line (Program.cs:297-298)
        if (taken) L.inv[pos] = lane;
    }
    w.sync();
    w.tick(20);
    LEAN_STOP(4, f_val ^ f_fw ^ f_tix ^ n_act ^ n_dropped ^ nr ^ L.inv[lane])

    // ---- the selected tags by position; EvaluateCoverage (Program.cs:1056-1095) --------------------------------------------
    // position p < n_prefix: a prefix tag; else the candidate of lane inv[p]
    const bool in_sel = lane < nr;
    uint32_t p_val = LEAN_DEAD, p_fl = FL_BLANK, p_fw = FW_BLANK, p_tix = TIX_NONE;
    uint32_t k_val[2] = {LEAN_DEAD, LEAN_DEAD}, k_fw[2] = {0, 0}, k_tix[2] = {TIX_NONE, TIX_NONE};
    uint32_t k_cnt = 0;
    if (in_sel) {
        if (lane >= t.n_prefix) {
            const uint32_t cl = L.inv[lane];
            uint32_t c0[4], c1[4];
            w.load16((const uint8_t*)&L.cr[0][cl], c0);
            w.load16((const uint8_t*)&L.cr[1][cl], c1);
            p_val = lean_val(lane, L.ids[cl], true); p_fl = c0[0]; p_fw = c0[1]; p_tix = (c0[2] >> 12) & TIX_NONE; k_cnt = (c0[2] >> 24) & 3u;
            k_val[0] = lean_val(64 + 2 * lane, c1[0] & LEAN_ID_MASK, c1[0] != NO_ID); k_val[1] = lean_val(65 + 2 * lane, c1[1] & LEAN_ID_MASK, c1[1] != NO_ID);
            k_tix[0] = c1[0] >> LEAN_ID_BITS; k_tix[1] = c1[1] >> LEAN_ID_BITS;
            k_fw[0] = c1[2]; k_fw[1] = c1[3];                      // the steps' final-list words (zero for an absent step)
        } else {
            p_val = lean_val(lane, S.prefix_ids[lane], true); p_fl = S.prefix_flags[lane]; p_fw = S.prefix_fw[lane]; p_tix = S.prefix_tix[lane];
        }
    }
    uint32_t coverage;
    {
        const uint32_t cv = w.or_all((in_sel && !(p_fw & FW_BLANK)) ? p_fw : 0u);
        coverage = ((cv & FW_TRIGGER) && (cv & FW_GIRL) ? TP_COV_IDENTITY : 0) | ((cv & FW_HAIR) ? TP_COV_HAIR : 0) | ((cv & FW_EYES) ? TP_COV_EYES : 0) |
                   ((cv & FW_OUTFIT) ? TP_COV_OUTFIT : 0) | ((cv & FW_POSE) ? TP_COV_POSE_ACTION : 0) | ((cv & FW_BACKGROUND) ? TP_COV_BACKGROUND : 0);
    }
    // a selected tag with FL_LEAF2 has steps NormalizeTag leaves alone and that lead nowhere further; the others' steps are followed one level down
    const bool deep = in_sel && (p_fl & FL_EDGE) && !(p_fl & FL_LEAF2);
    const bool any_deep = w.ballot(deep) != 0;
    if (!any_deep) { w.sync(); records_free(); }                    // (the deeper tags take the records' place)
    if (any_deep && lean_deep(w, t, L, deep, k_cnt, k_val[0], k_val[1], k_fw[0], k_fw[1])) return LEAN_NO(15);
    LEAN_STOP(5, f_val ^ f_fw ^ f_tix ^ n_act ^ n_dropped ^ nr ^ p_val ^ p_fl ^ p_fw ^ p_tix ^ k_val[0] ^ k_val[1] ^ k_fw[0] ^ k_fw[1] ^ k_tix[0] ^ k_tix[1] ^ coverage)

    // ---- the final set in the reference's insertion order (Program.cs:303-307): the selected tags, their implied tags by
    // position (and those one level further, by their parents' positions), the action tags by rule, the facet tags by
    // facet; an id already present is not added again.  Then the verdicts on the final list (Program.cs:309-329),
    // GetUnknownTags (:817-824) and the caption's length (:971-974), element by element ----------------------------------------
    const bool wide = any_deep || n_act != 0;                       // the usual file has the first four kinds of element only
    uint32_t fb, grown2 = 0, nact = 0;
    if (wide) {
        const LeanWide r = lean_first_wide(w, L, p_val, k_val[0], k_val[1], f_val, any_deep, n_act);
        fb = r.fb; grown2 = r.grown2; nact = r.nact;
    } else if (t.n_tix <= LEAN_TIX_MAX) {
        // Only a step or a facet tag can be in the list twice, and each of those strings has a target index: the lowest
        // order per index, in a table indexed by it, says which element is the first of its string.  The selected tags are
        // distinct and come first, so each of them is the first of its string; it enters the table only to be seen there.
        static_assert(LEAN_TIX_MAX <= 2 * LeanLds::TAB && LEAN_TIX_MAX % 256 == 0, "the table area holds one word per target index");
        TP_UNROLL for (uint32_t k = 0; k < LEAN_TIX_MAX / 256; ++k) w.fill16(&L.tab[256 * k + 4 * lane], 0xFFFFFFFFu);
        w.sync();
        const bool lk0 = k_val[0] != LEAN_DEAD, lk1 = k_val[1] != LEAN_DEAD, lf = f_val != LEAN_DEAD;
        if (w.ballot((lk0 && k_tix[0] >= LEAN_TIX_MAX) || (lk1 && k_tix[1] >= LEAN_TIX_MAX) || (lf && f_tix >= LEAN_TIX_MAX)) != 0) return LEAN_NO(13);   // (no such element exists)
        if (p_tix != TIX_NONE) w.lds_min(&L.tab[p_tix], p_val);
        if (lk0) w.lds_min(&L.tab[k_tix[0]], k_val[0]);
        if (lk1) w.lds_min(&L.tab[k_tix[1]], k_val[1]);
        if (lf) w.lds_min(&L.tab[f_tix], f_val);
        w.sync();
        fb = (uint32_t)in_sel | (uint32_t)(lk0 && L.tab[lk0 ? k_tix[0] : 0] == k_val[0]) << 1 | (uint32_t)(lk1 && L.tab[lk1 ? k_tix[1] : 0] == k_val[1]) << 2 |
             (uint32_t)(lf && L.tab[lf ? f_tix : 0] == f_val) << 3;
    } else {
        const uint32_t v4[4] = {p_val, k_val[0], k_val[1], f_val};
        fb = lean_first<4>(w, L, v4);
    }
    if (fb >> 31) return LEAN_NO(13);
    const uint64_t mk0 = w.ballot(fb & 2u), mk1 = w.ballot(fb & 4u), mf = w.ballot(fb & 8u);
    const uint32_t grown1 = popc64(mk0) + popc64(mk1), nfac = popc64(mf);
    const uint32_t pos_k0 = nr + w.rank(mk0) + w.rank(mk1), rank_f = w.rank(mf);
    const uint32_t nf = nr + grown1 + grown2 + nact + nfac;
    if (nf > FAST_FIN_MAX) return LEAN_NO(14);
    w.tick(23);
    LEAN_STOP(6, fb ^ nf ^ pos_k0 ^ rank_f ^ f_fw ^ n_dropped ^ p_fl ^ p_fw ^ k_fw[0] ^ k_fw[1] ^ coverage)

    LeanAcc acc = {0, 0};
    const uint32_t pos_k1 = pos_k0 + ((fb >> 1) & 1u), pos_f = nr + grown1 + grown2 + nact + rank_f;
    lean_emit(acc, fin32, (fb & 1u) != 0, p_val, p_fw, lane);
    lean_emit(acc, fin32, (fb & 2u) != 0, k_val[0], k_fw[0], pos_k0);
    lean_emit(acc, fin32, (fb & 4u) != 0, k_val[1], k_fw[1], pos_k1);
    lean_emit(acc, fin32, (fb & 8u) != 0, f_val, f_fw, pos_f);
    uint32_t wide_safe_key = 0xFFFFFFFFu;
    if (wide) {
        const LeanAcc wa = lean_emit_wide(w, t, L, fin32, fb, nr + grown1, nr + grown1 + grown2, any_deep, n_act, wide_safe_key);
        acc.sum += wa.sum; acc.bits |= wa.bits;
    }
    const uint32_t all_bits = w.or_all(acc.bits);
    uint32_t flags = 0;
    if (!((all_bits & FW_FACE) && (all_bits & FW_OUTFIT))) flags |= TP_F_LOW;
    int safe_hit = -1;
    if (t.safe_mode && (all_bits & FW_SAFE)) {                      // first final tag in SafeModeRejectTags (Program.cs:942-945)
        const uint32_t safe_key = min32(min32(lean_safe_key((fb & 1u) != 0, p_fw, lane), lean_safe_key((fb & 2u) != 0, k_fw[0], pos_k0)),
                                        min32(min32(lean_safe_key((fb & 4u) != 0, k_fw[1], pos_k1), lean_safe_key((fb & 8u) != 0, f_fw, pos_f)), wide_safe_key));
        if (lane == 0) L.scratch[0] = 0xFFFFFFFFu;
        w.sync();
        if (safe_key != 0xFFFFFFFFu) w.lds_min(&L.scratch[0], safe_key);
        w.sync();
        safe_hit = (int)(L.scratch[0] & 255u);
        flags |= TP_F_SAFE_REJECT;
    }
    const uint32_t sums = w.sum(acc.sum);
    const uint32_t bytes = sums & ((1u << LA_UNKNOWN_SHIFT) - 1u), unknown = t.known_active ? sums >> LA_UNKNOWN_SHIFT : 0u;
    if (aud.row) {                                                  // the kept tags are the final list; the unknown column walks it and keeps what qualifies
        auto ulen = [](bool first, uint32_t fw) { return first && (fw & FW_UNKNOWN) ? fw & FW_LEN_MASK : 0u; };
        uint32_t mine = ulen((fb & 1u) != 0, p_fw) + ulen((fb & 2u) != 0, k_fw[0]) + ulen((fb & 4u) != 0, k_fw[1]) + ulen((fb & 8u) != 0, f_fw);
        if (wide && unknown) {
            if (any_deep) mine += ulen((fb & 16u) != 0, L.deep[0][lane].fw) + ulen((fb & 32u) != 0, L.deep[1][lane].fw);
            if (lane < n_act) mine += ulen((fb & 64u) != 0, L.act[lane][1]);
        }
        const uint32_t ubytes = unknown ? w.sum(mine) : 0u;
        if (lane == 0) {
            aud.cnt[AUDIT_KEPT] = nf; aud.len[AUDIT_KEPT] = bytes + (nf - 1);
            aud.cnt[AUDIT_UNKNOWN] = nf; aud.len[AUDIT_UNKNOWN] = ubytes + (unknown ? unknown - 1 : 0u);
        }
    }
    if (ev_feet && ev_penis) { flags |= TP_F_ELIGIBLE; if (all_bits & FW_FOOTJOB) flags |= TP_F_FOOTJOB; }
    tp_verdict& v = out.verdict;
    v.coverage = (uint8_t)coverage;
    v.flags = (uint8_t)flags;
    v.n_tags = (uint16_t)nf;
    v.n_candidates = (uint16_t)n_cand;
    v.n_selected = (uint16_t)nr;
    v.safe_hit = (int16_t)safe_hit;
    v.n_evidence = (uint16_t)n_evidence;
    v.n_unknown = (uint16_t)unknown;
    v.n_dropped = (uint16_t)n_dropped;
    uint32_t n_cap = nf, cap_bytes = bytes;                         // the caption's list is the final list, unless NormalizeTag changes one of its tags
    if constexpr (FIX) {
        if (chg != 0) {
            const LeanCap c = lean_caption_fix(w, t, L, fin32, fb, nr + grown1 + grown2, n_act, chg, nf, aud.row);
            if (c.n == LEAN_DEAD) return LEAN_NO(16);
            n_cap = c.n; cap_bytes = c.bytes;
            v.n_tags = (uint16_t)n_cap;
            v.n_unknown = (uint16_t)c.unknown;
            if (aud.row && lane == 0) { aud.cnt[AUDIT_UNKNOWN] = n_cap; aud.len[AUDIT_UNKNOWN] = c.ubytes + (c.unknown ? c.unknown - 1 : 0u); }
        }
        if (aud.row && lane == 0) aud.row[AUD_FLAGS] = chg != 0 ? 1u : 0u;
    }
    out.n_final = n_cap;
    out.caption_len = cap_bytes + 2 * (n_cap - 1) + t.newline_len;
    w.tick(24);
    return out.caption_len <= FAST_CAPTION_MAX;
}

// Both halves in a row (the host-side logic test; the GPU runs them as two kernels with the candidates' ids in between).
template <class W, class TT>
TP_HD_FLAT bool lean_file(W& w, const TT& t, LeanFrontLds& LF, LeanLds& L, const LeanShared& S, const uint32_t* rc, const uint8_t* text, uint32_t len, const uint32_t (&x)[4],
                     uint32_t* row, FastOut& out, uint32_t* fin32, uint32_t* fired, uint16_t* facet_choice, const LeanAudit aud = LeanAudit{nullptr, nullptr, nullptr, nullptr}) {
    uint32_t n_cand = 0;
    if (!lean_front(w, t, LF, S, text, len, x, row, n_cand)) return false;
    w.sync();
    const uint32_t id = w.lane() < n_cand ? row[w.lane()] : NO_ID;
    if (w.lane() < n_cand) {
        uint32_t c0[4], c1[4];
        w.load16((const uint8_t*)(t.crec + 8 * (size_t)id), c0);
        w.load16((const uint8_t*)(t.crec + 8 * (size_t)id + 4), c1);
        w.store16((uint8_t*)&L.cr[0][w.lane()], c0);
        w.store16((uint8_t*)&L.cr[1][w.lane()], c1);
    }
    w.sync();
    return lean_back(w, t, L, S, rc, id, n_cand, out, fin32, fired, facet_choice, []() {}, aud);
}

}  // namespace tp
