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
constexpr uint32_t LEAN_TAB_BITS = 8;
constexpr uint32_t LEAN_TAB = 1u << LEAN_TAB_BITS;   // slots per min-order table
constexpr uint32_t LEAN_ID_BITS = 20;         // table values are order << 20 | id
constexpr uint32_t LEAN_ID_MASK = (1u << LEAN_ID_BITS) - 1;
constexpr uint32_t LEAN_DEAD = 0xFFFFFFFFu;   // no element
constexpr uint32_t LEAN_ROUNDS = 5;
constexpr uint32_t LEAN_FACETS_MAX = 64;      // one facet per lane
constexpr uint32_t LEAN_LIT_WORDS = 64;       // evidence bitset words (2048 rule literals)
constexpr uint32_t LEAN_CAND_WORDS = 12;      // words of a candidate's record in LDS

// What the rest of the pipeline needs to know about a candidate, kept in LDS from the token lookup to the point where the
// selected tags are put on the lanes by position (so that it does not occupy registers in between).
struct LeanCand {
    uint32_t id, flags, len, step_cnt;
    uint32_t step0, step1, s0_flags, s1_flags;   // the two steps of one ExpandImpliedTags visit (Rec) and, with FL_LEAF2, their flags ...
    uint32_t s0_len, s1_len, pad0, pad1;         // ... and byte lengths
};
static_assert(sizeof(LeanCand) == 4 * LEAN_CAND_WORDS, "three 16-byte pieces");
// Tags one level further down (steps of a selected tag's steps), per position: two per lane.
struct LeanDeep { uint32_t val, flags, len, pad; };

struct alignas(16) LeanLds {
    union {
        uint8_t line[16 + 1024 + 48];         // 16 spare bytes in front, 64 pieces, slack for 32-byte token reads
        LeanCand cand[64];                    // after the token lookup
        LeanDeep deep[2][64];                 // after the selected tags are on the lanes
    };
    uint32_t cpos[68];                        // cpos[1 + k]: position of comma k in the text; cpos[0] = -1, cpos[T] = len
    uint32_t tab[2 * LEAN_TAB];               // two min-order tables
    uint32_t ebits[LEAN_LIT_WORDS];           // evidence bitset over the rule literals
    uint32_t fhit[FAST_FRULES_MAX / 32];      // bit p: facet rule p matches the evidence
    uint32_t inv[64];                         // selected position -> candidate lane
    uint32_t scratch[4];
};
// Per block: masks that cut a 32-byte read down to a token's length; the tags SelectTags adds before the candidates.
struct alignas(16) LeanShared {
    uint64_t tail_mask[33][4];
    uint32_t prefix_ids[MAX_TOTAL_TAGS], prefix_flags[MAX_TOTAL_TAGS], prefix_len[MAX_TOTAL_TAGS];
};
TP_HD void lean_shared_init(const DevTables& t, LeanShared& S, uint32_t tid, uint32_t nthreads) {
    for (uint32_t k = tid; k < 33 * 4; k += nthreads) S.tail_mask[k >> 2][k & 3] = word_mask(k >> 2, k & 3);
    for (uint32_t k = tid; k < (uint32_t)MAX_TOTAL_TAGS; k += nthreads) {
        S.prefix_ids[k] = k < t.n_prefix ? t.prefix_ids[k] : NO_ID;
        S.prefix_flags[k] = k < t.n_prefix ? t.prefix_flags[k] : 0;
        S.prefix_len[k] = k < t.n_prefix ? t.prefix_len[k] : 0;
    }
}
// The tables fit this stage at all (checked once on the host).
inline bool lean_tables_ok(const DevTables& t) {
    return t.fast_ok && t.prefix_simple && t.n_strings <= LEAN_ID_MASK && t.n_facets <= LEAN_FACETS_MAX && t.n_prefix <= 8 && !t.rules_vacuous &&
           t.lit_words <= LEAN_LIT_WORDS;
}

// Bit k of the result: byte k of x differs from the byte pat repeats (exact for any bytes).
TP_HD uint32_t ne_nibble(uint32_t x, uint32_t pat) {
    const uint32_t z = x ^ pat;
    uint32_t t = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;   // 0x80 where the byte differs
    t |= t << 7;
    t |= t << 14;
    return t >> 28;
}
TP_HD uint32_t ws1(uint32_t c) { return (uint32_t)(c == 0x20u) | (uint32_t)(c - 9u <= 4u); }   // char.IsWhiteSpace below U+0080
TP_HD uint32_t lean_val(uint32_t order, uint32_t id, bool live) { return live ? (order << LEAN_ID_BITS) | id : LEAN_DEAD; }

// First occurrences by min-order tables.  val[e] = order << 20 | id describes element e of this lane (orders are distinct
// and below 2^12, ids below 2^20), LEAN_DEAD means no element.  Bit e of the result is set for the element with the lowest
// order among those with its id; bit 31 is set when elements are left undecided after LEAN_ROUNDS rounds (wave-uniform).
template <int NE, class W>
TP_HD __attribute__((always_inline)) uint32_t lean_first(W& w, LeanLds& L, const uint32_t (&val)[NE]) {
    uint32_t pend = 0, first = 0;
    TP_UNROLL for (int e = 0; e < NE; ++e) pend |= (uint32_t)(val[e] != LEAN_DEAD) << e;
    for (uint32_t round = 0; round < LEAN_ROUNDS; ++round) {
        static_assert((2 * LEAN_TAB) % 256 == 0, "whole 16-byte stores per lane");
        TP_UNROLL for (uint32_t k = 0; k < 2 * LEAN_TAB / 256; ++k) w.fill16(&L.tab[256 * k + 4 * w.lane()], 0xFFFFFFFFu);
        w.sync();
        const uint32_t K = round == 0 ? 2654435761u : (round == 1 ? 0x85EBCA77u : (round == 2 ? 0xC2B2AE3Du : (round == 3 ? 0x27D4EB2Fu : 0x165667B1u)));
        uint32_t p[NE];
        TP_UNROLL for (int e = 0; e < NE; ++e) {
            p[e] = (val[e] & LEAN_ID_MASK) * K;
            if ((pend >> e) & 1u) {
                w.lds_min(&L.tab[p[e] >> (32 - LEAN_TAB_BITS)], val[e]);
                w.lds_min(&L.tab[LEAN_TAB + ((p[e] >> (32 - 2 * LEAN_TAB_BITS)) & (LEAN_TAB - 1))], val[e]);
            }
        }
        w.sync();
        TP_UNROLL for (int e = 0; e < NE; ++e) {
            if ((pend >> e) & 1u) {
                const uint32_t va = L.tab[p[e] >> (32 - LEAN_TAB_BITS)], vb = L.tab[LEAN_TAB + ((p[e] >> (32 - 2 * LEAN_TAB_BITS)) & (LEAN_TAB - 1))];
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
TP_HD bool lean_plain_unknown(const uint8_t* p, uint32_t n) {
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

// One record of the rule cache against the evidence bitset (MatchesFacetRule and its kin, Program.cs:777-794).
TP_HD bool lean_rc_match(const uint32_t* ebits, const uint32_t* rc, uint32_t rec) {
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

// The pipeline for one file; false = declined.  rc: the rule cache; fin32: the caption's tag ids (FAST_FIN_MAX entries);
// fired (fired_words words) and facet_choice may be null; with fired the evidence set is counted (tp_verdict::n_evidence).
template <class W>
TP_HD bool lean_file(W& w, const DevTables& t, LeanLds& L, const LeanShared& S, const uint32_t* rc, const uint8_t* text, uint32_t len, FastOut& out,
                     uint32_t* fin32, uint32_t* fired, uint16_t* facet_choice) {
    const uint32_t lane = w.lane();
    if (len > LEAN_LINE_MAX || len == 0) return LEAN_NO(1);
    w.tick(16);
    // ---- the line, 16 bytes per lane: to LDS for the token reads, and scanned for commas in registers -------------------
    const uint32_t shift = (uint32_t)((uintptr_t)text & 15);
    uint32_t T;
    {
        const uint32_t end = shift + len, pieces = (end + 15) >> 4;   // at most 64
        uint32_t x[4] = {0, 0, 0, 0};
        if (lane < pieces) w.load16(text - shift + 16 * lane, x);
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
    LEAN_STOP(1, L.cpos[lane] ^ T)

    // ---- tokens -> interned ids (NormalizeRawTagLine's per-token steps and Distinct, Program.cs:598-602) -----------------
    uint32_t id = NO_ID, f = FL_BLANK;                              // the candidate of this lane and its flags
    uint32_t lit = NO_ID, s0_info = 0xFFFFFFu, s1_info = 0xFFFFFFu, ev_step[2] = {NO_ID, NO_ID};   // what the evidence step reads of its record
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
        LeanCand c;
        c.id = NO_ID; c.flags = FL_BLANK; c.len = 0; c.step_cnt = 0; c.step0 = c.step1 = NO_ID; c.s0_flags = c.s1_flags = c.s0_len = c.s1_len = 0; c.pad0 = c.pad1 = 0;
        if (tok && n > 0) {
            const uint64_t* mk = S.tail_mask[n];
            const uint64_t w0 = ld64u(s + i2) & mk[0], w1 = ld64u(s + i2 + 8) & mk[1], w2 = ld64u(s + i2 + 16) & mk[2], w3 = ld64u(s + i2 + 24) & mk[3];
            const uint64_t h = hash_finish(hash_word(hash_word(hash_word(hash_word(HASH_SEED, w0, 0), w1, 1), w2, 2), w3, 3), n);
            uint32_t slot = (uint32_t)h & t.ht_mask;
            for (;;) {
                const Slot* sl = t.slots + slot;
                const uint32_t slen = sl->slen, val = sl->val, kflags = sl->kflags;
                uint64_t k[4];
                __builtin_memcpy(k, sl->key, 32);
                const Rec rr = sl->rec;
                if (slen == SLOT_EMPTY) { miss = true; break; }
                if (slen == n && ((k[0] ^ w0) | (k[1] ^ w1) | (k[2] ^ w2) | (k[3] ^ w3)) == 0) {
                    if (kflags & SLOT_KEY_BASIC) {
                        id = val; f = rr.flags; lit = rr.lit; s0_info = rr.s0_info; s1_info = rr.s1_info;
                        const bool edge = (rr.flags & FL_EDGE) != 0;
                        c.id = val; c.flags = rr.flags; c.len = rr.len; c.step_cnt = edge ? rr.step_cnt : 0;
                        c.step0 = edge && rr.step_cnt > 0 ? rr.step0 : NO_ID; c.step1 = edge && rr.step_cnt > 1 ? rr.step1 : NO_ID;
                        c.s0_flags = rr.s0_flags; c.s1_flags = rr.s1_flags; c.s0_len = rr.s0_len; c.s1_len = rr.s1_len;
                        if (rr.next != val) c.pad0 = 1;            // NormalizeTag would change it again at the next site
                    } else decline = true;                         // not its own normal form: the general stages normalise it
                    break;
                }
                slot = (slot + 1) & t.ht_mask;
            }
        }
        if (w.ballot(miss) != 0) {
            // in no table: dropped under strict vocabulary, provided it is plain; otherwise it may survive as a tag of its own
            if (miss && !(t.drop_unknown && lean_plain_unknown(s + i2, n))) decline = true;
        }
        // LooksBad's phrases and code fences are looked for in the raw line (Program.cs:592-596): any token holding one declines
        if (id != NO_ID && (f & (FL_PHRASE | FL_FENCE))) decline = true;
        kept = id != NO_ID && (f & FL_LEN2) && !(f & FL_DROP);     // Length >= 2, !ShouldDropTag (Program.cs:600-601)
        if (kept && c.pad0) decline = true;                        // tp_fast.h follows such hops
        if (w.ballot(decline) != 0) return LEAN_NO(5);
        ev_step[0] = c.step0; ev_step[1] = c.step1;
        w.sync();                                                   // every lane has read its token: the records take the line's place
        w.store16((uint8_t*)&L.cand[lane], (const uint32_t(&)[4])c.id);
        w.store16((uint8_t*)&L.cand[lane] + 16, (const uint32_t(&)[4])c.step0);
        w.store16((uint8_t*)&L.cand[lane] + 32, (const uint32_t(&)[4])c.s0_len);
    }
    w.tick(18);
    LEAN_STOP(2, id ^ f ^ lit ^ s0_info ^ (uint32_t)kept)
    bool act;
    {
        const uint32_t v1[1] = {lean_val(lane, id, kept)};
        const uint32_t fb = lean_first<1>(w, L, v1);
        if (fb >> 31) return LEAN_NO(6);
        act = (fb & 1u) != 0;
    }
    if (!act) f = FL_BLANK;
    w.tick(19);
    LEAN_STOP(3, id ^ (uint32_t)act ^ f ^ lit ^ s0_info)

    // ---- candidates (Program.cs:625-655): scores fall with the candidate index, so sorted order is candidate order ----------
    const uint64_t m_act = w.ballot(act);
    const uint32_t n_cand = popc64(m_act), ci = w.rank(m_act);
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
            if (lit != NO_ID) w.lds_or(&L.ebits[lit >> 5], 1u << (lit & 31));
            ev_feet = (f & FL_NFEET) != 0; ev_penis = (f & FL_NPENIS) != 0;
            if ((f & FL_EDGE) && !(f & FL_REACH2)) {
                const uint32_t m0 = t.reach_off[id], m1 = t.reach_off[id + 1];
                if (fired && m1 - m0 > 2) ev_far = true;           // more members than this lane can hand to the count
                for (uint32_t m = m0; m < m1; ++m) {
                    const uint32_t info = t.reach_ent[2 * m + 1], li = info & 0xFFFFFFu;
                    if (li != 0xFFFFFFu) w.lds_or(&L.ebits[li >> 5], 1u << (li & 31));
                    ev_feet |= ((info >> 24) & 1u) != 0;
                    ev_penis |= ((info >> 25) & 1u) != 0;
                    if (m - m0 < 2) ev_id[1 + (m - m0)] = t.reach_ent[2 * m];
                }
            } else if (f & FL_EDGE) {
                const uint32_t i0 = ev_step[0] != NO_ID ? s0_info : 0xFFFFFFu, i1 = ev_step[1] != NO_ID ? s1_info : 0xFFFFFFu;
                if ((i0 & 0xFFFFFFu) != 0xFFFFFFu) w.lds_or(&L.ebits[(i0 & 0xFFFFFFu) >> 5], 1u << (i0 & 31));
                if ((i1 & 0xFFFFFFu) != 0xFFFFFFu) w.lds_or(&L.ebits[(i1 & 0xFFFFFFu) >> 5], 1u << (i1 & 31));
                ev_feet |= (((i0 | i1) >> 24) & 1u) != 0;
                ev_penis |= (((i0 | i1) >> 25) & 1u) != 0;
                ev_id[1] = ev_step[0]; ev_id[2] = ev_step[1];
            }
        }
        ev_feet = w.ballot(ev_feet) != 0;
        ev_penis = w.ballot(ev_penis) != 0;
        if (fired) {                                                // the evidence set's size is a detail output: distinct ids among seeds and their closures
            if (w.ballot(ev_far) != 0) return LEAN_NO(9);
            const uint32_t v3[3] = {lean_val(3 * lane, ev_id[0], seed), lean_val(3 * lane + 1, ev_id[1], seed && ev_id[1] != NO_ID),
                                    lean_val(3 * lane + 2, ev_id[2], seed && ev_id[2] != NO_ID)};
            const uint32_t fb = lean_first<3>(w, L, v3);
            if (fb >> 31) return LEAN_NO(10);
            n_evidence = popc64(w.ballot(fb & 1u)) + popc64(w.ballot(fb & 2u)) + popc64(w.ballot(fb & 4u));
        }
    }
    w.sync();
    w.tick(21);
    LEAN_STOP(5, id ^ f ^ L.ebits[lane & 1] ^ (uint32_t)ev_feet ^ (uint32_t)ev_penis ^ n_evidence)

    // ---- InferActionTags / EnsureRequiredFacets (Program.cs:731-775): both read the evidence only -------------------------
    bool frules_possible;
    uint32_t a_val = LEAN_DEAD, a_fl = 0, a_len = 0;                // lane q: the q-th tag the action rules offer, in rule order
    uint32_t n_act = 0;
    {
        bool lit_r = false, lit_f = false;
        for (uint32_t k = lane; k < t.lit_words; k += 64) {
            const uint32_t e = L.ebits[k];
            lit_r = lit_r || (e & rc[t.rc_rule_mask + k]) != 0;
            lit_f = lit_f || (e & rc[t.rc_frule_mask + k]) != 0;
        }
        frules_possible = t.frules_vacuous || w.ballot(lit_f) != 0;
        bool rules_fired = false;
        if (w.ballot(lit_r) != 0) {                                 // the evidence holds a literal some action rule tests positively: look closer
            // offered tags and fired bits are put together in the (idle) table area: 3 words per tag, then the bitmap
            uint32_t* acts = L.tab;
            uint32_t* firedw = L.tab + 3 * 64;
            static_assert(3 * 64 + FAST_FIRED_WORDS <= 2 * LEAN_TAB, "the action list fits the table area");
            if (lane < t.fired_words) firedw[lane] = 0;
            w.sync();
            bool changes = false;
            for (uint32_t r0 = 0; r0 < t.n_rules; r0 += 64) {
                const uint32_t ri = r0 + lane;
                const bool hit = ri < t.n_rules && lean_rc_match(L.ebits, rc, ri * RC_REC_WORDS);   // MatchesActionRule, Program.cs:777-782
                uint64_t m = w.ballot(hit);
                if (hit) w.lds_or(&firedw[ri >> 5], 1u << (ri & 31));                              // "action:{parent}" (Program.cs:741)
                while (m) {
                    rules_fired = true;
                    const uint32_t* rr = rc + (r0 + ctz64(m)) * RC_REC_WORDS;
                    m &= m - 1;
                    const uint32_t sub_off = rr[1], sub_cnt = rr[3] >> 16;
                    if (rr[3] & 0x8000u) changes = true;           // a tag NormalizeTag changes
                    if (n_act >= 64) return LEAN_NO(11);
                    if (lane == 0) { acts[3 * n_act] = rr[0]; acts[3 * n_act + 1] = rr[8]; acts[3 * n_act + 2] = rr[9]; }
                    ++n_act;
                    for (uint32_t s0 = 0; s0 < sub_cnt; s0 += 64) {
                        const uint32_t si = s0 + lane;
                        const uint32_t srec = t.rc_subs + (sub_off + (si < sub_cnt ? si : 0)) * RC_REC_WORDS;
                        const bool shit = si < sub_cnt && lean_rc_match(L.ebits, rc, srec);        // MatchesSubtypeRule, Program.cs:784-788
                        const uint64_t ms = w.ballot(shit);
                        if (n_act + popc64(ms) > 64) return LEAN_NO(11);
                        if (shit) {
                            const uint32_t bit = t.n_rules + sub_off + si;                           // "subtype:{parent}:{tag}" (Program.cs:748)
                            w.lds_or(&firedw[bit >> 5], 1u << (bit & 31));
                            const uint32_t* sr = rc + srec;
                            if (sr[3] & 0x8000u) changes = true;
                            const uint32_t q = n_act + w.rank(ms);
                            acts[3 * q] = sr[0]; acts[3 * q + 1] = sr[8]; acts[3 * q + 2] = sr[9];
                        }
                        n_act += popc64(ms);
                    }
                }
            }
            if (w.ballot(changes) != 0) return LEAN_NO(11);
            w.sync();
            if (lane < n_act) { a_val = lean_val(640 + lane, acts[3 * lane], true); a_fl = acts[3 * lane + 1]; a_len = acts[3 * lane + 2]; }
            if (fired && lane < t.fired_words) fired[lane] = firedw[lane];
            w.sync();                                               // the table area is read before the next use clears it
        }
        if (fired && !rules_fired && lane < t.fired_words) fired[lane] = 0;
    }
    uint32_t f_val = LEAN_DEAD, f_fl = 0, f_len = 0;               // lane q: the tag facet q asks for
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
            f_fl = fr[3]; f_len = fr[4]; changes = fr[5] != 0;
            uint16_t choice = 0xFFFF;
            if (frules_possible) {
                const uint32_t rule_off = fr[1], rule_cnt = fr[2];
                for (uint32_t q = 0; q < rule_cnt; ++q) {           // first matching rule wins (Program.cs:760-766)
                    const uint32_t p = rule_off + q;
                    if ((L.fhit[p >> 5] >> (p & 31)) & 1u) {
                        const uint32_t* pr = rc + t.rc_frules + p * RC_REC_WORDS;
                        fsel = pr[0]; f_fl = pr[8]; f_len = pr[9]; changes = (pr[3] & 0x8000u) != 0;
                        choice = (uint16_t)q;
                        break;
                    }
                }
            }
            f_val = lean_val(1024 + lane, fsel, true);
            if (facet_choice) facet_choice[lane] = choice;
        }
        if (w.ballot(changes) != 0) return LEAN_NO(12);             // a facet tag NormalizeTag changes
    }
    w.tick(22);
    LEAN_STOP(6, id ^ f ^ f_val ^ f_fl ^ f_len ^ a_val ^ a_fl ^ a_len)

    // ---- SelectTags' bucket fills (Program.cs:661-699) ------------------------------------------------------------------------
    uint32_t nr = t.n_prefix;
    {
        const uint32_t style_cap = style_n < 2 ? 2u : (style_n > 4 ? 4u : style_n);
        const bool ok = nonblank && !bad && !(f & FL_PREFIX);       // AddOne's tests; prefix tags are already in
        const uint32_t bk_own = fl_bucket(f);
        bool taken = false;
        uint32_t pos = 0;
        TP_UNROLL for (uint32_t bk = 0; bk <= 6; ++bk) {           // FillBucket for the six capped buckets, then the final sweep (Program.cs:670-680)
            const bool el = ok && !taken && (bk == 6 || bk_own == bk);
            const uint64_t m = w.ballot(el);
            const uint32_t room = nr < (uint32_t)MAX_TOTAL_TAGS ? (uint32_t)MAX_TOTAL_TAGS - nr : 0u;
            const uint32_t cap = bk == 0 ? 3u : bk == 1 ? 10u : bk == 2 ? 8u : bk == 3 ? 6u : bk == 4 ? 4u : bk == 5 ? style_cap : 64u;
            uint32_t take = popc64(m);
            take = take < cap ? take : cap;
            take = take < room ? take : room;
            const uint32_t rk = w.rank(m);
            if (el && rk < take) { taken = true; pos = nr + rk; }
            nr += take;
        }
        if (nr < 3) return LEAN_NO(7);                              // the minimal fallback line (Program.cs:297-298)
        if (taken) L.inv[pos] = lane;
    }
    w.sync();
    w.tick(20);
    LEAN_STOP(4, id ^ nr ^ L.inv[lane] ^ n_dropped ^ n_cand ^ f_val ^ a_val)

    // ---- the selected tags by position; EvaluateCoverage (Program.cs:1056-1095) --------------------------------------------
    // position p < n_prefix: a prefix tag; else the candidate of lane inv[p]
    const bool in_sel = lane < nr;
    uint32_t p_val = LEAN_DEAD, p_fl = FL_BLANK, p_len = 0;
    uint32_t k_val[2] = {LEAN_DEAD, LEAN_DEAD}, k_fl[2] = {0, 0}, k_len[2] = {0, 0};
    uint32_t k_cnt = 0;
    if (in_sel) {
        if (lane >= t.n_prefix) {
            const LeanCand& c = L.cand[L.inv[lane]];
            p_val = lean_val(lane, c.id, true); p_fl = c.flags; p_len = c.len; k_cnt = c.step_cnt;
            k_val[0] = lean_val(64 + 2 * lane, c.step0, c.step0 != NO_ID); k_val[1] = lean_val(65 + 2 * lane, c.step1, c.step1 != NO_ID);
            k_fl[0] = c.s0_flags; k_fl[1] = c.s1_flags; k_len[0] = c.s0_len; k_len[1] = c.s1_len;
        } else {
            p_val = lean_val(lane, S.prefix_ids[lane], true); p_fl = S.prefix_flags[lane]; p_len = S.prefix_len[lane];
        }
    }
    uint32_t coverage;
    {
        const bool nb = in_sel && !(p_fl & FL_BLANK);
        const uint32_t p_id = p_val & LEAN_ID_MASK;
        const bool has_trigger = w.ballot(nb && p_id == t.trigger_id) != 0, has_girl = w.ballot(nb && p_id == t.girl_lit_id) != 0;
        coverage = ((has_trigger && has_girl) ? TP_COV_IDENTITY : 0) | (w.ballot(nb && (p_fl & FL_HAIR)) ? TP_COV_HAIR : 0) |
                   (w.ballot(nb && (p_fl & FL_EYES)) ? TP_COV_EYES : 0) | (w.ballot(nb && fl_bucket(p_fl) == B_OUTFIT) ? TP_COV_OUTFIT : 0) |
                   (w.ballot(nb && fl_bucket(p_fl) == B_POSE) ? TP_COV_POSE_ACTION : 0) | (w.ballot(nb && fl_bucket(p_fl) == B_BACKGROUND) ? TP_COV_BACKGROUND : 0);
    }
    // A selected tag whose steps lead further (no FL_LEAF2): the steps' own records give their flags and lengths and, when a
    // step is itself settled (FL_LEAF2), the tags its visit appends in the next round of ExpandImpliedTags (Program.cs:1160-1179);
    // at most two of those per selected tag, at most two steps, and nothing beyond them, or the file is declined.
    const bool deep = in_sel && (p_fl & FL_EDGE) && !(p_fl & FL_LEAF2);
    const bool any_deep = w.ballot(deep) != 0;
    if (any_deep) {
        w.sync();                                                   // the candidates' records have been read: their place is taken by the deeper tags
        bool bad_deep = deep && k_cnt > 2;
        LeanDeep g[2] = {{LEAN_DEAD, 0, 0, 0}, {LEAN_DEAD, 0, 0, 0}};
        if (deep && !bad_deep) {
            uint32_t ng = 0;
            for (uint32_t c = 0; c < 2; ++c) {
                if (k_val[c] == LEAN_DEAD) continue;
                const uint32_t kid = k_val[c] & LEAN_ID_MASK;
                const Rec rk = t.rec[kid];
                if (rk.next != kid || (rk.flags & FL_BLANK)) bad_deep = true;
                k_fl[c] = rk.flags; k_len[c] = rk.len;
                if (rk.flags & FL_EDGE) {
                    if (!(rk.flags & FL_LEAF2) || ng + rk.step_cnt > 2) { bad_deep = true; continue; }
                    for (uint32_t j = 0; j < rk.step_cnt; ++j) {
                        g[ng].val = lean_val(320 + 4 * lane + 2 * c + j, j == 0 ? rk.step0 : rk.step1, true);
                        g[ng].flags = j == 0 ? rk.s0_flags : rk.s1_flags; g[ng].len = j == 0 ? rk.s0_len : rk.s1_len;
                        ++ng;
                    }
                }
            }
        }
        if (w.ballot(bad_deep) != 0) return LEAN_NO(15);
        L.deep[0][lane] = g[0]; L.deep[1][lane] = g[1];
        w.sync();
    }

    // ---- the final set in the reference's insertion order (Program.cs:303-307): the selected tags, their implied tags by
    // position (and those one level further, by their parents' positions), the action tags by rule, the facet tags by
    // facet; an id already present is not added again.  Then the verdicts on the final list (Program.cs:309-329),
    // GetUnknownTags (:817-824) and the caption's length (:971-974), element by element ----------------------------------------
    const bool wide = any_deep || n_act != 0;                       // the usual file has the first four kinds of element only
    uint32_t fb;
    if (wide) {
        const uint32_t v7[7] = {p_val, k_val[0], k_val[1], f_val, any_deep ? L.deep[0][lane].val : LEAN_DEAD, any_deep ? L.deep[1][lane].val : LEAN_DEAD, a_val};
        fb = lean_first<7>(w, L, v7);
    } else {
        const uint32_t v4[4] = {p_val, k_val[0], k_val[1], f_val};
        fb = lean_first<4>(w, L, v4);
    }
    if (fb >> 31) return LEAN_NO(13);
    const uint64_t mk0 = w.ballot(fb & 2u), mk1 = w.ballot(fb & 4u), mf = w.ballot(fb & 8u);
    const uint32_t grown1 = popc64(mk0) + popc64(mk1), nfac = popc64(mf);
    const uint32_t pos_k0 = nr + w.rank(mk0) + w.rank(mk1), rank_f = w.rank(mf);
    uint32_t grown2 = 0, nact = 0, pos_g0 = 0, rank_a = 0;
    if (wide) {
        const uint64_t mg0 = w.ballot(fb & 16u), mg1 = w.ballot(fb & 32u), ma = w.ballot(fb & 64u);
        grown2 = popc64(mg0) + popc64(mg1); nact = popc64(ma);
        pos_g0 = nr + grown1 + w.rank(mg0) + w.rank(mg1); rank_a = w.rank(ma);
    }
    const uint32_t nf = nr + grown1 + grown2 + nact + nfac;
    if (nf > FAST_FIN_MAX) return LEAN_NO(14);
    w.tick(23);
    LEAN_STOP(7, pos_k0 ^ rank_f ^ nf ^ coverage ^ k_val[0] ^ k_val[1] ^ k_fl[0] ^ k_len[1] ^ fb)

    bool face = false, outfit = false, safe = false, foot = false;
    uint32_t bytes = 0, unknown = 0, safe_key = 0xFFFFFFFFu;
    auto emit = [&](bool first, uint32_t val, uint32_t fl, uint32_t ln, uint32_t at) {
        if (first) {
            const uint32_t bk = fl_bucket(fl), eid = val & LEAN_ID_MASK;
            face |= bk == B_FACE;                                  // MissingCoreBuckets, Program.cs:1097-1101
            outfit |= bk == B_OUTFIT;
            if (fl & FL_SAFE) { safe = true; const uint32_t key = (at << 8) | ((fl >> FL_SAFE_IDX_SHIFT) & 31u); safe_key = key < safe_key ? key : safe_key; }
            foot |= eid == t.footjob_n;
            bytes += ln;
            unknown += (fl & FL_KNOWN) ? 0u : 1u;
            fin32[at] = eid;
        }
    };
    emit((fb & 1u) != 0, p_val, p_fl, p_len, lane);
    emit((fb & 2u) != 0, k_val[0], k_fl[0], k_len[0], pos_k0);
    emit((fb & 4u) != 0, k_val[1], k_fl[1], k_len[1], pos_k0 + ((fb >> 1) & 1u));
    if (wide) {
        if (any_deep) {
            const LeanDeep g0 = L.deep[0][lane], g1 = L.deep[1][lane];
            emit((fb & 16u) != 0, g0.val, g0.flags, g0.len, pos_g0);
            emit((fb & 32u) != 0, g1.val, g1.flags, g1.len, pos_g0 + ((fb >> 4) & 1u));
        }
        emit((fb & 64u) != 0, a_val, a_fl, a_len, nr + grown1 + grown2 + rank_a);
    }
    emit((fb & 8u) != 0, f_val, f_fl, f_len, nr + grown1 + grown2 + nact + rank_f);
    uint32_t flags = 0;
    if (!(w.ballot(face) && w.ballot(outfit))) flags |= TP_F_LOW;
    int safe_hit = -1;
    if (t.safe_mode && w.ballot(safe) != 0) {                       // first final tag in SafeModeRejectTags (Program.cs:942-945)
        if (lane == 0) L.scratch[0] = 0xFFFFFFFFu;
        w.sync();
        if (safe) w.lds_min(&L.scratch[0], safe_key);
        w.sync();
        safe_hit = (int)(L.scratch[0] & 255u);
        flags |= TP_F_SAFE_REJECT;
    }
    bytes = w.sum(bytes);
    unknown = t.known_active ? w.sum(unknown) : 0u;
    if (ev_feet && ev_penis) { flags |= TP_F_ELIGIBLE; if (w.ballot(foot) != 0) flags |= TP_F_FOOTJOB; }
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
    out.n_final = nf;
    out.caption_len = bytes + 2 * (nf - 1) + t.newline_len;
    w.tick(24);
    return out.caption_len <= FAST_CAPTION_MAX;
}

}  // namespace tp
