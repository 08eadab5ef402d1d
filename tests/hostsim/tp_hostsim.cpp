// TEST-ONLY harness: compiles the kernels' per-file pipeline source (tp_pipeline.h) and the
// table compiler for the HOST so that the pipeline LOGIC can be checked against the oracle
// in this GPU-less container.  It is not part of the product: nothing under
// imagecaptioner_amd/ loads it, the C ABI never routes here, and the GPU tests do not use it.
// The lane-interleaved workspace layout is kept exactly as the kernel uses it (lane = file & 63).
#define TP_LEAN_STATS 1
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tagpipe.h"
#include "../../imagecaptioner_amd/csrc/tp_fast.h"
#include "../../imagecaptioner_amd/csrc/tp_lean.h"
#include "../../imagecaptioner_amd/csrc/tp_pipeline.h"
#include "wave_emu.h"
#include "../../imagecaptioner_amd/csrc/tp_tables.hpp"

using namespace tp;

extern "C" { unsigned long long g_lean_declines[32]; }   // why the lean stage declined files (reason = position of the test in lean_file)

static const uint8_t* cls_table() {
    struct Tab { uint8_t v[256]; Tab() { for (uint32_t i = 0; i < 256; ++i) v[i] = byte_class(i); } };
    static const Tab tab;                                          // initialised once, thread-safe
    return tab.v;
}

namespace {
thread_local std::string g_err;

struct Sim {
    HostTables ht;
    DevTables dt;
};

struct SimOut {
    std::vector<uint8_t> captions;
    std::vector<uint64_t> offsets, hist;
    std::vector<tp_verdict> verdicts;
    std::vector<uint32_t> fired;
    std::vector<uint16_t> facets;
    tp_counters ctr{};
};

uint32_t commas(const uint8_t* p, uint64_t n) { uint32_t c = 0; for (uint64_t i = 0; i < n; ++i) c += p[i] == ','; return c; }

// Runs files [0,n) wave by wave; `emit` receives each file's final handle list.
void run(const Sim& s, const uint8_t* arena, const uint64_t* offsets, const uint64_t* tag_begin, const double* scores, const uint8_t* sources, uint64_t n,
         int mode, SimOut& o, std::vector<std::vector<H>>* fin_lists, const uint64_t* secondary_begin = nullptr, int secondary_cap = 0) {
    const DevTables& t = s.dt;
    o.offsets.assign(n + 1, 0);
    o.verdicts.assign(n, tp_verdict{});
    o.fired.assign(std::max<size_t>(n * t.fired_words, 1), 0);   // never empty: a non-null pointer asks for the detail outputs
    o.facets.assign(n * t.n_facets, 0xFFFF);
    o.hist.assign((size_t)t.n_public + 1, 0);
    std::memset(&o.ctr, 0, sizeof o.ctr);
    if (fin_lists) fin_lists->assign(n, {});
    for (uint64_t w0 = 0; w0 < n; w0 += 64) {
        uint64_t w1 = std::min<uint64_t>(n, w0 + 64);
        uint32_t T = 1, LB = 0;
        for (uint64_t f = w0; f < w1; ++f) {
            uint32_t tf, lf;
            if (tag_begin) { tf = (uint32_t)(tag_begin[f + 1] - tag_begin[f]); lf = (uint32_t)(offsets[tag_begin[f + 1]] - offsets[tag_begin[f]]); }
            else { lf = (uint32_t)(offsets[f + 1] - offsets[f]); tf = (commas(arena + offsets[f], lf) + 1) * t.split_grow; }
            T = std::max(T, tf); LB = std::max(LB, lf);
        }
        WsLayout L = ws_layout(T, LB, t.final_cap, t.evidence_grow, t.lit_words);
        std::vector<uint8_t> wsmem(L.total, 0xCD);
        for (uint64_t f = w0; f < w1; ++f) {
            FileWs ws = ws_bind(wsmem.data(), L, (uint32_t)(f & 63));
            FileIn in{};
            if (tag_begin) { in.arena = arena; in.tag_offsets = offsets; in.tag_lo = tag_begin[f]; in.tag_hi = tag_begin[f + 1]; in.scores = scores; in.sources = sources;
                             in.merged = secondary_begin != nullptr; in.tag_mid = secondary_begin ? secondary_begin[f] : in.tag_hi; in.secondary_cap = secondary_cap; }
            else { in.text = arena + offsets[f]; in.len = (uint32_t)(offsets[f + 1] - offsets[f]); }
            FileOut fo;
            process_file(t, in, ws, fo, o.fired.data() + f * t.fired_words, t.n_facets ? &o.facets[f * t.n_facets] : nullptr, mode);
            o.verdicts[f] = fo.verdict;
            if (fin_lists) for (uint32_t k = 0; k < fo.n_final; ++k) (*fin_lists)[f].push_back(ws.fin[k]);
            // caption bytes (what k_emit writes)
            size_t before = o.captions.size();
            if (mode == 0 && fo.caption_len) {
                for (uint32_t k = 0; k < fo.n_final; ++k) {
                    H h = ws.fin[k];
                    if (k) { o.captions.push_back(','); o.captions.push_back(' '); }
                    if (h_is_dyn(h)) { o.captions.insert(o.captions.end(), ws.norm + h_off(h), ws.norm + h_off(h) + h_len(h)); o.hist[t.n_public]++; }
                    else { const uint8_t* p = str_ptr(t, h_id(h)); o.captions.insert(o.captions.end(), p, p + str_len(t, h_id(h))); o.hist[std::min<uint32_t>(h_id(h), t.n_public)]++; }
                }
                o.captions.insert(o.captions.end(), t.newline, t.newline + t.newline_len);
                if (o.captions.size() - before != fo.caption_len) { g_err = "caption_len mismatch"; }
            }
            o.offsets[f + 1] = o.captions.size();
            uint8_t fl = fo.verdict.flags;
            o.ctr.files++;
            o.ctr.low += (fl & TP_F_LOW) != 0; o.ctr.action_eligible += (fl & TP_F_ELIGIBLE) != 0; o.ctr.action_footjob += (fl & TP_F_FOOTJOB) != 0;
            o.ctr.fallback += (fl & TP_F_FALLBACK) != 0; o.ctr.line_rejected += (fl & TP_F_LINE_REJECTED) != 0; o.ctr.unsupported += (fl & TP_F_UNSUPPORTED) != 0;
            o.ctr.safe_rejected += (fl & TP_F_SAFE_REJECT) != 0;
        }
    }
}
}  // namespace

extern "C" {

const char* sim_last_error(void) { return g_err.c_str(); }

// learned: strings to intern behind the configuration's own (what the library learns from its input), '\n'-separated; may be null
void* sim_create_learned(const tp_config* cfg, const char* learned, uint64_t learned_len) {
    try {
        std::unique_ptr<Sim> s(new Sim());
        std::vector<std::string> extra;
        if (learned) {
            std::string cur;
            for (uint64_t i = 0; i < learned_len; ++i) { if (learned[i] == '\n') { if (!cur.empty()) extra.push_back(cur); cur.clear(); } else cur.push_back(learned[i]); }
            if (!cur.empty()) extra.push_back(cur);
        }
        compile_tables(*cfg, s->ht, learned ? &extra : nullptr);
        s->dt = bind_host_pointers(s->ht);
        // minimal fallback line, as tp_create evaluates it (mode 1, weight 1.0, source OpenAI)
        DevTables saved = s->dt;
        s->dt.line_weight = 1.0;
        s->dt.line_source = TP_SRC_OPENAI;
        uint64_t offs[2] = {0, s->ht.fallback_line.size()};
        SimOut o;
        std::vector<std::vector<H>> fin;
        std::vector<uint8_t> line = s->ht.fallback_line;
        line.resize(line.size() + 16);
        run(*s, line.data(), offs, nullptr, nullptr, nullptr, 1, 1, o, &fin);
        s->dt = saved;
        std::vector<uint32_t> ids;
        for (H h : fin[0]) ids.push_back((uint32_t)h);
        set_fallback(s->ht, s->dt, ids.data(), (uint32_t)ids.size());
        return s.release();
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void* sim_create(const tp_config* cfg) { return sim_create_learned(cfg, nullptr, 0); }
void sim_destroy(void* h) { delete (Sim*)h; }

int sim_info(void* h, tp_table_info* out) {
    Sim* s = (Sim*)h;
    std::memset(out, 0, sizeof *out);
    out->n_strings = s->dt.n_public; out->n_action_rules = s->dt.n_rules; out->n_action_subtypes = s->dt.n_subs; out->n_facets = s->dt.n_facets;
    out->fired_words = s->dt.fired_words; out->n_known_tags = s->ht.n_known; out->strict_known = s->ht.strict; out->known_found = s->ht.known_found;
    return 0;
}
// tp_format_top_tags' text (the same top_tags_text the library calls); returns the full size, writes at most `capacity` bytes
uint64_t sim_format_top_tags(void* h, const uint64_t* histogram, uint8_t* out, uint64_t capacity) {
    const std::string text = top_tags_text(((Sim*)h)->ht, histogram);
    if (out && capacity) std::memcpy(out, text.data(), std::min<uint64_t>(capacity, text.size()));
    return text.size();
}

// One tagger reply through the library's decoder (decode_joytag_reply).  Returns the number of tags, or -1 when a buffer is too
// small; arena / offsets (tags + 1) / scores are the caller's.
int64_t sim_decode_reply(void* h, const uint8_t* json, uint64_t len, uint8_t* arena, uint64_t arena_cap, uint64_t* offsets, double* scores, uint64_t tag_cap,
                         uint8_t* status, uint8_t* has_score) {
    Sim& s = *(Sim*)h;
    static thread_local const Sim* indexed = nullptr;
    static thread_local std::unordered_map<std::string, uint32_t> ids;
    if (indexed != &s) { ids = index_strings(s.ht); indexed = &s; }
    DecodedReplies d;
    decode_joytag_reply(s.ht, ids, json, (size_t)len, d, *status, *has_score);
    if (d.arena.size() > arena_cap || d.scores.size() > tag_cap) return -1;
    if (!d.arena.empty()) std::memcpy(arena, d.arena.data(), d.arena.size());
    std::memcpy(offsets, d.tag_offsets.data(), d.tag_offsets.size() * 8);
    if (!d.scores.empty()) std::memcpy(scores, d.scores.data(), d.scores.size() * 8);
    return (int64_t)d.scores.size();
}

const char* sim_string(void* h, uint32_t id) { return ((Sim*)h)->ht.strings[id].c_str(); }

struct sim_result {
    uint64_t n_files; const uint8_t* captions; const uint64_t* offsets; const tp_verdict* verdicts; tp_counters counters;
    const uint64_t* histogram; const uint32_t* fired; const uint16_t* facets; void* opaque;
};
static int finish(SimOut* o, uint64_t n, sim_result* r) {
    r->n_files = n; r->captions = o->captions.data(); r->offsets = o->offsets.data(); r->verdicts = o->verdicts.data(); r->counters = o->ctr;
    r->histogram = o->hist.data(); r->fired = o->fired.data(); r->facets = o->facets.data(); r->opaque = o;
    return g_err.empty() ? 0 : -1;
}
int sim_run_lines(void* h, const uint8_t* arena, const uint64_t* offsets, uint64_t n, sim_result* r) {
    g_err.clear();
    SimOut* o = new SimOut();
    run(*(Sim*)h, arena, offsets, nullptr, nullptr, nullptr, n, 0, *o, nullptr);
    return finish(o, n, r);
}
int sim_run_scored(void* h, const uint8_t* arena, const uint64_t* tag_offsets, const uint64_t* tag_begin, const double* scores, const uint8_t* sources, uint64_t n,
                   sim_result* r) {
    g_err.clear();
    SimOut* o = new SimOut();
    run(*(Sim*)h, arena, tag_offsets, tag_begin, scores, sources, n, 0, *o, nullptr);
    return finish(o, n, r);
}
int sim_run_merged(void* h, const uint8_t* arena, const uint64_t* tag_offsets, const uint64_t* tag_begin, const uint64_t* secondary_begin, const double* scores,
                   int secondary_cap, uint64_t n, sim_result* r) {
    g_err.clear();
    SimOut* o = new SimOut();
    run(*(Sim*)h, arena, tag_offsets, tag_begin, scores, nullptr, n, 0, *o, nullptr, secondary_begin, secondary_cap);
    return finish(o, n, r);
}
// The GPU flow: every file first goes through the wave-per-file kernel body (tp_fast.h) on an emulated wavefront;
// files it declines take the general pipeline.  *n_general receives how many did.
static int run_lines_staged(void* h, const uint8_t* arena_in, const uint64_t* offsets, uint64_t n, sim_result* r, uint64_t* n_general, bool lean, bool details,
                            uint64_t* n_lean);
int sim_run_lines_fast(void* h, const uint8_t* arena_in, const uint64_t* offsets, uint64_t n, sim_result* r, uint64_t* n_general) {
    return run_lines_staged(h, arena_in, offsets, n, r, n_general, false, true, nullptr);
}
// The same with the lean stage (tp_lean.h) in front, as the library runs it.  details == 0: the run of a caller that takes
// no detail arrays (the evidence set is then not counted: tp_verdict::n_evidence is 0xFFFF for files of the wave stages).
// *n_lean receives how many files the lean stage took.
int sim_run_lines_lean(void* h, const uint8_t* arena_in, const uint64_t* offsets, uint64_t n, int details, sim_result* r, uint64_t* n_general, uint64_t* n_lean) {
    return run_lines_staged(h, arena_in, offsets, n, r, n_general, true, details != 0, n_lean);
}
static int run_lines_staged(void* h, const uint8_t* arena_in, const uint64_t* offsets, uint64_t n, sim_result* r, uint64_t* n_general, bool lean, bool details,
                            uint64_t* n_lean) {
    g_err.clear();
    Sim& s = *(Sim*)h;
    const DevTables& t = s.dt;
    SimOut* o = new SimOut();
    // the kernel reads whole aligned 16-byte pieces around a line: give the arena slack on both sides
    std::vector<uint8_t> padded((size_t)offsets[n] + 128, 0);
    uint8_t* arena = padded.data() + 64;
    std::memcpy(arena, arena_in, (size_t)offsets[n]);
    SimOut general;
    std::vector<std::vector<H>> unused;
    o->offsets.assign(n + 1, 0);
    o->verdicts.assign(n, tp_verdict{});
    o->fired.assign(std::max<size_t>(n * t.fired_words, 1), 0);
    o->facets.assign(n * t.n_facets, 0xFFFF);
    o->hist.assign((size_t)t.n_public + 1, 0);
    uint64_t slow = 0, n_lean_files = 0;
    emu::Shared sh;
    for (uint64_t f = 0; f < n; ++f) {
        static FastLds lds;
        std::memset(&lds, 0xCD, sizeof lds);
        std::vector<uint32_t> fin32(FAST_FIN_MAX, 0xDEADBEEF);
        bool done[64];
        FastOut outs[64];
        const uint8_t* text = arena + offsets[f];
        uint32_t len = (uint32_t)(offsets[f + 1] - offsets[f]);
        bool ok = t.fast_ok && len <= FAST_LINE_MAX;
        bool try_long = t.fast_ok && len <= LONG_LINE_MAX;
        uint32_t* fired_f = details ? o->fired.data() + f * t.fired_words : nullptr;
        bool lean_done = false;
        if (lean && lean_tables_ok(t) && len <= LEAN_LINE_MAX) {
            static LeanFrontLds flds;
            static LeanLds llds;
            static LeanShared lshared;
            std::memset((void*)&flds, 0xCD, sizeof flds);
            std::memset((void*)&llds, 0xCD, sizeof llds);
            lean_shared_init(t, lshared, 0, 1);
            std::vector<uint32_t> row(64, 0xDEADBEEF);
            sh.run([&](int lane) {
                emu::Wave w{&sh, (uint32_t)lane};
                uint32_t piece[4];
                lean_fetch(w, text, len, piece);
                // SIM_LEAN_AUDIT=1: the lean stage also lists what the audit row needs (into scratch arrays: which files it then
                // declines is what a test may want to know)
                static const bool with_audit = [] { const char* e = std::getenv("SIM_LEAN_AUDIT"); return e && e[0] == '1'; }();
                static uint32_t aud_row[AUD_STRIDE], aud_cnt[AUDIT_COLS], aud_len[AUDIT_COLS], aud_bits;
                const LeanAudit aud = with_audit ? LeanAudit{aud_row, aud_cnt, aud_len, &aud_bits} : LeanAudit{nullptr, nullptr, nullptr, nullptr};
                done[lane] = lean_file(w, t, flds, llds, lshared, t.rc, text, len, piece, row.data(), outs[lane], fin32.data(), fired_f, t.n_facets ? &o->facets[f * t.n_facets] : nullptr, aud);
            });
            for (int l = 1; l < 64; ++l) if (done[l] != done[0]) { g_err = "lanes disagree on the lean verdict"; }
            lean_done = done[0];
            if (lean_done) ++n_lean_files;
        }
        if (lean_done) ok = true;
        else if (ok) {
            // first stage (one element per lane), then the second (two per lane) for what the first declines
            for (int stage = 1; stage <= 2; ++stage) {
                std::memset(&lds, 0xCD, sizeof lds);
                sh.run([&](int lane) {
                    emu::Wave w{&sh, (uint32_t)lane};
                    uint32_t* fired = fired_f;
                    uint16_t* facets = t.n_facets ? &o->facets[f * t.n_facets] : nullptr;
                    done[lane] = stage == 1 ? fast_file<1, 2>(w, t, lds, t.rc, cls_table(), text, len, outs[lane], fin32.data(), fired, facets)
                                            : fast_file<2, 2>(w, t, lds, t.rc, cls_table(), text, len, outs[lane], fin32.data(), fired, facets);
                });
                for (int l = 1; l < 64; ++l) if (done[l] != done[0]) { g_err = "lanes disagree on the fast-path verdict"; }
                if (done[0]) break;
            }
            ok = done[0];
        }
        if (!ok && try_long) {                                       // third stage: long files
            static LongLds big;
            std::memset((void*)&big, 0xCD, sizeof big);
            sh.run([&](int lane) {
                emu::Wave w{&sh, (uint32_t)lane};
                done[lane] = long_file(w, t, big, t.rc, cls_table(), text, len, outs[lane], fin32.data(), fired_f,
                                       t.n_facets ? &o->facets[f * t.n_facets] : nullptr);
            });
            for (int l = 1; l < 64; ++l) if (done[l] != done[0]) { g_err = "lanes disagree on the long-file verdict"; }
            ok = done[0];
        }
        size_t before = o->captions.size();
        if (ok) {
            const FastOut& fo = outs[0];
            o->verdicts[f] = fo.verdict;
            for (uint32_t k = 0; k < fo.n_final; ++k) {
                uint32_t id = fin32[k];
                if (k) { o->captions.push_back(','); o->captions.push_back(' '); }
                const uint8_t* p = str_ptr(t, id);
                o->captions.insert(o->captions.end(), p, p + str_len(t, id));
                o->hist[std::min<uint32_t>(id, t.n_public)]++;
            }
            o->captions.insert(o->captions.end(), t.newline, t.newline + t.newline_len);
            if (o->captions.size() - before != fo.caption_len) g_err = "fast caption_len mismatch";
        } else {
            ++slow;
            uint64_t offs2[2] = {0, len};
            SimOut one;
            run(s, text, offs2, nullptr, nullptr, nullptr, 1, 0, one, nullptr);
            o->verdicts[f] = one.verdicts[0];
            for (uint32_t k = 0; k < t.fired_words; ++k) o->fired[f * t.fired_words + k] = one.fired[k];
            for (uint32_t k = 0; k < t.n_facets; ++k) o->facets[f * t.n_facets + k] = one.facets[k];
            o->captions.insert(o->captions.end(), one.captions.begin(), one.captions.end());
            for (size_t k = 0; k < one.hist.size(); ++k) o->hist[k] += one.hist[k];
        }
        o->offsets[f + 1] = o->captions.size();
        uint8_t fl = o->verdicts[f].flags;
        o->ctr.files++;
        o->ctr.low += (fl & TP_F_LOW) != 0; o->ctr.action_eligible += (fl & TP_F_ELIGIBLE) != 0; o->ctr.action_footjob += (fl & TP_F_FOOTJOB) != 0;
        o->ctr.fallback += (fl & TP_F_FALLBACK) != 0; o->ctr.line_rejected += (fl & TP_F_LINE_REJECTED) != 0; o->ctr.unsupported += (fl & TP_F_UNSUPPORTED) != 0;
        o->ctr.safe_rejected += (fl & TP_F_SAFE_REJECT) != 0;
    }
    if (n_general) *n_general = slow;
    if (n_lean) *n_lean = n_lean_files;
    return finish(o, n, r);
}

// The GPU flow for scored candidates: the wave-per-file kernel body with the scored front on an emulated wavefront, the
// general pipeline for the files it declines.  secondary_begin == nullptr: one reply per file with per-tag sources.
int sim_run_scored_fast(void* h, const uint8_t* arena_in, const uint64_t* tag_offsets, const uint64_t* tag_begin, const uint64_t* secondary_begin,
                        const double* scores, const uint8_t* sources, int secondary_cap, uint64_t n, sim_result* r, uint64_t* n_general) {
    g_err.clear();
    Sim& s = *(Sim*)h;
    const DevTables& t = s.dt;
    SimOut* o = new SimOut();
    const uint64_t n_tags = tag_begin[n];
    std::vector<uint8_t> padded((size_t)tag_offsets[n_tags] + 128, 0);
    uint8_t* arena = padded.data() + 64;
    std::memcpy(arena, arena_in, (size_t)tag_offsets[n_tags]);
    o->offsets.assign(n + 1, 0);
    o->verdicts.assign(n, tp_verdict{});
    o->fired.assign(std::max<size_t>(n * t.fired_words, 1), 0);
    o->facets.assign(n * t.n_facets, 0xFFFF);
    o->hist.assign((size_t)t.n_public + 1, 0);
    uint64_t slow = 0;
    emu::Shared sh;
    for (uint64_t f = 0; f < n; ++f) {
        static FastLds lds;
        std::memset(&lds, 0xCD, sizeof lds);
        std::vector<uint32_t> fin32(FAST_FIN_MAX, 0xDEADBEEF);
        bool done[64];
        FastOut outs[64];
        bool ok = t.fast_ok;
        if (ok) {
            sh.run([&](int lane) {
                emu::Wave w{&sh, (uint32_t)lane};
                ScoredFront<emu::Wave> front;
                front.in = ScoredIn{arena, tag_offsets, tag_begin[f], secondary_begin ? secondary_begin[f] : tag_begin[f + 1], tag_begin[f + 1], scores, sources,
                                    secondary_begin != nullptr, secondary_cap};
                done[lane] = fast_file<2, 2, emu::Wave, ScoredFront<emu::Wave>>(w, t, lds, t.rc, cls_table(), nullptr, 0, outs[lane], fin32.data(),
                                                                                 o->fired.data() + f * t.fired_words, t.n_facets ? &o->facets[f * t.n_facets] : nullptr,
                                                                                 false, &front);
            });
            for (int l = 1; l < 64; ++l) if (done[l] != done[0]) { g_err = "lanes disagree on the scored fast-path verdict"; }
            ok = done[0];
        }
        size_t before = o->captions.size();
        if (ok) {
            const FastOut& fo = outs[0];
            o->verdicts[f] = fo.verdict;
            for (uint32_t k = 0; k < fo.n_final; ++k) {
                uint32_t id = fin32[k];
                if (k) { o->captions.push_back(','); o->captions.push_back(' '); }
                const uint8_t* p = str_ptr(t, id);
                o->captions.insert(o->captions.end(), p, p + str_len(t, id));
                o->hist[std::min<uint32_t>(id, t.n_public)]++;
            }
            o->captions.insert(o->captions.end(), t.newline, t.newline + t.newline_len);
            if (o->captions.size() - before != fo.caption_len) g_err = "fast caption_len mismatch";
        } else {
            ++slow;
            SimOut one;
            run(s, arena, tag_offsets, tag_begin + f, scores, sources, 1, 0, one, nullptr, secondary_begin ? secondary_begin + f : nullptr, secondary_cap);
            o->verdicts[f] = one.verdicts[0];
            for (uint32_t k = 0; k < t.fired_words; ++k) o->fired[f * t.fired_words + k] = one.fired[k];
            for (uint32_t k = 0; k < t.n_facets; ++k) o->facets[f * t.n_facets + k] = one.facets[k];
            o->captions.insert(o->captions.end(), one.captions.begin(), one.captions.end());
            for (size_t k = 0; k < one.hist.size(); ++k) o->hist[k] += one.hist[k];
        }
        o->offsets[f + 1] = o->captions.size();
        uint8_t fl = o->verdicts[f].flags;
        o->ctr.files++;
        o->ctr.low += (fl & TP_F_LOW) != 0; o->ctr.action_eligible += (fl & TP_F_ELIGIBLE) != 0; o->ctr.action_footjob += (fl & TP_F_FOOTJOB) != 0;
        o->ctr.fallback += (fl & TP_F_FALLBACK) != 0; o->ctr.line_rejected += (fl & TP_F_LINE_REJECTED) != 0; o->ctr.unsupported += (fl & TP_F_UNSUPPORTED) != 0;
        o->ctr.safe_rejected += (fl & TP_F_SAFE_REJECT) != 0;
    }
    if (n_general) *n_general = slow;
    return finish(o, n, r);
}

// Existing-caption triage in the GPU flow: the three wave stages in triage mode, then the general pipeline in mode 2.
// which: 0 the flow as the library runs it, 1 general pipeline only.  bad[i]: 1 bad, 0 keep, 2 unsupported.
int sim_triage_lines(void* h, const uint8_t* arena_in, const uint64_t* offsets, uint64_t n, int which, uint8_t* bad, uint64_t* n_general) {
    g_err.clear();
    Sim& s = *(Sim*)h;
    const DevTables& t = s.dt;
    std::vector<uint8_t> padded((size_t)offsets[n] + 128, 0);
    uint8_t* arena = padded.data() + 64;
    std::memcpy(arena, arena_in, (size_t)offsets[n]);
    uint64_t slow = 0;
    emu::Shared sh;
    for (uint64_t f = 0; f < n; ++f) {
        static FastLds lds;
        std::vector<uint32_t> fin32(FAST_FIN_MAX, 0xDEADBEEF);
        bool done[64];
        FastOut outs[64];
        const uint8_t* text = arena + offsets[f];
        uint32_t len = (uint32_t)(offsets[f + 1] - offsets[f]);
        bool ok = false;
        if (which == 0 && t.fast_ok && len <= FAST_LINE_MAX) {
            for (int stage = 1; stage <= 2 && !ok; ++stage) {
                std::memset(&lds, 0xCD, sizeof lds);
                sh.run([&](int lane) {
                    emu::Wave w{&sh, (uint32_t)lane};
                    done[lane] = stage == 1 ? fast_file<1, 2>(w, t, lds, t.rc, cls_table(), text, len, outs[lane], fin32.data(), nullptr, nullptr, true)
                                            : fast_file<2, 2>(w, t, lds, t.rc, cls_table(), text, len, outs[lane], fin32.data(), nullptr, nullptr, true);
                });
                for (int l = 1; l < 64; ++l) if (done[l] != done[0] || outs[l].verdict.flags != outs[0].verdict.flags) g_err = "lanes disagree in triage";
                ok = done[0];
            }
        }
        if (which == 0 && !ok && t.fast_ok && len <= LONG_LINE_MAX) {
            static LongLds big;
            std::memset((void*)&big, 0xCD, sizeof big);
            sh.run([&](int lane) {
                emu::Wave w{&sh, (uint32_t)lane};
                done[lane] = long_file(w, t, big, t.rc, cls_table(), text, len, outs[lane], fin32.data(), nullptr, nullptr, true);
            });
            for (int l = 1; l < 64; ++l) if (done[l] != done[0] || outs[l].verdict.flags != outs[0].verdict.flags) g_err = "lanes disagree in long-file triage";
            ok = done[0];
        }
        uint8_t fl;
        if (ok) fl = outs[0].verdict.flags;
        else {
            ++slow;
            uint64_t offs2[2] = {0, len};
            SimOut one;
            run(s, text, offs2, nullptr, nullptr, nullptr, 1, 2, one, nullptr);
            fl = one.verdicts[0].flags;
        }
        bad[f] = (fl & TP_F_UNSUPPORTED) ? 2 : (fl & TP_F_TRIAGE_BAD) ? 1 : 0;
    }
    if (n_general) *n_general = slow;
    return g_err.empty() ? 0 : -1;
}

// The audit row's three tag lists (process_file mode 3 + audit_elem, i.e. what k_pipeline and k_emit_audit do on the GPU).
// Line ingest when tag_begin is null, scored ingest otherwise (two replies to merge when secondary_begin is given).
// offs3 has AUDIT_COLS (5) n + 1 entries; returns the text size, or -1 when cap is too small (offs3 is filled either way).
// reply_flags: per file, bit 0 = ParseJoyTags sorted the (primary) reply (tp_replies::has_score); null = every reply was sorted.
long long sim_run_audit_flags(void* h, const uint8_t* arena, const uint64_t* offsets, const uint64_t* tag_begin, const uint64_t* secondary_begin, const double* scores,
                              const uint8_t* sources, int secondary_cap, uint64_t n, uint8_t* text, uint64_t cap, uint64_t* offs3, tp_verdict* verdicts, uint32_t* review_bits,
                              const uint8_t* reply_flags) {
    const Sim& s = *(Sim*)h;
    const DevTables& t = s.dt;
    std::vector<uint8_t> all;
    std::vector<uint32_t> fired(std::max<size_t>(t.fired_words, 1));
    std::vector<uint16_t> facets(std::max<size_t>(t.n_facets, 1));
    offs3[0] = 0;
    for (uint64_t f = 0; f < n; ++f) {
        uint32_t T, LB;
        if (tag_begin) { T = (uint32_t)(tag_begin[f + 1] - tag_begin[f]); LB = (uint32_t)(offsets[tag_begin[f + 1]] - offsets[tag_begin[f]]); }
        else { LB = (uint32_t)(offsets[f + 1] - offsets[f]); T = (commas(arena + offsets[f], LB) + 1) * t.split_grow; }
        WsLayout L = ws_layout(std::max(T, 1u), LB, t.final_cap, t.evidence_grow, t.lit_words);
        std::vector<uint8_t> wsmem(L.total, 0xCD);
        FileWs ws = ws_bind(wsmem.data(), L, (uint32_t)(f & 63));
        FileIn in{};
        if (tag_begin) { in.arena = arena; in.tag_offsets = offsets; in.tag_lo = tag_begin[f]; in.tag_hi = tag_begin[f + 1]; in.scores = scores; in.sources = sources;
                         in.merged = secondary_begin != nullptr; in.tag_mid = secondary_begin ? secondary_begin[f] : in.tag_hi; in.secondary_cap = secondary_cap;
                         in.keep_order = reply_flags != nullptr && !(reply_flags[f] & 1u); }
        else { in.text = arena + offsets[f]; in.len = LB; }
        FileOut fo;
        process_file(t, in, ws, fo, fired.data(), t.n_facets ? facets.data() : nullptr, 3);
        if (verdicts) verdicts[f] = fo.verdict;
        if (review_bits) review_bits[f] = fo.review_bits;
        for (int c = 0; c < AUDIT_COLS; ++c) {
            const size_t before = all.size();
            bool first = true;
            for (uint32_t k = 0; k < fo.audit_cnt[c]; ++k) {
                AuditElem e = audit_elem(t, ws, c, k);
                if (!e.keep) continue;
                if (!first) all.push_back(c >= AUDIT_TOP ? '\n' : ';');
                first = false;
                all.insert(all.end(), e.p, e.p + e.len);
                if (e.reason) all.insert(all.end(), e.reason, e.reason + AUDIT_REASON_LEN);
            }
            if (all.size() - before != fo.audit_len[c]) return -2;   // the sizes process_file reports are what the emit kernel relies on
            offs3[f * AUDIT_COLS + c + 1] = all.size();
        }
    }
    if (all.size() > cap) return -1;
    if (!all.empty()) std::memcpy(text, all.data(), all.size());
    return (long long)all.size();
}

long long sim_run_audit(void* h, const uint8_t* arena, const uint64_t* offsets, const uint64_t* tag_begin, const uint64_t* secondary_begin, const double* scores,
                        const uint8_t* sources, int secondary_cap, uint64_t n, uint8_t* text, uint64_t cap, uint64_t* offs3, tp_verdict* verdicts, uint32_t* review_bits) {
    return sim_run_audit_flags(h, arena, offsets, tag_begin, secondary_begin, scores, sources, secondary_cap, n, text, cap, offs3, verdicts, review_bits, nullptr);
}

// The wave stages in audit mode on an emulated wavefront: the lean stage (front half, then the back half's detail build), then
// the general wave stage with one and with two tokens per lane.  Per file: which stage took it (0 none: the general kernel's
// file, 1 lean, 2 / 3 the general wave stage), its verdict and review bits, and the five columns of list text put together from
// the interned ids the stage left in the file's audit row - as k_emit_audit does for such a file.  offs5 has 5 n + 1 entries;
// returns the text size, -1 when cap is too small, -2 when a column's length differs from what the stage reported.
long long sim_run_audit_wave(void* h, const uint8_t* arena_in, const uint64_t* offsets, uint64_t n, uint8_t* text, uint64_t cap, uint64_t* offs5, tp_verdict* verdicts,
                             uint32_t* review_bits, uint8_t* stage_of) {
    g_err.clear();
    const Sim& s = *(Sim*)h;
    const DevTables& t = s.dt;
    std::vector<uint8_t> padded((size_t)offsets[n] + 128, 0);
    uint8_t* arena = padded.data() + 64;
    std::memcpy(arena, arena_in, (size_t)offsets[n]);
    std::vector<uint8_t> all;
    std::vector<uint32_t> fired(std::max<size_t>(t.fired_words, 1));
    std::vector<uint16_t> facets(std::max<size_t>(t.n_facets, 1));
    emu::Shared sh;
    offs5[0] = 0;
    bool mismatch = false;
    for (uint64_t f = 0; f < n; ++f) {
        const uint8_t* line = arena + offsets[f];
        const uint32_t len = (uint32_t)(offsets[f + 1] - offsets[f]);
        std::vector<uint32_t> fin32(FAST_FIN_MAX, 0xDEADBEEF), row(AUD_STRIDE, 0xDEADBEEF);
        uint32_t cnt[AUDIT_COLS], alen[AUDIT_COLS], rbits = 0xDEADBEEF;      // stale on the device: a stage that takes the file writes every one
        for (int c = 0; c < AUDIT_COLS; ++c) cnt[c] = alen[c] = 0xDEADBEEF;
        const LeanAudit aud{row.data(), cnt, alen, &rbits};
        bool done[64];
        FastOut outs[64];
        int stage = 0;
        if (lean_tables_ok(t) && len <= LEAN_LINE_MAX) {
            static LeanFrontLds flds;
            static LeanLds llds;
            static LeanShared lshared;
            std::memset((void*)&flds, 0xCD, sizeof flds);
            std::memset((void*)&llds, 0xCD, sizeof llds);
            lean_shared_init(t, lshared, 0, 1);
            std::vector<uint32_t> cand(64, 0xDEADBEEF);
            sh.run([&](int lane) {
                emu::Wave w{&sh, (uint32_t)lane};
                uint32_t piece[4];
                lean_fetch(w, line, len, piece);
                done[lane] = lean_file(w, t, flds, llds, lshared, t.rc, line, len, piece, cand.data(), outs[lane], fin32.data(), fired.data(), t.n_facets ? facets.data() : nullptr, aud);
            });
            if (done[0]) stage = 1;
        }
        for (int nt = 1; nt <= 2 && !stage && t.fast_ok && len <= FAST_LINE_MAX; ++nt) {
            static FastLds lds;
            std::memset(&lds, 0xCD, sizeof lds);
            sh.run([&](int lane) {
                emu::Wave w{&sh, (uint32_t)lane};
                done[lane] = nt == 1 ? fast_file<1, 2, emu::Wave, LineFront, true>(w, t, lds, t.rc, cls_table(), line, len, outs[lane], fin32.data(), fired.data(),
                                                                                   t.n_facets ? facets.data() : nullptr, false, nullptr, aud)
                                     : fast_file<2, 2, emu::Wave, LineFront, true>(w, t, lds, t.rc, cls_table(), line, len, outs[lane], fin32.data(), fired.data(),
                                                                                   t.n_facets ? facets.data() : nullptr, false, nullptr, aud);
            });
            for (int l = 1; l < 64; ++l) if (done[l] != done[0]) g_err = "lanes disagree on the audit-mode verdict";
            if (done[0]) stage = 1 + nt;
        }
        for (int size = 0; size < 3 && !stage && t.fast_ok; ++size) {   // the long-file stage's three sizes of state (4 + size)
            static LongLds256 l256;
            static LongLds512 l512;
            static LongLds l1024;
            if (len > (size == 0 ? LongLds256::LINE_CAP : LONG_LINE_MAX)) continue;
            std::memset((void*)&l256, 0xCD, sizeof l256); std::memset((void*)&l512, 0xCD, sizeof l512); std::memset((void*)&l1024, 0xCD, sizeof l1024);
            sh.run([&](int lane) {
                emu::Wave w{&sh, (uint32_t)lane};
                done[lane] = size == 0 ? long_file<true>(w, t, l256, t.rc, cls_table(), line, len, outs[lane], fin32.data(), fired.data(), t.n_facets ? facets.data() : nullptr, false, aud)
                           : size == 1 ? long_file<true>(w, t, l512, t.rc, cls_table(), line, len, outs[lane], fin32.data(), fired.data(), t.n_facets ? facets.data() : nullptr, false, aud)
                                       : long_file<true>(w, t, l1024, t.rc, cls_table(), line, len, outs[lane], fin32.data(), fired.data(), t.n_facets ? facets.data() : nullptr, false, aud);
            });
            for (int l = 1; l < 64; ++l) if (done[l] != done[0]) g_err = "lanes disagree on the long-file audit verdict";
            if (done[0]) stage = 4 + size;
        }
        stage_of[f] = (uint8_t)stage;
        verdicts[f] = stage ? outs[0].verdict : tp_verdict{};
        review_bits[f] = stage ? rbits : 0;
        for (int c = 0; c < AUDIT_COLS; ++c) {
            const size_t before = all.size();
            bool first = true;
            for (uint32_t k = 0; stage && k < cnt[c]; ++k) {
                const uint32_t v = c == AUDIT_UNKNOWN ? fin32[k] : c == AUDIT_KEPT ? (row[AUD_FLAGS] ? row[AUD_KEPT + k] : fin32[k]) : c == AUDIT_EVIDENCE ? row[AUD_EVIDENCE + k]
                                   : c == AUDIT_DROPPED ? row[AUD_DROPPED + k] : row[AUD_TOP + k];
                const uint32_t id = v & 0x7FFFFFFFu;
                if (id >= t.n_strings) { g_err = "audit row entry is not an interned id"; return -2; }
                if (c == AUDIT_UNKNOWN && !(t.known_active && !(t.flags[id] & FL_KNOWN))) continue;
                if (!first) all.push_back(c >= AUDIT_TOP ? '\n' : ';');
                first = false;
                all.insert(all.end(), str_ptr(t, id), str_ptr(t, id) + str_len(t, id));
                if (c == AUDIT_DROPPED) { const char* r = (v >> 31) ? ":policy_drop" : ":invalid_tag"; all.insert(all.end(), r, r + AUDIT_REASON_LEN); }
            }
            if (stage && all.size() - before != alen[c]) mismatch = true;
            offs5[f * AUDIT_COLS + c + 1] = all.size();
        }
    }
    if (mismatch) return -2;
    if (all.size() > cap) return -1;
    if (!all.empty()) std::memcpy(text, all.data(), all.size());
    return (long long)all.size();
}

// The wave kernel of tagger replies in audit mode on an emulated wavefront (one reply per file with per-tag sources, or two to
// merge when secondary_begin is given): per file whether the kernel took it (stage_of 1, else 0: the general kernel's file), its
// verdict, review bits and the five columns of list text put together from the ids of its audit row, as sim_run_audit_wave does.
long long sim_run_audit_scored_wave(void* h, const uint8_t* arena_in, const uint64_t* tag_offsets, const uint64_t* tag_begin, const uint64_t* secondary_begin,
                                    const double* scores, const uint8_t* sources, int secondary_cap, uint64_t n, uint8_t* text, uint64_t cap, uint64_t* offs5,
                                    tp_verdict* verdicts, uint32_t* review_bits, uint8_t* stage_of) {
    g_err.clear();
    const Sim& s = *(Sim*)h;
    const DevTables& t = s.dt;
    const uint64_t n_tags = tag_begin[n];
    std::vector<uint8_t> padded((size_t)tag_offsets[n_tags] + 128, 0);
    uint8_t* arena = padded.data() + 64;
    std::memcpy(arena, arena_in, (size_t)tag_offsets[n_tags]);
    std::vector<uint8_t> all;
    std::vector<uint32_t> fired(std::max<size_t>(t.fired_words, 1));
    std::vector<uint16_t> facets(std::max<size_t>(t.n_facets, 1));
    emu::Shared sh;
    offs5[0] = 0;
    bool mismatch = false;
    for (uint64_t f = 0; f < n; ++f) {
        std::vector<uint32_t> fin32(FAST_FIN_MAX, 0xDEADBEEF), row(AUD_STRIDE, 0xDEADBEEF);
        uint32_t cnt[AUDIT_COLS], alen[AUDIT_COLS], rbits = 0xDEADBEEF;      // stale on the device: a stage that takes the file writes every one
        for (int c = 0; c < AUDIT_COLS; ++c) cnt[c] = alen[c] = 0xDEADBEEF;
        const LeanAudit aud{row.data(), cnt, alen, &rbits};
        bool done[64];
        FastOut outs[64];
        int stage = 0;
        if (t.fast_ok && t.reach_levels_ok) {
            static FastLds lds;
            std::memset(&lds, 0xCD, sizeof lds);
            sh.run([&](int lane) {
                emu::Wave w{&sh, (uint32_t)lane};
                ScoredFront<emu::Wave> front;
                front.in = ScoredIn{arena, tag_offsets, tag_begin[f], secondary_begin ? secondary_begin[f] : tag_begin[f + 1], tag_begin[f + 1], scores, sources,
                                    secondary_begin != nullptr, secondary_cap};
                front.aud = aud;
                done[lane] = fast_file<2, 2, emu::Wave, ScoredFront<emu::Wave>, true>(w, t, lds, t.rc, cls_table(), nullptr, 0, outs[lane], fin32.data(), fired.data(),
                                                                                       t.n_facets ? facets.data() : nullptr, false, &front, aud);
            });
            for (int l = 1; l < 64; ++l) if (done[l] != done[0]) g_err = "lanes disagree on the scored audit-mode verdict";
            if (done[0]) stage = 1;
        }
        stage_of[f] = (uint8_t)stage;
        verdicts[f] = stage ? outs[0].verdict : tp_verdict{};
        review_bits[f] = stage ? rbits : 0;
        for (int c = 0; c < AUDIT_COLS; ++c) {
            const size_t before = all.size();
            bool first = true;
            for (uint32_t k = 0; stage && k < cnt[c]; ++k) {
                const uint32_t v = c == AUDIT_UNKNOWN ? fin32[k] : c == AUDIT_KEPT ? (row[AUD_FLAGS] ? row[AUD_KEPT + k] : fin32[k]) : c == AUDIT_EVIDENCE ? row[AUD_EVIDENCE + k]
                                   : c == AUDIT_DROPPED ? row[AUD_DROPPED + k] : row[AUD_TOP + k];
                const uint32_t id = v & 0x7FFFFFFFu;
                if (id >= t.n_strings) { g_err = "audit row entry is not an interned id"; return -2; }
                if (c == AUDIT_UNKNOWN && !(t.known_active && !(t.flags[id] & FL_KNOWN))) continue;
                if (!first) all.push_back(c >= AUDIT_TOP ? '\n' : ';');
                first = false;
                all.insert(all.end(), str_ptr(t, id), str_ptr(t, id) + str_len(t, id));
                if (c == AUDIT_DROPPED) { const char* r = (v >> 31) ? ":policy_drop" : ":invalid_tag"; all.insert(all.end(), r, r + AUDIT_REASON_LEN); }
            }
            if (stage && all.size() - before != alen[c]) mismatch = true;
            offs5[f * AUDIT_COLS + c + 1] = all.size();
        }
    }
    if (mismatch) return -2;
    if (all.size() > cap) return -1;
    if (!all.empty()) std::memcpy(text, all.data(), all.size());
    return (long long)all.size();
}

void sim_lean_declines(unsigned long long* out32, int reset) {
    for (int k = 0; k < 32; ++k) { out32[k] = g_lean_declines[k]; if (reset) g_lean_declines[k] = 0; }
}
void sim_release(sim_result* r) { delete (SimOut*)r->opaque; r->opaque = nullptr; }

}  // extern "C"
