// C ABI of the batch tag pipeline (include/tagpipe.h): context life cycle, table upload, batch runs.
// Every run goes through the HIP kernels; there is no host implementation of the pipeline in this library.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/tagpipe.h"
#include "tp_launch.h"
#include "tp_pipeline.h"
#include "tp_tables.hpp"

using namespace tp;

namespace {

thread_local std::string g_error;
int fail(int code, const std::string& msg) { g_error = msg; return code; }

struct HipFail { hipError_t e; const char* what; };
#define HIP_OK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) throw HipFail{e__, #expr}; } while (0)

struct ScratchFull { size_t bytes, cap; };

// Grow-only device buffer.  A bound one is a piece of memory that someone else owns: it cannot grow and is never freed.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool bound = false;
    void bind(void* q, size_t bytes) { p = q; cap = bytes; bound = true; }
    void need(size_t bytes, bool keep = false, hipStream_t s = nullptr) {
        if (bytes <= cap) return;
        if (bound) throw ScratchFull{bytes, cap};
        size_t want = std::max(bytes, cap + cap / 2);
        want = (want + 255) & ~(size_t)255;
        void* q = nullptr;
        HIP_OK(hipMalloc(&q, want));
        if (keep && p && cap) {
            HIP_OK(hipMemcpyAsync(q, p, cap, hipMemcpyDeviceToDevice, s));
            HIP_OK(hipStreamSynchronize(s));
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
    }
    void release() { if (p && !bound) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

template <class T>
const T* upload(std::vector<void*>& owned, const std::vector<T>& v) {
    void* d = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIP_OK(hipMalloc(&d, bytes));
    owned.push_back(d);
    if (!v.empty()) HIP_OK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return (const T*)d;
}

// Host memory of the results the host-pointer entry points hand out: page-locked, so that the copies back from the device
// run at the bus's rate (into pageable memory they take several times as long as the kernels and the bus together), and
// pooled for the process, because locking a gigabyte of pages costs as much as copying it.  tp_release gives a result's
// buffers back to the pool; the pool keeps a bounded number of them and frees the rest.  When the pages cannot be locked
// the buffer is ordinary memory: slower, not wrong.
struct PinBuf { void* p = nullptr; size_t cap = 0; bool pinned = false; };
class PinPool {
    std::mutex mu;
    std::vector<PinBuf> idle;
    size_t idle_bytes = 0;
    static constexpr size_t MAX_IDLE = 32;
    // what the pool keeps locked while nobody uses it: 16 GiB (the results of ten million files of the headline corpus are 11), or
    // TAGPIPE_PIN_POOL_GB
    const size_t MAX_IDLE_BYTES = [] {
        size_t gb = 16;
        if (const char* e = std::getenv("TAGPIPE_PIN_POOL_GB")) { const long v = std::strtol(e, nullptr, 10); if (v >= 0 && v <= 4096) gb = (size_t)v; }
        return gb << 30;
    }();
    static void drop(PinBuf& b) { if (b.p) { if (b.pinned) (void)hipHostFree(b.p); else std::free(b.p); } b = PinBuf{}; }
public:
    PinBuf take(size_t bytes) {
        bytes = std::max<size_t>(bytes, 64);
        {
            std::lock_guard<std::mutex> lock(mu);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i].cap >= bytes && idle[i].cap <= 2 * bytes + (1u << 20) && (best == idle.size() || idle[i].cap < idle[best].cap)) best = i;
            if (best != idle.size()) { PinBuf b = idle[best]; idle_bytes -= b.cap; idle.erase(idle.begin() + best); return b; }
        }
        PinBuf b;
        b.cap = (bytes + bytes / 8 + 4095) & ~(size_t)4095;            // some slack: the next batch of about this size fits too
        if (hipHostMalloc(&b.p, b.cap, hipHostMallocPortable) == hipSuccess && b.p) b.pinned = true;
        else { (void)hipGetLastError(); b.p = std::malloc(b.cap); if (!b.p) throw std::bad_alloc(); }
        return b;
    }
    void give(PinBuf b) {
        if (!b.p) return;
        std::lock_guard<std::mutex> lock(mu);
        if (idle.size() >= MAX_IDLE || idle_bytes + b.cap > MAX_IDLE_BYTES) { drop(b); return; }
        idle_bytes += b.cap;
        idle.push_back(b);
    }
    ~PinPool() { for (PinBuf& b : idle) drop(b); }
};
PinPool& pin_pool() { static PinPool* pool = new PinPool(); return *pool; }   // never destroyed: results may outlive every context
template <class T>
class PinArr {
    PinBuf b;
    size_t n = 0;
public:
    PinArr() = default;
    PinArr(const PinArr&) = delete;
    PinArr& operator=(const PinArr&) = delete;
    ~PinArr() { pin_pool().give(b); }
    void resize(size_t count) {                                     // contents are not kept and not cleared: every user fills what it reads
        if (count * sizeof(T) > b.cap) { pin_pool().give(b); b = PinBuf{}; b = pin_pool().take(count * sizeof(T)); }
        n = count;
    }
    T* data() { return (T*)b.p; }
    const T* data() const { return (const T*)b.p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) { return ((T*)b.p)[i]; }
    const T& operator[](size_t i) const { return ((const T*)b.p)[i]; }
};
struct HostResult {
    PinArr<uint8_t> captions;
    PinArr<uint64_t> caption_offsets, histogram;
    PinArr<tp_verdict> verdicts;
    PinArr<uint32_t> fired;
    PinArr<uint16_t> facet_choice;
    PinArr<uint8_t> audit_text;        // audit mode only
    PinArr<uint64_t> audit_offsets;
    PinArr<uint32_t> review_bits;
};

}  // namespace

// What one run works in besides the caller's inputs and outputs.
struct WorkBufs {
    DevBuf wave_T, wave_LB, wave_bytes, wave_off, tile_sums, cap_len, n_final, slot_of, fin32, slow_list, slow_list0, slow_list1, slow_list2, slow_list_b, tile_state, scalars, workspace;
    DevBuf hist_parts;               // EmitArgs::hist_parts (grown buffers only: the fixed working memory has no room for it)
    DevBuf side_list0, side_list1, side_list2;   // the side chain's lists (grown buffers only)
    void release() {
        for (DevBuf* b : {&wave_T, &wave_LB, &wave_bytes, &wave_off, &tile_sums, &cap_len, &n_final, &slot_of, &fin32, &slow_list, &slow_list0, &slow_list1, &slow_list2, &slow_list_b,
                          &tile_state, &scalars, &workspace, &hist_parts, &side_list0, &side_list1, &side_list2})
            b->release();
    }
};

struct tp_ctx {
    int device = 0;
    HostTables ht;
    DevTables dt;
    DevBuf dt_dev;                   // dt again, in device memory (the lean stage reads it from there)
    std::vector<void*> table_mem;
    // Strings learned from the input (see learn_unknown_strings below)
    tp_config cfg{};                                 // a copy of the configuration, its strings and blobs held in cfg_text
    std::vector<std::unique_ptr<std::string>> cfg_text;
    std::vector<std::string> public_strings;         // the configuration's interned strings (what tp_get_string hands out pointers into)
    std::vector<std::string> learned;                // in the order they were learned
    std::unordered_map<std::string, uint32_t> learned_set;
    DevBuf unk_keys, unk_ent, unk_text, unk_count, unk_out;
    bool learning = false;
    WorkBufs own;                    // grown with hipMalloc as runs need them
    WorkBufs fixed;                  // pieces of this device's static scratch array: tp_run_lines_device_noalloc works in these
    bool fixed_bound = false;        //   (bound by that entry point's first call on the context)
    WorkBufs* w = &own;
    uint64_t* host_scalars = nullptr;   // pinned host words the runs read device scalars back into
    DevBuf in_arena, in_offsets, in_tag_begin, in_scores, in_sources, in_secondary, in_reply_flags;           // host-pointer entry points stage their inputs here
    DevBuf out_captions, out_offsets, out_verdicts, out_hist, out_counters, out_fired, out_facets;
    DevBuf audit_cnt, audit_len, out_audit, out_audit_offsets, out_review_bits, aud32;
    std::unordered_map<std::string, uint32_t> string_ids;   // text of every interned string -> id (made on first use by tp_decode_joytag_replies)
    bool audit = false;              // tp_set_audit: the host-pointer runs also return the audit row's tag lists
    std::vector<hipEvent_t> events;
    tp_timing timing{};
    size_t ws_budget = (size_t)64 << 30;
    bool no_fast = false;            // TAGPIPE_NO_FAST=1: every file takes the general kernel (testing aid)
    bool use_lean = false;           // the lean stage runs ahead of the wave stages (TAGPIPE_NO_LEAN=1 turns it off: testing aid)
    // A corpus whose files the lean stage mostly declines (a kept alias-chain member in every file, say) pays for its front
    // half and gets nothing: after a range of which it took less than a quarter, the next lean_rest ranges go to the general
    // wave stages directly, then one range tries it again.  Results do not depend on it (TAGPIPE_LEAN_ADAPT=0: always try).
    bool lean_adapt = true;
    int lean_rest = 0;
    // The stages behind the lean stage's front half run on a stream of their own, beside the back half, when the front half
    // declined between one file in 256 and one in 16 of the last range.  Measured: c1 (6 % of the files) gains 2 - 3 %; c5, where
    // a tenth of the files goes that way, loses 2 %: the chain then outlasts the back half.  A dozen such files in a range (c2, c3,
    // c4) cost the main stream 0.04 ms as five nearly empty launches in a row, each waiting for its one file, which the side
    // chain would hide too (1 - 3 % of the step); it is left out there because kernels that overlap stretch one another's
    // durations in a kernel trace (k_lean_b: 1.37 ms by events on its stream, 1.72 ms in the trace), so that per-kernel times
    // no longer add up to the step.  TAGPIPE_SIDE_CHAIN=0 / 1: never / always; results do not depend on it
    bool side_chain = false, side_adapt = true;
    uint64_t last_front = 0;         // files the front half declined in the last range
    uint64_t last_stage1_out = 0;    // files the first general wave stage handed on in the last range
    hipStream_t side_q = nullptr;
    hipEvent_t side_ev[2] = {nullptr, nullptr};
    hipStream_t side_stream() {
        if (!side_q) HIP_OK(hipStreamCreateWithFlags(&side_q, hipStreamNonBlocking));
        return side_q;
    }
    hipEvent_t side_event(int k) {
        if (!side_ev[k]) HIP_OK(hipEventCreateWithFlags(&side_ev[k], hipEventDisableTiming));
        return side_ev[k];
    }
    bool stage1_idle = false;        // the first general wave stage took under a tenth of what the lean stage declined in the last range: a small grid for it
    bool no_fused = true;            // TAGPIPE_FUSED=1: the plain run's back half, caption offsets and caption bytes as ONE kernel (k_lean_bx).  Off by default:
                                     // measured on c2 (1M files) the fused kernel takes 2.86 ms where k_lean_b + scan + k_emit_runs take 2.2 ms
    uint64_t fused_reruns = 0;       // ranges the fused back half had to hand back to the separate kernels
    uint64_t last_slow = 0;
    uint32_t last_fpw = 64;          // files per wavefront of the last general-kernel run (the stride of its workspace rows)

    ~tp_ctx() {
        (void)hipSetDevice(device);
        for (void* p : table_mem) (void)hipFree(p);
        own.release();
        if (host_scalars) (void)hipHostFree(host_scalars);
        for (DevBuf* b : {&unk_keys, &unk_ent, &unk_text, &unk_count, &unk_out}) b->release();
        for (DevBuf* b : {&dt_dev, &in_arena, &in_offsets, &in_tag_begin,
                          &in_scores, &in_sources, &in_secondary, &in_reply_flags, &out_captions, &out_offsets, &out_verdicts, &out_hist, &out_counters, &out_fired, &out_facets,
                          &audit_cnt, &audit_len, &out_audit, &out_audit_offsets, &out_review_bits, &aud32})
            b->release();
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        for (hipEvent_t e : pipe_events) (void)hipEventDestroy(e);
        for (hipStream_t q : pipe_streams) if (q) (void)hipStreamDestroy(q);
        for (hipEvent_t e : side_ev) if (e) (void)hipEventDestroy(e);
        if (side_q) (void)hipStreamDestroy(side_q);
    }
    hipEvent_t event(size_t i) {
        while (events.size() <= i) { hipEvent_t e; HIP_OK(hipEventCreate(&e)); events.push_back(e); }
        return events[i];
    }
    // tp_run_lines over host buffers in pieces: the next piece's lines go to the device, the piece before's captions come
    // back, while the kernels work on the current one - three streams, events between them (no timing on these)
    hipStream_t pipe_streams[3] = {nullptr, nullptr, nullptr};       // in, run, out
    std::vector<hipEvent_t> pipe_events;
    hipStream_t pipe_stream(int k) {
        if (!pipe_streams[k]) HIP_OK(hipStreamCreateWithFlags(&pipe_streams[k], hipStreamNonBlocking));
        return pipe_streams[k];
    }
    hipEvent_t pipe_event(size_t i) {
        while (pipe_events.size() <= i) { hipEvent_t e; HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); pipe_events.push_back(e); }
        return pipe_events[i];
    }
    bool no_overlap = false;         // TAGPIPE_NO_OVERLAP=1: the host-pointer run copies in, computes and copies out in turn (testing aid)
};

namespace {

struct RunIo {
    const uint8_t* arena; const uint64_t* offsets; const uint64_t* tag_begin; const double* scores; const uint8_t* sources;
    const uint64_t* secondary_begin; int secondary_cap;
    const uint8_t* reply_flags;                                         // scored ingest: tp_replies::has_score per file on the device, or null
    uint64_t n_files;
    uint8_t* captions; uint64_t caption_capacity; bool grow_captions;   // grow_captions: captions live in ctx->out_captions and may be reallocated
    uint64_t* caption_offsets; tp_verdict* verdicts; uint64_t* histogram; tp_counters* counters; uint32_t* fired; uint16_t* facet_choice;
    hipStream_t stream; int mode;
    bool audit; uint64_t* audit_offsets; uint32_t* review_bits;    // audit: every file takes the general kernel in mode 3; the lists' text goes to ctx->out_audit
    // the host-pointer run in pieces: files per range (0: the usual 2^24), what to do before a range's first kernel and after
    // its last one (bytes0 .. bytes1: the range's captions)
    uint64_t range_files = 0;
    std::function<void(uint64_t f0, uint64_t f1)> before_range;
    std::function<void(uint64_t f0, uint64_t f1, uint64_t bytes0, uint64_t bytes1)> after_range;
};

// One range of files [f0, f1): wave-per-file kernel, general kernel for what it declined, caption offsets, emit.
// When the general kernel's workspace would exceed the budget the range is halved and each half runs on its own.
struct RunState {
    uint64_t total = 0;          // caption bytes so far (host copy of the device running total)
    int chunk = 0;               // parity of the device running-total slot
    bool over_capacity = false;
    uint64_t audit_total = 0;    // bytes of audit text so far
    int audit_chunk = 0;
    size_t ev = 2;               // next free event
    bool timed = true;           // false: no events are created or recorded (the run must not allocate)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> fast_ev, pipe_ev, emit_ev;
    struct Lean { hipEvent_t begin, mid, end; };
    std::vector<Lean> lean_ev;     // lean stage: front kernel between begin and mid, back kernel between mid and end
    std::vector<std::pair<hipEvent_t, hipEvent_t>> fused_ev;   // the fused back half (k_lean_bx)
};
constexpr size_t SCALARS_BYTES = 256;   // WorkBufs::scalars: [0, 64) totals, status and list counts; [64, 128) the fused back half's own batch counters;
                                        // [128, 136) its ticket counter and the number of files it declined

int run_range(tp_ctx* c, RunIo& io, uint64_t f0, uint64_t f1, RunState& st, bool allow_fused = true) {
    hipStream_t s = io.stream;
    const uint64_t n = f1 - f0;
    auto next_event = [&]() -> hipEvent_t { return st.timed ? c->event(st.ev++) : nullptr; };
    auto record = [&](hipEvent_t e) { if (e) HIP_OK(hipEventRecord(e, s)); };
    uint64_t* d_total = c->w->scalars.as<uint64_t>();          // [0], [1]: running caption total, ping-pong
    uint32_t* d_status = (uint32_t*)(d_total + 2);
    uint32_t* d_slow_count = d_status + 1;
    // audit mode over tagger replies: the scored stage lists what it takes (a reply that keeps its own order is left to the
    // general kernel: the stage's lanes hold the candidates sorted)
    const bool audit_scored = io.audit && c->dt.fast_ok && io.tag_begin && io.mode == 0 && !c->no_fast && c->dt.reach_levels_ok;
    const bool use_scored = c->dt.fast_ok && io.tag_begin && io.mode == 0 && !c->no_fast && (!io.audit || audit_scored);   // tagger replies: the wave kernel's scored front
    // audit mode over tag lines: the lean stage lists what it takes (interned ids, tp_lean.h), everything else takes the general kernel
    const bool audit_lean = io.audit && c->use_lean && c->dt.fast_ok && !io.tag_begin && io.mode == 0 && !c->no_fast;
    bool lean_now = c->use_lean && io.mode == 0 && !io.tag_begin;
    if (lean_now && c->lean_adapt && !c->learning && c->lean_rest > 0 && n >= 4096) { --c->lean_rest; lean_now = false; }
    bool lean_ran = false, side_ran = false;
    const bool use_fast = use_scored || audit_lean || (c->dt.fast_ok && !io.tag_begin && (io.mode == 0 || io.mode == 2) && !c->no_fast && !io.audit);
    // The plain run over tag lines (no detail arrays, no audit lists): the lean stage's back half, the caption offsets and the
    // caption bytes are one kernel (k_lean_bx); the stages behind the front half run before it, so that every file's caption
    // length is known or computed there.
    const bool fused = allow_fused && use_fast && !use_scored && !audit_lean && lean_now && io.mode == 0 && !io.fired && !io.facet_choice && !c->no_fused;
    uint8_t* const sc_bytes = (uint8_t*)c->w->scalars.p;
    tp_counters* const d_bx_counters = (tp_counters*)(sc_bytes + 64);
    uint32_t* const d_ticket = (uint32_t*)(sc_bytes + 128);
    uint32_t* const d_back_count = d_ticket + 1;
    FastArgs bxa{};

    c->w->cap_len.need(n * 4); c->w->n_final.need(n * 4); c->w->slot_of.need(n * 4);
    uint32_t* cap_len = c->w->cap_len.as<uint32_t>() - f0;      // indexed by absolute file number below
    uint32_t* n_final = c->w->n_final.as<uint32_t>() - f0;
    uint32_t* slot_of = c->w->slot_of.as<uint32_t>() - f0;
    uint32_t n_slow = (uint32_t)n;
    const uint32_t* item_list = nullptr;
    if (use_fast) {
        c->w->fin32.need(n * fast_fin_stride() * 4);
        c->w->slow_list.need(n * 4);
        c->w->slow_list0.need(n * 4);
        c->w->slow_list1.need(n * 4);
        c->w->slow_list2.need(n * 4);
        uint32_t* d_slow_count1 = d_slow_count + 1;
        uint32_t* d_slow_count2 = d_slow_count + 2;
        HIP_OK(hipMemsetAsync(d_slow_count, 0, 12, s));
        FastArgs fa{};
        fa.arena = io.arena; fa.offsets = io.offsets; fa.file0 = f0; fa.file_end = f1;
        fa.verdicts = io.verdicts; fa.n_final = n_final; fa.cap_len = cap_len; fa.fin32 = c->w->fin32.as<uint32_t>(); fa.slot_of = slot_of;
        fa.fired = io.fired; fa.facet_choice = io.facet_choice; fa.triage = io.mode == 2 ? 1 : 0;
        hipEvent_t p0 = next_event(), p1 = next_event();
        record(p0);
        if (use_scored) {
            fa.tag_begin = io.tag_begin; fa.scores = io.scores; fa.sources = io.sources; fa.secondary_begin = io.secondary_begin; fa.secondary_cap = io.secondary_cap;
            fa.stage = 4; fa.slow_list = c->w->slow_list.as<uint32_t>(); fa.slow_count = d_slow_count; fa.item_list = nullptr; fa.item_count = nullptr;
            if (audit_scored) {
                fa.reply_flags = io.reply_flags;
                c->aud32.need(n * AUD_STRIDE * 4); c->audit_cnt.need(n * AUDIT_COLS * 4); c->audit_len.need(n * AUDIT_COLS * 4);
                fa.aud32 = c->aud32.as<uint32_t>(); fa.audit_cnt = c->audit_cnt.as<uint32_t>(); fa.audit_len = c->audit_len.as<uint32_t>(); fa.review_bits = io.review_bits + f0;
            }
            HIP_OK(launch_fast(c->dt, fa, s));
        } else if (audit_lean) {
            c->aud32.need(n * AUD_STRIDE * 4); c->audit_cnt.need(n * AUDIT_COLS * 4); c->audit_len.need(n * AUDIT_COLS * 4);
            fa.aud32 = c->aud32.as<uint32_t>(); fa.audit_cnt = c->audit_cnt.as<uint32_t>(); fa.audit_len = c->audit_len.as<uint32_t>(); fa.review_bits = io.review_bits + f0;
            fa.item_list = nullptr; fa.item_count = nullptr;
            c->w->slow_list0.need(n * 4);
            uint32_t* d_slow_count0 = (uint32_t*)(d_total + 6);   // behind the audit totals
            HIP_OK(hipMemsetAsync(d_slow_count0, 0, 4, s));
            if (lean_now) {
                fa.stage = 0; fa.slow_list = c->w->slow_list0.as<uint32_t>(); fa.slow_count = d_slow_count0; fa.tables = c->dt_dev.as<DevTables>();
                hipEvent_t pm = next_event(), pe = next_event();
                HIP_OK(launch_fast(c->dt, fa, s, pm));
                record(pe);
                if (st.timed) st.lean_ev.push_back({p0, pm, pe});
                lean_ran = true;
                // what the lean stage declines: the general wave stages in audit mode (one token per lane, then two, then the
                // long-file stage); what they all decline is the general kernel's
                fa.item_list = c->w->slow_list0.as<uint32_t>(); fa.item_count = d_slow_count0;
            }
            fa.stage = 1; fa.slow_list = c->w->slow_list1.as<uint32_t>(); fa.slow_count = d_slow_count1;
            HIP_OK(launch_fast(c->dt, fa, s));
            fa.stage = 2; fa.item_list = c->w->slow_list1.as<uint32_t>(); fa.item_count = d_slow_count1;
            fa.slow_list = c->w->slow_list2.as<uint32_t>(); fa.slow_count = d_slow_count2;
            HIP_OK(launch_fast(c->dt, fa, s));
            // the long-file stage, three sizes as in the plain run (the first two stages' lists and counters are free again)
            if (lean_ran) HIP_OK(hipMemcpyAsync(&c->host_scalars[3], d_slow_count0, 4, hipMemcpyDeviceToHost, s));   // how many the lean stage declined
            HIP_OK(hipMemsetAsync(d_slow_count0, 0, 4, s));
            HIP_OK(hipMemsetAsync(d_slow_count1, 0, 4, s));
            fa.stage = 3; fa.long_size = 0; fa.item_list = c->w->slow_list2.as<uint32_t>(); fa.item_count = d_slow_count2;
            fa.slow_list = c->w->slow_list0.as<uint32_t>(); fa.slow_count = d_slow_count0;
            HIP_OK(launch_fast(c->dt, fa, s));
            fa.long_size = 1; fa.item_list = c->w->slow_list0.as<uint32_t>(); fa.item_count = d_slow_count0;
            fa.slow_list = c->w->slow_list1.as<uint32_t>(); fa.slow_count = d_slow_count1;
            HIP_OK(launch_fast(c->dt, fa, s));
            fa.long_size = 2; fa.item_list = c->w->slow_list1.as<uint32_t>(); fa.item_count = d_slow_count1;
            fa.slow_list = c->w->slow_list.as<uint32_t>(); fa.slow_count = d_slow_count;
            HIP_OK(launch_fast(c->dt, fa, s));
        } else {
        // first stage: one token per lane; what it declines goes through the second stage (two per lane), what that declines
        // through the long-file stage, and what is left to the general kernel.  Later stages learn their item counts on the device.
        fa.item_list = nullptr; fa.item_count = nullptr;
        uint32_t* const d_slow_count0 = (uint32_t*)(d_total + 6);   // behind the audit totals
        uint32_t* const d_count_b = (uint32_t*)(sc_bytes + 192);    // the lean stage's back half lists its files from the end of slow_list0
        uint32_t* const d_side_count = d_count_b + 1;               // [0 .. 2]: the side chain's lists; [3]: the main chain's fourth list while the side chain runs
        // The files the front half declines (too many tokens for it, as a rule: the expensive ones for the stages below) can go through
        // those stages on a stream of their own while the back half runs; the files the back half declines follow on this stream.
        const bool side = lean_now && !fused && c->side_chain && c->w == &c->own && n >= 4096;   // (its stream and events are made on first use: not for the run that must not allocate)
        if (lean_now) {
            lean_ran = true;
            // stage 0: the lean stage takes the common shape of a file; the stages below see what it declines
            c->w->slow_list0.need(n * 4);
            HIP_OK(hipMemsetAsync(d_slow_count0, 0, 4, s));
            HIP_OK(hipMemsetAsync(d_count_b, 0, 20, s));
            fa.stage = 0; fa.slow_list = c->w->slow_list0.as<uint32_t>(); fa.slow_count = d_slow_count0; fa.tables = c->dt_dev.as<DevTables>();
            fa.front_only = fused ? 1 : 0;
            if (!fused) { fa.slow_count_b = d_count_b; fa.item_cap = (uint32_t)n; }
            hipEvent_t pm = next_event(), pe = next_event();
            HIP_OK(launch_fast(c->dt, fa, s, pm, side ? c->side_event(0) : nullptr));
            record(pe);
            if (st.timed) st.lean_ev.push_back({p0, pm, pe});
            HIP_OK(hipMemcpyAsync(&c->host_scalars[3], d_slow_count0, 4, hipMemcpyDeviceToHost, s));   // how many each half declined (the counters are used again below)
            if (!fused) HIP_OK(hipMemcpyAsync(&c->host_scalars[6], d_count_b, 4, hipMemcpyDeviceToHost, s));
            else c->host_scalars[6] = 0;
            bxa = fa;
            fa.front_only = 0; fa.slow_count_b = nullptr;
            fa.item_list = c->w->slow_list0.as<uint32_t>(); fa.item_count = d_slow_count0;
            if (!fused) fa.item_count_b = d_count_b;
        }
        side_ran = side;
        if (side) {
            // the front half's files: every stage below on the side stream, lists of its own, the general kernel's list shared
            hipStream_t q = c->side_stream();
            c->w->side_list0.need(n * 4); c->w->side_list1.need(n * 4); c->w->side_list2.need(n * 4); c->w->slow_list_b.need(n * 4);
            uint32_t* const l0 = c->w->side_list0.as<uint32_t>(); uint32_t* const l1 = c->w->side_list1.as<uint32_t>(); uint32_t* const l2 = c->w->side_list2.as<uint32_t>();
            HIP_OK(hipStreamWaitEvent(q, c->side_event(0), 0));
            FastArgs sa = fa;
            sa.item_count_b = nullptr;
            // grids for about as many files as the front half declined last time (a wavefront each, and as many again): the blocks of a
            // full-size grid that find nothing to do would still pass through the CUs between the back half's
            sa.grid_cap = (unsigned)std::min<uint64_t>(64 + c->last_front / 2, 4096);
            sa.stage = 1; sa.slow_list = l0; sa.slow_count = d_side_count;
            HIP_OK(launch_fast(c->dt, sa, q));
            sa.stage = 2; sa.item_list = l0; sa.item_count = d_side_count; sa.slow_list = l1; sa.slow_count = d_side_count + 1;
            HIP_OK(launch_fast(c->dt, sa, q));
            sa.stage = 3; sa.long_size = 0; sa.item_list = l1; sa.item_count = d_side_count + 1; sa.slow_list = l2; sa.slow_count = d_side_count + 2;
            HIP_OK(launch_fast(c->dt, sa, q));
            HIP_OK(hipMemsetAsync(d_side_count, 0, 4, q));        // (the second stage has read it)
            sa.long_size = 1; sa.item_list = l2; sa.item_count = d_side_count + 2; sa.slow_list = l0; sa.slow_count = d_side_count;
            HIP_OK(launch_fast(c->dt, sa, q));
            sa.long_size = 2; sa.item_list = l0; sa.item_count = d_side_count; sa.slow_list = c->w->slow_list.as<uint32_t>(); sa.slow_count = d_slow_count;
            HIP_OK(launch_fast(c->dt, sa, q));
            HIP_OK(hipEventRecord(c->side_event(1), q));
            fa.item_count = nullptr;                               // this stream's stages: the back half's files only
        }
        fa.stage = 1; fa.slow_list = c->w->slow_list1.as<uint32_t>(); fa.slow_count = d_slow_count1;
        fa.grid_cap = lean_ran && c->stage1_idle ? 512u : 0u;     // (a stage that took next to nothing of its list last time: blocks cost their set-up)
        HIP_OK(launch_fast(c->dt, fa, s));
        fa.grid_cap = 0;
        fa.item_count_b = nullptr;
        if (lean_ran) HIP_OK(hipMemcpyAsync(&c->host_scalars[4], d_slow_count1, 4, hipMemcpyDeviceToHost, s));   // how many of them it handed on
        fa.stage = 2; fa.item_list = c->w->slow_list1.as<uint32_t>(); fa.item_count = d_slow_count1;
        fa.slow_list = c->w->slow_list2.as<uint32_t>(); fa.slow_count = d_slow_count2;
        // behind the lean stage the second stage's grid is small (few files reach it as a rule); a wavefront per eight files of
        // what reached it last time when that asks for more (c5, 21,000 of 200,000 files: 0.26 -> 0.22 ms)
        if (lean_ran) fa.grid_cap = (unsigned)std::min<uint64_t>(std::max<uint64_t>(512, c->last_stage1_out / 8), 4096);
        HIP_OK(launch_fast(c->dt, fa, s));
        fa.grid_cap = 0;
        // third stage, three sizes of per-file state (smallest first: more files at once); the first two stages' lists and
        // counters have been read by now and take what the sizes hand on (while the side chain reads the lean stage's list, another)
        uint32_t* const list3 = side ? c->w->slow_list_b.as<uint32_t>() : c->w->slow_list0.as<uint32_t>();
        uint32_t* const count3 = side ? d_side_count + 3 : d_slow_count0;
        if (!side) HIP_OK(hipMemsetAsync(d_slow_count0, 0, 4, s));
        HIP_OK(hipMemsetAsync(d_slow_count1, 0, 4, s));
        fa.stage = 3; fa.long_size = 0; fa.item_list = c->w->slow_list2.as<uint32_t>(); fa.item_count = d_slow_count2;
        fa.slow_list = list3; fa.slow_count = count3;
        HIP_OK(launch_fast(c->dt, fa, s));
        fa.long_size = 1; fa.item_list = list3; fa.item_count = count3;
        fa.slow_list = c->w->slow_list1.as<uint32_t>(); fa.slow_count = d_slow_count1;
        HIP_OK(launch_fast(c->dt, fa, s));
        fa.long_size = 2; fa.item_list = c->w->slow_list1.as<uint32_t>(); fa.item_count = d_slow_count1;
        fa.slow_list = c->w->slow_list.as<uint32_t>(); fa.slow_count = d_slow_count;
        HIP_OK(launch_fast(c->dt, fa, s));
        if (side) HIP_OK(hipStreamWaitEvent(s, c->side_event(1), 0));
        }
        record(p1);
        if (st.timed) st.fast_ev.emplace_back(p0, p1);
        HIP_OK(hipMemcpyAsync(&c->host_scalars[0], d_slow_count, 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        n_slow = (uint32_t)c->host_scalars[0];
        if (lean_ran && n >= 4096) {
            const uint64_t declined_front = (uint32_t)c->host_scalars[3];
            const uint64_t declined = declined_front + (audit_lean ? 0u : (uint32_t)c->host_scalars[6]);
            c->last_front = declined_front;
            c->last_stage1_out = (uint32_t)c->host_scalars[4];
            if (!audit_lean && c->side_adapt) c->side_chain = declined_front * 256 >= n && declined_front * 16 <= n;   // enough of them to matter, no more than the back half's run hides
            c->lean_rest = declined * 4 > n * 3 ? 15 : 0;          // it took less than a quarter: rest
            const uint64_t stage1_saw = side_ran ? declined - declined_front : declined;
            if (!audit_lean) c->stage1_idle = stage1_saw >= 1024 && ((uint64_t)(uint32_t)c->host_scalars[4]) * 10 > stage1_saw * 9;   // the stage behind it handed on over nine in ten
        }
        item_list = c->w->slow_list.as<uint32_t>();
    }
    const bool want_ids = use_fast || (io.histogram != nullptr && io.mode == 0);   // rows of tag ids: the wave kernels' output, and what the histogram kernel counts
    if (want_ids && !use_fast) c->w->fin32.need(n * fast_fin_stride() * 4);
    c->last_slow += n_slow;
    uint32_t fpw_used = 64;
    if (n_slow) {
        // The general kernel runs one file per lane, and the lanes of a wavefront wait for one another at every branch where
        // their files differ: a few thousand files go four to a wavefront, many files 64 to a wavefront.
        uint32_t fpw = 64;
        while (fpw > 4 && (uint64_t)n_slow / fpw < 2048) fpw >>= 1;   // measured on 4000 files: 64 per wavefront 10.4 ms, 16: 8.3, 4: 6.5, 1: 8.6
        if (n_slow <= 1024) fpw = 1;                                  // a few hundred files (456 of 200k c3c files): 1 per wavefront 2.9 ms, 2: 3.3, 4: 3.7, 8: 4.3
        if (const char* e = std::getenv("TAGPIPE_GENERAL_FPW")) { int v = atoi(e); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) fpw = (uint32_t)v; }   // testing aid
        const uint64_t spread_budget = std::min<uint64_t>(c->ws_budget, (uint64_t)4 << 30);   // a workspace row is 64 lanes wide however few hold a file
        uint64_t n_waves = 0, ws_bytes = 0;
        uint32_t status = 0;
        for (;;) {
            n_waves = ((uint64_t)n_slow + fpw - 1) / fpw;
            c->w->wave_T.need(n_waves * 4); c->w->wave_LB.need(n_waves * 4); c->w->wave_bytes.need(n_waves * 8); c->w->wave_off.need((n_waves + 1) * 8);
            c->w->tile_sums.need(scan_tiles_needed(std::max<uint64_t>(n, n_waves)) * 8);
            HIP_OK(launch_wave_sizes(c->dt, io.arena, io.offsets, io.tag_begin, item_list, f0, n_slow, fpw, c->w->wave_T.as<uint32_t>(), c->w->wave_LB.as<uint32_t>(),
                                     c->w->wave_bytes.as<uint64_t>(), d_status, s));
            HIP_OK(launch_scan_u64(c->w->wave_bytes.as<uint64_t>(), n_waves, c->w->wave_off.as<uint64_t>(), c->w->tile_sums.as<uint64_t>(), nullptr, nullptr, s));
            HIP_OK(hipMemcpyAsync(&c->host_scalars[1], c->w->wave_off.as<uint64_t>() + n_waves, 8, hipMemcpyDeviceToHost, s));
            HIP_OK(hipMemcpyAsync(&c->host_scalars[2], d_status, 4, hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
            ws_bytes = c->host_scalars[1];
            status = (uint32_t)c->host_scalars[2];
            if (status || fpw == 64 || ws_bytes <= spread_budget) break;
            fpw <<= 1;                                             // too much memory for so thin a spread
        }
        if (status) return fail(TP_E_INVALID_ARG, "a file is longer than 2 GiB");
        if (ws_bytes > c->ws_budget && n > 64) {
            uint64_t mid = f0 + n / 2;
            c->last_slow -= n_slow;
            int rc = run_range(c, io, f0, mid, st, allow_fused);
            if (rc != TP_OK && rc != TP_E_CAPACITY) return rc;
            return run_range(c, io, mid, f1, st, allow_fused);
        }
        c->w->workspace.need(ws_bytes);
        PipelineArgs pa{};
        pa.arena = io.arena; pa.offsets = io.offsets; pa.tag_begin = io.tag_begin; pa.scores = io.scores; pa.sources = io.sources;
        pa.secondary_begin = io.secondary_begin; pa.secondary_cap = io.secondary_cap; pa.reply_flags = io.reply_flags;
        pa.item_list = item_list; pa.file0 = f0; pa.n_items = n_slow; pa.fpw = fpw;
        fpw_used = fpw;
        c->last_fpw = fpw;
        pa.wave_T = c->w->wave_T.as<uint32_t>(); pa.wave_LB = c->w->wave_LB.as<uint32_t>(); pa.wave_off = c->w->wave_off.as<uint64_t>();
        pa.workspace = c->w->workspace.as<uint8_t>();
        pa.verdicts = io.verdicts; pa.n_final = n_final; pa.cap_len = cap_len; pa.slot_of = slot_of;
        pa.fired = io.fired; pa.facet_choice = io.facet_choice; pa.mode = io.audit ? 3 : io.mode;
        if (io.audit) {
            c->audit_cnt.need(n * AUDIT_COLS * 4); c->audit_len.need(n * AUDIT_COLS * 4);
            pa.audit_cnt = c->audit_cnt.as<uint32_t>(); pa.audit_len = c->audit_len.as<uint32_t>(); pa.review_bits = io.review_bits + f0;
        }
        hipEvent_t p0 = next_event(), p1 = next_event();
        record(p0);
        HIP_OK(launch_pipeline(c->dt, pa, s));
        record(p1);
        if (st.timed) st.pipe_ev.emplace_back(p0, p1);
    } else {
        c->w->tile_sums.need(scan_tiles_needed(n) * 8);
    }
    if (fused) {
        // back half, caption offsets and caption bytes of the wave-kernel files in one kernel; once more when it had to decline
        // files (the later stages finish them first) or when the caption buffer had to grow
        c->w->tile_state.need(lean_bx_tiles(n) * 8);
        c->w->slow_list_b.need(n * 4);
        bxa.slow_list = c->w->slow_list_b.as<uint32_t>(); bxa.slow_count = d_back_count;
        bxa.caption_offsets = io.caption_offsets; bxa.base_total = d_total + (st.chunk & 1); bxa.total_out = d_total + ((st.chunk + 1) & 1);
        bxa.tile_state = c->w->tile_state.as<unsigned long long>(); bxa.ticket = d_ticket; bxa.bx_counters = io.counters ? d_bx_counters : nullptr;
        bool declined_once = false;
        uint64_t total = 0;
        for (int pass = 0;; ++pass) {
            bxa.captions = io.captions; bxa.caption_capacity = st.over_capacity ? 0 : io.caption_capacity;
            HIP_OK(hipMemsetAsync(c->w->tile_state.p, 0, lean_bx_tiles(n) * 8, s));
            HIP_OK(hipMemsetAsync(sc_bytes + 64, 0, 72, s));        // the kernel's own counters, ticket, declined count
            hipEvent_t b0 = next_event(), b1 = next_event();
            record(b0);
            HIP_OK(launch_lean_bx(c->dt, bxa, s));
            record(b1);
            if (st.timed) st.fused_ev.emplace_back(b0, b1);
            HIP_OK(hipMemcpyAsync(&c->host_scalars[3], bxa.total_out, 8, hipMemcpyDeviceToHost, s));
            HIP_OK(hipMemcpyAsync(&c->host_scalars[5], d_back_count, 4, hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
            total = c->host_scalars[3];
            const uint32_t n_back = (uint32_t)c->host_scalars[5];
            if (n_back) {
                // the later wave stages take what the back half declined; a file they decline too would need the general kernel a
                // second time: the range then goes through the separate kernels from the start
                uint32_t* d_slow_count0 = (uint32_t*)(d_total + 6);
                bool give_up = declined_once || pass >= 3;
                if (!give_up) {
                    declined_once = true;
                    HIP_OK(hipMemsetAsync(d_slow_count + 1, 0, 8, s));
                    HIP_OK(hipMemsetAsync(d_slow_count0, 0, 4, s));
                    FastArgs fb = bxa;
                    fb.front_only = 0; fb.triage = 0;
                    fb.stage = 1; fb.item_list = c->w->slow_list_b.as<uint32_t>(); fb.item_count = d_back_count;
                    fb.slow_list = c->w->slow_list1.as<uint32_t>(); fb.slow_count = d_slow_count + 1;
                    HIP_OK(launch_fast(c->dt, fb, s));
                    fb.stage = 2; fb.item_list = c->w->slow_list1.as<uint32_t>(); fb.item_count = d_slow_count + 1;
                    fb.slow_list = c->w->slow_list2.as<uint32_t>(); fb.slow_count = d_slow_count + 2;
                    HIP_OK(launch_fast(c->dt, fb, s));
                    fb.stage = 3; fb.long_size = 2; fb.item_list = c->w->slow_list2.as<uint32_t>(); fb.item_count = d_slow_count + 2;
                    fb.slow_list = c->w->slow_list0.as<uint32_t>(); fb.slow_count = d_slow_count0;
                    HIP_OK(launch_fast(c->dt, fb, s));
                    HIP_OK(hipMemcpyAsync(&c->host_scalars[5], d_slow_count0, 4, hipMemcpyDeviceToHost, s));
                    HIP_OK(hipStreamSynchronize(s));
                    give_up = (uint32_t)c->host_scalars[5] != 0;
                }
                if (give_up) {
                    ++c->fused_reruns;
                    c->last_slow -= n_slow;
                    return run_range(c, io, f0, f1, st, false);
                }
                continue;
            }
            if (io.grow_captions && total > io.caption_capacity) {
                c->out_captions.need(std::max<uint64_t>(total + total / 8, 16), true, s);
                io.captions = c->out_captions.as<uint8_t>();
                io.caption_capacity = c->out_captions.cap;
                continue;
            }
            break;
        }
        st.total = total;
        ++st.chunk;
        if (st.total > io.caption_capacity) st.over_capacity = true;
        if (st.over_capacity) return TP_E_CAPACITY;
        if (io.counters) HIP_OK(launch_add_counters(io.counters, d_bx_counters, s));
    } else {
    // caption offsets of the range, continuing the running total
    HIP_OK(launch_scan_u32(c->w->cap_len.as<uint32_t>(), n, io.caption_offsets + f0, c->w->tile_sums.as<uint64_t>(), d_total + (st.chunk & 1),
                           d_total + ((st.chunk + 1) & 1), s));
    HIP_OK(hipMemcpyAsync(&c->host_scalars[3], d_total + ((st.chunk + 1) & 1), 8, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    st.total = c->host_scalars[3];
    ++st.chunk;
    if (io.mode != 0) return TP_OK;
    if (io.grow_captions) {
        c->out_captions.need(std::max<uint64_t>(st.total, 16), true, s);
        io.captions = c->out_captions.as<uint8_t>();
        io.caption_capacity = c->out_captions.cap;
    }
    if (st.total > io.caption_capacity) st.over_capacity = true;
    if (st.over_capacity) return TP_E_CAPACITY;
    }
    EmitArgs ea{};
    ea.file0 = f0; ea.file_end = f1;
    ea.wave_T = c->w->wave_T.as<uint32_t>(); ea.wave_LB = c->w->wave_LB.as<uint32_t>(); ea.wave_off = c->w->wave_off.as<uint64_t>(); ea.workspace = c->w->workspace.as<uint8_t>();
    ea.fin32 = want_ids ? c->w->fin32.as<uint32_t>() : nullptr; ea.fast_ran = use_fast ? 1 : 0; ea.wave_files_written = fused ? 1 : 0; ea.item_list = item_list; ea.n_items = n_slow; ea.slot_of = slot_of; ea.n_final = n_final;
    ea.caption_offsets = io.caption_offsets; ea.verdicts = io.verdicts; ea.captions = io.captions;
    ea.histogram = io.histogram; ea.counters = io.counters; ea.fpw = fpw_used;
    if (io.histogram && c->w == &c->own && n >= 4096) {
        c->w->hist_parts.need((size_t)HIST_PART_BLOCKS * HIST_PART_WORDS * 4);
        ea.hist_parts = c->w->hist_parts.as<uint32_t>();
    }
    hipEvent_t q0 = next_event(), q1 = next_event();
    record(q0);
    HIP_OK(launch_emit(c->dt, ea, s));
    record(q1);
    if (st.timed) st.emit_ev.emplace_back(q0, q1);
    if (io.audit) {
        // offsets of the range's audit columns, continuing the running total, then their text out of the workspaces (files of
        // the general kernel) and the id rows (files of the lean stage)
        c->audit_len.need(n * AUDIT_COLS * 4);
        uint64_t* d_atotal = d_total + 4;
        c->w->tile_sums.need(scan_tiles_needed(n * AUDIT_COLS) * 8);
        HIP_OK(launch_scan_u32(c->audit_len.as<uint32_t>(), n * AUDIT_COLS, io.audit_offsets + f0 * AUDIT_COLS, c->w->tile_sums.as<uint64_t>(),
                               d_atotal + (st.audit_chunk & 1), d_atotal + ((st.audit_chunk + 1) & 1), s));
        HIP_OK(hipMemcpyAsync(&c->host_scalars[4], d_atotal + ((st.audit_chunk + 1) & 1), 8, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        st.audit_total = c->host_scalars[4];
        ++st.audit_chunk;
        c->out_audit.need(std::max<uint64_t>(st.audit_total, 16), true, s);
        AuditArgs aa{};
        aa.file0 = f0; aa.n_items = n; aa.item_list = nullptr; aa.slot_of = slot_of; aa.fpw = fpw_used;
        aa.fin32 = audit_lean || audit_scored ? c->w->fin32.as<uint32_t>() : nullptr; aa.aud32 = audit_lean || audit_scored ? c->aud32.as<uint32_t>() : nullptr;
        aa.wave_T = c->w->wave_T.as<uint32_t>(); aa.wave_LB = c->w->wave_LB.as<uint32_t>(); aa.wave_off = c->w->wave_off.as<uint64_t>(); aa.workspace = c->w->workspace.as<uint8_t>();
        aa.audit_cnt = c->audit_cnt.as<uint32_t>(); aa.offsets = io.audit_offsets; aa.text = c->out_audit.as<uint8_t>();
        HIP_OK(launch_audit(c->dt, aa, s));
    }
    return TP_OK;
}

int run_device(tp_ctx* c, RunIo io, uint64_t* caption_bytes, uint64_t* audit_bytes = nullptr, bool timed = true) {
    HIP_OK(hipSetDevice(c->device));
    hipStream_t s = io.stream;
    const uint64_t n = io.n_files;
    c->timing = tp_timing{};
    c->last_slow = 0;
    c->w->scalars.need(SCALARS_BYTES);
    HIP_OK(hipMemsetAsync(c->w->scalars.p, 0, SCALARS_BYTES, s));
    if (n == 0) {
        if (io.audit) HIP_OK(hipMemsetAsync(io.audit_offsets, 0, 8, s));
        if (audit_bytes) *audit_bytes = 0;
        HIP_OK(hipMemsetAsync(io.caption_offsets, 0, 8, s));
        HIP_OK(hipStreamSynchronize(s));
        if (caption_bytes) *caption_bytes = 0;
        return TP_OK;
    }
    RunState st;
    st.timed = timed;
    hipEvent_t e_begin = timed ? c->event(0) : nullptr, e_end = timed ? c->event(1) : nullptr;
    if (timed) HIP_OK(hipEventRecord(e_begin, s));
    int rc = TP_OK;
    const uint64_t range_max = io.range_files ? io.range_files : (uint64_t)1 << 24;   // files per range: bounds the per-range index buffers
    for (uint64_t f0 = 0; f0 < n; f0 += range_max) {
        const uint64_t f1 = std::min(n, f0 + range_max), bytes0 = st.total;
        if (io.before_range) io.before_range(f0, f1);
        int r = run_range(c, io, f0, f1, st);
        if (r != TP_OK) rc = r;
        if (r != TP_OK && r != TP_E_CAPACITY) break;
        if (io.after_range && r == TP_OK) io.after_range(f0, f1, bytes0, st.total);
    }
    if (timed) HIP_OK(hipEventRecord(e_end, s));
    HIP_OK(hipStreamSynchronize(s));
    if (rc != TP_OK && rc != TP_E_CAPACITY) return rc;
    float ms = 0;
    if (timed) HIP_OK(hipEventElapsedTime(&ms, e_begin, e_end));
    c->timing.total_ms = ms;
    for (auto& p : st.fast_ev) { HIP_OK(hipEventElapsedTime(&ms, p.first, p.second)); c->timing.select_ms += ms; }
    for (auto& p : st.lean_ev) {
        HIP_OK(hipEventElapsedTime(&ms, p.begin, p.mid)); c->timing.lean_front_ms += ms;
        HIP_OK(hipEventElapsedTime(&ms, p.mid, p.end)); c->timing.lean_back_ms += ms;
    }
    for (auto& p : st.fused_ev) { HIP_OK(hipEventElapsedTime(&ms, p.first, p.second)); c->timing.fused_ms += ms; }
    for (auto& p : st.pipe_ev) { HIP_OK(hipEventElapsedTime(&ms, p.first, p.second)); c->timing.general_ms += ms; }
    for (auto& p : st.emit_ev) { HIP_OK(hipEventElapsedTime(&ms, p.first, p.second)); c->timing.emit_ms += ms; }
    c->timing.other_ms = c->timing.total_ms - c->timing.select_ms - c->timing.fused_ms - c->timing.general_ms - c->timing.emit_ms;
    c->timing.fused_reruns = c->fused_reruns;
    c->timing.general_files = c->last_slow;
    if (caption_bytes) *caption_bytes = st.total;
    if (audit_bytes) *audit_bytes = st.audit_total;
    if (st.over_capacity) return fail(TP_E_CAPACITY, "caption buffer too small: " + std::to_string(st.total) + " bytes needed");
    return TP_OK;
}

// Cuts this device's static working memory into the pieces a run of at most TP_NOALLOC_MAX_FILES files works in;
// what is left over is the general kernel's workspace.  The device of the context is current.
void bind_fixed(tp_ctx* c) {
    void* base = nullptr;
    size_t bytes = 0;
    HIP_OK(noalloc_scratch(c->device, &base, &bytes));
    size_t o = 0;
    auto take = [&](DevBuf& b, size_t sz) {
        sz = (sz + 255) & ~(size_t)255;
        if (o + sz > bytes) throw std::bad_alloc();
        b.bind((uint8_t*)base + o, sz);
        o += sz;
    };
    const size_t N = TP_NOALLOC_MAX_FILES;
    WorkBufs& w = c->fixed;
    take(w.scalars, SCALARS_BYTES);
    for (DevBuf* b : {&w.cap_len, &w.n_final, &w.slot_of, &w.slow_list, &w.slow_list0, &w.slow_list1, &w.slow_list2, &w.slow_list_b, &w.wave_T, &w.wave_LB}) take(*b, N * 4);
    take(w.tile_state, lean_bx_tiles(N) * 8);
    take(w.wave_bytes, N * 8);
    take(w.wave_off, (N + 1) * 8);
    take(w.tile_sums, scan_tiles_needed(N) * 8);
    take(w.fin32, N * fast_fin_stride() * 4);
    take(w.workspace, (bytes - o) & ~(size_t)255);
}

template <class F>
int guarded(F&& body) {
    try {
        return body();
    } catch (const ScratchFull& f) {
        return fail(TP_E_NOMEM, "this input needs " + std::to_string(f.bytes) + " bytes in a piece of the fixed working memory that holds " + std::to_string(f.cap));
    } catch (const HipFail& h) {
        return fail(TP_E_DEVICE, std::string("HIP error: ") + hipGetErrorString(h.e) + " in " + h.what);
    } catch (const ConfigError& e) {
        return fail(TP_E_CONFIG, e.what());
    } catch (const std::bad_alloc&) {
        return fail(TP_E_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(TP_E_INVALID_ARG, e.what());
    }
}

// Host-pointer run shared by tp_run_lines and tp_run_scored.
int run_host(tp_ctx* c, const uint8_t* arena, uint64_t arena_bytes, const uint64_t* offsets, uint64_t n_offsets, const uint64_t* tag_begin,
             const double* scores, const uint8_t* sources, uint64_t n_tags, uint64_t n_files, tp_result* out,
             const uint64_t* secondary_begin = nullptr, int secondary_cap = 0, const uint8_t* reply_flags = nullptr) {
    HIP_OK(hipSetDevice(c->device));
    const DevTables& t = c->dt;
    // Tag lines in pieces of PIECE files when there are several of them: while the kernels work on one piece the next one's
    // lines are on their way to the device and the captions of the one before on their way back (the bus carries both
    // directions at once, and either copy takes several times as long as the kernels).  Audit mode and tagger replies go in turn.
    const uint64_t PIECE = 131072;
    const bool overlap = !tag_begin && !c->audit && !c->no_overlap && n_files >= 2 * PIECE;
    hipStream_t s = overlap ? c->pipe_stream(1) : nullptr;
    hipStream_t s_in = overlap ? c->pipe_stream(0) : s, s_out = overlap ? c->pipe_stream(2) : s;
    const uint64_t n_pieces = overlap ? (n_files + PIECE - 1) / PIECE : 0;
    c->in_arena.need(arena_bytes + 64);
    c->in_offsets.need(n_offsets * 8);
    auto piece_in = [&](uint64_t k) {                               // piece k's lines to the device, then its event
        if (k >= n_pieces) return;
        const uint64_t f0 = k * PIECE, f1 = std::min(n_files, f0 + PIECE), b0 = offsets[f0], b1 = offsets[f1];
        if (b1 > b0) HIP_OK(hipMemcpyAsync((uint8_t*)c->in_arena.p + b0, arena + b0, b1 - b0, hipMemcpyHostToDevice, s_in));
        if (f1 == n_files) HIP_OK(hipMemsetAsync((uint8_t*)c->in_arena.p + arena_bytes, 0, 64, s_in));
        HIP_OK(hipEventRecord(c->pipe_event(1 + k), s_in));
    };
    if (overlap) {
        HIP_OK(hipMemcpyAsync(c->in_offsets.p, offsets, n_offsets * 8, hipMemcpyHostToDevice, s_in));
        HIP_OK(hipEventRecord(c->pipe_event(0), s_in));
        HIP_OK(hipStreamWaitEvent(s, c->pipe_event(0), 0));
        piece_in(0);
    } else {
    if (arena_bytes) HIP_OK(hipMemcpyAsync(c->in_arena.p, arena, arena_bytes, hipMemcpyHostToDevice, s));
    HIP_OK(hipMemsetAsync((uint8_t*)c->in_arena.p + arena_bytes, 0, 64, s));   // the lean stage reads whole 16-byte pieces: what lies behind the last line is not left to chance (a byte outside ASCII there sends that file the long way)
    HIP_OK(hipMemcpyAsync(c->in_offsets.p, offsets, n_offsets * 8, hipMemcpyHostToDevice, s));
    }
    if (tag_begin) {
        c->in_tag_begin.need((n_files + 1) * 8); c->in_scores.need(std::max<uint64_t>(n_tags, 1) * 8); c->in_sources.need(std::max<uint64_t>(n_tags, 1));
        HIP_OK(hipMemcpyAsync(c->in_tag_begin.p, tag_begin, (n_files + 1) * 8, hipMemcpyHostToDevice, s));
        if (n_tags) {
            HIP_OK(hipMemcpyAsync(c->in_scores.p, scores, n_tags * 8, hipMemcpyHostToDevice, s));
            if (sources) HIP_OK(hipMemcpyAsync(c->in_sources.p, sources, n_tags, hipMemcpyHostToDevice, s));
        }
        if (secondary_begin) {
            c->in_secondary.need(std::max<uint64_t>(n_files, 1) * 8);
            if (n_files) HIP_OK(hipMemcpyAsync(c->in_secondary.p, secondary_begin, n_files * 8, hipMemcpyHostToDevice, s));
        }
        if (reply_flags) {
            c->in_reply_flags.need(std::max<uint64_t>(n_files, 1));
            if (n_files) HIP_OK(hipMemcpyAsync(c->in_reply_flags.p, reply_flags, n_files, hipMemcpyHostToDevice, s));
        }
    }
    const size_t hist_n = (size_t)t.n_public + 1;
    c->out_offsets.need((n_files + 1) * 8);
    c->out_verdicts.need(std::max<uint64_t>(n_files, 1) * sizeof(tp_verdict));
    c->out_hist.need(hist_n * 8);
    c->out_counters.need(sizeof(tp_counters));
    c->out_fired.need(std::max<uint64_t>(n_files * t.fired_words, 1) * 4);
    c->out_facets.need(std::max<uint64_t>(n_files * t.n_facets, 1) * 2);
    c->out_captions.need(std::max<uint64_t>(overlap ? 2 * arena_bytes + n_files * 128 : arena_bytes + n_files * 96, 4096));
    HIP_OK(hipMemsetAsync(c->out_hist.p, 0, hist_n * 8, s));
    HIP_OK(hipMemsetAsync(c->out_counters.p, 0, sizeof(tp_counters), s));

    RunIo io{};
    io.arena = c->in_arena.as<uint8_t>(); io.offsets = c->in_offsets.as<uint64_t>();
    io.tag_begin = tag_begin ? c->in_tag_begin.as<uint64_t>() : nullptr;
    io.scores = tag_begin ? c->in_scores.as<double>() : nullptr;
    io.sources = tag_begin ? c->in_sources.as<uint8_t>() : nullptr;
    io.secondary_begin = secondary_begin ? c->in_secondary.as<uint64_t>() : nullptr; io.secondary_cap = secondary_cap;
    io.reply_flags = tag_begin && reply_flags ? c->in_reply_flags.as<uint8_t>() : nullptr;
    io.n_files = n_files;
    io.captions = c->out_captions.as<uint8_t>(); io.caption_capacity = c->out_captions.cap; io.grow_captions = true;
    io.caption_offsets = c->out_offsets.as<uint64_t>(); io.verdicts = c->out_verdicts.as<tp_verdict>();
    io.histogram = c->out_hist.as<uint64_t>(); io.counters = c->out_counters.as<tp_counters>();
    io.fired = c->out_fired.as<uint32_t>(); io.facet_choice = c->out_facets.as<uint16_t>();
    io.stream = s; io.mode = 0;
    io.audit = c->audit;
    if (io.audit) {
        c->out_audit_offsets.need((n_files * AUDIT_COLS + 1) * 8);
        io.audit_offsets = c->out_audit_offsets.as<uint64_t>();
        c->out_review_bits.need(std::max<uint64_t>(n_files, 1) * 4);
        io.review_bits = c->out_review_bits.as<uint32_t>();
    }
    uint64_t cap_bytes = 0, audit_bytes = 0;
    std::unique_ptr<HostResult> r(new HostResult());
    bool captions_back = false;                                   // the pieces' captions went back as they were written
    if (overlap) {
        // the result's arrays exist before the run; the captions' size is a guess (twice the input and 128 bytes a file: an
        // array that turns out too small is given up for one copy at the end)
        r->captions.resize(c->out_captions.cap);
        r->caption_offsets.resize(n_files + 1);
        r->verdicts.resize(n_files);
        r->fired.resize(std::max<uint64_t>(n_files * t.fired_words, 1));
        r->facet_choice.resize(std::max<uint64_t>(n_files * t.n_facets, 1));
        const uint64_t host_cap = c->out_captions.cap;
        captions_back = true;
        io.range_files = PIECE;
        io.before_range = [&](uint64_t f0, uint64_t) {
            const uint64_t k = f0 / PIECE;
            piece_in(k + 1);                                       // the next piece's lines start out before this piece's kernels are queued
            HIP_OK(hipStreamWaitEvent(s, c->pipe_event(1 + k), 0));
        };
        io.after_range = [&](uint64_t f0, uint64_t f1, uint64_t b0, uint64_t b1) {
            const uint64_t k = f0 / PIECE;
            hipEvent_t done = c->pipe_event(1 + n_pieces + k);
            HIP_OK(hipEventRecord(done, s));
            HIP_OK(hipStreamWaitEvent(s_out, done, 0));
            if (b1 > host_cap) captions_back = false;
            if (captions_back && b1 > b0) HIP_OK(hipMemcpyAsync(r->captions.data() + b0, (const uint8_t*)c->out_captions.p + b0, b1 - b0, hipMemcpyDeviceToHost, s_out));
            HIP_OK(hipMemcpyAsync(r->caption_offsets.data() + f0, (const uint64_t*)c->out_offsets.p + f0, (f1 - f0 + 1) * 8, hipMemcpyDeviceToHost, s_out));
            HIP_OK(hipMemcpyAsync(r->verdicts.data() + f0, (const tp_verdict*)c->out_verdicts.p + f0, (f1 - f0) * sizeof(tp_verdict), hipMemcpyDeviceToHost, s_out));
            if (t.fired_words) HIP_OK(hipMemcpyAsync(r->fired.data() + f0 * t.fired_words, (const uint32_t*)c->out_fired.p + f0 * t.fired_words, (f1 - f0) * t.fired_words * 4, hipMemcpyDeviceToHost, s_out));
            if (t.n_facets) HIP_OK(hipMemcpyAsync(r->facet_choice.data() + f0 * t.n_facets, (const uint16_t*)c->out_facets.p + f0 * t.n_facets, (f1 - f0) * t.n_facets * 2, hipMemcpyDeviceToHost, s_out));
        };
    }
    int rc = run_device(c, io, &cap_bytes, &audit_bytes);
    if (rc != TP_OK) { if (overlap) { (void)hipStreamSynchronize(s_in); (void)hipStreamSynchronize(s_out); } return rc; }

    r->histogram.resize(hist_n);
    tp_counters ctr{};
    if (overlap) {
        HIP_OK(hipStreamSynchronize(s_out));                       // (the run's own stream is idle: run_device waited for it)
        if (!captions_back) { r->captions.resize(std::max<uint64_t>(cap_bytes, 1)); if (cap_bytes) HIP_OK(hipMemcpyAsync(r->captions.data(), c->out_captions.p, cap_bytes, hipMemcpyDeviceToHost, s)); }
        HIP_OK(hipMemcpyAsync(r->histogram.data(), c->out_hist.p, hist_n * 8, hipMemcpyDeviceToHost, s));
    } else {
    r->captions.resize(std::max<uint64_t>(cap_bytes, 1));
    r->caption_offsets.resize(n_files + 1);
    r->verdicts.resize(std::max<uint64_t>(n_files, 1));
    r->fired.resize(std::max<uint64_t>(n_files * t.fired_words, 1));
    r->facet_choice.resize(std::max<uint64_t>(n_files * t.n_facets, 1));
    // every copy back is queued on the stream and waited for once (the buffers are page-locked: the copies are DMA transfers)
    if (cap_bytes) HIP_OK(hipMemcpyAsync(r->captions.data(), c->out_captions.p, cap_bytes, hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(r->caption_offsets.data(), c->out_offsets.p, (n_files + 1) * 8, hipMemcpyDeviceToHost, s));
    if (n_files) HIP_OK(hipMemcpyAsync(r->verdicts.data(), c->out_verdicts.p, n_files * sizeof(tp_verdict), hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(r->histogram.data(), c->out_hist.p, hist_n * 8, hipMemcpyDeviceToHost, s));
    if (n_files * t.fired_words) HIP_OK(hipMemcpyAsync(r->fired.data(), c->out_fired.p, n_files * t.fired_words * 4, hipMemcpyDeviceToHost, s));
    if (n_files * t.n_facets) HIP_OK(hipMemcpyAsync(r->facet_choice.data(), c->out_facets.p, n_files * t.n_facets * 2, hipMemcpyDeviceToHost, s));
    }
    if (io.audit) {
        r->audit_text.resize(std::max<uint64_t>(audit_bytes, 1));
        r->audit_offsets.resize(n_files * AUDIT_COLS + 1);
        if (audit_bytes) HIP_OK(hipMemcpyAsync(r->audit_text.data(), c->out_audit.p, audit_bytes, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(r->audit_offsets.data(), c->out_audit_offsets.p, (n_files * AUDIT_COLS + 1) * 8, hipMemcpyDeviceToHost, s));
        r->review_bits.resize(std::max<uint64_t>(n_files, 1));
        if (n_files) HIP_OK(hipMemcpyAsync(r->review_bits.data(), c->out_review_bits.p, n_files * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_OK(hipMemcpyAsync(&c->host_scalars[0], c->out_counters.p, sizeof ctr, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    std::memcpy(&ctr, &c->host_scalars[0], sizeof ctr);
    out->n_files = n_files;
    out->captions = r->captions.data();
    out->caption_offsets = r->caption_offsets.data();
    out->verdicts = r->verdicts.data();
    out->counters = ctr;
    out->histogram = r->histogram.data();
    out->histogram_other = r->histogram[t.n_public];
    out->fired = r->fired.data();
    out->facet_choice = r->facet_choice.data();
    out->opaque = r.release();
    return TP_OK;
}

// SelectTags of the minimal fallback line (Program.cs:298), evaluated once with the pipeline kernel itself.
void compute_fallback(tp_ctx* c) {
    const std::vector<uint8_t>& line = c->ht.fallback_line;
    hipStream_t s = nullptr;
    c->in_arena.need(line.size() + 64);
    c->in_offsets.need(16);
    uint64_t offs[2] = {0, line.size()};
    if (!line.empty()) HIP_OK(hipMemcpy(c->in_arena.p, line.data(), line.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(c->in_offsets.p, offs, 16, hipMemcpyHostToDevice));
    c->out_offsets.need(16); c->out_verdicts.need(sizeof(tp_verdict));
    RunIo io{};
    io.arena = c->in_arena.as<uint8_t>(); io.offsets = c->in_offsets.as<uint64_t>(); io.n_files = 1;
    io.caption_offsets = c->out_offsets.as<uint64_t>(); io.verdicts = c->out_verdicts.as<tp_verdict>();
    io.stream = s; io.mode = 1;
    // The fallback line always takes the line-ingest defaults of Program.cs:298: weight 1.0, source OpenAI.
    DevTables saved = c->dt;
    c->dt.line_weight = 1.0;
    c->dt.line_source = TP_SRC_OPENAI;
    int rc = run_device(c, io, nullptr);
    c->dt = saved;
    if (rc != TP_OK) throw ConfigError("could not evaluate the minimal fallback line: " + g_error);
    uint32_t nfin = 0;
    HIP_OK(hipMemcpy(&nfin, c->w->n_final.p, 4, hipMemcpyDeviceToHost));
    tp_verdict v{};
    HIP_OK(hipMemcpy(&v, c->out_verdicts.p, sizeof v, hipMemcpyDeviceToHost));
    if (v.flags & TP_F_UNSUPPORTED) throw ConfigError("the minimal fallback line holds text this build does not process");
    if (nfin > (uint32_t)MAX_FALLBACK) throw ConfigError("internal: fallback selection too long");
    // the final list of file 0 sits in lane 0 of the first wavefront's workspace
    uint32_t T = 0, LB = 0;
    HIP_OK(hipMemcpy(&T, c->w->wave_T.p, 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&LB, c->w->wave_LB.p, 4, hipMemcpyDeviceToHost));
    WsLayout lay = ws_layout(T, LB, c->dt.final_cap, c->dt.evidence_grow, c->dt.lit_words);
    std::vector<uint32_t> ids(nfin);
    for (uint32_t k = 0; k < nfin; ++k) {
        uint64_t h = 0;
        HIP_OK(hipMemcpy(&h, c->w->workspace.as<uint8_t>() + lay.o_fin + (size_t)k * c->last_fpw * 8, 8, hipMemcpyDeviceToHost));
        if (h >> 32) throw ConfigError("internal: fallback tag outside the tables");
        ids[k] = (uint32_t)h;
    }
    set_fallback(c->ht, c->dt, ids.data(), nfin);
}

}  // namespace

namespace {
// Rows of a log file for files [0, n): row_for(i, row) appends file i's row (without the newline; nothing for a file that has
// none) and returns false on an error it has reported with fail().  The files are cut into contiguous chunks, one per host
// thread; every thread puts its chunk's rows together, then the chunks are copied to their places in `out`, again in parallel.
// Conventions of the tp_format_* calls: out_offsets has n + 1 entries, *needed receives the size, TP_E_CAPACITY when it
// does not fit (nothing is written then).
template <class RowFor>
int format_rows_parallel(uint64_t n, const std::string& nl, RowFor&& row_for, uint8_t* out, uint64_t capacity, uint64_t* out_offsets, uint64_t* needed) {
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("TAGPIPE_FORMAT_THREADS")) { const long v = std::strtol(e, nullptr, 10); if (v >= 1 && v <= 256) hw = (unsigned)v; }
    const uint64_t want = n / 2048 + 1;
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(hw ? hw : 1, 32), want));
    std::vector<std::string> text(T);
    std::vector<int> ok(T, 1);
    std::vector<std::string> err(T);
    auto lo_of = [&](unsigned k) { return n * k / T; };
    auto work = [&](unsigned k) {
        std::string row;
        std::string& rows = text[k];
        for (uint64_t i = lo_of(k); i < lo_of(k + 1); ++i) {
            out_offsets[i] = rows.size();                          // relative to the chunk for now
            row.clear();
            if (!row_for(i, row)) { ok[k] = 0; err[k] = g_error; return; }
            if (!row.empty()) { rows += row; rows += nl; }
        }
    };
    auto run_all = [&](auto&& fn) {
        if (T == 1) { fn(0u); return; }
        std::vector<std::thread> th;
        for (unsigned k = 1; k < T; ++k) th.emplace_back([&fn, k]() { fn(k); });
        fn(0u);
        for (auto& x : th) x.join();
    };
    run_all(work);
    for (unsigned k = 0; k < T; ++k) if (!ok[k]) return fail(TP_E_INVALID_ARG, err[k]);
    std::vector<uint64_t> base(T + 1, 0);
    for (unsigned k = 0; k < T; ++k) base[k + 1] = base[k] + text[k].size();
    const uint64_t total = base[T];
    const bool fits = total == 0 || (out && capacity >= total);
    run_all([&](unsigned k) {
        for (uint64_t i = lo_of(k); i < lo_of(k + 1); ++i) out_offsets[i] += base[k];
        if (fits && !text[k].empty()) std::memcpy(out + base[k], text[k].data(), text[k].size());
    });
    out_offsets[n] = total;
    *needed = total;
    if (!fits) return fail(TP_E_CAPACITY, "output buffer too small");
    return TP_OK;
}
}  // namespace


extern "C" {

const char* tp_last_error(void) { return g_error.c_str(); }

// Tags in no table, learned.  A run without a strict vocabulary keeps such tags, and a wave kernel that meets one hands
// its file to the thread-per-file kernel, which is a hundred times slower per file.  The lean stage therefore notes every
// such string once (DevTables::unk_keys), the entry points that take tag lines collect them when a run is over, and the
// tables are compiled again with those strings interned behind the configuration's own (same ids for those, same attributes
// for the new ones as the pipeline computes for a string in no table, counted with the tags in no table in the histogram).
// The files of the run that met them went the slow way; later runs find them in the tables.
constexpr uint32_t UNK_SLOTS = 1u << 20, UNK_HARVEST = 1u << 18, UNK_LEARN_MAX = 400000;

void own_config(tp_ctx* c, const tp_config& cfg) {
    c->cfg = cfg;
    auto keep = [&](const char*& p) { if (p) { c->cfg_text.emplace_back(new std::string(p)); p = c->cfg_text.back()->c_str(); } };
    auto keep_src = [&](tp_source& src) {
        keep(src.path);
        if (src.data) { c->cfg_text.emplace_back(new std::string((const char*)src.data, (size_t)src.len)); src.data = (const uint8_t*)c->cfg_text.back()->data(); }
    };
    for (tp_source* src : {&c->cfg.lexicon, &c->cfg.actions, &c->cfg.required_facets, &c->cfg.profile, &c->cfg.vocabulary, &c->cfg.allowlist, &c->cfg.denylist}) keep_src(*src);
    keep(c->cfg.trigger); keep(c->cfg.auto_prefix_tags); keep(c->cfg.minimal_subject_tags); keep(c->cfg.newline);
}

void attach_unknown_set(tp_ctx* c) {
    c->dt.unk_keys = c->learning ? c->unk_keys.as<unsigned long long>() : nullptr;
    c->dt.unk_ent = c->learning ? c->unk_ent.as<unsigned long long>() : nullptr;
    c->dt.unk_text = c->learning ? c->unk_text.as<uint8_t>() : nullptr;
    c->dt.unk_count = c->learning ? c->unk_count.as<uint32_t>() : nullptr;
    c->dt.unk_mask = UNK_SLOTS - 1;
    HIP_OK(hipMemcpy(c->dt_dev.p, &c->dt, sizeof(DevTables), hipMemcpyHostToDevice));
}

// Compiles the configuration (plus the learned strings) and puts the result in place of the context's tables.
void load_tables(tp_ctx* c, const std::vector<std::string>* learned) {
    HostTables fresh;
    compile_tables(c->cfg, fresh, learned);
    std::vector<void*> old_mem;
    old_mem.swap(c->table_mem);
    c->ht = std::move(fresh);
    c->string_ids.clear();
    const HostTables& h = c->ht;
    DevTables& d = c->dt;
    d = h.scalars;
    d.str_bytes = upload(c->table_mem, h.str_bytes);
    d.str_off = upload(c->table_mem, h.str_off);
    d.flags = upload(c->table_mem, h.flags);
    d.next = upload(c->table_mem, h.next);
    d.alias1 = upload(c->table_mem, h.alias1);
    d.lit = upload(c->table_mem, h.lit);
    d.imp_off = upload(c->table_mem, h.imp_off);
    d.imp_ids = upload(c->table_mem, h.imp_ids);
    d.ht_hash = upload(c->table_mem, h.ht_hash);
    d.ht_id = upload(c->table_mem, h.ht_id);
    d.dfa_next = upload(c->table_mem, h.dfa_next);
    d.dfa_out = upload(c->table_mem, h.dfa_out);
    d.pool = upload(c->table_mem, h.pool);
    d.rules = upload(c->table_mem, h.rules);
    d.subs = upload(c->table_mem, h.subs);
    d.facets = upload(c->table_mem, h.facets);
    d.facet_rules = upload(c->table_mem, h.facet_rules);
    d.rec = upload(c->table_mem, h.rec);
    d.ht_info = upload(c->table_mem, h.ht_info);
    d.str_words = upload(c->table_mem, h.str_words);
    d.rc = upload(c->table_mem, h.rule_cache);
    d.slots = upload(c->table_mem, h.slots);
    d.lslots = h.lslots.empty() ? nullptr : upload(c->table_mem, h.lslots);
    d.lkey_hi = upload(c->table_mem, h.lkey_hi);
    d.attr2 = upload(c->table_mem, h.attr2);
    d.emit_rec = upload(c->table_mem, h.emit_rec);
    d.emit_rec32 = upload(c->table_mem, h.emit_rec32);
    d.case_map = upload(c->table_mem, h.case_map);
    d.crec = upload(c->table_mem, h.crec);
    d.fw_tix = upload(c->table_mem, h.fw_tix);
    d.step_off = upload(c->table_mem, h.step_off);
    d.step_ids = upload(c->table_mem, h.step_ids);
    d.reach_off = upload(c->table_mem, h.reach_off);
    d.reach_ent = upload(c->table_mem, h.reach_ent);
    compute_fallback(c);
    c->use_lean = lean_ok(c->dt);
    if (const char* b = std::getenv("TAGPIPE_NO_LEAN")) if (b[0] == '1') c->use_lean = false;
    if (const char* b = std::getenv("TAGPIPE_LEAN_ADAPT")) c->lean_adapt = b[0] != '0';
    if (const char* b = std::getenv("TAGPIPE_SIDE_CHAIN")) { c->side_adapt = false; c->side_chain = b[0] == '1'; }
    if (const char* b = std::getenv("TAGPIPE_NO_OVERLAP")) c->no_overlap = b[0] == '1';
    c->dt_dev.need(sizeof(DevTables));
    HIP_OK(hipMemcpy(c->dt_dev.p, &c->dt, sizeof(DevTables), hipMemcpyHostToDevice));
    for (void* p : old_mem) (void)hipFree(p);
}

// After a run over tag lines: the strings the lean stage noted are collected and, when there are new ones, the tables are
// compiled again with them.  The stream must be idle.
int learn_unknown_strings(tp_ctx* c, hipStream_t s) {
    if (!c->learning) return TP_OK;
    uint32_t noted = 0;
    HIP_OK(hipMemcpy(&noted, c->unk_count.p, 4, hipMemcpyDeviceToHost));
    if (noted == 0) return TP_OK;
    uint32_t* d_out_count = c->unk_count.as<uint32_t>() + 1;
    HIP_OK(hipMemsetAsync(c->unk_count.p, 0, 8, s));
    HIP_OK(launch_collect_unknown(c->unk_keys.as<unsigned long long>(), c->unk_ent.as<unsigned long long>(), c->unk_text.as<uint8_t>(), UNK_SLOTS, c->unk_out.as<uint8_t>(), UNK_HARVEST, d_out_count, s));
    uint32_t got = 0;
    HIP_OK(hipMemcpyAsync(&got, d_out_count, 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    got = std::min(got, UNK_HARVEST);
    std::vector<uint8_t> recs((size_t)got * 40);
    if (got) HIP_OK(hipMemcpy(recs.data(), c->unk_out.p, recs.size(), hipMemcpyDeviceToHost));
    if (c->string_ids.empty()) c->string_ids = index_strings(c->ht);
    size_t fresh = 0;
    for (uint32_t i = 0; i < got; ++i) {
        uint32_t n = 0;
        std::memcpy(&n, &recs[(size_t)i * 40 + 32], 4);
        if (n == 0 || n > 32) continue;
        std::string str((const char*)&recs[(size_t)i * 40], n);
        if (c->string_ids.count(str) || c->learned_set.count(str)) continue;
        c->learned_set.emplace(str, (uint32_t)c->learned.size());
        c->learned.push_back(std::move(str));
        ++fresh;
    }
    if (fresh == 0) return TP_OK;
    if (c->learned.size() > UNK_LEARN_MAX) {                       // no room for more: what is learned so far stays, nothing further is noted
        c->learned.resize(c->learned.size() - fresh);
        c->learning = false;
        attach_unknown_set(c);
        return TP_OK;
    }
    const tp_timing run_timing = c->timing;                        // (the tables' load evaluates the fallback line with a run of its own)
    const uint64_t run_slow = c->last_slow;
    load_tables(c, &c->learned);
    c->timing = run_timing;
    c->last_slow = run_slow;
    if (!c->use_lean) c->learning = false;
    attach_unknown_set(c);
    return TP_OK;
}

void tp_config_init(tp_config* cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg;
    cfg->vocabulary_is_json = -1;
    cfg->trigger = "sgrui";
    cfg->auto_prefix_tags = "";
    cfg->minimal_subject_tags = "";
    cfg->newline = "\n";
    cfg->safe_mode = 1;
    cfg->line_weight = 1.0;
    cfg->line_source = TP_SRC_OPENAI;
}

tp_ctx* tp_create(const tp_config* cfg) {
    if (!cfg || cfg->struct_size != sizeof(tp_config)) { fail(TP_E_INVALID_ARG, "tp_config is null or has the wrong struct_size"); return nullptr; }
    std::unique_ptr<tp_ctx> c(new tp_ctx());
    int rc = guarded([&]() {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0) return fail(TP_E_DEVICE, "no HIP device is available: this library only runs on a GPU");
        if (cfg->device < 0 || cfg->device >= count) return fail(TP_E_INVALID_ARG, "device ordinal out of range");
        c->device = cfg->device;
        HIP_OK(hipSetDevice(c->device));
        if (const char* b = std::getenv("TAGPIPE_WORKSPACE_BYTES")) { unsigned long long v = std::strtoull(b, nullptr, 10); if (v >= (1ull << 20)) c->ws_budget = (size_t)v; }
        if (const char* b = std::getenv("TAGPIPE_NO_FAST")) c->no_fast = b[0] == '1';
        if (const char* b = std::getenv("TAGPIPE_FUSED")) c->no_fused = b[0] != '1';
        HIP_OK(hipHostMalloc((void**)&c->host_scalars, 64, hipHostMallocDefault));
        std::memset(c->host_scalars, 0, 64);
        own_config(c.get(), *cfg);
        load_tables(c.get(), nullptr);
        c->public_strings = c->ht.strings;
        // Tags in no table are kept when no strict vocabulary drops them; the lean stage notes the ones it meets so that
        // later calls find them interned (TAGPIPE_NO_LEARN=1: off)
        c->learning = c->use_lean && !c->dt.drop_unknown;
        if (const char* b = std::getenv("TAGPIPE_NO_LEARN")) if (b[0] == '1') c->learning = false;
        if (c->learning) {
            c->unk_keys.need((size_t)UNK_SLOTS * 8); c->unk_ent.need((size_t)UNK_SLOTS * 8); c->unk_text.need((size_t)UNK_SLOTS * 32); c->unk_count.need(64); c->unk_out.need((size_t)UNK_HARVEST * 40);
            HIP_OK(hipMemset(c->unk_keys.p, 0, (size_t)UNK_SLOTS * 8));
            HIP_OK(hipMemset(c->unk_count.p, 0, 64));
            attach_unknown_set(c.get());
        }
        return TP_OK;
    });
    if (rc != TP_OK) return nullptr;
    return c.release();
}

void tp_destroy(tp_ctx* ctx) { delete ctx; }

int tp_get_table_info(const tp_ctx* ctx, tp_table_info* out) {
    if (!ctx || !out) return fail(TP_E_INVALID_ARG, "null argument");
    const DevTables& t = ctx->dt;
    out->n_strings = t.n_public;                                   // learned strings have no histogram entry of their own and no public id
    out->n_action_rules = t.n_rules;
    out->n_action_subtypes = t.n_subs;
    out->n_facets = t.n_facets;
    out->fired_words = t.fired_words;
    out->n_known_tags = ctx->ht.n_known;
    out->strict_known = ctx->ht.strict ? 1 : 0;
    out->known_found = ctx->ht.known_found ? 1 : 0;
    return TP_OK;
}

int tp_get_string(const tp_ctx* ctx, uint32_t id, const char** utf8, uint32_t* len) {
    if (!ctx || !utf8 || !len || id >= ctx->public_strings.size()) return fail(TP_E_INVALID_ARG, "bad string id");
    *utf8 = ctx->public_strings[id].c_str();
    *len = (uint32_t)ctx->public_strings[id].size();
    return TP_OK;
}

int tp_get_name(const tp_ctx* ctx, int kind, uint32_t index, const char** utf8, uint32_t* len) {
    if (!ctx || !utf8 || !len) return fail(TP_E_INVALID_ARG, "null argument");
    const HostTables& h = ctx->ht;
    const std::string* s = nullptr;
    switch (kind) {
        case TP_NAME_ACTION_RULE: if (index < h.rule_names.size()) s = &h.rule_names[index]; break;
        case TP_NAME_ACTION_SUBTYPE: if (index < h.sub_names.size()) s = &h.sub_names[index]; break;
        case TP_NAME_FACET: if (index < h.facet_names.size()) s = &h.facet_names[index]; break;
        case TP_NAME_FACET_DEFAULT: if (index < h.facet_defaults.size()) s = &h.facet_defaults[index]; break;
        case TP_NAME_FACET_RULE: {
            uint32_t f = index >> 16, r = index & 0xFFFF;
            if (f < h.facet_rule_tags.size() && r < h.facet_rule_tags[f].size()) s = &h.facet_rule_tags[f][r];
            break;
        }
        case TP_NAME_SAFE_TAG: if (index < h.safe_tags.size()) s = &h.safe_tags[index]; break;
        default: break;
    }
    if (!s) return fail(TP_E_INVALID_ARG, "name index out of range");
    *utf8 = s->c_str();
    *len = (uint32_t)s->size();
    return TP_OK;
}

int tp_run_lines(tp_ctx* ctx, const uint8_t* arena, const uint64_t* offsets, uint64_t n_files, tp_result* out) {
    if (!ctx || !out || !offsets || (n_files && offsets[n_files] && !arena)) return fail(TP_E_INVALID_ARG, "null argument");
    for (uint64_t i = 0; i < n_files; ++i) if (offsets[i + 1] < offsets[i]) return fail(TP_E_INVALID_ARG, "offsets must be non-decreasing");
    return guarded([&]() {
        const int rc = run_host(ctx, arena, offsets[n_files], offsets, n_files + 1, nullptr, nullptr, nullptr, 0, n_files, out);
        if (rc == TP_OK) { const int lrc = learn_unknown_strings(ctx, nullptr); if (lrc != TP_OK) return lrc; }
        return rc;
    });
}

int tp_run_merged(tp_ctx* ctx, const uint8_t* arena, const uint64_t* tag_offsets, const uint64_t* tag_begin, const uint64_t* secondary_begin,
                  const double* scores, int secondary_cap, uint64_t n_files, tp_result* out) {
    if (!ctx || !out || !tag_offsets || !tag_begin || (n_files && !secondary_begin)) return fail(TP_E_INVALID_ARG, "null argument");
    for (uint64_t i = 0; i < n_files; ++i) {
        if (tag_begin[i + 1] < tag_begin[i]) return fail(TP_E_INVALID_ARG, "tag_begin must be non-decreasing");
        if (secondary_begin[i] < tag_begin[i] || secondary_begin[i] > tag_begin[i + 1]) return fail(TP_E_INVALID_ARG, "secondary_begin must lie within the file's tags");
    }
    uint64_t n_tags = tag_begin[n_files];
    if (n_tags && !scores) return fail(TP_E_INVALID_ARG, "null argument");
    for (uint64_t i = 0; i < n_tags; ++i) {
        if (tag_offsets[i + 1] < tag_offsets[i]) return fail(TP_E_INVALID_ARG, "tag_offsets must be non-decreasing");
        if (!(scores[i] == scores[i]) || scores[i] - scores[i] != 0.0) return fail(TP_E_INVALID_ARG, "scores must be finite");
    }
    uint64_t arena_bytes = tag_offsets[n_tags];
    if (arena_bytes && !arena) return fail(TP_E_INVALID_ARG, "null argument");
    static const uint64_t none = 0;
    return guarded([&]() { return run_host(ctx, arena, arena_bytes, tag_offsets, n_tags + 1, tag_begin, scores, nullptr, n_tags, n_files, out,
                                           n_files ? secondary_begin : &none, secondary_cap); });
}

// Existing-caption triage (Program.cs:203-210): the line-ingest front of the pipeline only, one byte per file back.
static int triage_host(tp_ctx* c, const uint8_t* arena, const uint64_t* offsets, uint64_t n_files, uint8_t* bad) {
    HIP_OK(hipSetDevice(c->device));
    hipStream_t s = nullptr;
    if (n_files == 0) return TP_OK;
    const uint64_t arena_bytes = offsets[n_files];
    c->in_arena.need(arena_bytes + 64);
    c->in_offsets.need((n_files + 1) * 8);
    if (arena_bytes) HIP_OK(hipMemcpyAsync(c->in_arena.p, arena, arena_bytes, hipMemcpyHostToDevice, s));
    HIP_OK(hipMemsetAsync((uint8_t*)c->in_arena.p + arena_bytes, 0, 64, s));   // the lean stage reads whole 16-byte pieces: what lies behind the last line is not left to chance (a byte outside ASCII there sends that file the long way)
    HIP_OK(hipMemcpyAsync(c->in_offsets.p, offsets, (n_files + 1) * 8, hipMemcpyHostToDevice, s));
    c->out_offsets.need((n_files + 1) * 8);
    c->out_verdicts.need(n_files * sizeof(tp_verdict));
    RunIo io{};
    io.arena = c->in_arena.as<uint8_t>(); io.offsets = c->in_offsets.as<uint64_t>();
    io.n_files = n_files;
    io.caption_offsets = c->out_offsets.as<uint64_t>(); io.verdicts = c->out_verdicts.as<tp_verdict>();
    io.stream = s; io.mode = 2;
    int rc = run_device(c, io, nullptr);
    if (rc != TP_OK) return rc;
    std::vector<tp_verdict> v(n_files);
    HIP_OK(hipMemcpy(v.data(), c->out_verdicts.p, n_files * sizeof(tp_verdict), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n_files; ++i) bad[i] = (v[i].flags & TP_F_UNSUPPORTED) ? 2 : (v[i].flags & TP_F_TRIAGE_BAD) ? 1 : 0;
    return TP_OK;
}

int tp_triage_lines(tp_ctx* ctx, const uint8_t* arena, const uint64_t* offsets, uint64_t n_files, uint8_t* bad) {
    if (!ctx || !offsets || (n_files && !bad) || (n_files && offsets[n_files] && !arena)) return fail(TP_E_INVALID_ARG, "null argument");
    for (uint64_t i = 0; i < n_files; ++i) if (offsets[i + 1] < offsets[i]) return fail(TP_E_INVALID_ARG, "offsets must be non-decreasing");
    return guarded([&]() { return triage_host(ctx, arena, offsets, n_files, bad); });
}

int tp_run_scored(tp_ctx* ctx, const uint8_t* arena, const uint64_t* tag_offsets, const uint64_t* tag_begin, const double* scores, const uint8_t* sources,
                  uint64_t n_files, tp_result* out) {
    if (!ctx || !out || !tag_offsets || !tag_begin) return fail(TP_E_INVALID_ARG, "null argument");
    for (uint64_t i = 0; i < n_files; ++i) if (tag_begin[i + 1] < tag_begin[i]) return fail(TP_E_INVALID_ARG, "tag_begin must be non-decreasing");
    uint64_t n_tags = tag_begin[n_files];
    if (n_tags && (!scores || !sources)) return fail(TP_E_INVALID_ARG, "null argument");
    for (uint64_t i = 0; i < n_tags; ++i) {
        if (tag_offsets[i + 1] < tag_offsets[i]) return fail(TP_E_INVALID_ARG, "tag_offsets must be non-decreasing");
        if (!(scores[i] == scores[i]) || scores[i] - scores[i] != 0.0) return fail(TP_E_INVALID_ARG, "scores must be finite");
    }
    if (tag_offsets[n_tags] && !arena) return fail(TP_E_INVALID_ARG, "null argument");
    return guarded([&]() { return run_host(ctx, arena, tag_offsets[n_tags], tag_offsets, n_tags + 1, tag_begin, scores, sources, n_tags, n_files, out); });
}

// tp_run_replies: one reply per file (secondary_begin null: tp_run_scored's checks and run) or two (tp_run_merged's), with the
// replies' HasScore flags handed on.
static int run_scored_checked(tp_ctx* ctx, const uint8_t* arena, const uint64_t* tag_offsets, const uint64_t* tag_begin, const uint64_t* secondary_begin,
                              const double* scores, const uint8_t* sources, int secondary_cap, const uint8_t* reply_flags, uint64_t n_files, tp_result* out) {
    for (uint64_t i = 0; i < n_files; ++i) {
        if (tag_begin[i + 1] < tag_begin[i]) return fail(TP_E_INVALID_ARG, "tag_begin must be non-decreasing");
        if (secondary_begin && (secondary_begin[i] < tag_begin[i] || secondary_begin[i] > tag_begin[i + 1])) return fail(TP_E_INVALID_ARG, "secondary_begin must lie within the file's tags");
    }
    const uint64_t n_tags = tag_begin[n_files];
    if (n_tags && (!scores || (!secondary_begin && !sources))) return fail(TP_E_INVALID_ARG, "null argument");
    for (uint64_t i = 0; i < n_tags; ++i) {
        if (tag_offsets[i + 1] < tag_offsets[i]) return fail(TP_E_INVALID_ARG, "tag_offsets must be non-decreasing");
        if (!(scores[i] == scores[i]) || scores[i] - scores[i] != 0.0) return fail(TP_E_INVALID_ARG, "scores must be finite");
    }
    const uint64_t arena_bytes = tag_offsets[n_tags];
    if (arena_bytes && !arena) return fail(TP_E_INVALID_ARG, "null argument");
    static const uint64_t none = 0;
    return guarded([&]() {
        return run_host(ctx, arena, arena_bytes, tag_offsets, n_tags + 1, tag_begin, scores, secondary_begin ? nullptr : sources, n_tags, n_files, out,
                        secondary_begin ? (n_files ? secondary_begin : &none) : nullptr, secondary_cap, reply_flags);
    });
}

static int run_scored_checked(tp_ctx* ctx, const uint8_t* arena, const uint64_t* tag_offsets, const uint64_t* tag_begin, const uint64_t* secondary_begin,
                              const double* scores, const uint8_t* sources, int secondary_cap, const uint8_t* reply_flags, uint64_t n_files, tp_result* out);

int tp_run_replies(tp_ctx* ctx, const tp_replies* r, int secondary_cap, tp_result* out) {
    if (!ctx || !r || !out || !r->tag_offsets || !r->tag_begin || (r->n_files && !r->has_score)) return fail(TP_E_INVALID_ARG, "null argument");
    return run_scored_checked(ctx, r->arena, r->tag_offsets, r->tag_begin, r->secondary_begin, r->scores, r->sources, secondary_cap, r->has_score, r->n_files, out);
}

namespace {
struct HostReplies {
    DecodedReplies d;
    std::vector<uint64_t> tag_begin, secondary_begin;
    std::vector<uint8_t> sources, status, has_score;
};
}  // namespace

int tp_decode_joytag_replies(tp_ctx* ctx, const uint8_t* primary, const uint64_t* primary_offsets, const uint8_t* secondary,
                             const uint64_t* secondary_offsets, uint64_t n_files, tp_replies* out) {
    if (!ctx || !out || !primary_offsets || (secondary_offsets && !secondary && n_files && secondary_offsets[n_files]) ||
        (n_files && primary_offsets[n_files] && !primary))
        return fail(TP_E_INVALID_ARG, "null argument");
    for (uint64_t i = 0; i < n_files; ++i)
        if (primary_offsets[i + 1] < primary_offsets[i] || (secondary_offsets && secondary_offsets[i + 1] < secondary_offsets[i]))
            return fail(TP_E_INVALID_ARG, "offsets must be non-decreasing");
    return guarded([&]() {
        if (ctx->string_ids.empty()) ctx->string_ids = index_strings(ctx->ht);
        std::unique_ptr<HostReplies> r(new HostReplies());
        r->tag_begin.assign(1, 0);
        r->status.assign(n_files, 0);
        r->has_score.assign(n_files, 0);
        for (uint64_t i = 0; i < n_files; ++i) {
            uint8_t st = 0, hs = 0;
            decode_joytag_reply(ctx->ht, ctx->string_ids, primary + primary_offsets[i], (size_t)(primary_offsets[i + 1] - primary_offsets[i]), r->d, st, hs);
            r->sources.resize(r->d.scores.size(), (uint8_t)TP_SRC_PRIMARY_JOY);
            r->status[i] = st;
            r->has_score[i] = hs;
            if (secondary_offsets) {
                r->secondary_begin.push_back(r->d.scores.size());
                decode_joytag_reply(ctx->ht, ctx->string_ids, secondary + secondary_offsets[i], (size_t)(secondary_offsets[i + 1] - secondary_offsets[i]), r->d, st, hs);
                r->sources.resize(r->d.scores.size(), (uint8_t)TP_SRC_SECONDARY_JOY);
                r->status[i] = std::max(r->status[i], st);
                r->has_score[i] |= (uint8_t)(hs << 1);
            }
            r->tag_begin.push_back(r->d.scores.size());
        }
        if (r->d.arena.empty()) r->d.arena.push_back(0);          // never a null pointer, also without tags
        if (r->d.scores.empty()) { r->d.scores.push_back(0.0); r->sources.push_back(0); }
        out->n_files = n_files;
        out->n_tags = r->tag_begin.back();
        out->arena = r->d.arena.data();
        out->tag_offsets = r->d.tag_offsets.data();
        out->tag_begin = r->tag_begin.data();
        out->secondary_begin = secondary_offsets ? (r->secondary_begin.empty() ? r->tag_begin.data() : r->secondary_begin.data()) : nullptr;
        out->scores = r->d.scores.data();
        out->sources = r->sources.data();
        out->status = r->status.data();
        out->has_score = r->has_score.data();
        out->opaque = r.release();
        return TP_OK;
    });
}

void tp_release_replies(tp_replies* replies) {
    if (replies && replies->opaque) { delete (HostReplies*)replies->opaque; std::memset(replies, 0, sizeof *replies); }
}

void tp_release(tp_ctx* ctx, tp_result* res) {
    (void)ctx;
    if (res && res->opaque) { delete (HostResult*)res->opaque; std::memset(res, 0, sizeof *res); }
}

int tp_run_lines_device(tp_ctx* ctx, const tp_device_io* io, uint64_t* caption_bytes) {
    if (!ctx || !io || !io->offsets || !io->caption_offsets || (io->n_files && (!io->verdicts || !io->captions))) return fail(TP_E_INVALID_ARG, "null argument");
    return guarded([&]() {
        RunIo r{};
        r.arena = io->arena; r.offsets = io->offsets; r.n_files = io->n_files;
        r.captions = io->captions; r.caption_capacity = io->caption_capacity; r.grow_captions = false;
        r.caption_offsets = io->caption_offsets; r.verdicts = io->verdicts; r.histogram = io->histogram; r.counters = io->counters;
        r.fired = io->fired; r.facet_choice = io->facet_choice; r.stream = (hipStream_t)io->stream; r.mode = 0;
        const int rc = run_device(ctx, r, caption_bytes);
        if (rc == TP_OK || rc == TP_E_CAPACITY) { const int lrc = learn_unknown_strings(ctx, r.stream); if (lrc != TP_OK) return lrc; }
        return rc;
    });
}

int tp_run_lines_device_noalloc(tp_ctx* ctx, const tp_device_io* io, uint64_t* caption_bytes) {
    if (!ctx || !io || !io->offsets || !io->caption_offsets || (io->n_files && (!io->verdicts || !io->captions))) return fail(TP_E_INVALID_ARG, "null argument");
    if (io->n_files > TP_NOALLOC_MAX_FILES)
        return fail(TP_E_INVALID_ARG, "tp_run_lines_device_noalloc takes at most " + std::to_string((unsigned long long)TP_NOALLOC_MAX_FILES) + " files a call, got " +
                                      std::to_string((unsigned long long)io->n_files));
    if (!ctx->fixed_bound) {                                       // first call on this context: find this GPU's static array (no allocation)
        const int rc = guarded([&]() { HIP_OK(hipSetDevice(ctx->device)); bind_fixed(ctx); ctx->fixed_bound = true; return (int)TP_OK; });
        if (rc != TP_OK) return rc;
    }
    // the run works in the pieces of the static array instead of the context's own buffers, and within that workspace
    struct Swap {
        tp_ctx* c; WorkBufs* w; size_t budget;
        explicit Swap(tp_ctx* c_) : c(c_), w(c_->w), budget(c_->ws_budget) { c->w = &c->fixed; c->ws_budget = c->fixed.workspace.cap; }
        ~Swap() { c->w = w; c->ws_budget = budget; }
    } swap(ctx);
    return guarded([&]() {
        RunIo r{};
        r.arena = io->arena; r.offsets = io->offsets; r.n_files = io->n_files;
        r.captions = io->captions; r.caption_capacity = io->caption_capacity; r.grow_captions = false;
        r.caption_offsets = io->caption_offsets; r.verdicts = io->verdicts; r.histogram = io->histogram; r.counters = io->counters;
        r.fired = io->fired; r.facet_choice = io->facet_choice; r.stream = (hipStream_t)io->stream; r.mode = 0;
        return run_device(ctx, r, caption_bytes, nullptr, /*timed=*/false);
    });
}

int tp_format_top_tags(const tp_ctx* ctx, const uint64_t* histogram, uint8_t* out, uint64_t capacity, uint64_t* needed) {
    if (!ctx || !histogram || !needed) return fail(TP_E_INVALID_ARG, "null argument");
    return guarded([&]() {
        const std::string text = top_tags_text(ctx->ht, histogram);
        *needed = text.size();
        if (!out || capacity < text.size()) return text.empty() ? TP_OK : fail(TP_E_CAPACITY, "output buffer too small");
        std::memcpy(out, text.data(), text.size());
        return TP_OK;
    });
}

const char* tp_error_needtag_text(void) { return "error; check later"; }   // Program.cs:353

int tp_fallback_caption(const tp_ctx* ctx, uint8_t* out, uint64_t capacity, uint64_t* needed) {
    if (!ctx || !needed) return fail(TP_E_INVALID_ARG, "null argument");
    return guarded([&]() {
        std::string text;                                          // string.Join(", ", fb) + Environment.NewLine, Program.cs:349-350
        for (uint32_t k = 0; k < ctx->dt.n_fallback; ++k) { if (k) text += ", "; text += ctx->ht.strings[ctx->dt.fallback_ids[k]]; }
        text.append((const char*)ctx->dt.newline, ctx->dt.newline_len);
        *needed = text.size();
        if (!out || capacity < text.size()) return fail(TP_E_CAPACITY, "output buffer too small");
        std::memcpy(out, text.data(), text.size());
        return (int)TP_OK;
    });
}

int tp_format_fired_rules(const tp_ctx* ctx, const uint32_t* fired, const uint16_t* facet_choice, uint64_t n_files, uint8_t* out, uint64_t capacity,
                          uint64_t* out_offsets, uint64_t* needed) {
    if (!ctx || !out_offsets || !needed) return fail(TP_E_INVALID_ARG, "null argument");
    const HostTables& h = ctx->ht;
    const uint32_t n_rules = ctx->dt.n_rules, n_facets = ctx->dt.n_facets, fw = ctx->dt.fired_words;
    if (n_files && ((fw && !fired) || (n_facets && !facet_choice))) return fail(TP_E_INVALID_ARG, "null argument");
    return guarded([&]() {
        std::string text;
        // two passes over the same generator: sizes first, then the bytes
        for (int pass = 0; pass < 2; ++pass) {
            uint64_t pos = 0;
            for (uint64_t i = 0; i < n_files; ++i) {
                if (pass == 0) out_offsets[i] = pos;
                bool first = true;
                auto put = [&](const std::string& s) {
                    uint64_t n = s.size() + (first ? 0 : 1);
                    if (pass == 1) { if (!first) out[pos] = ';'; std::memcpy(out + pos + (first ? 0 : 1), s.data(), s.size()); }
                    pos += n;
                    first = false;
                };
                const uint32_t* fb = fired + i * fw;
                auto bit = [&](uint32_t b) { return (fb[b >> 5] >> (b & 31)) & 1u; };
                for (uint32_t r = 0; r < n_rules; ++r) {
                    if (!bit(r)) continue;
                    put("action:" + h.rule_names[r]);
                    for (uint32_t s = h.rules[r].sub_off; s < h.rules[r].sub_off + h.rules[r].sub_cnt; ++s)
                        if (bit(n_rules + s)) put("subtype:" + h.sub_names[s]);
                }
                for (uint32_t f = 0; f < n_facets; ++f) {
                    uint16_t c = facet_choice[i * n_facets + f];
                    const std::string& selected = c == 0xFFFF ? h.facet_defaults[f] : h.facet_rule_tags[f].at(c);
                    if (c != 0xFFFF) put("facet:" + h.facet_names[f] + ":" + selected);
                    if (selected == h.facet_defaults[f]) put("facet:" + h.facet_names[f] + ":" + h.facet_defaults[f] + ":default");
                }
            }
            if (pass == 0) {
                out_offsets[n_files] = pos;
                *needed = pos;
                if (pos == 0) return (int)TP_OK;
                if (!out || capacity < pos) return fail(TP_E_CAPACITY, "output buffer too small");
            }
        }
        return (int)TP_OK;
    });
}

const char* tp_needtag_text(void) { return "low tags; short caption or missing core; check later"; }   // Program.cs:323

int tp_format_side_files(const tp_ctx* ctx, const tp_verdict* verdicts, uint64_t n_files, int keep_needtag, uint8_t* needtag, uint8_t* reject,
                         uint64_t capacity, uint64_t* reject_offsets, uint64_t* needed) {
    if (!ctx || !reject_offsets || !needed || (n_files && (!verdicts || !needtag))) return fail(TP_E_INVALID_ARG, "null argument");
    const HostTables& h = ctx->ht;
    const std::string nl((const char*)ctx->dt.newline, ctx->dt.newline_len);
    return guarded([&]() {
        uint64_t pos = 0;
        for (uint64_t i = 0; i < n_files; ++i) {
            const tp_verdict& v = verdicts[i];
            reject_offsets[i] = pos;
            if (v.flags & TP_F_UNSUPPORTED) { needtag[i] = TP_NEEDTAG_LEAVE; continue; }
            needtag[i] = (v.flags & TP_F_LOW) ? (keep_needtag ? TP_NEEDTAG_WRITE : TP_NEEDTAG_LEAVE) : TP_NEEDTAG_DELETE;   // Program.cs:320-326
            if (v.flags & TP_F_SAFE_REJECT) {
                if (v.safe_hit < 0 || (size_t)v.safe_hit >= h.safe_tags.size()) return fail(TP_E_INVALID_ARG, "verdict with a safe-mode hit out of range");
                pos += 10 + h.safe_tags[v.safe_hit].size() + nl.size();      // "safe_mode:" + tag + newline, Program.cs:944, 955
            }
        }
        reject_offsets[n_files] = pos;
        *needed = pos;
        if (pos == 0) return (int)TP_OK;
        if (!reject || capacity < pos) return fail(TP_E_CAPACITY, "output buffer too small");
        for (uint64_t i = 0; i < n_files; ++i) {
            if (reject_offsets[i + 1] == reject_offsets[i]) continue;
            std::string s = "safe_mode:" + h.safe_tags[verdicts[i].safe_hit] + nl;
            std::memcpy(reject + reject_offsets[i], s.data(), s.size());
        }
        return (int)TP_OK;
    });
}

int tp_set_audit(tp_ctx* ctx, int on) {
    if (!ctx) return fail(TP_E_INVALID_ARG, "null argument");
    ctx->audit = on != 0;
    return TP_OK;
}

int tp_get_audit(const tp_result* res, tp_audit* out) {
    if (!res || !out) return fail(TP_E_INVALID_ARG, "null argument");
    const HostResult* r = (const HostResult*)res->opaque;
    if (!r || r->audit_offsets.empty()) return fail(TP_E_INVALID_ARG, "the result was not produced with tp_set_audit(ctx, 1)");
    out->text = r->audit_text.data();
    out->offsets = r->audit_offsets.data();
    return TP_OK;
}

const char* tp_audit_header(void) {   // Program.cs:964
    return "filename,final_tags,evidence_tags,fired_rules,dropped_tags_with_reason,reject_reason,used_vocab_file,used_profile,coverage_stop_reason,coverage_trace";
}

int tp_format_audit_rows(const tp_ctx* ctx, const tp_result* res, const uint8_t* names, const uint64_t* name_offsets, const char* vocab_file,
                         const char* used_profile, uint8_t* out, uint64_t capacity, uint64_t* out_offsets, uint64_t* needed) {
    if (!ctx || !res || !out_offsets || !needed || !name_offsets || !used_profile) return fail(TP_E_INVALID_ARG, "null argument");
    const HostResult* r = (const HostResult*)res->opaque;
    if (!r || r->audit_offsets.empty()) return fail(TP_E_INVALID_ARG, "the result was not produced with tp_set_audit(ctx, 1)");
    const uint64_t n_files = res->n_files;
    if (n_files && name_offsets[n_files] && !names) return fail(TP_E_INVALID_ARG, "null argument");
    for (uint64_t i = 0; i < n_files; ++i) if (name_offsets[i + 1] < name_offsets[i]) return fail(TP_E_INVALID_ARG, "name_offsets must be non-decreasing");
    const HostTables& h = ctx->ht;
    const uint32_t n_rules = ctx->dt.n_rules, n_facets = ctx->dt.n_facets, fw = ctx->dt.fired_words;
    const std::string nl((const char*)ctx->dt.newline, ctx->dt.newline_len);
    return guarded([&]() {
        static const char* const kCats[6] = {"identity", "hair", "eyes", "outfit", "pose_action", "background_scene"};   // Program.cs:1067-1074
        auto row_for = [&](uint64_t i, std::string& row) -> bool {
            auto csv = [&](const char* p, size_t n) {            // Csv(), Program.cs:976-981: quotes doubled, always quoted
                row.push_back('"');
                for (size_t k = 0; k < n; ++k) { if (p[k] == '"') row.push_back('"'); row.push_back(p[k]); }
                row.push_back('"');
            };
            auto csv_s = [&](const std::string& x) { csv(x.data(), x.size()); };
            std::string field;
            const tp_verdict& v = res->verdicts[i];
            if (v.flags & TP_F_UNSUPPORTED) return true;          // the device code did not process the file: no row
            csv((const char*)names + name_offsets[i], (size_t)(name_offsets[i + 1] - name_offsets[i]));
            const uint64_t* ao = r->audit_offsets.data() + i * AUDIT_COLS;
            row.push_back(','); csv((const char*)r->audit_text.data() + ao[AUDIT_KEPT], (size_t)(ao[AUDIT_KEPT + 1] - ao[AUDIT_KEPT]));
            row.push_back(','); csv((const char*)r->audit_text.data() + ao[AUDIT_EVIDENCE], (size_t)(ao[AUDIT_EVIDENCE + 1] - ao[AUDIT_EVIDENCE]));
            // firedRules, Program.cs:741, 748, 764, 772
            auto put = [&](const std::string& x) { if (!field.empty()) field.push_back(';'); field += x; };
            const uint32_t* fb = r->fired.data() + i * fw;
            auto bit = [&](uint32_t b) { return (fb[b >> 5] >> (b & 31)) & 1u; };
            for (uint32_t q = 0; q < n_rules; ++q) {
                if (!bit(q)) continue;
                put("action:" + h.rule_names[q]);
                for (uint32_t sb = h.rules[q].sub_off; sb < h.rules[q].sub_off + h.rules[q].sub_cnt; ++sb)
                    if (bit(n_rules + sb)) put("subtype:" + h.sub_names[sb]);
            }
            for (uint32_t f = 0; f < n_facets; ++f) {
                uint16_t c = r->facet_choice[i * n_facets + f];
                const std::string& selected = c == 0xFFFF ? h.facet_defaults[f] : h.facet_rule_tags[f].at(c);
                if (c != 0xFFFF) put("facet:" + h.facet_names[f] + ":" + selected);
                if (selected == h.facet_defaults[f]) put("facet:" + h.facet_names[f] + ":" + h.facet_defaults[f] + ":default");
            }
            row.push_back(','); csv_s(field);
            row.push_back(','); csv((const char*)r->audit_text.data() + ao[AUDIT_DROPPED], (size_t)(ao[AUDIT_DROPPED + 1] - ao[AUDIT_DROPPED]));
            field.clear();                                        // rejectReason ?? "", Program.cs:329, 944
            if (v.flags & TP_F_SAFE_REJECT) {
                if (v.safe_hit < 0 || (size_t)v.safe_hit >= h.safe_tags.size()) { fail(TP_E_INVALID_ARG, "verdict with a safe-mode hit out of range"); return false; }
                field = "safe_mode:" + h.safe_tags[v.safe_hit];
            }
            row.push_back(','); csv_s(field);
            row.push_back(','); csv_s(vocab_file ? std::string(vocab_file) : std::string());
            row.push_back(','); csv_s(std::string(used_profile));
            // coverageStopReason and the one coverageTrace entry of a run without the OpenAI passes (Program.cs:226-228, 246-248, 292-294)
            std::string missing, required, details;
            for (int k = 0; k < 6; ++k) {
                const bool ok = (v.coverage >> k) & 1u;
                if (!ok) {
                    if (!missing.empty()) missing.push_back('|');
                    missing += kCats[k];
                    if (k < 5) { if (!required.empty()) required.push_back('|'); required += kCats[k]; }
                }
                if (k) details.push_back('|');
                details += kCats[k]; details += ok ? "=ok" : "=missing";
            }
            row.push_back(','); csv_s(missing.empty() ? std::string("coverage_satisfied") : std::string("openai_disabled_or_missing_api_key"));
            row.push_back(','); csv_s("pass0_joy missing=" + missing + " required_missing=" + required + " details=" + details + " tags=" + std::to_string(v.n_selected));
            return true;
        };
        return format_rows_parallel(n_files, nl, row_for, out, capacity, out_offsets, needed);
    });
}

const char* tp_review_csv_header(void) {   // Program.cs:886
    return "image_path,caption_path,reasons,banned_tokens_detected,dropped_tags,final_tag_count,short_caption,thresholds_used,top_joy_tags,unknown_final_tags,"
           "known_dict_found,known_dict_count,known_dict_path,coverage_stop_reason,coverage_trace";
}

namespace {
// A string as JsonSerializer.Serialize writes it with its default encoder (Program.cs:874): of Basic Latin everything but
// the control characters, DEL and " & ' + < > \ ` stays; backslash and the five short control escapes are written short;
// the rest, everything beyond Basic Latin included, as \uXXXX in upper-case hex, UTF-16 code unit by code unit.
void json_string(const std::string& s, std::string& o) {
    static const char* const hex = "0123456789ABCDEF";
    auto unit = [&](uint32_t u) { o += "\\u"; o += hex[(u >> 12) & 15]; o += hex[(u >> 8) & 15]; o += hex[(u >> 4) & 15]; o += hex[u & 15]; };
    o += '"';
    for (size_t i = 0; i < s.size();) {
        const uint8_t c = (uint8_t)s[i];
        if (c < 0x80) {
            ++i;
            switch (c) {
                case '\\': o += "\\\\"; break;
                case '\b': o += "\\b"; break;
                case '\f': o += "\\f"; break;
                case '\n': o += "\\n"; break;
                case '\r': o += "\\r"; break;
                case '\t': o += "\\t"; break;
                case '"': case '&': case '\'': case '+': case '<': case '>': case '`': case 0x7F: unit(c); break;
                default: if (c < 0x20) unit(c); else o += (char)c;
            }
            continue;
        }
        uint32_t cp = 0xFFFD;
        size_t len = 1;
        if ((c & 0xE0) == 0xC0 && i + 1 < s.size()) { cp = ((c & 0x1Fu) << 6) | ((uint8_t)s[i + 1] & 0x3Fu); len = 2; }
        else if ((c & 0xF0) == 0xE0 && i + 2 < s.size()) { cp = ((c & 0x0Fu) << 12) | (((uint8_t)s[i + 1] & 0x3Fu) << 6) | ((uint8_t)s[i + 2] & 0x3Fu); len = 3; }
        else if ((c & 0xF8) == 0xF0 && i + 3 < s.size()) {
            cp = ((c & 0x07u) << 18) | (((uint8_t)s[i + 1] & 0x3Fu) << 12) | (((uint8_t)s[i + 2] & 0x3Fu) << 6) | ((uint8_t)s[i + 3] & 0x3Fu);
            len = 4;
        }
        i += len;
        if (cp >= 0x10000) { cp -= 0x10000; unit(0xD800 + (cp >> 10)); unit(0xDC00 + (cp & 0x3FF)); }
        else unit(cp);
    }
    o += '"';
}
void json_list(const std::vector<std::string>& v, std::string& o) {
    o += '[';
    for (size_t k = 0; k < v.size(); ++k) { if (k) o += ','; json_string(v[k], o); }
    o += ']';
}
std::vector<std::string> split_lines(const uint8_t* p, uint64_t n) {
    std::vector<std::string> v;
    if (n == 0) return v;
    uint64_t a = 0;
    for (uint64_t k = 0; k <= n; ++k) if (k == n || p[k] == '\n') { v.emplace_back((const char*)p + a, (size_t)(k - a)); a = k + 1; }
    return v;
}
}  // namespace

int tp_format_review_rows(const tp_ctx* ctx, const tp_result* res, int format, const uint8_t* image_paths, const uint64_t* image_offsets,
                          const uint8_t* caption_paths, const uint64_t* caption_offsets, const char* thresholds_used, const char* vocab_file,
                          uint8_t* out, uint64_t capacity, uint64_t* out_offsets, uint64_t* needed) {
    if (!ctx || !res || !out_offsets || !needed || !image_offsets || !caption_offsets || !thresholds_used) return fail(TP_E_INVALID_ARG, "null argument");
    if (format != TP_REVIEW_JSONL && format != TP_REVIEW_CSV) return fail(TP_E_INVALID_ARG, "unknown review format");
    const HostResult* r = (const HostResult*)res->opaque;
    if (!r || r->audit_offsets.empty()) return fail(TP_E_INVALID_ARG, "the result was not produced with tp_set_audit(ctx, 1)");
    const uint64_t n_files = res->n_files;
    if (n_files && ((image_offsets[n_files] && !image_paths) || (caption_offsets[n_files] && !caption_paths))) return fail(TP_E_INVALID_ARG, "null argument");
    for (uint64_t i = 0; i < n_files; ++i)
        if (image_offsets[i + 1] < image_offsets[i] || caption_offsets[i + 1] < caption_offsets[i]) return fail(TP_E_INVALID_ARG, "path offsets must be non-decreasing");
    const HostTables& h = ctx->ht;
    const std::string nl((const char*)ctx->dt.newline, ctx->dt.newline_len);
    return guarded([&]() {
        static const char* const kCats[6] = {"identity", "hair", "eyes", "outfit", "pose_action", "background_scene"};
        auto join = [](const std::vector<std::string>& v) { std::string o; for (size_t k = 0; k < v.size(); ++k) { if (k) o += ';'; o += v[k]; } return o; };
        auto row_for = [&](uint64_t i, std::string& row) -> bool {
            auto csv = [&](const std::string& x) { row += '"'; for (char ch : x) { if (ch == '"') row += '"'; row += ch; } row += '"'; };   // Program.cs:976-981
            const tp_verdict& v = res->verdicts[i];
            if (v.flags & TP_F_UNSUPPORTED) return true;
            const uint32_t rb = r->review_bits[i];
            // BuildReviewRecord, Program.cs:826-868
            std::vector<std::string> banned, reasons;
            for (int k = 0; k < N_BANNED; ++k) if (rb & (1u << k)) banned.push_back(h.strings[ctx->dt.banned_ids[k]]);
            const bool has_male = !banned.empty() || (rb & REVIEW_MALE), has_girl = (rb & REVIEW_GIRL) != 0, multi = (rb & REVIEW_MULTI) != 0;
            const bool low = (v.flags & TP_F_LOW) != 0;
            if (has_male) reasons.push_back("male_subject_token");
            if (has_male && has_girl) reasons.push_back("mixed_subject_tokens");
            if (multi) reasons.push_back("multi_person");
            if (low) reasons.push_back("short_caption");
            const uint64_t* ao = r->audit_offsets.data() + i * AUDIT_COLS;
            std::vector<std::string> top = split_lines(r->audit_text.data() + ao[AUDIT_TOP], ao[AUDIT_TOP + 1] - ao[AUDIT_TOP]);
            std::vector<std::string> unknown = split_lines(r->audit_text.data() + ao[AUDIT_UNKNOWN], ao[AUDIT_UNKNOWN + 1] - ao[AUDIT_UNKNOWN]);
            std::stable_sort(unknown.begin(), unknown.end(), less_ordinal_ignore_case);     // OrderBy(OrdinalIgnoreCase), Program.cs:822
            std::string missing, required, details;
            for (int k = 0; k < 6; ++k) {
                const bool ok = (v.coverage >> k) & 1u;
                if (!ok) {
                    if (!missing.empty()) missing.push_back('|');
                    missing += kCats[k];
                    if (k < 5) { if (!required.empty()) required.push_back('|'); required += kCats[k]; }
                }
                if (k) details.push_back('|');
                details += kCats[k]; details += ok ? "=ok" : "=missing";
            }
            const std::string stop = missing.empty() ? "coverage_satisfied" : "openai_disabled_or_missing_api_key";
            const bool has_coverage_log = !missing.empty();                                  // Program.cs:844 with a one-entry trace
            if (reasons.empty() && banned.empty() && unknown.empty() && !has_coverage_log) return true;   // Program.cs:845: no record
            const std::vector<std::string> trace = {"pass0_joy missing=" + missing + " required_missing=" + required + " details=" + details + " tags=" +
                                                    std::to_string(v.n_selected)};
            const std::string image((const char*)image_paths + image_offsets[i], (size_t)(image_offsets[i + 1] - image_offsets[i]));
            const std::string caption((const char*)caption_paths + caption_offsets[i], (size_t)(caption_offsets[i + 1] - caption_offsets[i]));
            const std::string vocab = vocab_file ? vocab_file : "";
            if (format == TP_REVIEW_JSONL) {                                                 // JsonSerializer.Serialize(record), Program.cs:874, 415-431
                row += "{\"ImagePath\":"; json_string(image, row);
                row += ",\"CaptionPath\":"; json_string(caption, row);
                row += ",\"Reasons\":"; json_list(reasons, row);
                row += ",\"BannedTokensDetected\":"; json_list(banned, row);
                row += ",\"DroppedTags\":"; json_list(banned, row);                       // DroppedMaleSubjectTags: the same list (Program.cs:618-623, 856)
                row += ",\"FinalTagCount\":" + std::to_string(v.n_selected);
                row += std::string(",\"ShortCaption\":") + (low ? "true" : "false");
                row += ",\"ThresholdsUsed\":"; json_string(thresholds_used, row);
                row += ",\"TopJoyTags\":"; json_list(top, row);
                row += ",\"UnknownFinalTags\":"; json_list(unknown, row);
                row += std::string(",\"KnownTagDictionaryFound\":") + (h.known_found ? "true" : "false");
                row += ",\"KnownTagDictionaryCount\":" + std::to_string(h.n_known);
                row += ",\"KnownTagDictionaryPath\":"; json_string(vocab, row);
                row += ",\"CoverageTrace\":"; json_list(trace, row);
                row += ",\"CoverageStopReason\":"; json_string(stop, row);
                row += '}';
            } else {                                                                         // Program.cs:875-881
                csv(image); row += ','; csv(caption); row += ','; csv(join(reasons)); row += ','; csv(join(banned)); row += ','; csv(join(banned)); row += ',';
                csv(std::to_string(v.n_selected)); row += ','; csv(low ? "true" : "false"); row += ','; csv(thresholds_used); row += ','; csv(join(top)); row += ',';
                csv(join(unknown)); row += ','; csv(h.known_found ? "true" : "false"); row += ','; csv(std::to_string(h.n_known)); row += ','; csv(vocab); row += ',';
                csv(stop); row += ','; csv(join(trace));
            }
            return true;
        };
        return format_rows_parallel(n_files, nl, row_for, out, capacity, out_offsets, needed);
    });
}

int tp_get_last_timing(const tp_ctx* ctx, tp_timing* out) {
    if (!ctx || !out) return fail(TP_E_INVALID_ARG, "null argument");
    *out = ctx->timing;
    return TP_OK;
}

}  // extern "C"
