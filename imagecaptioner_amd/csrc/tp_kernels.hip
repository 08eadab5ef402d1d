// HIP kernels of the batch tag pipeline for gfx950 (MI355X) and their launchers.
//
// Kernel map (all integer/byte work, HBM- and latency-bound; no MFMA):
//   k_fast         one WAVEFRONT per file (tp_fast.h): line staged in LDS, lanes = bytes / tokens / candidates / tags,
//                  set operations by ballot and an LDS hash set.  Takes the common shape of a file; files it declines
//                  are appended to a list for k_pipeline.
//   k_pipeline     one THREAD per file (tp_pipeline.h): the general form of the same pipeline; lane-private arrays are
//                  interleaved across the wavefront's 64 lanes so equal element indices share 512-byte rows.
//   k_wave_sizes   sizes k_pipeline's workspace: per item the token upper bound (comma count, 8 bytes per step) and the
//                  line length, reduced to per-wavefront maxima.
//   k_scan_*       three-pass exclusive scan (u32/u64 -> u64) for workspace offsets and caption offsets.
//   k_emit         one wavefront per file: caption bytes from the final tag list (wave prefix sum of tag lengths),
//                  tag histogram (block-private LDS counters for the low, frequent ids; global atomics for the tail),
//                  batch counters.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "tp_fast.h"
#include "tp_lean.h"
#include "tp_launch.h"
#include "tp_pipeline.h"

namespace tp {

// ------------------------------------------------------------------------------------------------
// k_fast
// ------------------------------------------------------------------------------------------------
// Wave collectives for tp_fast.h.  LDS traffic between lanes of one wavefront needs no barrier in hardware (the lanes
// run in lock step and LDS operations of a wave complete in order); the fence keeps the compiler from moving LDS
// accesses across the exchange point.
#ifdef TP_PROF
__device__ unsigned long long g_prof[32];   // cycles per pipeline section, summed over wavefronts (diagnostic builds only)
#endif

// Inclusive prefix sum over the 64 lanes with DPP row shifts and row broadcasts (no LDS traffic).
__device__ inline uint32_t wave_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// The OR over the 64 lanes, the same way; the result is in lane 63.
__device__ inline uint32_t wave_or_to_last(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

// The number of this wavefront in its block and in the grid.  (Handing these to the compiler as wave-uniform values - through
// readfirstlane - turns the per-file loads of sizes and offsets into scalar loads; measured, that is no gain in the lean
// kernels and a loss of 7 % in k_emit: scalar loads return out of order and share a counter with LDS, so every wait
// for one of them also waits for all LDS traffic.)
__device__ __forceinline__ uint32_t wave_in_block() { return threadIdx.x >> 6; }
__device__ __forceinline__ uint64_t wave_index() { return ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; }

struct DevWave {
#ifdef TP_PROF
    uint64_t last = 0, acc[32] = {0};
    __device__ __forceinline__ void tick(int k) {
        uint64_t now = __builtin_readcyclecounter();
        if (k > 0) acc[k] += now - last;      // section 0 starts the file's clock (the lean stage uses 16 .. 24; 16 is the time between its files)
        last = now;
    }
    __device__ void flush() {
        if (lane() == 0) for (int k = 0; k < 32; ++k) if (acc[k]) atomicAdd(&g_prof[k], (unsigned long long)acc[k]);
    }
#else
    __device__ void tick(int) const {}
#endif
    __device__ uint32_t lane() const { return threadIdx.x & 63u; }
    __device__ uint64_t ballot(bool p) const { return __ballot(p ? 1 : 0); }
    // how many set bits of m belong to lanes below this one
    __device__ uint32_t rank(uint64_t m) const { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
    // the value lane `src` holds; src is the same in every lane
    __device__ uint32_t bcast(uint32_t v, uint32_t src) const { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)src)); }
    // inclusive prefix sum over the lanes
    __device__ uint32_t scan_incl(uint32_t v) const { return wave_scan_incl(v); }
    // the value of the lane below; lane 0 receives `fill`
    __device__ uint32_t shift_up(uint32_t v, uint32_t fill) const { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false); }   // wave_shr:1
    // the sum over all lanes, in every lane
    __device__ uint32_t sum(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(v), 63); }
    // the OR over all lanes, in every lane
    __device__ uint32_t or_all(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readlane((int)wave_or_to_last(v), 63); }
    __device__ uint32_t shfl(uint32_t v, uint32_t src) const { return (uint32_t)__shfl((int)v, (int)(src & 63u), 64); }
    // Orders the wave's own LDS traffic (each wavefront owns its LDS state): LDS instructions of one wave execute in
    // order, so wavefront scope is enough and, unlike a wider scope, does not wait for outstanding global loads and stores.
    __device__ void sync() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ uint32_t lds_cas(uint32_t* p, uint32_t cmp, uint32_t val) const { return atomicCAS(p, cmp, val); }
    __device__ void lds_min(uint32_t* p, uint32_t v) const { atomicMin(p, v); }
    __device__ void lds_or(uint32_t* p, uint32_t v) const { atomicOr(p, v); }
    __device__ void lds_min64(uint64_t* p, uint64_t v) const { atomicMin((unsigned long long*)p, (unsigned long long)v); }
    __device__ void copy16(uint8_t* lds_dst, const uint8_t* src) const { *(uint4*)lds_dst = *(const uint4*)src; }
    __device__ void load16(const uint8_t* src, uint32_t (&x)[4]) const { const uint4 v = *(const uint4*)src; x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
    __device__ void store16(uint8_t* lds_dst, const uint32_t (&x)[4]) const { *(uint4*)lds_dst = make_uint4(x[0], x[1], x[2], x[3]); }
    __device__ void fill16(uint32_t* lds_dst, uint32_t v) const { *(uint4*)lds_dst = make_uint4(v, v, v, v); }
};

// The commas of a text, counted by the whole wavefront eight bytes per lane and step (aligned words; what they hold beyond the
// text's ends is masked out).
__device__ inline uint32_t wave_count_commas(const uint8_t* p, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63u, shift = (uint32_t)((uintptr_t)p & 7);
    const uint64_t* base = (const uint64_t*)(p - shift);
    const uint32_t words = (shift + n + 7) >> 3;
    uint32_t c = 0;
    for (uint32_t i = lane; i < words; i += 64) {
        uint64_t x = base[i] ^ 0x2C2C2C2C2C2C2C2Cull;               // a zero byte where the text holds ','
        uint64_t z = (x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full;
        z = ~(z | x | 0x7F7F7F7F7F7F7F7Full);                        // bit 7 of every zero byte
        const uint32_t lo = i == 0 ? shift : 0u, hi = (i + 1) * 8 > shift + n ? shift + n - i * 8 : 8u;   // bytes [lo, hi) of the word are text
        if (lo) z &= ~0ull << (8 * lo);
        if (hi < 8) z &= ~(~0ull << (8 * hi));
        c += (uint32_t)__popcll(z);
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(c), 63);
}

constexpr int FAST_WAVES_PER_BLOCK = 4;

// The files a wavefront declines, handed on in batches: one atomic on the list's counter per DECLINE_BATCH files instead
// of one per file (a single counter takes about 90 atomics per microsecond, which a stage that declines a quarter of a
// million files would spend milliseconds on).
constexpr uint32_t DECLINE_BATCH = 32;
struct DeclineBuf { uint32_t item[DECLINE_BATCH]; };
// from_end: the list is filled from entry from_end - 1 downwards (0: from entry 0 upwards)
__device__ inline void decline_flush(DeclineBuf& b, uint32_t& n, uint32_t* list, uint32_t* count, uint32_t from_end = 0) {   // wave-uniform n
    if (n == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(count, n);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (lane < n) list[from_end ? from_end - 1 - (base + lane) : base + lane] = b.item[lane];
    n = 0;
}
__device__ inline void decline_push(DeclineBuf& b, uint32_t& n, uint32_t item, uint32_t* list, uint32_t* count, uint32_t from_end = 0) {
    if ((threadIdx.x & 63u) == 0) b.item[n] = item;
    ++n;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (n == DECLINE_BATCH) decline_flush(b, n, list, count, from_end);
}

// The first stage serves files of at most 64 tokens (and 128 final tags) over the whole range [file0, file_end); the second
// serves the files the first declined (a.item_list, a.item_count entries on the device) with 128 tokens.  Files both decline
// go to a.slow_list for the general kernel.
// TR: existing-caption triage (a compile-time choice, so that the caption kernels carry none of it).
// AUD: audit mode - the file's other lists go to its row of a.aud32 (LeanAudit in tp_fast.h), as the lean stage leaves them.
template <int NT, bool TR, bool AUD = false>
__device__ __forceinline__ void fast_body(const DevTables& t, const FastArgs& a) {
    __shared__ FastLds lds[FAST_WAVES_PER_BLOCK];
    __shared__ uint8_t cls[256];                                // byte classes of the line scan
    extern __shared__ uint32_t rc_lds[];                        // the rule cache, t.rc_words words (dynamic LDS)
    // the list's entries from its front, then those from its end (the lean stage's two halves list their files at either end)
    const uint32_t n_front = a.item_list && a.item_count ? *a.item_count : 0u, n_back = a.item_list && a.item_count_b ? *a.item_count_b : 0u;
    const uint64_t n_items = a.item_list ? (uint64_t)n_front + n_back : a.file_end - a.file0;
    if (a.item_list && (uint64_t)blockIdx.x * FAST_WAVES_PER_BLOCK >= n_items) return;   // a later stage with nothing left for this block
    auto item = [&](uint64_t i) { return a.item_list[i < n_front ? i : a.item_cap - 1 - (i - n_front)]; };
    __shared__ DeclineBuf dbuf[FAST_WAVES_PER_BLOCK];
    uint32_t n_declined = 0;
    DevWave w;
    const uint64_t wave_global = wave_index();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    // A stage that works through an earlier stage's list: most of what reaches it and is not its to take has too many tokens,
    // which a count of the commas shows for a fraction of the line scan.  A block none of whose files fits hands them on before
    // it has set anything up (the rule cache is 18 KB with a few hundred rules, and such blocks can be most of the grid).
    auto fits = [&](uint64_t f) {
        const uint64_t o = a.offsets[f], len64 = a.offsets[f + 1] - o;
        return len64 <= FAST_LINE_MAX && !(a.item_list && wave_count_commas(a.arena + o, (uint32_t)len64) >= 64u * NT);
    };
    if (a.item_list) {
        bool any = false;
        for (uint64_t i = wave_global; i < n_items && !any; i += n_waves) any = fits(a.file0 + item(i));
        if (!__syncthreads_or(any ? 1 : 0)) {
            for (uint64_t i = wave_global; i < n_items; i += n_waves) decline_push(dbuf[wave_in_block()], n_declined, item(i), a.slow_list, a.slow_count);
            decline_flush(dbuf[wave_in_block()], n_declined, a.slow_list, a.slow_count);
            return;
        }
    }
    for (uint32_t i = threadIdx.x; i < t.rc_words; i += blockDim.x) rc_lds[i] = t.rc[i];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) cls[i] = byte_class(i);
    __syncthreads();
    FastLds& L = lds[wave_in_block()];
    for (uint64_t i = wave_global; i < n_items; i += n_waves) {
        const uint64_t f = a.file0 + (a.item_list ? item(i) : i);
        uint64_t o = a.offsets[f];
        uint64_t len64 = a.offsets[f + 1] - o;
        bool done = false;
        FastOut out;
        if (fits(f)) {
            LeanAudit aud{nullptr, nullptr, nullptr, nullptr};
            if (AUD) { const uint64_t r = f - a.file0; aud.row = a.aud32 + r * AUD_STRIDE; aud.cnt = a.audit_cnt + r * AUDIT_COLS; aud.len = a.audit_len + r * AUDIT_COLS; aud.rbits = a.review_bits + r; }
            done = fast_file<NT, 2, DevWave, LineFront, AUD>(w, t, L, rc_lds, cls, a.arena + o, (uint32_t)len64, out, a.fin32 + (f - a.file0) * FAST_FIN_MAX,
                                                             a.fired ? a.fired + f * t.fired_words : nullptr, a.facet_choice ? a.facet_choice + f * t.n_facets : nullptr, TR, nullptr, aud);
        }
        if (done) {
            if (w.lane() == 0) {
                a.verdicts[f] = out.verdict;
                a.n_final[f] = out.n_final;
                a.cap_len[f] = out.caption_len;
                a.slot_of[f] = SLOT_FAST;
            }
        } else {
            decline_push(dbuf[wave_in_block()], n_declined, (uint32_t)(f - a.file0), a.slow_list, a.slow_count);
        }
        w.sync();
    }
    decline_flush(dbuf[wave_in_block()], n_declined, a.slow_list, a.slow_count);
#ifdef TP_PROF
    w.flush();
#endif
}
#ifndef TP_FAST_WAVES_PER_SIMD
#define TP_FAST_WAVES_PER_SIMD 5
#endif
__global__ __launch_bounds__(64 * FAST_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(TP_FAST_WAVES_PER_SIMD))) void k_fast(DevTables t, FastArgs a) {
    fast_body<1, false>(t, a);
}
__global__ __launch_bounds__(64 * FAST_WAVES_PER_BLOCK) void k_fast2(DevTables t, FastArgs a) { fast_body<2, false>(t, a); }
__global__ __launch_bounds__(64 * FAST_WAVES_PER_BLOCK) void k_fast_triage(DevTables t, FastArgs a) { fast_body<1, true>(t, a); }
__global__ __launch_bounds__(64 * FAST_WAVES_PER_BLOCK) void k_fast2_triage(DevTables t, FastArgs a) { fast_body<2, true>(t, a); }
__global__ __launch_bounds__(64 * FAST_WAVES_PER_BLOCK) void k_fast_audit(DevTables t, FastArgs a) { fast_body<1, false, true>(t, a); }
__global__ __launch_bounds__(64 * FAST_WAVES_PER_BLOCK) void k_fast2_audit(DevTables t, FastArgs a) { fast_body<2, false, true>(t, a); }

// The lean stage (tp_lean.h) over the whole range, as two kernels: the front half (line scan, token lookup, Distinct) leaves
// each file's candidates - interned ids - in the file's row of the id array and marks the file SLOT_LEAN; the back half takes
// every file so marked from the candidates to the final tag ids.  Files either half declines go to a.slow_list for the
// stages above.  The table image is read from device memory through the constant address space: its scalars come by scalar
// loads at the places that use them instead of occupying registers for the whole kernel.
constexpr int LEAN_WAVES_PER_BLOCK = 4;
#ifndef TP_LEAN_A_WAVES_PER_SIMD
#define TP_LEAN_A_WAVES_PER_SIMD 8
#endif
#ifndef TP_LEAN_B_WAVES_PER_SIMD
#define TP_LEAN_B_WAVES_PER_SIMD 5
#endif
typedef const __attribute__((address_space(4))) DevTables ConstTables;
__global__ __launch_bounds__(64 * LEAN_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(TP_LEAN_A_WAVES_PER_SIMD))) void k_lean_a(FastArgs a) {
    __shared__ LeanFrontLds lds[LEAN_WAVES_PER_BLOCK];
    __shared__ LeanShared shared;
    __shared__ DeclineBuf dbuf[LEAN_WAVES_PER_BLOCK];
    ConstTables& t = *(ConstTables*)(uintptr_t)a.tables;
    lean_shared_init(t, shared, threadIdx.x, blockDim.x);
    if ((threadIdx.x & 63u) == 0) lds[wave_in_block()].cpos[LEAN_CPOS_NOTE] = 0;
    __syncthreads();
    uint32_t n_declined = 0;
    LeanFrontLds& L = lds[wave_in_block()];
    DevWave w;
    const uint64_t wave_global = wave_index();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t n_items = a.file_end - a.file0;
    // Two files ahead the wave reads a file's offsets, one file ahead its piece of the line, so that neither load is
    // waited for when its file's turn comes.
    auto bounds = [&](uint64_t i, uint64_t& o, uint32_t& len) {
        o = 0; len = 0xFFFFFFFFu;
        if (i < n_items) { o = a.offsets[a.file0 + i]; const uint64_t l = a.offsets[a.file0 + i + 1] - o; len = l <= LEAN_LINE_MAX ? (uint32_t)l : 0xFFFFFFFFu; }
    };
    uint64_t o0, o1, o2;
    uint32_t len0, len1, len2;
    uint32_t x0[4], x1[4] = {0, 0, 0, 0};
    bounds(wave_global, o0, len0);
    bounds(wave_global + n_waves, o1, len1);
    x0[0] = x0[1] = x0[2] = x0[3] = 0;
    if (len0 != 0xFFFFFFFFu) lean_fetch(w, a.arena + o0, len0, x0);
    for (uint64_t i = wave_global; i < n_items; i += n_waves) {
        const uint64_t f = a.file0 + i;
        bounds(i + 2 * n_waves, o2, len2);
        if (len1 != 0xFFFFFFFFu) lean_fetch(w, a.arena + o1, len1, x1);
        bool done = false;
        uint32_t n_cand = 0;
        const uint64_t fo_off = o0;
        const uint32_t fo_len = len0;
        if (len0 != 0xFFFFFFFFu) done = lean_front(w, t, L, shared, a.arena + o0, len0, x0, a.fin32 + i * FAST_FIN_MAX, n_cand);
        o0 = o1; len0 = len1; o1 = o2; len1 = len2;
        x0[0] = x1[0]; x0[1] = x1[1]; x0[2] = x1[2]; x0[3] = x1[3];
        if (done) {
            if (w.lane() == 0) { a.n_final[f] = n_cand; a.slot_of[f] = SLOT_LEAN; }
        } else {
            if (t.unk_keys != nullptr && fo_len != 0xFFFFFFFFu && L.cpos[LEAN_CPOS_NOTE] != 0) {   // declined over plain tokens in no table: note them
                lean_note_tokens(w, t, L, a.arena + fo_off, fo_len);
                w.sync();
                if (w.lane() == 0) L.cpos[LEAN_CPOS_NOTE] = 0;
            }
            if (w.lane() == 0) a.slot_of[f] = SLOT_NONE;             // not left to whatever an earlier run wrote there: the back half looks at it
            decline_push(dbuf[wave_in_block()], n_declined, (uint32_t)i, a.slow_list, a.slow_count);
        }
        w.sync();
    }
    decline_flush(dbuf[wave_in_block()], n_declined, a.slow_list, a.slow_count);
#ifdef TP_PROF
    w.flush();
#endif
}
// DETAIL: the run wants the fired rules, the facets' choices or the audit lists (a.fired, a.facet_choice, a.aud32); the plain run's build has no code for either
template <bool DETAIL>
__device__ __forceinline__ void lean_b_body(const FastArgs& a) {
    __shared__ LeanLds lds[LEAN_WAVES_PER_BLOCK];
    __shared__ LeanShared shared;
    __shared__ DeclineBuf dbuf[LEAN_WAVES_PER_BLOCK];
    extern __shared__ uint32_t rc_lds[];                        // the rule cache, t.rc_words words (dynamic LDS)
    ConstTables& t = *(ConstTables*)(uintptr_t)a.tables;
    for (uint32_t i = threadIdx.x; i < t.rc_words; i += blockDim.x) rc_lds[i] = t.rc[i];
    lean_shared_init(t, shared, threadIdx.x, blockDim.x);
    __syncthreads();
    uint32_t n_declined = 0;
    LeanLds& L = lds[wave_in_block()];
    DevWave w;
    const uint64_t wave_global = wave_index();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t n_items = a.file_end - a.file0;
    // Two files ahead the wave reads whether the front half passed a file, its candidate count and ids; one file ahead it
    // has the candidates' records (DevTables::crec, 32 bytes each) loaded straight into L.cr, as soon as the file before
    // has read its own out of there; so neither the ids nor the records are waited for when a file's turn comes.
    auto peek = [&](uint64_t i, uint32_t& n_cand, uint32_t& id) {
        n_cand = 0xFFFFFFFFu; id = NO_ID;
        if (i < n_items && a.slot_of[a.file0 + i] == SLOT_LEAN) {
            n_cand = a.n_final[a.file0 + i];
            id = a.fin32[i * FAST_FIN_MAX + w.lane()];           // entries behind the count are never used
        }
    };
    auto fetch_records = [&](uint32_t n_cand, uint32_t id) {       // n_cand is the same in every lane
        if (n_cand != 0xFFFFFFFFu && w.lane() < n_cand) {
            const uint32_t* src = t.crec + 8 * (size_t)id;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)&L.cr[0][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4), (__attribute__((address_space(3))) void*)&L.cr[1][0], 16, 0, 0);
        }
    };
    uint32_t* const back_count = a.slow_count_b ? a.slow_count_b : a.slow_count;   // its own end of the list, or the front half's
    const uint32_t from_end = a.slow_count_b ? a.item_cap : 0u;
    uint32_t n0, id0, n1, id1, n2, id2;
    peek(wave_global, n0, id0);
    peek(wave_global + n_waves, n1, id1);
    fetch_records(n0, id0);
    for (uint64_t i = wave_global; i < n_items; i += n_waves) {
        const uint64_t f = a.file0 + i;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this file's records are in L.cr (and the next file's ids in registers)
        peek(i + 2 * n_waves, n2, id2);
        bool fetched = false;
        auto records_free = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // what this file read out of L.cr has arrived
            fetch_records(n1, id1);
            fetched = true;
        };
        if (n0 != 0xFFFFFFFFu) {
            FastOut out;
            LeanAudit aud{nullptr, nullptr, nullptr, nullptr};
            if (DETAIL && a.aud32) { aud.row = a.aud32 + i * AUD_STRIDE; aud.cnt = a.audit_cnt + i * AUDIT_COLS; aud.len = a.audit_len + i * AUDIT_COLS; aud.rbits = a.review_bits + i; }
            const bool done = lean_back<DETAIL, true>(w, t, L, shared, rc_lds, w.lane() < n0 ? id0 : NO_ID, n0, out, a.fin32 + i * FAST_FIN_MAX,
                                        a.fired ? a.fired + f * t.fired_words : nullptr, a.facet_choice ? a.facet_choice + f * t.n_facets : nullptr, records_free, aud);
            if (done) {
                if (w.lane() == 0) {
                    a.verdicts[f] = out.verdict;
                    a.n_final[f] = out.n_final;
                    a.cap_len[f] = out.caption_len;
                    a.slot_of[f] = SLOT_FAST;
                }
            } else {
                decline_push(dbuf[wave_in_block()], n_declined, (uint32_t)i, a.slow_list, back_count, from_end);
            }
            w.sync();
        }
        if (!fetched) records_free();
        n0 = n1; id0 = id1; n1 = n2; id1 = id2;
    }
    decline_flush(dbuf[wave_in_block()], n_declined, a.slow_list, back_count, from_end);
#ifdef TP_PROF
    w.flush();
#endif
}
__global__ __launch_bounds__(64 * LEAN_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(TP_LEAN_B_WAVES_PER_SIMD))) void k_lean_b(FastArgs a) { lean_b_body<false>(a); }
#ifndef TP_LEAN_B_DETAIL_WAVES_PER_SIMD
#define TP_LEAN_B_DETAIL_WAVES_PER_SIMD 4                            // the lists want more registers: 5 waves spill about twenty
#endif
__global__ __launch_bounds__(64 * LEAN_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(TP_LEAN_B_DETAIL_WAVES_PER_SIMD))) void k_lean_b_detail(FastArgs a) { lean_b_body<true>(a); }

// Scored candidates (tagger replies as text and score per tag) through the wave kernel's scored front, two candidates per lane.
// AUD: audit mode - the file's other lists go to its row of a.aud32, as the line stages leave them.
template <bool AUD>
__device__ __forceinline__ void scored_body(const DevTables& t, const FastArgs& a) {
    __shared__ FastLds lds[FAST_WAVES_PER_BLOCK];
    __shared__ uint8_t cls[256];
    extern __shared__ uint32_t rc_lds[];
    for (uint32_t i = threadIdx.x; i < t.rc_words; i += blockDim.x) rc_lds[i] = t.rc[i];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) cls[i] = byte_class(i);
    __syncthreads();
    __shared__ DeclineBuf dbuf[FAST_WAVES_PER_BLOCK];
    uint32_t n_declined = 0;
    FastLds& L = lds[wave_in_block()];
    DevWave w;
    const uint64_t wave_global = wave_index();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t n_items = a.file_end - a.file0;
    for (uint64_t i = wave_global; i < n_items; i += n_waves) {
        const uint64_t f = a.file0 + i;
        ScoredFront<DevWave> front;
        const uint64_t lo = a.tag_begin[f], hi = a.tag_begin[f + 1];
        front.in = ScoredIn{a.arena, a.offsets, lo, a.secondary_begin ? a.secondary_begin[f] : hi, hi, a.scores, a.sources, a.secondary_begin != nullptr, a.secondary_cap};
        FastOut out;
        LeanAudit aud{nullptr, nullptr, nullptr, nullptr};
        if (AUD) { const uint64_t r = f - a.file0; aud.row = a.aud32 + r * AUD_STRIDE; aud.cnt = a.audit_cnt + r * AUDIT_COLS; aud.len = a.audit_len + r * AUDIT_COLS; aud.rbits = a.review_bits + r; }
        front.aud = aud;
        // a reply without scores keeps its order (Program.cs:487): its lists follow that order, which the lanes do not hold
        const bool as_it_came = AUD && a.reply_flags && !(a.reply_flags[f] & 1u);
        const bool done = !as_it_came && fast_file<2, 2, DevWave, ScoredFront<DevWave>, AUD>(w, t, L, rc_lds, cls, nullptr, 0, out, a.fin32 + (f - a.file0) * FAST_FIN_MAX,
                                                                              a.fired ? a.fired + f * t.fired_words : nullptr,
                                                                              a.facet_choice ? a.facet_choice + f * t.n_facets : nullptr, false, &front, aud);
        if (done) {
            if (w.lane() == 0) {
                a.verdicts[f] = out.verdict;
                a.n_final[f] = out.n_final;
                a.cap_len[f] = out.caption_len;
                a.slot_of[f] = SLOT_FAST;
            }
        } else {
            decline_push(dbuf[wave_in_block()], n_declined, (uint32_t)(f - a.file0), a.slow_list, a.slow_count);
        }
        w.sync();
    }
    decline_flush(dbuf[wave_in_block()], n_declined, a.slow_list, a.slow_count);
}
__global__ __launch_bounds__(64 * FAST_WAVES_PER_BLOCK) void k_scored(DevTables t, FastArgs a) { scored_body<false>(t, a); }
__global__ __launch_bounds__(64 * FAST_WAVES_PER_BLOCK) void k_scored_audit(DevTables t, FastArgs a) { scored_body<true>(t, a); }

// Third stage: the files the second declined, up to LL::TOK_CAP tokens in LL::LINE_CAP bytes (three sizes, smallest first: what
// one declines for its size the next takes); one wavefront per block.
template <bool TR, class LL, bool AUD = false>
__device__ __forceinline__ void long_body(const DevTables& t, const FastArgs& a) {
    __shared__ LL L;
    __shared__ uint8_t cls[256];
    // The rule cache stays in device memory here: a copy per one-wavefront block (18 KB with a few hundred rules) would decide
    // how many files a CU holds at once, and a long file reads it once for hundreds of tokens.
    const uint32_t* rc_lds = t.rc;
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) cls[i] = byte_class(i);
    __syncthreads();
    DevWave w;
    const uint64_t n_items = (uint64_t)*a.item_count;
    for (uint64_t i = blockIdx.x; i < n_items; i += gridDim.x) {
        const uint64_t f = a.file0 + a.item_list[i];
        uint64_t o = a.offsets[f];
        uint64_t len64 = a.offsets[f + 1] - o;
        bool done = false;
        FastOut out;
        if (len64 <= LL::LINE_CAP && (LL::TOK_CAP == LONG_TOK_MAX || wave_count_commas(a.arena + o, (uint32_t)len64) < LL::TOK_CAP)) {
            LeanAudit aud{nullptr, nullptr, nullptr, nullptr};
            if (AUD) { const uint64_t r = f - a.file0; aud.row = a.aud32 + r * AUD_STRIDE; aud.cnt = a.audit_cnt + r * AUDIT_COLS; aud.len = a.audit_len + r * AUDIT_COLS; aud.rbits = a.review_bits + r; }
            done = long_file<AUD>(w, t, L, rc_lds, cls, a.arena + o, (uint32_t)len64, out, a.fin32 + (f - a.file0) * FAST_FIN_MAX,
                                  a.fired ? a.fired + f * t.fired_words : nullptr, a.facet_choice ? a.facet_choice + f * t.n_facets : nullptr, TR, aud);
        }
        if (w.lane() == 0) {
            if (done) {
                a.verdicts[f] = out.verdict;
                a.n_final[f] = out.n_final;
                a.cap_len[f] = out.caption_len;
                a.slot_of[f] = SLOT_FAST;
            } else {
                uint32_t k = atomicAdd(a.slow_count, 1u);
                a.slow_list[k] = (uint32_t)(f - a.file0);
            }
        }
        w.sync();
    }
#ifdef TP_PROF
    w.flush();
#endif
}
__global__ __launch_bounds__(64) void k_long(DevTables t, FastArgs a) { long_body<false, LongLds>(t, a); }
__global__ __launch_bounds__(64) void k_long512(DevTables t, FastArgs a) { long_body<false, LongLds512>(t, a); }
__global__ __launch_bounds__(64) void k_long256(DevTables t, FastArgs a) { long_body<false, LongLds256>(t, a); }
__global__ __launch_bounds__(64) void k_long_triage(DevTables t, FastArgs a) { long_body<true, LongLds>(t, a); }
__global__ __launch_bounds__(64) void k_long512_triage(DevTables t, FastArgs a) { long_body<true, LongLds512>(t, a); }
__global__ __launch_bounds__(64) void k_long256_triage(DevTables t, FastArgs a) { long_body<true, LongLds256>(t, a); }
__global__ __launch_bounds__(64) void k_long_audit(DevTables t, FastArgs a) { long_body<false, LongLds, true>(t, a); }
__global__ __launch_bounds__(64) void k_long512_audit(DevTables t, FastArgs a) { long_body<false, LongLds512, true>(t, a); }
__global__ __launch_bounds__(64) void k_long256_audit(DevTables t, FastArgs a) { long_body<false, LongLds256, true>(t, a); }

// ------------------------------------------------------------------------------------------------
// k_wave_sizes
// ------------------------------------------------------------------------------------------------
__device__ inline uint32_t count_commas(const uint8_t* p, uint32_t n) {
    uint32_t c = 0, i = 0;
    while (i < n && (((uintptr_t)(p + i)) & 7)) { c += (p[i] == ','); ++i; }
    for (; i + 8 <= n; i += 8) {
        uint64_t w = *(const uint64_t*)(p + i) ^ 0x2C2C2C2C2C2C2C2Cull;   // zero byte where the input holds ','
        uint64_t t = (w & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full;
        t = ~(t | w | 0x7F7F7F7F7F7F7F7Full);
        c += (uint32_t)__popcll(t);
    }
    for (; i < n; ++i) c += (p[i] == ',');
    return c;
}

__global__ __launch_bounds__(256) void k_wave_sizes(DevTables t, const uint8_t* arena, const uint64_t* offsets, const uint64_t* tag_begin,
                                                    const uint32_t* item_list, uint64_t file0, uint64_t n_items, uint32_t* wave_T, uint32_t* wave_LB,
                                                    uint64_t* wave_bytes, uint32_t* status, uint32_t fpw) {
    // fpw files per wavefront (lanes 0 .. fpw - 1 hold one each): few files are spread over many wavefronts, see run_range
    const uint64_t th = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = (th >> 6) * fpw + (th & 63);
    const bool mine = (th & 63) < fpw && i < n_items;
    uint32_t T = 1, LB = 0;
    if (mine) {
        uint64_t f = file0 + (item_list ? (uint64_t)item_list[i] : i);
        if (tag_begin) {   // scored ingest: tag count and total text length are read off the offset arrays
            uint64_t lo = tag_begin[f], hi = tag_begin[f + 1];
            uint64_t ntag = hi - lo, bytes = offsets[hi] - offsets[lo];
            if (ntag > 0x7FFFFFFFull || bytes > 0x7FFFFFFFull) { atomicOr(status, 1u); ntag = 0; bytes = 0; }
            T = (uint32_t)ntag;
            LB = (uint32_t)bytes;
        } else {
            uint64_t len = offsets[f + 1] - offsets[f];
            if (len > 0x7FFFFFFFull) { atomicOr(status, 1u); len = 0; }
            LB = (uint32_t)len;
            T = (count_commas(arena + offsets[f], LB) + 1) * t.split_grow;   // (a token can become several candidates: FL_SPLIT)
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t oT = __shfl_xor(T, d), oL = __shfl_xor(LB, d);
        T = T > oT ? T : oT;
        LB = LB > oL ? LB : oL;
    }
    if ((threadIdx.x & 63) == 0 && i < n_items) {
        uint64_t w = th >> 6;
        WsLayout L = ws_layout(T, LB, t.final_cap, t.evidence_grow, t.lit_words);
        wave_T[w] = T;
        wave_LB[w] = LB;
        wave_bytes[w] = L.total;
    }
}

// ------------------------------------------------------------------------------------------------
// exclusive scan: out[i] = base + sum(in[0..i)), out[n] = base + total
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256, SCAN_PER_THREAD = 16, SCAN_TILE = SCAN_THREADS * SCAN_PER_THREAD;

template <class Tin>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(const Tin* in, uint64_t n, uint64_t* out, uint64_t* tile_sums) {
    __shared__ uint64_t wave_tot[SCAN_THREADS / 64];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    uint64_t v[SCAN_PER_THREAD];
    uint64_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) { v[k] = (base + k < n) ? (uint64_t)in[base + k] : 0; sum += v[k]; }
    uint64_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) { uint64_t o = __shfl_up(incl, d); if ((int)(threadIdx.x & 63) >= d) incl += o; }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t prefix = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) prefix += wave_tot[w];
    uint64_t run = prefix + incl - sum;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == SCAN_THREADS - 1) tile_sums[blockIdx.x] = prefix + incl;
}
__global__ __launch_bounds__(64) void k_scan_sums(uint64_t* tile_sums, uint64_t n_tiles) {
    // one wavefront walks the tile sums 64 at a time
    uint64_t carry = 0;
    for (uint64_t b = 0; b < n_tiles; b += 64) {
        uint64_t i = b + threadIdx.x;
        uint64_t x = i < n_tiles ? tile_sums[i] : 0;
        uint64_t incl = x;
        for (int d = 1; d < 64; d <<= 1) { uint64_t o = __shfl_up(incl, d); if ((int)threadIdx.x >= d) incl += o; }
        if (i < n_tiles) tile_sums[i] = carry + incl - x;
        carry += __shfl(incl, 63);
    }
    if (threadIdx.x == 0) tile_sums[n_tiles] = carry;
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_add(uint64_t* out, uint64_t n, const uint64_t* tile_sums, uint64_t n_tiles, const uint64_t* base_ptr,
                                                           uint64_t* total_out) {
    uint64_t base = base_ptr ? *base_ptr : 0;
    uint64_t add = tile_sums[blockIdx.x] + base;
    uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE;
    for (int k = threadIdx.x; k < SCAN_TILE; k += SCAN_THREADS) if (i0 + k < n) out[i0 + k] += add;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[n] = tile_sums[n_tiles] + base;
        if (total_out) *total_out = tile_sums[n_tiles] + base;
    }
}

template <class Tin>
static hipError_t scan_exclusive(const Tin* in, uint64_t n, uint64_t* out, uint64_t* tile_sums, const uint64_t* base_ptr, uint64_t* total_out, hipStream_t s) {
    uint64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (n_tiles == 0) n_tiles = 1;
    hipLaunchKernelGGL(k_scan_tiles<Tin>, dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, s, in, n, out, tile_sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, s, tile_sums, n_tiles);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, s, out, n, tile_sums, n_tiles, base_ptr, total_out);
    return hipGetLastError();
}
uint64_t scan_tiles_needed(uint64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 2; }

// ------------------------------------------------------------------------------------------------
// k_pipeline
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pipeline(DevTables t, PipelineArgs a) {
    const uint64_t th = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // workspace slot: wavefront th >> 6, lane th & 63
    const uint64_t i = (th >> 6) * a.fpw + (th & 63);                  // item index: a.fpw files per wavefront
    if ((th & 63) >= a.fpw || i >= a.n_items) return;
    uint64_t f = a.file0 + (a.item_list ? (uint64_t)a.item_list[i] : i);
    uint64_t w = th >> 6;
    uint32_t lane = (uint32_t)(th & 63);
    WsLayout L = ws_layout(a.wave_T[w], a.wave_LB[w], t.final_cap, t.evidence_grow, t.lit_words);
    FileWs ws = ws_bind(a.workspace + a.wave_off[w], L, lane, a.fpw);
    FileIn in;
    if (a.tag_begin) {
        in.text = nullptr; in.len = 0;
        in.arena = a.arena; in.tag_offsets = a.offsets; in.tag_lo = a.tag_begin[f]; in.tag_hi = a.tag_begin[f + 1];
        in.scores = a.scores; in.sources = a.sources;
        in.merged = a.secondary_begin != nullptr; in.tag_mid = in.merged ? a.secondary_begin[f] : in.tag_hi; in.secondary_cap = a.secondary_cap;
        in.keep_order = a.reply_flags != nullptr && !(a.reply_flags[f] & 1u);
    } else {
        uint64_t o = a.offsets[f];
        in.text = a.arena + o; in.len = (uint32_t)(a.offsets[f + 1] - o);
        in.arena = nullptr; in.tag_offsets = nullptr; in.tag_lo = in.tag_hi = 0; in.scores = nullptr; in.sources = nullptr;
        in.merged = false; in.tag_mid = 0; in.secondary_cap = 0; in.keep_order = false;
    }
    FileOut out;
    process_file(t, in, ws, out, a.fired ? a.fired + f * t.fired_words : nullptr, a.facet_choice ? a.facet_choice + f * t.n_facets : nullptr, a.mode);
    a.verdicts[f] = out.verdict;
    a.n_final[f] = out.n_final;
    a.cap_len[f] = out.caption_len;
    a.slot_of[f] = (uint32_t)th;
    if (a.audit_cnt) {
        for (int c = 0; c < AUDIT_COLS; ++c) {
            a.audit_cnt[(f - a.file0) * AUDIT_COLS + c] = out.audit_cnt[c];
            a.audit_len[(f - a.file0) * AUDIT_COLS + c] = out.audit_len[c];
        }
        a.review_bits[f - a.file0] = out.review_bits;
    }
}

constexpr uint32_t GEN_STAGE = 4096, AUDIT_STAGE = 4096;   // bytes of LDS staging per wavefront in the general emit kernels (multiples of 16)
__device__ inline void put_bytes(uint8_t* dst, const uint8_t* src, uint32_t len) {   // src is readable eight bytes past its end
    uint32_t b = 0;
    for (; b + 8 <= len; b += 8) { const uint64_t v = ld64u(src + b); __builtin_memcpy(dst + b, &v, 8); }
    if (b < len) {
        uint64_t v = ld64u(src + b);
        const uint32_t r = len - b;
        if (r & 4) { const uint32_t y = (uint32_t)v; __builtin_memcpy(dst + b, &y, 4); v >>= 32; b += 4; }
        if (r & 2) { const uint16_t y = (uint16_t)v; __builtin_memcpy(dst + b, &y, 2); v >>= 16; b += 2; }
        if (r & 1) dst[b] = (uint8_t)v;
    }
}
// The tag lists of the audit row and the review record as text: one wavefront per file, one entry per lane and step.  The
// audit row's entries are joined with ";" (Program.cs:966), dropped tags carry their reason (Program.cs:630-631); the review
// record's lists are separated by "\n" (the host sorts and joins them); the unknown-tag column keeps only entries that qualify.
__global__ __launch_bounds__(256) void k_emit_audit(DevTables t, AuditArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[4][AUDIT_STAGE];
    __shared__ uint32_t ids_all[4][AUD_STRIDE + FAST_FIN_MAX];   // a wave-stage file's lists: its audit row, then its caption's tag ids
    uint8_t* stage = stage_all[wave_in_block()];
    uint32_t* ids = ids_all[wave_in_block()];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave_global = wave_index();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    // One file ahead the wave reads what it needs to know before it can ask for a file's lists: the file, its kind, the columns'
    // entry counts and places in the text (lanes 0 .. 5), whether the kept tags differ from the caption's (lane 6).
    struct Meta { uint64_t f; uint32_t slot, cnt; uint64_t off; };
    auto load_meta = [&](uint64_t i, Meta& m) {
        m.f = 0; m.slot = 0; m.cnt = 0; m.off = 0;
        if (i >= a.n_items) return;
        m.f = a.file0 + (a.item_list ? a.item_list[i] : i);
        m.slot = a.slot_of[m.f];
        if (lane < (uint32_t)AUDIT_COLS) m.cnt = a.audit_cnt[(m.f - a.file0) * AUDIT_COLS + lane];
        if (lane == 6 && a.aud32 != nullptr && m.slot == SLOT_FAST) m.cnt = a.aud32[(m.f - a.file0) * AUD_STRIDE + AUD_FLAGS];
        if (lane <= (uint32_t)AUDIT_COLS) m.off = a.offsets[m.f * AUDIT_COLS + lane];
    };
    Meta m0, m1;
    load_meta(wave_global, m0);
    for (uint64_t i = wave_global; i < a.n_items; i += n_waves, m0 = m1) {
        load_meta(i + n_waves, m1);
        const uint64_t f = m0.f;
        const uint32_t slot = m0.slot;
        // a file of the lean stage: its lists are interned ids in its rows of fin32 and aud32; else they are in its workspace
        const bool lean_file = a.aud32 != nullptr && slot == SLOT_FAST;
        const uint32_t* fin_row = lean_file ? a.fin32 + (f - a.file0) * FAST_FIN_MAX : nullptr;
        const uint32_t* aud_row = lean_file ? a.aud32 + (f - a.file0) * AUD_STRIDE : nullptr;
        const uint64_t w = lean_file ? 0 : slot >> 6;
        WsLayout L{};
        FileWs ws{};
        if (!lean_file) { L = ws_layout(a.wave_T[w], a.wave_LB[w], t.final_cap, t.evidence_grow, t.lit_words); ws = ws_bind(a.workspace + a.wave_off[w], L, slot & 63, a.fpw); }
        auto lane_of = [](uint32_t v, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); };
        uint32_t cnt_[AUDIT_COLS];
        uint64_t off_[AUDIT_COLS + 1];
        for (int c = 0; c < AUDIT_COLS; ++c) cnt_[c] = lane_of(m0.cnt, c);
        for (int c = 0; c <= AUDIT_COLS; ++c) off_[c] = lane_of((uint32_t)m0.off, c) | (uint64_t)lane_of((uint32_t)(m0.off >> 32), c) << 32;
        if (lean_file) {
            // every list of the file in one round trip: the entries the counts name, into LDS
            const bool kept_in_row = lane_of(m0.cnt, 6) != 0;
            const bool unknown_text = off_[AUDIT_UNKNOWN + 1] != off_[AUDIT_UNKNOWN];
            const uint32_t fin_unknown = unknown_text ? cnt_[AUDIT_UNKNOWN] : 0u, fin_kept = kept_in_row ? 0u : cnt_[AUDIT_KEPT];   // the caption's ids are the unknown-tag column's, and the kept tags' when those are the caption's
            const uint32_t n_fin = fin_unknown > fin_kept ? fin_unknown : fin_kept;
            uint32_t ev[3], dr = 0, tp = 0, kp[2], fn[2];
            for (uint32_t j = 0; j < 3; ++j) { const uint32_t k = lane + 64 * j; ev[j] = k < cnt_[AUDIT_EVIDENCE] && k < AUD_DROPPED - AUD_EVIDENCE ? aud_row[AUD_EVIDENCE + k] : 0u; }
            if (lane < cnt_[AUDIT_DROPPED] && lane < AUD_TOP - AUD_DROPPED) dr = aud_row[AUD_DROPPED + lane];
            if (lane < cnt_[AUDIT_TOP] && lane < AUD_FLAGS - AUD_TOP) tp = aud_row[AUD_TOP + lane];
            for (uint32_t j = 0; j < 2; ++j) {
                const uint32_t k = lane + 64 * j;
                kp[j] = kept_in_row && k < cnt_[AUDIT_KEPT] && k < FAST_FIN_MAX ? aud_row[AUD_KEPT + k] : 0u;
                fn[j] = k < n_fin && k < FAST_FIN_MAX ? fin_row[k] : 0u;
            }
            __builtin_amdgcn_wave_barrier();                        // the last file's entries have been read
            for (uint32_t j = 0; j < 3; ++j) ids[AUD_EVIDENCE + lane + 64 * j] = ev[j];
            ids[AUD_DROPPED + lane] = dr;
            if (lane < AUD_FLAGS - AUD_TOP) ids[AUD_TOP + lane] = tp;
            for (uint32_t j = 0; j < 2; ++j) { ids[AUD_KEPT + lane + 64 * j] = kept_in_row ? kp[j] : fn[j]; ids[AUD_STRIDE + lane + 64 * j] = fn[j]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // an entry's string into the staging area: from its 32-byte caption record, or from the string arena when it is longer than 31 bytes
        auto put32 = [&](uint8_t* dst, const uint4& r0, const uint4& r1, uint32_t n, bool in_rec) {
            const uint8_t* sp = t.str_bytes + (in_rec ? 0u : r0.x);
            uint64_t w8[4];
            w8[0] = r0.x | (uint64_t)r0.y << 32; w8[1] = r0.z | (uint64_t)r0.w << 32;
            w8[2] = r1.x | (uint64_t)r1.y << 32; w8[3] = r1.z | (uint64_t)(r1.w & 0xFFFFFFu) << 32;
            if (!in_rec && n) { __builtin_memcpy(&w8[0], sp, 16); __builtin_memcpy(&w8[2], sp + 16, 16); }   // (the string arena has slack behind its end)
            const uint32_t n32 = n < 32 ? n : 32, nfull = n32 >> 3;
            for (uint32_t q = 0; q < 4; ++q)
                if (q < nfull) __builtin_memcpy(dst + 8 * q, &w8[q], 8);
            if (n32 & 7) {                                           // the last, partial word in pieces of 4, 2 and 1 bytes
                uint64_t x = nfull == 0 ? w8[0] : (nfull == 1 ? w8[1] : (nfull == 2 ? w8[2] : w8[3]));
                uint32_t o = 8 * nfull;
                if (n32 & 4) { uint32_t y = (uint32_t)x; __builtin_memcpy(dst + o, &y, 4); x >>= 32; o += 4; }
                if (n32 & 2) { uint16_t y = (uint16_t)x; __builtin_memcpy(dst + o, &y, 2); x >>= 16; o += 2; }
                if (n32 & 1) dst[o] = (uint8_t)x;
            }
            for (uint32_t done = 32; done < n; ++done) dst[done] = sp[done];
        };
        if (lean_file) {
            // The five columns of a file are consecutive in the text: when they fit the staging area together they are ONE list of
            // entries (a lane tells its column by the columns' cumulative counts, as the caption writer tells a run's files apart):
            // full lane blocks instead of five columns of which three hold a handful of entries.  (Asking for the records of all the
            // file's blocks before the first is put together was slower: c2 0.88 against 0.84 ms, c3c 1.29 against 1.01.)
            uint32_t cum[AUDIT_COLS + 1];
            cum[0] = 0;
            for (int c = 0; c < AUDIT_COLS; ++c) cum[c + 1] = cum[c] + (off_[c + 1] != off_[c] ? cnt_[c] : 0u);
            const uint32_t T = cum[AUDIT_COLS];
            const uint64_t total64 = off_[AUDIT_COLS] - off_[0];
            uint8_t* out = a.text + off_[0];
            const uint32_t head = (uint32_t)((uintptr_t)out & 15);
            if (T == 0 || total64 == 0) continue;
            if (head + total64 <= (uint64_t)AUDIT_STAGE) {
                const uint32_t total = (uint32_t)total64;
                uint32_t at = head, unknown_seen = 0;
                for (uint32_t g0 = 0; g0 < T; g0 += 64) {
                    const uint32_t gi = g0 + lane;
                    const uint32_t col = (uint32_t)(gi >= cum[1]) + (uint32_t)(gi >= cum[2]) + (uint32_t)(gi >= cum[3]) + (uint32_t)(gi >= cum[4]);
                    const uint32_t k = gi - (col == 0 ? 0u : col == 1 ? cum[1] : col == 2 ? cum[2] : col == 3 ? cum[3] : cum[4]);
                    const uint32_t base = col == (uint32_t)AUDIT_KEPT ? AUD_KEPT : col == (uint32_t)AUDIT_EVIDENCE ? AUD_EVIDENCE : col == (uint32_t)AUDIT_DROPPED ? AUD_DROPPED
                                          : col == (uint32_t)AUDIT_TOP ? AUD_TOP : AUD_STRIDE;
                    const bool in = gi < T;
                    const uint32_t v = in ? ids[base + k] : 0u;
                    const uint32_t id = v & 0x7FFFFFFFu;
                    const bool unk_col = col == (uint32_t)AUDIT_UNKNOWN;
                    const bool live = in && (!unk_col || (t.known_active && !(t.flags[id] & FL_KNOWN)));
                    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
                    if (live) { const uint4* rp = (const uint4*)(t.emit_rec32 + 8 * (size_t)id); r0 = rp[0]; r1 = rp[1]; }
                    const uint64_t mu = __ballot(live && unk_col ? 1 : 0);
                    const bool first = unk_col ? unknown_seen + __builtin_amdgcn_mbcnt_hi((uint32_t)(mu >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mu, 0u)) == 0 : k == 0;
                    unknown_seen += (uint32_t)__popcll(mu);
                    const uint32_t lenb = r1.w >> 24;
                    const bool in_rec = lenb != 0xFFu;
                    const uint32_t n = live ? (in_rec ? lenb : r0.y) : 0u;
                    const bool reason = col == (uint32_t)AUDIT_DROPPED;
                    const uint32_t seg = live ? n + (reason ? AUDIT_REASON_LEN : 0u) + (first ? 0u : 1u) : 0u;
                    const uint32_t incl = wave_scan_incl(seg);
                    if (live) {
                        uint8_t* dst = stage + at + (incl - seg);
                        if (!first) *dst++ = col >= (uint32_t)AUDIT_TOP ? (uint8_t)'\n' : (uint8_t)';';
                        put32(dst, r0, r1, n, in_rec);
                        if (reason) { const char* rs = (v >> 31) ? ":policy_drop" : ":invalid_tag"; for (uint32_t q = 0; q < AUDIT_REASON_LEN; ++q) dst[n + q] = (uint8_t)rs[q]; }
                    }
                    at += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint8_t* gbase = out - head;                            // 16-byte aligned
                const uint32_t end = head + total, nchunks = (end + 15) >> 4;
                for (uint32_t ch = lane; ch < nchunks; ch += 64)
                    if (16 * ch >= head && 16 * ch + 16 <= end) *(uint4*)(gbase + 16 * ch) = *(const uint4*)(stage + 16 * ch);
                if (lane < 32) {                                        // the partial first and last pieces, one byte per lane
                    const uint32_t ch = lane < 16 ? 0u : nchunks - 1;
                    const uint32_t off = 16 * ch + (lane & 15);
                    const bool partial = !(16 * ch >= head && 16 * ch + 16 <= end) && (lane < 16 || nchunks > 1);
                    if (partial && off >= head && off < end) gbase[off] = stage[off];
                }
                __builtin_amdgcn_wave_barrier();                        // the area is read before the next file's bytes are written
                continue;
            }
        }
        for (int c = 0; c < AUDIT_COLS; ++c) {
            const uint32_t cnt = cnt_[c];
            if (off_[c + 1] == off_[c]) continue;                    // no text: none of the entries qualifies (the usual unknown-tag column)
            uint8_t* out = a.text + off_[c];
            const uint8_t sep = c >= AUDIT_TOP ? (uint8_t)'\n' : (uint8_t)';';
            uint32_t running = 0, kept_before = 0;
            if (lean_file) {
                // Interned ids: an entry's text comes with its 32-byte caption record (strings up to 31 bytes inline) in one round trip, the
                // group is put together in LDS at the output's 16-byte phase and stored 16 bytes a lane, as the caption writer does.
                for (uint32_t k0 = 0; k0 < cnt; k0 += 64) {
                    const uint32_t k = k0 + lane;
                    uint32_t v = 0;
                    if (k < cnt) v = ids[(c == AUDIT_UNKNOWN ? AUD_STRIDE : c == AUDIT_KEPT ? AUD_KEPT : c == AUDIT_EVIDENCE ? AUD_EVIDENCE : c == AUDIT_DROPPED ? AUD_DROPPED : AUD_TOP) + k];
                    const uint32_t id = v & 0x7FFFFFFFu;
                    const bool live = k < cnt && (c != AUDIT_UNKNOWN || (t.known_active && !(t.flags[id] & FL_KNOWN)));
                    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
                    if (live) { const uint4* rp = (const uint4*)(t.emit_rec32 + 8 * (size_t)id); r0 = rp[0]; r1 = rp[1]; }
                    const uint32_t lenb = r1.w >> 24;
                    const bool in_rec = lenb != 0xFFu;
                    const uint32_t n = live ? (in_rec ? lenb : r0.y) : 0u;
                    const bool reason = c == AUDIT_DROPPED;
                    const uint64_t m = __ballot(live ? 1 : 0);
                    const bool first = kept_before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0;
                    kept_before += (uint32_t)__popcll(m);
                    const uint32_t seg = live ? n + (reason ? AUDIT_REASON_LEN : 0u) + (first ? 0u : 1u) : 0u;
                    const uint32_t incl = wave_scan_incl(seg);
                    const uint32_t group = __shfl(incl, 63);
                    const uint32_t head = (uint32_t)((uintptr_t)(out + running) & 15);
                    const bool staged = head + group <= AUDIT_STAGE;
                    if (live) {
                        uint8_t* dst = (staged ? stage + head : out + running) + (incl - seg);
                        if (!first) *dst++ = sep;
                        const uint8_t* sp = t.str_bytes + (in_rec ? 0u : r0.x);
                        if (staged) put32(dst, r0, r1, n, in_rec);
                        else {
                            if (in_rec) { uint8_t bytes[32]; __builtin_memcpy(bytes, &r0, 16); __builtin_memcpy(bytes + 16, &r1, 16); for (uint32_t q = 0; q < n; ++q) dst[q] = bytes[q]; }
                            else for (uint32_t q = 0; q < n; ++q) dst[q] = sp[q];
                        }
                        if (reason) { const char* rs = (v >> 31) ? ":policy_drop" : ":invalid_tag"; for (uint32_t q = 0; q < AUDIT_REASON_LEN; ++q) dst[n + q] = (uint8_t)rs[q]; }
                    }
                    if (staged) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        uint8_t* gbase = out + running - head;             // 16-byte aligned
                        const uint32_t end = head + group, nchunks = (end + 15) >> 4;
                        for (uint32_t ch = lane; ch < nchunks; ch += 64)
                            if (16 * ch >= head && 16 * ch + 16 <= end) *(uint4*)(gbase + 16 * ch) = *(const uint4*)(stage + 16 * ch);
                        if (lane < 32 && nchunks) {                         // the partial first and last pieces, one byte per lane
                            const uint32_t ch = lane < 16 ? 0u : nchunks - 1;
                            const uint32_t off = 16 * ch + (lane & 15);
                            const bool partial = !(16 * ch >= head && 16 * ch + 16 <= end) && (lane < 16 || nchunks > 1);
                            if (partial && off >= head && off < end) gbase[off] = stage[off];
                        }
                        __builtin_amdgcn_wave_barrier();                // the area is read before the next group's bytes are written
                    }
                    running += group;
                }
                continue;
            }
            for (uint32_t k0 = 0; k0 < cnt; k0 += 64) {
                const uint32_t k = k0 + lane;
                AuditElem e{nullptr, 0, nullptr, false};
                if (k < cnt) e = audit_elem(t, ws, c, k);
                const bool live = e.keep;
                const uint64_t m = __ballot(live ? 1 : 0);
                const bool first = kept_before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0;
                kept_before += (uint32_t)__popcll(m);
                const uint32_t seg = live ? e.len + (e.reason ? AUDIT_REASON_LEN : 0u) + (first ? 0u : 1u) : 0u;
                const uint32_t incl = wave_scan_incl(seg);
                const uint32_t group = __shfl(incl, 63);
                // the group's text goes through the staging area (eight bytes per load, consecutive lanes store consecutive bytes);
                // a group too long for it is stored lane by lane
                const bool staged = group <= AUDIT_STAGE;
                if (live) {
                    uint8_t* dst = (staged ? stage : out + running) + (incl - seg);
                    if (!first) *dst++ = sep;
                    if (staged) put_bytes(dst, e.p, e.len);
                    else for (uint32_t b = 0; b < e.len; ++b) dst[b] = e.p[b];
                    if (e.reason) for (uint32_t b = 0; b < AUDIT_REASON_LEN; ++b) dst[e.len + b] = (uint8_t)e.reason[b];
                }
                if (staged) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    for (uint32_t o = lane; o < group; o += 64) out[running + o] = stage[o];
                    __builtin_amdgcn_wave_barrier();                // the area is read before the next group's bytes are written
                }
                running += group;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_emit
// ------------------------------------------------------------------------------------------------
constexpr int EMIT_WAVES = 4;           // wavefronts (files in flight) per block of the staged emit kernel
constexpr uint32_t HIST_IDS = 65536;    // ids one histogram pass counts in LDS: 16-bit counters, two to a word (128 KB)
constexpr uint32_t HIST_FILES_PER_FLUSH = 60000;   // files a block counts before its 16-bit counters are flushed
constexpr uint32_t EMIT_STAGE = FAST_CAPTION_MAX;   // captions are put together in LDS and stored 16 bytes per lane

struct EmitCounts {
    uint64_t files = 0, low = 0, elig = 0, foot = 0, fb = 0, rej = 0, uns = 0, safe = 0;
    __device__ void add(uint32_t fl) {
        ++files;
        low += (fl & TP_F_LOW) != 0; elig += (fl & TP_F_ELIGIBLE) != 0; foot += (fl & TP_F_FOOTJOB) != 0; fb += (fl & TP_F_FALLBACK) != 0;
        rej += (fl & TP_F_LINE_REJECTED) != 0; uns += (fl & TP_F_UNSUPPORTED) != 0; safe += (fl & TP_F_SAFE_REJECT) != 0;
    }
    // The block's wavefronts add their counts up in LDS and one thread hands the sums on: every thread of the block must
    // call this (it holds a block barrier).  One global atomic per wavefront and counter - thousands of wavefronts ending at
    // about the same time, all on the same eight addresses - used to cost k_emit a fifth of its time.
    __device__ void flush_block(tp_counters* out, unsigned long long* blk8) const {
        if (threadIdx.x < 8) blk8[threadIdx.x] = 0;
        __syncthreads();
        if ((threadIdx.x & 63u) == 0) {
            const uint64_t v[8] = {files, low, elig, foot, fb, rej, uns, safe};
            for (int k = 0; k < 8; ++k) if (v[k]) atomicAdd(&blk8[k], (unsigned long long)v[k]);
        }
        __syncthreads();
        if (threadIdx.x < 8 && blk8[threadIdx.x]) atomicAdd((unsigned long long*)out + threadIdx.x, blk8[threadIdx.x]);
    }
    __device__ void flush(tp_counters* out) const {
        unsigned long long* c = (unsigned long long*)out;
        if (files) atomicAdd(&c[0], (unsigned long long)files);
        if (low) atomicAdd(&c[1], (unsigned long long)low);
        if (elig) atomicAdd(&c[2], (unsigned long long)elig);
        if (foot) atomicAdd(&c[3], (unsigned long long)foot);
        if (fb) atomicAdd(&c[4], (unsigned long long)fb);
        if (rej) atomicAdd(&c[5], (unsigned long long)rej);
        if (uns) atomicAdd(&c[6], (unsigned long long)uns);
        if (safe) atomicAdd(&c[7], (unsigned long long)safe);
    }
};

// Files of the wave-per-file kernel: every tag is an interned string, at most 128 tags, at most EMIT_STAGE caption bytes.
// One wavefront per file.  The caption is put together in LDS at the same 16-byte phase as its place in the output,
// then stored with aligned 16-byte stores.  Two files ahead the wave loads a file's sizes and tag ids, one file ahead
// the tags' string offsets, so that the dependent loads of three files are in flight at once.
#ifndef TP_EMIT_WAVES_PER_SIMD
#define TP_EMIT_WAVES_PER_SIMD 5
#endif
__global__ __launch_bounds__(64 * EMIT_WAVES) __attribute__((amdgpu_waves_per_eu(TP_EMIT_WAVES_PER_SIMD))) void k_emit(DevTables t, EmitArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[EMIT_WAVES][EMIT_STAGE + 32];
    const uint32_t lane = threadIdx.x & 63;
    uint8_t* stage = stage_all[wave_in_block()];
    const uint64_t wave_global = wave_index();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    EmitCounts cnt;

    struct Meta { uint32_t nfin, total, slot, fl; uint64_t out0; uint32_t id[2]; };
    struct Strs { uint4 rec[2]; };                                     // DevTables::emit_rec of the lane's two tags
    auto load_meta = [&](uint64_t f, Meta& m) {
        m.nfin = 0; m.total = 0; m.slot = 0; m.fl = 0; m.out0 = 0; m.id[0] = m.id[1] = 0;
        if (f >= a.file_end) return;
        m.nfin = a.n_final[f];
        m.out0 = a.caption_offsets[f];
        m.total = (uint32_t)(a.caption_offsets[f + 1] - m.out0);
        m.fl = a.verdicts[f].flags;
        m.slot = a.slot_of[f];
        // ids are read whatever the file's kind and size turn out to be (the loads must not wait for those): a slot of the
        // id array exists for every file, and anything that is not an id of this file is clamped to a valid string here and
        // ignored later.
        const uint32_t* ids = a.fin32 + (f - a.file0) * FAST_FIN_MAX;
        const uint32_t i0 = ids[lane], i1 = ids[lane + 64];
        m.id[0] = i0 < t.n_strings ? i0 : 0;
        m.id[1] = i1 < t.n_strings ? i1 : 0;
    };
    auto load_strs = [&](const Meta& m, Strs& q) {
#ifdef EMIT_X_NOGATHER
        for (int blk = 0; blk < 2; ++blk) q.rec[blk] = make_uint4(m.id[blk], m.id[blk], 7, 12u << 24);
#else
        for (int blk = 0; blk < 2; ++blk) q.rec[blk] = *(const uint4*)(t.emit_rec + 4 * (size_t)m.id[blk]);   // most strings whole, in one 16-byte load
#endif
    };
    Meta m0, m1, m2;
    Strs q0, q1;
    const uint64_t f_first = a.file0 + wave_global;
    load_meta(f_first, m0);
    load_meta(f_first + n_waves, m1);
    load_strs(m0, q0);
    for (uint64_t f = f_first; f < a.file_end; f += n_waves, m0 = m1, m1 = m2, q0 = q1) {
        load_meta(f + 2 * n_waves, m2);
        load_strs(m1, q1);
        if (m0.slot != SLOT_FAST) continue;                        // a file of the general kernel: k_emit_general writes it
        const uint32_t nfin = m0.nfin, total = m0.total;
        const uint64_t out0 = m0.out0;
        if (lane == 0) cnt.add(m0.fl);
        if (total == 0) continue;
        const uint32_t head = (uint32_t)((uintptr_t)(a.captions + out0) & 15);
        uint32_t len[2], seg[2], incl[2];
        for (int blk = 0; blk < 2; ++blk) {
            const uint32_t k = (uint32_t)blk * 64 + lane;
            const bool live = k < nfin;
            len[blk] = live ? ((q0.rec[blk].w >> 24) == 0xFFu ? q0.rec[blk].y : q0.rec[blk].w >> 24) : 0;
            seg[blk] = len[blk] + ((live && k + 1 < nfin) ? 2u : 0u);
            incl[blk] = (blk == 0 || nfin > 64) ? wave_scan_incl(seg[blk]) : 0;
        }
        const uint32_t tot0 = __shfl(incl[0], 63);
        for (int blk = 0; blk < 2; ++blk) {
            if (blk == 1 && nfin <= 64) break;
            uint8_t* dst = stage + head + (blk ? tot0 : 0) + incl[blk] - seg[blk];
            const uint4 er = q0.rec[blk];
#ifdef EMIT_X_NOLONG
            const bool in_rec = true;
#else
            const bool in_rec = (er.w >> 24) != 0xFFu;                // up to 15 bytes: the record holds them
#endif
            const uint8_t* sp = t.str_bytes + (in_rec ? 0u : er.x);
            const uint32_t n = len[blk];
            // a longer string's first 32 bytes are fetched at once (the string arena has slack behind its end); stored exactly
            uint64_t v[4];
            v[0] = er.x | (uint64_t)er.y << 32; v[1] = er.z | (uint64_t)(er.w & 0xFFFFFFu) << 32; v[2] = 0; v[3] = 0;
            if (!in_rec && n) {
                __builtin_memcpy(&v[0], sp, 16);
                __builtin_memcpy(&v[2], sp + 16, 16);
            }
            const uint32_t n32 = n < 32 ? n : 32, nfull = n32 >> 3;
            for (uint32_t k = 0; k < 4; ++k)
                if (k < nfull) __builtin_memcpy(dst + 8 * k, &v[k], 8);
            if (n32 & 7) {                                           // the last, partial word in pieces of 4, 2 and 1 bytes
                uint64_t x = nfull == 0 ? v[0] : (nfull == 1 ? v[1] : (nfull == 2 ? v[2] : v[3]));
                uint32_t o = 8 * nfull;
                if (n32 & 4) { uint32_t y = (uint32_t)x; __builtin_memcpy(dst + o, &y, 4); x >>= 32; o += 4; }
                if (n32 & 2) { uint16_t y = (uint16_t)x; __builtin_memcpy(dst + o, &y, 2); x >>= 16; o += 2; }
                if (n32 & 1) dst[o] = (uint8_t)x;
            }
            for (uint32_t done = 32; done < n; ++done) dst[done] = sp[done];
            if (seg[blk] != len[blk]) { dst[n] = ','; dst[n + 1] = ' '; }
        }
        if (lane < t.newline_len) stage[head + total - t.newline_len + lane] = t.newline[lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the staging area is this wave's own: no wait for global traffic
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint8_t* gbase = a.captions + out0 - head;                    // 16-byte aligned
        const uint32_t end = head + total, nchunks = (end + 15) >> 4;
#ifdef EMIT_X_NOSTORE
        if (total == 0xFFFFFFFFu)
#endif
        for (uint32_t c = lane; c < nchunks; c += 64) {
            if (16 * c >= head && 16 * c + 16 <= end) *(uint4*)(gbase + 16 * c) = *(const uint4*)(stage + 16 * c);
        }
        if (lane < 32) {                                              // the partial first and last pieces, one byte per lane
            const uint32_t c = lane < 16 ? 0u : nchunks - 1;
            const uint32_t off = 16 * c + (lane & 15);
            const bool partial = !(16 * c >= head && 16 * c + 16 <= end) && (lane < 16 || nchunks > 1);
            if (partial && off >= head && off < end) gbase[off] = stage[off];
        }
        __builtin_amdgcn_wave_barrier();                              // the staging bytes are read before the next file's are written
    }
    __shared__ unsigned long long blk_cnt[8];
    if (a.counters) cnt.flush_block(a.counters, blk_cnt);
}

// The same files, several to a wavefront.  The captions of consecutive files are consecutive in the output, so a run of
// wave-kernel files is one list of tags in which the last tag of each file is followed by the line end instead of ", ".
// A wavefront takes a group of RUN_FILES consecutive files: as one run when all of them are wave-kernel files and the run
// fits (RUN_BLOCKS lane blocks of tags, RUN_STAGE caption bytes), else file by file.  Every lane block of the run loads its
// tags' ids, then their 32-byte caption records (strings up to 31 bytes inline), and only then the text is put together in
// LDS at the output's 16-byte phase and stored 16 bytes per lane.  Per file this takes about two thirds of the
// instructions of one file per wavefront (c2's 73 tags fill 1.14 of that kernel's two lane blocks) and a third of its
// dependent memory round trips.
constexpr int RUN_FILES = 4;
constexpr int RUN_BLOCKS = 5;
constexpr uint32_t RUN_TAGS = 64 * RUN_BLOCKS;
constexpr uint32_t RUN_STAGE = 6144;
static_assert(RUN_STAGE >= FAST_CAPTION_MAX && RUN_TAGS >= FAST_FIN_MAX, "a single file always fits a run");

// One group of up to RUN_FILES consecutive files: rows = the first file's row of tag ids (the others follow at FAST_FIN_MAX
// words each); bit j of fastmask: file j is a wave-kernel file (the others are written elsewhere); n_[j] its tag count;
// off_[j] .. off_[j + 1] its place in `captions`.  stage: RUN_STAGE + 48 bytes of LDS of this wavefront's own.
// HIST: every tag id of the group is also counted in `bins` (LDS, two 16-bit counters a word, ids below n_public) and the
// learned strings among them in n_learned; STAGE_CAP is the staging area's size.
#ifndef TP_EMIT_DIAG
#define TP_EMIT_DIAG 0        // diagnostic builds of the caption writer: 1 = no caption stores, 3 = made-up records instead of the gather (wrong captions: timing only)
#endif
template <bool HIST = false, uint32_t STAGE_CAP = RUN_STAGE, class TT>
__device__ __forceinline__ void emit_group(const TT& t, const uint32_t* rows, uint8_t* captions, uint8_t* stage, uint32_t lane, uint32_t nfile, uint32_t fastmask,
                                           const uint32_t (&n_)[RUN_FILES], const uint64_t (&off_)[RUN_FILES + 1], uint32_t* bins = nullptr, uint32_t* n_learned = nullptr) {
    static_assert(STAGE_CAP >= FAST_CAPTION_MAX, "a single file always fits the staging area");
    uint32_t tags_all = 0;
    for (int j = 0; j < RUN_FILES; ++j) tags_all += ((fastmask >> j) & 1u) ? n_[j] : 0u;
    const bool whole = fastmask == (1u << nfile) - 1u && tags_all <= RUN_TAGS && off_[nfile] - off_[0] <= (uint64_t)STAGE_CAP;
    // the runs of the group: [0, nfile) as a whole, or every wave-kernel file by itself
    for (uint32_t lo = 0; lo < nfile; lo = whole ? nfile : lo + 1) {
        const uint32_t hi = whole ? nfile : lo + 1;
        if (!whole && !((fastmask >> lo) & 1u)) continue;
        uint32_t nn[RUN_FILES];
        for (int j = 0; j < RUN_FILES; ++j) nn[j] = ((uint32_t)j >= lo && (uint32_t)j < hi) ? n_[j] : 0u;
        const uint32_t c1 = nn[0], c2 = c1 + nn[1], c3 = c2 + nn[2], T = c3 + nn[3];
        static_assert(RUN_FILES == 4, "the cumulative counts above are written out for four files");
        uint64_t out0 = off_[0], out1 = off_[RUN_FILES];
        for (int j = 1; j < RUN_FILES; ++j) { if ((uint32_t)j == lo) out0 = off_[j]; if ((uint32_t)j == hi) out1 = off_[j]; }
        const uint32_t total = (uint32_t)(out1 - out0);
        if (T == 0 || total == 0) continue;
        const uint32_t head = (uint32_t)((uintptr_t)(captions + out0) & 15);

        // which file and which of its tags a lane holds in each lane block; the ids, then the records
        uint32_t id[RUN_BLOCKS];                                   // bit 31: the last tag of its file
        TP_UNROLL for (int b = 0; b < RUN_BLOCKS; ++b) {
            const uint32_t gi = 64u * b + lane;
            const uint32_t j = (uint32_t)(gi >= c1) + (uint32_t)(gi >= c2) + (uint32_t)(gi >= c3);
            const uint32_t cj = j == 0 ? 0u : j == 1 ? c1 : j == 2 ? c2 : c3, nj = j == 0 ? nn[0] : j == 1 ? nn[1] : j == 2 ? nn[2] : nn[3];
            const uint32_t kb = gi - cj;
            id[b] = 0;
            if (64u * b < T && gi < T) {
                const uint32_t v = rows[(size_t)j * FAST_FIN_MAX + kb];
                id[b] = (v < t.n_strings ? v : 0) | (uint32_t)(kb + 1 == nj) << 31;
            }
        }
        if constexpr (HIST) {                                       // a tag occurs once in a file: the 16-bit counters hold as many files as the launch gives a block
            TP_UNROLL for (int b = 0; b < RUN_BLOCKS; ++b) {
                if (64u * b < T && 64u * b + lane < T) {
                    const uint32_t v = id[b] & 0x7FFFFFFFu;
                    if (v < t.n_public) atomicAdd(&bins[v >> 1], 1u << (16 * (v & 1)));
                    else *n_learned += 1;                           // (v < n_strings: a learned string, counted with the tags in no table)
                }
            }
        }
        uint4 r0[RUN_BLOCKS], r1[RUN_BLOCKS];
        TP_UNROLL for (int b = 0; b < RUN_BLOCKS; ++b) {
            r0[b] = make_uint4(0, 0, 0, 0); r1[b] = r0[b];
#if TP_EMIT_DIAG == 3
            r0[b] = make_uint4(id[b], 0x61616161u, 0x62626262u, 0x63636363u); r1[b] = make_uint4(0, 0, 0, (8u + (id[b] & 7u)) << 24);   // no record loads (diagnostic build)
            if (false) {
#else
            if (64u * b < T) {
#endif
                const uint4* rp = (const uint4*)(t.emit_rec32 + 8 * (size_t)(id[b] & 0x7FFFFFFFu));
                r0[b] = rp[0]; r1[b] = rp[1];
            }
        }
        uint32_t at = head;                                        // where the lane block's text begins in the staging area
        TP_UNROLL for (int b = 0; b < RUN_BLOCKS; ++b) {
            if (64u * b >= T) break;
            const bool live = 64u * b + lane < T;
            const uint32_t lenb = r1[b].w >> 24;
            const bool in_rec = lenb != 0xFFu;
            const uint32_t n = live ? (in_rec ? lenb : r0[b].y) : 0u;
            const bool last = (id[b] >> 31) != 0;
            const uint32_t seg = live ? n + (last ? t.newline_len : 2u) : 0u;
            const uint32_t incl = wave_scan_incl(seg);
            uint8_t* dst = stage + at + incl - seg;
            uint64_t v[4];
            v[0] = r0[b].x | (uint64_t)r0[b].y << 32; v[1] = r0[b].z | (uint64_t)r0[b].w << 32;
            v[2] = r1[b].x | (uint64_t)r1[b].y << 32; v[3] = r1[b].z | (uint64_t)(r1[b].w & 0xFFFFFFu) << 32;
            const uint8_t* sp = t.str_bytes + (in_rec ? 0u : r0[b].x);
            if (!in_rec && n) {                                    // longer than the record: its first 32 bytes at once (the string arena has slack behind its end)
                __builtin_memcpy(&v[0], sp, 16);
                __builtin_memcpy(&v[2], sp + 16, 16);
            }
            const uint32_t n32 = n < 32 ? n : 32, nfull = n32 >> 3;
            for (uint32_t k = 0; k < 4; ++k)
                if (k < nfull) __builtin_memcpy(dst + 8 * k, &v[k], 8);
            if (n32 & 7) {                                           // the last, partial word in pieces of 4, 2 and 1 bytes
                uint64_t x = nfull == 0 ? v[0] : (nfull == 1 ? v[1] : (nfull == 2 ? v[2] : v[3]));
                uint32_t o = 8 * nfull;
                if (n32 & 4) { uint32_t y = (uint32_t)x; __builtin_memcpy(dst + o, &y, 4); x >>= 32; o += 4; }
                if (n32 & 2) { uint16_t y = (uint16_t)x; __builtin_memcpy(dst + o, &y, 2); x >>= 16; o += 2; }
                if (n32 & 1) dst[o] = (uint8_t)x;
            }
            for (uint32_t done = 32; done < n; ++done) dst[done] = sp[done];
            if (live) {
                if (last) { for (uint32_t q = 0; q < t.newline_len; ++q) dst[n + q] = t.newline[q]; }
                else { dst[n] = ','; dst[n + 1] = ' '; }
            }
            at += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the staging area is this wave's own: no wait for global traffic
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint8_t* gbase = captions + out0 - head;                      // 16-byte aligned
        const uint32_t end = head + total, nchunks = (end + 15) >> 4;
#if TP_EMIT_DIAG == 1
        if (total == 0xFFFFFFFFu)                                      // no stores (diagnostic build)
#endif
        for (uint32_t c = lane; c < nchunks; c += 64) {
            if (16 * c >= head && 16 * c + 16 <= end) *(uint4*)(gbase + 16 * c) = *(const uint4*)(stage + 16 * c);
        }
        if (lane < 32) {                                              // the partial first and last pieces, one byte per lane
            const uint32_t c = lane < 16 ? 0u : nchunks - 1;
            const uint32_t off = 16 * c + (lane & 15);
            const bool partial = !(16 * c >= head && 16 * c + 16 <= end) && (lane < 16 || nchunks > 1);
            if (partial && off >= head && off < end) gbase[off] = stage[off];
        }
        __builtin_amdgcn_wave_barrier();                              // the staging bytes are read before the next run's are written
    }
}

// Lanes 0 .. RUN_FILES load the group's sizes, kinds and offsets (one instruction each for the whole group, one group ahead).
#ifndef TP_EMIT_RUNS_WAVES_PER_SIMD
#define TP_EMIT_RUNS_WAVES_PER_SIMD 4      // 128 registers: the records of five lane blocks stay in registers (measured: 0.79 ms against 0.82 at five waves with spills)
#endif
// k_emit_runs_hist: the same with the tag histogram of these files counted on the way (the ids are in registers), one block
// of 16 wavefronts per CU: its LDS holds the staging areas and 16-bit counters for HISTX_IDS ids, so the launch takes it when
// the tables have no more strings than that and a block sees fewer than 65536 files; k_hist then only counts the general
// kernel's files.
constexpr int EMITH_WAVES = 16;
constexpr uint32_t HISTX_IDS = 32768;
constexpr uint32_t RUNH_STAGE = 6064;                           // 16 x (6064 + 48) + 64 KB of counters = 160 KB less 512 bytes
template <bool HIST, int WAVES, uint32_t STAGE_CAP>
__device__ __forceinline__ void emit_runs_body(const DevTables& t, const EmitArgs& a) {
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[WAVES][STAGE_CAP + 48];
    __shared__ unsigned long long blk_cnt[8];
    __shared__ uint32_t bins[HIST ? HISTX_IDS / 2 : 1];
    __shared__ uint32_t learned;
    if constexpr (HIST) {
        for (uint32_t i = threadIdx.x; i < HISTX_IDS / 2; i += blockDim.x) bins[i] = 0;
        if (threadIdx.x == 0) learned = 0;
        __syncthreads();
    }
    uint32_t n_learned = 0;
    const uint32_t lane = threadIdx.x & 63;
    uint8_t* stage = stage_all[wave_in_block()];
    const uint64_t wave_global = wave_index();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t n_groups = (a.file_end - a.file0 + RUN_FILES - 1) / RUN_FILES;
    uint32_t cnt[4] = {0, 0, 0, 0};                                // lane j < RUN_FILES counts the files number j of its groups: counter k in the
                                                                   // 16 bits k & 1 of word k >> 1 (a wavefront sees far fewer than 65536 groups)

    struct Meta { uint32_t n, slot, fl; uint64_t off; };
    auto load_meta = [&](uint64_t g, Meta& m) {
        m.n = 0; m.slot = 0; m.fl = 0; m.off = 0;
        if (g >= n_groups) return;
        const uint64_t f = a.file0 + g * RUN_FILES + lane;
        if (lane < (uint32_t)RUN_FILES && f < a.file_end) { m.n = a.n_final[f]; m.slot = a.slot_of[f]; m.fl = a.verdicts[f].flags; }
        if (lane <= (uint32_t)RUN_FILES && f <= a.file_end) m.off = a.caption_offsets[f];
    };
    auto lane_of = [](uint32_t v, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); };
    Meta m0, m1;
    load_meta(wave_global, m0);
    for (uint64_t g = wave_global; g < n_groups; g += n_waves, m0 = m1) {
        load_meta(g + n_waves, m1);
        const uint64_t row0 = g * RUN_FILES;                         // the group's first file, relative to file0
        const uint64_t left = a.file_end - a.file0 - row0;
        const uint32_t nfile = left < (uint64_t)RUN_FILES ? (uint32_t)left : (uint32_t)RUN_FILES;
        const bool fast = lane < nfile && m0.slot == SLOT_FAST;
        const uint32_t fastmask = (uint32_t)__ballot(fast ? 1 : 0);
        if (fast) {
            const uint32_t fl = m0.fl;
            cnt[0] += 1u | (uint32_t)((fl & TP_F_LOW) != 0) << 16;
            cnt[1] += (uint32_t)((fl & TP_F_ELIGIBLE) != 0) | (uint32_t)((fl & TP_F_FOOTJOB) != 0) << 16;
            cnt[2] += (uint32_t)((fl & TP_F_FALLBACK) != 0) | (uint32_t)((fl & TP_F_LINE_REJECTED) != 0) << 16;
            cnt[3] += (uint32_t)((fl & TP_F_UNSUPPORTED) != 0) | (uint32_t)((fl & TP_F_SAFE_REJECT) != 0) << 16;
        }
        if (fastmask == 0) continue;
        uint32_t n_[RUN_FILES];
        uint64_t off_[RUN_FILES + 1];
        for (int j = 0; j < RUN_FILES; ++j) n_[j] = lane_of(m0.n, j);
        for (int j = 0; j <= RUN_FILES; ++j) off_[j] = lane_of((uint32_t)m0.off, j) | (uint64_t)lane_of((uint32_t)(m0.off >> 32), j) << 32;
        emit_group<HIST, STAGE_CAP>(t, a.fin32 + row0 * FAST_FIN_MAX, a.captions, stage, lane, nfile, fastmask, n_, off_, bins, &n_learned);
    }
    if constexpr (HIST) {
        if (n_learned) atomicAdd(&learned, n_learned);
        __syncthreads();
        if (threadIdx.x == 0 && learned) atomicAdd((unsigned long long*)&a.histogram[t.n_public], (unsigned long long)learned);
        if (a.hist_parts) {                                           // k_hist_sum adds the blocks' counters up
            uint32_t* part = a.hist_parts + (size_t)blockIdx.x * HIST_PART_WORDS;
            for (uint32_t i = threadIdx.x; i < (t.n_public + 1) / 2; i += blockDim.x) part[i] = bins[i];
        } else
        for (uint32_t i = threadIdx.x; i < (t.n_public + 1) / 2; i += blockDim.x) {
            const uint32_t v = bins[i];
            if (v & 0xFFFFu) atomicAdd((unsigned long long*)&a.histogram[2 * i], (unsigned long long)(v & 0xFFFFu));
            if (v >> 16) atomicAdd((unsigned long long*)&a.histogram[2 * i + 1], (unsigned long long)(v >> 16));
        }
    }
    if (a.counters) {                                                     // added up per block in LDS, then one global atomic per block and counter
        if (threadIdx.x < 8) blk_cnt[threadIdx.x] = 0;
        __syncthreads();
        if (lane < (uint32_t)RUN_FILES) for (int k = 0; k < 8; ++k) { const uint32_t v = (cnt[k >> 1] >> (16 * (k & 1))) & 0xFFFFu; if (v) atomicAdd(&blk_cnt[k], (unsigned long long)v); }
        __syncthreads();
        if (threadIdx.x < 8 && blk_cnt[threadIdx.x]) atomicAdd((unsigned long long*)a.counters + threadIdx.x, blk_cnt[threadIdx.x]);
    }
}
__global__ __launch_bounds__(64 * EMIT_WAVES) __attribute__((amdgpu_waves_per_eu(TP_EMIT_RUNS_WAVES_PER_SIMD))) void k_emit_runs(DevTables t, EmitArgs a) {
    emit_runs_body<false, EMIT_WAVES, RUN_STAGE>(t, a);
}
__global__ __launch_bounds__(64 * EMITH_WAVES) void k_emit_runs_hist(DevTables t, EmitArgs a) { emit_runs_body<true, EMITH_WAVES, RUNH_STAGE>(t, a); }

// The back half, the caption offsets and the caption bytes in one kernel (the plain run: no detail arrays, no audit lists).
// Files are handed out in tiles of BX_TILE consecutive files, in the order blocks ask for them (a ticket counter), BX_FILES
// consecutive files to each of the block's wavefronts.  A wavefront first runs the back half for those of its files the
// front half passed (final tag ids to the file's row of the id array, as k_lean_b leaves them: the histogram kernel counts
// them there, and the rows are in the L2 cache when the same wavefront reads them again for the text); files an earlier
// stage has finished bring their sizes.  The block adds up its files' caption bytes and publishes the sum in the tile's state
// word (flag AGG).  Where the tile's captions begin is the sum over all tiles before it: a decoupled look-back over the state
// words, nearest tile first, adding up aggregates until a tile is met whose inclusive prefix is known (flag PRE), after which
// the tile's own inclusive prefix is published.  A tile waits for nothing but state words of tiles handed out before it, which
// their blocks publish without waiting for anyone.  The look-back of a tile is put off until the block has done the back
// halves of its NEXT tile: by then the tiles just before it have long published their sums, so the walk finds every word it
// needs at the first try (tiles that run side by side all finish their back halves at about the same time; looking back
// right away, each would wait for its neighbours and walk through up to a thousand of them while its block idles).
// Then every wavefront writes its files' caption offsets and their text, RUN_FILES files to a run as k_emit_runs does.
// What this saves over k_lean_b + scan + k_emit_runs: the latency-bound text phase runs beside other wavefronts' back
// halves, which are bound by instruction issue; the sizes, kinds and offsets are not loaded again; three scan kernels go.
// A file the back half declines is put on a.slow_list and counted in *a.slow_count with length 0: the host then finishes such
// files with the later stages and runs this kernel again (files finished here are then files "an earlier stage has finished").
#ifndef BX_RUNS_N
#define BX_RUNS_N 2
#endif
constexpr int BX_RUNS = BX_RUNS_N;
constexpr int BX_FILES = BX_RUNS * RUN_FILES;                 // files per wavefront and tile
constexpr int BX_TILE = BX_FILES * LEAN_WAVES_PER_BLOCK;      // files per tile
constexpr unsigned long long BX_AGG = 1ull << 62, BX_PRE = 2ull << 62, BX_VAL = (1ull << 62) - 1;
constexpr int BX_WINDOWS = 4;                                  // state words per lane and look-back step
#ifndef TP_LEAN_BX_WAVES_PER_SIMD
#define TP_LEAN_BX_WAVES_PER_SIMD 4
#endif
union alignas(16) BxWaveLds {
    LeanLds lean;                                               // while the wavefront's back halves run
    uint8_t stage[RUN_STAGE + 48];                              // while its captions are put together
};
struct BxMeta { uint32_t n, len, fl, fast; };                  // one file: tags, caption bytes, verdict flags, whether its caption is written here

// The caption bytes of all files before tile `tile` (> 0) of this launch.
__device__ __forceinline__ unsigned long long bx_look_back(const unsigned long long* state, uint64_t tile, uint32_t lane) {
    long long look = (long long)tile - 1;
    unsigned long long part = 0;
    for (;;) {
        unsigned long long v[BX_WINDOWS];
        TP_UNROLL for (int q = 0; q < BX_WINDOWS; ++q) {
            const long long idx = look - 64 * q - (long long)lane;   // lane 0 of window 0: the nearest tile
            v[q] = BX_PRE;                                          // before the first tile: nothing, and nothing further back
            if (idx >= 0) v[q] = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int q = 0;
        bool done = false, again = false;
        for (; q < BX_WINDOWS; ++q) {
            const uint64_t m_pre = __ballot((v[q] >> 62) == 2 ? 1 : 0), m_none = __ballot((v[q] >> 62) == 0 ? 1 : 0);
            const uint32_t first_pre = m_pre ? (uint32_t)__builtin_ctzll(m_pre) : 64u;
            const uint64_t upto = first_pre >= 63u ? ~0ull : ((1ull << (first_pre + 1)) - 1ull);   // the lanes whose values count
            if (m_none & upto) { again = true; break; }           // a tile in between has not published yet: this window once more
            if ((upto >> lane) & 1ull) part += v[q] & BX_VAL;
            if (first_pre < 64u) { done = true; break; }
        }
        if (done) break;
        look -= 64 * q;
        if (again) __builtin_amdgcn_s_sleep(4);
    }
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
    return part;
}

__global__ __launch_bounds__(64 * LEAN_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(TP_LEAN_BX_WAVES_PER_SIMD))) void k_lean_bx(FastArgs a) {
    __shared__ BxWaveLds lds[LEAN_WAVES_PER_BLOCK];
    __shared__ LeanShared shared;
    __shared__ DeclineBuf dbuf[LEAN_WAVES_PER_BLOCK];
    __shared__ uint32_t wave_total[2][LEAN_WAVES_PER_BLOCK];   // by tile parity: a tile's sums are read while the next tile's are made
    __shared__ BxMeta meta[2][LEAN_WAVES_PER_BLOCK][BX_FILES];
    __shared__ uint32_t tile_s;
    __shared__ unsigned long long blk_cnt[8];
    extern __shared__ uint32_t rc_lds[];                        // the rule cache, t.rc_words words (dynamic LDS)
    ConstTables& t = *(ConstTables*)(uintptr_t)a.tables;
    for (uint32_t i = threadIdx.x; i < t.rc_words; i += blockDim.x) rc_lds[i] = t.rc[i];
    lean_shared_init(t, shared, threadIdx.x, blockDim.x);
    if (threadIdx.x < 8) blk_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t n_declined = 0;
    const uint32_t wv = wave_in_block();
    LeanLds& L = lds[wv].lean;
    DevWave w;
    const uint32_t lane = w.lane();
    const uint64_t n_items = a.file_end - a.file0;
    const uint64_t n_tiles = (n_items + BX_TILE - 1) / BX_TILE;
    auto lane_of = [](uint32_t v, uint32_t j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)j)); };

    // what the wavefront knows of a file before its turn: kind, sizes, and - for a file of the front half - its candidates' ids
    struct Peek { uint32_t slot, n, len, id; };
    auto peek = [&](uint64_t i, Peek& p) {
        p.slot = SLOT_NONE; p.n = 0; p.len = 0; p.id = NO_ID;
        if (i < n_items) {
            const uint64_t f = a.file0 + i;
            p.slot = a.slot_of[f];
            p.n = a.n_final[f];
            if (p.slot == SLOT_LEAN) p.id = a.fin32[i * FAST_FIN_MAX + lane];      // entries behind the count are never used
            else if (p.slot != SLOT_NONE) p.len = a.cap_len[f];
        }
    };
    auto fetch_records = [&](const Peek& p) {                    // p is the same in every lane but for id
        if (p.slot == SLOT_LEAN && lane < p.n) {
            const uint32_t* src = t.crec + 8 * (size_t)p.id;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)&L.cr[0][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4), (__attribute__((address_space(3))) void*)&L.cr[1][0], 16, 0, 0);
        }
    };

    bool pend = false;                                           // the tile before: back halves done, sums published, text still to write
    uint64_t pend_tile = 0;
    for (uint32_t it = 0;; ++it) {
#ifdef BX_X_STATIC
        const uint64_t tile = blockIdx.x + (uint64_t)it * gridDim.x;
#else
        if (threadIdx.x == 0) tile_s = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const uint64_t tile = tile_s;
#endif
        const bool have = tile < n_tiles;
        if (!have && !pend) break;
        const uint32_t par = it & 1u;
        if (have) {
            // ---- the wavefront's files of this tile: back halves, sizes ---------------------------------------------------------------
            const uint64_t i0 = tile * BX_TILE + (uint64_t)wv * BX_FILES;    // the wavefront's first file, relative to file0
            uint32_t tile_len = 0;
            Peek p0, p1;
            peek(i0, p0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fetch_records(p0);
            for (uint32_t j = 0; j < (uint32_t)BX_FILES; ++j) {
                const uint64_t i = i0 + j;
                const uint64_t f = a.file0 + i;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this file's records are in L.cr
                if (j + 1 < (uint32_t)BX_FILES) peek(i + 1, p1); else { p1.slot = SLOT_NONE; p1.n = 0; p1.len = 0; p1.id = NO_ID; }
                bool fetched = false;
                auto records_free = [&]() {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // what this file read out of L.cr has arrived, and the next file's ids
                    fetch_records(p1);
                    fetched = true;
                };
                BxMeta m{0, 0, 0, 0};
                if (p0.slot == SLOT_LEAN) {
                    FastOut out;
                    const bool done = lean_back<false, true>(w, t, L, shared, rc_lds, lane < p0.n ? p0.id : NO_ID, p0.n, out, a.fin32 + i * FAST_FIN_MAX, nullptr, nullptr, records_free);
                    if (done) {
                        m.n = out.n_final; m.len = out.caption_len; m.fl = out.verdict.flags; m.fast = 1;
                        if (lane == 0) {
                            a.verdicts[f] = out.verdict;
                            a.n_final[f] = m.n;
                            a.cap_len[f] = m.len;
                            a.slot_of[f] = SLOT_FAST;
                        }
                    } else {
                        if (lane == 0) a.slot_of[f] = SLOT_NONE;
                        decline_push(dbuf[wv], n_declined, (uint32_t)i, a.slow_list, a.slow_count);
                    }
                    w.sync();
                } else if (p0.slot == SLOT_FAST) {
                    m.n = p0.n; m.len = p0.len; m.fast = 1;
                    m.fl = a.verdicts[f].flags;
                } else {
                    m.len = p0.len;                               // a file of the general kernel: its caption is written from its workspace later
                }
                if (!fetched) records_free();
                if (lane == 0) meta[par][wv][j] = m;
                tile_len += m.len;
                p0 = p1;
            }
            if (lane == 0) wave_total[par][wv] = tile_len;
        }
#ifndef BX_X_NOBAR
        __syncthreads();
#endif
        if (have && wv == 0 && lane == 0) {                      // the tile's sum; the first tile knows its prefix already
            unsigned long long total = 0;
            for (int k = 0; k < LEAN_WAVES_PER_BLOCK; ++k) total += wave_total[par][k];
            __hip_atomic_store(&a.tile_state[tile], tile == 0 ? (BX_PRE | (*a.base_total + total)) : (BX_AGG | total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (pend) {
            // ---- the tile before: where its captions begin, then offsets and text of the wavefront's files ------------------------------
            const uint32_t pp = par ^ 1u;
#ifdef BX_X_NOLOOK
            unsigned long long base = *a.base_total + pend_tile * 1000;
#else
            unsigned long long base = pend_tile == 0 ? *a.base_total : bx_look_back(a.tile_state, pend_tile, lane);
#endif
            if (wv == 0 && lane == 0 && pend_tile != 0) {
                unsigned long long total = 0;
                for (int k = 0; k < LEAN_WAVES_PER_BLOCK; ++k) total += wave_total[pp][k];
                __hip_atomic_store(&a.tile_state[pend_tile], BX_PRE | (base + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (uint32_t k = 0; k < wv; ++k) base += wave_total[pp][k];
            const uint64_t i0 = pend_tile * BX_TILE + (uint64_t)wv * BX_FILES;
            BxMeta m{0, 0, 0, 0};
            if (lane < (uint32_t)BX_FILES) m = meta[pp][wv][lane];
            const uint32_t incl = w.scan_incl(m.len);
            const unsigned long long off = base + (incl - m.len);
            if (lane < (uint32_t)BX_FILES && i0 + lane < n_items) {
                a.caption_offsets[a.file0 + i0 + lane] = off;
                if (i0 + lane + 1 == n_items) { a.caption_offsets[a.file_end] = off + m.len; *a.total_out = off + m.len; }
                if (m.fast && a.bx_counters) {
                    const uint32_t fl = m.fl;
                    atomicAdd(&blk_cnt[0], 1ull);
                    if (fl & TP_F_LOW) atomicAdd(&blk_cnt[1], 1ull);
                    if (fl & TP_F_ELIGIBLE) atomicAdd(&blk_cnt[2], 1ull);
                    if (fl & TP_F_FOOTJOB) atomicAdd(&blk_cnt[3], 1ull);
                    if (fl & TP_F_FALLBACK) atomicAdd(&blk_cnt[4], 1ull);
                    if (fl & TP_F_LINE_REJECTED) atomicAdd(&blk_cnt[5], 1ull);
                    if (fl & TP_F_UNSUPPORTED) atomicAdd(&blk_cnt[6], 1ull);
                    if (fl & TP_F_SAFE_REJECT) atomicAdd(&blk_cnt[7], 1ull);
                }
            }
            const uint32_t fastbits = (uint32_t)w.ballot(m.fast != 0);
            for (uint32_t r = 0; r < (uint32_t)BX_RUNS; ++r) {
                const uint64_t row0 = i0 + (uint64_t)RUN_FILES * r;
                if (row0 >= n_items) break;
                const uint64_t left = n_items - row0;
                const uint32_t nfile = left < (uint64_t)RUN_FILES ? (uint32_t)left : (uint32_t)RUN_FILES;
                const uint32_t fastmask = (fastbits >> (RUN_FILES * r)) & ((1u << nfile) - 1u);
                if (fastmask == 0) continue;
                uint32_t n_[RUN_FILES];
                uint64_t off_[RUN_FILES + 1];
                for (uint32_t q = 0; q < (uint32_t)RUN_FILES; ++q) {
                    n_[q] = lane_of(m.n, RUN_FILES * r + q);
                    off_[q] = lane_of((uint32_t)off, RUN_FILES * r + q) | (uint64_t)lane_of((uint32_t)(off >> 32), RUN_FILES * r + q) << 32;
                }
                off_[RUN_FILES] = off_[RUN_FILES - 1] + lane_of(m.len, RUN_FILES * r + RUN_FILES - 1);
                for (uint32_t q = nfile; q < (uint32_t)RUN_FILES; ++q) { n_[q] = 0; off_[q + 1] = off_[nfile]; }     // (files behind the range's end)
                if (off_[nfile] > a.caption_capacity) continue;   // no room: the host learns the total and reports it
#ifdef BX_X_NOEMIT
                if (a.caption_capacity == 12345)
#endif
                emit_group(t, a.fin32 + row0 * FAST_FIN_MAX, a.captions, lds[wv].stage, lane, nfile, fastmask, n_, off_);
            }
            w.sync();
        }
        pend = have;
        pend_tile = tile;
    }
    decline_flush(dbuf[wv], n_declined, a.slow_list, a.slow_count);
    if (a.bx_counters) {
        __syncthreads();
        if (threadIdx.x < 8 && blk_cnt[threadIdx.x]) atomicAdd((unsigned long long*)a.bx_counters + threadIdx.x, blk_cnt[threadIdx.x]);
    }
}
__global__ __launch_bounds__(64) void k_add_counters(tp_counters* dst, const tp_counters* src) {
    if (threadIdx.x < 8) { const unsigned long long v = ((const unsigned long long*)src)[threadIdx.x]; if (v) atomicAdd((unsigned long long*)dst + threadIdx.x, v); }
}

// Tag histogram of the wave-kernel files: ids id0 .. id0 + HIST_IDS are counted in LDS (a tag occurs at most once per file,
// so a block's 16-bit counters cannot overflow before HIST_FILES_PER_FLUSH files), then added to the global histogram.
// Wider id ranges take further launches.  One wavefront per file.
// general != 0: the files of the general kernel whose ids its emit kernel left in the same rows are counted too; 2: only those
// (the caption writer has counted the wave kernels' files).
__global__ __launch_bounds__(1024) void k_hist(DevTables t, EmitArgs a, uint32_t id0, int general) {
    __shared__ uint32_t bins[HIST_IDS / 2];
    static_assert(HIST_IDS / 2 == HIST_PART_WORDS && HISTX_IDS / 2 <= HIST_PART_WORDS, "a block's counters fit its part of EmitArgs::hist_parts");
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint32_t span = t.n_public - id0 < HIST_IDS ? t.n_public - id0 : HIST_IDS;
    __shared__ uint32_t learned;                                // tags that are learned strings: counted with the tags in no table (first pass only)
    if (threadIdx.x == 0) learned = 0;
    const uint64_t n = a.file_end - a.file0;
    const uint64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t b0 = a.file0 + per_block * blockIdx.x, b1 = b0 + per_block < a.file_end ? b0 + per_block : a.file_end;
    if (a.hist_parts && per_block <= HIST_FILES_PER_FLUSH && b0 >= b1)      // no files: the part still takes part in the sum
        for (uint32_t i = threadIdx.x; i < (span + 1) / 2; i += blockDim.x) a.hist_parts[(size_t)blockIdx.x * HIST_PART_WORDS + i] = 0;
    for (uint64_t c0 = b0; c0 < b1; c0 += HIST_FILES_PER_FLUSH) {
        const uint64_t c1 = c0 + HIST_FILES_PER_FLUSH < b1 ? c0 + HIST_FILES_PER_FLUSH : b1;
        for (uint32_t i = threadIdx.x; i < HIST_IDS / 2; i += blockDim.x) bins[i] = 0;
        __syncthreads();
        // eight files per step, so that their loads overlap; a lane takes two neighbouring ids of each (only the part of a
        // row that holds ids is fetched); the sizes that say which part are loaded one step ahead
        constexpr int NF = 8;
        auto sizes = [&](uint64_t f, uint32_t (&nf)[NF]) {
            TP_UNROLL for (int k = 0; k < NF; ++k) {
                const uint64_t g = f + (uint64_t)k * waves;
                nf[k] = 0;
                if (g < c1) {
                    const uint32_t n = a.n_final[g];
                    const bool fast = a.slot_of[g] == SLOT_FAST;
                    if ((fast && general != 2) || (!fast && general && n <= FAST_FIN_MAX)) nf[k] = n;
                }
            }
        };
        // sizes two steps ahead, rows one step ahead: a step's loads are on their way while the step before counts
        uint32_t nf[NF], nf1[NF], nf2[NF];
        uint32_t n_learned = 0;
        auto rows = [&](uint64_t f, const uint32_t (&n_)[NF], uint2 (&x)[NF]) {
            TP_UNROLL for (int k = 0; k < NF; ++k) {
                x[k] = make_uint2(0, 0);
                if (2 * lane < n_[k]) x[k] = *(const uint2*)(a.fin32 + (f + (uint64_t)k * waves - a.file0) * FAST_FIN_MAX + 2 * lane);
            }
        };
        uint2 x[NF], x1[NF];
        sizes(c0 + wave, nf);
        sizes(c0 + wave + NF * waves, nf1);
        rows(c0 + wave, nf, x);
        for (uint64_t f = c0 + wave; f < c1; f += NF * waves) {
            sizes(f + 2 * NF * waves, nf2);
            rows(f + NF * waves, nf1, x1);
            auto count = [&](uint32_t id, bool live) {
                const uint32_t k = id - id0;
#ifdef HIST_X_NOATOM
                if (live && k == span + 12345)
#else
                if (live && k < span)
#endif
                    atomicAdd(&bins[k >> 1], 1u << (16 * (k & 1)));
                n_learned += (uint32_t)(live && id0 == 0 && id >= t.n_public && id < t.n_strings);
            };
            TP_UNROLL for (int k = 0; k < NF; ++k) { count(x[k].x, 2 * lane < nf[k]); count(x[k].y, 2 * lane + 1 < nf[k]); }
            TP_UNROLL for (int k = 0; k < NF; ++k) { nf[k] = nf1[k]; nf1[k] = nf2[k]; x[k] = x1[k]; }
        }
        if (n_learned) atomicAdd(&learned, n_learned);
        __syncthreads();
        if (threadIdx.x == 0 && learned) { atomicAdd((unsigned long long*)&a.histogram[t.n_public], (unsigned long long)learned); learned = 0; }
        if (a.hist_parts && per_block <= HIST_FILES_PER_FLUSH) {      // (one flush per block: k_hist_sum adds the blocks' counters up)
            uint32_t* part = a.hist_parts + (size_t)blockIdx.x * HIST_PART_WORDS;
            for (uint32_t i = threadIdx.x; i < (span + 1) / 2; i += blockDim.x) part[i] = bins[i];
        } else
#ifdef HIST_X_NOFLUSH
        if (span == 12345)
#endif
        for (uint32_t i = threadIdx.x; i < (span + 1) / 2; i += blockDim.x) {
            const uint32_t v = bins[i];
            if (v & 0xFFFFu) atomicAdd((unsigned long long*)&a.histogram[id0 + 2 * i], (unsigned long long)(v & 0xFFFFu));
            if (v >> 16) atomicAdd((unsigned long long*)&a.histogram[id0 + 2 * i + 1], (unsigned long long)(v >> 16));
        }
        __syncthreads();
    }
}

// The blocks' counters (EmitArgs::hist_parts) added up into the histogram: one thread per word of two counters.
__global__ __launch_bounds__(256) void k_hist_sum(const uint32_t* parts, uint32_t n_blocks, uint32_t words, uint64_t* histogram) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    uint32_t lo = 0, hi = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) { const uint32_t v = parts[(size_t)b * HIST_PART_WORDS + i]; lo += v & 0xFFFFu; hi += v >> 16; }
    if (lo) histogram[2 * i] += lo;
    if (hi) histogram[2 * i + 1] += hi;
}

// Files of the general kernel (a.item_list, or every file of the range when it is null): tags may be strings outside the
// tables, kept in the file's workspace.  One wavefront per file, 64 tags at a time: every lane puts its tag (and separator)
// into an LDS staging area, eight bytes per load, then the wavefront stores the area with consecutive lanes on consecutive
// bytes.  A group of tags too long for the area is stored lane by lane.
__global__ __launch_bounds__(256) void k_emit_general(DevTables t, EmitArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[4][GEN_STAGE];
    uint8_t* stage = stage_all[wave_in_block()];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave_global = wave_index();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    EmitCounts cnt;
    uint64_t c_other = 0;
    for (uint64_t i = wave_global; i < a.n_items; i += n_waves) {
        const uint64_t f = a.file0 + (a.item_list ? a.item_list[i] : i);
        const uint32_t nfin = a.n_final[f];
        const uint64_t out0 = a.caption_offsets[f];
        const uint32_t total = (uint32_t)(a.caption_offsets[f + 1] - out0);
        if (lane == 0) cnt.add(a.verdicts[f].flags);
        if (total == 0) continue;   // unsupported file: nothing is written
        const uint32_t slot = a.slot_of[f];
        const uint64_t w = slot >> 6;
        WsLayout L = ws_layout(a.wave_T[w], a.wave_LB[w], t.final_cap, t.evidence_grow, t.lit_words);
        FileWs ws = ws_bind(a.workspace + a.wave_off[w], L, slot & 63, a.fpw);
        uint32_t running = 0;
        const bool ids_out = a.fin32 != nullptr && nfin <= FAST_FIN_MAX;
        for (uint32_t k0 = 0; k0 < nfin; k0 += 64) {
            uint32_t k = k0 + lane;
            bool live = k < nfin;
            H h = 0;
            if (live) h = ws.fin[k];
            const uint8_t* src = nullptr;
            uint32_t len = 0;
            if (live) {
                if (h_is_dyn(h)) { src = ws.norm + h_off(h); len = h_len(h); }
                else { uint32_t id = h_id(h); uint32_t o = t.str_off[id]; src = t.str_bytes + o; len = t.str_off[id + 1] - o; }
            }
            bool sep = live && (k + 1 < nfin);
            uint32_t seg = len + (sep ? 2u : 0u);
            uint32_t incl = wave_scan_incl(seg);
            const uint32_t group = __shfl(incl, 63);
            uint8_t* gdst = a.captions + out0 + running;
            if (group <= GEN_STAGE) {
                if (live) {
                    uint8_t* dst = stage + (incl - seg);
                    put_bytes(dst, src, len);
                    if (sep) { dst[len] = ','; dst[len + 1] = ' '; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (uint32_t o = lane; o < group; o += 64) gdst[o] = stage[o];
                __builtin_amdgcn_wave_barrier();                    // the area is read before the next group's bytes are written
            } else if (live) {
                uint8_t* dst = gdst + (incl - seg);
                for (uint32_t b = 0; b < len; ++b) dst[b] = src[b];
                if (sep) { dst[len] = ','; dst[len + 1] = ' '; }
            }
            running += group;
            if (live && a.histogram) {
                if (h_is_dyn(h)) ++c_other;
                if (ids_out) a.fin32[(f - a.file0) * FAST_FIN_MAX + k] = h_is_dyn(h) ? NO_ID : h_id(h);   // counted by the histogram kernel
                else if (!h_is_dyn(h)) atomicAdd((unsigned long long*)&a.histogram[h_id(h)], 1ull);
            }
        }
        if (lane < t.newline_len) a.captions[out0 + running + lane] = t.newline[lane];
    }
    if (a.histogram) {
        for (int d = 32; d >= 1; d >>= 1) c_other += __shfl_xor(c_other, d);
        if (lane == 0 && c_other) atomicAdd((unsigned long long*)&a.histogram[t.n_public], (unsigned long long)c_other);
    }
    if (a.counters && lane == 0) cnt.flush(a.counters);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Working memory of the entry point that may not allocate: one array per GPU, part of the code object, held until the
// process ends.  In host code the array's name is only a handle for hipGetSymbolAddress; its address on a device is
// asked for once per device, with that device current, and kept.  Kernels get that address as an argument.
// ------------------------------------------------------------------------------------------------
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

hipError_t noalloc_scratch(int device, void** ptr, size_t* bytes) {
    static std::mutex mu;
    static void* known[NOALLOC_MAX_DEVICES];       // known[d]: address of `scratch` on device d, null until asked
    *ptr = nullptr;
    *bytes = NOALLOC_SCRATCH_BYTES;
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!known[device]) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) return e;
        if (current != device) return hipErrorInvalidDevice;      // the caller makes the device current first
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p || ((uintptr_t)p & 255)) return hipErrorInvalidSymbol;
        known[device] = p;
    }
    *ptr = known[device];
    return hipSuccess;
}

// The strings the lean stage noted (tp_lean.h, lean_note_unknown): every used slot's text goes to `out` and the slot is freed.
__global__ __launch_bounds__(256) void k_collect_unknown(unsigned long long* keys, const unsigned long long* ent, const uint8_t* text, uint32_t slots, uint8_t* out, uint32_t out_cap,
                                                         uint32_t* out_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots || keys[i] == 0) return;
    keys[i] = 0;
    const uint32_t n = (uint32_t)ent[i];
    const uint8_t* p = text + 32 * (size_t)i;                     // the copy lean_note_unknown made: nothing here points into a caller's input
    const uint32_t at = atomicAdd(out_count, 1u);
    if (at >= out_cap || n == 0 || n > 32) return;
    uint8_t* o = out + (size_t)at * 40;
    for (uint32_t k = 0; k < 32; ++k) o[k] = k < n ? p[k] : 0;
    *(uint32_t*)(o + 32) = n;
    *(uint32_t*)(o + 36) = 0;
}
hipError_t launch_collect_unknown(unsigned long long* keys, const unsigned long long* ent, const uint8_t* text, uint32_t slots, uint8_t* out, uint32_t out_cap, uint32_t* out_count,
                                  hipStream_t s) {
    hipLaunchKernelGGL(k_collect_unknown, dim3((slots + 255) / 256), dim3(256), 0, s, keys, ent, text, slots, out, out_cap, out_count);
    return hipGetLastError();
}

uint32_t fast_fin_stride() { return FAST_FIN_MAX; }
bool lean_ok(const DevTables& t) { return lean_tables_ok(t); }

// A grid-size override from the environment (testing aid): a positive number, anything else leaves the default.
static unsigned env_blocks(const char* name, unsigned dflt) {
    const char* e = getenv(name);
    if (!e) return dflt;
    const long v = strtol(e, nullptr, 10);
    return v >= 1 && v <= 0x7FFFFFFFl ? (unsigned)v : dflt;
}

hipError_t launch_fast(const DevTables& t, const FastArgs& a, hipStream_t s, hipEvent_t between, hipEvent_t after_front) {
    uint64_t n = a.file_end - a.file0;
    if (n == 0) return hipSuccess;
    uint64_t want = (n + FAST_WAVES_PER_BLOCK - 1) / FAST_WAVES_PER_BLOCK;
    // Wavefronts stride over the files.  Many more blocks than fit at once (measured best around 80 per CU): blocks that
    // finish early make room for others, which evens out the files' different costs.  Later stages usually find little to do.
    static const unsigned max_blocks = env_blocks("TAGPIPE_FAST_BLOCKS", 256u * 80u);
    // Behind the lean stage (a.tables is set) the stages find few files as a rule: smaller grids, so that blocks which find
    // nothing to do are not what these launches cost.
    static const unsigned after_lean1 = env_blocks("TAGPIPE_AFTER_LEAN_BLOCKS", 256u * 16u);
    static const unsigned after_lean2 = env_blocks("TAGPIPE_AFTER_LEAN_BLOCKS2", 512u);
    const bool after_lean = a.tables != nullptr && a.stage >= 1 && a.stage <= 3;
    const unsigned cap = a.grid_cap ? a.grid_cap : after_lean ? (a.stage == 1 ? after_lean1 : after_lean2) : (a.stage == 1 ? max_blocks : 256u * 8u);
    unsigned blocks = (unsigned)(want < cap ? want : cap);
    const bool tr = a.triage != 0;
    if (a.stage == 0) {
        // (measured on 1M c2 files, blocks: front / back ms: 1280: 0.89 / 1.47, 4096: 0.74 / 1.41, 20480: 0.71 / 1.33, 81920: 0.72 / 1.45)
        static const unsigned lean_blocks_a = env_blocks("TAGPIPE_LEAN_A_BLOCKS", 256u * 80u);
        static const unsigned lean_blocks_b = env_blocks("TAGPIPE_LEAN_B_BLOCKS", 256u * 80u);
        // about a dozen files per wavefront (shorter loops leave the wavefronts of a CU in different phases of their files; far
        // fewer files per wavefront and the per-block set-up and the per-wavefront hand-over of declined files begin to show)
        want = (n + 12 * LEAN_WAVES_PER_BLOCK - 1) / (12 * LEAN_WAVES_PER_BLOCK);
        if (want < 256u * 5u) want = (n + LEAN_WAVES_PER_BLOCK - 1) / LEAN_WAVES_PER_BLOCK < 256u * 5u ? (n + LEAN_WAVES_PER_BLOCK - 1) / LEAN_WAVES_PER_BLOCK : 256u * 5u;
        hipLaunchKernelGGL(k_lean_a, dim3((unsigned)(want < lean_blocks_a ? want : lean_blocks_a)), dim3(64 * LEAN_WAVES_PER_BLOCK), 0, s, a);
        if (between) { hipError_t e = hipEventRecord(between, s); if (e != hipSuccess) return e; }
        if (after_front) { hipError_t e = hipEventRecord(after_front, s); if (e != hipSuccess) return e; }
        if (a.front_only) return hipGetLastError();               // the fused back half follows later (launch_lean_bx)
        const dim3 grid_b((unsigned)(want < lean_blocks_b ? want : lean_blocks_b));
        if (a.fired || a.facet_choice || a.aud32) hipLaunchKernelGGL(k_lean_b_detail, grid_b, dim3(64 * LEAN_WAVES_PER_BLOCK), t.rc_words * sizeof(uint32_t), s, a);
        else hipLaunchKernelGGL(k_lean_b, grid_b, dim3(64 * LEAN_WAVES_PER_BLOCK), t.rc_words * sizeof(uint32_t), s, a);
    } else if (a.stage == 4) hipLaunchKernelGGL(a.aud32 ? k_scored_audit : k_scored, dim3((unsigned)(want < 256u * 16u ? want : 256u * 16u)), dim3(64 * FAST_WAVES_PER_BLOCK), t.rc_words * sizeof(uint32_t), s, t, a);
    else if (a.stage == 3) {
        // as many one-wavefront blocks as the CUs hold at once for the size's LDS (the grid strides over the list)
        const unsigned per_cu = a.long_size == 0 ? 9u : a.long_size == 1 ? 5u : 3u;
        unsigned blocks3 = (unsigned)(n < 256u * per_cu ? n : 256u * per_cu);
        if (a.grid_cap && a.grid_cap < blocks3) blocks3 = a.grid_cap;
        const dim3 grid(blocks3);
        const size_t dyn = 0;
        if (a.long_size == 0) hipLaunchKernelGGL(a.aud32 ? k_long256_audit : tr ? k_long256_triage : k_long256, grid, dim3(64), dyn, s, t, a);
        else if (a.long_size == 1) hipLaunchKernelGGL(a.aud32 ? k_long512_audit : tr ? k_long512_triage : k_long512, grid, dim3(64), dyn, s, t, a);
        else hipLaunchKernelGGL(a.aud32 ? k_long_audit : tr ? k_long_triage : k_long, grid, dim3(64), dyn, s, t, a);
    }
    else if (a.stage == 2) hipLaunchKernelGGL(a.aud32 ? k_fast2_audit : tr ? k_fast2_triage : k_fast2, dim3(blocks), dim3(64 * FAST_WAVES_PER_BLOCK), t.rc_words * sizeof(uint32_t), s, t, a);
    else hipLaunchKernelGGL(a.aud32 ? k_fast_audit : tr ? k_fast_triage : k_fast, dim3(blocks), dim3(64 * FAST_WAVES_PER_BLOCK), t.rc_words * sizeof(uint32_t), s, t, a);
    return hipGetLastError();
}
uint64_t lean_bx_tiles(uint64_t n_files) { return (n_files + BX_TILE - 1) / BX_TILE; }
hipError_t launch_lean_bx(const DevTables& t, const FastArgs& a, hipStream_t s) {
    const uint64_t n = a.file_end - a.file0;
    if (n == 0) return hipSuccess;
    const uint64_t tiles = lean_bx_tiles(n);
    // as many blocks as fit at once take tiles until none is left (a block that starts late simply finds fewer)
    static const unsigned per_cu = env_blocks("TAGPIPE_LEAN_BX_BLOCKS_PER_CU", (unsigned)TP_LEAN_BX_WAVES_PER_SIMD);
    const uint64_t cap = 256ull * per_cu;
    hipLaunchKernelGGL(k_lean_bx, dim3((unsigned)(tiles < cap ? tiles : cap)), dim3(64 * LEAN_WAVES_PER_BLOCK), t.rc_words * sizeof(uint32_t), s, a);
    return hipGetLastError();
}
hipError_t launch_add_counters(tp_counters* dst, const tp_counters* src, hipStream_t s) {
    hipLaunchKernelGGL(k_add_counters, dim3(1), dim3(64), 0, s, dst, src);
    return hipGetLastError();
}
hipError_t launch_wave_sizes(const DevTables& t, const uint8_t* arena, const uint64_t* offsets, const uint64_t* tag_begin, const uint32_t* item_list,
                             uint64_t file0, uint64_t n_items, uint32_t fpw, uint32_t* wave_T, uint32_t* wave_LB, uint64_t* wave_bytes, uint32_t* status, hipStream_t s) {
    if (n_items == 0) return hipSuccess;
    const uint64_t waves = (n_items + fpw - 1) / fpw;
    unsigned blocks = (unsigned)((waves * 64 + 255) / 256);
    hipLaunchKernelGGL(k_wave_sizes, dim3(blocks), dim3(256), 0, s, t, arena, offsets, tag_begin, item_list, file0, n_items, wave_T, wave_LB, wave_bytes, status, fpw);
    return hipGetLastError();
}
hipError_t launch_scan_u64(const uint64_t* in, uint64_t n, uint64_t* out, uint64_t* tile_sums, const uint64_t* base_ptr, uint64_t* total_out, hipStream_t s) {
    return scan_exclusive<uint64_t>(in, n, out, tile_sums, base_ptr, total_out, s);
}
hipError_t launch_scan_u32(const uint32_t* in, uint64_t n, uint64_t* out, uint64_t* tile_sums, const uint64_t* base_ptr, uint64_t* total_out, hipStream_t s) {
    return scan_exclusive<uint32_t>(in, n, out, tile_sums, base_ptr, total_out, s);
}
hipError_t launch_pipeline(const DevTables& t, const PipelineArgs& a, hipStream_t s) {
    if (a.n_items == 0) return hipSuccess;
    const uint64_t waves = (a.n_items + a.fpw - 1) / a.fpw;
    unsigned blocks = (unsigned)((waves * 64 + 255) / 256);
    hipLaunchKernelGGL(k_pipeline, dim3(blocks), dim3(256), 0, s, t, a);
    return hipGetLastError();
}
hipError_t launch_audit(const DevTables& t, const AuditArgs& a, hipStream_t s) {
    if (a.n_items == 0) return hipSuccess;
    uint64_t want = (a.n_items + 3) / 4;                    // four wavefronts per block
    // many more blocks than fit at once (five or six a CU): a grid of one or two rounds of blocks ends with most CUs idle
    static const unsigned cap = env_blocks("TAGPIPE_AUDIT_BLOCKS", 256u * 40u);
    unsigned blocks = (unsigned)(want < cap ? want : cap);
    hipLaunchKernelGGL(k_emit_audit, dim3(blocks), dim3(256), 0, s, t, a);
    return hipGetLastError();
}
hipError_t launch_emit(const DevTables& t, const EmitArgs& a, hipStream_t s) {
    uint64_t n = a.file_end - a.file0;
    if (n == 0) return hipSuccess;
    bool wave_files_counted = false;
    if (a.fin32 && a.fast_ran && !a.wave_files_written) {   // the wave-per-file kernel ran: its files first
        uint64_t want = (n + EMIT_WAVES - 1) / EMIT_WAVES;
        static const unsigned per_cu = env_blocks("TAGPIPE_EMIT_BLOCKS_PER_CU", (unsigned)TP_EMIT_WAVES_PER_SIMD);
        unsigned blocks = (unsigned)(want < 256 * per_cu ? want : 256 * per_cu);   // as many blocks (of four wavefronts) per CU as fit at once
        static const bool by_file = [] { const char* e = getenv("TAGPIPE_EMIT_BY_FILE"); return e && e[0] == '1'; }();   // testing aid: one file per wavefront
        if (by_file) hipLaunchKernelGGL(k_emit, dim3(blocks), dim3(64 * EMIT_WAVES), 0, s, t, a);
        else {
            want = ((n + RUN_FILES - 1) / RUN_FILES + EMIT_WAVES - 1) / EMIT_WAVES;
            static const unsigned runs_per_cu = env_blocks("TAGPIPE_EMIT_BLOCKS_PER_CU", (unsigned)TP_EMIT_RUNS_WAVES_PER_SIMD);
            blocks = (unsigned)(want < 256 * runs_per_cu ? want : 256 * runs_per_cu);
            static const bool no_hist = [] { const char* e = getenv("TAGPIPE_EMIT_HIST"); return e && e[0] == '0'; }();   // testing aid: the separate histogram kernel
            const uint64_t hwant = ((n + RUN_FILES - 1) / RUN_FILES + EMITH_WAVES - 1) / EMITH_WAVES;
            const unsigned hblocks = (unsigned)(hwant < 256 ? hwant : 256);
            if (!no_hist && a.histogram && t.n_public <= HISTX_IDS && n >= 4096 && n / hblocks + RUN_FILES * EMITH_WAVES < 65536) {
                EmitArgs own = a;
                own.hist_parts = nullptr;                              // (measured: with so few counters the blocks' own atomics are the cheaper flush)
                hipLaunchKernelGGL(k_emit_runs_hist, dim3(hblocks), dim3(64 * EMITH_WAVES), 0, s, t, own);
                wave_files_counted = true;
            } else
                hipLaunchKernelGGL(k_emit_runs, dim3(blocks), dim3(64 * EMIT_WAVES), 0, s, t, a);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (a.n_items) {
        uint64_t want = (a.n_items + 3) / 4;                // four wavefronts per block
        unsigned blocks = (unsigned)(want < 1024 ? want : 1024);
        hipLaunchKernelGGL(k_emit_general, dim3(blocks), dim3(256), 0, s, t, a);
    }
    if (a.fin32 && a.histogram) {                           // after both emit kernels: the general one leaves its files' tag ids in the same rows
        uint64_t hb = (n + 1023) / 1024;                      // a block is worth launching for a thousand files
        unsigned hblocks = (unsigned)(hb < 256 ? hb : 256);   // one per CU: the counters take most of its LDS
        const int general = a.n_items ? (wave_files_counted ? 2 : 1) : 0;
        if (!(wave_files_counted && !a.n_items))
            for (uint32_t id0 = 0; id0 < t.n_public; id0 += HIST_IDS) {
                hipLaunchKernelGGL(k_hist, dim3(hblocks), dim3(1024), 0, s, t, a, id0, general);
                if (a.hist_parts && (n + hblocks - 1) / hblocks <= HIST_FILES_PER_FLUSH) {
                    const uint32_t span = t.n_public - id0 < HIST_IDS ? t.n_public - id0 : HIST_IDS, words = (span + 1) / 2;
                    hipLaunchKernelGGL(k_hist_sum, dim3((words + 255) / 256), dim3(256), 0, s, a.hist_parts, hblocks, words, a.histogram + id0);
                }
            }
    }
    return hipGetLastError();
}

}  // namespace tp

#ifdef TP_PROF
extern "C" int tp_debug_prof(unsigned long long* out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(tp::g_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(tp::g_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
