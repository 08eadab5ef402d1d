// TEST-ONLY: lock-step emulation of one 64-lane wavefront on the host, so that tp_fast.h (the wave-per-file kernel
// body) can be checked against the oracle without a GPU.  Each lane is a ucontext fiber running the same function;
// a collective (ballot / shfl / sync) parks the lane until all 64 have arrived at it, then every lane reads the
// exchanged values.  A collective reached from different call sites by different lanes — i.e. executed in divergent
// control flow, which would be undefined on the GPU — aborts the test.
#pragma once
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace emu {

constexpr int LANES = 64;

struct Shared {
    ucontext_t main_ctx;
    ucontext_t ctx[LANES];
    std::vector<char> stacks;
    uint64_t slot[2][LANES];
    uint32_t kind[2][LANES];
    uint64_t gen_tag[2] = {~0ull, ~0ull};
    int arrived[2] = {0, 0};
    bool finished[LANES];
    int current = 0;
    std::function<void(int)> body;

    Shared() : stacks((size_t)LANES * (256 << 10)) {}

    static void trampoline(unsigned lo, unsigned hi, int lane) {
        Shared* s = (Shared*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
        s->body(lane);
        s->finished[lane] = true;
        swapcontext(&s->ctx[lane], &s->main_ctx);
    }
    void run(std::function<void(int)> fn) {
        body = std::move(fn);
        gen_tag[0] = gen_tag[1] = ~0ull;
        arrived[0] = arrived[1] = 0;
        for (int l = 0; l < LANES; ++l) {
            finished[l] = false;
            getcontext(&ctx[l]);
            ctx[l].uc_stack.ss_sp = stacks.data() + (size_t)l * (256 << 10);
            ctx[l].uc_stack.ss_size = 256 << 10;
            ctx[l].uc_link = &main_ctx;
            uintptr_t p = (uintptr_t)this;
            makecontext(&ctx[l], (void (*)())trampoline, 3, (unsigned)(p & 0xFFFFFFFFu), (unsigned)(p >> 32), l);
        }
        for (;;) {
            bool any = false;
            for (int l = 0; l < LANES; ++l) {
                if (finished[l]) continue;
                any = true;
                current = l;
                swapcontext(&main_ctx, &ctx[l]);
            }
            if (!any) break;
        }
    }
    // All lanes exchange one 64-bit value; returns after every lane has contributed.
    void exchange(int lane, uint64_t gen, uint32_t what, uint64_t value) {
        int b = (int)(gen & 1);
        if (gen_tag[b] != gen) { gen_tag[b] = gen; arrived[b] = 0; }
        slot[b][lane] = value;
        kind[b][lane] = what;
        ++arrived[b];
        while (arrived[b] < LANES) {
            for (int l = 0; l < LANES; ++l)
                if (finished[l]) { std::fprintf(stderr, "wave_emu: lane %d finished while lane %d waits at a collective\n", l, lane); std::abort(); }
            swapcontext(&ctx[lane], &main_ctx);
        }
        for (int l = 0; l < LANES; ++l)
            if (kind[b][l] != what) { std::fprintf(stderr, "wave_emu: divergent collective (lane %d kind %u, lane %d kind %u)\n", lane, what, l, kind[b][l]); std::abort(); }
    }
};

// The `W` policy of tp_fast.h, host flavour.
struct Wave {
    Shared* sh;
    uint32_t id;
    uint64_t gen = 0;
    uint32_t lane() const { return id; }
    uint64_t ballot(bool p) {
        sh->exchange((int)id, gen, 1, p ? 1 : 0);
        uint64_t m = 0;
        for (int l = 0; l < LANES; ++l) m |= (sh->slot[gen & 1][l] & 1ull) << l;
        ++gen;
        return m;
    }
    uint32_t shfl(uint32_t v, uint32_t src) {
        sh->exchange((int)id, gen, 2, v);
        uint32_t r = (uint32_t)sh->slot[gen & 1][src & 63];
        ++gen;
        return r;
    }
    void sync() {
        sh->exchange((int)id, gen, 3, 0);
        ++gen;
    }
    uint32_t lds_cas(uint32_t* p, uint32_t cmp, uint32_t val) { uint32_t old = *p; if (old == cmp) *p = val; return old; }
    void lds_min(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
    void lds_or(uint32_t* p, uint32_t v) { *p |= v; }
    void lds_min64(uint64_t* p, uint64_t v) { if (v < *p) *p = v; }
    void copy16(uint8_t* dst, const uint8_t* src) { std::memcpy(dst, src, 16); }
    void fill16(uint32_t* dst, uint32_t v) { for (int k = 0; k < 4; ++k) dst[k] = v; }
    void load16(const uint8_t* src, uint32_t (&x)[4]) { std::memcpy(x, src, 16); }
    void store16(uint8_t* dst, const uint32_t (&x)[4]) { std::memcpy(dst, x, 16); }
    void tick(int) {}
    uint32_t bcast(uint32_t v, uint32_t src) { return shfl(v, src); }
    uint32_t shift_up(uint32_t v, uint32_t fill) { uint32_t r = shfl(v, lane() == 0 ? 0 : lane() - 1); return lane() == 0 ? fill : r; }
    uint32_t scan_incl(uint32_t v) { for (uint32_t d = 1; d < 64; d <<= 1) { uint32_t o = shfl(v, lane() >= d ? lane() - d : lane()); if (lane() >= d) v += o; } return v; }
    uint32_t sum(uint32_t v) { for (uint32_t d = 32; d >= 1; d >>= 1) v += shfl(v, lane() ^ d); return v; }
    uint32_t or_all(uint32_t v) { for (uint32_t d = 32; d >= 1; d >>= 1) v |= shfl(v, lane() ^ d); return v; }
    uint32_t rank(uint64_t m) { return (uint32_t)__builtin_popcountll(m & (lane() == 0 ? 0ull : (~0ull >> (64 - lane())))); }
};

}  // namespace emu
