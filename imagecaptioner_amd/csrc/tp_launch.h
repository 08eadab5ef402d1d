// Kernel argument blocks and launcher entry points (tp_kernels.hip), used by the C ABI layer (tp_api.cpp).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/tagpipe.h"
#include "tp_common.h"

namespace tp {

constexpr uint32_t SLOT_FAST = 0xFFFFFFFFu;   // slot_of[file]: the file was finished by the wave-per-file kernel
constexpr uint32_t SLOT_LEAN = 0xFFFFFFFEu;   // ... the lean stage's front half has left the file's candidates for its back half
constexpr uint32_t SLOT_NONE = 0xFFFFFFFDu;   // ... no stage has finished the file yet (the lean stage's front half declined it)

// Wave-per-file kernel over files [file0, file_end) (line ingest only).
struct FastArgs {
    const uint8_t* arena;
    const uint64_t* offsets;
    uint64_t file0, file_end;
    tp_verdict* verdicts;
    uint32_t* n_final;
    uint32_t* cap_len;
    uint32_t* fin32;              // FAST_FIN_MAX ids per file, indexed by file - file0
    uint32_t* slot_of;            // per file: SLOT_FAST, or (set later) its position in the general kernel's item list
    uint32_t* slow_list;          // files (relative to file0) this launch declines, in no particular order
    uint32_t* slow_count;
    const uint32_t* item_list;    // null: first stage over the whole range; else the files (relative to file0) the stage before declined:
    const uint32_t* item_count;   //   *item_count of them (read on the device)
    // The lean stage's list has two ends: the front half's files from entry 0 upwards (slow_count), the back half's from entry
    // item_cap - 1 downwards (slow_count_b), so that a stage can take either end or both (null: none from that end)
    const uint32_t* item_count_b;
    uint32_t item_cap;
    uint32_t* slow_count_b;       // stage 0: the back half counts its declined files here and lists them from the end (null: with the front half's)
    // scored candidates (stage 4): offsets are tag offsets; the tags of file f are tag_begin[f] .. tag_begin[f + 1]
    const uint64_t* tag_begin;
    const double* scores;
    const uint8_t* sources;
    const uint64_t* secondary_begin;   // two replies per file to merge: first tag of the secondary reply (else null)
    int secondary_cap;
    const uint8_t* reply_flags;   // stage 4 in audit mode: per file, bit 0 = ParseJoyTags sorted the (primary) reply; a reply that kept its order is declined (null: every reply was sorted)
    int triage;                   // 1: existing-caption triage: stop after the candidates, verdict flags carry TP_F_TRIAGE_BAD
    int front_only;               // stage 0: launch the front half alone
    int stage;                    // 0: the lean stage (tp_lean.h); 1: up to 64 tokens per file; 2: up to 128; 3: long files (one wavefront per block, candidates in LDS)
    unsigned grid_cap;            // stages 1 to 3: at most so many blocks (0: the launch's own choice)
    int long_size;                // stage 3: 0 = up to 256 tokens in 4 KB, 1 = up to 512 in 8 KB, 2 = up to 1024 in 8 KB
    uint32_t* fired;
    uint16_t* facet_choice;
    const DevTables* tables;      // the lean stage: the table image in device memory (the other kernels take it as an argument)
    // audit mode (the lean stage only; null otherwise): per file AUD_STRIDE words of list entries (LeanAudit in tp_lean.h) and the
    // entries, bytes and review bits the general kernel reports in the same arrays (PipelineArgs)
    uint32_t* aud32;              // indexed by file - file0
    uint32_t* audit_cnt;          // AUDIT_COLS per file, indexed by file - file0
    uint32_t* audit_len;
    uint32_t* review_bits;        // indexed by file - file0
    // the fused back half (k_lean_bx: back half, caption offsets by look-back over tiles of files, caption bytes); null / 0 otherwise
    uint64_t* caption_offsets;    // indexed by file; the kernel writes entries file0 .. file_end
    uint8_t* captions;
    uint64_t caption_capacity;    // a tile whose captions would end behind this writes none (the total is still counted)
    const uint64_t* base_total;   // caption bytes of the files before file0 (device memory)
    uint64_t* total_out;          // receives *base_total + the range's caption bytes
    unsigned long long* tile_state;   // one word per tile, zero before the launch
    uint32_t* ticket;             // zero before the launch: tiles are handed out in the order blocks ask for them
    tp_counters* bx_counters;     // accumulated (may be null)
};

// General kernel: items are files file0 + i, or slow_list[i] when a list is given.  Item i uses lane i & 63 of
// workspace wave i >> 6.
struct PipelineArgs {
    const uint8_t* arena;
    const uint64_t* offsets;      // line ingest: file offsets; scored ingest: tag offsets
    const uint64_t* tag_begin;    // scored ingest only
    const double* scores;
    const uint8_t* sources;
    const uint64_t* secondary_begin;   // scored ingest of two tagger replies: first secondary tag of every file (else null)
    int secondary_cap;
    const uint8_t* reply_flags;   // scored ingest: per file, bit 0 = the (primary) reply was sorted by ParseJoyTags (null: every reply was)
    const uint32_t* item_list;    // may be null
    uint64_t file0;
    uint64_t n_items;
    uint32_t fpw;                 // files per wavefront of the workspace layout (64, or fewer to spread few files over many wavefronts)
    const uint32_t* wave_T;       // per workspace wave: token capacity and line-length capacity
    const uint32_t* wave_LB;
    const uint64_t* wave_off;     // per workspace wave: byte offset of its workspace
    uint8_t* workspace;
    tp_verdict* verdicts;
    uint32_t* n_final;
    uint32_t* cap_len;
    uint32_t* slot_of;
    uint32_t* fired;
    uint16_t* facet_choice;
    int mode;
    uint32_t* audit_cnt;          // mode 3: AUDIT_COLS entries per file, indexed by file - file0: entries and bytes of the audit row's lists
    uint32_t* audit_len;
    uint32_t* review_bits;        // mode 3: per file (indexed by file - file0) the REVIEW_* bits of the review record
};

// Text of the audit row's tag lists (keptTags, evidenceTags, DroppedTagsWithReason joined with ";", Program.cs:966) for the
// files of one general-kernel launch, out of their workspaces.
struct AuditArgs {
    uint64_t file0;
    uint64_t n_items;
    const uint32_t* item_list;    // may be null
    const uint32_t* slot_of;      // indexed by file
    const uint32_t* wave_T;
    const uint32_t* wave_LB;
    const uint64_t* wave_off;
    uint8_t* workspace;
    const uint32_t* audit_cnt;    // indexed by (file - file0) * AUDIT_COLS + column
    const uint64_t* offsets;      // indexed by file * AUDIT_COLS + column; the column's text is text[offsets[k] .. offsets[k + 1])
    uint8_t* text;
    uint32_t fpw;                 // files per wavefront of the general kernel's run (the stride of its workspace rows)
    const uint32_t* fin32;        // files of the lean stage (slot_of == SLOT_FAST): their caption's tag ids, FAST_FIN_MAX per file, and
    const uint32_t* aud32;        //   their other lists' entries, AUD_STRIDE per file, both indexed by file - file0 (null: no such files)
};

struct EmitArgs {
    uint64_t file0, file_end;
    const uint32_t* wave_T;
    const uint32_t* wave_LB;
    const uint64_t* wave_off;
    uint8_t* workspace;
    uint32_t* fin32;              // FAST_FIN_MAX tag ids per file, indexed by file - file0: written by the wave-per-file kernels for their files and by
                                  // the general emit kernel for the others (what the histogram kernel counts); null when neither is wanted
    int fast_ran;                 // the wave-per-file kernels ran over the range
    int wave_files_written;       // ... and the fused back half has written their captions and counted them: only the histogram is left to do for them
    const uint32_t* item_list;    // the files of the general kernel, relative to file0 (null: all files of the range)
    uint64_t n_items;             // how many of them
    const uint32_t* slot_of;
    const uint32_t* n_final;
    const uint64_t* caption_offsets;
    const tp_verdict* verdicts;
    uint8_t* captions;
    uint64_t* histogram;          // n_strings + 1, accumulated; may be null
    tp_counters* counters;        // accumulated; may be null
    uint32_t fpw;                 // files per wavefront of the general kernel's run (the stride of its workspace rows)
    // May be null.  HIST_PART_WORDS words for each of up to HIST_PART_BLOCKS blocks: the histogram kernels' blocks leave their
    // counters there (two 16-bit counters a word) and one thread per word adds the blocks' words up into the histogram,
    // instead of every block adding every counter that is not zero to the histogram with an atomic of its own.
    uint32_t* hist_parts;
};
constexpr uint32_t HIST_PART_WORDS = 32768, HIST_PART_BLOCKS = 256;

// Address and size of this device's static working memory (see tp_kernels.hip); `device` must be the current device.
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)16 << 20;
constexpr int NOALLOC_MAX_DEVICES = 64;
hipError_t noalloc_scratch(int device, void** ptr, size_t* bytes);

// `between`: when not null and the stage launches two kernels (stage 0), recorded on the stream after the first of them.
// stage 0: `between` and `after_front` (either may be null) are recorded between the front half and the back half
hipError_t launch_fast(const DevTables& t, const FastArgs& a, hipStream_t s, hipEvent_t between = nullptr, hipEvent_t after_front = nullptr);
hipError_t launch_wave_sizes(const DevTables& t, const uint8_t* arena, const uint64_t* offsets, const uint64_t* tag_begin, const uint32_t* item_list,
                             uint64_t file0, uint64_t n_items, uint32_t fpw, uint32_t* wave_T, uint32_t* wave_LB, uint64_t* wave_bytes, uint32_t* status, hipStream_t s);
hipError_t launch_scan_u64(const uint64_t* in, uint64_t n, uint64_t* out, uint64_t* tile_sums, const uint64_t* base_ptr, uint64_t* total_out, hipStream_t s);
hipError_t launch_scan_u32(const uint32_t* in, uint64_t n, uint64_t* out, uint64_t* tile_sums, const uint64_t* base_ptr, uint64_t* total_out, hipStream_t s);
uint64_t scan_tiles_needed(uint64_t n);
hipError_t launch_pipeline(const DevTables& t, const PipelineArgs& a, hipStream_t s);
hipError_t launch_audit(const DevTables& t, const AuditArgs& a, hipStream_t s);
hipError_t launch_emit(const DevTables& t, const EmitArgs& a, hipStream_t s);
// The strings noted in DevTables::unk_keys / unk_ent / unk_text, as records of 40 bytes {32 bytes of text, length, 0} in `out` (at most
// out_cap of them; *out_count receives how many there were); the set is emptied.
hipError_t launch_collect_unknown(unsigned long long* keys, const unsigned long long* ent, const uint8_t* text, uint32_t slots, uint8_t* out, uint32_t out_cap, uint32_t* out_count, hipStream_t s);
// The fused back half over [file0, file_end): a.tables, the lean front half's outputs and the fields above must be set.
hipError_t launch_lean_bx(const DevTables& t, const FastArgs& a, hipStream_t s);
uint64_t lean_bx_tiles(uint64_t n_files);                      // words of FastArgs::tile_state a range of n_files files needs
// dst[k] += src[k] for the eight batch counters (both device memory)
hipError_t launch_add_counters(tp_counters* dst, const tp_counters* src, hipStream_t s);
uint32_t fast_fin_stride();
bool lean_ok(const DevTables& t);   // the lean stage (tp_lean.h) can serve this configuration

}  // namespace tp
