// Kernel argument blocks and launcher entry points (tp_kernels.hip), used by the C ABI layer (tp_api.cpp).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/tagpipe.h"
#include "tp_common.h"

namespace tp {

// One chunk of files [file0, file_end); file0 is a multiple of 64 so that a wavefront's lanes are 64 consecutive files.
struct PipelineArgs {
    const uint8_t* arena;
    const uint64_t* offsets;      // line ingest: file offsets; scored ingest: tag offsets
    const uint64_t* tag_begin;    // scored ingest only
    const double* scores;
    const uint8_t* sources;
    uint64_t file0, file_end;
    const uint32_t* wave_T;       // per wavefront: token capacity and line-length capacity of its workspace
    const uint32_t* wave_LB;
    const uint64_t* wave_off;     // per wavefront: byte offset of its workspace (global scan; the chunk's first wave is subtracted)
    uint8_t* workspace;
    tp_verdict* verdicts;
    uint32_t* n_final;
    uint32_t* cap_len;
    uint32_t* fired;
    uint16_t* facet_choice;
    int mode;
};

struct EmitArgs {
    uint64_t file0, file_end;
    const uint32_t* wave_T;
    const uint32_t* wave_LB;
    const uint64_t* wave_off;
    uint8_t* workspace;
    const uint32_t* n_final;
    const uint64_t* caption_offsets;
    const tp_verdict* verdicts;
    uint8_t* captions;
    uint64_t* histogram;          // n_strings + 1, accumulated; may be null
    tp_counters* counters;        // accumulated; may be null
};

hipError_t launch_wave_sizes(const DevTables& t, const uint8_t* arena, const uint64_t* offsets, const uint64_t* tag_begin, uint64_t n_files,
                             uint32_t* wave_T, uint32_t* wave_LB, uint64_t* wave_bytes, uint32_t* status, hipStream_t s);
hipError_t launch_scan_u64(const uint64_t* in, uint64_t n, uint64_t* out, uint64_t* tile_sums, const uint64_t* base_ptr, uint64_t* total_out, hipStream_t s);
hipError_t launch_scan_u32(const uint32_t* in, uint64_t n, uint64_t* out, uint64_t* tile_sums, const uint64_t* base_ptr, uint64_t* total_out, hipStream_t s);
uint64_t scan_tiles_needed(uint64_t n);
hipError_t launch_pipeline(const DevTables& t, const PipelineArgs& a, hipStream_t s);
hipError_t launch_emit(const DevTables& t, const EmitArgs& a, hipStream_t s);

}  // namespace tp
