// pack.hpp -- host-callable interface of pack.hip (the per-cell step packer on the device).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "records.hpp"

namespace mfsgd {

struct PackArgs {
    const int32_t* u;        // device: P-row index of every rating
    const int32_t* i;        // device: Q-row index
    const float* r;          // device: rating
    const long long* orig;   // device or null: caller-visible index of rating j (null: j itself)
    const unsigned* sorted;  // device: rating indices in bucket order
    const long long* bptr;   // device: B*B*W*W + 1 bucket starts
    const int32_t* urank;    // device: rank of a P row among the rows of its block (ascending index)
    const int32_t* irank;    // device: same for Q rows / tiles
    int B, W, G, L;
    float lr, c;
    int solo_ok;
    int max_m, max_rows;     // capacities of the kernel's LDS arrays (ratings per cell, rows per cell)
    int u_words, i_words;    // bitmap words: ceil(max rank + 1 / 32)
    int emit;                // 0: COUNT pass, 1: EMIT pass, 2: one pass -- COUNT outputs AND rows / entries into scratch arrays at
                             //    worst-case offsets (rows, entries point at the scratch), order at its final place; launch_compact
                             //    then moves the cells that are kept to their offsets
    PackCellInfo* info;      // COUNT out: per cell
    SubDesc* subs;           // COUNT out / EMIT in: per cell W*W
    const uint32_t* row_off; // EMIT in: per cell
    const uint32_t* ent_off; // EMIT in: per cell (step units)
    const long long* ord_off;  // EMIT in: per cell position in the canonical order
    uint32_t* rows;          // EMIT out
    Entry* entries;          // EMIT out
    long long* order;        // EMIT out
};

struct MixedSegment;
// one workgroup per segment: dst[seg.dst + x] = src[seg.src + x], x < seg.n, elements of elem_bytes (multiple of 4)
hipError_t launch_scatter(void* dst, const void* src, const MixedSegment* segs, long long n_segs, int elem_bytes, hipStream_t st);
size_t pack_lds_bytes(const PackArgs& a);
hipError_t launch_pack(const PackArgs& a, long long n_cells, hipStream_t st);
// one-pass mode: a.rows / a.entries / a.row_off / a.ent_off FINAL, a.info / a.subs from the pass, scratch arrays as written by it
hipError_t launch_compact(const PackArgs& a, long long n_cells, const uint32_t* srows, const Entry* sent, hipStream_t st);
size_t pack_scratch_steps(long long n, long long n_cells, int W);

}  // namespace mfsgd
