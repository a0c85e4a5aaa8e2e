// ingest.hpp -- rating ingestion on the device (first phase):
// degree histograms and COO -> (cell, sub-round, wave) bucket order on the GPU.  The host
// scheduler (schedule.cpp, HIP-free) sees it only through these callbacks and falls back to
// its own loops when they are absent or fail; both produce identical arrays.
#pragma once

#include <cstdint>
#include <vector>

#include "records.hpp"

namespace mfsgd {

struct DeviceIngest {
    void* ctx = nullptr;
    // degu[U], degi[I]: number of ratings per row.  Returns 0 on success.
    int (*degrees)(void* ctx, const int32_t* u, const int32_t* i, int64_t n, int32_t U, int32_t I, int64_t* degu,
                   int64_t* degi) = nullptr;
    // Stable sort of the rating indices by bucket key
    //   key = (((ub * B + it) * W + s) * W + us),  ub = ubin[u] % B, us = ubin[u] / B, it = ibin[i] % B,
    //   is = ibin[i] / B, s = (is - us + W) % W; an item with a tile of its own (ibin[i] < giants: schedule.cpp,
    //   lpt_assign) takes us = 0 for all its ratings -- its tile holds nothing else, so its cells are ONE sub-cell
    // bptr[nb + 1] = first position of every bucket, sorted[n] = rating indices in key order (ties in
    // input order).  Returns 0 on success.
    int (*bucket)(void* ctx, const int32_t* u, const int32_t* i, int64_t n, const int32_t* ubin, const int32_t* ibin,
                  int32_t U, int32_t I, int B, int W, int giants, int64_t* bptr, int64_t* sorted) = nullptr;
    // Drops the device copy of the triples (and everything derived from it).  The copy is recognised by the host
    // pointers and the length, so it must be forgotten before another rating set is built from buffers that may
    // sit at the same addresses.
    void (*forget)(void* ctx) = nullptr;
    const struct DeviceIngestExt* ext = nullptr;  // the device packer, when available
};

// ---- second phase: the per-cell step packer on the device (pack.hip) ------------------------------
struct PackRequest {
    const int32_t* u = nullptr;     // host arrays handed to degrees() / bucket_dev() (identity check)
    const int32_t* i = nullptr;
    const float* r = nullptr;       // host: ratings
    const int64_t* orig = nullptr;  // host or null: caller-visible rating indices
    int64_t n = 0;
    int32_t U = 0, I = 0;
    const int32_t* ubin = nullptr;  // host: fine bin of every P row / Q row (block = bin % B)
    const int32_t* ibin = nullptr;
    int B = 0, W = 0, G = 0, L = 0;
    float lr = 0.f, c = 0.f;
    bool solo_ok = false;
    int64_t max_cell_nnz = 0;       // from bptr
    // the most rows a chunk of the training kernel's LDS image can hold (0: not said).  A cell with more is cut whatever
    // it packs into, so the packing kernel need not provide for it: it reports such a cell as too large (status 2, row
    // counts valid) and keeps the smaller per-wave arrays -- and the better occupancy -- for everything else.
    int fit_rows = 0;
    bool want_subs = true;  // false: the per-cell sub-cell tables stay on the device (the caller will ask the emit call for the final table)
    // [r3] host, per cell, or null: where the cell's ratings start in the canonical order (it follows from the bucket
    // starts alone).  With it the COUNT pass also WRITES what it packs -- rows and entries into scratch arrays at
    // worst-case offsets, the order at its final place -- and the emit calls below only move the cells that are kept to
    // their offsets: the packing runs once.  Without it (or when the scratch does not fit) they pack a second time.
    const int64_t* ord_off = nullptr;
};
// Buffers a successful emit() leaves on the device; the receiver frees them with `release`.
struct DevicePacked {
    void* rows = nullptr;     // uint32 x n_rows (+4 padding words)
    void* entries = nullptr;  // Entry x n_entries
    void* order = nullptr;    // int64 x n
    // [r3] the sub-cell tables of all chunk descriptors (n_subs SubDesc records, the two padding records included), when
    // the emit call was asked for them (n_descs > 0): the device wrote them and the training kernel reads them, so
    // they need not come to the host and go back
    void* subs = nullptr;
    int64_t n_subs = 0;
    void (*release)(void*) = nullptr;
};

// What the host packed in a mixed build: concatenated pieces and where each goes in the final arrays
// (element offsets: uint32 words of rows[], Entry records, int64 order positions).
struct MixedSegment {
    uint64_t dst, src, n;
};
struct MixedPieces {
    std::vector<uint32_t> rows;
    std::vector<Entry> entries;
    std::vector<int64_t> order;
    std::vector<MixedSegment> seg_rows, seg_entries, seg_order;
};

struct DeviceIngestExt {
    // like DeviceIngest::bucket, but the sorted indices stay on the device (only bptr comes back)
    int (*bucket_dev)(void* ctx, const int32_t* u, const int32_t* i, int64_t n, const int32_t* ubin, const int32_t* ibin,
                      int32_t U, int32_t I, int B, int W, int giants, int64_t* bptr) = nullptr;
    // the sorted indices after bucket_dev, for the host packer (fallback)
    int (*fetch_sorted)(void* ctx, int64_t* sorted) = nullptr;
    int (*fetch_sorted32)(void* ctx, uint32_t* sorted) = nullptr;  // the same, as the 32-bit indices the device holds
    // [r3] ... and only `n_ranges` pieces of them -- positions [lo[x], lo[x] + len[x]) of the bucket order, concatenated
    // into `out` (sum of len entries): the cells whose chunks the host has to decide (a gather on the device, ONE copy)
    int (*fetch_sorted_ranges)(void* ctx, int64_t n_ranges, const int64_t* lo, const int64_t* len, uint32_t* out) = nullptr;
    // COUNT pass: 0 = done (info: B*B, subs: B*B*W*W), 1 = this rating set is outside what the kernel
    // handles (nothing produced), -1 = a HIP call failed
    int (*pack_count)(void* ctx, const PackRequest& req, std::vector<PackCellInfo>& info, std::vector<SubDesc>& subs) = nullptr;
    // EMIT pass at the offsets the caller derived from the COUNT pass (per cell, B*B entries each)
    // n_descs > 0: also leave the final sub-cell table (n_descs * W*W + 2 records: the cells' tables as counted) in out->subs
    int (*pack_emit)(void* ctx, const uint32_t* row_off, const uint32_t* ent_off, const int64_t* ord_off, int64_t n_rows,
                     int64_t n_steps, int64_t n_descs, DevicePacked* out) = nullptr;
    // EMIT for the cells whose row_off is not 0xFFFFFFFF, then the host-packed pieces scattered to their places
    int (*pack_emit_mixed)(void* ctx, const uint32_t* row_off, const uint32_t* ent_off, const int64_t* ord_off, int64_t n_rows,
                           int64_t n_steps, const MixedPieces& host, DevicePacked* out) = nullptr;
    // [r3] Chunks on the device.  A chunk of a cell is a subset of its ratings (those inside a rectangle of user and
    // item ids, schedule.cpp) packed as a complete little cell; to the packing kernel it IS a cell, given as a list:
    // `sorted` = rating indices of the parts one after another, each in the cell's bucket order, `cptr` = n_parts * W*W
    // + 1 sub-cell starts into that list.  COUNT over such a list (any number of times, between pack_count and the
    // emit): info[n_parts], subs[n_parts * W*W].  Returns 0, or -1 when a HIP call failed.
    int (*pack_count_parts)(void* ctx, int64_t n_parts, const uint32_t* sorted, int64_t n_sorted, const int64_t* cptr,
                            PackCellInfo* info, SubDesc* subs) = nullptr;
    // EMIT for the whole cells whose row_off is not 0xFFFFFFFF AND for a final list of parts (the chunks of the cells
    // that were cut), each at the offsets the caller gives (p_*: per part), into ONE set of arrays
    int (*pack_emit_parts)(void* ctx, const uint32_t* row_off, const uint32_t* ent_off, const int64_t* ord_off, int64_t n_rows,
                           int64_t n_steps, int64_t n_parts, const uint32_t* sorted, int64_t n_sorted, const int64_t* cptr,
                           const uint32_t* p_row_off, const uint32_t* p_ent_off, const int64_t* p_ord_off,
                           int64_t n_descs, const int64_t* p_desc, DevicePacked* out) = nullptr;
    // (n_descs > 0 with p_desc = the chunk descriptor of every part: out->subs = the cells' tables, the parts' tables at
    // their descriptors, zeros elsewhere)
    // device -> host copies of what emit() produced (debug / get_order); any pointer may be null
    int (*download)(const DevicePacked& d, uint32_t* rows, int64_t n_rows, Entry* entries, int64_t n_entries, int64_t* order,
                    int64_t n) = nullptr;
    int (*download_raw)(const void* dev, void* host, size_t bytes) = nullptr;  // (the sub-cell tables, for the debug getter)
};

// Implemented in ingest.hip.  `device` must already be usable (capi checks).
DeviceIngest make_device_ingest(int device);
void destroy_device_ingest(DeviceIngest& d);

}  // namespace mfsgd
