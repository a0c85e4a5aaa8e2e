// schedule.hpp -- host-side rating bucketing and conflict-free step packing.
//
// Product code (part of libmfsgd.so).  No reference counterpart exists
// (/root/reference/README.md:1-2 is the whole reference); this implements
// SURVEY.md 2.2 row B4 ("rating bucketing / schedule for coalesced gather")
// and the blocking half of row B7 (DSGD user x item blocks).
//
// Model:
//   * users are cut into B user blocks, items into B item tiles, both balanced
//     by rating count (LPT), each further cut into W sub-groups;
//   * cell (ub, it) holds the ratings of user block ub on item tile it;
//     round rd of an epoch runs the B cells (b, (b + rd) % B) -- no two of them
//     share a user or an item, so they run on B workgroups with no ordering;
//   * inside a cell, sub-round s gives wave w the sub-cell
//     (user sub-group w, item sub-group (w + s) % W): again disjoint;
//   * inside a sub-cell the ratings are packed into steps of G slots (G = 64/L
//     lane groups of one wave); the G ratings of a step share no row.
// Executing rounds, cells, sub-rounds, waves, steps and slots in index order
// is the canonical sequential order; any conflict-free parallel execution
// gives bit-identical factors.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

#include "ingest.hpp"
#include "records.hpp"

namespace mfsgd {

struct Geometry {
    int k;         // latent dimension
    int L;         // lanes per rating: smallest power of two >= ceil(k/4)
    int G;         // ratings per wave step = 64 / L
    int kp;        // padded row length in floats = 4 * L
    int rowbytes;  // 16 * L
};
Geometry geometry_for_k(int k);
// LDS image of a workgroup: [control 16 B][schedule buffer 0][schedule buffer 1][rows].
// A schedule buffer holds one cell's step entries, sub-cell table and row ids; there are
// two so that the next cell's can be fetched while the current cell is being applied.
int64_t sched_bytes_for(const Geometry& geo, int W, int nrows, int64_t n_steps);
int64_t rows_bytes_for(const Geometry& geo, int nrows);

// Big flat arrays of the schedule: allocated without being cleared (a std::vector would
// write 0.9 GB of zeros for 20 M ratings before the packer overwrites every byte).
template <class T>
class PodVec {
public:
    PodVec() = default;
    PodVec(const PodVec&) = delete;
    PodVec& operator=(const PodVec&) = delete;
    PodVec(PodVec&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    PodVec& operator=(PodVec&& o) noexcept {
        if (this != &o) {
            std::free(p_);
            p_ = o.p_;
            n_ = o.n_;
            o.p_ = nullptr;
            o.n_ = 0;
        }
        return *this;
    }
    ~PodVec() { std::free(p_); }
    // contents are unspecified afterwards (not preserved, not cleared)
    void resize_uninit(size_t n) {
        std::free(p_);
        p_ = nullptr;
        n_ = 0;
        if (n) {
            p_ = static_cast<T*>(std::malloc(n * sizeof(T)));
            if (!p_) throw std::bad_alloc();
            n_ = n;
        }
    }
    T* data() { return p_; }
    const T* data() const { return p_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T& operator[](size_t i) { return p_[i]; }
    const T& operator[](size_t i) const { return p_[i]; }

private:
    T* p_ = nullptr;
    size_t n_ = 0;
};

// Device buffers of a schedule the device packer built (rows, entries, order never existed on the
// host); whoever ends up holding them frees them through `release`.
struct DeviceSchedule {
    DevicePacked buf{};
    DeviceSchedule() = default;
    DeviceSchedule(const DeviceSchedule&) = delete;
    DeviceSchedule& operator=(const DeviceSchedule&) = delete;
    DeviceSchedule(DeviceSchedule&& o) noexcept : buf(o.buf) { o.buf = DevicePacked{}; }
    DeviceSchedule& operator=(DeviceSchedule&& o) noexcept {
        if (this != &o) {
            reset();
            buf = o.buf;
            o.buf = DevicePacked{};
        }
        return *this;
    }
    ~DeviceSchedule() { reset(); }
    void reset() {
        if (buf.release) {
            buf.release(buf.rows);
            buf.release(buf.entries);
            buf.release(buf.order);
            buf.release(buf.subs);
        }
        buf = DevicePacked{};
    }
    bool present() const { return buf.entries != nullptr; }
};

struct SchedParams {
    int32_t U = 0, I = 0;   // row counts of P and of this partition's Q block
    int k = 0;
    float lr = 0.f, lambda = 0.f;  // baked into the step entries (lr*r, 1 - lr*lambda)
    int B = 0;              // 0 = auto
    int W = 0;              // 0 = auto
    int lds_budget = 160 * 1024 - 512;
    int n_cu = 256;
    int threads = 0;        // 0 = hardware_concurrency
    const DeviceIngest* ingest = nullptr;  // optional: degrees and bucket order computed on the GPU
    // optional: rating counts per P row / Q row the caller already has (skips that pass), and a
    // promise that every (u, i) has been range-checked
    const int64_t* degu = nullptr;
    const int64_t* degi = nullptr;
    bool validated = false;
    bool solo = true;  // allow solo runs (MFSGD_FLAG_NO_SOLO clears it: A/B measurements, tests)
    bool device_pack = true;  // let the device pack the cells when it can (needs `ingest` with the packer)
    bool lone_giants = true;  // an item that fills a fine bin by itself gets its tile to itself (schedule.cpp, lpt_assign)
};

struct Schedule {
    Geometry geo{};
    int B = 0, W = 0;
    int64_t nnz = 0;
    int lds_bytes = 0;   // 16 + 2 * sched_cap + largest rows image
    int sched_cap = 0;   // bytes of one schedule buffer (largest cell, multiple of 16)
    std::vector<CellDesc> cells;    // chunk descriptors: [0, B*B) first chunks (index ub*B + it), then the rest
    PodVec<uint32_t> rows;          // per chunk: nu user rows then ni item rows
    std::vector<SubDesc> subs;      // (desc*W + s)*W + w
    PodVec<Entry> entries;          // (desc.ent_off + step)*G + slot
    PodVec<int64_t> order;          // canonical order -> caller's rating index
    std::vector<int64_t> cell_ptr;  // B*B+1, round-major: rd*B + b
    // statistics
    int64_t total_steps = 0, total_rows = 0;
    int64_t max_cell_nnz = 0, max_cell_rows = 0, max_cell_steps = 0, sum_round_steps = 0;
    int64_t split_cells = 0;  // cells cut into more than one chunk
    double build_seconds = 0;
    bool device_ingest = false;  // degrees + bucket order came from the GPU
    // Device-packed schedules: rows / entries / order live in `dev` only (the PodVecs above stay empty
    // until somebody asks for a host copy); the counts are always valid.
    bool device_packed = false;
    DeviceSchedule dev;
    const DeviceIngestExt* dev_ops = nullptr;  // for the host copies on demand
    int64_t n_rows_words = 0;  // rows[] length incl. the 4 padding words
    int64_t n_entry_recs = 0;  // entries[] length
    int64_t n_sub_recs = 0;    // subs[] length incl. the 2 padding records (the array itself may live in dev.buf.subs only)
};

// u/i are row indices into P and into this partition's Q block; orig[j] is the
// caller-visible index of rating j (nullptr = j itself).
// Returns 0, or -1 with `err` set.  Cells too large for the LDS budget are chunked, so the
// budget alone never makes a schedule fail.
int build_schedule(const SchedParams& prm, const int32_t* u, const int32_t* i, const float* r,
                   const int64_t* orig, int64_t n, Schedule& out, std::string& err);

// Picks B, W and the LDS budget (workgroups per CU) when B / W are 0.
int build_schedule_auto(SchedParams prm, const int32_t* u, const int32_t* i, const float* r,
                        const int64_t* orig, int64_t n, Schedule& out, std::string& err);

// DSGD over G devices of ONE global rating set: users are cut into G contiguous
// ranges balanced by rating count -- device g keeps the P rows of users [user_begin[g],
// user_begin[g+1]) -- and items into G partitions balanced by rating count (LPT; items nobody
// rated are dealt out so that the row counts even out).  A pure function of the two degree
// arrays: every rank computes the same plan from the same (all-reduced) degrees.
// user_begin: G + 1 entries; item_part: I entries in [0, G).
void dsgd_plan(const int64_t* degu, const int64_t* degi, int32_t U, int32_t I, int32_t G, int32_t* user_begin,
               int32_t* item_part);
// The two halves (schedule.cpp): users over `G` ranks; items over `G` partitions of a job of `world` ranks at rank
// k -- balanced by rating count (LPT) and, for chain_crit > 0, chain-aware (the chain-critical items packed into as
// few partitions as the balance allows; schedule.cpp says when that pays and when it does not).  info (nullable): {sum of the partitions' heaviest items, critical items, partitions filled
// sequentially, threshold}.
void dsgd_plan_users(const int64_t* degu, int32_t U, int32_t G, int32_t* user_begin);
void dsgd_plan_items(const int64_t* degi, int32_t I, int32_t G, int32_t world, int32_t k, double chain_crit, int32_t* item_part,
                     int64_t* info);

}  // namespace mfsgd
