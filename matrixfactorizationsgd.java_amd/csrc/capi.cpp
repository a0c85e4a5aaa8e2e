// capi.cpp -- the C-ABI of include/mfsgd.h over the HIP kernels.
//
// No reference counterpart exists (/root/reference/README.md:1-2 is the whole
// reference); the surface follows SURVEY.md section 8b.  There is no CPU
// compute path in this library: every compute entry point needs a gfx950
// device and fails with MFSGD_ERR_NO_DEVICE otherwise.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/mfsgd.h"
#include "ingest.hpp"
#include "jrandom.hpp"
#include "kernels.hpp"
#include "schedule.hpp"

using namespace mfsgd;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct Part {
    Schedule sched;
    int32_t q_rows = 0;  // rows of this partition's Q block
    // Forwarding and register-resident runs exist on the kernel's q side only.  The update is
    // symmetric in p and q, so when the heaviest USER outweighs the heaviest item the schedule
    // is built with the roles exchanged and the kernels get (Q, P) instead of (P, Q).
    bool swapped = false;
    bool on_device = false;
    DevBuf d_cells, d_rows, d_subs, d_entries, d_sse_partial, d_sse_out;
    DevBuf d_sync;        // persistent kernel: done[B] words (kDoneStride apart) + the abort word
    int persistent_np = -1;  // co-resident workgroups of the epoch kernel; 0 = use round launches; -1 = not probed
    // training graphs keyed by the (P, Q) pointers they were captured with (both are baked into the
    // kernel node; P changes when the factors are re-seeded, Q with every caller-owned block)
    std::map<std::pair<const void*, const void*>, hipGraphExec_t> graphs;
};

}  // namespace

struct mfsgd_handle {
    mfsgd_config cfg{};
    Geometry geo{};
    int n_parts = 1;
    std::vector<Part> parts;
    bool have_ratings = false;
    int64_t nnz_total = 0;
    // DSGD item map (n_parts > 1): item i lives in partition item_part[i], row item_row[i] of that
    // partition's Q block.  Default: i % n_parts, i / n_parts; mfsgd_set_item_partition replaces it.
    std::vector<int32_t> item_part, item_row, part_q_rows;
    bool custom_item_map = false;

    // factors: host staging (kp-padded rows) until the device copy is created
    enum class Where { None, Host, Device } where = Where::None;
    std::vector<float> hP, hQ;
    DevBuf dP, dQ;

    bool device_ready = false;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;  // diagnostics only (mfsgd_debug_occupy)
    unsigned* occupy_started = nullptr; // ... pinned host word its workgroups count themselves in
    int64_t n_not_resident = 0;         // persistent launches that gave up at the residency check
    // identity of the rating set the schedules were built from: its length and a 128-bit hash of every
    // byte of u, i and r -- a repeated mfsgd_set_ratings with the same triples keeps the schedules
    uint64_t ratings_hash[2] = {0, 0};
    int64_t n_schedule_builds = 0, n_schedule_reuses = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    mutable std::string err;
};

namespace {

int fail(const mfsgd_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((h), e_ == hipErrorOutOfMemory ? MFSGD_ERR_OOM : MFSGD_ERR_HIP,         \
                        std::string(#call) + ": " + hipGetErrorString(e_));                     \
    } while (0)

int usable_devices(int* count, std::string* why) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        *count = 0;
        if (why) *why = std::string("no HIP device visible (") + hipGetErrorString(e) + ")";
        return 0;
    }
    *count = n;
    return 0;
}

int ensure_device(mfsgd_handle* h) {
    if (h->device_ready) {
        HIPCHK(h, hipSetDevice(h->cfg.device));
        return MFSGD_OK;
    }
    int n = 0;
    std::string why;
    usable_devices(&n, &why);
    if (n <= 0) return fail(h, MFSGD_ERR_NO_DEVICE, "libmfsgd has no CPU fallback: " + why);
    if (h->cfg.device < 0 || h->cfg.device >= n)
        return fail(h, MFSGD_ERR_NO_DEVICE, "device ordinal " + std::to_string(h->cfg.device) +
                                                " out of range (" + std::to_string(n) + " visible)");
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, h->cfg.device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(h, MFSGD_ERR_NO_DEVICE,
                    std::string("device is ") + prop.gcnArchName + "; libmfsgd is built for gfx950 only");
    h->n_cu = prop.multiProcessorCount;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreate(&h->ev0));
    HIPCHK(h, hipEventCreate(&h->ev1));
    h->device_ready = true;
    return MFSGD_OK;
}

int dev_alloc(mfsgd_handle* h, DevBuf& b, size_t bytes) {
    if (b.p && b.bytes >= bytes) return MFSGD_OK;
    b.release();
    if (bytes == 0) bytes = 16;
    HIPCHK(h, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return MFSGD_OK;
}

template <class T>
int upload(mfsgd_handle* h, DevBuf& b, const T& v) {
    using E = std::remove_reference_t<decltype(*v.data())>;
    int rc = dev_alloc(h, b, v.size() * sizeof(E));
    if (rc) return rc;
    if (!v.empty()) HIPCHK(h, hipMemcpy(b.p, v.data(), v.size() * sizeof(E), hipMemcpyHostToDevice));
    return MFSGD_OK;
}

void drop_graphs(Part& p) {
    for (auto& kv : p.graphs)
        if (kv.second) (void)hipGraphExecDestroy(kv.second);
    p.graphs.clear();
}

// The device copies of the factors go away (re-seed, set_factors, load): nothing captured with the
// old pointers may be replayed, and nothing may still be running on them.
void release_device_factors(mfsgd_handle* h) {
    if (h->where != mfsgd_handle::Where::Device) return;
    if (h->device_ready) {
        (void)hipSetDevice(h->cfg.device);
        (void)hipDeviceSynchronize();
    }
    for (Part& p : h->parts) drop_graphs(p);
    h->dP.release();
    h->dQ.release();
}

void default_item_map(mfsgd_handle* h) {
    const int G = h->n_parts;
    const int32_t I = h->cfg.n_items;
    h->item_part.resize((size_t)I);
    h->item_row.resize((size_t)I);
    h->part_q_rows.assign((size_t)G, 0);
    for (int32_t x = 0; x < I; ++x) {
        h->item_part[(size_t)x] = x % G;
        h->item_row[(size_t)x] = x / G;
    }
    for (int g = 0; g < G; ++g) h->part_q_rows[(size_t)g] = (I - g + G - 1) / G;
    h->custom_item_map = false;
}

// d_sync: done[B] words (kDoneStride apart), then {arrivals, generation, -, -} of the kernel's start-of-launch
// barrier, then {abort code, launches that started, -, -}.  Zeroed once, when allocated; the kernel keeps it
// consistent from launch to launch by itself.
// Behind them the tile mailboxes of the persistent kernel: B x kp granules of 8 bytes ({value, tag}; only tiles the
// scheduler marked kCellLoneTile use theirs).
size_t sync_bytes(const Part& p) {
    return ((size_t)p.sched.B * kDoneStride + 8) * sizeof(unsigned) + (size_t)p.sched.B * (size_t)p.sched.geo.L * 4 * 8;
}
unsigned* abort_word(const Part& p) { return static_cast<unsigned*>(p.d_sync.p) + (size_t)p.sched.B * kDoneStride + 4; }

int ensure_part_on_device(mfsgd_handle* h, Part& p) {
    if (p.on_device) return MFSGD_OK;
    int rc;
    if ((rc = upload(h, p.d_cells, p.sched.cells))) return rc;
    if (p.sched.device_packed && p.sched.dev.buf.subs) {
        // (the device assembled the sub-cell tables too)
        p.d_subs.release();
        p.d_subs.p = p.sched.dev.buf.subs;
        p.d_subs.bytes = (size_t)p.sched.dev.buf.n_subs * sizeof(SubDesc);
        p.sched.dev.buf.subs = nullptr;
    } else if ((rc = upload(h, p.d_subs, p.sched.subs))) {
        return rc;
    }
    if (p.sched.device_packed) {
        // the device packer left rows and entries where they are needed
        p.d_rows.release();
        p.d_entries.release();
        p.d_rows.p = p.sched.dev.buf.rows;
        p.d_rows.bytes = (size_t)p.sched.n_rows_words * sizeof(uint32_t);
        p.d_entries.p = p.sched.dev.buf.entries;
        p.d_entries.bytes = (size_t)p.sched.n_entry_recs * sizeof(Entry);
        p.sched.dev.buf.rows = nullptr;  // owned by the DevBufs from here on (same allocator: hipFree)
        p.sched.dev.buf.entries = nullptr;
    } else {
        if ((rc = upload(h, p.d_rows, p.sched.rows))) return rc;
        if ((rc = upload(h, p.d_entries, p.sched.entries))) return rc;
    }
    if ((rc = dev_alloc(h, p.d_sse_partial, sizeof(double) * p.sched.cells.size()))) return rc;
    if ((rc = dev_alloc(h, p.d_sse_out, sizeof(double)))) return rc;
    if ((rc = dev_alloc(h, p.d_sync, sync_bytes(p)))) return rc;
    HIPCHK(h, hipMemset(p.d_sync.p, 0, sync_bytes(p)));
    HIPCHK(h, hipDeviceSynchronize());  // (memset is asynchronous; the first launch may be on another stream)
    p.on_device = true;
    return MFSGD_OK;
}

// factors host <-> device -------------------------------------------------------
int factors_to_device(mfsgd_handle* h) {
    int rc = ensure_device(h);
    if (rc) return rc;
    if (h->where == mfsgd_handle::Where::Device) return MFSGD_OK;
    if (h->where == mfsgd_handle::Where::None)
        return fail(h, MFSGD_ERR_STATE, "factors not initialised: call mfsgd_init_factors or mfsgd_set_factors");
    if ((rc = upload(h, h->dP, h->hP))) return rc;
    if (h->n_parts == 1 && (rc = upload(h, h->dQ, h->hQ))) return rc;
    h->where = mfsgd_handle::Where::Device;
    std::vector<float>().swap(h->hP);
    std::vector<float>().swap(h->hQ);
    return MFSGD_OK;
}

CellLaunch make_launch(const mfsgd_handle* h, const Part& p, float* Q) {
    CellLaunch a{};
    a.P = p.swapped ? Q : static_cast<float*>(h->dP.p);
    a.Q = p.swapped ? static_cast<float*>(h->dP.p) : Q;
    a.cells = static_cast<const CellDesc*>(p.d_cells.p);
    a.rows = static_cast<const uint32_t*>(p.d_rows.p);
    a.subs = static_cast<const SubDesc*>(p.d_subs.p);
    a.entries = static_cast<const Entry*>(p.d_entries.p);
    a.B = p.sched.B;
    a.rd = 0;
    a.grid = p.sched.B;
    a.lds_bytes = p.sched.lds_bytes;
    a.sched_cap = p.sched.sched_cap;
    a.lr = h->cfg.lr;
    a.c = 1.0f - h->cfg.lr * h->cfg.lambda;
    a.sse_partial = static_cast<double*>(p.d_sse_partial.p);
    return a;
}

int launch_epoch_eager(mfsgd_handle* h, Part& p, float* Q, hipStream_t st) {
    CellLaunch a = make_launch(h, p, Q);
    for (int rd = 0; rd < p.sched.B; ++rd) {
        a.rd = rd;
        HIPCHK(h, launch_cell(true, h->geo.L, p.sched.W, a, st));
    }
    return MFSGD_OK;
}

// The persistent epoch kernel needs every one of its workgroups resident at once.
int probe_persistent(mfsgd_handle* h, Part& p) {
    if (p.persistent_np >= 0) return MFSGD_OK;
    p.persistent_np = 0;
    if (h->cfg.flags & MFSGD_FLAG_ROUND_LAUNCH) return MFSGD_OK;
    CellLaunch a = make_launch(h, p, nullptr);
    int per_cu = 0;
    hipError_t e = epoch_blocks_per_cu(h->geo.L, p.sched.W, a, &per_cu);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return MFSGD_OK;  // fall back to one launch per round
    }
    // DSGD partitions share the GPU with RCCL's send / recv kernels (a few workgroups, on their own stream): leave
    // them some CUs, or a persistent launch that needs the whole chip would sit in its residency check until the
    // exchange in flight has finished
    long np = (long)per_cu * (h->n_parts > 1 ? std::max(1, h->n_cu - 8) : h->n_cu);
    // test hook: pretend the chip holds this many times more workgroups than it does, so that the
    // residency check of the epoch kernel has to fail (tests/test_gpu_parity.py)
    if (const char* f = std::getenv("MFSGD_TEST_OVERSUBSCRIBE")) np *= std::max(1, std::atoi(f));
    p.persistent_np = (int)std::min<long>(np, p.sched.B);
    return MFSGD_OK;
}

int launch_epoch_body(mfsgd_handle* h, Part& p, float* Q, hipStream_t st) {
    if (p.persistent_np > 0) {
        CellLaunch a = make_launch(h, p, Q);
        a.grid = p.persistent_np;
        // flags are counted within the launch: zero them (and the abort word) every time
        // no memset: the kernel resets its own hand-off flags behind a device-side barrier (kernels.hip,
        // run_ring) -- a memset node in a replayed graph is not reliably ordered before the kernel node
        HIPCHK(h, launch_epoch_persistent(h->geo.L, p.sched.W, a, p.sched.B, static_cast<unsigned*>(p.d_sync.p),
                                          abort_word(p), st));
        return MFSGD_OK;
    }
    return launch_epoch_eager(h, p, Q, st);
}

// One epoch of partition p against Q on stream st (asynchronous).
int launch_epoch(mfsgd_handle* h, Part& p, float* Q, hipStream_t st) {
    if (p.sched.nnz == 0) return MFSGD_OK;
    int rc = probe_persistent(h, p);
    if (rc) return rc;
    if (h->cfg.flags & MFSGD_FLAG_NO_GRAPH) return launch_epoch_body(h, p, Q, st);
    const auto key = std::make_pair((const void*)h->dP.p, (const void*)Q);
    auto it = p.graphs.find(key);
    if (it == p.graphs.end()) {
        // capture the launch(es) of one epoch once; replayed every epoch
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        rc = launch_epoch_body(h, p, Q, h->stream);
        hipError_t e = hipStreamEndCapture(h->stream, &graph);
        if (rc) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        HIPCHK(h, e);
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIPCHK(h, e);
        if (p.graphs.size() >= 32) {
            // replays of the old graphs may still be in flight on a caller's stream
            HIPCHK(h, hipDeviceSynchronize());
            drop_graphs(p);
        }
        it = p.graphs.emplace(key, exec).first;
    }
    HIPCHK(h, hipGraphLaunch(it->second, st));
    return MFSGD_OK;
}

// After a synchronisation point: did a persistent launch give up?  Returns MFSGD_OK, MFSGD_ERR_HIP (a
// hand-off or a solo helper timed out mid-epoch: the factors are invalid), or kNotResident: the
// launch found its workgroups not co-resident and did NOTHING (nor did any launch queued behind it);
// *started receives the number of launches since the last check that did run.
constexpr int kNotResident = 1;
int check_abort(mfsgd_handle* h, Part& p, unsigned* started = nullptr) {
    if (started) *started = 0;
    if (p.persistent_np <= 0 || !p.d_sync.p) return MFSGD_OK;
    unsigned w[2] = {0, 0};
    HIPCHK(h, hipMemcpy(w, abort_word(p), sizeof w, hipMemcpyDeviceToHost));
    if (started) *started = w[1];
    if (w[0] != 0 || w[1] != 0) {
        // the device is idle on this stream (the caller has synchronised).  A give-up also leaves arrivals (and the
        // "somebody left" bit) in the start barrier's counter: zero it.  The GENERATION word next to it is never reset:
        // the tile mailboxes are tagged with it, and a launch that reused a generation could take a granule an aborted
        // launch left behind for this launch's (advisor finding, round 2).
        const unsigned zeros[2] = {0, 0};
        if (w[0] != 0) (void)hipMemcpy(abort_word(p) - 4, zeros, sizeof(unsigned), hipMemcpyHostToDevice);
        (void)hipMemcpy(abort_word(p), zeros, 2 * sizeof(unsigned), hipMemcpyHostToDevice);
    }
    if (w[0] == 2u) return kNotResident;
    if (w[0] != 0)
        return fail(h, MFSGD_ERR_HIP, "persistent epoch kernel timed out waiting for a tile hand-off (results invalid)");
    return MFSGD_OK;
}

// The persistent kernel could not get all its workgroups onto the chip (something else is running
// there): from now on this partition is trained with one launch per round, which needs no co-residency.
// `idle` (nullable): the one stream this partition's launches went to, already synchronised by the caller -- then
// nothing of the partition is in flight and the device-wide wait (which would also wait for a DSGD ring's exchange
// with a slower peer on its communication stream) is not needed.
void give_up_persistence(mfsgd_handle* h, Part& p, const hipStream_t* idle = nullptr) {
    if (!idle) (void)hipDeviceSynchronize();
    drop_graphs(p);
    p.persistent_np = 0;
    h->n_not_resident++;
}

// For callers that cannot re-run what was skipped (asynchronous DSGD sub-epochs on caller-owned blocks).
int check_abort_strict(mfsgd_handle* h, Part& p) {
    const int rc = check_abort(h, p);
    if (rc != kNotResident) return rc;
    give_up_persistence(h, p);
    return fail(h, MFSGD_ERR_HIP,
                "persistent epoch kernel: workgroups not co-resident (another kernel holds the GPU); the launch and those "
                "queued behind it of THIS partition were NOT applied -- the partition now uses one launch per round.  If other "
                "work depended on it (a DSGD ring that passed the block on), the factors are invalid: seed or load them again");
}

int launch_sse(mfsgd_handle* h, Part& p, const float* Q, hipStream_t st) {
    CellLaunch a = make_launch(h, p, const_cast<float*>(Q));
    const int n_cells = (int)p.sched.cells.size();  // chunk descriptors: every one is independent here
    if (h->cfg.flags & MFSGD_FLAG_ROUND_LAUNCH) {  // reference form: one workgroup per chunk
        a.grid = n_cells;
        HIPCHK(h, launch_cell(false, h->geo.L, p.sched.W, a, st));
    } else {
        const int per_cu = std::max(1, std::min(4, (160 * 1024) / std::max(1, p.sched.lds_bytes)));
        a.grid = std::min(n_cells, per_cu * std::max(1, h->n_cu));
        HIPCHK(h, launch_sse_persistent(h->geo.L, p.sched.W, a, n_cells, st));
    }
    HIPCHK(h, launch_reduce_sse(a.sse_partial, (int64_t)a.grid, static_cast<double*>(p.d_sse_out.p), st));
    return MFSGD_OK;
}

int part_sse_sync(mfsgd_handle* h, Part& p, const float* Q, hipStream_t st, double* sse) {
    if (p.sched.nnz == 0) {
        *sse = 0.0;
        return MFSGD_OK;
    }
    int rc = launch_sse(h, p, Q, st);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(sse, p.d_sse_out.p, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    return check_abort_strict(h, p);
}

int launch_epoch(mfsgd_handle* h, Part& p, float* Q, hipStream_t st);

// `launched` epochs of a single-partition handle are in flight on st: wait, and if the persistent
// kernel found itself not resident (it then did nothing, nor did the launches behind it), run what
// is missing as one launch per round.
int settle_epochs(mfsgd_handle* h, Part& p, float* Q, hipStream_t st, int launched) {
    HIPCHK(h, hipStreamSynchronize(st));
    unsigned started = 0;
    int rc = check_abort(h, p, &started);
    if (rc != kNotResident) return rc;
    give_up_persistence(h, p, &st);
    for (int e = (int)std::min<unsigned>(started, (unsigned)launched); e < launched; ++e)
        if ((rc = launch_epoch(h, p, Q, st))) return rc;
    HIPCHK(h, hipStreamSynchronize(st));
    return kNotResident;  // recovered: the missing epochs ran as round launches
}

int settle_epochs_ok(mfsgd_handle* h, Part& p, float* Q, hipStream_t st, int launched) {
    const int rc = settle_epochs(h, p, Q, st, launched);
    return rc == kNotResident ? MFSGD_OK : rc;
}

// Host copies of a device-packed schedule's big arrays, made when somebody asks for them
// (mfsgd_get_order, the debug getters): they never existed on the host.
int host_copies(const mfsgd_handle* h, const Part& cp, bool want_order, bool want_arrays) {
    Part& p = const_cast<Part&>(cp);
    Schedule& s = p.sched;
    if (!s.device_packed || !s.dev_ops || !s.dev_ops->download) return MFSGD_OK;
    try {
        DevicePacked d = s.dev.buf;
        if (p.on_device) {  // rows / entries have moved into the DevBufs
            d.rows = p.d_rows.p;
            d.entries = p.d_entries.p;
        }
        if (want_order && s.order.empty() && s.nnz > 0) {
            s.order.resize_uninit((size_t)s.nnz);
            if (s.dev_ops->download(d, nullptr, 0, nullptr, 0, s.order.data(), s.nnz) != 0)
                return fail(h, MFSGD_ERR_HIP, "could not copy the canonical order from the device");
        }
        if (want_arrays && s.subs.empty() && s.n_sub_recs > 0 && s.dev_ops->download_raw) {
            const void* dsubs = p.on_device ? p.d_subs.p : s.dev.buf.subs;
            if (dsubs) {
                s.subs.resize((size_t)s.n_sub_recs);
                if (s.dev_ops->download_raw(dsubs, s.subs.data(), s.subs.size() * sizeof(SubDesc)) != 0)
                    return fail(h, MFSGD_ERR_HIP, "could not copy the sub-cell tables from the device");
            }
        }
        if (want_arrays && s.entries.empty() && s.n_entry_recs > 0) {
            s.rows.resize_uninit((size_t)s.n_rows_words);
            s.entries.resize_uninit((size_t)s.n_entry_recs);
            if (s.dev_ops->download(d, s.rows.data(), s.n_rows_words, s.entries.data(), s.n_entry_recs, nullptr, 0) != 0)
                return fail(h, MFSGD_ERR_HIP, "could not copy the schedule from the device");
        }
        return MFSGD_OK;
    } catch (const std::bad_alloc&) {
        return fail(h, MFSGD_ERR_OOM, "out of host memory");
    }
}

int prepare_compute(mfsgd_handle* h) {
    if (!h->have_ratings) return fail(h, MFSGD_ERR_STATE, "no ratings: call mfsgd_set_ratings first");
    int rc = factors_to_device(h);
    if (rc) return rc;
    for (Part& p : h->parts)
        if ((rc = ensure_part_on_device(h, p))) return rc;
    return MFSGD_OK;
}

// 2 x 64-bit multiply-xorshift hash of the three rating arrays (every byte; the array boundaries and
// the length are mixed in), computed in parallel over fixed 1 MiB pieces so that it does not depend on
// the thread count.  Not cryptographic; 128 bits make an accidental match of two different rating sets
// (the only thing it guards against) a non-event.
void hash_ratings(const int32_t* u, const int32_t* i, const float* r, int64_t n, int threads, uint64_t out[2]) {
    constexpr uint64_t M1 = 0x9E3779B97F4A7C15ull, M2 = 0xC2B2AE3D27D4EB4Full;
    constexpr int64_t kPiece = 1 << 18;  // elements per piece
    const int64_t pieces = (n + kPiece - 1) / kPiece;
    std::vector<uint64_t> ph((size_t)pieces * 6, 0);
    auto piece_hash = [&](const void* base, int64_t lo, int64_t hi, uint64_t& a, uint64_t& b) {
        const unsigned char* p = static_cast<const unsigned char*>(base) + lo * 4;
        const int64_t bytes = (hi - lo) * 4;
        uint64_t h1 = 0x243F6A8885A308D3ull ^ (uint64_t)bytes, h2 = 0x13198A2E03707344ull + (uint64_t)bytes;
        int64_t x = 0;
        for (; x + 8 <= bytes; x += 8) {
            uint64_t w;
            std::memcpy(&w, p + x, 8);
            h1 = (h1 ^ w) * M1;
            h1 ^= h1 >> 32;
            h2 = (h2 + w) * M2;
            h2 ^= h2 >> 29;
        }
        if (x < bytes) {
            uint64_t w = 0;
            std::memcpy(&w, p + x, (size_t)(bytes - x));
            h1 = (h1 ^ w) * M1;
            h1 ^= h1 >> 32;
            h2 = (h2 + w) * M2;
            h2 ^= h2 >> 29;
        }
        a = h1;
        b = h2;
    };
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t c = next.fetch_add(1);
            if (c >= pieces) break;
            const int64_t lo = c * kPiece, hi = std::min(n, lo + kPiece);
            piece_hash(u, lo, hi, ph[(size_t)c * 6 + 0], ph[(size_t)c * 6 + 1]);
            piece_hash(i, lo, hi, ph[(size_t)c * 6 + 2], ph[(size_t)c * 6 + 3]);
            piece_hash(r, lo, hi, ph[(size_t)c * 6 + 4], ph[(size_t)c * 6 + 5]);
        }
    };
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(nt, 64), pieces));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    uint64_t h1 = 0x452821E638D01377ull ^ (uint64_t)n, h2 = 0xBE5466CF34E90C6Cull + (uint64_t)n;
    for (size_t x = 0; x < ph.size(); x += 2) {
        h1 = (h1 ^ ph[x]) * M1;
        h1 ^= h1 >> 32;
        h2 = (h2 + ph[x + 1]) * M2;
        h2 ^= h2 >> 29;
    }
    out[0] = h1;
    out[1] = h2;
}

void fill_rows(JRandom& g, float* dst, int64_t rows, int k, int kp, float scale) {
    for (int64_t x = 0; x < rows; ++x) {
        float* row = dst + x * kp;
        for (int f = 0; f < k; ++f) row[f] = g.nextFloat() * scale;
        for (int f = k; f < kp; ++f) row[f] = 0.0f;
    }
}

}  // namespace

// =============================================================================
extern "C" {

int mfsgd_abi_version(void) { return MFSGD_ABI_VERSION; }

int mfsgd_device_count(int32_t* out) {
    if (!out) return MFSGD_ERR_INVALID_ARG;
    int n = 0;
    usable_devices(&n, nullptr);
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    *out = ok;
    return MFSGD_OK;
}

int mfsgd_create(const mfsgd_config* cfg, mfsgd_handle** out) {
    if (out) *out = nullptr;
    if (!cfg || !out) {
        g_create_error = "mfsgd_create: null argument";
        return MFSGD_ERR_INVALID_ARG;
    }
    auto bad = [&](const char* m, int code = MFSGD_ERR_INVALID_ARG) {
        g_create_error = std::string("mfsgd_create: ") + m;
        return code;
    };
    if (cfg->n_users < 1 || cfg->n_items < 1) return bad("n_users and n_items must be >= 1");
    if (cfg->k < 1) return bad("k must be >= 1");
    if (cfg->k > MFSGD_MAX_K) return bad("k exceeds MFSGD_MAX_K (256)", MFSGD_ERR_UNSUPPORTED);
    if (!(cfg->lr == cfg->lr) || !(cfg->lambda == cfg->lambda)) return bad("lr / lambda is NaN");
    if (cfg->blocks < 0 || cfg->waves < 0 || cfg->n_parts < 0 || cfg->device < 0 || cfg->host_threads < 0)
        return bad("negative geometry field");
    if (cfg->waves != 0 && cfg->waves != 1 && cfg->waves != 2 && cfg->waves != 4 && cfg->waves != 8)
        return bad("waves must be 0 (auto), 1, 2, 4 or 8");
    for (int x = 0; x < 5; ++x)
        if (cfg->reserved[x] != 0) return bad("reserved fields must be zero");
    mfsgd_handle* h = new (std::nothrow) mfsgd_handle();
    if (!h) return bad("out of host memory", MFSGD_ERR_OOM);
    h->cfg = *cfg;
    h->geo = geometry_for_k(cfg->k);
    h->n_parts = cfg->n_parts > 1 ? cfg->n_parts : 1;
    if (h->n_parts > cfg->n_items) {
        delete h;
        return bad("n_parts exceeds n_items");
    }
    try {
        if (h->n_parts > 1) default_item_map(h);
    } catch (const std::bad_alloc&) {
        delete h;
        return bad("out of host memory", MFSGD_ERR_OOM);
    }
    *out = h;
    return MFSGD_OK;
}

void mfsgd_destroy(mfsgd_handle* h) {
    if (!h) return;
    if (h->device_ready) {
        (void)hipSetDevice(h->cfg.device);
        (void)hipStreamSynchronize(h->stream);
    }
    for (Part& p : h->parts) {
        drop_graphs(p);
        p.d_cells.release();
        p.d_rows.release();
        p.d_subs.release();
        p.d_entries.release();
        p.d_sse_partial.release();
        p.d_sse_out.release();
        p.d_sync.release();
    }
    h->dP.release();
    h->dQ.release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->side_stream) {
        (void)hipStreamSynchronize(h->side_stream);
        if (h->occupy_started) (void)hipHostFree(h->occupy_started);
        h->occupy_started = nullptr;
        (void)hipStreamDestroy(h->side_stream);
    }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* mfsgd_last_error(const mfsgd_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mfsgd_set_ratings(mfsgd_handle* h, const int32_t* u, const int32_t* i, const float* r, int64_t nnz) {
    if (!h) return MFSGD_ERR_INVALID_ARG;
    if (nnz < 0 || (nnz > 0 && (!u || !i || !r))) return fail(h, MFSGD_ERR_INVALID_ARG, "set_ratings: null array or negative nnz");
    const bool trace = std::getenv("MFSGD_SCHED_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[set_ratings] %-25s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    try {
        // The same triples again (Java / C++ / Python hosts hand train() the same arrays every call): keep
        // the schedules and their device copies.  Exact: length + 128-bit hash of every byte.
        uint64_t hash[2];
        hash_ratings(u, i, r, nnz, h->cfg.host_threads, hash);
        if (h->have_ratings && h->nnz_total == nnz && hash[0] == h->ratings_hash[0] && hash[1] == h->ratings_hash[1]) {
            h->n_schedule_reuses++;
            return MFSGD_OK;
        }
        lap("hash of the triples");
        // drop what an earlier call built (device copies included)
        if (h->device_ready) {
            (void)hipSetDevice(h->cfg.device);
            (void)hipStreamSynchronize(h->stream);
        }
        for (Part& p : h->parts) {
            drop_graphs(p);
            p.d_cells.release();
            p.d_rows.release();
            p.d_subs.release();
            p.d_entries.release();
            p.d_sse_partial.release();
            p.d_sse_out.release();
            p.d_sync.release();
        }
        h->parts.clear();
        h->have_ratings = false;
        {
            // range check, on the host threads: first offending rating, if any
            int nt = h->cfg.host_threads > 0 ? h->cfg.host_threads : (int)std::thread::hardware_concurrency();
            nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(nt, 64), nnz >> 16));
            std::vector<int64_t> first_bad((size_t)nt, nnz);
            auto scan = [&](int t) {
                const int64_t lo = nnz * t / nt, hi = nnz * (t + 1) / nt;
                const int32_t nu = h->cfg.n_users, ni = h->cfg.n_items;
                for (int64_t j = lo; j < hi; ++j)
                    if ((uint32_t)u[j] >= (uint32_t)nu || (uint32_t)i[j] >= (uint32_t)ni) {
                        first_bad[(size_t)t] = j;
                        return;
                    }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) th.emplace_back(scan, t);
            scan(0);
            for (auto& t : th) t.join();
            const int64_t j = *std::min_element(first_bad.begin(), first_bad.end());
            if (j < nnz)
                return fail(h, MFSGD_ERR_INVALID_ARG, "set_ratings: rating " + std::to_string(j) + " has (u,i) = (" +
                                                          std::to_string(u[j]) + "," + std::to_string(i[j]) + ") out of range");
        }
        lap("release + range check");
        const int G = h->n_parts;
        h->parts.resize((size_t)G);
        // Ingestion (degree histograms, bucket order) runs on the GPU when there is one and the
        // rating set is large enough to pay for the upload; the host loops are the fallback.
        DeviceIngest ingest;
        const bool want_dev = !(h->cfg.flags & MFSGD_FLAG_HOST_INGEST) &&
                              ((h->cfg.flags & MFSGD_FLAG_DEVICE_INGEST) || nnz >= (int64_t)1 << 20);
        if (want_dev) {
            if (ensure_device(h) == MFSGD_OK) ingest = make_device_ingest(h->cfg.device);
            else h->err.clear();  // no device: not an error for a host-side call
        }
        struct IngestGuard {
            DeviceIngest& d;
            ~IngestGuard() { destroy_device_ingest(d); }
        } ingest_guard{ingest};
        lap("device ingest context");
        SchedParams prm;
        prm.ingest = ingest.ctx ? &ingest : nullptr;
        prm.U = h->cfg.n_users;
        prm.k = h->cfg.k;
        prm.lr = h->cfg.lr;
        prm.lambda = h->cfg.lambda;
        prm.B = h->cfg.blocks;
        prm.W = h->cfg.waves;
        prm.threads = h->cfg.host_threads;
        prm.solo = !(h->cfg.flags & MFSGD_FLAG_NO_SOLO);
        prm.device_pack = !(h->cfg.flags & MFSGD_FLAG_HOST_PACK);
        if (const char* e = std::getenv("MFSGD_LDS_BUDGET")) {  // (A/B measurements: e.g. 81408 = two workgroups per CU)
            const int v = std::atoi(e);
            if (v >= 16 * 1024 && v <= prm.lds_budget) prm.lds_budget = v;
        }
        if (G == 1) {
            Part& p = h->parts[0];
            p.q_rows = h->cfg.n_items;
            // rating counts per user and per item: on the device when it ingests, else here; they
            // decide which side carries the longest chain and are handed on to the scheduler
            std::vector<int64_t> du((size_t)h->cfg.n_users, 0), di((size_t)h->cfg.n_items, 0);
            if (!(ingest.ctx && ingest.degrees &&
                  ingest.degrees(ingest.ctx, u, i, nnz, h->cfg.n_users, h->cfg.n_items, du.data(), di.data()) == 0)) {
                std::fill(du.begin(), du.end(), 0);
                std::fill(di.begin(), di.end(), 0);
                for (int64_t j = 0; j < nnz; ++j) {
                    du[(size_t)u[j]]++;
                    di[(size_t)i[j]]++;
                }
            }
            const int64_t mu = du.empty() ? 0 : *std::max_element(du.begin(), du.end());
            const int64_t mi = di.empty() ? 0 : *std::max_element(di.begin(), di.end());
            prm.validated = true;  // the loop above checked every (u, i)
            lap("degrees + role decision");
            p.swapped = mu > mi;
            std::string err;
            int rc;
            prm.degu = p.swapped ? di.data() : du.data();
            prm.degi = p.swapped ? du.data() : di.data();
            if (p.swapped) {
                prm.U = h->cfg.n_items;
                prm.I = h->cfg.n_users;
                rc = build_schedule_auto(prm, i, u, r, nullptr, nnz, p.sched, err);
                prm.U = h->cfg.n_users;
            } else {
                prm.I = p.q_rows;
                rc = build_schedule_auto(prm, u, i, r, nullptr, nnz, p.sched, err);
            }
            if (rc != 0) return fail(h, MFSGD_ERR_SCHEDULE, err);
        } else {
            // item i -> partition item_part[i], local row item_row[i]: one counting sort of the rating
            // indices by partition, then one schedule per partition
            std::vector<int64_t> pptr((size_t)G + 1, 0);
            for (int64_t j = 0; j < nnz; ++j) pptr[(size_t)h->item_part[(size_t)i[j]] + 1]++;
            for (int g = 0; g < G; ++g) pptr[(size_t)g + 1] += pptr[(size_t)g];
            std::vector<int64_t> orig((size_t)nnz);
            {
                std::vector<int64_t> cur(pptr.begin(), pptr.end() - 1);
                for (int64_t j = 0; j < nnz; ++j) orig[(size_t)cur[(size_t)h->item_part[(size_t)i[j]]]++] = j;
            }
            lap("partition split");
            for (int g = 0; g < G; ++g) {
                const int64_t lo = pptr[(size_t)g], m = pptr[(size_t)g + 1] - lo;
                std::vector<int32_t> uu((size_t)m), ii((size_t)m);
                std::vector<float> rr((size_t)m);
                for (int64_t x = 0; x < m; ++x) {
                    const int64_t j = orig[(size_t)(lo + x)];
                    uu[(size_t)x] = u[j];
                    ii[(size_t)x] = h->item_row[(size_t)i[j]];
                    rr[(size_t)x] = r[j];
                }
                Part& p = h->parts[(size_t)g];
                p.q_rows = h->part_q_rows[(size_t)g];
                // uu / ii of two partitions of equal size sit at the same addresses (the allocator hands the block
                // back): the ingest context must not take them for the arrays it already holds on the device
                if (ingest.ctx && ingest.forget) ingest.forget(ingest.ctx);
                prm.I = std::max<int32_t>(1, p.q_rows);
                prm.validated = true;  // checked above; local rows are in range by construction
                // the partition's own rating counts per row: build_schedule_auto compares the longest chain with the
                // partition's work when it picks the wave count (a chain-bound partition runs on two waves, a
                // work-bound one on four), exactly as for a single-partition handle
                std::vector<int64_t> du((size_t)h->cfg.n_users, 0), di((size_t)prm.I, 0);
                for (int64_t x = 0; x < m; ++x) {
                    du[(size_t)uu[(size_t)x]]++;
                    di[(size_t)ii[(size_t)x]]++;
                }
                prm.degu = du.data();
                prm.degi = di.data();
                std::string err;
                if (build_schedule_auto(prm, uu.data(), ii.data(), rr.data(), orig.data() + lo, m, p.sched, err) != 0)
                    return fail(h, MFSGD_ERR_SCHEDULE, "partition " + std::to_string(g) + ": " + err);
            }
        }
        lap("schedules");
        h->nnz_total = nnz;
        h->ratings_hash[0] = hash[0];
        h->ratings_hash[1] = hash[1];
        h->n_schedule_builds++;
        h->have_ratings = true;
        return MFSGD_OK;
    } catch (const std::bad_alloc&) {
        return fail(h, MFSGD_ERR_OOM, "set_ratings: out of host memory");
    } catch (const std::exception& e) {
        return fail(h, MFSGD_ERR_INVALID_ARG, std::string("set_ratings: ") + e.what());
    }
}

// Seeds P (stream position of row u: (u_offset + u) * k) and, for single-partition handles with with_q, Q
// (row i: (n_users + i) * k).  On the device when there is one (a kernel per matrix; nothing crosses PCIe);
// on the host otherwise (host-only callers: the factors are uploaded when the first compute call comes).
static int seed_factors(mfsgd_handle* h, int64_t seed, int64_t u_offset, bool with_q) {
    const int k = h->cfg.k, kp = h->geo.kp;
    const float scale = (float)(1.0 / std::sqrt((double)k));
    release_device_factors(h);
    h->where = mfsgd_handle::Where::None;
    if (ensure_device(h) == MFSGD_OK) {
        int rc;
        if ((rc = dev_alloc(h, h->dP, sizeof(float) * (size_t)h->cfg.n_users * kp))) return rc;
        HIPCHK(h, launch_init_rows(static_cast<float*>(h->dP.p), h->cfg.n_users, k, kp, seed, (unsigned long long)u_offset * (unsigned long long)k,
                                   scale, h->stream));
        if (with_q) {
            if ((rc = dev_alloc(h, h->dQ, sizeof(float) * (size_t)h->cfg.n_items * kp))) return rc;
            HIPCHK(h, launch_init_rows(static_cast<float*>(h->dQ.p), h->cfg.n_items, k, kp, seed,
                                       (unsigned long long)h->cfg.n_users * (unsigned long long)k, scale, h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::vector<float>().swap(h->hP);
        std::vector<float>().swap(h->hQ);
        h->where = mfsgd_handle::Where::Device;
        return MFSGD_OK;
    }
    h->err.clear();  // no device: not an error for this call
    h->hP.assign((size_t)h->cfg.n_users * kp, 0.0f);
    JRandom g(seed);
    g.skip((uint64_t)u_offset * (uint64_t)k);
    fill_rows(g, h->hP.data(), h->cfg.n_users, k, kp, scale);
    h->hQ.clear();
    if (with_q) {
        h->hQ.assign((size_t)h->cfg.n_items * kp, 0.0f);
        JRandom gq(seed);
        gq.skip((uint64_t)h->cfg.n_users * (uint64_t)k);
        fill_rows(gq, h->hQ.data(), h->cfg.n_items, k, kp, scale);
    }
    h->where = mfsgd_handle::Where::Host;
    return MFSGD_OK;
}

int mfsgd_init_p_offset(mfsgd_handle* h, int64_t seed, int64_t u_offset) {
    if (!h || u_offset < 0) return fail(h, MFSGD_ERR_INVALID_ARG, "init_p_offset: bad argument");
    try {
        return seed_factors(h, seed, u_offset, false);
    } catch (const std::bad_alloc&) {
        return fail(h, MFSGD_ERR_OOM, "init_p_offset: out of host memory");
    }
}

int mfsgd_init_factors(mfsgd_handle* h, int64_t seed) {
    if (!h) return MFSGD_ERR_INVALID_ARG;
    try {
        return seed_factors(h, seed, 0, h->n_parts == 1);  // n_parts > 1: Q lives in caller-owned blocks
    } catch (const std::bad_alloc&) {
        return fail(h, MFSGD_ERR_OOM, "init_factors: out of host memory");
    }
}

int mfsgd_set_factors(mfsgd_handle* h, const float* P, const float* Q) {
    if (!h || !P) return fail(h, MFSGD_ERR_INVALID_ARG, "set_factors: P is null");
    if (h->n_parts == 1 && !Q) return fail(h, MFSGD_ERR_INVALID_ARG, "set_factors: Q is null");
    try {
        const int k = h->cfg.k, kp = h->geo.kp;
        release_device_factors(h);
        h->hP.assign((size_t)h->cfg.n_users * kp, 0.0f);
        for (int64_t x = 0; x < h->cfg.n_users; ++x) std::memcpy(&h->hP[(size_t)x * kp], P + x * k, sizeof(float) * (size_t)k);
        h->hQ.clear();
        if (h->n_parts == 1) {
            h->hQ.assign((size_t)h->cfg.n_items * kp, 0.0f);
            for (int64_t x = 0; x < h->cfg.n_items; ++x) std::memcpy(&h->hQ[(size_t)x * kp], Q + x * k, sizeof(float) * (size_t)k);
        }
        h->where = mfsgd_handle::Where::Host;
        return MFSGD_OK;
    } catch (const std::bad_alloc&) {
        return fail(h, MFSGD_ERR_OOM, "set_factors: out of host memory");
    }
}

int mfsgd_get_factors(mfsgd_handle* h, float* P, float* Q) {
    if (!h) return MFSGD_ERR_INVALID_ARG;
    if (h->where == mfsgd_handle::Where::None) return fail(h, MFSGD_ERR_STATE, "get_factors: factors not initialised");
    try {
        const int k = h->cfg.k, kp = h->geo.kp;
        const std::vector<float>*sp = &h->hP, *sq = &h->hQ;
        std::vector<float> tp, tq;
        if (h->where == mfsgd_handle::Where::Device) {
            HIPCHK(h, hipSetDevice(h->cfg.device));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (P) {
                tp.resize((size_t)h->cfg.n_users * kp);
                HIPCHK(h, hipMemcpy(tp.data(), h->dP.p, tp.size() * sizeof(float), hipMemcpyDeviceToHost));
            }
            if (Q && h->n_parts == 1) {
                tq.resize((size_t)h->cfg.n_items * kp);
                HIPCHK(h, hipMemcpy(tq.data(), h->dQ.p, tq.size() * sizeof(float), hipMemcpyDeviceToHost));
            }
            sp = &tp;
            sq = &tq;
        }
        if (P)
            for (int64_t x = 0; x < h->cfg.n_users; ++x) std::memcpy(P + x * k, &(*sp)[(size_t)x * kp], sizeof(float) * (size_t)k);
        if (Q) {
            if (h->n_parts != 1) return fail(h, MFSGD_ERR_STATE, "get_factors: Q lives in caller-owned blocks when n_parts > 1");
            for (int64_t x = 0; x < h->cfg.n_items; ++x) std::memcpy(Q + x * k, &(*sq)[(size_t)x * kp], sizeof(float) * (size_t)k);
        }
        return MFSGD_OK;
    } catch (const std::bad_alloc&) {
        return fail(h, MFSGD_ERR_OOM, "get_factors: out of host memory");
    }
}

int mfsgd_train(mfsgd_handle* h, int32_t epochs, double* rmse_per_epoch) {
    if (!h || epochs < 0) return fail(h, MFSGD_ERR_INVALID_ARG, "train: bad argument");
    if (h->n_parts != 1) return fail(h, MFSGD_ERR_STATE, "train: handle has n_parts > 1, drive it with mfsgd_part_train");
    int rc = prepare_compute(h);
    if (rc) return rc;
    Part& p = h->parts[0];
    float* Q = static_cast<float*>(h->dQ.p);
    for (int e = 0; e < epochs; ++e) {
        if ((rc = launch_epoch(h, p, Q, h->stream))) return rc;
        if (rmse_per_epoch) {
            double sse = 0.0;
            if ((rc = settle_epochs_ok(h, p, Q, h->stream, 1))) return rc;
            if ((rc = part_sse_sync(h, p, Q, h->stream, &sse))) return rc;
            rmse_per_epoch[e] = p.sched.nnz > 0 ? std::sqrt(sse / (double)p.sched.nnz) : 0.0;
        }
    }
    return settle_epochs_ok(h, p, Q, h->stream, rmse_per_epoch ? 0 : epochs);
}

int mfsgd_train_timed(mfsgd_handle* h, int32_t epochs, double* elapsed_ms, int64_t* launches) {
    if (!h || epochs < 0 || !elapsed_ms) return fail(h, MFSGD_ERR_INVALID_ARG, "train_timed: bad argument");
    if (h->n_parts != 1) return fail(h, MFSGD_ERR_STATE, "train_timed: single-partition handles only");
    int rc = prepare_compute(h);
    if (rc) return rc;
    Part& p = h->parts[0];
    float* Q = static_cast<float*>(h->dQ.p);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    for (int e = 0; e < epochs; ++e)
        if ((rc = launch_epoch(h, p, Q, h->stream))) return rc;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *elapsed_ms = (double)ms;
    if (launches) *launches = p.sched.nnz > 0 ? (int64_t)epochs * (p.persistent_np > 0 ? 1 : p.sched.B) : 0;
    return check_abort_strict(h, p);  // a timing of launches that did nothing would be meaningless
}

int mfsgd_rmse(mfsgd_handle* h, double* out) {
    if (!h || !out) return fail(h, MFSGD_ERR_INVALID_ARG, "rmse: null argument");
    if (h->n_parts != 1) return fail(h, MFSGD_ERR_STATE, "rmse: handle has n_parts > 1, use mfsgd_part_sse");
    int rc = prepare_compute(h);
    if (rc) return rc;
    Part& p = h->parts[0];
    double sse = 0.0;
    if ((rc = part_sse_sync(h, p, static_cast<const float*>(h->dQ.p), h->stream, &sse))) return rc;
    *out = p.sched.nnz > 0 ? std::sqrt(sse / (double)p.sched.nnz) : 0.0;
    return MFSGD_OK;
}

int mfsgd_predict(mfsgd_handle* h, const int32_t* u, const int32_t* i, float* out, int64_t n) {
    if (!h || n < 0 || (n > 0 && (!u || !i || !out))) return fail(h, MFSGD_ERR_INVALID_ARG, "predict: bad argument");
    if (h->n_parts != 1) return fail(h, MFSGD_ERR_STATE, "predict: single-partition handles only");
    for (int64_t j = 0; j < n; ++j)
        if (u[j] < 0 || u[j] >= h->cfg.n_users || i[j] < 0 || i[j] >= h->cfg.n_items)
            return fail(h, MFSGD_ERR_INVALID_ARG, "predict: pair " + std::to_string(j) + " out of range");
    if (n == 0) return MFSGD_OK;
    int rc = factors_to_device(h);
    if (rc) return rc;
    DevBuf du, di, dout;
    auto cleanup = [&]() {
        du.release();
        di.release();
        dout.release();
    };
    rc = dev_alloc(h, du, sizeof(int32_t) * (size_t)n);
    if (!rc) rc = dev_alloc(h, di, sizeof(int32_t) * (size_t)n);
    if (!rc) rc = dev_alloc(h, dout, sizeof(float) * (size_t)n);
    if (rc) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemcpyAsync(du.p, u, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(di.p, i, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess)
        e = launch_predict(h->geo.L, static_cast<const float*>(h->dP.p), static_cast<const float*>(h->dQ.p),
                           static_cast<const int32_t*>(du.p), static_cast<const int32_t*>(di.p),
                           static_cast<float*>(dout.p), n, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) return fail(h, MFSGD_ERR_HIP, std::string("predict: ") + hipGetErrorString(e));
    return MFSGD_OK;
}

int mfsgd_recommend(mfsgd_handle* h, const int32_t* users, int32_t n_users, int32_t topn, int32_t* out_items,
                    float* out_scores) {
    if (!h || n_users < 0 || topn < 1 || (n_users > 0 && (!users || !out_items || !out_scores)))
        return fail(h, MFSGD_ERR_INVALID_ARG, "recommend: bad argument");
    if (h->n_parts != 1) return fail(h, MFSGD_ERR_STATE, "recommend: single-partition handles only");
    if (topn > h->cfg.n_items) return fail(h, MFSGD_ERR_INVALID_ARG, "recommend: topn exceeds the number of items");
    for (int32_t j = 0; j < n_users; ++j)
        if (users[j] < 0 || users[j] >= h->cfg.n_users)
            return fail(h, MFSGD_ERR_INVALID_ARG, "recommend: user " + std::to_string(j) + " out of range");
    if (n_users == 0) return MFSGD_OK;
    int rc = factors_to_device(h);
    if (rc) return rc;
    const int32_t I = h->cfg.n_items;
    const bool fused = recommend_is_fused(I, topn);  // score + select in one kernel, no score buffers
    // users per batch: about 64 M scores at a time (the sort path materialises them)
    int batch = (int)std::max<int64_t>(1, std::min<int64_t>(n_users, ((int64_t)64 << 20) / std::max(1, I)));
    if (fused) batch = n_users;
    batch = std::min(batch, 65535);
    DevBuf d_users, s_in, s_out, id_in, id_out, d_off, o_s, o_i;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    auto cleanup = [&]() {
        d_users.release(); s_in.release(); s_out.release(); id_in.release(); id_out.release(); d_off.release();
        o_s.release(); o_i.release();
        if (temp) (void)hipFree(temp);
    };
    const size_t cells = (size_t)batch * (size_t)I;
    rc = dev_alloc(h, d_users, sizeof(int32_t) * (size_t)batch);
    if (!fused) {
        if (!rc) rc = dev_alloc(h, s_in, 4 * cells);
        if (!rc) rc = dev_alloc(h, s_out, 4 * cells);
        if (!rc) rc = dev_alloc(h, id_in, 4 * cells);
        if (!rc) rc = dev_alloc(h, id_out, 4 * cells);
        if (!rc) rc = dev_alloc(h, d_off, sizeof(long long) * ((size_t)batch + 1));
    }
    if (!rc) rc = dev_alloc(h, o_s, 4 * (size_t)batch * topn);
    if (!rc) rc = dev_alloc(h, o_i, 4 * (size_t)batch * topn);
    if (rc) {
        cleanup();
        return rc;
    }
    hipError_t e = hipSuccess;
    for (int32_t done = 0; done < n_users && e == hipSuccess; done += batch) {
        const int nb = std::min<int32_t>(batch, n_users - done);
        e = hipMemcpyAsync(d_users.p, users + done, sizeof(int32_t) * (size_t)nb, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess && fused)
            e = recommend_fused(h->geo.L, static_cast<const float*>(h->dP.p), static_cast<const float*>(h->dQ.p),
                                static_cast<const int32_t*>(d_users.p), nb, I, topn, static_cast<float*>(o_s.p),
                                static_cast<int32_t*>(o_i.p), h->stream);
        else if (e == hipSuccess)
            e = recommend_batch(h->geo.L, static_cast<const float*>(h->dP.p), static_cast<const float*>(h->dQ.p),
                                static_cast<const int32_t*>(d_users.p), nb, I, topn, static_cast<float*>(s_in.p),
                                static_cast<float*>(s_out.p), static_cast<int32_t*>(id_in.p), static_cast<int32_t*>(id_out.p),
                                static_cast<long long*>(d_off.p), temp, temp_bytes, static_cast<float*>(o_s.p),
                                static_cast<int32_t*>(o_i.p), h->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(out_scores + (size_t)done * topn, o_s.p, 4 * (size_t)nb * topn, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(out_items + (size_t)done * topn, o_i.p, 4 * (size_t)nb * topn, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    cleanup();
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(h, e == hipErrorOutOfMemory ? MFSGD_ERR_OOM : MFSGD_ERR_HIP, std::string("recommend: ") + hipGetErrorString(e));
    }
    return MFSGD_OK;
}

int mfsgd_get_dims(const mfsgd_handle* h, int32_t* n_users, int32_t* n_items, int32_t* k) {
    if (!h) return MFSGD_ERR_INVALID_ARG;
    if (n_users) *n_users = h->cfg.n_users;
    if (n_items) *n_items = h->cfg.n_items;
    if (k) *k = h->cfg.k;
    return MFSGD_OK;
}

int mfsgd_get_schedule_info(const mfsgd_handle* h, int32_t part, mfsgd_schedule_info* out) {
    if (!h || !out) return fail(h, MFSGD_ERR_INVALID_ARG, "get_schedule_info: null argument");
    if (!h->have_ratings) return fail(h, MFSGD_ERR_STATE, "get_schedule_info: no ratings");
    if (part < 0 || part >= h->n_parts) return fail(h, MFSGD_ERR_INVALID_ARG, "get_schedule_info: bad partition");
    const Schedule& s = h->parts[(size_t)part].sched;
    std::memset(out, 0, sizeof *out);
    out->nnz = s.nnz;
    out->part = part;
    out->blocks = s.B;
    out->waves = s.W;
    out->group_lanes = s.geo.L;
    out->slots = s.geo.G;
    out->kp = s.geo.kp;
    out->rounds = s.B;
    out->lds_bytes = s.lds_bytes;
    out->total_steps = s.total_steps;
    out->total_rows = s.total_rows;
    out->max_cell_nnz = s.max_cell_nnz;
    out->max_cell_rows = s.max_cell_rows;
    out->max_cell_steps = s.max_cell_steps;
    out->sum_round_steps = s.sum_round_steps;
    out->build_seconds = s.build_seconds;
    out->swapped = h->parts[(size_t)part].swapped ? 1 : 0;
    out->device_ingest = s.device_packed ? 2 : s.device_ingest ? 1 : 0;
    out->chunks = (int64_t)s.cells.size();
    out->split_cells = s.split_cells;
    return MFSGD_OK;
}

int mfsgd_get_order(const mfsgd_handle* h, int32_t part, int64_t* order, int64_t* cell_ptr) {
    if (!h) return MFSGD_ERR_INVALID_ARG;
    if (!h->have_ratings) return fail(h, MFSGD_ERR_STATE, "get_order: no ratings");
    if (part < 0 || part >= h->n_parts) return fail(h, MFSGD_ERR_INVALID_ARG, "get_order: bad partition");
    if (order) {
        const int rc = host_copies(h, h->parts[(size_t)part], true, false);
        if (rc) return rc;
    }
    const Schedule& s = h->parts[(size_t)part].sched;
    if (order && !s.order.empty()) std::memcpy(order, s.order.data(), s.order.size() * sizeof(int64_t));
    if (cell_ptr) std::memcpy(cell_ptr, s.cell_ptr.data(), s.cell_ptr.size() * sizeof(int64_t));
    return MFSGD_OK;
}

int mfsgd_debug_schedule_sizes(const mfsgd_handle* h, int32_t part, int64_t* n_cells, int64_t* n_rows,
                               int64_t* n_subs, int64_t* n_entries) {
    if (!h || !n_cells || !n_rows || !n_subs || !n_entries) return fail(h, MFSGD_ERR_INVALID_ARG, "debug_schedule_sizes: null argument");
    if (!h->have_ratings) return fail(h, MFSGD_ERR_STATE, "debug_schedule_sizes: no ratings");
    if (part < 0 || part >= h->n_parts) return fail(h, MFSGD_ERR_INVALID_ARG, "debug_schedule_sizes: bad partition");
    const Schedule& s = h->parts[(size_t)part].sched;
    *n_cells = (int64_t)s.cells.size();
    *n_rows = s.n_rows_words;
    *n_subs = s.subs.empty() ? s.n_sub_recs : (int64_t)s.subs.size();
    *n_entries = s.n_entry_recs;
    return MFSGD_OK;
}

int mfsgd_debug_get_schedule(const mfsgd_handle* h, int32_t part, uint32_t* cells, uint32_t* rows, uint32_t* subs,
                             uint32_t* entries) {
    if (!h) return MFSGD_ERR_INVALID_ARG;
    if (!h->have_ratings) return fail(h, MFSGD_ERR_STATE, "debug_get_schedule: no ratings");
    if (part < 0 || part >= h->n_parts) return fail(h, MFSGD_ERR_INVALID_ARG, "debug_get_schedule: bad partition");
    if (rows || entries || subs) {
        const int rc = host_copies(h, h->parts[(size_t)part], false, true);
        if (rc) return rc;
    }
    const Schedule& s = h->parts[(size_t)part].sched;
    if (cells && !s.cells.empty()) std::memcpy(cells, s.cells.data(), s.cells.size() * sizeof(CellDesc));
    if (rows && !s.rows.empty()) std::memcpy(rows, s.rows.data(), s.rows.size() * sizeof(uint32_t));
    if (subs && !s.subs.empty()) std::memcpy(subs, s.subs.data(), s.subs.size() * sizeof(SubDesc));
    if (entries && !s.entries.empty()) std::memcpy(entries, s.entries.data(), s.entries.size() * sizeof(Entry));
    return MFSGD_OK;
}

int mfsgd_debug_epoch_profile(mfsgd_handle* h, uint64_t* out, int32_t* n_workgroups) {
    if (!h || !out || !n_workgroups) return fail(h, MFSGD_ERR_INVALID_ARG, "debug_epoch_profile: null argument");
    if (h->n_parts != 1) return fail(h, MFSGD_ERR_INVALID_ARG, "debug_epoch_profile: single-partition handles only");
    int rc = prepare_compute(h);
    if (rc) return rc;
    Part& p = h->parts[0];
    if ((rc = probe_persistent(h, p))) return rc;
    if (p.persistent_np <= 0) return fail(h, MFSGD_ERR_STATE, "debug_epoch_profile: persistent kernel not in use");
    const size_t words = (size_t)p.persistent_np * 16;
    if ((rc = dev_alloc(h, p.d_sse_partial, std::max(words * sizeof(uint64_t), sizeof(double) * p.sched.cells.size())))) return rc;
    CellLaunch a = make_launch(h, p, static_cast<float*>(h->dQ.p));
    a.grid = p.persistent_np;
    a.diag = true;
    a.sse_partial = static_cast<double*>(p.d_sse_partial.p);
    HIPCHK(h, hipMemsetAsync(p.d_sse_partial.p, 0, words * sizeof(uint64_t), h->stream));
    HIPCHK(h, launch_epoch_persistent(h->geo.L, p.sched.W, a, p.sched.B, static_cast<unsigned*>(p.d_sync.p), abort_word(p), h->stream));
    HIPCHK(h, hipMemcpyAsync(out, p.d_sse_partial.p, words * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *n_workgroups = p.persistent_np;
    return check_abort_strict(h, p);
}

int mfsgd_debug_counters(const mfsgd_handle* h, int64_t* out4) {
    if (!h || !out4) return MFSGD_ERR_INVALID_ARG;
    out4[0] = h->n_not_resident;
    out4[1] = out4[2] = 0;
    out4[3] = h->n_schedule_builds;
    for (const Part& p : h->parts) {
        out4[1] += p.persistent_np > 0 ? 1 : 0;  // partitions trained by the persistent kernel
        out4[2] += (int64_t)p.graphs.size();
    }
    return MFSGD_OK;
}

int mfsgd_debug_occupy(mfsgd_handle* h, int32_t milliseconds) {
    if (!h || milliseconds < 0 || milliseconds > 5000) return fail(h, MFSGD_ERR_INVALID_ARG, "debug_occupy: bad argument");
    int rc = ensure_device(h);
    if (rc) return rc;
    if (!h->side_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    // one workgroup on all but four CUs, each with (nearly) the whole LDS: nothing that needs LDS fits beside
    // it, and a persistent launch of more than a handful of workgroups finds only SOME of them resident
    // ... once they are ON the CUs: a launch on another stream is not ordered against what the caller launches
    // next, and a training launch that overtook this kernel met an empty chip (the not-resident test failed once in
    // five full runs that way).  The workgroups count themselves in a pinned host word; this call returns when all
    // have started (or after 0.2 s: a chip too busy to take them is occupied enough).
    if (!h->occupy_started) HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->occupy_started), sizeof(unsigned), hipHostMallocDefault));
    HIPCHK(h, hipStreamSynchronize(h->side_stream));  // (an earlier occupation has ended: the word is ours)
    *h->occupy_started = 0u;
    const int wgs = std::max(1, h->n_cu - 4);
    HIPCHK(h, launch_occupy(wgs, 160 * 1024 - 1024, (unsigned long long)milliseconds * 100000ull, h->occupy_started, h->side_stream));
    const auto t0 = std::chrono::steady_clock::now();
    while (__atomic_load_n(h->occupy_started, __ATOMIC_ACQUIRE) < (unsigned)wgs &&
           std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(200))
        std::this_thread::yield();
    return MFSGD_OK;
}

int mfsgd_debug_round_stamps(mfsgd_handle* h, int32_t part, int32_t round, uint64_t* out) {
    if (!h || !out) return fail(h, MFSGD_ERR_INVALID_ARG, "debug_round_stamps: null argument");
    if (h->n_parts != 1 || part != 0) return fail(h, MFSGD_ERR_INVALID_ARG, "debug_round_stamps: single-partition handles only");
    int rc = prepare_compute(h);
    if (rc) return rc;
    Part& p = h->parts[0];
    if (round < 0 || round >= p.sched.B) return fail(h, MFSGD_ERR_INVALID_ARG, "debug_round_stamps: bad round");
    const size_t words = (size_t)p.sched.B * (6 + (size_t)p.sched.W * p.sched.W * 4);
    if (words * sizeof(uint64_t) > p.d_sse_partial.bytes) {
        int rc2 = dev_alloc(h, p.d_sse_partial, words * sizeof(uint64_t));
        if (rc2) return rc2;
    }
    CellLaunch a = make_launch(h, p, static_cast<float*>(h->dQ.p));
    a.rd = round;
    a.diag = true;
    
    HIPCHK(h, hipMemsetAsync(p.d_sse_partial.p, 0, words * sizeof(uint64_t), h->stream));
    a.sse_partial = static_cast<double*>(p.d_sse_partial.p);
    HIPCHK(h, launch_cell(true, h->geo.L, p.sched.W, a, h->stream));
    HIPCHK(h, hipMemcpyAsync(out, p.d_sse_partial.p, words * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MFSGD_OK;
}

int mfsgd_dsgd_plan(const int64_t* deg_user, const int64_t* deg_item, int32_t n_users, int32_t n_items, int32_t n_parts,
                    int32_t* user_begin, int32_t* item_part) {
    if (!deg_user || !deg_item || !user_begin || !item_part || n_users < 1 || n_items < 1 || n_parts < 1)
        return MFSGD_ERR_INVALID_ARG;
    for (int32_t x = 0; x < n_users; ++x)
        if (deg_user[x] < 0) return MFSGD_ERR_INVALID_ARG;
    for (int32_t x = 0; x < n_items; ++x)
        if (deg_item[x] < 0) return MFSGD_ERR_INVALID_ARG;
    try {
        dsgd_plan(deg_user, deg_item, n_users, n_items, n_parts, user_begin, item_part);
    } catch (const std::bad_alloc&) {
        return MFSGD_ERR_OOM;
    }
    return MFSGD_OK;
}

int mfsgd_dsgd_plan_ex(const int64_t* deg_user, const int64_t* deg_item, int32_t n_users, int32_t n_items, int32_t world,
                       int32_t parts_per_rank, int32_t k, float chain_crit, int32_t* user_begin, int32_t* item_part,
                       int64_t* info4) {
    if (!deg_user || !deg_item || !user_begin || !item_part || n_users < 1 || n_items < 1 || world < 1 || parts_per_rank < 1 ||
        k < 1 || k > MFSGD_MAX_K || (int64_t)world * parts_per_rank > INT32_MAX || !(chain_crit >= 0.f))
        return MFSGD_ERR_INVALID_ARG;
    for (int32_t x = 0; x < n_users; ++x)
        if (deg_user[x] < 0) return MFSGD_ERR_INVALID_ARG;
    for (int32_t x = 0; x < n_items; ++x)
        if (deg_item[x] < 0) return MFSGD_ERR_INVALID_ARG;
    try {
        dsgd_plan_users(deg_user, n_users, world, user_begin);
        dsgd_plan_items(deg_item, n_items, world * parts_per_rank, world, k, (double)chain_crit, item_part, info4);
    } catch (const std::bad_alloc&) {
        return MFSGD_ERR_OOM;
    }
    return MFSGD_OK;
}

int mfsgd_set_item_partition(mfsgd_handle* h, const int32_t* item_part) {
    if (!h) return MFSGD_ERR_INVALID_ARG;
    if (h->n_parts <= 1) return fail(h, MFSGD_ERR_STATE, "set_item_partition: handle has a single partition");
    if (h->have_ratings) return fail(h, MFSGD_ERR_STATE, "set_item_partition: call it before mfsgd_set_ratings");
    try {
        if (!item_part) {
            default_item_map(h);
            return MFSGD_OK;
        }
        const int G = h->n_parts;
        const int32_t I = h->cfg.n_items;
        for (int32_t x = 0; x < I; ++x)
            if (item_part[x] < 0 || item_part[x] >= G)
                return fail(h, MFSGD_ERR_INVALID_ARG, "set_item_partition: item " + std::to_string(x) + " has partition " +
                                                          std::to_string(item_part[x]) + " out of range");
        h->item_part.assign(item_part, item_part + I);
        h->item_row.assign((size_t)I, 0);
        h->part_q_rows.assign((size_t)G, 0);
        for (int32_t x = 0; x < I; ++x) h->item_row[(size_t)x] = h->part_q_rows[(size_t)item_part[x]]++;
        h->custom_item_map = true;
        return MFSGD_OK;
    } catch (const std::bad_alloc&) {
        return fail(h, MFSGD_ERR_OOM, "set_item_partition: out of host memory");
    }
}

int mfsgd_get_item_partition(const mfsgd_handle* h, int32_t* item_part, int32_t* item_row) {
    if (!h) return MFSGD_ERR_INVALID_ARG;
    if (h->n_parts <= 1) return fail(h, MFSGD_ERR_STATE, "get_item_partition: handle has a single partition");
    if (item_part) std::memcpy(item_part, h->item_part.data(), h->item_part.size() * sizeof(int32_t));
    if (item_row) std::memcpy(item_row, h->item_row.data(), h->item_row.size() * sizeof(int32_t));
    return MFSGD_OK;
}

int mfsgd_part_rows(const mfsgd_handle* h, int32_t part, int32_t* rows) {
    if (!h || !rows) return fail(h, MFSGD_ERR_INVALID_ARG, "part_rows: null argument");
    if (part < 0 || part >= h->n_parts) return fail(h, MFSGD_ERR_INVALID_ARG, "part_rows: bad partition");
    *rows = h->n_parts > 1 ? h->part_q_rows[(size_t)part] : h->cfg.n_items;
    return MFSGD_OK;
}

int mfsgd_part_init_q(const mfsgd_handle* h, int32_t part, int64_t seed, int64_t u_total, float* q_block_host) {
    if (!h || !q_block_host || u_total < 0) return fail(h, MFSGD_ERR_INVALID_ARG, "part_init_q: bad argument");
    if (part < 0 || part >= h->n_parts) return fail(h, MFSGD_ERR_INVALID_ARG, "part_init_q: bad partition");
    const int k = h->cfg.k, kp = h->geo.kp;
    const float scale = (float)(1.0 / std::sqrt((double)k));
    // items of this partition in ascending id order = ascending row order; the stream position of
    // item i is (u_total + i) * k
    JRandom g(seed);
    int64_t pos = 0;  // floats drawn so far
    for (int32_t x = 0; x < h->cfg.n_items; ++x) {
        if (h->n_parts > 1 && h->item_part[(size_t)x] != part) continue;
        const int64_t want = ((int64_t)u_total + x) * k;
        g.skip((uint64_t)(want - pos));
        float* row = q_block_host + (size_t)(h->n_parts > 1 ? h->item_row[(size_t)x] : x) * kp;
        for (int f = 0; f < k; ++f) row[f] = g.nextFloat() * scale;
        for (int f = k; f < kp; ++f) row[f] = 0.0f;
        pos = want + k;
    }
    return MFSGD_OK;
}

int mfsgd_part_train(mfsgd_handle* h, int32_t part, float* q_block_dev, void* stream) {
    if (!h || !q_block_dev) return fail(h, MFSGD_ERR_INVALID_ARG, "part_train: null argument");
    if (part < 0 || part >= h->n_parts) return fail(h, MFSGD_ERR_INVALID_ARG, "part_train: bad partition");
    int rc = prepare_compute(h);
    if (rc) return rc;
    // the caller owns the Q block, so the caller names the stream (NULL = HIP's null stream)
    return launch_epoch(h, h->parts[(size_t)part], q_block_dev, static_cast<hipStream_t>(stream));
}

int mfsgd_part_settle(mfsgd_handle* h, int32_t part, float* q_block_dev, void* stream, int32_t* rerun) {
    if (rerun) *rerun = 0;
    if (!h || !q_block_dev) return fail(h, MFSGD_ERR_INVALID_ARG, "part_settle: null argument");
    if (part < 0 || part >= h->n_parts) return fail(h, MFSGD_ERR_INVALID_ARG, "part_settle: bad partition");
    if (!h->device_ready || !h->have_ratings) return MFSGD_OK;  // nothing can have been launched
    HIPCHK(h, hipSetDevice(h->cfg.device));
    Part& p = h->parts[(size_t)part];
    if (p.sched.nnz == 0) {
        HIPCHK(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
        return MFSGD_OK;
    }
    const int rc = settle_epochs(h, p, q_block_dev, static_cast<hipStream_t>(stream), 1);
    if (rc == kNotResident) {
        if (rerun) *rerun = 1;
        return MFSGD_OK;
    }
    return rc;
}

int mfsgd_part_sync(mfsgd_handle* h, int32_t part, void* stream) {
    if (!h) return MFSGD_ERR_INVALID_ARG;
    if (part < 0 || part >= h->n_parts) return fail(h, MFSGD_ERR_INVALID_ARG, "part_sync: bad partition");
    if (!h->device_ready || !h->have_ratings) return MFSGD_OK;  // nothing can have been launched
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return check_abort_strict(h, h->parts[(size_t)part]);
}

int mfsgd_get_parts(const mfsgd_handle* h, int32_t* n_parts, int32_t* kp, int32_t* device) {
    if (!h) return MFSGD_ERR_INVALID_ARG;
    if (n_parts) *n_parts = h->n_parts;
    if (kp) *kp = h->geo.kp;
    if (device) *device = h->cfg.device;
    return MFSGD_OK;
}

int mfsgd_part_sse(mfsgd_handle* h, int32_t part, const float* q_block_dev, void* stream, double* sse) {
    if (!h || !q_block_dev || !sse) return fail(h, MFSGD_ERR_INVALID_ARG, "part_sse: null argument");
    if (part < 0 || part >= h->n_parts) return fail(h, MFSGD_ERR_INVALID_ARG, "part_sse: bad partition");
    int rc = prepare_compute(h);
    if (rc) return rc;
    return part_sse_sync(h, h->parts[(size_t)part], q_block_dev, static_cast<hipStream_t>(stream), sse);
}

}  // extern "C"
