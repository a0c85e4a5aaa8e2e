// dsgd.cpp -- DSGD over the GPUs of one node, under the C-ABI (mfsgd_dsgd_*, include/mfsgd.h).
//
// No reference counterpart exists (/root/reference/README.md:1-2 is the whole reference); this is
// SURVEY.md section 3's call stack (train -> kernel -> ncclGroupStart / ncclSend / ncclRecv /
// ncclGroupEnd) and section 8e: one process per GPU, rank g keeps the P rows of its users for the
// whole run, the item-factor blocks travel along a ring -- one point-to-point message per block and
// sub-epoch over xGMI, no all-to-all, no data-path all-reduce; RMSE is one 2-double all-reduce.
//
// A pure client of the library's own public entry points (mfsgd_part_*), HIP streams / events and
// RCCL.  RCCL is bound at run time (dlopen of librccl.so.1 on first use): single-GPU hosts never
// load it, and a process that already holds a copy (PyTorch bundles one) shares it by SONAME.
//
// Streams: the partitions of a rank's group are trained one after another on the compute stream
// (they update the same P rows); block j leaves on the communication stream as soon as ITS training
// has finished -- while block j + 1 is being trained -- and the next sub-epoch's training of slot j
// waits for slot j's arrival only.  With one partition per rank that is train -> shift -> train with
// no host involvement; with m > 1 the shifts hide behind the training of the other blocks.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mfsgd.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {std::getenv("MFSGD_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
            r.why = dlerror();
        }
        if (!r.lib) return;
        auto sym = [&](const char* s) {
            void* p = dlsym(r.lib, s);
            if (!p) r.why = std::string("librccl lacks ") + s;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv ||
            !r.AllReduce || !r.GetErrorString) {
            dlclose(r.lib);
            r.lib = nullptr;
        }
    });
    return r;
}

thread_local std::string g_dsgd_error;

// ---- the rehearsal transport: blocks staged through POSIX shared memory ------------------------------
// [r3] NOT in the product library: compiled only with -DMFSGD_DSGD_REHEARSAL, i.e. into lib/libmfsgd_rehearsal.so (the
// Makefile builds it beside libmfsgd.so from the same sources), which the multi-process tests and bench.py
// --rehearse-on-one-gpu load through MFSGD_LIBRARY.  libmfsgd.so moves blocks with RCCL or not at all.
// RCCL cannot put two ranks on one GPU, and a host without RCCL has no ring at all.  With
// MFSGD_DSGD_TRANSPORT=shm mfsgd_dsgd_unique_id() hands out the name of a shared-memory segment instead of
// an RCCL id, and a ring created from such an id moves its blocks device -> segment -> device with blocking
// copies and sequence counters.  Same ring, same order of events, no xGMI: it exists so that the multi-rank
// logic of this file (groups, slots, double buffering, event ordering, the RMSE reduction) runs -- and is
// tested -- with several REAL processes on one GPU.  Not a performance path.
constexpr char kShmMagic[8] = {'M', 'F', 'S', 'G', 'D', 'S', 'H', 'M'};  // an id that names a segment, not an RCCL id
#ifdef MFSGD_DSGD_REHEARSAL
constexpr int kShmMaxWorld = 16, kShmMaxSlots = 64;
struct ShmHeader {
    std::atomic<uint32_t> ready[kShmMaxWorld];
    std::atomic<uint64_t> written[kShmMaxWorld][kShmMaxSlots];  // channel (rank -> rank - 1, slot): blocks written
    std::atomic<uint64_t> taken[kShmMaxWorld][kShmMaxSlots];    // ... and taken by the receiver
    std::atomic<uint64_t> ar_seq[kShmMaxWorld], ar_done[kShmMaxWorld];
    double ar_val[kShmMaxWorld][2];
};
constexpr size_t kShmDataOffset = (sizeof(ShmHeader) + 4095) & ~(size_t)4095;

struct ShmRing {
    std::string name;
    int fd = -1;
    unsigned char* base = nullptr;
    size_t bytes = 0, slot_bytes = 0;
    int rank = 0, world = 1, m = 1;
    uint64_t ar_round = 0;
    ShmHeader* hdr() const { return reinterpret_cast<ShmHeader*>(base); }
    unsigned char* slot(int r, int j) const { return base + kShmDataOffset + ((size_t)r * m + j) * slot_bytes; }
};

template <class Pred>
bool spin_until(Pred ok, double seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned n = 0; !ok(); ++n) {
        if ((n & 1023u) == 1023u) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
            std::this_thread::yield();
        }
    }
    return true;
}
#else
struct ShmRing;  // (the product library has no such transport: the pointer below stays null)
#endif  // MFSGD_DSGD_REHEARSAL

}  // namespace

struct mfsgd_dsgd {
    mfsgd_handle* h = nullptr;
    int rank = 0, world = 1, m = 1;  // m: partitions a rank holds at a time (its "group")
    int n_parts = 1, kp = 0, k = 0, device = 0;
    int32_t max_rows = 0;
    int64_t nnz_local = 0;
    ncclComm_t comm = nullptr;
    ShmRing* shm = nullptr;  // the rehearsal transport instead of RCCL (MFSGD_DSGD_TRANSPORT=shm)
    hipStream_t compute = nullptr, wire = nullptr;
    float* buf[2] = {nullptr, nullptr};  // [cur, nxt]: m blocks of max_rows x kp floats each
    int cur = 0;
    int group = 0;  // group currently held: partitions group * m .. group * m + m - 1
    std::vector<hipEvent_t> trained, arrived;  // per slot j
    double* d_red = nullptr;                   // 2 doubles for the RMSE all-reduce
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // counters (mfsgd_dsgd_stats): sub-epoch trainings enqueued, of those with the recovery point, of those re-run
    // as round launches, bytes sent
    int64_t n_trained = 0, n_checked = 0, n_rerun = 0, bytes_sent = 0;
    bool always_check = false;  // MFSGD_DSGD_CHECK=1: the recovery point in every sub-epoch
    std::string err;

    float* block(int which, int j) const { return buf[which] + (size_t)j * max_rows * kp; }
    int part(int j) const { return group * m + j; }
};

namespace {

int dfail(mfsgd_dsgd* d, int code, const std::string& msg) {
    if (d) d->err = msg;
    else g_dsgd_error = msg;
    return code;
}

#define DHIP(d, call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return dfail((d), e_ == hipErrorOutOfMemory ? MFSGD_ERR_OOM : MFSGD_ERR_HIP,                \
                         std::string(#call) + ": " + hipGetErrorString(e_));                            \
    } while (0)
#define DNCCL(d, call)                                                                                  \
    do {                                                                                                \
        ncclResult_t r_ = (call);                                                                       \
        if (r_ != ncclSuccess) return dfail((d), MFSGD_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(r_)); \
    } while (0)
#define DLIB(d, call)                                                                                   \
    do {                                                                                                \
        int rc_ = (call);                                                                               \
        if (rc_ != MFSGD_OK) return dfail((d), rc_, std::string(#call) + ": " + mfsgd_last_error((d)->h)); \
    } while (0)

// One ring shift of slot j: the block goes to rank - 1, the next one arrives from rank + 1.
int shift_slot(mfsgd_dsgd* d, int j, bool after_training) {
    const size_t count = (size_t)d->max_rows * d->kp;
#ifdef MFSGD_DSGD_REHEARSAL
    if (d->shm) {
        // rehearsal transport: the same shift with blocking copies through shared memory
        ShmRing& r = *d->shm;
        ShmHeader* H = r.hdr();
        const int src = (d->rank + 1) % d->world;
        if (after_training) DHIP(d, hipEventSynchronize(d->trained[(size_t)j]));
        auto& wr = H->written[d->rank][j];
        auto& tk = H->taken[d->rank][j];
        if (!spin_until([&] { return tk.load(std::memory_order_acquire) == wr.load(std::memory_order_relaxed); }, 120.0))
            return dfail(d, MFSGD_ERR_HIP, "dsgd (shm transport): rank " + std::to_string((d->rank + d->world - 1) % d->world) + " never took the last block");
        DHIP(d, hipMemcpy(r.slot(d->rank, j), d->block(d->cur, j), count * sizeof(float), hipMemcpyDeviceToHost));
        d->bytes_sent += (int64_t)(count * sizeof(float));
        wr.store(wr.load(std::memory_order_relaxed) + 1, std::memory_order_release);
        auto& swr = H->written[src][j];
        auto& stk = H->taken[src][j];
        if (!spin_until([&] { return swr.load(std::memory_order_acquire) > stk.load(std::memory_order_relaxed); }, 120.0))
            return dfail(d, MFSGD_ERR_HIP, "dsgd (shm transport): rank " + std::to_string(src) + " never sent its block");
        DHIP(d, hipMemcpy(d->block(d->cur ^ 1, j), r.slot(src, j), count * sizeof(float), hipMemcpyHostToDevice));
        stk.store(stk.load(std::memory_order_relaxed) + 1, std::memory_order_release);
        DHIP(d, hipEventRecord(d->arrived[(size_t)j], d->wire));
        return MFSGD_OK;
    }
#endif  // MFSGD_DSGD_REHEARSAL
    Rccl& R = rccl();
    if (after_training) DHIP(d, hipStreamWaitEvent(d->wire, d->trained[(size_t)j], 0));
    DNCCL(d, R.GroupStart());
    DNCCL(d, R.Send(d->block(d->cur, j), count, ncclFloat, (d->rank + d->world - 1) % d->world, d->comm, d->wire));
    DNCCL(d, R.Recv(d->block(d->cur ^ 1, j), count, ncclFloat, (d->rank + 1) % d->world, d->comm, d->wire));
    DNCCL(d, R.GroupEnd());
    DHIP(d, hipEventRecord(d->arrived[(size_t)j], d->wire));
    d->bytes_sent += (int64_t)(count * sizeof(float));
    return MFSGD_OK;
}

void rotated(mfsgd_dsgd* d) {
    d->cur ^= 1;
    d->group = (d->group + 1) % d->world;
}

// One epoch: `world` sub-epochs of train-the-group, pass-it-on.  Asynchronous -- no host synchronisation inside --
// unless `checked`: then every block passes the recovery point (mfsgd_part_settle) before it leaves.  A persistent
// training launch that finds its workgroups not co-resident (an exchange of the ring, or a foreign kernel, holds CUs)
// changes nothing; unnoticed, the untrained block would travel on and the ranks' factors diverge for good.  At the
// recovery point the host waits for the training, and a launch that gave up is repeated as round launches before the
// block is sent.  Cost: the launch latencies of one sub-epoch, exposed once per sub-epoch.  When it is on:
//  - the first epoch of every train call: what a launch meets on this node shows there (the partition then stays on
//    round launches, so later epochs cannot fail the same way);
//  - every epoch when a rank holds several partitions (m > 1): slot j's exchange is in flight while slot j + 1 is
//    trained, for the whole run, and how long a peer keeps an exchange waiting is not this rank's to know;
//  - every epoch under MFSGD_DSGD_CHECK=1.
// With m = 1 nothing of the ring runs beside a training launch (train -> shift -> train), so the later epochs go
// unchecked; a launch that still gives up (a foreign process) surfaces in finish() as "factors invalid".
int enqueue_epoch(mfsgd_dsgd* d, bool checked) {
    for (int s = 0; s < d->world; ++s) {
        for (int j = 0; j < d->m; ++j) {
            DHIP(d, hipStreamWaitEvent(d->compute, d->arrived[(size_t)j], 0));
            DLIB(d, mfsgd_part_train(d->h, d->part(j), d->block(d->cur, j), d->compute));
            d->n_trained++;
            if (checked) {
                int32_t rerun = 0;
                DLIB(d, mfsgd_part_settle(d->h, d->part(j), d->block(d->cur, j), d->compute, &rerun));
                d->n_checked++;
                d->n_rerun += rerun;
            }
            DHIP(d, hipEventRecord(d->trained[(size_t)j], d->compute));
            int rc = shift_slot(d, j, true);
            if (rc) return rc;
        }
        rotated(d);
    }
    return MFSGD_OK;
}

int finish(mfsgd_dsgd* d) {
    DHIP(d, hipStreamSynchronize(d->compute));
    DHIP(d, hipStreamSynchronize(d->wire));
    for (int p = 0; p < d->n_parts; ++p) DLIB(d, mfsgd_part_sync(d->h, p, d->compute));
    return MFSGD_OK;
}

// Sum of squared errors of this rank's ratings: one read-only rotation (blocks come home again).
int local_sse(mfsgd_dsgd* d, double* out) {
    int rc = finish(d);
    if (rc) return rc;
    double total = 0.0;
    for (int s = 0; s < d->world; ++s) {
        for (int j = 0; j < d->m; ++j) {
            DHIP(d, hipStreamWaitEvent(d->compute, d->arrived[(size_t)j], 0));
            double sse = 0.0;
            DLIB(d, mfsgd_part_sse(d->h, d->part(j), d->block(d->cur, j), d->compute, &sse));  // synchronous
            total += sse;
            DHIP(d, hipEventRecord(d->trained[(size_t)j], d->compute));
            rc = shift_slot(d, j, true);
            if (rc) return rc;
        }
        rotated(d);
    }
    DHIP(d, hipStreamSynchronize(d->wire));
    *out = total;
    return MFSGD_OK;
}

int allreduce2(mfsgd_dsgd* d, double* v, ncclRedOp_t op) {
#ifdef MFSGD_DSGD_REHEARSAL
    if (d->shm) {
        ShmRing& r = *d->shm;
        ShmHeader* H = r.hdr();
        const uint64_t q = ++r.ar_round;
        // nobody may still be reading the previous round's values
        if (!spin_until([&] {
                for (int x = 0; x < d->world; ++x)
                    if (H->ar_done[x].load(std::memory_order_acquire) + 1 < q) return false;
                return true;
            }, 120.0))
            return dfail(d, MFSGD_ERR_HIP, "dsgd (shm transport): all-reduce, a rank is missing");
        H->ar_val[d->rank][0] = v[0];
        H->ar_val[d->rank][1] = v[1];
        H->ar_seq[d->rank].store(q, std::memory_order_release);
        if (!spin_until([&] {
                for (int x = 0; x < d->world; ++x)
                    if (H->ar_seq[x].load(std::memory_order_acquire) < q) return false;
                return true;
            }, 120.0))
            return dfail(d, MFSGD_ERR_HIP, "dsgd (shm transport): all-reduce, a rank is missing");
        double a = H->ar_val[0][0], b = H->ar_val[0][1];
        for (int x = 1; x < d->world; ++x) {  // rank order: every rank gets the same bits
            a = op == ncclSum ? a + H->ar_val[x][0] : std::max(a, H->ar_val[x][0]);
            b = op == ncclSum ? b + H->ar_val[x][1] : std::max(b, H->ar_val[x][1]);
        }
        v[0] = a;
        v[1] = b;
        H->ar_done[d->rank].store(q, std::memory_order_release);
        return MFSGD_OK;
    }
#endif  // MFSGD_DSGD_REHEARSAL
    DHIP(d, hipMemcpyAsync(d->d_red, v, 2 * sizeof(double), hipMemcpyHostToDevice, d->wire));
    DNCCL(d, rccl().AllReduce(d->d_red, d->d_red, 2, ncclDouble, op, d->comm, d->wire));
    DHIP(d, hipMemcpyAsync(v, d->d_red, 2 * sizeof(double), hipMemcpyDeviceToHost, d->wire));
    DHIP(d, hipStreamSynchronize(d->wire));
    return MFSGD_OK;
}

}  // namespace

extern "C" {

const char* mfsgd_dsgd_last_error(const mfsgd_dsgd* d) { return d ? d->err.c_str() : g_dsgd_error.c_str(); }

int mfsgd_dsgd_unique_id(void* id_out) {
    if (!id_out) return dfail(nullptr, MFSGD_ERR_INVALID_ARG, "dsgd_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) <= MFSGD_DSGD_ID_BYTES, "id buffer");
#ifdef MFSGD_DSGD_REHEARSAL
    if (const char* tr = std::getenv("MFSGD_DSGD_TRANSPORT"))
        if (std::strcmp(tr, "shm") == 0) {
            // the rehearsal transport: the id is the name of a shared-memory segment
            std::memset(id_out, 0, MFSGD_DSGD_ID_BYTES);
            std::memcpy(id_out, kShmMagic, sizeof kShmMagic);
            const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
            std::snprintf(static_cast<char*>(id_out) + 8, MFSGD_DSGD_ID_BYTES - 8, "/mfsgd_%d_%llx", (int)getpid(), (unsigned long long)now);
            return MFSGD_OK;
        }
#else
    if (const char* tr = std::getenv("MFSGD_DSGD_TRANSPORT"))
        if (std::strcmp(tr, "shm") == 0)
            return dfail(nullptr, MFSGD_ERR_UNSUPPORTED, "MFSGD_DSGD_TRANSPORT=shm: this library has no rehearsal transport; load lib/libmfsgd_rehearsal.so (MFSGD_LIBRARY)");
#endif
    Rccl& R = rccl();
    if (!R.lib) return dfail(nullptr, MFSGD_ERR_UNSUPPORTED, "RCCL is not available: " + R.why);
    ncclUniqueId id;
    ncclResult_t r = R.GetUniqueId(&id);
    if (r != ncclSuccess) return dfail(nullptr, MFSGD_ERR_HIP, std::string("ncclGetUniqueId: ") + R.GetErrorString(r));
    std::memset(id_out, 0, MFSGD_DSGD_ID_BYTES);
    std::memcpy(id_out, &id, sizeof id);
    return MFSGD_OK;
}

int mfsgd_dsgd_create(mfsgd_handle* h, int32_t rank, int32_t world, const void* id, mfsgd_dsgd** out) {
    if (out) *out = nullptr;
    if (!h || !id || !out || world < 1 || rank < 0 || rank >= world)
        return dfail(nullptr, MFSGD_ERR_INVALID_ARG, "dsgd_create: bad argument");
    const bool use_shm = std::memcmp(id, kShmMagic, sizeof kShmMagic) == 0;
    Rccl& R = rccl();
    if (!use_shm && !R.lib) return dfail(nullptr, MFSGD_ERR_UNSUPPORTED, "RCCL is not available: " + R.why);
#ifdef MFSGD_DSGD_REHEARSAL
    if (use_shm && world > kShmMaxWorld) return dfail(nullptr, MFSGD_ERR_UNSUPPORTED, "dsgd (shm transport): at most 16 ranks");
#else
    if (use_shm)
        return dfail(nullptr, MFSGD_ERR_UNSUPPORTED, "dsgd_create: the id names a shared-memory segment, and this library has no rehearsal transport; "
                                                       "load lib/libmfsgd_rehearsal.so (MFSGD_LIBRARY)");
#endif
    int32_t n_parts = 0, kp = 0, device = 0, k = 0;
    if (mfsgd_get_parts(h, &n_parts, &kp, &device) != MFSGD_OK || mfsgd_get_dims(h, nullptr, nullptr, &k) != MFSGD_OK)
        return dfail(nullptr, MFSGD_ERR_INVALID_ARG, "dsgd_create: bad handle");
    if (n_parts < 2 && world > 1) return dfail(nullptr, MFSGD_ERR_STATE, "dsgd_create: the handle was created with n_parts <= 1");
    if (n_parts % world != 0)
        return dfail(nullptr, MFSGD_ERR_INVALID_ARG, "dsgd_create: n_parts (" + std::to_string(n_parts) + ") is not a multiple of world (" +
                                                         std::to_string(world) + ")");
    mfsgd_dsgd* d = new (std::nothrow) mfsgd_dsgd();
    if (!d) return dfail(nullptr, MFSGD_ERR_OOM, "dsgd_create: out of host memory");
    auto bail = [&](int rc) {
        g_dsgd_error = d->err;
        mfsgd_dsgd_destroy(d);
        return rc;
    };
    d->h = h;
    d->rank = rank;
    d->world = world;
    d->n_parts = n_parts;
    d->m = n_parts / world;
    d->kp = kp;
    d->k = k;
    d->device = device;
    d->group = rank;
    if (const char* c = std::getenv("MFSGD_DSGD_CHECK")) d->always_check = std::atoi(c) != 0;
    for (int p = 0; p < n_parts; ++p) {
        int32_t rows = 0;
        mfsgd_schedule_info info;
        if (mfsgd_part_rows(h, p, &rows) != MFSGD_OK || mfsgd_get_schedule_info(h, p, &info) != MFSGD_OK) {
            d->err = std::string("dsgd_create: ") + mfsgd_last_error(h) + " (set the ratings first)";
            return bail(MFSGD_ERR_STATE);
        }
        d->max_rows = std::max(d->max_rows, rows);
        d->nnz_local += info.nnz;
    }
    if (d->max_rows < 1) d->max_rows = 1;
    auto hip = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return false;
        d->err = std::string("dsgd_create: ") + what + ": " + hipGetErrorString(e);
        return true;
    };
    if (hip(hipSetDevice(device), "hipSetDevice")) return bail(MFSGD_ERR_NO_DEVICE);
    if (hip(hipStreamCreateWithFlags(&d->compute, hipStreamNonBlocking), "hipStreamCreate") ||
        hip(hipStreamCreateWithFlags(&d->wire, hipStreamNonBlocking), "hipStreamCreate") ||
        hip(hipEventCreateWithFlags(&d->ev0, hipEventDefault), "hipEventCreate") ||
        hip(hipEventCreateWithFlags(&d->ev1, hipEventDefault), "hipEventCreate"))
        return bail(MFSGD_ERR_HIP);
    const size_t bytes = (size_t)d->m * d->max_rows * kp * sizeof(float);
    for (int b = 0; b < 2; ++b) {
        if (hip(hipMalloc(reinterpret_cast<void**>(&d->buf[b]), bytes), "hipMalloc(Q blocks)")) return bail(MFSGD_ERR_OOM);
        if (hip(hipMemset(d->buf[b], 0, bytes), "hipMemset")) return bail(MFSGD_ERR_HIP);
    }
    if (hip(hipMalloc(reinterpret_cast<void**>(&d->d_red), 2 * sizeof(double)), "hipMalloc")) return bail(MFSGD_ERR_OOM);
    d->trained.assign((size_t)d->m, nullptr);
    d->arrived.assign((size_t)d->m, nullptr);
    for (int j = 0; j < d->m; ++j)
        if (hip(hipEventCreateWithFlags(&d->trained[(size_t)j], hipEventDisableTiming), "hipEventCreate") ||
            hip(hipEventCreateWithFlags(&d->arrived[(size_t)j], hipEventDisableTiming), "hipEventCreate"))
            return bail(MFSGD_ERR_HIP);
#ifdef MFSGD_DSGD_REHEARSAL
    if (use_shm) {
        if (d->m > kShmMaxSlots) {
            d->err = "dsgd_create (shm transport): at most 64 partitions per rank";
            return bail(MFSGD_ERR_UNSUPPORTED);
        }
        ShmRing* sr = new (std::nothrow) ShmRing();
        if (!sr) return bail(MFSGD_ERR_OOM);
        d->shm = sr;
        sr->name.assign(static_cast<const char*>(id) + 8, strnlen(static_cast<const char*>(id) + 8, MFSGD_DSGD_ID_BYTES - 9));
        sr->rank = rank;
        sr->world = world;
        sr->m = d->m;
        sr->slot_bytes = (size_t)d->max_rows * kp * sizeof(float);
        sr->bytes = kShmDataOffset + (size_t)world * d->m * sr->slot_bytes;
        sr->fd = shm_open(sr->name.c_str(), O_CREAT | O_RDWR, 0600);
        if (sr->fd < 0 || ftruncate(sr->fd, (off_t)sr->bytes) != 0) {
            d->err = "dsgd_create (shm transport): cannot create " + sr->name;
            return bail(MFSGD_ERR_OOM);
        }
        void* mp = mmap(nullptr, sr->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, sr->fd, 0);
        if (mp == MAP_FAILED) {
            d->err = "dsgd_create (shm transport): cannot map " + sr->name;
            return bail(MFSGD_ERR_OOM);
        }
        sr->base = static_cast<unsigned char*>(mp);  // a fresh segment is all zeros: every counter starts at 0
        ShmHeader* H = sr->hdr();
        H->ready[rank].store(1, std::memory_order_release);
        if (!spin_until([&] {
                for (int x = 0; x < world; ++x)
                    if (H->ready[x].load(std::memory_order_acquire) == 0) return false;
                return true;
            }, 120.0)) {
            d->err = "dsgd_create (shm transport): not every rank arrived";
            return bail(MFSGD_ERR_HIP);
        }
        *out = d;
        return MFSGD_OK;
    }
#endif  // MFSGD_DSGD_REHEARSAL
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclResult_t r = R.CommInitRank(&d->comm, world, uid, rank);
    if (r != ncclSuccess) {
        d->comm = nullptr;
        d->err = std::string("dsgd_create: ncclCommInitRank: ") + R.GetErrorString(r);
        return bail(MFSGD_ERR_HIP);
    }
    *out = d;
    return MFSGD_OK;
}

void mfsgd_dsgd_destroy(mfsgd_dsgd* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->compute) (void)hipStreamSynchronize(d->compute);
    if (d->wire) (void)hipStreamSynchronize(d->wire);
    if (d->comm) (void)rccl().CommDestroy(d->comm);
#ifdef MFSGD_DSGD_REHEARSAL
    if (d->shm) {
        if (d->shm->base) (void)munmap(d->shm->base, d->shm->bytes);
        if (d->shm->fd >= 0) (void)close(d->shm->fd);
        if (d->rank == 0 && !d->shm->name.empty()) (void)shm_unlink(d->shm->name.c_str());
        delete d->shm;
    }
#endif  // MFSGD_DSGD_REHEARSAL
    for (hipEvent_t e : d->trained)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : d->arrived)
        if (e) (void)hipEventDestroy(e);
    if (d->ev0) (void)hipEventDestroy(d->ev0);
    if (d->ev1) (void)hipEventDestroy(d->ev1);
    for (int b = 0; b < 2; ++b)
        if (d->buf[b]) (void)hipFree(d->buf[b]);
    if (d->d_red) (void)hipFree(d->d_red);
    if (d->compute) (void)hipStreamDestroy(d->compute);
    if (d->wire) (void)hipStreamDestroy(d->wire);
    delete d;
}

int mfsgd_dsgd_init_q(mfsgd_dsgd* d, int64_t seed, int64_t u_total) {
    if (!d || u_total < 0) return dfail(d, MFSGD_ERR_INVALID_ARG, "dsgd_init_q: bad argument");
    try {
        int rc = finish(d);
        if (rc) return rc;
        if (d->group != d->rank) return dfail(d, MFSGD_ERR_STATE, "dsgd_init_q: blocks are not home");
        std::vector<float> host((size_t)d->max_rows * d->kp);
        for (int j = 0; j < d->m; ++j) {
            std::fill(host.begin(), host.end(), 0.0f);
            DLIB(d, mfsgd_part_init_q(d->h, d->part(j), seed, u_total, host.data()));
            DHIP(d, hipMemcpy(d->block(d->cur, j), host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        return MFSGD_OK;
    } catch (const std::bad_alloc&) {
        return dfail(d, MFSGD_ERR_OOM, "dsgd_init_q: out of host memory");
    }
}

int mfsgd_dsgd_set_q(mfsgd_dsgd* d, int32_t j, const float* block_host) {
    if (!d || !block_host || j < 0 || j >= d->m) return dfail(d, MFSGD_ERR_INVALID_ARG, "dsgd_set_q: bad argument");
    try {
        int rc = finish(d);
        if (rc) return rc;
        int32_t rows = 0;
        DLIB(d, mfsgd_part_rows(d->h, d->part(j), &rows));
        std::vector<float> host((size_t)d->max_rows * d->kp, 0.0f);
        for (int32_t x = 0; x < rows; ++x) std::memcpy(&host[(size_t)x * d->kp], block_host + (size_t)x * d->k, sizeof(float) * (size_t)d->k);
        DHIP(d, hipMemcpy(d->block(d->cur, j), host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
        return MFSGD_OK;
    } catch (const std::bad_alloc&) {
        return dfail(d, MFSGD_ERR_OOM, "dsgd_set_q: out of host memory");
    }
}

int mfsgd_dsgd_get_q(mfsgd_dsgd* d, int32_t j, int32_t* part, int32_t* rows_out, float* block_host) {
    if (!d || j < 0 || j >= d->m) return dfail(d, MFSGD_ERR_INVALID_ARG, "dsgd_get_q: bad argument");
    try {
        int rc = finish(d);
        if (rc) return rc;
        int32_t rows = 0;
        DLIB(d, mfsgd_part_rows(d->h, d->part(j), &rows));
        if (part) *part = d->part(j);
        if (rows_out) *rows_out = rows;
        if (block_host) {
            std::vector<float> host((size_t)d->max_rows * d->kp);
            DHIP(d, hipMemcpy(host.data(), d->block(d->cur, j), host.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (int32_t x = 0; x < rows; ++x) std::memcpy(block_host + (size_t)x * d->k, &host[(size_t)x * d->kp], sizeof(float) * (size_t)d->k);
        }
        return MFSGD_OK;
    } catch (const std::bad_alloc&) {
        return dfail(d, MFSGD_ERR_OOM, "dsgd_get_q: out of host memory");
    }
}

int mfsgd_dsgd_rmse(mfsgd_dsgd* d, double* out) {
    if (!d || !out) return dfail(d, MFSGD_ERR_INVALID_ARG, "dsgd_rmse: null argument");
    double v[2] = {0.0, (double)d->nnz_local};
    int rc = local_sse(d, &v[0]);
    if (rc) return rc;
    if ((rc = allreduce2(d, v, ncclSum))) return rc;
    *out = v[1] > 0 ? std::sqrt(v[0] / v[1]) : 0.0;
    return MFSGD_OK;
}

int mfsgd_dsgd_train(mfsgd_dsgd* d, int32_t epochs, double* rmse_per_epoch) {
    if (!d || epochs < 0) return dfail(d, MFSGD_ERR_INVALID_ARG, "dsgd_train: bad argument");
    for (int e = 0; e < epochs; ++e) {
        int rc = enqueue_epoch(d, e == 0 || d->m > 1 || d->always_check);
        if (rc) return rc;
        if (rmse_per_epoch && (rc = mfsgd_dsgd_rmse(d, &rmse_per_epoch[e]))) return rc;
    }
    return finish(d);
}

int mfsgd_dsgd_train_timed(mfsgd_dsgd* d, int32_t epochs, double* elapsed_ms) {
    if (!d || epochs < 0 || !elapsed_ms) return dfail(d, MFSGD_ERR_INVALID_ARG, "dsgd_train_timed: bad argument");
    int rc = finish(d);
    if (rc) return rc;
    DHIP(d, hipEventRecord(d->ev0, d->compute));
    for (int e = 0; e < epochs; ++e)
        if ((rc = enqueue_epoch(d, e == 0 || d->m > 1 || d->always_check))) return rc;
    // the last blocks arrive on the communication stream: the epoch ends when they are home
    for (int j = 0; j < d->m; ++j) DHIP(d, hipStreamWaitEvent(d->compute, d->arrived[(size_t)j], 0));
    DHIP(d, hipEventRecord(d->ev1, d->compute));
    DHIP(d, hipEventSynchronize(d->ev1));
    float ms = 0.f;
    DHIP(d, hipEventElapsedTime(&ms, d->ev0, d->ev1));
    *elapsed_ms = (double)ms;
    return finish(d);
}

int mfsgd_dsgd_stats(const mfsgd_dsgd* d, int64_t* out4) {
    if (!d || !out4) return MFSGD_ERR_INVALID_ARG;
    out4[0] = d->n_trained;
    out4[1] = d->n_checked;
    out4[2] = d->n_rerun;
    out4[3] = d->bytes_sent;
    return MFSGD_OK;
}

int mfsgd_dsgd_allreduce(mfsgd_dsgd* d, double* values2, int32_t op) {
    if (!d || !values2 || op < 0 || op > 1) return dfail(d, MFSGD_ERR_INVALID_ARG, "dsgd_allreduce: bad argument");
    return allreduce2(d, values2, op == 0 ? ncclSum : ncclMax);
}

}  // extern "C"
