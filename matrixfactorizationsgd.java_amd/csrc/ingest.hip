// ingest.hip -- see ingest.hpp.  Streaming passes over the COO triples: HBM-bound integer work
// (coalesced reads of u/i, random 4-byte gathers of the bin maps, one LSD radix sort).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdint>
#include <vector>

#include "ingest.hpp"
#include "hugepages.hpp"
#include "pack.hpp"

namespace mfsgd {

namespace {

struct Ctx {
    int device = 0;
    // the triples stay on the device between degrees() and bucket() of the same arrays.  "The same arrays" is
    // recognised by (pointer, pointer, length), which holds only WITHIN one build: a caller that builds several
    // schedules from buffers it frees and allocates again (the partitions of a DSGD handle) gets the same addresses
    // back from the allocator for different contents and must call DeviceIngest::forget between the builds
    const int32_t* host_u = nullptr;
    const int32_t* host_i = nullptr;
    int64_t n = 0;
    int32_t *du = nullptr, *di = nullptr;
    // kept after bucket_dev() for the device packer
    unsigned* d_sorted = nullptr;    // rating indices in bucket order
    long long* d_bptr = nullptr;     // nb + 1
    int64_t nb = 0;
    float* d_r = nullptr;
    long long* d_orig = nullptr;
    int32_t *d_urank = nullptr, *d_irank = nullptr;
    PackCellInfo* d_info = nullptr;
    SubDesc* d_subs = nullptr;
    PackArgs args{};                 // as launched for COUNT; EMIT reuses it
    int64_t n_cells = 0;
    int rows_full = 0;               // the row capacity no cell (or part of one) can exceed
    // one-pass mode: what the COUNT pass already wrote
    uint32_t* d_srows = nullptr;     // scratch rows (2 per rating)
    Entry* d_sent = nullptr;         // scratch entries (worst-case strides)
    long long* d_order = nullptr;    // the canonical order, final
    long long* d_ord_off = nullptr;
};

#define ING_CHK(call)                      \
    do {                                   \
        if ((call) != hipSuccess) {        \
            (void)hipGetLastError();       \
            goto fail;                     \
        }                                  \
    } while (0)

void drop_pack_state(Ctx* c) {
    void* ptrs[] = {c->d_sorted, c->d_bptr, c->d_r, c->d_orig, c->d_urank, c->d_irank, c->d_info, c->d_subs,
                    c->d_srows, c->d_sent, c->d_order, c->d_ord_off};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    c->d_srows = nullptr;
    c->d_sent = nullptr;
    c->d_order = nullptr;
    c->d_ord_off = nullptr;
    c->d_sorted = nullptr;
    c->d_bptr = nullptr;
    c->d_r = nullptr;
    c->d_orig = nullptr;
    c->d_urank = c->d_irank = nullptr;
    c->d_info = nullptr;
    c->d_subs = nullptr;
}

void drop_triples(Ctx* c) {
    drop_pack_state(c);
    if (c->du) (void)hipFree(c->du);
    if (c->di) (void)hipFree(c->di);
    c->du = c->di = nullptr;
    c->host_u = c->host_i = nullptr;
    c->n = 0;
}

bool ensure_triples(Ctx* c, const int32_t* u, const int32_t* i, int64_t n) {
    if (c->host_u == u && c->host_i == i && c->n == n && c->du) return true;
    drop_triples(c);
    if (hipSetDevice(c->device) != hipSuccess) return false;
    const size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(int32_t);
    if (hipMalloc(&c->du, bytes) != hipSuccess || hipMalloc(&c->di, bytes) != hipSuccess ||
        hipMemcpy(c->du, u, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->di, i, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        drop_triples(c);
        return false;
    }
    c->host_u = u;
    c->host_i = i;
    c->n = n;
    return true;
}

__global__ void __launch_bounds__(256) degree_kernel(const int32_t* __restrict__ u, const int32_t* __restrict__ i,
                                                     const int64_t n, unsigned* __restrict__ degu,
                                                     unsigned* __restrict__ degi) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        atomicAdd(&degu[u[j]], 1u);
        atomicAdd(&degi[i[j]], 1u);
    }
}

// The same with ONE side's counts kept in the LDS and flushed at the end: the global atomics of a popular row all hit
// one address (232 K increments of one word at the Netflix shape: 11 ms for 100 M ratings); a workgroup's LDS takes
// them at LDS speed and hands on one sum per row it saw.  `small` = the side with at most kDegLdsRows rows.
constexpr int kDegLdsRows = 36 * 1024;  // x 4 B = 144 KiB of LDS
__global__ void __launch_bounds__(1024) degree_lds_kernel(const int32_t* __restrict__ big, const int32_t* __restrict__ small_ids,
                                                          const int64_t n, unsigned* __restrict__ deg_big,
                                                          unsigned* __restrict__ deg_small, const int n_small) {
    extern __shared__ unsigned hist[];
    for (int x = threadIdx.x; x < n_small; x += blockDim.x) hist[x] = 0u;
    __syncthreads();
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        atomicAdd(&deg_big[big[j]], 1u);
        atomicAdd(&hist[small_ids[j]], 1u);
    }
    __syncthreads();
    for (int x = threadIdx.x; x < n_small; x += blockDim.x)
        if (hist[x] != 0u) atomicAdd(&deg_small[x], hist[x]);
}

__global__ void __launch_bounds__(256) key_kernel(const int32_t* __restrict__ u, const int32_t* __restrict__ i,
                                                  const int64_t n, const int32_t* __restrict__ ubin,
                                                  const int32_t* __restrict__ ibin, const int B, const int W,
                                                  const int giants, unsigned* __restrict__ key,
                                                  unsigned* __restrict__ val) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        const int fu = ubin[u[j]], fi = ibin[i[j]];
        const int ub = fu % B, us = fi < giants ? 0 : fu / B, it = fi % B, is = fi / B;
        const int s = (is - us + W) % W;
        key[j] = (unsigned)((((long long)ub * B + it) * W + s) * W + us);
        val[j] = (unsigned)j;
    }
}

// bptr[b] = first position whose key is >= b (keys sorted ascending); bptr[nb] = n
__global__ void __launch_bounds__(256) bound_kernel(const unsigned* __restrict__ key, const int64_t n, const int64_t nb,
                                                    long long* __restrict__ bptr) {
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b <= nb; b += (int64_t)gridDim.x * 256) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)key[mid] < b) lo = mid + 1;
            else hi = mid;
        }
        bptr[b] = lo;
    }
}

int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 256 * 8) g = 256 * 8;  // 2048 workgroups, grid-stride beyond
    return (int)(g < 1 ? 1 : g);
}

int degrees_cb(void* vctx, const int32_t* u, const int32_t* i, int64_t n, int32_t U, int32_t I, int64_t* degu,
               int64_t* degi) {
    Ctx* c = static_cast<Ctx*>(vctx);
    if (n >= (int64_t)1 << 32) return -1;
    if (!ensure_triples(c, u, i, n)) return -1;
    unsigned *d_u = nullptr, *d_i = nullptr;
    std::vector<unsigned> hu((size_t)U), hi((size_t)I);
    ING_CHK(hipMalloc(&d_u, sizeof(unsigned) * (size_t)U));
    ING_CHK(hipMalloc(&d_i, sizeof(unsigned) * (size_t)I));
    ING_CHK(hipMemset(d_u, 0, sizeof(unsigned) * (size_t)U));
    ING_CHK(hipMemset(d_i, 0, sizeof(unsigned) * (size_t)I));
    if (std::min(U, I) <= kDegLdsRows && n >= (1 << 20)) {
        const bool items_small = I <= U;
        const int n_small = items_small ? I : U;
        const size_t lds = 4 * (size_t)n_small;
        ING_CHK(hipFuncSetAttribute((const void*)degree_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(degree_lds_kernel, dim3(256), dim3(1024), lds, 0, items_small ? c->du : c->di, items_small ? c->di : c->du, n,
                           items_small ? d_u : d_i, items_small ? d_i : d_u, n_small);
    } else {
        hipLaunchKernelGGL(degree_kernel, dim3(grid_for(n)), dim3(256), 0, 0, c->du, c->di, n, d_u, d_i);
    }
    ING_CHK(hipGetLastError());
    ING_CHK(hipMemcpy(hu.data(), d_u, sizeof(unsigned) * (size_t)U, hipMemcpyDeviceToHost));
    ING_CHK(hipMemcpy(hi.data(), d_i, sizeof(unsigned) * (size_t)I, hipMemcpyDeviceToHost));
    (void)hipFree(d_u);
    (void)hipFree(d_i);
    for (int32_t x = 0; x < U; ++x) degu[x] = hu[(size_t)x];
    for (int32_t x = 0; x < I; ++x) degi[x] = hi[(size_t)x];
    return 0;
fail:
    if (d_u) (void)hipFree(d_u);
    if (d_i) (void)hipFree(d_i);
    return -1;
}

// Keys, one stable LSD radix sort of (key, index) pairs, bucket starts.  Leaves the sorted indices and
// the bucket starts on the device (c->d_sorted, c->d_bptr).
int bucket_on_device(Ctx* c, const int32_t* u, const int32_t* i, int64_t n, const int32_t* ubin, const int32_t* ibin,
                     int32_t U, int32_t I, int B, int W, int giants) {
    const int64_t nb = (int64_t)B * B * W * W;
    if (n >= (int64_t)1 << 32 || nb >= (int64_t)1 << 32) return -1;
    if (!ensure_triples(c, u, i, n)) return -1;
    drop_pack_state(c);
    int32_t *d_ubin = nullptr, *d_ibin = nullptr;
    unsigned *k0 = nullptr, *k1 = nullptr, *v0 = nullptr, *v1 = nullptr;
    long long* d_bptr = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    unsigned bits = 1;
    while (((int64_t)1 << bits) < nb) ++bits;
    ING_CHK(hipMalloc(&d_ubin, sizeof(int32_t) * (size_t)U));
    ING_CHK(hipMalloc(&d_ibin, sizeof(int32_t) * (size_t)I));
    ING_CHK(hipMemcpy(d_ubin, ubin, sizeof(int32_t) * (size_t)U, hipMemcpyHostToDevice));
    ING_CHK(hipMemcpy(d_ibin, ibin, sizeof(int32_t) * (size_t)I, hipMemcpyHostToDevice));
    ING_CHK(hipMalloc(&k0, 4 * nn));
    ING_CHK(hipMalloc(&k1, 4 * nn));
    ING_CHK(hipMalloc(&v0, 4 * nn));
    ING_CHK(hipMalloc(&v1, 4 * nn));
    ING_CHK(hipMalloc(&d_bptr, sizeof(long long) * (size_t)(nb + 1)));
    hipLaunchKernelGGL(key_kernel, dim3(grid_for(n)), dim3(256), 0, 0, c->du, c->di, n, d_ubin, d_ibin, B, W, giants, k0, v0);
    ING_CHK(hipGetLastError());
    // LSD radix sort: stable, so equal keys keep their input order -- the host counting sort's order
    ING_CHK(rocprim::radix_sort_pairs(nullptr, temp_bytes, k0, k1, v0, v1, (size_t)n, 0u, bits, (hipStream_t)0));
    ING_CHK(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    ING_CHK(rocprim::radix_sort_pairs(temp, temp_bytes, k0, k1, v0, v1, (size_t)n, 0u, bits, (hipStream_t)0));
    hipLaunchKernelGGL(bound_kernel, dim3(grid_for(nb + 1)), dim3(256), 0, 0, k1, n, nb, d_bptr);
    ING_CHK(hipGetLastError());
    ING_CHK(hipDeviceSynchronize());
    (void)hipFree(d_ubin); (void)hipFree(d_ibin); (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(v0); (void)hipFree(temp);
    c->d_sorted = v1;
    c->d_bptr = d_bptr;
    c->nb = nb;
    return 0;
fail:
    if (d_ubin) (void)hipFree(d_ubin);
    if (d_ibin) (void)hipFree(d_ibin);
    if (k0) (void)hipFree(k0);
    if (k1) (void)hipFree(k1);
    if (v0) (void)hipFree(v0);
    if (v1) (void)hipFree(v1);
    if (d_bptr) (void)hipFree(d_bptr);
    if (temp) (void)hipFree(temp);
    return -1;
}

int fetch_bptr(Ctx* c, int64_t* bptr) {
    static_assert(sizeof(long long) == sizeof(int64_t), "the bucket starts come down as they are");
    if (hipMemcpy(bptr, c->d_bptr, sizeof(long long) * (size_t)(c->nb + 1), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return 0;
}

int fetch_sorted_cb(void* vctx, int64_t* sorted) {
    Ctx* c = static_cast<Ctx*>(vctx);
    if (!c->d_sorted) return -1;
    std::vector<unsigned> hv((size_t)(c->n > 0 ? c->n : 1));
    if (hipMemcpy(hv.data(), c->d_sorted, 4 * (size_t)c->n, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    for (int64_t j = 0; j < c->n; ++j) sorted[j] = (int64_t)hv[(size_t)j];
    return 0;
}

int fetch_sorted32_cb(void* vctx, uint32_t* sorted) {
    Ctx* c = static_cast<Ctx*>(vctx);
    if (!c->d_sorted) return -1;
    if (c->n > 0 && hipMemcpy(sorted, c->d_sorted, 4 * (size_t)c->n, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return 0;
}

// out[dst[x] + y] = sorted[lo[x] + y], y < len[x]: one workgroup per range
__global__ void __launch_bounds__(256) gather_ranges_kernel(const unsigned* __restrict__ sorted, const long long* __restrict__ lo,
                                                            const long long* __restrict__ len, const long long* __restrict__ dst,
                                                            unsigned* __restrict__ out) {
    const long long a = lo[blockIdx.x], n = len[blockIdx.x], d = dst[blockIdx.x];
    for (long long y = threadIdx.x; y < n; y += 256) out[d + y] = sorted[a + y];
}

int fetch_sorted_ranges_cb(void* vctx, int64_t n_ranges, const int64_t* lo, const int64_t* len, uint32_t* out) {
    Ctx* c = static_cast<Ctx*>(vctx);
    if (!c->d_sorted || n_ranges < 0) return -1;
    if (n_ranges == 0) return 0;
    std::vector<long long> dst((size_t)n_ranges);
    long long total = 0;
    for (int64_t x = 0; x < n_ranges; ++x) {
        if (lo[x] < 0 || len[x] < 0 || lo[x] + len[x] > c->n) return -1;
        dst[(size_t)x] = total;
        total += len[x];
    }
    if (total == 0) return 0;
    long long *d_lo = nullptr, *d_len = nullptr, *d_dst = nullptr;
    unsigned* d_out = nullptr;
    int rc = -1;
    static_assert(sizeof(long long) == sizeof(int64_t), "ranges are uploaded as they are");
    ING_CHK(hipMalloc(&d_lo, 8 * (size_t)n_ranges));
    ING_CHK(hipMalloc(&d_len, 8 * (size_t)n_ranges));
    ING_CHK(hipMalloc(&d_dst, 8 * (size_t)n_ranges));
    ING_CHK(hipMalloc(&d_out, 4 * (size_t)total));
    ING_CHK(hipMemcpy(d_lo, lo, 8 * (size_t)n_ranges, hipMemcpyHostToDevice));
    ING_CHK(hipMemcpy(d_len, len, 8 * (size_t)n_ranges, hipMemcpyHostToDevice));
    ING_CHK(hipMemcpy(d_dst, dst.data(), 8 * (size_t)n_ranges, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gather_ranges_kernel, dim3((unsigned)n_ranges), dim3(256), 0, 0, c->d_sorted, d_lo, d_len, d_dst, d_out);
    ING_CHK(hipGetLastError());
    ING_CHK(hipMemcpy(out, d_out, 4 * (size_t)total, hipMemcpyDeviceToHost));
    rc = 0;
fail:
    if (d_lo) (void)hipFree(d_lo);
    if (d_len) (void)hipFree(d_len);
    if (d_dst) (void)hipFree(d_dst);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

int bucket_dev_cb(void* vctx, const int32_t* u, const int32_t* i, int64_t n, const int32_t* ubin, const int32_t* ibin,
                  int32_t U, int32_t I, int B, int W, int giants, int64_t* bptr) {
    Ctx* c = static_cast<Ctx*>(vctx);
    if (bucket_on_device(c, u, i, n, ubin, ibin, U, I, B, W, giants) != 0) return -1;
    return fetch_bptr(c, bptr);
}

int bucket_cb(void* vctx, const int32_t* u, const int32_t* i, int64_t n, const int32_t* ubin, const int32_t* ibin,
              int32_t U, int32_t I, int B, int W, int giants, int64_t* bptr, int64_t* sorted) {
    Ctx* c = static_cast<Ctx*>(vctx);
    if (bucket_on_device(c, u, i, n, ubin, ibin, U, I, B, W, giants) != 0) return -1;
    const int rc = fetch_bptr(c, bptr) == 0 && fetch_sorted_cb(c, sorted) == 0 ? 0 : -1;
    drop_pack_state(c);
    return rc;
}

// ---- the device packer (pack.hip) behind the DeviceIngestExt callbacks ----------------------------
void release_cb(void* p) {
    if (p) (void)hipFree(p);
}

// rank of every row among the rows of its block (block = bin % B), ascending row index
void block_ranks(const int32_t* bin, int32_t n, int B, std::vector<int32_t>& rank, int32_t& max_rank) {
    std::vector<int32_t> next((size_t)B, 0);
    rank.resize((size_t)n);
    for (int32_t x = 0; x < n; ++x) rank[(size_t)x] = next[(size_t)(bin[x] % B)]++;
    max_rank = 0;
    for (int b = 0; b < B; ++b) max_rank = std::max(max_rank, next[(size_t)b]);
}

int pack_count_cb(void* vctx, const PackRequest& q, std::vector<PackCellInfo>& info, std::vector<SubDesc>& subs) {
    Ctx* c = static_cast<Ctx*>(vctx);
    if (!c->d_sorted || !c->d_bptr || c->host_u != q.u || c->host_i != q.i || c->n != q.n) return -1;
    const int64_t n_cells = (int64_t)q.B * q.B, WW = (int64_t)q.W * q.W;
    if (n_cells >= (int64_t)1 << 31 || q.G > 64) return 1;
    std::vector<int32_t> ur, ir;
    int32_t mu = 0, mi = 0;
    block_ranks(q.ubin, q.U, q.B, ur, mu);
    block_ranks(q.ibin, q.I, q.B, ir, mi);
    PackArgs a{};
    a.max_m = (int)((std::max<int64_t>(q.max_cell_nnz, 64) + 63) / 64 * 64);
    if (mu > 65535 || mi > 65535 || a.max_m > 2048) return 1;  // 16-bit ranks, 32 candidates per lane
    a.u_words = (mu + 31) / 32 + 1;
    a.i_words = (mi + 31) / 32 + 1;
    // rows a cell can touch: at most 2 per rating and at most what the blocks hold; the training kernel's LDS
    // image (160 KiB) cannot hold more than 10240 16-byte units of rows anyway.  Most cells touch far fewer, and
    // the kernel's occupancy hangs on this number (its per-wave state arrays), so a first launch provides for
    // 512 and only a set with fuller cells pays for a second launch with the full bound.
    int rows_full = std::max(8, (int)std::min<int64_t>(std::min<int64_t>(2 * (int64_t)a.max_m, (int64_t)mu + mi), 32767 / std::max(1, q.L)));
    if (q.fit_rows > 0) rows_full = std::max(8, std::min(rows_full, q.fit_rows));  // (more rows than fit: cut anyway)
    a.max_rows = std::min(rows_full, 512);
    a.B = q.B;
    a.W = q.W;
    a.G = q.G;
    a.L = q.L;
    a.lr = q.lr;
    a.c = q.c;
    a.solo_ok = q.solo_ok ? 1 : 0;
    {
        PackArgs worst = a;
        worst.max_rows = rows_full;
        if (pack_lds_bytes(worst) > 160 * 1024 - 256) return 1;
    }
    int rc = -1;
    ING_CHK(hipMalloc(&c->d_r, sizeof(float) * (size_t)std::max<int64_t>(q.n, 1)));
    ING_CHK(hipMemcpy(c->d_r, q.r, sizeof(float) * (size_t)q.n, hipMemcpyHostToDevice));
    if (q.orig) {
        ING_CHK(hipMalloc(&c->d_orig, sizeof(long long) * (size_t)std::max<int64_t>(q.n, 1)));
        ING_CHK(hipMemcpy(c->d_orig, q.orig, sizeof(long long) * (size_t)q.n, hipMemcpyHostToDevice));
    }
    ING_CHK(hipMalloc(&c->d_urank, sizeof(int32_t) * (size_t)q.U));
    ING_CHK(hipMalloc(&c->d_irank, sizeof(int32_t) * (size_t)q.I));
    ING_CHK(hipMemcpy(c->d_urank, ur.data(), sizeof(int32_t) * (size_t)q.U, hipMemcpyHostToDevice));
    ING_CHK(hipMemcpy(c->d_irank, ir.data(), sizeof(int32_t) * (size_t)q.I, hipMemcpyHostToDevice));
    ING_CHK(hipMalloc(&c->d_info, sizeof(PackCellInfo) * (size_t)n_cells));
    ING_CHK(hipMalloc(&c->d_subs, sizeof(SubDesc) * (size_t)(n_cells * WW)));
    a.u = c->du;
    a.i = c->di;
    a.r = c->d_r;
    a.orig = c->d_orig;
    a.sorted = c->d_sorted;
    a.bptr = c->d_bptr;
    a.urank = c->d_urank;
    a.irank = c->d_irank;
    a.info = c->d_info;
    a.subs = c->d_subs;
    a.emit = 0;
    if (q.ord_off && !std::getenv("MFSGD_PACK_TWICE")) {  // (the variable: A/B measurements)
        // one-pass mode: scratch for rows and entries, the order array itself; if any of it does not fit, count only
        // (nor when the scratch would take more than a third of what is free: the final arrays come after it)
        const size_t steps = pack_scratch_steps(q.n, n_cells, q.W);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        const size_t need = sizeof(Entry) * steps * (size_t)q.G + 16 * (size_t)std::max<int64_t>(q.n, 1);
        if (need <= free_b / 3 && hipMalloc(&c->d_srows, 4 * (2 * (size_t)std::max<int64_t>(q.n, 1))) == hipSuccess &&
            hipMalloc(&c->d_sent, sizeof(Entry) * steps * (size_t)q.G) == hipSuccess &&
            hipMalloc(&c->d_order, 8 * (size_t)std::max<int64_t>(q.n, 1)) == hipSuccess &&
            hipMalloc(&c->d_ord_off, 8 * (size_t)n_cells) == hipSuccess &&
            hipMemcpy(c->d_ord_off, q.ord_off, 8 * (size_t)n_cells, hipMemcpyHostToDevice) == hipSuccess) {
            a.emit = 2;
            a.rows = c->d_srows;
            a.entries = c->d_sent;
            a.order = c->d_order;
            a.ord_off = c->d_ord_off;
        } else {
            (void)hipGetLastError();
            void* ptrs[] = {c->d_srows, c->d_sent, c->d_order, c->d_ord_off};
            for (void* p : ptrs)
                if (p) (void)hipFree(p);
            c->d_srows = nullptr;
            c->d_sent = nullptr;
            c->d_order = nullptr;
            c->d_ord_off = nullptr;
        }
    }
    reserve_huge(info, (size_t)n_cells);
    info.resize((size_t)n_cells);
    if (q.want_subs) {
        reserve_huge(subs, (size_t)(n_cells * WW));
        subs.resize((size_t)(n_cells * WW));
    } else {
        subs.clear();
    }
    for (;;) {
        ING_CHK(launch_pack(a, n_cells, (hipStream_t)0));
        ING_CHK(hipMemcpy(info.data(), c->d_info, sizeof(PackCellInfo) * (size_t)n_cells, hipMemcpyDeviceToHost));
        bool more_rows = false;
        for (const PackCellInfo& ci : info) more_rows = more_rows || ci.status == 2;
        if (!more_rows || a.max_rows >= rows_full) break;
        a.max_rows = rows_full;
    }
    if (q.want_subs) ING_CHK(hipMemcpy(subs.data(), c->d_subs, sizeof(SubDesc) * (size_t)(n_cells * WW), hipMemcpyDeviceToHost));
    c->args = a;
    c->n_cells = n_cells;
    c->rows_full = rows_full;
    return 0;
fail:
    return rc;
}

// A list of parts (chunks) as the packing kernel's cells: uploads it and returns the arguments of a launch over it
// (COUNT outputs allocated); the caller frees what `owned` holds.
struct PartList {
    unsigned* d_sorted = nullptr;
    long long* d_cptr = nullptr;
    PackCellInfo* d_info = nullptr;
    SubDesc* d_subs = nullptr;
    void release() {
        void* ptrs[] = {d_sorted, d_cptr, d_info, d_subs};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        d_sorted = nullptr;
        d_cptr = nullptr;
        d_info = nullptr;
        d_subs = nullptr;
    }
};

bool upload_parts(Ctx* c, int64_t n_parts, const uint32_t* sorted, int64_t n_sorted, const int64_t* cptr, PartList& pl, PackArgs& a) {
    const int64_t WW = (int64_t)c->args.W * c->args.W;
    static_assert(sizeof(long long) == sizeof(int64_t), "cptr is uploaded as it is");
    if (hipMalloc(&pl.d_sorted, 4 * (size_t)std::max<int64_t>(n_sorted, 1)) != hipSuccess ||
        hipMalloc(&pl.d_cptr, 8 * (size_t)(n_parts * WW + 1)) != hipSuccess ||
        hipMalloc(&pl.d_info, sizeof(PackCellInfo) * (size_t)n_parts) != hipSuccess ||
        hipMalloc(&pl.d_subs, sizeof(SubDesc) * (size_t)(n_parts * WW)) != hipSuccess ||
        hipMemcpy(pl.d_sorted, sorted, 4 * (size_t)n_sorted, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(pl.d_cptr, cptr, 8 * (size_t)(n_parts * WW + 1), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        pl.release();
        return false;
    }
    a = c->args;
    a.rows = nullptr;  // (a COUNT pass proper: the cells' one-pass scratch is not the parts')
    a.entries = nullptr;
    a.order = nullptr;
    a.ord_off = nullptr;
    a.max_rows = c->rows_full;  // a part can hold any number of rows a cell can
    a.sorted = pl.d_sorted;
    a.bptr = pl.d_cptr;
    a.info = pl.d_info;
    a.subs = pl.d_subs;
    a.emit = 0;
    return true;
}

int pack_count_parts_cb(void* vctx, int64_t n_parts, const uint32_t* sorted, int64_t n_sorted, const int64_t* cptr,
                        PackCellInfo* info, SubDesc* subs) {
    Ctx* c = static_cast<Ctx*>(vctx);
    if (!c->d_info || n_parts < 0) return -1;
    if (n_parts == 0) return 0;
    const int64_t WW = (int64_t)c->args.W * c->args.W;
    PartList pl;
    PackArgs a{};
    if (!upload_parts(c, n_parts, sorted, n_sorted, cptr, pl, a)) return -1;
    int rc = -1;
    ING_CHK(launch_pack(a, n_parts, (hipStream_t)0));
    ING_CHK(hipMemcpy(info, pl.d_info, sizeof(PackCellInfo) * (size_t)n_parts, hipMemcpyDeviceToHost));
    if (subs) ING_CHK(hipMemcpy(subs, pl.d_subs, sizeof(SubDesc) * (size_t)(n_parts * WW), hipMemcpyDeviceToHost));
    rc = 0;
fail:
    pl.release();
    return rc;
}

struct PartsToEmit {
    int64_t n_parts = 0, n_sorted = 0;
    const uint32_t* sorted = nullptr;
    const int64_t* cptr = nullptr;
    const uint32_t *row_off = nullptr, *ent_off = nullptr;
    const int64_t* ord_off = nullptr;
    const int64_t* desc = nullptr;  // the chunk descriptor of every part (final sub-cell table)
};

// fin[desc[y] * WW + x] = src[y * WW + x]: the parts' sub-cell tables to their chunk descriptors
__global__ void __launch_bounds__(256) table_scatter_kernel(SubDesc* __restrict__ fin, const SubDesc* __restrict__ src,
                                                            const long long* __restrict__ desc, const long long n_parts, const int WW) {
    const long long x = (long long)blockIdx.x * 256 + threadIdx.x;
    if (x >= n_parts * WW) return;
    fin[desc[x / WW] * WW + x % WW] = src[x];
}

int emit_common(Ctx* c, const uint32_t* row_off, const uint32_t* ent_off, const int64_t* ord_off, int64_t n_rows,
                int64_t n_steps, const MixedPieces* host, DevicePacked* out, const PartsToEmit* parts = nullptr,
                int64_t n_descs = 0) {
    if (!c->d_info || !out) return -1;
    PackArgs a = c->args;
    SubDesc* d_fin = nullptr;  // the final sub-cell table (n_descs > 0)
    long long* d_pdesc = nullptr;
    uint32_t *d_ro = nullptr, *d_eo = nullptr, *d_rows = nullptr;
    long long *d_oo = nullptr, *d_order = nullptr;
    Entry* d_ent = nullptr;
    void* staged[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t nc = (size_t)c->n_cells;
    const bool one_pass = c->d_sent != nullptr;  // the COUNT pass wrote what it packed: move it, do not pack again
    std::vector<long long> oo(ord_off, ord_off + nc);
    auto stage = [&](int slot, const void* p, size_t bytes) -> bool {
        if (bytes == 0) return true;
        return hipMalloc(&staged[slot], bytes) == hipSuccess && hipMemcpy(staged[slot], p, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    ING_CHK(hipMalloc(&d_ro, 4 * nc));
    ING_CHK(hipMalloc(&d_eo, 4 * nc));
    ING_CHK(hipMalloc(&d_oo, 8 * nc));
    ING_CHK(hipMemcpy(d_ro, row_off, 4 * nc, hipMemcpyHostToDevice));
    ING_CHK(hipMemcpy(d_eo, ent_off, 4 * nc, hipMemcpyHostToDevice));
    ING_CHK(hipMemcpy(d_oo, oo.data(), 8 * nc, hipMemcpyHostToDevice));
    ING_CHK(hipMalloc(&d_rows, 4 * (size_t)(n_rows + 4)));
    ING_CHK(hipMemset(d_rows + n_rows, 0, 16));  // the staging DMA reads whole 16-byte units
    ING_CHK(hipMalloc(&d_ent, sizeof(Entry) * (size_t)std::max<int64_t>(n_steps * a.G, 1)));
    if (one_pass) {
        d_order = c->d_order;  // written by the COUNT pass, at its final place
        c->d_order = nullptr;
    } else {
        ING_CHK(hipMalloc(&d_order, 8 * (size_t)std::max<int64_t>(c->n, 1)));
    }
    a.row_off = d_ro;
    a.ent_off = d_eo;
    a.ord_off = d_oo;
    a.rows = d_rows;
    a.entries = d_ent;
    a.order = d_order;
    if (n_descs > 0) {
        // the final sub-cell table: the cells' tables as the COUNT pass left them (a cell that is cut gets its first
        // chunk's below), zeros for the descriptors nothing is written to and for the two padding records
        const size_t WWs = (size_t)a.W * a.W;
        if (n_descs < c->n_cells) goto fail;
        ING_CHK(hipMalloc(&d_fin, sizeof(SubDesc) * ((size_t)n_descs * WWs + 2)));
        ING_CHK(hipMemsetAsync(d_fin + (size_t)c->n_cells * WWs, 0, sizeof(SubDesc) * ((size_t)(n_descs - c->n_cells) * WWs + 2), (hipStream_t)0));
        ING_CHK(hipMemcpyAsync(d_fin, c->d_subs, sizeof(SubDesc) * (size_t)c->n_cells * WWs, hipMemcpyDeviceToDevice, (hipStream_t)0));
    }
    if (one_pass) {
        a.emit = 2;
        ING_CHK(launch_compact(a, c->n_cells, c->d_srows, c->d_sent, (hipStream_t)0));
    } else {
        a.emit = 1;
        ING_CHK(launch_pack(a, c->n_cells, (hipStream_t)0));
    }
    if (parts && parts->n_parts > 0) {
        // the chunks of the cells that were cut: COUNT over the final list (the EMIT pass reads the sub-cell table the
        // COUNT pass of the SAME list left on the device), then EMIT at the caller's offsets into the same arrays
        PartList pl;
        PackArgs pa{};
        uint32_t *p_ro = nullptr, *p_eo = nullptr;
        long long* p_oo = nullptr;
        const size_t np = (size_t)parts->n_parts;
        bool ok = upload_parts(c, parts->n_parts, parts->sorted, parts->n_sorted, parts->cptr, pl, pa);
        ok = ok && launch_pack(pa, parts->n_parts, (hipStream_t)0) == hipSuccess;
        ok = ok && hipMalloc(&p_ro, 4 * np) == hipSuccess && hipMalloc(&p_eo, 4 * np) == hipSuccess && hipMalloc(&p_oo, 8 * np) == hipSuccess;
        ok = ok && hipMemcpy(p_ro, parts->row_off, 4 * np, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(p_eo, parts->ent_off, 4 * np, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(p_oo, parts->ord_off, 8 * np, hipMemcpyHostToDevice) == hipSuccess;
        if (ok) {
            pa.emit = 1;
            pa.row_off = p_ro;
            pa.ent_off = p_eo;
            pa.ord_off = p_oo;
            pa.rows = d_rows;
            pa.entries = d_ent;
            pa.order = d_order;
            ok = launch_pack(pa, parts->n_parts, (hipStream_t)0) == hipSuccess;
            if (ok && d_fin) {
                // (the EMIT pass does not touch the table the COUNT pass of the same list left in pl.d_subs)
                ok = parts->desc != nullptr && hipMalloc(&d_pdesc, 8 * np) == hipSuccess &&
                     hipMemcpy(d_pdesc, parts->desc, 8 * np, hipMemcpyHostToDevice) == hipSuccess;
                if (ok) {
                    const long long tot = (long long)np * a.W * a.W;
                    hipLaunchKernelGGL(table_scatter_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)0, d_fin, pl.d_subs,
                                       d_pdesc, (long long)np, a.W * a.W);
                    ok = hipGetLastError() == hipSuccess;
                }
            }
            ok = ok && hipDeviceSynchronize() == hipSuccess;
        }
        if (p_ro) (void)hipFree(p_ro);
        if (p_eo) (void)hipFree(p_eo);
        if (p_oo) (void)hipFree(p_oo);
        pl.release();
        if (!ok) {
            (void)hipGetLastError();
            goto fail;
        }
    }
    if (host) {
        // what the host packed: three staging arrays and their segment lists, then one copy kernel each
        if (!stage(0, host->rows.data(), host->rows.size() * 4) || !stage(1, host->seg_rows.data(), host->seg_rows.size() * sizeof(MixedSegment)) ||
            !stage(2, host->entries.data(), host->entries.size() * sizeof(Entry)) ||
            !stage(3, host->seg_entries.data(), host->seg_entries.size() * sizeof(MixedSegment)) ||
            !stage(4, host->order.data(), host->order.size() * 8) ||
            !stage(5, host->seg_order.data(), host->seg_order.size() * sizeof(MixedSegment))) {
            (void)hipGetLastError();
            goto fail;
        }
        ING_CHK(launch_scatter(d_rows, staged[0], static_cast<const MixedSegment*>(staged[1]), (long long)host->seg_rows.size(), 4, (hipStream_t)0));
        ING_CHK(launch_scatter(d_ent, staged[2], static_cast<const MixedSegment*>(staged[3]), (long long)host->seg_entries.size(), 16, (hipStream_t)0));
        ING_CHK(launch_scatter(d_order, staged[4], static_cast<const MixedSegment*>(staged[5]), (long long)host->seg_order.size(), 8, (hipStream_t)0));
    }
    ING_CHK(hipDeviceSynchronize());
    (void)hipFree(d_ro); (void)hipFree(d_eo); (void)hipFree(d_oo);
    for (void* p : staged)
        if (p) (void)hipFree(p);
    if (d_pdesc) (void)hipFree(d_pdesc);
    out->rows = d_rows;
    out->entries = d_ent;
    out->order = d_order;
    out->subs = d_fin;
    out->n_subs = d_fin ? n_descs * (int64_t)a.W * a.W + 2 : 0;
    out->release = release_cb;
    drop_pack_state(c);
    return 0;
fail:
    if (d_ro) (void)hipFree(d_ro);
    if (d_eo) (void)hipFree(d_eo);
    if (d_oo) (void)hipFree(d_oo);
    if (d_rows) (void)hipFree(d_rows);
    if (d_ent) (void)hipFree(d_ent);
    if (d_order) (void)hipFree(d_order);
    if (d_fin) (void)hipFree(d_fin);
    if (d_pdesc) (void)hipFree(d_pdesc);
    for (void* p : staged)
        if (p) (void)hipFree(p);
    return -1;
}

int pack_emit_cb(void* vctx, const uint32_t* row_off, const uint32_t* ent_off, const int64_t* ord_off, int64_t n_rows,
                 int64_t n_steps, int64_t n_descs, DevicePacked* out) {
    return emit_common(static_cast<Ctx*>(vctx), row_off, ent_off, ord_off, n_rows, n_steps, nullptr, out, nullptr, n_descs);
}

int pack_emit_mixed_cb(void* vctx, const uint32_t* row_off, const uint32_t* ent_off, const int64_t* ord_off, int64_t n_rows,
                       int64_t n_steps, const MixedPieces& host, DevicePacked* out) {
    return emit_common(static_cast<Ctx*>(vctx), row_off, ent_off, ord_off, n_rows, n_steps, &host, out);
}

int pack_emit_parts_cb(void* vctx, const uint32_t* row_off, const uint32_t* ent_off, const int64_t* ord_off, int64_t n_rows,
                       int64_t n_steps, int64_t n_parts, const uint32_t* sorted, int64_t n_sorted, const int64_t* cptr,
                       const uint32_t* p_row_off, const uint32_t* p_ent_off, const int64_t* p_ord_off, int64_t n_descs,
                       const int64_t* p_desc, DevicePacked* out) {
    PartsToEmit pe;
    pe.desc = p_desc;
    pe.n_parts = n_parts;
    pe.n_sorted = n_sorted;
    pe.sorted = sorted;
    pe.cptr = cptr;
    pe.row_off = p_row_off;
    pe.ent_off = p_ent_off;
    pe.ord_off = p_ord_off;
    return emit_common(static_cast<Ctx*>(vctx), row_off, ent_off, ord_off, n_rows, n_steps, nullptr, out, &pe, n_descs);
}

int download_cb(const DevicePacked& d, uint32_t* rows, int64_t n_rows, Entry* entries, int64_t n_entries, int64_t* order,
                int64_t n) {
    if (rows && n_rows > 0 && hipMemcpy(rows, d.rows, 4 * (size_t)n_rows, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (entries && n_entries > 0 &&
        hipMemcpy(entries, d.entries, sizeof(Entry) * (size_t)n_entries, hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    if (order && n > 0 && hipMemcpy(order, d.order, 8 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}

int download_raw_cb(const void* dev, void* host, size_t bytes) {
    if (bytes == 0) return 0;
    if (!dev || !host || hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return 0;
}

const DeviceIngestExt kExt = {bucket_dev_cb, fetch_sorted_cb, fetch_sorted32_cb, fetch_sorted_ranges_cb, pack_count_cb, pack_emit_cb, pack_emit_mixed_cb,
                              pack_count_parts_cb, pack_emit_parts_cb, download_cb, download_raw_cb};

}  // namespace

DeviceIngest make_device_ingest(int device) {
    DeviceIngest d;
    Ctx* c = new Ctx();
    c->device = device;
    d.ctx = c;
    d.degrees = degrees_cb;
    d.bucket = bucket_cb;
    d.forget = [](void* vctx) { drop_triples(static_cast<Ctx*>(vctx)); };
    d.ext = &kExt;
    return d;
}

void destroy_device_ingest(DeviceIngest& d) {
    if (d.ctx) {
        Ctx* c = static_cast<Ctx*>(d.ctx);
        drop_triples(c);
        delete c;
    }
    d = DeviceIngest{};
}

}  // namespace mfsgd
