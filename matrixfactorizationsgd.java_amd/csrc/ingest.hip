// ingest.hip -- see ingest.hpp.  Streaming passes over the COO triples: HBM-bound integer work
// (coalesced reads of u/i, random 4-byte gathers of the bin maps, one LSD radix sort).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>
#include <vector>

#include "ingest.hpp"

namespace mfsgd {

namespace {

struct Ctx {
    int device = 0;
    // the triples stay on the device between degrees() and bucket() of the same arrays
    const int32_t* host_u = nullptr;
    const int32_t* host_i = nullptr;
    int64_t n = 0;
    int32_t *du = nullptr, *di = nullptr;
};

#define ING_CHK(call)                      \
    do {                                   \
        if ((call) != hipSuccess) {        \
            (void)hipGetLastError();       \
            goto fail;                     \
        }                                  \
    } while (0)

void drop_triples(Ctx* c) {
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

__global__ void __launch_bounds__(256) key_kernel(const int32_t* __restrict__ u, const int32_t* __restrict__ i,
                                                  const int64_t n, const int32_t* __restrict__ ubin,
                                                  const int32_t* __restrict__ ibin, const int B, const int W,
                                                  unsigned* __restrict__ key, unsigned* __restrict__ val) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        const int fu = ubin[u[j]], fi = ibin[i[j]];
        const int ub = fu % B, us = fu / B, it = fi % B, is = fi / B;
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
    hipLaunchKernelGGL(degree_kernel, dim3(grid_for(n)), dim3(256), 0, 0, c->du, c->di, n, d_u, d_i);
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

int bucket_cb(void* vctx, const int32_t* u, const int32_t* i, int64_t n, const int32_t* ubin, const int32_t* ibin,
              int32_t U, int32_t I, int B, int W, int64_t* bptr, int64_t* sorted) {
    Ctx* c = static_cast<Ctx*>(vctx);
    const int64_t nb = (int64_t)B * B * W * W;
    if (n >= (int64_t)1 << 32 || nb >= (int64_t)1 << 32) return -1;
    if (!ensure_triples(c, u, i, n)) return -1;
    int32_t *d_ubin = nullptr, *d_ibin = nullptr;
    unsigned *k0 = nullptr, *k1 = nullptr, *v0 = nullptr, *v1 = nullptr;
    long long* d_bptr = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    std::vector<unsigned> hv;
    std::vector<long long> hb;
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
    hipLaunchKernelGGL(key_kernel, dim3(grid_for(n)), dim3(256), 0, 0, c->du, c->di, n, d_ubin, d_ibin, B, W, k0, v0);
    ING_CHK(hipGetLastError());
    // LSD radix sort: stable, so equal keys keep their input order -- the host counting sort's order
    ING_CHK(rocprim::radix_sort_pairs(nullptr, temp_bytes, k0, k1, v0, v1, (size_t)n, 0u, bits, (hipStream_t)0));
    ING_CHK(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    ING_CHK(rocprim::radix_sort_pairs(temp, temp_bytes, k0, k1, v0, v1, (size_t)n, 0u, bits, (hipStream_t)0));
    hipLaunchKernelGGL(bound_kernel, dim3(grid_for(nb + 1)), dim3(256), 0, 0, k1, n, nb, d_bptr);
    ING_CHK(hipGetLastError());
    hv.resize(nn);
    hb.resize((size_t)(nb + 1));
    ING_CHK(hipMemcpy(hv.data(), v1, 4 * (size_t)n, hipMemcpyDeviceToHost));
    ING_CHK(hipMemcpy(hb.data(), d_bptr, sizeof(long long) * (size_t)(nb + 1), hipMemcpyDeviceToHost));
    for (int64_t j = 0; j < n; ++j) sorted[j] = (int64_t)hv[(size_t)j];
    for (int64_t b = 0; b <= nb; ++b) bptr[b] = (int64_t)hb[(size_t)b];
    (void)hipFree(d_ubin); (void)hipFree(d_ibin); (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(v0); (void)hipFree(v1);
    (void)hipFree(d_bptr); (void)hipFree(temp);
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

}  // namespace

DeviceIngest make_device_ingest(int device) {
    DeviceIngest d;
    Ctx* c = new Ctx();
    c->device = device;
    d.ctx = c;
    d.degrees = degrees_cb;
    d.bucket = bucket_cb;
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
