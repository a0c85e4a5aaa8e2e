// ingest.hpp -- rating ingestion on the device (first phase):
// degree histograms and COO -> (cell, sub-round, wave) bucket order on the GPU.  The host
// scheduler (schedule.cpp, HIP-free) sees it only through these callbacks and falls back to
// its own loops when they are absent or fail; both produce identical arrays.
#pragma once

#include <cstdint>

namespace mfsgd {

struct DeviceIngest {
    void* ctx = nullptr;
    // degu[U], degi[I]: number of ratings per row.  Returns 0 on success.
    int (*degrees)(void* ctx, const int32_t* u, const int32_t* i, int64_t n, int32_t U, int32_t I, int64_t* degu,
                   int64_t* degi) = nullptr;
    // Stable sort of the rating indices by bucket key
    //   key = (((ub * B + it) * W + s) * W + us),  ub = ubin[u] % B, us = ubin[u] / B, it = ibin[i] % B,
    //   is = ibin[i] / B, s = (is - us + W) % W
    // bptr[nb + 1] = first position of every bucket, sorted[n] = rating indices in key order (ties in
    // input order).  Returns 0 on success.
    int (*bucket)(void* ctx, const int32_t* u, const int32_t* i, int64_t n, const int32_t* ubin, const int32_t* ibin,
                  int32_t U, int32_t I, int B, int W, int64_t* bptr, int64_t* sorted) = nullptr;
};

// Implemented in ingest.hip.  `device` must already be usable (capi checks).
DeviceIngest make_device_ingest(int device);
void destroy_device_ingest(DeviceIngest& d);

}  // namespace mfsgd
