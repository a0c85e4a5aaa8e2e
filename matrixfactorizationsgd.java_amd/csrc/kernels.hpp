// kernels.hpp -- host-callable launchers of kernels.hip.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "schedule.hpp"

namespace mfsgd {

struct CellLaunch {
    float* P;
    float* Q;
    const CellDesc* cells;
    const uint32_t* rows;
    const SubDesc* subs;
    const Entry* entries;
    int B;
    int rd;          // round (training only)
    int grid;        // workgroups: B for a training round, B*B for an SSE pass
    int lds_bytes;   // dynamic LDS per workgroup
    int sched_cap;   // bytes of one of its two schedule buffers
    float lr;
    float c;         // 1 - lr*lambda
    double* sse_partial;
    bool diag = false;  // diagnostic launch: in-kernel cycle stamps are written to sse_partial (as u64 words)
};

// One training round (train = true) or the SSE pass over every cell.
hipError_t launch_cell(bool train, int L, int W, const CellLaunch& a, hipStream_t st);
// Persistent epoch kernel: a.grid workgroups (all must be co-resident: at most
// blocks_per_cu x CUs) run n_rounds rounds; `done` holds B words spaced kDoneStride
// apart (zeroed by the caller before every launch), `abort_word` one word.
constexpr int kDoneStride = 32;
hipError_t epoch_blocks_per_cu(int L, int W, const CellLaunch& a, int* blocks_per_cu);
hipError_t launch_epoch_persistent(int L, int W, const CellLaunch& a, int n_rounds, unsigned* done,
                                   unsigned* abort_word, hipStream_t st);
// SSE pass, persistent form: a.grid workgroups walk n_cells cells; a.sse_partial gets a.grid doubles.
hipError_t launch_sse_persistent(int L, int W, const CellLaunch& a, int n_cells, hipStream_t st);
hipError_t launch_reduce_sse(const double* partial, int64_t n, double* out, hipStream_t st);
hipError_t launch_predict(int L, const float* P, const float* Q, const int32_t* u, const int32_t* i,
                          float* out, int64_t n, hipStream_t st);

// rows x kp floats: java.util.Random(seed) draws first_pos + row * k ... scaled, zero padded (device-side seeding).
hipError_t launch_init_rows(float* dst, long long rows, int k, int kp, long long seed, unsigned long long first_pos, float scale,
                            hipStream_t st);

// Diagnostic (tests): `workgroups` one-wave workgroups, each holding lds_bytes of LDS, spin for `ticks` x 10 ns.
// `started` (device-accessible host memory, or null): every workgroup adds one to it as it starts
hipError_t launch_occupy(int workgroups, int lds_bytes, unsigned long long ticks, unsigned* started, hipStream_t st);

// recommend.hip: fused score + select (one workgroup per user, nothing but the winners goes to memory) for
// the (n_items, topn) recommend_is_fused() accepts ...
bool recommend_is_fused(int32_t n_items, int32_t topn);
hipError_t recommend_fused(int L, const float* P, const float* Q, const int32_t* d_users, int nb, int32_t n_items,
                           int32_t topn, float* out_s, int32_t* out_i, hipStream_t st);
// ... and for the rest: scores of nb users against every item, top `topn` of each into out_s / out_i.
hipError_t recommend_batch(int L, const float* P, const float* Q, const int32_t* d_users, int nb, int32_t n_items,
                           int32_t topn, float* s_in, float* s_out, int32_t* id_in, int32_t* id_out, long long* d_off,
                           void*& temp, size_t& temp_bytes, float* out_s, int32_t* out_i, hipStream_t st);

}  // namespace mfsgd
