// recommend.hip -- top-N scoring: for each requested user the N items with
// the largest dot(P[u], Q[i]), ties broken by the smaller item index.  Scores use the canonical
// dot of DESIGN.md section 3, so they are bit-identical to mfsgd_predict() and to the oracle.
// Score pass: one lane group per (user, item) pair, the user's row held in registers across
// items.  Selection: one stable, descending segmented radix sort per batch of users (rocPRIM).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <cstdint>

#include "kernels.hpp"

#pragma clang fp contract(off)

namespace mfsgd {

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float swap_add16(float v) {
    float t;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(v), "=&v"(t));
    return v + t;
}
__device__ __forceinline__ float swap_add32(float v) {
    float t;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(v), "=&v"(t));
    return v + t;
}
template <int L>
__device__ __forceinline__ float group_allreduce(float v) {
    if constexpr (L >= 2) v = v + dpp_move<0xB1>(v);
    if constexpr (L >= 4) v = v + dpp_move<0x4E>(v);
    if constexpr (L >= 8) v = v + dpp_move<0x141>(v);
    if constexpr (L >= 16) v = v + dpp_move<0x140>(v);
    if constexpr (L >= 32) v = swap_add16(v);
    if constexpr (L >= 64) v = swap_add32(v);
    return v;
}
__device__ __forceinline__ float chunk_dot(const float4 p, const float4 q) {
    float t0 = p.x * q.x;
    float t1 = p.y * q.y;
    t0 = __builtin_fmaf(p.z, q.z, t0);
    t1 = __builtin_fmaf(p.w, q.w, t1);
    return t0 + t1;
}

// scores[b * n_items + i] = dot(P[users[b]], Q[i]); ids[...] = i.  grid = (item blocks, users)
template <int L>
__global__ void __launch_bounds__(256) score_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                    const int32_t* __restrict__ users, const int32_t n_items,
                                                    float* __restrict__ scores, int32_t* __restrict__ ids) {
    constexpr int KP = 4 * L;
    constexpr int GPB = 256 / L;
    const int lig = threadIdx.x % L;
    const int grp = threadIdx.x / L;
    const int b = blockIdx.y;
    const float4 p = *reinterpret_cast<const float4*>(P + (size_t)users[b] * KP + lig * 4);
    const int stride = (int)gridDim.x * GPB;
    const int iters = (n_items + stride - 1) / stride;  // uniform trip count: DPP needs every lane live
    for (int it = 0; it < iters; ++it) {
        const int i = (int)blockIdx.x * GPB + grp + it * stride;
        const bool ok = i < n_items;
        const float4 q = *reinterpret_cast<const float4*>(Q + (size_t)(ok ? i : 0) * KP + lig * 4);
        const float d = group_allreduce<L>(chunk_dot(p, q));
        if (ok && lig == 0) {
            scores[(size_t)b * n_items + i] = d;
            ids[(size_t)b * n_items + i] = i;
        }
    }
}

__global__ void __launch_bounds__(256) take_top_kernel(const float* __restrict__ s, const int32_t* __restrict__ id,
                                                       const int32_t n_items, const int32_t topn,
                                                       float* __restrict__ out_s, int32_t* __restrict__ out_i) {
    const int b = blockIdx.x;
    for (int x = threadIdx.x; x < topn; x += 256) {
        out_s[(size_t)b * topn + x] = s[(size_t)b * n_items + x];
        out_i[(size_t)b * topn + x] = id[(size_t)b * n_items + x];
    }
}

__global__ void __launch_bounds__(256) offsets_kernel(long long* __restrict__ off, const int n, const int32_t n_items) {
    for (int x = threadIdx.x; x <= n; x += 256) off[x] = (long long)x * n_items;
}

template <int L>
hipError_t score_L(const float* P, const float* Q, const int32_t* users, int nb, int32_t n_items, float* scores,
                   int32_t* ids, hipStream_t st) {
    const int gpb = 256 / L;
    int bx = (n_items + gpb - 1) / gpb;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL((score_kernel<L>), dim3((unsigned)bx, (unsigned)nb), dim3(256), 0, st, P, Q, users, n_items, scores, ids);
    return hipGetLastError();
}

}  // namespace

// Device buffers are the caller's (capi.cpp): scores/ids in and out (nb * n_items each), offsets nb+1.
hipError_t recommend_batch(int L, const float* P, const float* Q, const int32_t* d_users, int nb, int32_t n_items,
                           int32_t topn, float* s_in, float* s_out, int32_t* id_in, int32_t* id_out, long long* d_off,
                           void*& temp, size_t& temp_bytes, float* out_s, int32_t* out_i, hipStream_t st) {
    hipError_t e;
    switch (L) {
        case 1: e = score_L<1>(P, Q, d_users, nb, n_items, s_in, id_in, st); break;
        case 2: e = score_L<2>(P, Q, d_users, nb, n_items, s_in, id_in, st); break;
        case 4: e = score_L<4>(P, Q, d_users, nb, n_items, s_in, id_in, st); break;
        case 8: e = score_L<8>(P, Q, d_users, nb, n_items, s_in, id_in, st); break;
        case 16: e = score_L<16>(P, Q, d_users, nb, n_items, s_in, id_in, st); break;
        case 32: e = score_L<32>(P, Q, d_users, nb, n_items, s_in, id_in, st); break;
        case 64: e = score_L<64>(P, Q, d_users, nb, n_items, s_in, id_in, st); break;
        default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(256), 0, st, d_off, nb, n_items);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    size_t need = 0;
    e = rocprim::segmented_radix_sort_pairs_desc(nullptr, need, s_in, s_out, id_in, id_out, (unsigned)((size_t)nb * n_items),
                                                 (unsigned)nb, d_off, d_off + 1, 0u, 32u, st);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) {
        if (temp) (void)hipFree(temp);
        temp = nullptr;
        temp_bytes = 0;
        if ((e = hipMalloc(&temp, need)) != hipSuccess) return e;
        temp_bytes = need;
    }
    e = rocprim::segmented_radix_sort_pairs_desc(temp, need, s_in, s_out, id_in, id_out, (unsigned)((size_t)nb * n_items),
                                                 (unsigned)nb, d_off, d_off + 1, 0u, 32u, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(take_top_kernel, dim3((unsigned)nb), dim3(256), 0, st, s_out, id_out, n_items, topn, out_s, out_i);
    return hipGetLastError();
}

}  // namespace mfsgd
