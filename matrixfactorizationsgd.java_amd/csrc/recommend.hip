// recommend.hip -- top-N scoring: for each requested user the N items with
// the largest dot(P[u], Q[i]), ties broken by the smaller item index.  Scores use the canonical
// dot of DESIGN.md section 3, so they are bit-identical to mfsgd_predict() and to the oracle.
// Score pass: one lane group per (user, item) pair, the user's row held in registers across
// items.  Selection, for topn <= kTopnFused: fused behind the scores in ONE kernel -- a workgroup per
// user keeps the scores of a tile of items in LDS as order-preserving integers, finds the topn-th
// largest by a 4 x 8-bit radix select on LDS histograms, collects what lies above it (ties in
// ascending item order) and sorts only those; no score ever goes to memory.  Larger topn: scores to
// memory and one stable, descending segmented radix sort per batch of users (rocPRIM).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <cstdint>

#include "canon.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace mfsgd {

namespace {

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

// ---- fused score + select ----------------------------------------------------------------------------
constexpr int kTopnFused = 128;    // largest topn the fused kernel takes
constexpr int kTopnTile = 14336;   // items scored per tile (56 KiB of keys in LDS: two workgroups per CU)
constexpr int kTopnCand = 2048;    // candidates kept across tiles (tiles x topn must fit)

// float -> unsigned whose order is the float order (-0 counts as +0, as a comparison would)
__device__ __forceinline__ unsigned order_key(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int kTopnThreads = 1024;  // 16 waves: the scores are a latency-bound row gather, so many loads in flight

template <int L>
__global__ void __launch_bounds__(kTopnThreads) topn_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                            const int32_t* __restrict__ users, const int32_t n_items,
                                                            const int32_t topn, float* __restrict__ out_s,
                                                            int32_t* __restrict__ out_i) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = kTopnThreads, NW = NT / 64;
    unsigned* keys = reinterpret_cast<unsigned*>(smem);                                   // kTopnTile
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(keys + kTopnTile);   // kTopnCand
    unsigned* hist = reinterpret_cast<unsigned*>(cand + kTopnCand);                        // 256
    unsigned* wtot = hist + 256;                                                           // NW wave totals
    int* ctl = reinterpret_cast<int*>(wtot + NW);  // [0] candidates so far, [1] bin, [2] need
    constexpr int KP = 4 * L;
    constexpr int GPB = NT / L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lig = tid % L, grp = tid / L;
    const int b = blockIdx.x;
    const float4 p = *reinterpret_cast<const float4*>(P + (size_t)users[b] * KP + lig * 4);
    if (tid == 0) ctl[0] = 0;
    __syncthreads();
    for (int tile0 = 0; tile0 < n_items; tile0 += kTopnTile) {
        const int nt = min(kTopnTile, n_items - tile0);
        // scores of the tile (uniform trip count: the DPP reduction needs every lane live), two rows in flight
        const int iters = (nt + GPB - 1) / GPB;
        int it = 0;
        for (; it + 1 < iters; it += 2) {
            const int x0 = grp + it * GPB, x1 = x0 + GPB;
            const bool ok1 = x1 < nt;
            const float4 q0 = *reinterpret_cast<const float4*>(Q + (size_t)(tile0 + x0) * KP + lig * 4);
            const float4 q1 = *reinterpret_cast<const float4*>(Q + (size_t)(ok1 ? tile0 + x1 : 0) * KP + lig * 4);
            const float d0 = group_allreduce<L>(chunk_dot(p, q0));
            const float d1 = group_allreduce<L>(chunk_dot(p, q1));
            if (lig == 0) {
                keys[x0] = order_key(d0);
                if (ok1) keys[x1] = order_key(d1);
            }
        }
        for (; it < iters; ++it) {
            const int x = grp + it * GPB;
            const bool ok = x < nt;
            const float4 q = *reinterpret_cast<const float4*>(Q + (size_t)(ok ? tile0 + x : 0) * KP + lig * 4);
            const float d = group_allreduce<L>(chunk_dot(p, q));
            if (ok && lig == 0) keys[x] = order_key(d);
        }
        __syncthreads();
        // radix select: the key of the need-th largest score of the tile
        int need = min(topn, nt);
        unsigned prefix = 0u, mask = 0u;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0u;
            __syncthreads();
            // (LDS atomics on a handful of hot bins -- the first digit is sign + exponent -- were measured
            // faster than aggregating equal bins inside a wave first: 2.2 against 4.1 ms per 4,096 users)
            for (int x = tid; x < nt; x += NT) {
                const unsigned kx = keys[x];
                if ((kx & mask) == prefix) atomicAdd(&hist[(kx >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (wave == 0) {
                // bins from the top: lane l owns bins 255 - 4l .. 252 - 4l; counts above by a wave scan
                unsigned h[4], own = 0;
                for (int j = 0; j < 4; ++j) {
                    h[j] = hist[255 - 4 * lane - j];
                    own += h[j];
                }
                unsigned incl = own;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned o = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += o;
                }
                unsigned above = incl - own;  // scores in bins above this lane's
                if (above < (unsigned)need && incl >= (unsigned)need) {
                    for (int j = 0; j < 4; ++j) {
                        if (above + h[j] >= (unsigned)need) {
                            ctl[1] = 255 - 4 * lane - j;
                            ctl[2] = need - (int)above;
                            break;
                        }
                        above += h[j];
                    }
                }
            }
            __syncthreads();
            prefix |= (unsigned)ctl[1] << shift;
            mask |= 0xFFu << shift;
            need = ctl[2];
            __syncthreads();
        }
        // everything above the threshold, then `need` of the ties in ascending item order: each thread owns a
        // contiguous slice, so that "the first `need` by index" is a prefix over threads
        const unsigned T = prefix;
        const int per = (nt + NT - 1) / NT, lo = min(nt, tid * per), hi = min(nt, lo + per);
        int ties = 0;
        for (int x = lo; x < hi; ++x) ties += keys[x] == T ? 1 : 0;
        int incl = ties;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wtot[wave] = (unsigned)incl;
        __syncthreads();
        int tie_rank = incl - ties;
        for (int w2 = 0; w2 < wave; ++w2) tie_rank += (int)wtot[w2];
        for (int x = lo; x < hi; ++x) {
            const unsigned kx = keys[x];
            bool sel = kx > T;
            if (kx == T) {
                sel = tie_rank < need;
                ++tie_rank;
            }
            if (sel) {
                const int at = atomicAdd(&ctl[0], 1);
                // ascending sort of this = score descending, then item ascending
                if (at < kTopnCand) cand[at] = ((unsigned long long)(~kx) << 32) | (unsigned)(tile0 + x);
            }
        }
        __syncthreads();
    }
    // ---- the candidates of all tiles: bitonic sort, best first -----------------------------------------
    const int nc = min(ctl[0], kTopnCand);
    int n2 = 1;
    while (n2 < nc) n2 <<= 1;
    for (int x = nc + tid; x < n2; x += NT) cand[x] = ~0ull;
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int x = tid; x < n2 / 2; x += NT) {
                const int i0 = 2 * x - (x & (stride - 1)), i1 = i0 + stride;
                const bool up = (i0 & size) == 0;
                const unsigned long long a0 = cand[i0], a1 = cand[i1];
                if ((a0 > a1) == up) {
                    cand[i0] = a1;
                    cand[i1] = a0;
                }
            }
            __syncthreads();
        }
    // the winners; their scores recomputed with the canonical dot (the bits predict() returns)
    const int iters = (topn + GPB - 1) / GPB;
    for (int it = 0; it < iters; ++it) {
        const int x = grp + it * GPB;
        const bool ok = x < topn;
        const int item = ok ? (int)(cand[x] & 0xFFFFFFFFull) : 0;
        const float4 q = *reinterpret_cast<const float4*>(Q + (size_t)item * KP + lig * 4);
        const float d = group_allreduce<L>(chunk_dot(p, q));
        if (ok && lig == 0) {
            out_s[(size_t)b * topn + x] = d;
            out_i[(size_t)b * topn + x] = item;
        }
    }
}

template <int L>
hipError_t topn_L(const float* P, const float* Q, const int32_t* users, int nb, int32_t n_items, int32_t topn, float* out_s,
                  int32_t* out_i, hipStream_t st) {
    const size_t lds = (size_t)kTopnTile * 4 + (size_t)kTopnCand * 8 + 256 * 4 + (kTopnThreads / 64) * 4 + 16;
    hipError_t e = hipFuncSetAttribute((const void*)topn_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((topn_kernel<L>), dim3((unsigned)nb), dim3(kTopnThreads), lds, st, P, Q, users, n_items, topn, out_s, out_i);
    return hipGetLastError();
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

bool recommend_is_fused(int32_t n_items, int32_t topn) {
    const long long tiles = ((long long)n_items + kTopnTile - 1) / kTopnTile;
    return topn <= kTopnFused && tiles * topn <= kTopnCand;
}

// Fused score + select: no score buffers at all.
hipError_t recommend_fused(int L, const float* P, const float* Q, const int32_t* d_users, int nb, int32_t n_items,
                           int32_t topn, float* out_s, int32_t* out_i, hipStream_t st) {
    switch (L) {
        case 1: return topn_L<1>(P, Q, d_users, nb, n_items, topn, out_s, out_i, st);
        case 2: return topn_L<2>(P, Q, d_users, nb, n_items, topn, out_s, out_i, st);
        case 4: return topn_L<4>(P, Q, d_users, nb, n_items, topn, out_s, out_i, st);
        case 8: return topn_L<8>(P, Q, d_users, nb, n_items, topn, out_s, out_i, st);
        case 16: return topn_L<16>(P, Q, d_users, nb, n_items, topn, out_s, out_i, st);
        case 32: return topn_L<32>(P, Q, d_users, nb, n_items, topn, out_s, out_i, st);
        case 64: return topn_L<64>(P, Q, d_users, nb, n_items, topn, out_s, out_i, st);
        default: return hipErrorInvalidValue;
    }
}

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
