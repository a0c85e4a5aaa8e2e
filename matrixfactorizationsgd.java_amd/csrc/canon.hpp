// canon.hpp -- the canonical arithmetic of DESIGN.md section 3 on the device: ONE definition, shared by
// the training / RMSE / predict kernels (kernels.hip) and the top-N scorer (recommend.hip), so that
// predict(), recommend() and the oracle agree bit for bit by construction.
// Build every translation unit that includes this with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace mfsgd {
namespace {

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Butterfly all-reduce over a group of L consecutive lanes (L power of two,
// group aligned to L).  Level m adds the value of lane (l xor m); after the
// lower levels every lane of a 2^j sub-group holds the same value, so the
// mirror permutations used for m = 4 and m = 8 fetch exactly that value.
// xor-16 / xor-32 butterfly levels with the gfx950 lane-swap instructions (no LDS traffic):
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows
// of its second; with both holding v, one ends up with the even row of each row pair in
// both rows and the other with the odd row, so their sum is (even + odd) in every lane --
// the same bits in both partners.  v_permlane32_swap does the same for the two 32-lane halves.
__device__ __forceinline__ float swap_add16(float v) {
    float t;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1"
                 : "+v"(v), "=&v"(t));
    return v + t;
}
__device__ __forceinline__ float swap_add32(float v) {
    float t;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1"
                 : "+v"(v), "=&v"(t));
    return v + t;
}

template <int L>
__device__ __forceinline__ float group_allreduce(float v) {
    if constexpr (L >= 2) v = v + dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]   : xor 1
    if constexpr (L >= 4) v = v + dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]   : xor 2
    if constexpr (L >= 8) v = v + dpp_move<0x141>(v);  // row_half_mirror       : other quad
    if constexpr (L >= 16) v = v + dpp_move<0x140>(v); // row_mirror            : other half-row
    if constexpr (L == 32) v = swap_add16(v);
    if constexpr (L >= 64) {
        // One rating per wave: the sum is wave-uniform, so the two upper levels need no all-reduce.  Every
        // lane of row r holds S_r; row_bcast:15 (rows 1, 3) leaves S0+S1 and S2+S3 there, row_bcast:31
        // (rows 2, 3) leaves (S2+S3)+(S0+S1) in lane 63 -- the bits of the tree above, addition being
        // commutative -- and a readlane hands it to everybody.  (Rows the masks exclude add 0.)
        v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
        v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
        v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    }
    return v;
}

__device__ __forceinline__ float chunk_dot(const float4 p, const float4 q) {
    float t0 = p.x * q.x;
    float t1 = p.y * q.y;
    t0 = __builtin_fmaf(p.z, q.z, t0);
    t1 = __builtin_fmaf(p.w, q.w, t1);
    return t0 + t1;
}

__device__ __forceinline__ float4 axpy_row(const float s, const float4 x, const float c,
                                           const float4 y) {
    // y' = fma(s, x, c*y)
    float4 o;
    o.x = __builtin_fmaf(s, x.x, c * y.x);
    o.y = __builtin_fmaf(s, x.y, c * y.y);
    o.z = __builtin_fmaf(s, x.z, c * y.z);
    o.w = __builtin_fmaf(s, x.w, c * y.w);
    return o;
}

}  // namespace
}  // namespace mfsgd
