// kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the MF-SGD hot path.
//
// No reference counterpart exists (/root/reference/README.md:1-2 is the whole
// reference).  Implements SURVEY.md 2.2 rows B1-B3 (dot, error, rank-1 update),
// B5 (LDS-staged tiles), B6 (wavefront reduction, no MFMA) and B9 (RMSE).
//
// Shape of the work: one workgroup = one cell of the
// block schedule.  The cell's touched factor rows (users AND items) are
// gathered from HBM/L2 into LDS with 16-byte-per-lane loads (one row = L lanes
// x 16 B, a wave moves 64/L rows per instruction), every rating of the cell is
// then applied out of LDS, and the rows are scattered back.  A rating occupies
// a group of L lanes (4 floats per lane); a wave applies G = 64/L ratings per
// step; the dot product is reduced inside the lane group with DPP row
// operations (no LDS traffic, no MFMA: this is gather + axpy, not a dense
// contraction).
//
// Arithmetic is the contract of DESIGN.md section 3 and must stay bit-for-bit
// what the CPU checker under oracle/ computes: build with -ffp-contract=off.
#include <hip/hip_runtime.h>

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

// Butterfly all-reduce over a group of L consecutive lanes (L power of two,
// group aligned to L).  Level m adds the value of lane (l xor m); after the
// lower levels every lane of a 2^j sub-group holds the same value, so the
// mirror permutations used for m = 4 and m = 8 fetch exactly that value.
template <int L>
__device__ __forceinline__ float group_allreduce(float v) {
    if constexpr (L >= 2) v = v + dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]   : xor 1
    if constexpr (L >= 4) v = v + dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]   : xor 2
    if constexpr (L >= 8) v = v + dpp_move<0x141>(v);  // row_half_mirror       : other quad
    if constexpr (L >= 16) v = v + dpp_move<0x140>(v); // row_mirror            : other half-row
    if constexpr (L >= 32) v = v + __shfl_xor(v, 16, 64);
    if constexpr (L >= 64) v = v + __shfl_xor(v, 32, 64);
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

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

__device__ __forceinline__ float4 lds_ld(const unsigned char* base, unsigned off) {
    return *reinterpret_cast<const float4*>(base + off);
}
__device__ __forceinline__ void lds_st(unsigned char* base, unsigned off, const float4 v) {
    *reinterpret_cast<float4*>(base + off) = v;
}

// One workgroup = one cell.  TRAIN: round `rd` runs cells (b, (b + rd) % B).
// !TRAIN: blockIdx.x is the cell index, no writes, sum of squared errors out.
//
// LDS image: [rows: nrows x ROWB][2G zero rows][entries: n_steps x G x 8][subs: W*W x 8][row ids: nrows x 4]
template <int L, int W, bool TRAIN, bool DIAG = false>
__global__ void __launch_bounds__(64 * W)
cell_kernel(float* __restrict__ P, float* __restrict__ Q, const CellDesc* __restrict__ cells,
            const uint32_t* __restrict__ rows, const SubDesc* __restrict__ subs,
            const Entry* __restrict__ entries, const int B, const int rd, const float lr,
            const float c, double* __restrict__ sse_partial) {
    constexpr int G = 64 / L;
    constexpr int ROWB = 16 * L;
    constexpr int KP = 4 * L;
    constexpr int NT = 64 * W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane / L;
    const int lig = lane % L;
    const int cell = TRAIN ? (int)blockIdx.x * B + ((int)blockIdx.x + rd) % B : (int)blockIdx.x;
    const CellDesc cd = cells[cell];
    unsigned long long stamp0 = 0, stamp1 = 0, stamp2 = 0;
    if constexpr (DIAG) stamp0 = __builtin_amdgcn_s_memtime();
    const int nu = cd.nu;
    const int nrows = (int)cd.nu + (int)cd.ni;
    if (nrows == 0) {  // uniform over the workgroup
        if (!TRAIN && tid == 0) sse_partial[cell] = 0.0;
        return;
    }
    unsigned char* const lrows = smem;
    uint2* const lent = reinterpret_cast<uint2*>(smem + (size_t)(nrows + 2 * G) * ROWB);
    uint2* const lsub = lent + (size_t)cd.n_steps * G;
    uint32_t* const lids = reinterpret_cast<uint32_t*>(lsub + W * W);

    // ---- stage the cell's schedule: row ids, step entries, sub-cell table ---------
    // (one global latency for all of it; the row gather below depends on the ids)
    const uint32_t* const crow = rows + cd.row_off;
    for (int x = tid; x < nrows; x += NT) lids[x] = crow[x];
    {
        const uint2* gent = reinterpret_cast<const uint2*>(entries) + (size_t)cd.ent_off * G;
        const int ne = (int)cd.n_steps * G;
        for (int x = tid; x < ne; x += NT) lent[x] = gent[x];
        if (tid < W * W) lsub[tid] = reinterpret_cast<const uint2*>(subs)[(size_t)cell * W * W + tid];
    }
    // idle slots of a step point at these all-zero rows: r = 0 keeps them zero
    for (int x = tid; x < 2 * G * L; x += NT)
        lds_st(lrows, (unsigned)(nrows * ROWB + x * 16), make_float4(0.f, 0.f, 0.f, 0.f));
    __syncthreads();

    // ---- gather: touched factor rows -> LDS, straight from memory (LDS-DMA) -------
    // One wave instruction moves G whole rows (64 lanes x 16 B = G x ROWB contiguous
    // LDS bytes); the source address is per lane, so it is a row gather.  Every load
    // of the wave is in flight before the single wait.
    for (int s0 = wave * G; s0 < nrows; s0 += W * G) {
        const int sx = s0 + g;
        if (sx < nrows) {
            const uint32_t rid = lids[sx];
            const float* src = (sx < nu ? P : Q) + (size_t)rid * KP + lig * 4;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lrows + (size_t)s0 * ROWB), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (DIAG) stamp1 = __builtin_amdgcn_s_memtime();

    // ---- apply the ratings out of LDS ------------------------------------------
    // Software pipeline: the rows of step t+1 are read before the rows of step t are
    // written back.  The scheduler guarantees (schedule.cpp, "Eligibility") that a
    // row read that early was not written in step t, except a q-side row in the same
    // lane slot, which is flagged and taken from registers instead.
    double acc = 0.0;
    const unsigned laneoff = (unsigned)lig * 16u;
    // Two register sets (A, B) alternate between "current step" and "next step", so
    // the loop is unrolled by two and nothing is copied between iterations.
    struct StepRegs {
        uint2 en;      // entry: addresses | forward flag, rating
        unsigned pa, qa;
        float4 p, q;
    };
    auto set_addr = [&](StepRegs& x) {
        x.pa = ((x.en.x & 0xFFFFu) << 4) + laneoff;
        x.qa = (__builtin_amdgcn_ubfe(x.en.x, 16, 15) << 4) + laneoff;
    };
    // One step: `cur` holds step t (entry, addresses, rows); `nxt.en` holds entry t+1.
    // Leaves `nxt` complete for step t+1 and cur.en = entry t+2 (read from eptr[e2]).
    auto step = [&](StepRegs& cur, StepRegs& nxt, const uint2* eptr, const int e2) {
        const float r = __builtin_bit_cast(float, cur.en.y);
        set_addr(nxt);
        const float4 pn = lds_ld(lrows, nxt.pa);
        const float4 qn = lds_ld(lrows, nxt.qa);
        cur.en = eptr[e2];
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the arithmetic
        const float dot = group_allreduce<L>(chunk_dot(cur.p, cur.q));
        const float err = r - dot;
        if constexpr (TRAIN) {
            const float sc = lr * err;
            const float4 p2 = axpy_row(sc, cur.q, c, cur.p);
            const float4 q2 = axpy_row(sc, cur.p, c, cur.q);
            lds_st(lrows, cur.pa, p2);
            lds_st(lrows, cur.qa, q2);
            const bool fwd = (int)nxt.en.x < 0;
            nxt.q.x = fwd ? q2.x : qn.x;
            nxt.q.y = fwd ? q2.y : qn.y;
            nxt.q.z = fwd ? q2.z : qn.z;
            nxt.q.w = fwd ? q2.w : qn.w;
        } else {
            acc += (double)err * (double)err;
            nxt.q = qn;
        }
        nxt.p = pn;
    };
    // Run step: the slot's q row is resident in `rq` for the whole run (no q load, no
    // select, no q store); idle slots (flag bit) skip the update.
    float4 rq;
    auto run_step = [&](StepRegs& cur, StepRegs& nxt, const uint2* eptr, const int e2) {
        const float r = __builtin_bit_cast(float, cur.en.y);
        const bool active = (int)cur.en.x >= 0;
        nxt.pa = ((nxt.en.x & 0xFFFFu) << 4) + laneoff;
        const float4 pn = lds_ld(lrows, nxt.pa);
        cur.en = eptr[e2];
        const float dot = group_allreduce<L>(chunk_dot(cur.p, rq));
        const float err = r - dot;
        if constexpr (TRAIN) {
            // idle slot: s = 0 and c = 1 leave the resident row bit-identical
            // (fma(0, p, 1*q) == q) and rewrite zeros to the all-zero p row.
            const float sc = active ? lr * err : 0.0f;
            const float ce = active ? c : 1.0f;
            const float4 p2 = axpy_row(sc, rq, ce, cur.p);
            rq = axpy_row(sc, cur.p, ce, rq);
            lds_st(lrows, cur.pa, p2);
        } else {
            acc += (double)err * (double)err;  // idle: p row and r are zero, err == 0
        }
        nxt.p = pn;
    };
    for (int s = 0; s < W; ++s) {
        const uint2 sd = lsub[s * W + wave];
        const int nall = __builtin_amdgcn_readfirstlane((int)sd.y);
        const int n = nall & 0xFFFF;   // general steps
        const int nr = nall >> 16;     // run steps, stored after the general ones
        // entries of this wave's sub-cell; the host pads every cell with two idle
        // steps, so reading entries t+1 and t+2 past the end stays inside the image
        const uint2* ebase = lent + (size_t)__builtin_amdgcn_readfirstlane((int)sd.x) * G + g;
        if (n > 0) {
            const uint2* eptr = ebase;
            StepRegs A, B;
            A.en = eptr[0];
            B.en = eptr[G];
            set_addr(A);
            A.p = lds_ld(lrows, A.pa);
            A.q = lds_ld(lrows, A.qa);
            int t = 0;
            for (; t + 1 < n; t += 2, eptr += 2 * G) {
                step(A, B, eptr, 2 * G);
                step(B, A, eptr, 3 * G);
            }
            if (t < n) step(A, B, eptr, 2 * G);
        }
        if (nr > 0) {
            const uint2* eptr = ebase + (size_t)n * G;
            StepRegs A, B;
            A.en = eptr[0];
            B.en = eptr[G];
            // every run entry of a slot carries the slot's item address
            const unsigned rqa = (__builtin_amdgcn_ubfe(A.en.x, 16, 15) << 4) + laneoff;
            A.pa = ((A.en.x & 0xFFFFu) << 4) + laneoff;
            rq = lds_ld(lrows, rqa);
            A.p = lds_ld(lrows, A.pa);
            int t = 0;
            for (; t + 1 < nr; t += 2, eptr += 2 * G) {
                run_step(A, B, eptr, 2 * G);
                run_step(B, A, eptr, 3 * G);
            }
            if (t < nr) run_step(A, B, eptr, 2 * G);
            if constexpr (TRAIN) lds_st(lrows, rqa, rq);
        }
        if constexpr (TRAIN) __syncthreads();
    }

    if constexpr (DIAG) stamp2 = __builtin_amdgcn_s_memtime();
    if constexpr (TRAIN) {
        // ---- scatter: LDS -> rows ----------------------------------------------
        constexpr int UNR = 4;
        int s = wave * G + g;
        for (; s + (UNR - 1) * W * G < nrows; s += UNR * W * G) {
            uint32_t rid[UNR];
            float4 v[UNR];
#pragma unroll
            for (int x = 0; x < UNR; ++x) {
                rid[x] = lids[s + x * W * G];
                v[x] = lds_ld(lrows, (unsigned)((s + x * W * G) * ROWB) + laneoff);
            }
#pragma unroll
            for (int x = 0; x < UNR; ++x) {
                const int sx = s + x * W * G;
                float* dst = (sx < nu ? P : Q) + (size_t)rid[x] * KP + lig * 4;
                *reinterpret_cast<float4*>(dst) = v[x];
            }
        }
        for (; s < nrows; s += W * G) {
            const uint32_t rid = lids[s];
            float* dst = (s < nu ? P : Q) + (size_t)rid * KP + lig * 4;
            *reinterpret_cast<float4*>(dst) = lds_ld(lrows, (unsigned)(s * ROWB) + laneoff);
        }
        if constexpr (DIAG) {
            // diagnostic build only: phase stamps of this workgroup (shader clock ticks)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long stamp3 = __builtin_amdgcn_s_memtime();
            if (tid == 0) {
                unsigned long long* o = reinterpret_cast<unsigned long long*>(sse_partial) + (size_t)blockIdx.x * 4;
                o[0] = stamp0;
                o[1] = stamp1;
                o[2] = stamp2;
                o[3] = stamp3;
            }
        }
    } else {
        // ---- deterministic sum of squared errors --------------------------------
        // every lane of a group carries the group's sum: keep one copy, then a
        // fixed butterfly over the wave, then waves in index order.
        double v = lig == 0 ? acc : 0.0;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
        __syncthreads();  // everyone is done reading lsub/lent before they are reused
        double* wsum = reinterpret_cast<double*>(lent);
        if (lane == 0) wsum[wave] = v;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < W; ++w) t += wsum[w];
            sse_partial[cell] = t;
        }
    }
}

// Fixed-order reduction of the per-cell partial sums (one workgroup).
__global__ void __launch_bounds__(256) reduce_sse_kernel(const double* __restrict__ partial,
                                                         const int64_t n, double* __restrict__ out) {
    __shared__ double sh[256];
    double t = 0.0;
    for (int64_t x = threadIdx.x; x < n; x += 256) t += partial[x];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) sh[threadIdx.x] += sh[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

// out[j] = dot(P[u[j]], Q[i[j]]); one lane group per pair.
template <int L>
__global__ void __launch_bounds__(256) predict_kernel(const float* __restrict__ P,
                                                      const float* __restrict__ Q,
                                                      const int32_t* __restrict__ u,
                                                      const int32_t* __restrict__ i,
                                                      float* __restrict__ out, const int64_t n) {
    constexpr int KP = 4 * L;
    constexpr int GPB = 256 / L;  // groups per block
    const int lig = threadIdx.x % L;
    const int64_t grp0 = (int64_t)blockIdx.x * GPB + threadIdx.x / L;
    const int64_t stride = (int64_t)gridDim.x * GPB;
    // all lanes of a wave run the same number of iterations (DPP needs them live)
    const int64_t iters = (n + stride - 1) / stride;
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t j = grp0 + it * stride;
        const bool ok = j < n;
        const int64_t jj = ok ? j : 0;
        const float4 p = *reinterpret_cast<const float4*>(P + (size_t)u[jj] * KP + lig * 4);
        const float4 q = *reinterpret_cast<const float4*>(Q + (size_t)i[jj] * KP + lig * 4);
        const float d = group_allreduce<L>(chunk_dot(p, q));
        if (ok && lig == 0) out[j] = d;
    }
}

template <int L, int W>
hipError_t launch_cell_LW(bool train, const CellLaunch& a, hipStream_t st) {
    const void* fn = train ? (const void*)cell_kernel<L, W, true> : (const void*)cell_kernel<L, W, false>;
    // > 64 KiB of dynamic LDS has to be granted per function; cheap to repeat.
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, a.lds_bytes);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)a.grid), block(64 * W);
    if (train && a.diag) {
        const void* fd = (const void*)cell_kernel<L, W, true, true>;
        e = hipFuncSetAttribute(fd, hipFuncAttributeMaxDynamicSharedMemorySize, a.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((cell_kernel<L, W, true, true>), grid, block, (size_t)a.lds_bytes, st, a.P, a.Q,
                           a.cells, a.rows, a.subs, a.entries, a.B, a.rd, a.lr, a.c, a.sse_partial);
    } else if (train)
        hipLaunchKernelGGL((cell_kernel<L, W, true>), grid, block, (size_t)a.lds_bytes, st, a.P, a.Q,
                           a.cells, a.rows, a.subs, a.entries, a.B, a.rd, a.lr, a.c, a.sse_partial);
    else
        hipLaunchKernelGGL((cell_kernel<L, W, false>), grid, block, (size_t)a.lds_bytes, st, a.P, a.Q,
                           a.cells, a.rows, a.subs, a.entries, a.B, a.rd, a.lr, a.c, a.sse_partial);
    return hipGetLastError();
}

template <int L>
hipError_t launch_cell_L(bool train, int W, const CellLaunch& a, hipStream_t st) {
    switch (W) {
        case 1: return launch_cell_LW<L, 1>(train, a, st);
        case 2: return launch_cell_LW<L, 2>(train, a, st);
        case 4: return launch_cell_LW<L, 4>(train, a, st);
        case 8: return launch_cell_LW<L, 8>(train, a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

hipError_t launch_cell(bool train, int L, int W, const CellLaunch& a, hipStream_t st) {
    switch (L) {
        case 1: return launch_cell_L<1>(train, W, a, st);
        case 2: return launch_cell_L<2>(train, W, a, st);
        case 4: return launch_cell_L<4>(train, W, a, st);
        case 8: return launch_cell_L<8>(train, W, a, st);
        case 16: return launch_cell_L<16>(train, W, a, st);
        case 32: return launch_cell_L<32>(train, W, a, st);
        case 64: return launch_cell_L<64>(train, W, a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_reduce_sse(const double* partial, int64_t n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_sse_kernel, dim3(1), dim3(256), 0, st, partial, n, out);
    return hipGetLastError();
}

hipError_t launch_predict(int L, const float* P, const float* Q, const int32_t* u, const int32_t* i,
                          float* out, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int gpb = 256 / L;
    int64_t blocks = (n + gpb - 1) / gpb;
    if (blocks > 4096) blocks = 4096;
    const dim3 grid((unsigned)blocks), block(256);
    switch (L) {
#define MFSGD_PRED(LL)                                                                     \
    case LL:                                                                               \
        hipLaunchKernelGGL((predict_kernel<LL>), grid, block, 0, st, P, Q, u, i, out, n); \
        break;
        MFSGD_PRED(1)
        MFSGD_PRED(2)
        MFSGD_PRED(4)
        MFSGD_PRED(8)
        MFSGD_PRED(16)
        MFSGD_PRED(32)
        MFSGD_PRED(64)
#undef MFSGD_PRED
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mfsgd
