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

#include "canon.hpp"
#include "kernels.hpp"
#include "run_asm.hpp"

#pragma clang fp contract(off)


namespace mfsgd {

namespace {

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() would also
// drain the vector-memory counter, i.e. any LDS-DMA prefetch still in flight.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float4 lds_ld(const unsigned char* base, unsigned off) {
    return *reinterpret_cast<const float4*>(base + off);
}
__device__ __forceinline__ void lds_st(unsigned char* base, unsigned off, const float4 v) {
    *reinterpret_cast<float4*>(base + off) = v;
}

// (the hand-scheduled run loop text lives in run_asm.hpp, shared with tools/ubench3.hip)
template <int EST, int L>
__device__ __forceinline__ void run_loop_asm(float4& rq, const unsigned ea, const unsigned rowbase, int pairs,
                                             const float lr) {
    static_assert(L == 16 || L == 32 || L == 64, "hand-scheduled run loop: 16, 32 or 64 lanes per rating");
    using f4 = __attribute__((ext_vector_type(4))) float;
    f4 q = {rq.x, rq.y, rq.z, rq.w};
    constexpr int PADV = mfsgd_pad_run(L);
    if constexpr (L == 16)
        asm volatile(MFSGD_RUN_LOOP_ASM_TEXT("", MFSGD_SFMA_V) MFSGD_RUN_LOOP_ASM_OPERANDS);
    else if constexpr (L == 32)
        asm volatile(MFSGD_RUN_LOOP_ASM_TEXT(MFSGD_SWAP_ADD16, MFSGD_SFMA_V) MFSGD_RUN_LOOP_ASM_OPERANDS);
    else
        asm volatile(MFSGD_RUN_LOOP_ASM_TEXT(MFSGD_BCAST_ADD64, MFSGD_SFMA_S) MFSGD_RUN_LOOP_ASM_OPERANDS);
    rq = make_float4(q[0], q[1], q[2], q[3]);
}

// ---- general steps (run_asm.hpp, MFSGD_GEN_LOOP_ASM_TEXT) ---------------------------------
// `ea`: LDS byte address of this lane group's entry of step 0 (stride EST), n >= 1 steps, c2 = {c, c}.
template <int EST, int L>
__device__ __forceinline__ void gen_loop_asm(const unsigned ea, const unsigned rowbase, int n, const float lr, const uint64_t c2) {
    static_assert(L == 16 || L == 32 || L == 64, "hand-scheduled general loop: 16, 32 or 64 lanes per rating");
    n = __builtin_amdgcn_readfirstlane(n);
    constexpr int PADV = mfsgd_pad_gen(L);
    if constexpr (L == 16)
        asm volatile(MFSGD_GEN_LOOP_ASM_TEXT("", MFSGD_SFMA_V) MFSGD_GEN_LOOP_ASM_OPERANDS);
    else if constexpr (L == 32)
        asm volatile(MFSGD_GEN_LOOP_ASM_TEXT(MFSGD_SWAP_ADD16, MFSGD_SFMA_V) MFSGD_GEN_LOOP_ASM_OPERANDS);
    else
        asm volatile(MFSGD_GEN_LOOP_ASM_TEXT(MFSGD_BCAST_ADD64, MFSGD_SFMA_S) MFSGD_GEN_LOOP_ASM_OPERANDS);
}

// ---- solo run: chain wave / helper wave (run_asm.hpp) -------------------------------------
// `ea`: LDS byte address of the run's header record, `rowbase`: LDS byte address of row slot 0 plus
// this lane's 16-byte offset inside a row, n >= 1 steps, c2 = {c, c} as one 64-bit scalar.
template <int L>
__device__ __forceinline__ void solo_chain_asm(float4& rq, const unsigned ea, const unsigned rowbase, int n,
                                               const float lr, const uint64_t c2) {
    static_assert(L == 16 || L == 32 || L == 64, "solo loops: 16, 32 or 64 lanes per rating");
    using f4 = __attribute__((ext_vector_type(4))) float;
    f4 q = {rq.x, rq.y, rq.z, rq.w};
    n = __builtin_amdgcn_readfirstlane(n);
    constexpr int PADV = mfsgd_pad_chain(L);
    if constexpr (L == 16)
        asm volatile(MFSGD_SOLO_CHAIN_ASM_TEXT("", MFSGD_SFMA2_V) MFSGD_SOLO_CHAIN_OPERANDS);
    else if constexpr (L == 32)
        asm volatile(MFSGD_SOLO_CHAIN_ASM_TEXT(MFSGD_BCAST_ADD32, MFSGD_SFMA2_S) MFSGD_SOLO_CHAIN_OPERANDS);
    else
        asm volatile(MFSGD_SOLO_CHAIN_ASM_TEXT(MFSGD_BCAST_ADD64, MFSGD_SFMA2_S) MFSGD_SOLO_CHAIN_OPERANDS);
    rq = make_float4(q[0], q[1], q[2], q[3]);  // q after the n steps (the helper stores it; the caller needs it when it cuts a run)
}
// Returns false if it gave up waiting for the chain wave (bounded polling; cannot happen with a
// schedule the packer built -- the bound only keeps a corrupt one from hanging the GPU).
template <int L>
__device__ __forceinline__ bool solo_helper_asm(const unsigned ea, const unsigned rowbase, int n, const uint64_t c2,
                                                int fin = 1) {  // fin = 0: do not store q at the end (ubench3's cut runs)
    constexpr int PADV = mfsgd_pad_helper(L);
    n = __builtin_amdgcn_readfirstlane(n);  // (workgroup-uniform by construction; the compiler cannot always see it)
    fin = __builtin_amdgcn_readfirstlane(fin);
    asm volatile("" : "+s"(fin));  // a register, not an immediate, in the text below
    int spins = 1 << 22;
    asm volatile(MFSGD_SOLO_HELPER_ASM_TEXT MFSGD_SOLO_HELPER_OPERANDS);
    return spins != 0;
}

// -DMFSGD_GEN_STEP_CPP: the compiler-scheduled general step everywhere (A/B measurements; it stays the reference form
// of the step and is what k <= 32 and the RMSE pass run)
#ifdef MFSGD_GEN_STEP_CPP
constexpr bool kGenStepCpp = true;
#else
constexpr bool kGenStepCpp = false;
#endif

// copy waves of the persistent training kernel: as many again as apply waves, up to 8 waves in all
// (16 waves would leave each only 128 VGPRs; the assembly run loop uses v100..v143)
template <int L, int W>
constexpr int epoch_helpers() {
    return W <= 4 ? W : 0;
}

// A chunk descriptor through the scalar path: the index is the same in every lane (it comes from
// workgroup-uniform counters and from descriptors loaded this way), which the compiler cannot see
// once it has been through memory -- pin it to an SGPR so that the load is an s_load and the
// descriptor lives in SGPRs (it is live across the rating loops, where VGPRs are scarce).
__device__ __forceinline__ CellDesc load_desc(const CellDesc* __restrict__ cells, unsigned idx) {
    idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
    asm volatile("" : "+s"(idx));
    return cells[idx];
}

// Everything one workgroup does with one cell, phase by phase.  Shared by the
// per-round kernel, the SSE pass and the persistent epoch kernel.
//
// LDS image of a workgroup:
//   [control block 16 B][schedule buffer 0: sched_cap][schedule buffer 1: sched_cap][rows ...]
//   schedule buffer: [entries: n_steps x G x 16][subs: W*W x 8, padded to 16][row ids: nrows x 4]
//   rows: [nrows x ROWB][2G zero rows]; slots [0, nu) hold p-side (user) rows, [nu, nrows) q-side (item) rows.
// Two schedule buffers: the persistent kernel fetches the next cell's schedule (LDS-DMA)
// while the current cell is being worked on.
//
// NH copy waves (0, or W in the persistent training kernel): waves W .. W+NH-1.  They take part
// in staging, gathers and scatters like any other wave -- an LDS-DMA gather or a scatter is bound by
// how fast a wave can issue (~150 cycles per LDS-DMA instruction), so more waves shorten those
// phases -- and only keep the barriers company in apply(), whose W waves own the sub-cells.
template <int L, int W, int NH = 0>
struct Cell {
    static constexpr int G = 64 / L;
    static constexpr int ROWB = 16 * L;
    static constexpr int KP = 4 * L;
    static constexpr int NWV = W + NH;   // waves of the workgroup
    static constexpr int NT = 64 * NWV;
    static constexpr int CTL = 16;  // control block at the start of the dynamic LDS
    static constexpr int SUBB = (W * W * 8 + 15) & ~15;  // sub-cell table, padded to 16-byte units

    int tid, lane, wave, g, lig;
    int wave_all;      // index among all NWV waves (copy loops); `wave` is the sub-cell owner index
    bool helper;       // this wave is a copy wave (NH > 0 only)
    unsigned laneoff;
    int nu, nrows, n_steps;
    bool critical;  // the cell carries a long per-row chain (scheduler flag)
    volatile unsigned* fail_flag = nullptr;  // LDS control word a helper raises when it gives up (persistent kernel)
    // [r3] Early hand-off of a tile that is ONE row (run_ring's mailboxes): when the cell's last work is a solo run, the
    // chain wave posts the row from its registers the moment the run ends -- {value, tag} granules at post_at, tag
    // post_tag -- instead of leaving it to the workgroup behind the helper's stores, the sub-round barriers and an LDS
    // round trip; it says so in *posted_flag (an LDS control word), and the workgroup does not post again.
    unsigned long long* post_at = nullptr;
    unsigned post_tag = 0;
    volatile unsigned* posted_flag = nullptr;
    unsigned char* lrows;
    uint4* lent;
    uint2* lsub;
    uint32_t* lids;

    __device__ __forceinline__ void init_thread() {
        tid = threadIdx.x;
        lane = tid & 63;
        wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
        helper = NH > 0 && wave_all >= W;
        wave = helper ? wave_all - W : wave_all;
        g = lane / L;
        lig = lane % L;
        laneoff = (unsigned)lig * 16u;
    }
    __device__ __forceinline__ void bind(const CellDesc& cd, unsigned char* smem, int buf, int sched_cap) {
        nu = cd.nu;
        nrows = (int)cd.nu + (int)cd.ni;
        n_steps = (int)(cd.n_steps & 0x7FFFFFFFu);
        critical = (cd.n_steps >> 31) != 0;
        lrows = smem + CTL + 2 * (size_t)sched_cap;
        lent = reinterpret_cast<uint4*>(smem + CTL + (size_t)buf * sched_cap);
        lsub = reinterpret_cast<uint2*>(lent + (size_t)n_steps * G);
        lids = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(lsub) + SUBB);
    }

    // Zeroes the 2G rows idle step slots point at (r = 0 keeps them zero).
    __device__ __forceinline__ void zero_idle_rows() {
        for (int x = tid; x < 2 * G * L; x += NT)
            lds_st(lrows, (unsigned)(nrows * ROWB + x * 16), make_float4(0.f, 0.f, 0.f, 0.f));
    }

    // The schedule of ANOTHER cell -> schedule buffer `buf`, by LDS-DMA (no registers held,
    // nothing waited for here): entries, sub-cell table, row ids, each a contiguous copy of
    // whole 16-byte units.  The caller waits (vmcnt) and barriers before binding that buffer.
    __device__ __forceinline__ void prefetch_schedule(const CellDesc& nd, int ncell, unsigned char* smem, int buf,
                                                      int sched_cap, const uint32_t* __restrict__ rows,
                                                      const SubDesc* __restrict__ subs,
                                                      const Entry* __restrict__ entries) {
        unsigned char* const dst = smem + CTL + (size_t)buf * sched_cap;
        const int nn = (int)nd.nu + (int)nd.ni;
        const int ent_bytes = (int)(nd.n_steps & 0x7FFFFFFFu) * G * 16;
        const int sub_bytes = SUBB;
        const int ids_bytes = (nn * 4 + 15) & ~15;
        auto copy = [&](const unsigned char* src, unsigned char* d, int bytes) {
            for (int off0 = wave_all * 1024; off0 < bytes; off0 += NWV * 1024) {
                const int off = off0 + lane * 16;
                if (off < bytes)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + off), (lptr_t)(d + off0), 16, 0, 0);
            }
        };
        copy(reinterpret_cast<const unsigned char*>(entries + (size_t)nd.ent_off * G), dst, ent_bytes);
        copy(reinterpret_cast<const unsigned char*>(subs + (size_t)ncell * W * W), dst + ent_bytes, sub_bytes);
        copy(reinterpret_cast<const unsigned char*>(rows + nd.row_off), dst + ent_bytes + sub_bytes, ids_bytes);
    }

    // Row ids, step entries and the sub-cell table -> LDS; zeroes the idle rows.
    // One global latency for all of it.  Caller barriers before using any of it.
    __device__ __forceinline__ void stage_schedule(const CellDesc& cd, int cell, const uint32_t* __restrict__ rows,
                                                   const SubDesc* __restrict__ subs,
                                                   const Entry* __restrict__ entries) {
        const uint32_t* const crow = rows + cd.row_off;
        for (int x = tid; x < nrows; x += NT) lids[x] = crow[x];
        const uint4* gent = reinterpret_cast<const uint4*>(entries) + (size_t)cd.ent_off * G;
        const int ne = n_steps * G;
        for (int x = tid; x < ne; x += NT) lent[x] = gent[x];
        if (tid < W * W) lsub[tid] = reinterpret_cast<const uint2*>(subs)[(size_t)cell * W * W + tid];
        zero_idle_rows();
    }

    // Factor rows of LDS slots [lo, hi) -> LDS, straight from memory (LDS-DMA).  One
    // wave instruction moves G whole rows (64 lanes x 16 B = G x ROWB contiguous LDS
    // bytes); the source address is per lane, so it is a row gather.  Issues every
    // load of the wave back to back and does NOT wait: caller does vmcnt(0) + barrier.
    // COH: the loads carry sc1 (they bypass this CU's L1), for rows another workgroup stored write-through inside
    // the same launch -- the ring hand-off then needs no acquire fence in front of them (Guideline 16, form R1
    // with sc1 loads in place of the acquire).
    template <bool COH = false>
    __device__ __forceinline__ void gather(const float* __restrict__ P, const float* __restrict__ Q, int lo, int hi) {
        constexpr int AUX = COH ? 16 : 0;  // cache policy bits of the builtin: 16 = sc1
        constexpr int UNR = 4;  // row ids of UNR instructions are fetched before any of them is issued
        const int first = (lo / G) * G;  // keep wave instructions aligned to G-slot groups
        int s0 = first + wave_all * G;
        for (; s0 + (UNR - 1) * NWV * G < hi; s0 += UNR * NWV * G) {
            uint32_t rid[UNR];
            bool in[UNR];
#pragma unroll
            for (int x = 0; x < UNR; ++x) {
                const int sx = s0 + x * NWV * G + g;
                in[x] = sx >= lo && sx < hi;
                rid[x] = lids[in[x] ? sx : lo];
            }
#pragma unroll
            for (int x = 0; x < UNR; ++x) {
                const int sb = s0 + x * NWV * G;
                if (in[x]) {
                    const float* src = (sb + g < nu ? P : Q) + (size_t)rid[x] * KP + lig * 4;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lrows + (size_t)sb * ROWB), 16, 0, AUX);
                }
            }
        }
        for (; s0 < hi; s0 += NWV * G) {
            const int sx = s0 + g;
            if (sx >= lo && sx < hi) {
                const uint32_t rid = lids[sx];
                const float* src = (sx < nu ? P : Q) + (size_t)rid * KP + lig * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lrows + (size_t)s0 * ROWB), 16, 0, AUX);
            }
        }
    }

    // LDS slots [lo, hi) -> factor rows.  WT: write-through (sc1) stores, for rows another
    // workgroup will read inside the same launch (cdna guide, Guideline 16, form R1).
    template <bool WT>
    __device__ __forceinline__ void scatter(float* __restrict__ P, float* __restrict__ Q, int lo, int hi) {
        constexpr int UNR = 4;
        auto put = [&](int s, uint32_t rid, const float4 v) {
            float* dst = (s < nu ? P : Q) + (size_t)rid * KP + lig * 4;
            if constexpr (WT) {
                const f32x4 vv = {v.x, v.y, v.z, v.w};
                // hipcc pads nothing inside asm: a VALU write of a >64-bit store operand needs a wait
                // state before the store reads it, and the operands must not be rewritten right after
                asm volatile("s_nop 1\n\tglobal_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
            } else {
                *reinterpret_cast<float4*>(dst) = v;
            }
        };
        int s = lo + wave_all * G + g;
        for (; s + (UNR - 1) * NWV * G < hi; s += UNR * NWV * G) {
            uint32_t rid[UNR];
            float4 v[UNR];
#pragma unroll
            for (int x = 0; x < UNR; ++x) {
                rid[x] = lids[s + x * NWV * G];
                v[x] = lds_ld(lrows, (unsigned)((s + x * NWV * G) * ROWB) + laneoff);
            }
#pragma unroll
            for (int x = 0; x < UNR; ++x) put(s + x * NWV * G, rid[x], v[x]);
        }
        for (; s < hi; s += NWV * G) put(s, lids[s], lds_ld(lrows, (unsigned)(s * ROWB) + laneoff));
    }

    // ---- apply the ratings out of LDS ------------------------------------------
    // Software pipeline: the rows of step t+1 are read before the rows of step t are
    // written back.  The scheduler guarantees (schedule.cpp, "Eligibility") that a
    // row read that early was not written in step t, except a q-side row in the same
    // lane slot, which is flagged and taken from registers instead.
    struct StepRegs {
        uint4 en;  // entry: addresses | flag, rating, lr*rating, decay factor
        unsigned pa, qa;
        float4 p, q;
    };

    template <bool TRAIN, bool TIMED = false>
    __device__ __forceinline__ void apply(const float lr, const float c, double& acc,
                                          unsigned long long* timers = nullptr) {
        unsigned char* const lr_ = lrows;
        const unsigned lo = laneoff;
        if constexpr (NH > 0 && TRAIN) {
            if (helper) {
                // Copy waves keep the barriers company: one per sub-round like everybody else.  Copy wave h
                // is also the HELPER of apply wave (h + 1) % W -- a wave on another SIMD -- whenever that
                // wave's sub-cell ends in a solo run: it follows the chain wave through the run's mailboxes,
                // redoes the q recurrence and does all the stores (run_asm.hpp).
                for (int s = 0; s < W; ++s) {
                    if constexpr (L >= 16) {
                        const uint64_t c2 = ((uint64_t)__builtin_bit_cast(unsigned, c) << 32) | __builtin_bit_cast(unsigned, c);
                        const unsigned rowbase = (unsigned)(uintptr_t)(lptr_t)lr_ + lo;
                        // header record of apply wave a's solo run in this sub-round, and its length
                        auto run_of = [&](const int a, int& ns) -> const uint4* {
                            const uint2 sd = lsub[s * W + a];
                            ns = __builtin_amdgcn_readfirstlane((int)(sd.x >> 16));
                            const int first = __builtin_amdgcn_readfirstlane((int)((sd.x & 0xFFFFu) + (sd.y & 0xFFFFu) + (sd.y >> 16))) + kSoloPad;
                            return lent + (size_t)first * G;
                        };
                        int ns;
                        const uint4* hdr = run_of((wave + 1) % W, ns);
                        // (cutting a long run in two and giving the second half to a second, idle copy wave was
                        // measured -- tools/ubench3 mode 4: 128.7 against 133.5 cycles per step at 16 lanes, no gain
                        // at 32 / 64; in situ 4.120 against 4.125 ms per epoch -- and is not done)
                        if (ns > 0 && !solo_helper_asm<L>((unsigned)(uintptr_t)(lptr_t)hdr, rowbase, ns, c2) && fail_flag)
                            *fail_flag = 1;
                    }
                    wg_barrier();
                }
                return;
            }
        }
        auto set_addr = [&](StepRegs& x) {
            x.pa = ((x.en.x & 0xFFFFu) << 4) + lo;
            x.qa = (__builtin_amdgcn_ubfe(x.en.x, 16, 15) << 4) + lo;
        };
        // One general step: `cur` holds step t (entry, addresses, rows); `nxt.en` holds
        // entry t+1.  Leaves `nxt` complete for step t+1 and cur.en = entry t+2.
        // Two register sets alternate roles, so the loop is unrolled by two and nothing
        // is copied between iterations.
        auto step = [&](StepRegs& cur, StepRegs& nxt, const uint4* eptr, const int e2) {
            __builtin_amdgcn_sched_barrier(0);  // the prefetch below must not climb into the previous step
            asm volatile("" : "+v"(nxt.en.x));   // ... nor its address arithmetic (no instruction emitted)
            const float r = __builtin_bit_cast(float, TRAIN ? cur.en.z : cur.en.y);  // lr*r when training
            set_addr(nxt);
            const float4 pn = lds_ld(lr_, nxt.pa);
            const float4 qn = lds_ld(lr_, nxt.qa);
            cur.en = eptr[e2];
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the arithmetic
            const float dot = group_allreduce<L>(chunk_dot(cur.p, cur.q));
            if constexpr (TRAIN) {
                const float sc = __builtin_fmaf(-lr, dot, r);  // lr*(r - dot): one dependent operation
                const float4 p2 = axpy_row(sc, cur.q, c, cur.p);
                const float4 q2 = axpy_row(sc, cur.p, c, cur.q);
                lds_st(lr_, cur.pa, p2);
                lds_st(lr_, cur.qa, q2);
                const bool fwd = (int)nxt.en.x < 0;
                nxt.q.x = fwd ? q2.x : qn.x;
                nxt.q.y = fwd ? q2.y : qn.y;
                nxt.q.z = fwd ? q2.z : qn.z;
                nxt.q.w = fwd ? q2.w : qn.w;
            } else {
                const float err = r - dot;
                acc += (double)err * (double)err;
                nxt.q = qn;
            }
            nxt.p = pn;
        };
        // Run step: the slot's q row is resident in `rq` for the whole run (no q load, no
        // select, no q store); idle slots are flagged.
        float4 rq;
        auto run_step = [&](StepRegs& cur, StepRegs& nxt, const uint4* eptr, const int e2) {
            __builtin_amdgcn_sched_barrier(0);  // the prefetch below must not climb into the previous step
            asm volatile("" : "+v"(nxt.en.x));   // ... nor its address arithmetic (no instruction emitted)
            const float r = __builtin_bit_cast(float, TRAIN ? cur.en.z : cur.en.y);
            const float ce = __builtin_bit_cast(float, cur.en.w);
            nxt.pa = ((nxt.en.x & 0xFFFFu) << 4) + lo;
            const float4 pn = lds_ld(lr_, nxt.pa);
            cur.en = eptr[e2];
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the arithmetic
            const float dot = group_allreduce<L>(chunk_dot(cur.p, rq));
            if constexpr (TRAIN) {
                // idle slot: p = 0 and r = 0 give s == 0, and its entry carries ce = 1, so the
                // resident row stays bit-identical (fma(0, p, 1*q) == q) and zeros are
                // rewritten to the all-zero p row: no flag test, no select.
                const float sc = __builtin_fmaf(-lr, dot, r);
                const float4 p2 = axpy_row(sc, rq, ce, cur.p);
                rq = axpy_row(sc, cur.p, ce, rq);
                lds_st(lr_, cur.pa, p2);
            } else {
                const float err = r - dot;
                acc += (double)err * (double)err;  // idle: p row and r are zero, err == 0
            }
            nxt.p = pn;
        };
        // Training: sub-round s, this wave's sub-cell, a barrier after every sub-round.  The RMSE pass
        // writes nothing, so its sub-cells are independent: every wave of the workgroup (copy waves
        // included) takes sub-cells wave_all, wave_all + NWV, ... with no barrier in between.
        const int n_iter = TRAIN ? W : (W * W - wave_all + NWV - 1) / NWV;
        for (int s = 0; s < n_iter; ++s) {
            const uint2 sd = lsub[TRAIN ? s * W + wave : wave_all + s * NWV];
            const int nall = __builtin_amdgcn_readfirstlane((int)sd.y);
            const int n = nall & 0xFFFF;  // general steps
            const int nr = (int)((unsigned)nall >> 16);  // run steps, stored after the general ones
            const int offs = __builtin_amdgcn_readfirstlane((int)sd.x);
            const int nsolo = (int)((unsigned)offs >> 16);  // solo records, stored after the run steps
            // entries of this wave's sub-cell; the host pads every cell with two idle
            // steps, so reading entries t+1 and t+2 past the end stays inside the image
            const uint4* ebase = lent + (size_t)(offs & 0xFFFF) * G + g;
            unsigned long long tm0 = 0, tm1 = 0, tm2 = 0;
            if constexpr (TIMED) tm0 = __builtin_amdgcn_s_memtime();
            if (TRAIN && (L == 16 || L == 32 || L == 64) && n > 0 && !kGenStepCpp) {
                // hand-scheduled form of the loop below (k in 33..256)
                if constexpr (L == 16 || L == 32 || L == 64) {
                    const uint64_t c2 = ((uint64_t)__builtin_bit_cast(unsigned, c) << 32) | __builtin_bit_cast(unsigned, c);
                    gen_loop_asm<G * 16, L>((unsigned)(uintptr_t)(lptr_t)ebase, (unsigned)(uintptr_t)(lptr_t)lr_ + lo, n, lr, c2);
                }
            } else if (n > 0) {
                const uint4* eptr = ebase;
                StepRegs A, B;
                A.en = eptr[0];
                B.en = eptr[G];
                set_addr(A);
                A.p = lds_ld(lr_, A.pa);
                A.q = lds_ld(lr_, A.qa);
                int t = 0;
                for (; t + 1 < n; t += 2, eptr += 2 * G) {
                    step(A, B, eptr, 2 * G);
                    step(B, A, eptr, 3 * G);
                }
                if (t < n) step(A, B, eptr, 2 * G);
            }
            if constexpr (TIMED) tm1 = __builtin_amdgcn_s_memtime();
            if (TRAIN && (L == 16 || L == 32 || L == 64) && nr > 0 && (nr & 1) == 0) {
                // hand-scheduled form of the loop below (k in 33..256: 16, 32 or 64 lanes per rating)
                const uint4* eptr = ebase + (size_t)n * G;
                const unsigned ea = (unsigned)(uintptr_t)(lptr_t)eptr;
                const unsigned rowbase = (unsigned)(uintptr_t)(lptr_t)lr_ + lo;
                const unsigned rqa = (__builtin_amdgcn_ubfe(eptr->x, 16, 15) << 4) + lo;
                rq = lds_ld(lr_, rqa);
                if constexpr (L == 16 || L == 32 || L == 64) run_loop_asm<G * 16, L>(rq, ea, rowbase, nr >> 1, lr);
                lds_st(lr_, rqa, rq);
            } else if (nr > 0) {
                const uint4* eptr = ebase + (size_t)n * G;
                StepRegs A, B;
                A.en = eptr[0];
                B.en = eptr[G];
                // every run entry of a slot carries the slot's item address
                const unsigned rqa = (__builtin_amdgcn_ubfe(A.en.x, 16, 15) << 4) + lo;
                A.pa = ((A.en.x & 0xFFFFu) << 4) + lo;
                rq = lds_ld(lr_, rqa);
                A.p = lds_ld(lr_, A.pa);
                int t = 0;
                for (; t + 1 < nr; t += 2, eptr += 2 * G) {
                    run_step(A, B, eptr, 2 * G);
                    run_step(B, A, eptr, 3 * G);
                }
                if (t < nr) run_step(A, B, eptr, 2 * G);
                if constexpr (TRAIN) lds_st(lr_, rqa, rq);
            }
            if (TRAIN && nsolo > 0) {
                // Solo run: header record, then one 16-byte record per step {next slots, mailbox, lr*r, r}.
                const uint4* hdr = lent + (size_t)((offs & 0xFFFF) + n + nr + kSoloPad) * G;
                const unsigned s0 = hdr->x;
                const unsigned rqa = (__builtin_amdgcn_ubfe(s0, 16, 15) << 4) + lo;
                if constexpr (TRAIN && NH > 0 && L >= 16) {
                    // chain wave: dot -> s -> q' only; its helper (a copy wave) stores the p rows and q
                    const uint64_t c2 = ((uint64_t)__builtin_bit_cast(unsigned, c) << 32) | __builtin_bit_cast(unsigned, c);
                    const unsigned rowbase = (unsigned)(uintptr_t)(lptr_t)lr_ + lo;
                    float4 q = lds_ld(lr_, rqa);
                    solo_chain_asm<L>(q, (unsigned)(uintptr_t)(lptr_t)hdr, rowbase, nsolo, lr, c2);
                    // the run was the cell's last work (its records end where the cell's steps end): hand the row on now
                    const int units = (nsolo + 2 + G - 1) / G + kSoloPad;
                    if (post_at != nullptr && (offs & 0xFFFF) + n + nr + units + 2 == n_steps) {
                        if (lane < L) {  // lane group 0: lane l holds elements 4l .. 4l + 3 of the row
                            using gu64 = __attribute__((address_space(1))) unsigned long long;
                            gu64* dst = (gu64*)post_at + lig * 4;
                            const unsigned long long tag = (unsigned long long)post_tag << 32;
                            __hip_atomic_store(dst + 0, tag | __builtin_bit_cast(unsigned, q.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(dst + 1, tag | __builtin_bit_cast(unsigned, q.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(dst + 2, tag | __builtin_bit_cast(unsigned, q.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(dst + 3, tag | __builtin_bit_cast(unsigned, q.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (lane == 0) *posted_flag = 1u;
                    }
                } else if constexpr (TRAIN) {
                    // one wave does everything (kernels without copy waves); every lane group computes the
                    // same step -- the chain is sequential -- and they all store the same bits
                    float4 q = lds_ld(lr_, rqa);
                    unsigned pa = ((s0 & 0xFFFFu) << 4) + lo;
                    for (int t = 0; t < nsolo; ++t) {
                        const uint4 e = hdr[1 + t];
                        const float4 p = lds_ld(lr_, pa);
                        const float dot = group_allreduce<L>(chunk_dot(p, q));
                        const float sc = __builtin_fmaf(-lr, dot, __builtin_bit_cast(float, e.z));
                        const float4 p2 = axpy_row(sc, q, c, p);
                        q = axpy_row(sc, p, c, q);
                        lds_st(lr_, pa, p2);
                        pa = ((e.x & 0xFFFFu) << 4) + lo;
                    }
                    lds_st(lr_, rqa, q);
                }
                // (RMSE: the solo records of ALL sub-cells are shared out over all waves below)
            }
            if constexpr (TIMED) {
                tm2 = __builtin_amdgcn_s_memtime();
                if (lane == 0) {  // [wave][sub-round] -> {general cycles, run cycles, general steps, run steps}
                    unsigned long long* o = timers + ((size_t)wave * W + s) * 4;
                    o[0] = tm1 - tm0;
                    o[1] = tm2 - tm1;
                    o[2] = (unsigned long long)n;
                    o[3] = (unsigned long long)nr;
                }
            }
            if constexpr (TRAIN) wg_barrier();
        }
        if constexpr (!TRAIN) {
            // RMSE over the solo records: nothing is written, so the records of EVERY sub-cell are dealt out over
            // all waves of the workgroup and, inside a wave, lane group g takes record t0 + g (the address of
            // step t sits in record t - 1; groups past the end read the terminator: the zero row with r = 0).
            // (A cell that is one solo run -- an item with a tile of its own -- would otherwise be one wave's job.)
            for (int sc = 0; sc < W * W; ++sc) {
                const uint2 sd = lsub[sc];
                const int offs = __builtin_amdgcn_readfirstlane((int)sd.x);
                const int nsolo = (int)((unsigned)offs >> 16);
                if (nsolo == 0) continue;
                const int nall = __builtin_amdgcn_readfirstlane((int)sd.y);
                const uint4* hdr = lent + (size_t)((offs & 0xFFFF) + (nall & 0xFFFF) + (int)((unsigned)nall >> 16) + kSoloPad) * G;
                const float4 q = lds_ld(lr_, (__builtin_amdgcn_ubfe(hdr->x, 16, 15) << 4) + lo);
                for (int t0 = wave_all * G; t0 < nsolo; t0 += NWV * G) {
                    const int t = t0 + g;
                    const bool live = t < nsolo;
                    const uint4 e = hdr[1 + (live ? t : nsolo)];  // past the end: the terminator (r = 0)
                    const unsigned sl = hdr[live ? t : nsolo].x;    // ... whose predecessor addresses the zero row
                    const float4 p = lds_ld(lr_, ((sl & 0xFFFFu) << 4) + lo);
                    const float err = __builtin_bit_cast(float, e.w) - group_allreduce<L>(chunk_dot(p, q));
                    acc += (double)err * (double)err;
                }
            }
        }
    }
};

// One workgroup = one cell, one launch = one round.  TRAIN: round `rd` runs cells
// (b, (b + rd) % B), each workgroup walking the chunks of its cell.  !TRAIN: blockIdx.x
// is a chunk descriptor index (every chunk on its own), no writes, SSE out.
template <int L, int W, bool TRAIN, bool DIAG = false>
__global__ void __launch_bounds__(64 * W)
cell_kernel(float* __restrict__ P, float* __restrict__ Q, const CellDesc* __restrict__ cells,
            const uint32_t* __restrict__ rows, const SubDesc* __restrict__ subs,
            const Entry* __restrict__ entries, const int B, const int rd, const float lr,
            const float c, double* __restrict__ sse_partial, const int sched_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Cell<L, W> cx;
    cx.init_thread();
    int cell = TRAIN ? (int)blockIdx.x * B + ((int)blockIdx.x + rd) % B : (int)blockIdx.x;
    unsigned long long stamp0 = 0, stamp1 = 0, stamp2 = 0, real0 = 0;
    if constexpr (DIAG) {
        stamp0 = __builtin_amdgcn_s_memtime();
        real0 = __builtin_amdgcn_s_memrealtime();
    }
    double acc = 0.0;
    for (;;) {
        const CellDesc cd = load_desc(cells, (unsigned)cell);
        cx.bind(cd, smem, 0, sched_cap);
        if (cx.nrows == 0) break;  // uniform over the workgroup; an empty cell has no further chunk
        cx.stage_schedule(cd, cell, rows, subs, entries);
        __syncthreads();
        cx.gather(P, Q, 0, cx.nrows);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (DIAG) stamp1 = __builtin_amdgcn_s_memtime();
        if constexpr (DIAG)
            cx.template apply<TRAIN, true>(lr, c, acc, reinterpret_cast<unsigned long long*>(sse_partial) +
                                                           (size_t)gridDim.x * 6 + (size_t)blockIdx.x * W * W * 4);
        else
            cx.template apply<TRAIN>(lr, c, acc);
        if constexpr (DIAG) stamp2 = __builtin_amdgcn_s_memtime();
        if constexpr (!TRAIN) break;
        cx.template scatter<false>(P, Q, 0, cx.nrows);
        if (cd.next == 0) break;
        // next chunk of this cell: its gathers may read rows stored just now, and it reuses the LDS image
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cell = (int)cd.next;
    }
    if constexpr (TRAIN) {
        if constexpr (DIAG) {
            // diagnostic build only: phase stamps of this workgroup (of the last chunk it ran)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long stamp3 = __builtin_amdgcn_s_memtime();
            const unsigned long long real3 = __builtin_amdgcn_s_memrealtime();
            if (cx.tid == 0) {
                unsigned long long* o = reinterpret_cast<unsigned long long*>(sse_partial) + (size_t)blockIdx.x * 6;
                o[0] = stamp0;
                o[1] = stamp1;
                o[2] = stamp2;
                o[3] = stamp3;
                o[4] = real0;  // 100 MHz constant clock, common to all XCDs
                o[5] = real3;
            }
        }
    } else {
        // ---- deterministic sum of squared errors --------------------------------
        // every lane of a group carries the group's sum: keep one copy, then a
        // fixed butterfly over the wave, then waves in index order.
        double v = cx.lig == 0 ? acc : 0.0;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
        __syncthreads();  // everyone is done reading the schedule buffer before it is reused
        double* wsum = reinterpret_cast<double*>(smem + 16);
        if (cx.lane == 0) wsum[cx.wave] = v;
        __syncthreads();
        if (cx.tid == 0) {
            double t = 0.0;
            for (int w = 0; w < W; ++w) t += wsum[w];
            sse_partial[blockIdx.x] = t;
        }
    }
}

// ---- persistent epoch kernel -----------------------------------------------------
// One launch = n_rounds consecutive rounds (an epoch is B rounds).  Workgroup x owns
// user blocks x, x + NP, ... for the whole launch: their P rows are only ever touched
// by this workgroup (this CU), so they need no inter-workgroup protocol.  Item tiles
// move: tile (b + rd) % B is trained by block b in round rd and by block b - 1 in
// round rd + 1, i.e. block b waits for block b + 1 -- a ring hand-off between
// workgroups inside the GPU, the same shape as the DSGD ring between GPUs.
//   producer: q rows stored write-through (sc1) -> every wave s_waitcnt vmcnt(0) ->
//             workgroup barrier -> one lane stores done[b] = R + 1 (relaxed, agent scope);
//   consumer: one lane polls done[b + 1] >= R (relaxed, agent scope, s_sleep) -> workgroup
//             barrier -> the tile's rows are gathered with sc1 loads (they bypass this CU's L1,
//             which is all an acquire fence in front of plain loads would have done).
// (cdna_hip_programming.md Guideline 16, form R1 with sc1 loads in place of the acquire; round 1
// had the fence -- buffer_inv sc1 + s_waitcnt vmcnt(0), ~1.5 us per hop.)  A tile that is ONE item
// row does not use the flags at all: it travels through its mailbox (below).  While it waits, a workgroup has
// already staged the next cell's schedule and gathered its own P rows.  All NP
// workgroups must be co-resident (the host sizes NP from the occupancy query); every
// spin is bounded and raises *abort_word instead of hanging.
constexpr int kFlagStride = 32;  // one done[] word per 128-byte line

// One ring: workgroup `wg` of `NP` runs its share of the B x n_rounds cells of one schedule.
template <int L, int W, int NH>
__device__ __forceinline__ void run_ring(unsigned char* smem, float* __restrict__ P, float* __restrict__ Q,
                                         const CellDesc* __restrict__ cells, const uint32_t* __restrict__ rows,
                                         const SubDesc* __restrict__ subs, const Entry* __restrict__ entries,
                                         const int B, const int n_rounds, const float lr, const float c,
                                         unsigned* __restrict__ done, unsigned* __restrict__ abort_word,
                                         const int sched_cap, const int wg, const int NP,
                                         unsigned long long* __restrict__ prof) {
    using gu32 = __attribute__((address_space(1))) unsigned;
    volatile unsigned* const ctl = reinterpret_cast<volatile unsigned*>(smem);  // [0] = abort broadcast
    Cell<L, W, NH> cx;
    cx.init_thread();
    cx.fail_flag = ctl;
    cx.posted_flag = ctl + 3;
    // Start-of-launch rendezvous, before anything is touched.  It does two jobs with one device-side
    // barrier (sense reversing: abort_word[-4] counts arrivals, abort_word[-3] is the generation):
    //  * the hand-off flags are reset HERE, by their owners (block b's flag by the workgroup that runs
    //    block b), and nobody proceeds until everybody has -- the host zeroes nothing between launches
    //    (a memset node in a replayed hipGraph was measured NOT to be reliably ordered before the kernel
    //    node behind it: flags still standing from the previous epoch let consumers run ahead of their
    //    producers);
    //  * it proves that all NP workgroups are on the chip at once, which the hand-off protocol needs.
    //    When they are not -- another kernel holds CUs -- the launch gives up with the factors untouched
    //    (abort code 2) and the host runs the epoch as one launch per round instead.  A launch that finds
    //    the abort word already set (an earlier launch of the same stream gave up) does nothing either.
    if (cx.tid == 0) {
        unsigned bad = __hip_atomic_load((gu32*)abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned my_gen = 0;
        if (bad == 0u) {
            gu32* arrive = (gu32*)(abort_word - 4);
            gu32* gen = (gu32*)(abort_word - 3);
            for (int b = wg; b < B; b += NP)
                __hip_atomic_store((gu32*)(done + (size_t)b * kFlagStride), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned g0 = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // before arriving
            my_gen = (g0 + 1u) & 0xFFFFu;  // the same in every workgroup of this launch
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // my flags are zero before my arrival counts
            // Arrival counter: low 31 bits count arrivals, bit 31 says "a waiter has given up".  Giving up and
            // releasing are decided on this ONE word, so they cannot both happen: a waiter that times out sets the bit
            // and leaves; the last arriver finds it set and refuses to release (it raises the abort code instead).
            // (Round 2 had the waiter set the abort word and leave without looking back: the last workgroup could
            // arrive in that window, release the others and let them train with one workgroup missing.)
            constexpr unsigned kGaveUp = 0x80000000u;
            const unsigned old = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((old & ~kGaveUp) == (unsigned)NP - 1u) {
                // the last one in: NP -> 0 releases, and only if nobody has set the bit -- one compare-and-swap, so a
                // waiter's give-up (NP -> NP | bit) and the release exclude each other whichever comes first
                unsigned expected = (unsigned)NP;
                if ((old & kGaveUp) == 0u &&
                    __hip_atomic_compare_exchange_strong(arrive, &expected, 0u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __hip_atomic_store(gen, g0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    expected = 0u;
                    __hip_atomic_compare_exchange_strong((gu32*)abort_word, &expected, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                    bad = 2u;  // somebody left: nobody trains (the host zeroes the counter when it handles the abort)
                }
            } else {
                unsigned spins = 0;
                while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g0) {
                    __builtin_amdgcn_s_sleep(8);
                    if ((++spins & 63u) == 0u) {
                        bad = __hip_atomic_load((gu32*)abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (bad == 0u && spins > (1u << 19)) {
                            // give up -- unless the barrier completed meanwhile: the release zeroes the counter, so a
                            // release that has happened shows as a count of 0 here; then take the bit back and wait
                            // for the generation (which the last arriver advances next)
                            const unsigned was = __hip_atomic_fetch_or(arrive, kGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((was & ~kGaveUp) == 0u) {
                                __hip_atomic_fetch_and(arrive, ~kGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g0) __builtin_amdgcn_s_sleep(1);
                                break;
                            }
                            unsigned expected = 0u;  // only the first one to give up sets the code
                            __hip_atomic_compare_exchange_strong((gu32*)abort_word, &expected, 2u, __ATOMIC_RELAXED,
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            bad = 2u;
                        }
                        if (bad != 0u) break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        ctl[0] = bad != 0u ? 1u : 0u;
        ctl[1] = bad;
        ctl[2] = my_gen;
    }
    wg_barrier();
    if (ctl[0] != 0) return;  // uniform; nothing has been modified
    // Tile mailboxes (cells marked kCellLoneTile: a tile that is ONE item row in every cell -- the item whose chain
    // the epoch waits for).  The row travels as KP granules {value, tag}, each written by ONE 8-byte sc1 store and
    // read by ONE 8-byte sc1 load (a granule is never seen torn), tag = launch generation << 16 | round + 1: the
    // consumer polls the granules themselves until every tag is the one it expects -- one memory round trip per
    // hop, no drain, no flag, no gather -- and nothing of an earlier round or launch can be mistaken for it.
    // Only the holder in the launch's LAST round stores the row to Q; the first round takes it from Q.
    using gu64 = __attribute__((address_space(1))) unsigned long long;
    gu64* const mbox = (gu64*)(abort_word + 4);
    const unsigned tag_hi = ctl[2] << 16;
    constexpr int KP = Cell<L, W, NH>::KP, ROWB = Cell<L, W, NH>::ROWB;
    constexpr int NGR = KP >= 64 ? KP / 64 : 1;  // granules per lane of one wave

    // This workgroup's work list: (round R, block b) for b = blockIdx.x, +NP, ... in round order,
    // and within a cell its chunks in chain order.
    struct Item {
        int R, b;
        unsigned idx;  // chunk descriptor
        bool first;    // first chunk of its cell: the tile has to be waited for
    };
    auto cell_of = [&](int R, int b) { return (unsigned)(b * B + (b + R % B) % B); };
    auto next_item = [&](const Item& it, const CellDesc& d) {
        Item n = it;
        if (d.next != 0) {
            n.idx = d.next;
            n.first = false;
        } else {
            n.b += NP;
            if (n.b >= B) {
                n.b = wg;
                ++n.R;
            }
            n.idx = cell_of(n.R, n.b);
            n.first = true;
        }
        return n;
    };
    // Software pipeline over the list: descriptors are fetched two items ahead (registers),
    // schedules one item ahead (LDS-DMA into the other schedule buffer).
    Item it0{0, wg, cell_of(0, wg), true};
    CellDesc cd = load_desc(cells, it0.idx);
    Item it1 = next_item(it0, cd);
    CellDesc cd1 = it1.R < n_rounds ? load_desc(cells, it1.idx) : cd;
    int buf = 0;
    cx.bind(cd, smem, buf, sched_cap);
    cx.stage_schedule(cd, (int)it0.idx, rows, subs, entries);  // the first one synchronously
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();

    // optional phase accounting (diagnostic launches only): shader cycles of wave 0 per phase
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // [7]: the longest single "ratings" phase (its slowest cell)
    unsigned long long pcur[7] = {0, 0, 0, 0, 0, 0, 0}, pmax[7] = {0, 0, 0, 0, 0, 0, 0};  // this pass / the pass of [7]
    unsigned long long pt = 0;
    auto mark = [&](int k) {
        if (prof) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            pacc[k] += now - pt;
            pcur[k] = now - pt;
            if (k == 6 && pcur[4] > pacc[7]) {  // end of a pass whose ratings phase is the longest so far
                pacc[7] = pcur[4];
                for (int x = 0; x < 7; ++x) pmax[x] = pcur[x];
            }
            pt = now;
        }
    };
    if (prof) pt = __builtin_amdgcn_s_memtime();

    for (; it0.R < n_rounds;) {
        const int R = it0.R, b = it0.b;
        CellDesc cd2 = cd1;
        cx.bind(cd, smem, buf, sched_cap);
        const bool work = cx.nrows != 0;  // uniform over the workgroup
        const bool last = cd.next == 0;   // last chunk of its cell: the tile is handed on after it
        cx.zero_idle_rows();
        if (cx.tid == 0) ctl[3] = 0u;  // "the chain wave has posted the tile's row" (read behind the barriers below)
        // The rows stored at the end of the previous iteration may be gathered again below.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MFSGD_DIAG_SPLIT_PHASE0  // (a build for tools/phase_profile.py that takes phase 0 apart)
        mark(0);  // phase 0 = zeroing + drain only
#endif
        if (it1.R < n_rounds) cx.prefetch_schedule(cd1, (int)it1.idx, smem, buf ^ 1, sched_cap, rows, subs, entries);
        if (work) cx.gather(P, Q, 0, cx.nu);  // own rows: no dependency on other workgroups
        // descriptor used two iterations from now: a scalar load issued here, in front of the wait for the tile,
        // so that it completes in that wait's shadow (behind the tile gather it was exposed -- ~1.4 K cycles --
        // whenever there was no gather to hide it: a tile taken from its mailbox)
        const Item it2 = next_item(it1, cd1);
        const bool lone = KP >= 64 && (cd.rsv[0] & kCellLoneTile) != 0;  // uniform
#ifdef MFSGD_DIAG_SPLIT_PHASE0
        mark(2);  // diagnostic build: the issue of prefetch and gather, booked under "barrier"
#endif
        if (it2.R < n_rounds) cd2 = load_desc(cells, it2.idx);
#ifdef MFSGD_DIAG_SPLIT_PHASE0
        mark(6);  // diagnostic build: the descriptor load (s_memtime waits for it), booked under "own store"
#else
        mark(0);  // drain of the previous stores + issue of the prefetch and the P gather
#endif
        if (R > 0 && it0.first && lone) {
            // the tile is one row: take it from the tile's mailbox as soon as block b + 1 has posted it
            if (cx.wave_all == 0) {
                const unsigned tile = (unsigned)((b + R % B) % B);
                const unsigned want = tag_hi | (unsigned)R;  // posted in round R - 1
                const gu64* src = mbox + (size_t)tile * KP + cx.lane;
                unsigned long long v[NGR];
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < NGR; ++j) {
                        v[j] = __hip_atomic_load(src + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = ok && (unsigned)(v[j] >> 32) == want;
                    }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0u) {
                        if (__hip_atomic_load((gu32*)abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                            spins > (1u << 22)) {
                            if (cx.lane == 0) {
                                __hip_atomic_store((gu32*)abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ctl[0] = 1;
                            }
                            break;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < NGR; ++j)
                    *reinterpret_cast<unsigned*>(cx.lrows + (size_t)cx.nu * ROWB + (size_t)(cx.lane + 64 * j) * 4) = (unsigned)v[j];
            }
        } else if (R > 0 && it0.first) {
            // wait until block b + 1 has finished round R - 1 (it held our tile)
            if (cx.tid == 0) {
                gu32* flag = (gu32*)(done + (size_t)((b + 1) % B) * kFlagStride);
                unsigned spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)R) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 255u) == 0u) {
                        if (__hip_atomic_load((gu32*)abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                            spins > (1u << 22)) {
                            __hip_atomic_store((gu32*)abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ctl[0] = 1;
                            break;
                        }
                    }
                }
                asm volatile("" ::: "memory");  // the tile's rows are loaded sc1 below: no acquire fence
            }
        }
        mark(1);  // waiting for the tile (wave 0)
        const bool from_mbox = lone && R > 0;  // the tile's row is in LDS already (wave 0 put it there)
        // (Waiting for the own rows in front of this barrier and dropping the second one for a row that came from its
        // mailbox -- one barrier less on the hop the epoch waits for -- was measured: 4.00 against 3.97 ms per epoch
        // on the same box, three runs each; the second barrier stays.)
        wg_barrier();
        mark(2);  // the other waves' arrival
        if (ctl[0] != 0) {  // uniform: some workgroup timed out (or a solo helper of this one gave up)
            if (cx.tid == 0) __hip_atomic_store((gu32*)abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (work && !from_mbox) cx.template gather<true>(P, Q, cx.nu, cx.nrows);  // the tile's q rows, sc1: stored by another CU
        if (work) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // rows (and the prefetched schedule) have landed
            wg_barrier();
            mark(3);  // tile rows (and own rows, and the next schedule) landed
            double acc = 0.0;
            if (lone && R + 1 < n_rounds) {
                cx.post_at = (unsigned long long*)(abort_word + 4) + (size_t)((b + R % B) % B) * KP;  // the tile's mailbox
                cx.post_tag = tag_hi | (unsigned)(R + 1);
            } else {
                cx.post_at = nullptr;
            }
            cx.template apply<true>(lr, c, acc);  // ends with a workgroup barrier
            mark(4);  // the ratings
            // write-through even when more chunks of this cell follow: item rows that no later
            // chunk touches have to be visible to the next workgroup all the same
            if (lone && R + 1 < n_rounds) {
                // post the row for block b - 1 (round R + 1); Q gets it from the holder in the last round
                // (unless the chain wave has posted it from its registers already, Cell::post_at)
                if (cx.wave_all == 0 && ctl[3] == 0u) {
                    const unsigned tile = (unsigned)((b + R % B) % B);
                    const unsigned long long tag = (unsigned long long)(tag_hi | (unsigned)(R + 1)) << 32;
                    gu64* dst = mbox + (size_t)tile * KP + cx.lane;
#pragma unroll
                    for (int j = 0; j < NGR; ++j) {
                        const unsigned bits =
                            *reinterpret_cast<const unsigned*>(cx.lrows + (size_t)cx.nu * ROWB + (size_t)(cx.lane + 64 * j) * 4);
                        __hip_atomic_store(dst + 64 * j, tag | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else {
                cx.template scatter<true>(P, Q, cx.nu, cx.nrows);
            }
        }
        // publish the tile: every storing wave drains, then one lane signals
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();
        if (last && cx.tid == 0)
            __hip_atomic_store((gu32*)(done + (size_t)b * kFlagStride), (unsigned)(R + 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        mark(5);  // tile rows stored write-through, drained, flag published
        if (work) cx.template scatter<false>(P, Q, 0, cx.nu);
        // The rows image and this schedule buffer are reused from here on: their LDS reads (the
        // scatter above) are complete once every wave has passed this barrier.  The next
        // schedule has been complete since the vmcnt(0) + barrier above.
        wg_barrier();
        buf ^= 1;
        cd = cd1;
        cd1 = cd2;
        it0 = it1;
        it1 = it2;
        mark(6);  // own rows stored (not drained), end barrier
    }
    if (ctl[0] != 0 && cx.tid == 0)  // raised during the last cell
        __hip_atomic_store((gu32*)abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wg == 0 && cx.tid == 0)  // launches that got past the residency check (the host counts on it when one did not)
        __hip_atomic_fetch_add((gu32*)(abort_word + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prof && cx.tid == 0) {
        for (int k = 0; k < 8; ++k) prof[(size_t)wg * 16 + k] = pacc[k];
        for (int k = 0; k < 7; ++k) prof[(size_t)wg * 16 + 8 + k] = pmax[k];
    }
}

template <int L, int W>
__global__ void __launch_bounds__(64 * (W + epoch_helpers<L, W>()))
epoch_kernel(float* __restrict__ P, float* __restrict__ Q, const CellDesc* __restrict__ cells,
             const uint32_t* __restrict__ rows, const SubDesc* __restrict__ subs,
             const Entry* __restrict__ entries, const int B, const int n_rounds, const float lr,
             const float c, unsigned* __restrict__ done, unsigned* __restrict__ abort_word,
             const int sched_cap, unsigned long long* __restrict__ prof) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    run_ring<L, W, epoch_helpers<L, W>()>(smem, P, Q, cells, rows, subs, entries, B, n_rounds, lr, c, done, abort_word,
                                           sched_cap, (int)blockIdx.x, (int)gridDim.x, prof);
}

// Sum of squared errors, persistent form: gridDim.x workgroups walk the B*B cells with a stride,
// the next cell's schedule prefetched (LDS-DMA) while the current one is applied; no writes.
// One fp64 partial per workgroup (fixed order inside it), reduced by reduce_sse_kernel.
template <int L, int W>
__global__ void __launch_bounds__(64 * (W + epoch_helpers<L, W>()))
sse_kernel(const float* __restrict__ P, const float* __restrict__ Q, const CellDesc* __restrict__ cells,
           const uint32_t* __restrict__ rows, const SubDesc* __restrict__ subs, const Entry* __restrict__ entries,
           const int n_cells, double* __restrict__ sse_partial, const int sched_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Cell<L, W, epoch_helpers<L, W>()> cx;
    constexpr int NWV = W + epoch_helpers<L, W>();
    cx.init_thread();
    const int stride = (int)gridDim.x;
    int c = (int)blockIdx.x;
    double acc = 0.0;
    if (c < n_cells) {
        CellDesc cd = load_desc(cells, (unsigned)c);
        CellDesc cd1 = c + stride < n_cells ? load_desc(cells, (unsigned)(c + stride)) : cd;
        int buf = 0;
        cx.bind(cd, smem, buf, sched_cap);
        cx.stage_schedule(cd, c, rows, subs, entries);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();
        for (; c < n_cells; c += stride) {
            const int c1 = c + stride, c2 = c + 2 * stride;
            cx.bind(cd, smem, buf, sched_cap);
            cx.zero_idle_rows();
            if (c1 < n_cells) cx.prefetch_schedule(cd1, c1, smem, buf ^ 1, sched_cap, rows, subs, entries);
            if (cx.nrows != 0) cx.gather(P, Q, 0, cx.nrows);
            CellDesc cd2 = cd1;
            if (c2 < n_cells) cd2 = load_desc(cells, (unsigned)c2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // rows and the next schedule have landed
            wg_barrier();
            if (cx.nrows != 0) cx.template apply<false>(0.f, 0.f, acc);
            wg_barrier();  // every wave is done with the rows image and this schedule buffer
            buf ^= 1;
            cd = cd1;
            cd1 = cd2;
        }
    }
    // every lane of a group carries the group's sum: keep one copy, a fixed butterfly over the
    // wave, then waves in index order
    double v = cx.lig == 0 ? acc : 0.0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    double* wsum = reinterpret_cast<double*>(smem + 16);
    __syncthreads();
    if (cx.lane == 0) wsum[cx.wave_all] = v;
    __syncthreads();
    if (cx.tid == 0) {
        double t = 0.0;
        for (int w = 0; w < NWV; ++w) t += wsum[w];
        sse_partial[blockIdx.x] = t;
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
                           a.cells, a.rows, a.subs, a.entries, a.B, a.rd, a.lr, a.c, a.sse_partial, a.sched_cap);
    } else if (train)
        hipLaunchKernelGGL((cell_kernel<L, W, true>), grid, block, (size_t)a.lds_bytes, st, a.P, a.Q,
                           a.cells, a.rows, a.subs, a.entries, a.B, a.rd, a.lr, a.c, a.sse_partial, a.sched_cap);
    else
        hipLaunchKernelGGL((cell_kernel<L, W, false>), grid, block, (size_t)a.lds_bytes, st, a.P, a.Q,
                           a.cells, a.rows, a.subs, a.entries, a.B, a.rd, a.lr, a.c, a.sse_partial, a.sched_cap);
    return hipGetLastError();
}

template <int L, int W>
hipError_t epoch_LW(int what, const CellLaunch& a, int n_rounds, unsigned* done, unsigned* abort_word, int* out,
                    hipStream_t st) {
    const void* fn = (const void*)epoch_kernel<L, W>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, a.lds_bytes);
    if (e != hipSuccess) return e;
    constexpr int threads = 64 * (W + epoch_helpers<L, W>());
    if (what == 0) {  // occupancy query: workgroups per CU
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, fn, threads, (size_t)a.lds_bytes);
    }
    hipLaunchKernelGGL((epoch_kernel<L, W>), dim3((unsigned)a.grid), dim3(threads), (size_t)a.lds_bytes, st, a.P, a.Q,
                       a.cells, a.rows, a.subs, a.entries, a.B, n_rounds, a.lr, a.c, done, abort_word, a.sched_cap,
                       reinterpret_cast<unsigned long long*>(a.diag ? a.sse_partial : nullptr));
    return hipGetLastError();
}

template <int L>
hipError_t epoch_L(int what, int W, const CellLaunch& a, int n_rounds, unsigned* done, unsigned* abort_word, int* out,
                   hipStream_t st) {
    switch (W) {
        case 1: return epoch_LW<L, 1>(what, a, n_rounds, done, abort_word, out, st);
        case 2: return epoch_LW<L, 2>(what, a, n_rounds, done, abort_word, out, st);
        case 4: return epoch_LW<L, 4>(what, a, n_rounds, done, abort_word, out, st);
        case 8: return epoch_LW<L, 8>(what, a, n_rounds, done, abort_word, out, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t epoch_dispatch(int what, int L, int W, const CellLaunch& a, int n_rounds, unsigned* done,
                          unsigned* abort_word, int* out, hipStream_t st) {
    switch (L) {
        case 1: return epoch_L<1>(what, W, a, n_rounds, done, abort_word, out, st);
        case 2: return epoch_L<2>(what, W, a, n_rounds, done, abort_word, out, st);
        case 4: return epoch_L<4>(what, W, a, n_rounds, done, abort_word, out, st);
        case 8: return epoch_L<8>(what, W, a, n_rounds, done, abort_word, out, st);
        case 16: return epoch_L<16>(what, W, a, n_rounds, done, abort_word, out, st);
        case 32: return epoch_L<32>(what, W, a, n_rounds, done, abort_word, out, st);
        case 64: return epoch_L<64>(what, W, a, n_rounds, done, abort_word, out, st);
        default: return hipErrorInvalidValue;
    }
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

template <int L, int W>
hipError_t sse_LW(const CellLaunch& a, int n_cells, hipStream_t st) {
    const void* fn = (const void*)sse_kernel<L, W>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, a.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sse_kernel<L, W>), dim3((unsigned)a.grid), dim3(64 * (W + epoch_helpers<L, W>())), (size_t)a.lds_bytes, st, a.P, a.Q, a.cells,
                       a.rows, a.subs, a.entries, n_cells, a.sse_partial, a.sched_cap);
    return hipGetLastError();
}

template <int L>
hipError_t sse_L(int W, const CellLaunch& a, int n_cells, hipStream_t st) {
    switch (W) {
        case 1: return sse_LW<L, 1>(a, n_cells, st);
        case 2: return sse_LW<L, 2>(a, n_cells, st);
        case 4: return sse_LW<L, 4>(a, n_cells, st);
        case 8: return sse_LW<L, 8>(a, n_cells, st);
        default: return hipErrorInvalidValue;
    }
}

// Factor initialisation on the device: java.util.Random(seed).nextFloat() * scale, row-major, the stream
// position of row x being first_pos + x * k -- the 48-bit LCG jumped to that position per row (the same
// doubling recurrence as JRandom::skip), then k sequential draws.  Integer work plus one exact division and
// one multiply per element: the bits csrc/jrandom.hpp and the oracle produce.
__global__ void __launch_bounds__(256) init_rows_kernel(float* __restrict__ dst, const long long rows, const int k,
                                                        const int kp, const unsigned long long s0,
                                                        const unsigned long long first_pos, const float scale) {
    constexpr unsigned long long kMult = 0x5DEECE66DULL, kAdd = 0xBULL, kMask = (1ULL << 48) - 1;
    for (long long row = (long long)blockIdx.x * 256 + threadIdx.x; row < rows; row += (long long)gridDim.x * 256) {
        unsigned long long n = first_pos + (unsigned long long)row * (unsigned long long)k;
        unsigned long long acc_a = 1, acc_c = 0, cur_a = kMult, cur_c = kAdd;
        while (n) {
            if (n & 1) {
                acc_a = (acc_a * cur_a) & kMask;
                acc_c = (acc_c * cur_a + cur_c) & kMask;
            }
            cur_c = ((cur_a + 1) * cur_c) & kMask;
            cur_a = (cur_a * cur_a) & kMask;
            n >>= 1;
        }
        unsigned long long s = (acc_a * s0 + acc_c) & kMask;
        float* out = dst + (size_t)row * kp;
        for (int f = 0; f < k; ++f) {
            s = (s * kMult + kAdd) & kMask;
            const int v = (int)(unsigned)(s >> 24);  // next(24)
            out[f] = (float)v / 16777216.0f * scale;
        }
        for (int f = k; f < kp; ++f) out[f] = 0.0f;
    }
}

hipError_t launch_init_rows(float* dst, long long rows, int k, int kp, long long seed, unsigned long long first_pos, float scale,
                            hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    const unsigned long long s0 = ((unsigned long long)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
    long long g = (rows + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(init_rows_kernel, dim3((unsigned)g), dim3(256), 0, st, dst, rows, k, kp, s0, first_pos, scale);
    return hipGetLastError();
}

// Diagnostic: workgroups that hold a whole CU's LDS and spin for `ticks` of the 100 MHz clock.
__global__ void __launch_bounds__(64) occupy_kernel(const unsigned long long ticks, unsigned* __restrict__ started) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    smem[threadIdx.x] = 0;
    if (started && threadIdx.x == 0) __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

hipError_t launch_occupy(int workgroups, int lds_bytes, unsigned long long ticks, unsigned* started, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)workgroups), dim3(64), (size_t)lds_bytes, st, ticks, started);
    return hipGetLastError();
}

hipError_t epoch_blocks_per_cu(int L, int W, const CellLaunch& a, int* blocks_per_cu) {
    return epoch_dispatch(0, L, W, a, 0, nullptr, nullptr, blocks_per_cu, nullptr);
}

hipError_t launch_epoch_persistent(int L, int W, const CellLaunch& a, int n_rounds, unsigned* done,
                                   unsigned* abort_word, hipStream_t st) {
    return epoch_dispatch(1, L, W, a, n_rounds, done, abort_word, nullptr, st);
}

hipError_t launch_sse_persistent(int L, int W, const CellLaunch& a, int n_cells, hipStream_t st) {
    switch (L) {
        case 1: return sse_L<1>(W, a, n_cells, st);
        case 2: return sse_L<2>(W, a, n_cells, st);
        case 4: return sse_L<4>(W, a, n_cells, st);
        case 8: return sse_L<8>(W, a, n_cells, st);
        case 16: return sse_L<16>(W, a, n_cells, st);
        case 32: return sse_L<32>(W, a, n_cells, st);
        case 64: return sse_L<64>(W, a, n_cells, st);
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
