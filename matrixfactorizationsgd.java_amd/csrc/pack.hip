// pack.hip -- the per-cell step packer on the device (second phase).
//
// No reference counterpart exists (/root/reference/README.md:1-2).  This is schedule.cpp's
// pack_chunk -- distinct rows of a cell, run / solo detection per sub-cell, the greedy conflict-free
// packing of the general ratings (pack_subcell), the run packing (pack_run), the solo records
// (pack_solo) -- restated for one workgroup per cell, and it produces the SAME BYTES: the same
// selection rule evaluated in the same order with the same tie-breaks, so the schedule (and with
// it the canonical order the oracle replays) does not depend on where it was built
// (tests/test_gpu_parity.py compares device-built and host-built schedules word for word).
//
// Integer / byte work: no floating point except lr * r (one multiply per rating, as on the host).
// One workgroup = one cell; its W*W sub-cells are independent given the cell's row slots, and
// are dealt to the workgroup's waves; inside a sub-cell the greedy is sequential over steps, and a
// wave runs it with its 64 lanes scanning the candidates (wave-wide arg-max per slot taken).
//
// Two passes over the cells with the same code: COUNT (sizes per cell and sub-cell; the host
// turns them into offsets and checks that every cell fits the LDS image as a single chunk), then
// EMIT (rows, entries, order written at their final places).  Anything this kernel cannot hold
// (a cell with more ratings or rows than its LDS arrays, 16-bit counters that would overflow)
// is reported, and the caller falls back to the host packer -- which also does the chunking of
// cells too large for the training kernel's LDS image.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ingest.hpp"
#include "pack.hpp"

namespace mfsgd {

namespace {

constexpr int kWaves = 4;  // waves per workgroup
// one-pass mode: a sub-cell of n ratings packs into at most 2 n + kPackSlack steps -- a general or run step can come out
// empty when every rating left is blocked by the step before it, but never two in a row; the slack is a run's padding
// step, the idle steps in front of a solo run, its header and terminator records
constexpr int kPackSlack = 3 + kSoloPad;

struct WaveState {
    int* remdeg;         // per row slot: ratings left on the row in the current list (also the degree counter)
    short* last;         // stamp of the step that selected the row
    short* prev;         // stamp of the last emitted step that used the row
    signed char* slot;   // lane slot the row had in that step
    unsigned short* l0;  // the sub-cell's ratings partitioned: general | run | solo (positions in the cell arrays)
    unsigned short* tk;  // scratch: takes of a step (position, slot) / run queues
};

__device__ __forceinline__ unsigned enc_slots(int pslot, int qslot, bool flag, int L) {
    return (unsigned)(pslot * L) | ((unsigned)(qslot * L) << 16) | (flag ? 0x80000000u : 0u);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned long long o = __shfl_xor(v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}

[[maybe_unused]] __device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// exclusive prefix of a 0/1 flag over the lanes of a wave
__device__ __forceinline__ int lane_rank(bool flag, int lane, int& total) {
    const unsigned long long b = __ballot(flag);
    total = __popcll(b);
    return __popcll(b & ((1ull << lane) - 1ull));
}

}  // namespace

// One workgroup per cell.  Dynamic LDS, in this order:
//   [rat_p u16 x M][rat_q u16 x M][rat_r f32 x M][rat_i u32 x M]                (the cell's ratings)
//   [ubits u32 x UW][ibits u32 x IW][upre u16 x UW][ipre u16 x IW]              (touched-row bitmaps, prefixes)
//   per wave: [remdeg i32 x R][last i16 x R][prev i16 x R][slot i8 x R (padded)][l0 u16 x M][tk u16 x 2M]
//   [sub_info i32 x WW x 4][misc i32 x 8]
__global__ void __launch_bounds__(64 * kWaves) pack_kernel(const PackArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, W = a.W, WW = W * W, G = a.G, L = a.L, M = a.max_m, R = a.max_rows;
    const long long cell = blockIdx.x;
    const int ub = (int)(cell / B), it = (int)(cell % B);
    (void)ub;
    (void)it;

    unsigned short* rat_p = reinterpret_cast<unsigned short*>(smem);
    unsigned short* rat_q = rat_p + M;
    float* rat_r = reinterpret_cast<float*>(rat_q + M);
    unsigned* rat_i = reinterpret_cast<unsigned*>(rat_r + M);
    unsigned* ubits = rat_i + M;
    unsigned* ibits = ubits + a.u_words;
    unsigned short* upre = reinterpret_cast<unsigned short*>(ibits + a.i_words);
    unsigned short* ipre = upre + a.u_words;
    unsigned char* wbase = reinterpret_cast<unsigned char*>(ipre + a.i_words + ((a.u_words + a.i_words) & 1));
    const size_t slot_bytes = ((size_t)R + 3) & ~(size_t)3;
    const size_t wave_bytes = (size_t)R * 4 + (size_t)R * 2 * 2 + slot_bytes + (size_t)M * 2 + (size_t)M * 4;
    WaveState ws;
    {
        unsigned char* p = wbase + (size_t)wave * wave_bytes;
        ws.remdeg = reinterpret_cast<int*>(p);
        ws.last = reinterpret_cast<short*>(ws.remdeg + R);
        ws.prev = ws.last + R;
        ws.slot = reinterpret_cast<signed char*>(ws.prev + R);
        ws.l0 = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(ws.slot) + slot_bytes);
        ws.tk = ws.l0 + M;
    }
    int* sub_info = reinterpret_cast<int*>(wbase + (size_t)kWaves * wave_bytes);  // [x][0..3] = ns, nr, nsolo, units
    int* misc = sub_info + WW * 4;                                                  // [0] status, [1] nu, [2] ni

    if (a.emit == 1 && a.row_off[cell] == 0xFFFFFFFFu) return;  // mixed build: this cell is cut (its chunks come as parts)
    const long long lo = a.bptr[cell * WW], hi = a.bptr[(cell + 1) * WW];
    const int m = (int)(hi - lo);
    PackCellInfo info{};
    if (tid == 0) misc[0] = 0;
    if (m == 0) {
        if (a.emit != 1 && tid == 0) {
            a.info[cell] = info;
        }
        if (a.emit != 1)
            for (int x = tid; x < WW; x += 64 * kWaves) a.subs[cell * WW + x] = SubDesc{0u, 0u};
        return;
    }
    if (m > M) {
        if (a.emit != 1 && tid == 0) {
            info.status = 1;
            a.info[cell] = info;
        }
        return;
    }

    // ---- the cell's ratings; touched-row bitmaps ----------------------------------------------
    for (int x = tid; x < a.u_words + a.i_words; x += 64 * kWaves) ubits[x] = 0u;  // ubits and ibits are adjacent
    __syncthreads();
    for (int x = tid; x < m; x += 64 * kWaves) {
        const unsigned j = a.sorted[lo + x];
        const int uu = a.u[j], ii = a.i[j];
        rat_r[x] = a.r[j];
        rat_i[x] = j;
        const int ur = a.urank[uu], ir = a.irank[ii];
        rat_p[x] = (unsigned short)ur;  // ranks for now; turned into slots below
        rat_q[x] = (unsigned short)ir;
        atomicOr(&ubits[ur >> 5], 1u << (ur & 31));
        atomicOr(&ibits[ir >> 5], 1u << (ir & 31));
    }
    __syncthreads();
    // exclusive prefix of the popcounts (wave 0: users, wave 1: items), 64 words at a time
    if (wave < 2) {
        const unsigned* bits = wave == 0 ? ubits : ibits;
        unsigned short* pre = wave == 0 ? upre : ipre;
        const int nw = wave == 0 ? a.u_words : a.i_words;
        int run = 0;
        for (int w0 = 0; w0 < nw; w0 += 64) {
            const int x = w0 + lane;
            int v = x < nw ? __popc(bits[x]) : 0;
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                if (lane >= d) incl += o;
            }
            if (x < nw) pre[x] = (unsigned short)(run + incl - v);
            run += __shfl(incl, 63, 64);
        }
        if (lane == 0) misc[1 + wave] = run;
    }
    __syncthreads();
    const int nu = misc[1], ni = misc[2], nrows = nu + ni;
    // where this pass writes: the EMIT pass (emit == 1) at the offsets the host derived from a COUNT pass; the one-pass
    // mode (emit == 2, [r3]) into SCRATCH arrays at worst-case offsets known from the bucket starts alone -- at most two
    // rows per rating, and a sub-cell of n ratings packs into at most 2 n + kPackSlack steps -- which compact_kernel moves
    // to their final places once the host has the sizes: the packing itself runs once
    const size_t row_base = a.emit == 1 ? (size_t)a.row_off[cell] : 2 * (size_t)lo;
    bool bad = nrows > R || (long long)(nrows + 2 * G) * L > 32767 || nu > 0xFFFF || ni > 0xFFFF;
    if (bad) {
        if (a.emit != 1 && tid == 0) {
            info.status = nrows > R ? 2 : 1;  // 2: more rows than this launch's LDS arrays hold (a retry with more may do)
            info.nu = (unsigned)nu;
            info.ni = (unsigned)ni;
            a.info[cell] = info;
        }
        return;
    }
    for (int x = tid; x < m; x += 64 * kWaves) {
        const int ur = rat_p[x], ir = rat_q[x];
        const int ps = upre[ur >> 5] + __popc(ubits[ur >> 5] & ((1u << (ur & 31)) - 1u));
        const int qs = nu + ipre[ir >> 5] + __popc(ibits[ir >> 5] & ((1u << (ir & 31)) - 1u));
        rat_p[x] = (unsigned short)ps;
        rat_q[x] = (unsigned short)qs;
        if (a.emit) {  // row ids, users then items, ascending id = ascending rank (every occurrence writes the same value)
            const unsigned j = rat_i[x];
            a.rows[row_base + (size_t)ps] = (unsigned)a.u[j];
            a.rows[row_base + (size_t)qs] = (unsigned)a.i[j];
        }
    }
    __syncthreads();

    // ---- sub-cells, dealt to the waves -----------------------------------------------------------
    const float lr = a.lr, cdecay = a.c;
    const size_t ent_base = a.emit == 1 ? (size_t)a.ent_off[cell]
                            : 2 * (size_t)lo + (size_t)cell * (size_t)(WW * kPackSlack + 2);  // first step of the cell
    const long long ord_base = a.emit ? a.ord_off[cell] : 0;
    auto put_entry = [&](unsigned step, int g, unsigned slots, float r, float ce) {
        Entry e;
        e.slots = slots;
        e.r = r;
        e.lrr = lr * r;
        e.ce = ce;
        a.entries[(ent_base + step) * G + g] = e;
    };
    auto idle_general = [&](unsigned step) {  // an all-idle general step (every lane helps)
        for (int g = lane; g < G; g += 64) put_entry(step, g, enc_slots(nrows + 2 * g, nrows + 2 * g + 1, false, L), 0.0f, cdecay);
    };

    for (int x = wave; x < WW; x += kWaves) {
        const int slo = (int)(a.bptr[cell * WW + x] - lo), shi = (int)(a.bptr[cell * WW + x + 1] - lo);
        const int nsub = shi - slo;
        int ns = 0, nr = 0, nsolo = 0, units = 0;
        // offsets known from the COUNT pass when emitting
        unsigned stepcur = 0;
        long long ord_at = 0;
        if (a.emit) {
            stepcur = a.emit == 1 ? (a.subs[cell * WW + x].off & 0xFFFFu) : (unsigned)(2 * slo + x * kPackSlack);
            ord_at = ord_base + slo;
        }
        if (nsub > 0) {
            // state reset
            for (int rr = lane; rr < nrows; rr += 64) {
                ws.remdeg[rr] = 0;
                ws.last[rr] = 0;
                ws.prev[rr] = 0;
                ws.slot[rr] = (signed char)-1;
            }
            __builtin_amdgcn_wave_barrier();
            // ---- run items: d >= kRunMin and d * G >= nsub, heaviest first (ties: smaller slot) ------
            int nrun = 0;
            int run_q[64], run_d[64];
            int top_n = 0;
            if (nsub >= kRunMin) {
                for (int j = slo + lane; j < shi; j += 64) atomicAdd(&ws.remdeg[rat_q[j]], 1);
                __builtin_amdgcn_wave_barrier();
                // at most G items can satisfy d * G >= nsub: collect them by scanning the item slots
                for (int q0 = nu; q0 < nrows; q0 += 64) {
                    const int q = q0 + lane;
                    const int d = q < nrows ? ws.remdeg[q] : 0;
                    unsigned long long hit = __ballot(d >= kRunMin && (long long)d * G >= nsub);
                    while (hit) {
                        const int src = __builtin_ctzll(hit);
                        hit &= hit - 1;
                        const int dq = __shfl(d, src, 64);
                        // insertion by (d desc, q asc); q ascends in scan order, so equal d keeps its place
                        int pos = top_n;
                        while (pos > 0 && run_d[pos - 1] < dq) {
                            if (pos < 64) {
                                run_d[pos] = run_d[pos - 1];
                                run_q[pos] = run_q[pos - 1];
                            }
                            --pos;
                        }
                        if (pos < 64) {
                            run_d[pos] = dq;
                            run_q[pos] = q0 + src;
                        }
                        if (top_n < 64) ++top_n;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                for (int j = slo + lane; j < shi; j += 64) ws.remdeg[rat_q[j]] = 0;
                __builtin_amdgcn_wave_barrier();
                nrun = top_n < G ? top_n : G;
            }
            // ---- solo: the heaviest run item when it dwarfs the others and its users are distinct ----
            bool solo = false;
            int solo_q = -1;
            if (a.solo_ok && nrun > 0) {
                const int n1 = run_d[0], n2 = top_n > 1 ? run_d[1] : 0;
                if (n1 >= kSoloMin && 4 * n2 <= n1) {
                    solo = true;
                    solo_q = run_q[0];
                    bool dup = false;
                    for (int j0 = slo; j0 < shi; j0 += 64) {
                        const int j = j0 + lane;
                        if (j < shi && rat_q[j] == solo_q) {
                            // users are distinct iff nobody finds the mark already set
                            const int old = atomicExch(&ws.remdeg[rat_p[j]], 1);
                            if (old != 0) dup = true;
                        }
                    }
                    dup = __any(dup);
                    __builtin_amdgcn_wave_barrier();
                    for (int j = slo + lane; j < shi; j += 64)
                        if (rat_q[j] == solo_q) ws.remdeg[rat_p[j]] = 0;
                    __builtin_amdgcn_wave_barrier();
                    if (dup) solo = false;
                }
                if (solo) {
                    for (int g = 1; g < nrun; ++g) {
                        run_q[g - 1] = run_q[g];
                        run_d[g - 1] = run_d[g];
                    }
                    --nrun;
                }
            }
            // ---- stable partition of the sub-cell: general | run | solo ------------------------------
            auto kind = [&](int j) {
                const int q = rat_q[j];
                if (solo && q == solo_q) return 2;
                for (int g = 0; g < nrun; ++g)
                    if (run_q[g] == q) return 1;
                return 0;
            };
            int cnt[3] = {0, 0, 0};
            for (int j0 = slo; j0 < shi; j0 += 64) {
                const int j = j0 + lane;
                const int kd = j < shi ? kind(j) : -1;
                for (int c = 0; c < 3; ++c) {
                    int tot;
                    (void)lane_rank(kd == c, lane, tot);
                    cnt[c] += tot;
                }
            }
            const int ngen = cnt[0], nrn = cnt[1];
            nsolo = cnt[2];
            {
                int at[3] = {0, ngen, ngen + nrn};
                for (int j0 = slo; j0 < shi; j0 += 64) {
                    const int j = j0 + lane;
                    const int kd = j < shi ? kind(j) : -1;
                    for (int c = 0; c < 3; ++c) {
                        int tot;
                        const int rk = lane_rank(kd == c, lane, tot);
                        if (kd == c) ws.l0[at[c] + rk] = (unsigned short)j;
                        at[c] += tot;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();

            // ---- general ratings: greedy steps of up to G conflict-free slots ---------------------------
            {
                const unsigned short* list = ws.l0;
                const int n = ngen;
                for (int x2 = lane; x2 < n; x2 += 64) {
                    atomicAdd(&ws.remdeg[rat_p[list[x2]]], 1);
                    atomicAdd(&ws.remdeg[rat_q[list[x2]]], 1);
                }
                __builtin_amdgcn_wave_barrier();
                unsigned taken_mask = 0;  // bit k: this lane's candidate lane + 64 k is taken (M <= 2048)
                int remaining = n;
                short t = 1;
                while (remaining > 0) {
                    ++t;
                    int ntake = 0;
                    unsigned long long slot_taken = 0;
                    // the host takes the last G or fewer candidates in list order, everything before that in
                    // priority order (the busier row's remaining work, then the other row's, then list order)
                    const bool by_pos = remaining <= G;
                    for (int round = 0; round < G; ++round) {
                        unsigned long long best = 0;
                        for (int k = 0, x2 = lane; x2 < n; ++k, x2 += 64) {
                            if ((taken_mask >> k) & 1u) continue;
                            const int j = list[x2];
                            const int p = rat_p[j], q = rat_q[j];
                            if (ws.last[p] == t || ws.last[q] == t) continue;
                            if (ws.prev[p] == (short)(t - 1)) continue;
                            if (ws.prev[q] == (short)(t - 1) && ((slot_taken >> ws.slot[q]) & 1ull)) continue;
                            const unsigned aa = (unsigned)ws.remdeg[p], bb = (unsigned)ws.remdeg[q];
                            const unsigned long long hi2 = aa > bb ? aa : bb, lo2 = aa > bb ? bb : aa;
                            const unsigned long long key = (by_pos ? 0ull : ((hi2 << 44) | (lo2 << 24))) | (unsigned long long)(0xFFFFFF - (unsigned)x2);
                            best = key > best ? key : best;
                        }
                        best = wave_max_u64(best);
                        if (best == 0) break;
                        const int x2 = (int)(0xFFFFFFu - (unsigned)(best & 0xFFFFFFull));
                        const int j = list[x2];
                        const int p = rat_p[j], q = rat_q[j];
                        int req = -1;
                        if (ws.prev[q] == (short)(t - 1)) {
                            req = ws.slot[q];
                            slot_taken |= 1ull << req;
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) {
                            ws.last[p] = t;
                            ws.last[q] = t;
                            ws.tk[2 * ntake] = (unsigned short)x2;
                            ws.tk[2 * ntake + 1] = (unsigned short)(req & 0xFF);
                        }
                        if ((x2 & 63) == lane) taken_mask |= 1u << (x2 >> 6);
                        ++ntake;
                        __builtin_amdgcn_wave_barrier();
                    }
                    // lane slots: forwarded rows keep theirs, the rest fill the free ones in take order
                    unsigned long long freemask = (G >= 64 ? ~0ull : ((1ull << G) - 1ull)) & ~slot_taken;
                    if (a.emit) idle_general(stepcur + (unsigned)ns);
                    __builtin_amdgcn_wave_barrier();
                    // (uniform, sequential: at most G takes)
                    unsigned long long used = 0;  // slots that carry a rating, for the order
                    for (int y = 0; y < ntake; ++y) {
                        const int x2 = ws.tk[2 * y];
                        int g = (signed char)(ws.tk[2 * y + 1] & 0xFF);
                        const bool fwd = g >= 0;
                        if (!fwd) {
                            g = __builtin_ctzll(freemask);
                            freemask &= ~(1ull << g);
                        }
                        used |= 1ull << g;
                        const int j = list[x2];
                        const int p = rat_p[j], q = rat_q[j];
                        if (lane == 0) {
                            if (a.emit) put_entry(stepcur + (unsigned)ns, g, enc_slots(p, q, fwd, L), rat_r[j], cdecay);
                            ws.remdeg[p]--;
                            ws.remdeg[q]--;
                            ws.prev[p] = t;
                            ws.prev[q] = t;
                            ws.slot[p] = (signed char)g;
                            ws.slot[q] = (signed char)g;
                            ws.tk[2 * y + 1] = (unsigned short)g;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (a.emit && ntake > 0 && lane == 0) {
                        // order: the step's ratings in slot order
                        for (int g = 0; g < G; ++g) {
                            if (!((used >> g) & 1ull)) continue;
                            for (int y = 0; y < ntake; ++y)
                                if (ws.tk[2 * y + 1] == (unsigned short)g) {
                                    const unsigned jj = rat_i[list[ws.tk[2 * y]]];
                                    a.order[ord_at++] = a.orig ? a.orig[jj] : (long long)jj;
                                }
                        }
                    }
                    if (a.emit) ord_at = __shfl(ord_at, 0, 64);
                    ++ns;
                    remaining -= ntake;
                    if (ns > 0xFFFF) break;
                }
            }
            // ---- run ratings: item g keeps lane slot g, its row stays resident ----------------------------
            if (nrn > 0 && ns <= 0xFFFF) {
                const unsigned short* list = ws.l0 + ngen;
                // the run starts with fresh loads: no hazard against the general steps
                for (int rr = lane; rr < nrows; rr += 64) {
                    ws.last[rr] = 0;
                    ws.prev[rr] = 0;
                }
                __builtin_amdgcn_wave_barrier();
                // queues: positions of the list grouped by slot, in list order; tk[] = [qitems (nrn)] ; starts in regs
                int qstart[64], qlen[64], head[64];
                {
                    int at = 0;
                    for (int g = 0; g < nrun; ++g) {
                        qstart[g] = at;
                        int len = 0;
                        for (int x0 = 0; x0 < nrn; x0 += 64) {
                            const int x2 = x0 + lane;
                            const bool mine = x2 < nrn && rat_q[list[x2]] == run_q[g];
                            int tot;
                            const int rk = lane_rank(mine, lane, tot);
                            if (mine) ws.tk[at + len + rk] = (unsigned short)x2;
                            len += tot;
                        }
                        qlen[g] = len;
                        head[g] = 0;
                        at += len;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                int remaining = nrn;
                short t = 1;
                const unsigned step0 = stepcur + (unsigned)ns;
                while (remaining > 0) {
                    ++t;
                    // longest queue first (stable)
                    int ord[64];
                    for (int g = 0; g < nrun; ++g) ord[g] = g;
                    for (int x1 = 1; x1 < nrun; ++x1) {  // insertion sort, stable, descending by what is left
                        const int g = ord[x1];
                        const int left = qlen[g] - head[g];
                        int pos = x1;
                        while (pos > 0 && (qlen[ord[pos - 1]] - head[ord[pos - 1]]) < left) {
                            ord[pos] = ord[pos - 1];
                            --pos;
                        }
                        ord[pos] = g;
                    }
                    if (a.emit)
                        for (int g = lane; g < G; g += 64)
                            put_entry(step0 + (unsigned)nr, g, enc_slots(nrows + 2 * g, g < nrun ? run_q[g] : nrows + 2 * g + 1, true, L), 0.0f, 1.0f);
                    __builtin_amdgcn_wave_barrier();
                    unsigned long long used = 0;
                    int took_pos[64];
                    for (int oi = 0; oi < nrun; ++oi) {
                        const int g = ord[oi];
                        took_pos[g] = -1;
                        // first rating of this slot whose user is free now and was not used last step
                        for (int x0 = head[g]; x0 < qlen[g]; x0 += 64) {
                            const int xx = x0 + lane;
                            bool ok = false;
                            if (xx < qlen[g]) {
                                const unsigned short v = ws.tk[qstart[g] + xx];
                                if (v != 0xFFFFu) {
                                    const int p = rat_p[list[v]];
                                    ok = !(ws.last[p] == t || ws.prev[p] == (short)(t - 1));
                                }
                            }
                            const unsigned long long hit = __ballot(ok);
                            if (hit) {
                                const int xx2 = x0 + __builtin_ctzll(hit);
                                const int v = ws.tk[qstart[g] + xx2];
                                const int j = list[v];
                                const int p = rat_p[j];
                                __builtin_amdgcn_wave_barrier();
                                if (lane == 0) {
                                    ws.last[p] = t;
                                    ws.tk[qstart[g] + xx2] = 0xFFFFu;
                                    if (a.emit) put_entry(step0 + (unsigned)nr, g, enc_slots(p, rat_q[j], false, L), rat_r[j], cdecay);
                                }
                                took_pos[g] = v;
                                used |= 1ull << g;
                                --remaining;
                                __builtin_amdgcn_wave_barrier();
                                break;
                            }
                        }
                        // advance the head over taken entries
                        while (head[g] < qlen[g] && ws.tk[qstart[g] + head[g]] == 0xFFFFu) ++head[g];
                    }
                    if (a.emit && lane == 0)
                        for (int g = 0; g < nrun; ++g)
                            if ((used >> g) & 1ull) {
                                const unsigned jj = rat_i[list[took_pos[g]]];
                                a.order[ord_at++] = a.orig ? a.orig[jj] : (long long)jj;
                            }
                    if (a.emit) ord_at = __shfl(ord_at, 0, 64);
                    // users of this step become "previous step" users
                    if (lane == 0)
                        for (int g = 0; g < nrun; ++g)
                            if ((used >> g) & 1ull) ws.prev[rat_p[list[took_pos[g]]]] = t;
                    __builtin_amdgcn_wave_barrier();
                    ++nr;
                    if (nr > 0xFFFF) break;
                }
                if ((nr & 1) && nr <= 0xFFFF) {  // the run loop is unrolled by two: pad with an all-idle step
                    if (a.emit)
                        for (int g = lane; g < G; g += 64)
                            put_entry(step0 + (unsigned)nr, g, enc_slots(nrows + 2 * g, g < nrun ? run_q[g] : nrows + 2 * g + 1, true, L), 0.0f, 1.0f);
                    ++nr;
                }
            }
            // ---- solo records ----------------------------------------------------------------------------
            if (nsolo > 0) {
                const unsigned short* list = ws.l0 + ngen + nrn;
                units = (nsolo + 2 + G - 1) / G + kSoloPad;
                if (a.emit) {
                    const unsigned step0 = stepcur + (unsigned)ns + (unsigned)nr;
                    for (int pad = 0; pad < kSoloPad; ++pad) idle_general(step0 + (unsigned)pad);
                    Entry* rec = a.entries + ((size_t)ent_base + step0 + kSoloPad) * G;
                    const int q = solo_q;
                    const unsigned zero_slots = enc_slots(nrows, q, false, L);
                    const int total = (units - kSoloPad) * G;  // records incl. header, terminator, padding
                    for (int x2 = lane; x2 < total; x2 += 64) {
                        unsigned wd[4];
                        const int tt = x2 - 1;  // step of this record (-1: header)
                        auto slots_of = [&](int s2) { return s2 < nsolo ? enc_slots(rat_p[list[s2]], q, false, L) : zero_slots; };
                        // record = {slots of the NEXT step, mailbox, lr * r, r} (schedule.cpp pack_solo)
                        if (x2 == 0) {
                            wd[0] = slots_of(0); wd[1] = 0u; wd[2] = 0u; wd[3] = 0u;
                        } else if (tt < nsolo) {
                            const float rr = rat_r[list[tt]];
                            wd[0] = slots_of(tt + 1);
                            wd[1] = 0xFFFFFFFFu;
                            wd[2] = __builtin_bit_cast(unsigned, lr * rr);
                            wd[3] = __builtin_bit_cast(unsigned, rr);
                        } else {
                            wd[0] = zero_slots; wd[1] = 0xFFFFFFFFu; wd[2] = 0u; wd[3] = 0u;
                        }
                        Entry e;
                        __builtin_memcpy(&e, wd, 16);
                        rec[x2] = e;
                    }
                    for (int x2 = lane; x2 < nsolo; x2 += 64) {
                        const unsigned jj = rat_i[list[x2]];
                        a.order[ord_at + x2] = a.orig ? a.orig[jj] : (long long)jj;
                    }
                }
            }
        }
        if (lane == 0) {
            int* si = sub_info + x * 4;
            si[0] = ns;
            si[1] = nr;
            si[2] = nsolo;
            si[3] = units;
            if (ns > 0xFFFF || nr > 0xFFFF || nsolo > 0xFFFF) misc[0] = 1;
        }
    }
    __syncthreads();
    // ---- per-cell summary (COUNT) / trailing idle steps (EMIT) -------------------------------------------
    if (tid == 0) {
        unsigned stepcur = 0;
        long long crit = 0;
        bool has_run = false, fail = misc[0] != 0;
        for (int s = 0; s < W && !fail; ++s) {
            unsigned smax = 0;
            for (int w = 0; w < W; ++w) {
                const int* si = sub_info + (s * W + w) * 4;
                if (stepcur > 0xFFFFu) fail = true;
                if (a.emit != 1) a.subs[cell * WW + s * W + w] = SubDesc{stepcur | ((unsigned)si[2] << 16), (unsigned)si[0] | ((unsigned)si[1] << 16)};
                stepcur += (unsigned)(si[0] + si[1] + si[3]);
                if (si[1] > 0 || si[2] > 0) has_run = true;
                const unsigned cost = (unsigned)(si[0] + si[1] + si[2] * 3 / 4);
                smax = cost > smax ? cost : smax;
            }
            crit += smax;
        }
        if (a.emit != 1) {
            info.status = fail ? 1 : 0;
            info.nu = (unsigned)nu;
            info.ni = (unsigned)ni;
            info.n_steps = stepcur + 2;
            info.has_run = has_run ? 1u : 0u;
            info.crit = crit;
            a.info[cell] = info;
        }
        misc[3] = (int)stepcur;
    }
    __syncthreads();
    if (a.emit == 1 && wave == 0) {  // (one-pass mode: compact_kernel writes them)
        const unsigned stepcur = (unsigned)misc[3];
        idle_general(stepcur);
        idle_general(stepcur + 1);
    }
}

// One-pass mode, second half: the cells whose row_off is not 0xFFFFFFFF move from the scratch arrays (worst-case offsets,
// see pack_kernel) to their final places; the two trailing idle steps of a cell are written here.  One workgroup per cell.
__global__ void __launch_bounds__(256) compact_kernel(const PackArgs a, const uint32_t* __restrict__ srows, const Entry* __restrict__ sent) {
    const long long cell = blockIdx.x;
    if (a.row_off[cell] == 0xFFFFFFFFu) return;
    const PackCellInfo ci = a.info[cell];
    if (ci.n_steps == 0) return;
    const int W = a.W, WW = W * W, G = a.G, L = a.L;
    const long long lo = a.bptr[cell * WW];
    const int nrows = (int)(ci.nu + ci.ni);
    const size_t row_src = 2 * (size_t)lo, row_dst = (size_t)a.row_off[cell];
    for (int x = threadIdx.x; x < nrows; x += 256) a.rows[row_dst + (size_t)x] = srows[row_src + (size_t)x];
    const size_t ent_src = 2 * (size_t)lo + (size_t)cell * (size_t)(WW * kPackSlack + 2), ent_dst = (size_t)a.ent_off[cell];
    const uint4* src4 = reinterpret_cast<const uint4*>(sent);
    uint4* dst4 = reinterpret_cast<uint4*>(a.entries);
    for (int x = 0; x < WW; ++x) {
        const SubDesc sd = a.subs[cell * WW + x];
        const int nsolo = (int)(sd.off >> 16), ns = (int)(sd.n & 0xFFFFu), nr = (int)(sd.n >> 16);
        const int units = nsolo > 0 ? (nsolo + 2 + G - 1) / G + kSoloPad : 0;
        const long long cnt = (long long)(ns + nr + units) * G;
        const size_t s0 = (ent_src + 2 * (size_t)(a.bptr[cell * WW + x] - lo) + (size_t)x * kPackSlack) * (size_t)G;
        const size_t d0 = (ent_dst + (size_t)(sd.off & 0xFFFFu)) * (size_t)G;
        for (long long y = threadIdx.x; y < cnt; y += 256) dst4[d0 + (size_t)y] = src4[s0 + (size_t)y];
    }
    if ((int)threadIdx.x < 2 * G) {
        const int step = (int)ci.n_steps - 2 + (int)threadIdx.x / G, g = (int)threadIdx.x % G;
        Entry e;
        e.slots = enc_slots(nrows + 2 * g, nrows + 2 * g + 1, false, L);
        e.r = 0.0f;
        e.lrr = a.lr * 0.0f;
        e.ce = a.c;
        a.entries[(ent_dst + (size_t)step) * G + g] = e;
    }
}

hipError_t launch_compact(const PackArgs& a, long long n_cells, const uint32_t* srows, const Entry* sent, hipStream_t st) {
    if (n_cells <= 0) return hipSuccess;
    hipLaunchKernelGGL(compact_kernel, dim3((unsigned)n_cells), dim3(256), 0, st, a, srows, sent);
    return hipGetLastError();
}

// steps / rows of the scratch arrays of the one-pass mode for n ratings in n_cells cells of W*W sub-cells
size_t pack_scratch_steps(long long n, long long n_cells, int W) { return 2 * (size_t)n + (size_t)n_cells * (size_t)(W * W * kPackSlack + 2); }

// dst[seg.dst + x] = src[seg.src + x] for x < seg.n, elements of `elem` bytes (a multiple of 4): one workgroup
// per segment.  Places the pieces the host packed in a mixed build.
__global__ void __launch_bounds__(256) scatter_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src,
                                                      const MixedSegment* __restrict__ segs, const int words_per_elem) {
    const MixedSegment s = segs[blockIdx.x];
    unsigned* d = dst + s.dst * words_per_elem;
    const unsigned* q = src + s.src * words_per_elem;
    const unsigned long long nw = s.n * (unsigned long long)words_per_elem;
    for (unsigned long long x = threadIdx.x; x < nw; x += 256) d[x] = q[x];
}

hipError_t launch_scatter(void* dst, const void* src, const MixedSegment* segs, long long n_segs, int elem_bytes, hipStream_t st) {
    if (n_segs <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)n_segs), dim3(256), 0, st, static_cast<unsigned*>(dst),
                       static_cast<const unsigned*>(src), segs, elem_bytes / 4);
    return hipGetLastError();
}

size_t pack_lds_bytes(const PackArgs& a) {
    const size_t M = (size_t)a.max_m, R = (size_t)a.max_rows;
    const size_t words = (size_t)a.u_words + a.i_words;
    size_t b = M * (2 + 2 + 4 + 4) + words * 4 + (words + (words & 1)) * 2;
    const size_t slot_bytes = (R + 3) & ~(size_t)3;
    b += (size_t)kWaves * (R * 4 + R * 2 * 2 + slot_bytes + M * 2 + M * 4);
    b += ((size_t)a.W * a.W * 4 + 8) * 4;
    return (b + 15) & ~(size_t)15;
}

hipError_t launch_pack(const PackArgs& a, long long n_cells, hipStream_t st) {
    const size_t lds = pack_lds_bytes(a);
    hipError_t e = hipFuncSetAttribute((const void*)pack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)n_cells), dim3(64 * kWaves), lds, st, a);
    return hipGetLastError();
}

}  // namespace mfsgd
