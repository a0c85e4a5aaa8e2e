// schedule.cpp -- see schedule.hpp.  Host only; no HIP calls.
#include "schedule.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <queue>
#include <thread>

namespace mfsgd {

Geometry geometry_for_k(int k) {
    Geometry g{};
    g.k = k;
    int need = (k + 3) / 4, L = 1;
    while (L < need) L <<= 1;
    g.L = L;
    g.G = 64 / L;
    g.kp = 4 * L;
    g.rowbytes = 16 * L;
    return g;
}

int64_t sched_bytes_for(const Geometry& geo, int W, int nrows, int64_t n_steps) {
    const int64_t b = n_steps * geo.G * 16 + (int64_t)W * W * 8 + (int64_t)nrows * 4;
    return (b + 15) & ~(int64_t)15;
}

int64_t rows_bytes_for(const Geometry& geo, int nrows) { return (int64_t)(nrows + 2 * geo.G) * geo.rowbytes; }

namespace {

// Longest-processing-time-first assignment of rows to `nbins` bins by rating
// count.  Rows with no rating go to bin 0 (they are never touched).
// Deterministic: ties broken by row index / bin index.
void lpt_assign(const std::vector<int64_t>& deg, int nbins, std::vector<int32_t>& bin) {
    const int64_t n = (int64_t)deg.size();
    bin.assign((size_t)n, 0);
    std::vector<int32_t> idx;
    idx.reserve((size_t)n);
    for (int64_t x = 0; x < n; ++x)
        if (deg[(size_t)x] > 0) idx.push_back((int32_t)x);
    std::stable_sort(idx.begin(), idx.end(),
                     [&](int32_t a, int32_t b) { return deg[(size_t)a] > deg[(size_t)b]; });
    using Item = std::pair<int64_t, int32_t>;  // (load, bin): smallest load, then smallest bin
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
    for (int32_t b = 0; b < nbins; ++b) heap.push({0, b});
    for (int32_t x : idx) {
        Item t = heap.top();
        heap.pop();
        bin[(size_t)x] = t.second;
        t.first += deg[(size_t)x];
        heap.push(t);
    }
}

struct Rat {
    uint16_t p, q;  // LDS slots
    float r;
    int64_t idx;    // caller-visible rating index
};

struct CellOut {
    std::vector<uint32_t> rows;
    std::vector<Entry> entries;
    std::vector<SubDesc> subs;
    std::vector<int64_t> order;
    uint32_t nu = 0, ni = 0, n_steps = 0;
    int64_t crit = 0;
    bool has_run = false;
};

struct Scratch {
    std::vector<int32_t> remdeg;    // ratings left on a row inside the current sub-cell
    std::vector<int32_t> laststep;  // stamp of the step that selected the row (selection phase)
    std::vector<int32_t> prevstep;  // stamp of the last EMITTED step that used the row
    std::vector<int8_t> lastslot;   // lane slot the row had in that step
    std::vector<uint32_t> us, is;
    std::vector<Rat> rats;
    std::vector<int32_t> cand;
    std::vector<uint64_t> keys;
    std::vector<std::pair<int32_t, uint16_t>> top;
};

constexpr int kRunMin = 8;  // shortest item chain worth a register-resident run

// Packs the ratings rs[0..n) of one sub-cell into steps of G conflict-free
// slots.  `t0` is the running step stamp of the cell (unique per step).
// Entry word: p-side LDS address | q-side LDS address << 16 | forward flag << 31,
// addresses in 16-byte units (slot * L; at most 160 KiB / 16 = 10240 < 2^15).
inline uint32_t encode_slots(int pslot, int qslot, bool fwd_q, int Lg) {
    return (uint32_t)(pslot * Lg) | ((uint32_t)(qslot * Lg) << 16) | (fwd_q ? 0x80000000u : 0u);
}

struct Hyper {
    float lr, c;
};

inline Entry make_entry(uint32_t slots, float r, float ce, const Hyper& hy) {
    Entry e;
    e.slots = slots;
    e.r = r;
    e.lrr = hy.lr * r;
    e.ce = ce;
    return e;
}

void pack_subcell(const Rat* rs, int n, int G, int Lg, int nrows, const Hyper& hy, Scratch& sc, int32_t& tstamp,
                  std::vector<Entry>& entries, std::vector<int64_t>& order, uint32_t& n_steps) {
    n_steps = 0;
    if (n == 0) return;
    for (int j = 0; j < n; ++j) {
        sc.remdeg[rs[j].p]++;
        sc.remdeg[rs[j].q]++;
    }
    sc.cand.resize((size_t)n);
    std::iota(sc.cand.begin(), sc.cand.end(), 0);
    int remaining = n;
    int taken[64];
    int slot_of[64];
    while (remaining > 0) {
        const int32_t t = ++tstamp;
        // priority: the rating whose busier row has the most work left goes first
        sc.keys.resize((size_t)remaining);
        for (int c = 0; c < remaining; ++c) {
            const Rat& x = rs[sc.cand[(size_t)c]];
            const uint32_t a = (uint32_t)sc.remdeg[x.p], b = (uint32_t)sc.remdeg[x.q];
            const uint64_t hi = a > b ? a : b, lo = a > b ? b : a;
            // descending by (hi, lo), ascending by position: encode position inverted
            sc.keys[(size_t)c] = (hi << 44) | (lo << 24) | (uint64_t)(0xFFFFFF - (uint32_t)c);
        }
        // Eligibility (the kernel prefetches the rows of step t+1 before it stores
        // the rows of step t -- DESIGN.md section 4):
        //  * a p-side row used in step t-1 may not be used in step t at all;
        //  * a q-side row used in step t-1 may be used in step t only in the SAME
        //    lane slot (the kernel then forwards it in registers).
        int ntake = 0;
        uint64_t slot_taken = 0;
        auto try_take = [&](int c) {
            const Rat& x = rs[sc.cand[(size_t)c]];
            if (sc.laststep[x.p] == t || sc.laststep[x.q] == t) return;
            if (sc.prevstep[x.p] == t - 1) return;
            int req = -1;
            if (sc.prevstep[x.q] == t - 1) {
                req = sc.lastslot[x.q];
                if ((slot_taken >> req) & 1) return;
                slot_taken |= 1ull << req;
            }
            sc.laststep[x.p] = t;
            sc.laststep[x.q] = t;
            slot_of[ntake] = req;
            taken[ntake++] = c;
        };
        if (remaining <= G) {
            for (int c = 0; c < remaining && ntake < G; ++c) try_take(c);
        } else {
            static thread_local std::vector<int32_t> ordv;
            ordv.resize((size_t)remaining);
            std::iota(ordv.begin(), ordv.end(), 0);
            const int want = std::min(remaining, 4 * G + 8);
            auto by_key = [&](int32_t x, int32_t y) { return sc.keys[(size_t)x] > sc.keys[(size_t)y]; };
            if (remaining > want)
                std::partial_sort(ordv.begin(), ordv.begin() + want, ordv.end(), by_key);
            else
                std::sort(ordv.begin(), ordv.end(), by_key);
            int scanned = 0;
            for (; scanned < want && ntake < G; ++scanned) try_take(ordv[(size_t)scanned]);
            if (ntake < G && want < remaining) {
                std::sort(ordv.begin() + want, ordv.end(), by_key);
                for (; scanned < remaining && ntake < G; ++scanned) try_take(ordv[(size_t)scanned]);
            }
        }
        // lane slots: forwarded rows keep theirs, the rest fill the free ones
        uint64_t freemask = (G >= 64 ? ~0ull : ((1ull << G) - 1)) & ~slot_taken;
        for (int j = 0; j < ntake; ++j) {
            if (slot_of[j] >= 0) continue;
            const int g = __builtin_ctzll(freemask);
            slot_of[j] = g;
            freemask &= ~(1ull << g);
        }
        // emit
        const size_t base = entries.size();
        entries.resize(base + (size_t)G);
        for (int g = 0; g < G; ++g) {
            entries[base + (size_t)g] = make_entry(encode_slots(nrows + 2 * g, nrows + 2 * g + 1, false, Lg), 0.0f, hy.c, hy);
        }
        int64_t ord_tmp[64];
        for (int g = 0; g < G; ++g) ord_tmp[g] = -1;
        for (int j = 0; j < ntake; ++j) {
            const Rat& x = rs[sc.cand[(size_t)taken[j]]];
            const int g = slot_of[j];
            entries[base + (size_t)g] = make_entry(encode_slots(x.p, x.q, sc.prevstep[x.q] == t - 1, Lg), x.r, hy.c, hy);
            ord_tmp[g] = x.idx;
            sc.remdeg[x.p]--;
            sc.remdeg[x.q]--;
            sc.prevstep[x.p] = t;
            sc.prevstep[x.q] = t;
            sc.lastslot[x.p] = (int8_t)g;
            sc.lastslot[x.q] = (int8_t)g;
        }
        for (int g = 0; g < G; ++g)
            if (ord_tmp[g] >= 0) order.push_back(ord_tmp[g]);
        ++n_steps;
        if (ntake == 0) continue;  // bubble: every remaining rating waits out the hazard rule
        // remove taken candidates (positions in cand), keeping relative order
        std::sort(taken, taken + ntake);
        int wpos = taken[0], next = 0;
        for (int c = taken[0]; c < remaining; ++c) {
            if (next < ntake && taken[next] == c) {
                ++next;
                continue;
            }
            sc.cand[(size_t)wpos++] = sc.cand[(size_t)c];
        }
        remaining -= ntake;
    }
}


// Packs the ratings of up to G "run" items of a sub-cell: item j keeps lane slot j and
// its row stays in registers for the whole run (kernel: run loop).  Each step takes at
// most one rating per slot; users are distinct inside a step and never repeat in
// consecutive steps (their rows are prefetched one step ahead).  Idle slots carry the
// idle flag (bit 31) and the slot's item address, so the kernel can skip them.
void pack_run(const Rat* rs, int n, const uint16_t* run_q, int nrun, int G, int Lg, int nrows, const Hyper& hy,
              Scratch& sc, int32_t& tstamp, std::vector<Entry>& entries,
              std::vector<int64_t>& order, uint32_t& n_steps) {
    n_steps = 0;
    if (n == 0) return;
    // per-slot queues of rating positions, in input order
    std::vector<int32_t> qpos[64];
    for (int j = 0; j < n; ++j) {
        int slot = -1;
        for (int g = 0; g < nrun; ++g)
            if (run_q[g] == rs[j].q) slot = g;
        qpos[slot].push_back(j);
    }
    size_t head[64] = {0};
    int remaining = n;
    while (remaining > 0) {
        const int32_t t = ++tstamp;
        // longest queue first
        int ord[64];
        for (int g = 0; g < nrun; ++g) ord[g] = g;
        std::stable_sort(ord, ord + nrun, [&](int a, int b) {
            return qpos[a].size() - head[a] > qpos[b].size() - head[b];
        });
        const size_t base = entries.size();
        entries.resize(base + (size_t)G);
        for (int g = 0; g < G; ++g) {
            const int qslot = g < nrun ? (int)run_q[g] : nrows + 2 * g + 1;
            // idle run slot: zero p row, r = 0 (so s == 0) and ce = 1: the resident row is untouched
            entries[base + (size_t)g] = make_entry(encode_slots(nrows + 2 * g, qslot, true, Lg), 0.0f, 1.0f, hy);
        }
        int64_t ord_tmp[64];
        for (int g = 0; g < G; ++g) ord_tmp[g] = -1;
        for (int oi = 0; oi < nrun; ++oi) {
            const int g = ord[oi];
            // first rating of this slot whose user is free now and was not used last step
            for (size_t x = head[g]; x < qpos[g].size(); ++x) {
                const int j = qpos[g][x];
                if (j < 0) continue;
                const Rat& r = rs[j];
                if (sc.laststep[r.p] == t || sc.prevstep[r.p] == t - 1) continue;
                sc.laststep[r.p] = t;
                entries[base + (size_t)g] = make_entry(encode_slots(r.p, r.q, false, Lg), r.r, hy.c, hy);
                ord_tmp[g] = r.idx;
                qpos[g][x] = -1;
                --remaining;
                break;
            }
            while (head[g] < qpos[g].size() && qpos[g][head[g]] < 0) ++head[g];
        }
        for (int g = 0; g < G; ++g)
            if (ord_tmp[g] >= 0) order.push_back(ord_tmp[g]);
        // users of this step become "previous step" users
        for (int g = 0; g < G; ++g) {
            const uint32_t pa = entries[base + (size_t)g].slots & 0xFFFFu;
            const int pslot = (int)(pa / (uint32_t)Lg);
            if (pslot < nrows) sc.prevstep[pslot] = t;
        }
        ++n_steps;
    }
    // the kernel's run loop is unrolled by two: pad to an even number of steps with an
    // all-idle step (p = zero row, r = 0, ce = 1 leaves the resident rows untouched)
    if (n_steps & 1) {
        for (int g = 0; g < G; ++g) {
            const int qslot = g < nrun ? (int)run_q[g] : nrows + 2 * g + 1;
            entries.push_back(make_entry(encode_slots(nrows + 2 * g, qslot, true, Lg), 0.0f, 1.0f, hy));
        }
        ++n_steps;
    }
}

}  // namespace

int build_schedule(const SchedParams& prm, const int32_t* u, const int32_t* i, const float* r,
                   const int64_t* orig, int64_t n, Schedule& out, std::string& err) {
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_last = t_begin;
    const bool trace = std::getenv("MFSGD_SCHED_TRACE") != nullptr;
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[schedule] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    const Geometry geo = geometry_for_k(prm.k);
    const int B = prm.B, W = prm.W, G = geo.G;
    if (B < 1 || W < 1 || W > 8 || prm.k < 1 || geo.L > 64) {
        err = "build_schedule: bad geometry (B, W or k)";
        return -1;
    }
    if ((int64_t)B * B * W * W > (int64_t)1 << 28) {
        err = "build_schedule: B*W too large";
        return -1;
    }
    const int32_t U = prm.U, I = prm.I;
    int nthreads = prm.threads > 0 ? prm.threads : (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;

    // ---- degrees and LPT partition into B*W fine bins -----------------------
    std::vector<int64_t> degu((size_t)U, 0), degi((size_t)I, 0);
    bool on_device = prm.ingest && prm.ingest->degrees && prm.ingest->bucket;
    if (on_device) {
        for (int64_t j = 0; j < n && on_device; ++j)
            if (u[j] < 0 || u[j] >= U || i[j] < 0 || i[j] >= I) on_device = false;  // let the host loop report it
        if (on_device && prm.ingest->degrees(prm.ingest->ctx, u, i, n, U, I, degu.data(), degi.data()) != 0) {
            on_device = false;
            std::fill(degu.begin(), degu.end(), 0);
            std::fill(degi.begin(), degi.end(), 0);
        }
    }
    if (!on_device) {
        for (int64_t j = 0; j < n; ++j) {
            if (u[j] < 0 || u[j] >= U || i[j] < 0 || i[j] >= I) {
                err = "set_ratings: index out of range at rating " + std::to_string(j);
                return -1;
            }
            degu[(size_t)u[j]]++;
            degi[(size_t)i[j]]++;
        }
    }
    lap(on_device ? "degrees (device)" : "degrees");
    std::vector<int32_t> ubin, ibin;
    lpt_assign(degu, B * W, ubin);
    lpt_assign(degi, B * W, ibin);
    lap("LPT partition");
    // fine bin f -> block f % B, sub-group f / B

    // ---- counting sort by (cell, sub-round, wave) ---------------------------
    const int64_t nb = (int64_t)B * B * W * W;
    std::vector<int64_t> bptr((size_t)nb + 1, 0);
    auto bucket_of = [&](int64_t j) -> int64_t {
        const int32_t fu = ubin[(size_t)u[j]], fi = ibin[(size_t)i[j]];
        const int ub = fu % B, us = fu / B, it = fi % B, is = fi / B;
        const int s = (is - us + W) % W;
        return (((int64_t)ub * B + it) * W + s) * W + us;
    };
    std::vector<int64_t> sorted((size_t)n);
    if (on_device && prm.ingest->bucket(prm.ingest->ctx, u, i, n, ubin.data(), ibin.data(), U, I, B, W, bptr.data(),
                                        sorted.data()) != 0) {
        on_device = false;
        std::fill(bptr.begin(), bptr.end(), 0);
    }
    if (!on_device) {
        std::vector<int64_t> bkt((size_t)n);
        for (int64_t j = 0; j < n; ++j) {
            const int64_t b = bucket_of(j);
            bkt[(size_t)j] = b;
            bptr[(size_t)b + 1]++;
        }
        for (int64_t b = 0; b < nb; ++b) bptr[(size_t)b + 1] += bptr[(size_t)b];
        std::vector<int64_t> cur(bptr.begin(), bptr.end() - 1);
        for (int64_t j = 0; j < n; ++j) sorted[(size_t)cur[(size_t)bkt[(size_t)j]]++] = j;
    }
    lap(on_device ? "bucket (device radix sort)" : "bucket (counting sort)");

    // ---- per-cell packing (parallel over cells) ------------------------------
    const int64_t ncell = (int64_t)B * B;
    std::vector<CellOut> co((size_t)ncell);
    std::atomic<int64_t> next_cell{0};
    std::atomic<int> failed{0};
    std::string fail_msg;
    const int WW = W * W;
    const Hyper hy{prm.lr, 1.0f - prm.lr * prm.lambda};
    auto worker = [&]() {
        Scratch sc;
        for (;;) {
            const int64_t c = next_cell.fetch_add(1);
            if (c >= ncell || failed.load()) break;
            const int64_t lo = bptr[(size_t)(c * WW)], hi = bptr[(size_t)((c + 1) * WW)];
            CellOut& o = co[(size_t)c];
            o.subs.assign((size_t)WW, SubDesc{0, 0});
            if (hi == lo) continue;
            const int64_t m = hi - lo;
            sc.us.resize((size_t)m);
            sc.is.resize((size_t)m);
            for (int64_t x = 0; x < m; ++x) {
                const int64_t j = sorted[(size_t)(lo + x)];
                sc.us[(size_t)x] = (uint32_t)u[j];
                sc.is[(size_t)x] = (uint32_t)i[j];
            }
            std::vector<uint32_t> uu(sc.us.begin(), sc.us.begin() + m), ii(sc.is.begin(), sc.is.begin() + m);
            std::sort(uu.begin(), uu.end());
            uu.erase(std::unique(uu.begin(), uu.end()), uu.end());
            std::sort(ii.begin(), ii.end());
            ii.erase(std::unique(ii.begin(), ii.end()), ii.end());
            const int nu = (int)uu.size(), ni = (int)ii.size(), nrows = nu + ni;
            if ((int64_t)(nrows + 2 * G) * geo.L > 32767) {
                if (!failed.exchange(1)) fail_msg = "lds: cell touches more rows than LDS can address";
                break;
            }
            o.nu = (uint32_t)nu;
            o.ni = (uint32_t)ni;
            o.rows.reserve((size_t)nrows);
            o.rows.insert(o.rows.end(), uu.begin(), uu.end());
            o.rows.insert(o.rows.end(), ii.begin(), ii.end());
            sc.rats.resize((size_t)m);
            for (int64_t x = 0; x < m; ++x) {
                const int64_t j = sorted[(size_t)(lo + x)];
                Rat t;
                t.p = (uint16_t)(std::lower_bound(uu.begin(), uu.end(), (uint32_t)u[j]) - uu.begin());
                t.q = (uint16_t)(nu + (std::lower_bound(ii.begin(), ii.end(), (uint32_t)i[j]) - ii.begin()));
                t.r = r[j];
                t.idx = orig ? orig[j] : j;
                sc.rats[(size_t)x] = t;
            }
            sc.remdeg.assign((size_t)nrows, 0);
            sc.laststep.assign((size_t)nrows, 0);
            sc.lastslot.assign((size_t)nrows, (int8_t)-1);
            sc.prevstep.assign((size_t)nrows, 0);
            int32_t tstamp = 1;               // step stamps start at 2 so that t-1 never matches 0
            o.entries.reserve((size_t)(m + m / 2 + G));
            o.order.reserve((size_t)m);
            uint32_t stepcur = 0;
            int64_t crit = 0;
            for (int s = 0; s < W; ++s) {
                uint32_t smax = 0;
                for (int w = 0; w < W; ++w) {
                    const int64_t sb = c * WW + (int64_t)s * W + w;
                    const int64_t slo = bptr[(size_t)sb] - lo, shi = bptr[(size_t)sb + 1] - lo;
                    uint32_t ns = 0, nr = 0;
                    const int nsub = (int)(shi - slo);
                    Rat* sub = sc.rats.data() + slo;
                    // run items: the (at most G) items whose ratings would dominate the step
                    // count of this sub-cell; their ratings go last, in run mode
                    uint16_t run_q[64];
                    int nrun = 0;
                    if (nsub >= kRunMin) {
                        for (int j = 0; j < nsub; ++j) sc.remdeg[sub[j].q]++;
                        std::vector<std::pair<int32_t, uint16_t>>& top = sc.top;
                        top.clear();
                        for (int j = 0; j < nsub; ++j) {
                            const int32_t d = sc.remdeg[sub[j].q];
                            if (d >= kRunMin && d * G >= nsub) top.push_back({-d, sub[j].q});
                        }
                        for (int j = 0; j < nsub; ++j) sc.remdeg[sub[j].q] = 0;
                        std::sort(top.begin(), top.end());
                        top.erase(std::unique(top.begin(), top.end()), top.end());
                        for (size_t x = 0; x < top.size() && nrun < G; ++x) run_q[nrun++] = top[x].second;
                    }
                    int ngen = nsub;
                    if (nrun > 0) {
                        // stable partition: general ratings first, run ratings after
                        auto is_run = [&](const Rat& x) {
                            for (int g = 0; g < nrun; ++g)
                                if (run_q[g] == x.q) return true;
                            return false;
                        };
                        ngen = (int)(std::stable_partition(sub, sub + nsub, [&](const Rat& x) { return !is_run(x); }) - sub);
                    }
                    ++tstamp;  // break stickiness across sub-cells
                    pack_subcell(sub, ngen, G, geo.L, nrows, hy, sc, tstamp, o.entries, o.order, ns);
                    if (nrun > 0) {
                        ++tstamp;  // the run starts with fresh loads: no hazard against the last general step
                        pack_run(sub + ngen, nsub - ngen, run_q, nrun, G, geo.L, nrows, hy, sc, tstamp, o.entries,
                                 o.order, nr);
                    }
                    if (ns > 0xFFFF || nr > 0xFFFF) {
                        if (!failed.exchange(1)) fail_msg = "lds: sub-cell has more than 65535 steps";
                        break;
                    }
                    if (nr > 0) o.has_run = true;
                    o.subs[(size_t)(s * W + w)] = SubDesc{stepcur, ns | (nr << 16)};
                    stepcur += ns + nr;
                    smax = std::max(smax, ns + nr);
                }
                crit += smax;
            }
            // two trailing idle steps: the kernel reads entries t+1 and t+2 ahead
            for (int pad = 0; pad < 2; ++pad)
                for (int g = 0; g < G; ++g)
                    o.entries.push_back(make_entry(encode_slots(nrows + 2 * g, nrows + 2 * g + 1, false, geo.L), 0.0f, hy.c, hy));
            o.n_steps = stepcur + 2;
            o.crit = crit;
        }
    };
    {
        std::vector<std::thread> th;
        const int nt = (int)std::min<int64_t>(nthreads, ncell);
        for (int t = 1; t < nt; ++t) th.emplace_back(worker);
        worker();
        for (auto& t : th) t.join();
    }
    if (failed.load()) {
        err = fail_msg;
        return -1;
    }
    lap("per-cell packing");

    // ---- concatenate in cell order; canonical order is round-major ----------
    out = Schedule{};
    out.geo = geo;
    out.B = B;
    out.W = W;
    out.nnz = n;
    out.cells.resize((size_t)ncell);
    out.subs.resize((size_t)(ncell * WW));
    int64_t tot_rows = 0, tot_steps = 0;
    int64_t sched_cap = 0, rows_cap = 0;
    for (int64_t c = 0; c < ncell; ++c) {
        const CellOut& o = co[(size_t)c];
        if (tot_rows > 0xFFFFFFFFll - (int64_t)o.rows.size() || tot_steps > 0xFFFFFFFFll - o.n_steps) {
            err = "build_schedule: schedule exceeds 32-bit offsets";
            return -1;
        }
        CellDesc d;
        d.row_off = (uint32_t)tot_rows;
        d.ent_off = (uint32_t)tot_steps;
        d.n_steps = o.n_steps | (o.has_run ? kCellCritical : 0u);
        d.nu = (uint16_t)o.nu;
        d.ni = (uint16_t)o.ni;
        out.cells[(size_t)c] = d;
        for (int x = 0; x < WW; ++x)
            out.subs[(size_t)(c * WW + x)] = o.subs.empty() ? SubDesc{0, 0} : o.subs[(size_t)x];
        tot_rows += (int64_t)o.rows.size();
        tot_steps += o.n_steps;
        sched_cap = std::max(sched_cap, sched_bytes_for(geo, W, (int)(o.nu + o.ni), (int64_t)o.n_steps));
        rows_cap = std::max(rows_cap, rows_bytes_for(geo, (int)(o.nu + o.ni)));
        out.max_cell_nnz = std::max<int64_t>(out.max_cell_nnz, (int64_t)o.order.size());
        out.max_cell_rows = std::max<int64_t>(out.max_cell_rows, (int64_t)o.rows.size());
        out.max_cell_steps = std::max<int64_t>(out.max_cell_steps, o.crit);
    }
    lap("  offsets");
    {
        const int64_t need = 16 + 2 * sched_cap + rows_cap;
        if (need > prm.lds_budget) {
            err = "lds: the largest cell needs " + std::to_string(need) + " bytes of LDS (budget " +
                  std::to_string(prm.lds_budget) + "); use more blocks";
            out.lds_bytes = (int)std::min<int64_t>(need, 0x7FFFFFFF);  // tells build_schedule_auto how far off it was
            return -1;
        }
        out.lds_bytes = (int)((need + 15) & ~(int64_t)15);
        out.sched_cap = (int)sched_cap;
    }
    out.total_rows = tot_rows;
    out.total_steps = tot_steps;
    out.rows.resize((size_t)tot_rows + 4, 0u);  // +16 B: the staging DMA reads whole 16-byte units
    out.entries.resize((size_t)(tot_steps * G));
    {
        std::atomic<int64_t> nc{0};
        auto copier = [&]() {
            for (;;) {
                const int64_t c = nc.fetch_add(64);
                if (c >= ncell) break;
                for (int64_t x = c; x < std::min<int64_t>(c + 64, ncell); ++x) {
                    const CellOut& o = co[(size_t)x];
                    const CellDesc& d = out.cells[(size_t)x];
                    if (!o.rows.empty())
                        std::memcpy(&out.rows[d.row_off], o.rows.data(), o.rows.size() * sizeof(uint32_t));
                    if (!o.entries.empty())
                        std::memcpy(&out.entries[(size_t)d.ent_off * G], o.entries.data(),
                                    o.entries.size() * sizeof(Entry));
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(copier);
        copier();
        for (auto& t : th) t.join();
    }
    lap("  copy rows/entries");
    out.order.resize((size_t)n);
    out.cell_ptr.assign((size_t)ncell + 1, 0);
    int64_t pos = 0;
    for (int rd = 0; rd < B; ++rd) {
        int64_t worst = 0;
        for (int b = 0; b < B; ++b) {
            const int64_t c = (int64_t)b * B + (b + rd) % B;
            const CellOut& o = co[(size_t)c];
            out.cell_ptr[(size_t)((int64_t)rd * B + b)] = pos;
            if (!o.order.empty())
                std::memcpy(&out.order[(size_t)pos], o.order.data(), o.order.size() * sizeof(int64_t));
            pos += (int64_t)o.order.size();
            worst = std::max(worst, o.crit);
        }
        out.sum_round_steps += worst;
    }
    out.cell_ptr[(size_t)ncell] = pos;
    if (pos != n) {
        err = "build_schedule: internal error, packed " + std::to_string(pos) + " of " + std::to_string(n);
        return -1;
    }
    lap("concatenate + order");
    out.device_ingest = on_device;
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return 0;
}

int build_schedule_auto(SchedParams prm, const int32_t* u, const int32_t* i, const float* r,
                        const int64_t* orig, int64_t n, Schedule& out, std::string& err) {
    const Geometry geo = geometry_for_k(prm.k);
    const int32_t minrows = std::max<int32_t>(1, std::min(prm.U, prm.I));
    int W = prm.W;
    if (W <= 0) {
        W = 4;
        while (W > 1 && (int64_t)W * 8 > minrows) W >>= 1;
    }
    const bool autoB = prm.B <= 0;
    int B = prm.B;
    if (autoB) {
        // rows a cell may hold in LDS, leaving a tenth for step entries
        const double cap_rows = (double)prm.lds_budget * 0.9 / geo.rowbytes - 2.0 * geo.G;
        // a cell of m ratings touches at most 2m rows, typically ~1.1m
        const double target_nnz = std::max(16.0, 0.6 * cap_rows);
        double best = std::ceil(std::sqrt((double)std::max<int64_t>(n, 1) / target_nnz));
        int64_t bb = (int64_t)best;
        if (bb <= prm.n_cu) {
            bb = (bb + 7) / 8 * 8;
            if (bb > prm.n_cu) bb = prm.n_cu;
            // a few more blocks than strictly needed keeps every CU busy
            if (bb > prm.n_cu * 0.7) bb = prm.n_cu;
        } else {
            bb = (bb + prm.n_cu - 1) / prm.n_cu * prm.n_cu;
        }
        const int64_t lim = std::max<int64_t>(1, minrows / W);
        if (bb > lim) bb = lim;
        if (bb < 1) bb = 1;
        B = (int)bb;
    }
    for (int attempt = 0; attempt < 6; ++attempt) {
        prm.B = B;
        prm.W = W;
        const int rc = build_schedule(prm, u, i, r, orig, n, out, err);
        if (rc == 0) return 0;
        if (!autoB || err.compare(0, 4, "lds:") != 0) return rc;
        // the rows of the offending cell shrink roughly like 1/B: jump straight to a B that fits
        int64_t nb = B <= prm.n_cu / 2 ? (int64_t)B * 2 : ((int64_t)B / prm.n_cu + 1) * prm.n_cu;
        if (out.lds_bytes > prm.lds_budget) {
            const double ratio = (double)out.lds_bytes / (double)prm.lds_budget * 1.08;
            int64_t want = (int64_t)std::ceil((double)B * ratio);
            want = want <= prm.n_cu ? (want + 7) / 8 * 8 : (want + prm.n_cu - 1) / prm.n_cu * prm.n_cu;
            nb = std::max(nb, want);
        }
        const int64_t lim = std::max<int64_t>(1, minrows / W);
        if (nb > lim) {
            if (W > 1) {
                W >>= 1;
                continue;
            }
            return rc;
        }
        B = (int)nb;
    }
    return -1;
}

}  // namespace mfsgd
