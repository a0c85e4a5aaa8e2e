// schedule.cpp -- see schedule.hpp.  Host only; no HIP calls.
#include "schedule.hpp"
#include "hugepages.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <iterator>
#include <memory>
#include <numeric>
#include <queue>
#include <mutex>
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
    // three sections, each a whole number of 16-byte units (the staging DMA copies such units)
    return n_steps * geo.G * 16 + (((int64_t)W * W * 8 + 15) & ~(int64_t)15) + (((int64_t)nrows * 4 + 15) & ~(int64_t)15);
}

int64_t rows_bytes_for(const Geometry& geo, int nrows) { return (int64_t)(nrows + 2 * geo.G) * geo.rowbytes; }

namespace {

// Longest-processing-time-first assignment of rows to `nbins` bins by rating
// count.  Rows with no rating go to bin 0 (they are never touched).
// Deterministic: ties broken by row index / bin index.
// `stride` > 0 (the item side of a schedule: bin f belongs to tile f % stride): a row that fills a bin by
// itself -- its count is at least the mean bin load, so LPT adds nothing to its bin -- and whose chain of
// dependent updates is long enough to be what the epoch waits for (count >= giant_min)
// gets its TILE to itself as well: the tile's other bins stay empty.  With nothing else in its tile, the
// hand-off of that tile moves one row instead of a hundred and its cell holds no work but the chain.  (The giants are the first rows LPT
// places, into bins 0, 1, ... in this order, so their tiles are known before anything else is placed.)
// Returns the number of such rows: they sit in bins 0 .. n - 1.
int lpt_assign(const std::vector<int64_t>& deg, int nbins, std::vector<int32_t>& bin, int stride = 0,
               int64_t giant_min = 0) {
    const int64_t n = (int64_t)deg.size();
    bin.assign((size_t)n, 0);
    std::vector<int32_t> idx;
    idx.reserve((size_t)n);
    int64_t total = 0;
    for (int64_t x = 0; x < n; ++x)
        if (deg[(size_t)x] > 0) {
            idx.push_back((int32_t)x);
            total += deg[(size_t)x];
        }
    // descending by rating count, ties in index order -- a counting sort when the counts are small enough for one
    // (they are: a row has at most as many ratings as there are rows on the other side); [r3] std::stable_sort of
    // 480 K users took 30 of the 65 ms of this function at the Netflix shape, and a second for 10 M users
    std::vector<int32_t> sdeg;  // the counts in sorted order (beside idx), when they fit: the ring below reads them in sequence
    {
        int64_t dmax = 0;
        for (int32_t x : idx) dmax = std::max(dmax, deg[(size_t)x]);
        if (dmax <= (int64_t)1 << 26) {
            std::vector<int64_t> start((size_t)dmax + 2, 0);
            for (int32_t x : idx) start[(size_t)(dmax - deg[(size_t)x]) + 1]++;  // bucket 0 = the largest count
            for (size_t d = 1; d < start.size(); ++d) start[d] += start[d - 1];
            std::vector<int32_t> sorted_idx(idx.size());
            sdeg.resize(idx.size());
            for (int32_t x : idx) {  // idx ascends: stable
                const int64_t d = deg[(size_t)x];
                const size_t at = (size_t)start[(size_t)(dmax - d)]++;
                sorted_idx[at] = x;
                sdeg[at] = (int32_t)d;
            }
            idx.swap(sorted_idx);
        } else {
            std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return deg[(size_t)a] > deg[(size_t)b]; });
        }
    }
    int giants = 0;
    if (stride > 0 && stride < nbins) {
        const int64_t mean = (total + nbins - 1) / nbins;
        while (giants < (int)idx.size() && giants < stride / 4 &&
               deg[(size_t)idx[(size_t)giants]] >= std::max(mean, giant_min))
            ++giants;
    }
    // Every row goes to the bin with the smallest (load, bin).  [r3] The smallest load never decreases and no load is
    // ever more than the largest count above it, so the bins live in a ring of buckets indexed by load: a bucket is
    // appended to while it lies ahead, sorted by bin once when the front reaches it, and then consumed from its start
    // (nothing can be added to the front bucket: every count is at least 1).  The same assignment as the heap below,
    // which stays for small inputs and absurd counts: 1.23 -> 0.49 s at 10 M users and 14 848 bins, 30 -> 9 ms at 480 K.
    const int64_t dmax_all = idx.empty() ? 0 : deg[(size_t)idx[0]];
    if (idx.size() >= 4096 && nbins >= 16 && dmax_all < ((int64_t)1 << 20) && !std::getenv("MFSGD_LPT_HEAP")) {  // (ring: 24 B per count)
        const int64_t ring = dmax_all + 1;
        std::vector<std::vector<int32_t>> bucket((size_t)ring);
        for (int32_t b = 0; b < nbins; ++b)
            if (!(b >= stride && stride > 0 && b % stride < giants)) bucket[0].push_back(b);  // ascending: sorted
        // (a bucket is filled from the front bucket in ascending bin order; through a long stretch of equal counts --
        // the tail of any real degree distribution -- it is filled from ONE front bucket and arrives sorted: `mixed`
        // remembers the buckets that were not, and only those are sorted when the front reaches them)
        std::vector<uint8_t> mixed((size_t)ring, 0);
        int64_t at = 0;  // the front bucket: the smallest load modulo the ring (kept by increments: no division per row)
        size_t pos = 0;  // consumed so far of the front bucket
        std::vector<int32_t>* front = &bucket[0];
        for (size_t r = 0; r < idx.size(); ++r) {
            const int32_t x = idx[r];
            while (pos == front->size()) {
                front->clear();
                mixed[(size_t)at] = 0;
                if (++at == ring) at = 0;
                pos = 0;
                front = &bucket[(size_t)at];
                if (mixed[(size_t)at]) std::sort(front->begin(), front->end());
            }
            const int32_t b = (*front)[pos++];
            bin[(size_t)x] = b;
            int64_t to = at + (sdeg.empty() ? deg[(size_t)x] : (int64_t)sdeg[r]);  // 1 <= count <= ring - 1: never the front bucket
            if (to >= ring) to -= ring;
            std::vector<int32_t>& dst = bucket[(size_t)to];
            if (!dst.empty() && dst.back() > b) mixed[(size_t)to] = 1;
            dst.push_back(b);
        }
        return giants;
    }
    // (load, bin): smallest load, then smallest bin.  A binary min-heap whose top is replaced in place -- one
    // sift-down per row instead of priority_queue's pop + push; the order of equal keys cannot matter, the keys
    // (load, bin) are distinct
    using Item = std::pair<int64_t, int32_t>;
    std::vector<Item> heap;
    heap.reserve((size_t)nbins);
    for (int32_t b = 0; b < nbins; ++b)
        if (!(b >= stride && stride > 0 && b % stride < giants)) heap.push_back({0, b});  // not a giant's tile-mate
    // (all loads 0, bins ascending: already a heap)
    const size_t hn = heap.size();
    for (int32_t x : idx) {
        Item t = heap[0];
        bin[(size_t)x] = t.second;
        t.first += deg[(size_t)x];
        size_t at = 0;
        for (;;) {
            size_t ch = 2 * at + 1;
            if (ch >= hn) break;
            if (ch + 1 < hn && heap[ch + 1] < heap[ch]) ++ch;
            if (!(heap[ch] < t)) break;
            heap[at] = heap[ch];
            at = ch;
        }
        heap[at] = t;
    }
    return giants;
}

// Marks the tiles whose cells all are a single chunk with exactly one item row (an item with a tile of its own,
// lpt_assign): the persistent kernel passes that row from workgroup to workgroup through a mailbox of
// self-validating {value, tag} granules instead of "store the tile, drain, flag" / "poll, gather" -- one
// memory round trip per hop instead of two and a drain, on the hop the epoch waits for.  Rows shorter than one
// wave of granules (fewer than 16 lanes per rating) are left alone.
// `tile_items[t]`: rated items of tile t -- one item row per cell is not enough, it has to be the SAME row in all of them.
void mark_lone_tiles(std::vector<CellDesc>& cells, int B, const Geometry& geo, const std::vector<int32_t>& tile_items) {
    if (geo.L < 16 || B < 2 || std::getenv("MFSGD_NO_MAILBOX")) return;  // (the variable: A/B measurements)
    for (int t = 0; t < B; ++t) {
        bool lone = tile_items[(size_t)t] == 1;
        for (int b = 0; b < B && lone; ++b) {
            const CellDesc& d = cells[(size_t)b * B + t];
            lone = d.next == 0 && d.ni == 1 && d.nu > 0 && (d.n_steps & 0x7FFFFFFFu) != 0;
        }
        if (lone)
            for (int b = 0; b < B; ++b) cells[(size_t)b * B + t].rsv[0] |= kCellLoneTile;
    }
}

struct Rat {
    uint16_t p, q;  // LDS slots
    float r;
    int64_t idx;    // caller-visible rating index
};

struct RawRat {
    uint32_t u, i;
    float r;
    int64_t idx;
    uint16_t sb;  // sub-cell inside the cell: sub-round * W + wave
};

// What a cell's (or chunk's) packer wrote, when it is in host memory; boxed, because most cells of a large rating set
// are packed on the device and hold none of it -- 590 K records of six empty vectors each were 118 MB to fault in and
// to give back at the Netflix shape.
struct CellArrays {
    std::vector<uint32_t> rows;
    std::vector<Entry> entries;
    std::vector<SubDesc> subs;
    std::vector<int64_t> order;
};

struct CellOut {
    std::unique_ptr<CellArrays> arr;
    CellArrays& a() {
        if (!arr) arr.reset(new CellArrays);
        return *arr;
    }
    const CellArrays& a() const {
        static const CellArrays none;
        return arr ? *arr : none;
    }
    uint32_t nu = 0, ni = 0, n_steps = 0;
    int64_t crit = 0;
    bool has_run = false;
    // sizes, valid also when there are no arrays because the DEVICE holds (or will write) the data
    uint32_t n_rows = 0;
    int64_t n_order = 0;
    bool dev = false;       // packed by the device packer: rows / entries / order are not here
    // [r3] a CHUNK the device packs (a part of a cell that was cut): its ratings are the range [part_lo, part_lo +
    // n_order) of build_schedule's `part_ratings` (indices into the caller's arrays, in the cell's bucket order, the
    // sub-cell of each beside it) -- the packing kernel takes the range as a cell of its own; its sub-cell table is
    // entry part_tab of `part_subs`
    bool dev_part = false;
    int64_t part_lo = -1, part_tab = -1;
    mutable int64_t desc = -1;  // its chunk descriptor, once placed
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

// A SOLO run: the ratings of ONE item whose users are all distinct, as a compact stream of 16-byte
// records (kernels.hip, run_asm.hpp): [header][record 0] ... [record n-1][terminator], padded to
// whole steps.  record t = {slots_{t+1}, mailbox (0xFFFFFFFF), lr * r_t, r_t}; header = {slots_0, 0,
// 0, 0}; slots = p-row address | q-row address << 16; the address behind the last step is a zero row.
// ([r3] word order: {slots, mailbox} is an aligned 8-byte unit, so the helper wave fetches the words of TWO
// steps with one ds_read2_b64 -- run_asm.hpp.)
// Every step decays with the same factor (no idle slots), so none is stored.
void pack_solo(const Rat* rs, int n, int G, int Lg, int nrows, const Hyper& hy, std::vector<Entry>& entries,
               std::vector<int64_t>& order, uint32_t& n_units) {
    auto words = [](uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        Entry e;
        const uint32_t w[4] = {a, b, c, d};
        std::memcpy(&e, w, sizeof e);
        return e;
    };
    auto bits = [](float f) {
        uint32_t x;
        std::memcpy(&x, &f, 4);
        return x;
    };
    const uint32_t q = rs[0].q;
    const uint32_t zero_slots = encode_slots(nrows, (int)q, false, Lg);
    auto slots_of = [&](int t) { return t < n ? encode_slots(rs[t].p, (int)q, false, Lg) : zero_slots; };
    entries.push_back(words(slots_of(0), 0u, 0u, 0u));
    for (int t = 0; t < n; ++t) {
        entries.push_back(words(slots_of(t + 1), 0xFFFFFFFFu, bits(hy.lr * rs[t].r), bits(rs[t].r)));
        order.push_back(rs[t].idx);
    }
    entries.push_back(words(zero_slots, 0xFFFFFFFFu, 0u, 0u));
    n_units = (uint32_t)((n + 2 + G - 1) / G);
    for (int x = n + 2; x < (int)n_units * G; ++x) entries.push_back(words(zero_slots, 0xFFFFFFFFu, 0u, 0u));
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
    std::vector<int64_t> degu, degi;
    bool on_device = prm.ingest && prm.ingest->degrees && prm.ingest->bucket;
    if (on_device && !prm.validated) {
        for (int64_t j = 0; j < n && on_device; ++j)
            if (u[j] < 0 || u[j] >= U || i[j] < 0 || i[j] >= I) on_device = false;  // let the host loop report it
    }
    if (prm.degu && prm.degi && prm.validated) {
        degu.assign(prm.degu, prm.degu + U);
        degi.assign(prm.degi, prm.degi + I);
    } else {
        degu.assign((size_t)U, 0);
        degi.assign((size_t)I, 0);
        if (on_device && prm.ingest->degrees(prm.ingest->ctx, u, i, n, U, I, degu.data(), degi.data()) != 0) {
            on_device = false;
            std::fill(degu.begin(), degu.end(), 0);
            std::fill(degi.begin(), degi.end(), 0);
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
    }
    lap(on_device ? "degrees (device)" : "degrees");
    std::vector<int32_t> ubin, ibin;
    // An item whose chain alone (count x cycles per dependent step) comes to 0.8 of the work-bound estimate
    // of the epoch (the model choose_geometry() picks W with) is on or near the critical path: lpt_assign
    // gives it a tile of its own.
    int64_t giant_min = 0;
    int giants = 0;  // items with a tile of their own: fine bins 0 .. giants - 1
    {
        const double np = (double)std::min<int64_t>(B, std::max(1, prm.n_cu));
        const double passes = std::ceil((double)B / (double)std::max(1, prm.n_cu));
        const double t_rest = 48.0 * (4.0 / geo.G) * (double)n / np + 12000.0 * (double)B * passes;
        giant_min = (int64_t)(0.8 * t_rest / (170.0 + 2.0 * geo.L));
    }
    {
        // users and items are independent: one thread each (the LPT itself is a sequential heap walk)
        std::exception_ptr failed_u;
        std::thread tu([&]() {
            try {
                lpt_assign(degu, B * W, ubin);
            } catch (...) {
                failed_u = std::current_exception();  // no exception may leave a thread
            }
        });
        try {
            giants = lpt_assign(degi, B * W, ibin, prm.lone_giants && !std::getenv("MFSGD_NO_LONE_GIANTS") ? B : 0,
                                giant_min);  // (the variable: A/B measurements)
        } catch (...) {
            tu.join();
            throw;
        }
        tu.join();
        if (failed_u) std::rethrow_exception(failed_u);
    }
    lap("LPT partition");
    // fine bin f -> block f % B, sub-group f / B
    std::vector<int32_t> tile_items((size_t)B, 0);  // rated items per tile (mark_lone_tiles)
    for (int32_t x = 0; x < I; ++x)
        if (degi[(size_t)x] > 0) tile_items[(size_t)(ibin[(size_t)x] % B)]++;

    // ---- counting sort by (cell, sub-round, wave) ---------------------------
    const int64_t nb = (int64_t)B * B * W * W;
    std::vector<int64_t> bptr;
    reserve_huge(bptr, (size_t)nb + 1);
    bptr.assign((size_t)nb + 1, 0);
    auto bucket_of = [&](int64_t j) -> int64_t {
        const int32_t fu = ubin[(size_t)u[j]], fi = ibin[(size_t)i[j]];
        // (an item with a tile of its own: all its ratings in the cell's first sub-cell -- the tile holds nothing
        // else, so nothing of the same users runs beside it, and its chain is one run instead of W)
        const int ub = fu % B, us = fi < giants ? 0 : fu / B, it = fi % B, is = fi / B;
        const int s = (is - us + W) % W;
        return (((int64_t)ub * B + it) * W + s) * W + us;
    };
    std::vector<int64_t> sorted;
    std::vector<uint32_t> sorted32;  // the same thing when it was fetched from the device packer's context
    bool want_dev_chunks = false;    // mixed mode with the chunks packed on the device: the bucket order stays there
    const DeviceIngestExt* ext = (on_device && prm.device_pack && prm.ingest->ext && prm.ingest->ext->bucket_dev &&
                                  prm.ingest->ext->pack_count && prm.ingest->ext->pack_emit && n > 0)
                                     ? prm.ingest->ext
                                     : nullptr;
    bool sorted_on_device = false;
    if (ext && ext->bucket_dev(prm.ingest->ctx, u, i, n, ubin.data(), ibin.data(), U, I, B, W, giants, bptr.data()) == 0)
        sorted_on_device = true;
    else
        ext = nullptr;
    if (!sorted_on_device) {
        sorted.resize((size_t)n);
        if (on_device && prm.ingest->bucket(prm.ingest->ctx, u, i, n, ubin.data(), ibin.data(), U, I, B, W, giants,
                                            bptr.data(), sorted.data()) != 0) {
            on_device = false;
            std::fill(bptr.begin(), bptr.end(), 0);
        }
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
    const int WW = W * W;
    const Hyper hy{prm.lr, 1.0f - prm.lr * prm.lambda};
    // solo runs pay where the kernel has the two-wave loops for them: 16+ lanes per rating (k > 32) and
    // a workgroup with copy waves (W <= 4); elsewhere a solo run would only be a slower run
    const bool solo_ok = prm.solo && geo.L >= 16 && W <= 4;
    const int64_t avail = (int64_t)prm.lds_budget - 16;
    const int64_t min_sched = sched_bytes_for(geo, W, 2, 3), min_rows = rows_bytes_for(geo, 2);
    auto addressable = [&](int nrows) { return (int64_t)(nrows + 2 * G) * geo.L <= 32767; };
    if (avail < 2 * min_sched + min_rows) {
        err = "lds: the LDS budget cannot hold a single rating at this k";
        return -1;
    }
    // ---- the device packer: every cell as a single chunk, same bytes as the host packer below -----
    std::vector<PackCellInfo> info;  // per cell, from the device's COUNT pass (kept for the mixed mode)
    std::vector<SubDesc> dsubs;
    bool have_info = false;
    // [r3] The sub-cell tables (W*W records per chunk descriptor) are written by the device's packer and read by the
    // training kernel: when the device packs the chunks of cut cells as well, every table is its, and the final array
    // is assembled THERE (emit: the cells' tables, the chunks' scattered to their descriptors) instead of coming down
    // with the COUNT results, through place() and up again with the schedule -- 75 MB three times at the Netflix
    // shape, 1.76 GB three times at 1 B ratings.  Schedule::subs stays empty; the debug getter fetches a copy.
    const bool dev_tables = ext && ext->download_raw && ext->pack_count_parts && ext->pack_emit_parts && ext->fetch_sorted_ranges &&
                            !std::getenv("MFSGD_HOST_CHUNKS") && !std::getenv("MFSGD_HOST_TABLES");
    if (sorted_on_device) {
        const char* why = "";
        const int rc = [&]() -> int {
            PackRequest q;
            q.u = u;
            q.i = i;
            q.r = r;
            q.orig = orig;
            q.n = n;
            q.U = U;
            q.I = I;
            q.ubin = ubin.data();
            q.ibin = ibin.data();
            q.B = B;
            q.W = W;
            q.G = G;
            q.L = geo.L;
            q.lr = hy.lr;
            q.c = hy.c;
            q.solo_ok = solo_ok;
            q.want_subs = !dev_tables;  // (the sub-cell tables stay on the device then: emit leaves the final one there)
            {
                // rows of the largest chunk the training kernel can hold beside the smallest schedule
                int64_t fit = (avail - 2 * min_sched) / geo.rowbytes - 2 * G;
                while (fit > 0 && !(addressable((int)fit) && rows_bytes_for(geo, (int)fit) + 2 * min_sched <= avail)) --fit;
                q.fit_rows = (int)std::max<int64_t>(fit, 0);
            }
            for (int64_t cc = 0; cc < ncell; ++cc)
                q.max_cell_nnz = std::max(q.max_cell_nnz, bptr[(size_t)((cc + 1) * WW)] - bptr[(size_t)(cc * WW)]);
            // canonical order: rounds, then blocks; a cell's ratings are contiguous.  Known from the bucket starts alone,
            // so the device's pass can write the order array -- and, into scratch, everything else -- while it counts
            std::vector<int64_t> ord_off((size_t)ncell);
            std::vector<int64_t> cell_ptr((size_t)ncell + 1, 0);
            int64_t pos = 0;
            for (int rd = 0; rd < B; ++rd)
                for (int b = 0; b < B; ++b) {
                    const int64_t cc = (int64_t)b * B + (b + rd) % B;
                    cell_ptr[(size_t)((int64_t)rd * B + b)] = pos;
                    ord_off[(size_t)cc] = pos;
                    pos += bptr[(size_t)((cc + 1) * WW)] - bptr[(size_t)(cc * WW)];
                }
            cell_ptr[(size_t)ncell] = pos;
            if (pos != n) return -1;
            q.ord_off = ord_off.data();
            int prc = ext->pack_count(prm.ingest->ctx, q, info, dsubs);
            if (prc != 0) {
                why = "the rating set is outside the kernel's limits (cell size, ranks, LDS)";
                return prc;
            }
            have_info = ext->pack_emit_mixed != nullptr;  // from here on a refusal means "mixed mode", not "host"
            lap("  device pack: count");
            // every cell must fit the training kernel's LDS image as ONE chunk (chunking is the host's job)
            int64_t max_s = min_sched, max_r = min_rows, tot_rows = 0, tot_steps = 0;
            std::vector<uint32_t> row_off((size_t)ncell), ent_off((size_t)ncell);
            for (int64_t cc = 0; cc < ncell; ++cc) {
                const PackCellInfo& ci = info[(size_t)cc];
                if (ci.status != 0) {
                    why = "a cell overflowed the kernel's arrays or counters";
                    return 1;
                }
                const int nrows = (int)(ci.nu + ci.ni);
                if (ci.n_steps != 0) {
                    if (!addressable(nrows) || rows_bytes_for(geo, nrows) + 2 * min_sched > avail) {
                        why = "a cell's rows exceed the training kernel's LDS image";
                        return 1;
                    }
                    max_s = std::max(max_s, sched_bytes_for(geo, W, nrows, (int64_t)ci.n_steps));
                    max_r = std::max(max_r, rows_bytes_for(geo, nrows));
                }
                if (tot_rows > 0xFFFFFFFFll - nrows || tot_steps > 0xFFFFFFFFll - (int64_t)ci.n_steps) return 1;
                row_off[(size_t)cc] = (uint32_t)tot_rows;
                ent_off[(size_t)cc] = (uint32_t)tot_steps;
                tot_rows += nrows;
                tot_steps += ci.n_steps;
            }
            if (2 * max_s + max_r > avail) {
                why = "cells have to be chunked";
                return 1;
            }
            if (ncell > 0x7FFFFFFFll / WW) return 1;
            Schedule sch;
            sch.geo = geo;
            sch.B = B;
            sch.W = W;
            sch.nnz = n;
            sch.cells.resize((size_t)ncell);
            sch.n_sub_recs = ncell * WW + 2;
            if (!dev_tables) {
                sch.subs.assign((size_t)(ncell * WW) + 2, SubDesc{0, 0});
                std::memcpy(sch.subs.data(), dsubs.data(), sizeof(SubDesc) * (size_t)(ncell * WW));
            }
            for (int64_t cc = 0; cc < ncell; ++cc) {
                const PackCellInfo& ci = info[(size_t)cc];
                CellDesc d{};
                d.row_off = row_off[(size_t)cc];
                d.ent_off = ent_off[(size_t)cc];
                d.n_steps = ci.n_steps | (ci.has_run ? kCellCritical : 0u);
                d.nu = (uint16_t)ci.nu;
                d.ni = (uint16_t)ci.ni;
                sch.cells[(size_t)cc] = d;
                const int64_t m_c = bptr[(size_t)((cc + 1) * WW)] - bptr[(size_t)(cc * WW)];
                sch.max_cell_nnz = std::max(sch.max_cell_nnz, m_c);
                sch.max_cell_rows = std::max<int64_t>(sch.max_cell_rows, ci.nu + ci.ni);
                sch.max_cell_steps = std::max<int64_t>(sch.max_cell_steps, ci.crit);
            }
            mark_lone_tiles(sch.cells, B, geo, tile_items);
            sch.sched_cap = (int)max_s;
            sch.lds_bytes = (int)((16 + 2 * max_s + max_r + 15) & ~(int64_t)15);
            sch.total_rows = tot_rows;
            sch.total_steps = tot_steps;
            sch.n_rows_words = tot_rows + 4;
            sch.n_entry_recs = tot_steps * G;
            sch.cell_ptr = std::move(cell_ptr);
            for (int rd = 0; rd < B; ++rd) {
                int64_t worst = 0;
                for (int b = 0; b < B; ++b) worst = std::max<int64_t>(worst, info[(size_t)((int64_t)b * B + (b + rd) % B)].crit);
                sch.sum_round_steps += worst;
            }
            lap("  device pack: offsets");
            prc = ext->pack_emit(prm.ingest->ctx, row_off.data(), ent_off.data(), ord_off.data(), tot_rows, tot_steps,
                                 dev_tables ? ncell : 0, &sch.dev.buf);
            if (prc != 0) return -1;
            lap("  device pack: emit");
            sch.device_packed = true;
            sch.dev_ops = ext;
            sch.device_ingest = true;
            out = std::move(sch);
            return 0;
        }();
        if (rc == 0) {
            out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
            return 0;
        }
        if (rc != 1) have_info = false;
        // the host packer needs the bucket order on this side (32-bit, as the device holds it) -- unless the device packs
        // the chunks too ([r3]): then only the cells that are cut come here, once it is known which
        want_dev_chunks = have_info && ext->pack_count_parts && ext->pack_emit_parts && ext->fetch_sorted_ranges &&
                          !std::getenv("MFSGD_HOST_CHUNKS");
        if (!want_dev_chunks) {
            sorted32.resize((size_t)n);
            if (!ext->fetch_sorted32 || ext->fetch_sorted32(prm.ingest->ctx, sorted32.data()) != 0) {
                err = "build_schedule: could not fetch the bucket order from the device";
                return -1;
            }
        }
        if (trace)
            std::fprintf(stderr, "[schedule]   device pack %s: %s%s\n", rc == 1 ? "declined" : "FAILED", why,
                         have_info ? " -> mixed mode: the device keeps the cells that fit, the host packs the rest" : "");
        lap("  bucket order to the host");
    }
    std::atomic<int> failed{0};
    std::string fail_msg;

    // One chunk: the ratings `sel` (bucket order, so sub-cells are contiguous) packed as a complete
    // little cell.  Returns false when a sub-cell overflows the 16-bit step counts (caller splits).
    auto pack_chunk = [&](const std::vector<RawRat>& sel, CellOut& o, Scratch& sc) -> bool {
        const int64_t m = (int64_t)sel.size();
        o = CellOut{};
        o.a().subs.assign((size_t)WW, SubDesc{0, 0});
        std::vector<uint32_t>& uu = sc.us;
        std::vector<uint32_t>& ii = sc.is;
        uu.resize((size_t)m);
        ii.resize((size_t)m);
        for (int64_t x = 0; x < m; ++x) {
            uu[(size_t)x] = sel[(size_t)x].u;
            ii[(size_t)x] = sel[(size_t)x].i;
        }
        std::sort(uu.begin(), uu.end());
        uu.erase(std::unique(uu.begin(), uu.end()), uu.end());
        std::sort(ii.begin(), ii.end());
        ii.erase(std::unique(ii.begin(), ii.end()), ii.end());
        const int nu = (int)uu.size(), ni = (int)ii.size(), nrows = nu + ni;
        o.nu = (uint32_t)nu;
        o.ni = (uint32_t)ni;
        o.a().rows.reserve((size_t)nrows);
        o.a().rows.insert(o.a().rows.end(), uu.begin(), uu.end());
        o.a().rows.insert(o.a().rows.end(), ii.begin(), ii.end());
        sc.rats.resize((size_t)m);
        int64_t sub_lo[65];  // first rating of sub-cell x (x = s * W + w), sub_lo[WW] = m
        {
            int nxt = 0;
            for (int64_t x = 0; x < m; ++x) {
                const RawRat& a = sel[(size_t)x];
                while (nxt <= (int)a.sb) sub_lo[nxt++] = x;
                Rat t;
                t.p = (uint16_t)(std::lower_bound(uu.begin(), uu.end(), a.u) - uu.begin());
                t.q = (uint16_t)(nu + (std::lower_bound(ii.begin(), ii.end(), a.i) - ii.begin()));
                t.r = a.r;
                t.idx = a.idx;
                sc.rats[(size_t)x] = t;
            }
            while (nxt <= WW) sub_lo[nxt++] = m;
        }
        sc.remdeg.assign((size_t)nrows, 0);
        sc.laststep.assign((size_t)nrows, 0);
        sc.lastslot.assign((size_t)nrows, (int8_t)-1);
        sc.prevstep.assign((size_t)nrows, 0);
        int32_t tstamp = 1;  // step stamps start at 2 so that t-1 never matches 0
        o.a().entries.reserve((size_t)(m + m / 2 + G));
        o.a().order.reserve((size_t)m);
        uint32_t stepcur = 0;
        int64_t crit = 0;
        for (int s = 0; s < W; ++s) {
            uint32_t smax = 0;
            for (int w = 0; w < W; ++w) {
                const int64_t slo = sub_lo[s * W + w], shi = sub_lo[s * W + w + 1];
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
                // A solo run for the heaviest run item when it dwarfs the others (their run would be
                // over in a quarter of its steps: taking it out costs less than it saves) and its
                // users are all distinct (a repeated (user, item) pair has to see its own update).
                bool solo = false;
                uint16_t solo_q = 0;
                if (solo_ok && nrun > 0) {
                    const int n1 = -sc.top[0].first, n2 = sc.top.size() > 1 ? -sc.top[1].first : 0;
                    if (n1 >= kSoloMin && 4 * n2 <= n1) {
                        solo = true;
                        solo_q = sc.top[0].second;
                        const int32_t stamp = ++tstamp;
                        for (int j = 0; j < nsub && solo; ++j)
                            if (sub[j].q == solo_q) {
                                if (sc.laststep[sub[j].p] == stamp) solo = false;
                                sc.laststep[sub[j].p] = stamp;
                            }
                    }
                    if (solo) {
                        for (int g = 1; g < nrun; ++g) run_q[g - 1] = run_q[g];
                        --nrun;
                    }
                }
                int ngen = nsub, nsolo = 0;
                if (solo) {  // solo ratings last
                    nsolo = nsub - (int)(std::stable_partition(sub, sub + nsub, [&](const Rat& x) { return x.q != solo_q; }) - sub);
                    ngen = nsub - nsolo;
                }
                const int nnon = ngen;  // general + run ratings
                if (nrun > 0) {
                    // stable partition: general ratings first, run ratings after
                    auto is_run = [&](const Rat& x) {
                        for (int g = 0; g < nrun; ++g)
                            if (run_q[g] == x.q) return true;
                        return false;
                    };
                    ngen = (int)(std::stable_partition(sub, sub + nnon, [&](const Rat& x) { return !is_run(x); }) - sub);
                }
                ++tstamp;  // break stickiness across sub-cells
                pack_subcell(sub, ngen, G, geo.L, nrows, hy, sc, tstamp, o.a().entries, o.a().order, ns);
                if (nrun > 0) {
                    ++tstamp;  // the run starts with fresh loads: no hazard against the last general step
                    pack_run(sub + ngen, nnon - ngen, run_q, nrun, G, geo.L, nrows, hy, sc, tstamp, o.a().entries, o.a().order,
                             nr);
                }
                uint32_t nsu = 0;  // step units the solo records occupy
                if (nsolo > 0) {
                    ++tstamp;
                    // the general and run loops read one and two steps ahead: two idle steps keep that
                    // look-ahead on step-format entries, whatever follows
                    for (int pad = 0; pad < kSoloPad; ++pad)
                        for (int g = 0; g < G; ++g)
                            o.a().entries.push_back(make_entry(encode_slots(nrows + 2 * g, nrows + 2 * g + 1, false, geo.L), 0.0f, hy.c, hy));
                    pack_solo(sub + nnon, nsolo, G, geo.L, nrows, hy, o.a().entries, o.a().order, nsu);
                    nsu += kSoloPad;
                }
                if (ns > 0xFFFF || nr > 0xFFFF || nsolo > 0xFFFF || stepcur > 0xFFFF) return false;
                if (nr > 0 || nsolo > 0) o.has_run = true;
                o.a().subs[(size_t)(s * W + w)] = SubDesc{stepcur | ((uint32_t)nsolo << 16), ns | (nr << 16)};
                stepcur += ns + nr + nsu;
                // a solo step costs about three quarters of a run step (137 against 184 cycles at L = 16)
                smax = std::max(smax, ns + nr + (uint32_t)(nsolo * 3 / 4));
            }
            crit += smax;
        }
        // two trailing idle steps: the kernel reads entries t+1 and t+2 ahead
        for (int pad = 0; pad < 2; ++pad)
            for (int g = 0; g < G; ++g)
                o.a().entries.push_back(make_entry(encode_slots(nrows + 2 * g, nrows + 2 * g + 1, false, geo.L), 0.0f, hy.c, hy));
        o.n_steps = stepcur + 2;
        o.crit = crit;
        o.n_rows = (uint32_t)nrows;
        o.n_order = m;
        return true;
    };
    auto load_cell = [&](int64_t c, std::vector<RawRat>& sel) {
        const int64_t lo = bptr[(size_t)(c * WW)], hi = bptr[(size_t)((c + 1) * WW)];
        sel.resize((size_t)(hi - lo));
        int sb = 0;
        for (int64_t x = lo; x < hi; ++x) {
            while (bptr[(size_t)(c * WW + sb + 1)] <= x) ++sb;
            const int64_t j = sorted32.empty() ? sorted[(size_t)x] : (int64_t)sorted32[(size_t)x];
            sel[(size_t)(x - lo)] = RawRat{(uint32_t)u[j], (uint32_t)i[j], r[j], orig ? orig[j] : j, (uint16_t)sb};
        }
    };
    auto distinct_rows = [&](const std::vector<RawRat>& sel, Scratch& sc, int& nu, int& ni) {
        sc.us.resize(sel.size());
        sc.is.resize(sel.size());
        for (size_t x = 0; x < sel.size(); ++x) {
            sc.us[x] = sel[x].u;
            sc.is[x] = sel[x].i;
        }
        std::sort(sc.us.begin(), sc.us.end());
        std::sort(sc.is.begin(), sc.is.end());
        nu = (int)(std::unique(sc.us.begin(), sc.us.end()) - sc.us.begin());
        ni = (int)(std::unique(sc.is.begin(), sc.is.end()) - sc.is.begin());
    };
    auto run_parallel = [&](const std::function<void(Scratch&, std::vector<RawRat>&)>& body) {
        std::vector<std::thread> th;
        auto w = [&]() {
            Scratch sc;
            std::vector<RawRat> sel;
            body(sc, sel);
        };
        const int nt = (int)std::min<int64_t>(nthreads, ncell);
        for (int t = 1; t < nt; ++t) th.emplace_back(w);
        w();
        for (auto& t : th) t.join();
    };

    // [r3] Mixed mode with the chunks packed on the device as well (pack_count_parts / pack_emit_parts): the host only
    // DECIDES the cuts -- from the cells' ids -- and asks the device for the size of every candidate chunk.
    // MFSGD_HOST_CHUNKS=1: round 2's mixed mode (the host packs and chunks what the device declines; A/B measurements).
    const bool dev_chunks = have_info && ext && want_dev_chunks;
    // Phase 1: every cell as a single chunk, unless it cannot possibly fit.
    std::vector<CellOut> co;
    reserve_huge(co, (size_t)ncell);
    co.resize((size_t)ncell);
    std::vector<uint8_t> oversize((size_t)ncell, 0);
    {
        std::atomic<int64_t> next_cell{0};
        // cells are claimed in batches: in mixed mode most of them only copy five numbers, and one atomic per cell
        // (590 K of them at the Netflix shape, contended by 16 threads) cost more than the work
        const int64_t batch = have_info ? 256 : 1;
        run_parallel([&](Scratch& sc, std::vector<RawRat>& sel) {
            for (;;) {
                const int64_t c0 = next_cell.fetch_add(batch);
                if (c0 >= ncell) break;
                for (int64_t c = c0; c < std::min(ncell, c0 + batch); ++c) {
                CellOut& o = co[(size_t)c];
                if (bptr[(size_t)(c * WW)] == bptr[(size_t)((c + 1) * WW)]) continue;  // (an empty cell: place() writes zeros)
                if (have_info) {
                    // mixed mode: a cell the device packed as one chunk stays there (only its sizes come here)
                    const PackCellInfo& ci = info[(size_t)c];
                    const int nrows = (int)(ci.nu + ci.ni);
                    if (ci.status == 0 && addressable(nrows) && rows_bytes_for(geo, nrows) + 2 * min_sched <= avail) {
                        // (its sub-cell table stays in dsubs: place() takes it from there -- 590 K small vectors less at
                        // the Netflix shape)
                        o.nu = ci.nu;
                        o.ni = ci.ni;
                        o.n_steps = ci.n_steps;
                        o.crit = ci.crit;
                        o.has_run = ci.has_run != 0;
                        o.n_rows = (uint32_t)nrows;
                        o.n_order = bptr[(size_t)((c + 1) * WW)] - bptr[(size_t)(c * WW)];
                        o.dev = true;
                        continue;
                    }
                }
                if (dev_chunks) {
                    // [r3] the device packs the chunks too: a cell it declined as ONE chunk (its rows exceed the LDS
                    // image, or its counters overflowed) is exactly a cell the host packer would call oversize
                    const PackCellInfo& ci = info[(size_t)c];
                    o = CellOut{};
                    o.nu = ci.nu;
                    o.ni = ci.ni;
                    o.n_order = bptr[(size_t)((c + 1) * WW)] - bptr[(size_t)(c * WW)];
                    oversize[(size_t)c] = 1;
                    continue;
                }
                load_cell(c, sel);
                int nu, ni;
                distinct_rows(sel, sc, nu, ni);
                if (!addressable(nu + ni) || rows_bytes_for(geo, nu + ni) + 2 * min_sched > avail ||
                    !pack_chunk(sel, o, sc)) {
                    o = CellOut{};
                    o.nu = (uint32_t)nu;  // kept for the limit search below
                    o.ni = (uint32_t)ni;
                    o.n_order = (int64_t)sel.size();
                    oversize[(size_t)c] = 1;
                }
                }
            }
        });
    }
    lap("per-cell packing");

    // Limits of one chunk: S bytes of schedule, R bytes of rows, 16 + 2 S + R <= budget.  Chosen to
    // cut as few cells as possible; when every cell fits as it is nothing is cut (and the caps below
    // are the actual maxima, not these limits).
    int64_t lim_s = 0, lim_r = 0;
    {
        int64_t max_s = min_sched, max_r = min_rows;
        bool any_over = false;
        for (int64_t c = 0; c < ncell; ++c) {
            const CellOut& o = co[(size_t)c];
            if (oversize[(size_t)c]) {
                any_over = true;
                continue;
            }
            if (o.n_steps == 0) continue;
            max_s = std::max(max_s, sched_bytes_for(geo, W, (int)(o.nu + o.ni), (int64_t)o.n_steps));
            max_r = std::max(max_r, rows_bytes_for(geo, (int)(o.nu + o.ni)));
        }
        if (!any_over && 2 * max_s + max_r <= avail) {
            lim_s = max_s;
            lim_r = max_r;
        } else {
            constexpr int kCand = 48;
            // per cell, once: the bytes it needs (an unpacked -- oversize -- cell: steps guessed from its rating count)
            std::vector<int64_t> need_s((size_t)ncell, 0), need_r((size_t)ncell, 0);
            int64_t max_need_s = max_s;  // the cells that will be cut count too: when ONLY they are large (an item with
                                         // a tile of its own), the candidates must not stop at the small cells' sizes
            for (int64_t c = 0; c < ncell; ++c) {
                const CellOut& o = co[(size_t)c];
                if (o.nu + o.ni == 0) continue;
                need_r[(size_t)c] = rows_bytes_for(geo, (int)(o.nu + o.ni));
                need_s[(size_t)c] = oversize[(size_t)c] ? sched_bytes_for(geo, W, (int)(o.nu + o.ni), o.n_order * 2 / G + 2)
                                                        : sched_bytes_for(geo, W, (int)(o.nu + o.ni), (int64_t)o.n_steps);
                max_need_s = std::max(max_need_s, need_s[(size_t)c]);
            }
            // two grids: up to the largest cell that fits as it is (where the optimum usually sits), and from there
            // up to the largest cell there is
            const int64_t s_mid = std::min(max_s, (avail - min_rows) / 2);
            const int64_t s_hi = std::min(max_need_s, (avail - min_rows) / 2);
            const int n_cand = s_hi > s_mid ? 2 * kCand : kCand;
            // the candidates are independent: one thread each, the winner (lowest cost, then lowest index) as before
            std::vector<double> cand_cost((size_t)n_cand + 1, 0.0);
            std::vector<int64_t> cand_s((size_t)n_cand + 1, 0), cand_r((size_t)n_cand + 1, 0);
            std::atomic<int> next_cand{0};
            auto eval = [&]() {
                for (;;) {
                    const int x = next_cand.fetch_add(1);
                    if (x > n_cand) break;
                    int64_t S = x <= kCand ? min_sched + (s_mid - min_sched) * x / kCand
                                           : s_mid + (s_hi - s_mid) * (x - kCand) / kCand;
                    S = (S + 15) & ~(int64_t)15;
                    if (S > (avail - min_rows) / 2) S = ((avail - min_rows) / 2) & ~(int64_t)15;
                    const int64_t R = (avail - 2 * S) & ~(int64_t)15;
                    double cost = 0;
                    for (int64_t c = 0; c < ncell; ++c) {
                        const int64_t rb = need_r[(size_t)c];
                        if (rb == 0) continue;
                        const int64_t sb = need_s[(size_t)c];
                        if (sb <= S && rb <= R && !oversize[(size_t)c]) continue;
                        cost += std::max(1.0, std::max((double)sb / (double)S, (double)rb / (double)R));
                    }
                    cand_cost[(size_t)x] = cost;
                    cand_s[(size_t)x] = S;
                    cand_r[(size_t)x] = R;
                }
            };
            {
                std::vector<std::thread> th;
                for (int t = 1; t < std::min(nthreads, n_cand + 1); ++t) th.emplace_back(eval);
                eval();
                for (auto& t : th) t.join();
            }
            double best_cost = -1;
            for (int x = 0; x <= n_cand; ++x)
                if (best_cost < 0 || cand_cost[(size_t)x] < best_cost) {
                    best_cost = cand_cost[(size_t)x];
                    lim_s = cand_s[(size_t)x];
                    lim_r = cand_r[(size_t)x];
                }
        }
    }
    lap("  chunk limits");

    // Phase 2: cells over the limits are cut in two (by users or by items, whichever there are more
    // of; halves balanced by rating count) until every piece fits.
    std::vector<std::vector<CellOut>> extra;
    reserve_huge(extra, (size_t)ncell);
    extra.resize((size_t)ncell);
    std::vector<int64_t> todo;
    for (int64_t c = 0; c < ncell; ++c) {
        const CellOut& o = co[(size_t)c];
        if (oversize[(size_t)c] ||
            (o.n_steps != 0 && (sched_bytes_for(geo, W, (int)(o.nu + o.ni), (int64_t)o.n_steps) > lim_s ||
                                rows_bytes_for(geo, (int)(o.nu + o.ni)) > lim_r)))
            todo.push_back(c);
    }
    // [r3] The ratings of the cells that are cut (indices into the caller's arrays), cell after cell in bucket order, and
    // the sub-cell of each.  The cut tree partitions a cell's range IN PLACE (stably), so every part -- and in the end
    // every chunk -- is a range of these two arrays, a cell's chunks lie in chain order one behind the other, and the
    // arrays as they stand are the rating lists the device's EMIT pass takes: no per-part vectors, nothing concatenated.
    std::vector<uint32_t> part_ratings;
    std::vector<uint16_t> part_sbs;
    std::vector<SubDesc> part_subs;  // W*W per chunk, in the order the chunks were accepted
    if (dev_chunks && !todo.empty()) {
        // The same cut tree as the host recursion below, grown level by level: a part whose rows fit is a candidate
        // and the device's COUNT pass says how many steps it packs into; a candidate within the limits is a leaf (a
        // chunk), everything else is cut at the same pivot the recursion would take.  A leaf's place among its cell's
        // chunks is its path in the tree (left before right) -- which is its place in the arrays.
        struct Part {
            int64_t lo, hi;  // its range of part_ratings
            int depth;
            int nu = 0, ni = 0;
            bool candidate = false;
            bool by_user = false;   // the cut this part gets if it is not a leaf ...
            uint32_t pivot = 0;     // ... ids >= pivot go right
        };
        struct Leaf {
            int64_t lo, hi;
            PackCellInfo ci;
            int64_t tab;  // its sub-cell table in part_subs
        };
        std::vector<Leaf> leaves;
        std::vector<Part> level;
        level.reserve(todo.size());
        // the bucket order of the cells to be cut, and of those only
        std::vector<int64_t> r_lo(todo.size()), r_len(todo.size()), r_at(todo.size() + 1, 0);
        for (size_t x = 0; x < todo.size(); ++x) {
            const int64_t c = todo[x];
            r_lo[x] = bptr[(size_t)(c * WW)];
            r_len[x] = bptr[(size_t)((c + 1) * WW)] - r_lo[x];
            r_at[x + 1] = r_at[x] + r_len[x];
        }
        const int64_t n_cut = r_at[todo.size()];
        reserve_huge(part_ratings, (size_t)n_cut);
        part_ratings.resize((size_t)n_cut);
        if (ext->fetch_sorted_ranges(prm.ingest->ctx, (int64_t)todo.size(), r_lo.data(), r_len.data(), part_ratings.data()) != 0) {
            err = "build_schedule: could not fetch the bucket order of the cells to be cut from the device";
            return -1;
        }
        lap("  bucket order of the cut cells to the host");
        reserve_huge(part_sbs, (size_t)n_cut);
        part_sbs.resize((size_t)n_cut);
        auto on_all_threads = [&](size_t n_items, size_t grain, const std::function<void(size_t, size_t)>& body) {
            std::atomic<size_t> nx{0};
            auto work = [&]() {
                for (;;) {
                    const size_t x0 = nx.fetch_add(grain);
                    if (x0 >= n_items) break;
                    body(x0, std::min(n_items, x0 + grain));
                }
            };
            std::vector<std::thread> th;
            const size_t want = (n_items + grain - 1) / grain;
            for (int t = 1; t < (int)std::min<size_t>((size_t)nthreads, want); ++t) th.emplace_back(work);
            work();
            for (auto& t : th) t.join();
        };
        on_all_threads(todo.size(), 16, [&](size_t x0, size_t x1) {
            for (size_t x = x0; x < x1; ++x) {
                const int64_t c = todo[x];
                const int64_t lo = r_lo[x], hi = lo + r_len[x];
                int sbi = 0;
                for (int64_t y = lo; y < hi; ++y) {
                    while (bptr[(size_t)(c * WW + sbi + 1)] <= y) ++sbi;
                    part_sbs[(size_t)(r_at[x] + (y - lo))] = (uint16_t)sbi;
                }
            }
        });
        for (size_t x = 0; x < todo.size(); ++x) {
            Part p;
            p.lo = r_at[x];
            p.hi = r_at[x + 1];
            p.depth = 0;
            level.push_back(p);
        }
        bool root = true;
        double t_rows = 0, t_lists = 0, t_count = 0, t_cut = 0;  // MFSGD_SCHED_TRACE: where the levels' time goes
        int n_levels = 0;
        auto since = [](std::chrono::steady_clock::time_point& t) {
            const auto now = std::chrono::steady_clock::now();
            const double d = std::chrono::duration<double>(now - t).count();
            t = now;
            return d;
        };
        auto t_lvl = std::chrono::steady_clock::now();
        while (!level.empty() && !failed.load()) {
            ++n_levels;
            (void)since(t_lvl);
            // 1. distinct rows of every part; whole cells (the roots) are known not to fit: they are cut unseen
            on_all_threads(level.size(), 1, [&](size_t x0, size_t x1) {
                thread_local std::vector<uint32_t> us, is;
                for (size_t x = x0; x < x1; ++x) {
                    Part& p = level[x];
                    const size_t m = (size_t)(p.hi - p.lo);
                    us.resize(m);
                    is.resize(m);
                    for (size_t y = 0; y < m; ++y) {
                        const uint32_t j = part_ratings[(size_t)p.lo + y];
                        us[y] = (uint32_t)u[j];
                        is[y] = (uint32_t)i[j];
                    }
                    std::sort(us.begin(), us.end());
                    std::sort(is.begin(), is.end());
                    auto distinct = [](const std::vector<uint32_t>& v) {
                        int n = 0;
                        for (size_t y = 0; y < v.size(); ++y) n += y == 0 || v[y] != v[y - 1];
                        return n;
                    };
                    p.nu = distinct(us);
                    p.ni = distinct(is);
                    const int nrows = p.nu + p.ni;
                    p.candidate = !root && addressable(nrows) && rows_bytes_for(geo, nrows) <= lim_r;
                    // the pivot of the cut, should there be one: the first distinct id (position >= 1) at which the
                    // ratings of the ids before it reach half of the part (the host recursion's rule)
                    p.by_user = (p.nu >= p.ni && p.nu > 1) || p.ni <= 1;
                    const std::vector<uint32_t>& v = p.by_user ? us : is;
                    const int nid = p.by_user ? p.nu : p.ni;
                    const int64_t half = (int64_t)v.size() / 2;
                    int seen = 0;  // distinct ids passed
                    p.pivot = v.empty() ? 0u : v[0];
                    for (size_t y = 0; y < v.size(); ++y) {
                        if (y > 0 && v[y] != v[y - 1]) {
                            // y ratings belong to the `seen + 1` ids before v[y]
                            ++seen;
                            p.pivot = v[y];
                            if ((int64_t)y >= half || seen >= nid - 1) break;
                        }
                    }
                }
            });
            t_rows += since(t_lvl);
            // 2. the candidates' sizes, from the device
            std::vector<size_t> cand;
            for (size_t x = 0; x < level.size(); ++x)
                if (level[x].candidate) cand.push_back(x);
            std::vector<PackCellInfo> pinfo(cand.size());
            std::vector<SubDesc> psubs(dev_tables ? 0 : cand.size() * (size_t)WW);  // (device tables: not fetched)
            if (!cand.empty()) {
                std::vector<int64_t> c_at(cand.size() + 1, 0);
                for (size_t y = 0; y < cand.size(); ++y) c_at[y + 1] = c_at[y] + (level[cand[y]].hi - level[cand[y]].lo);
                std::vector<uint32_t> lst;
                reserve_huge(lst, (size_t)c_at[cand.size()]);
                lst.resize((size_t)c_at[cand.size()]);
                std::vector<int64_t> cptr(cand.size() * (size_t)WW + 1, 0);
                on_all_threads(cand.size(), 64, [&](size_t y0, size_t y1) {
                    for (size_t y = y0; y < y1; ++y) {
                        const Part& p = level[cand[y]];
                        const size_t m = (size_t)(p.hi - p.lo);
                        std::memcpy(&lst[(size_t)c_at[y]], &part_ratings[(size_t)p.lo], m * sizeof(uint32_t));
                        const uint16_t* sb = &part_sbs[(size_t)p.lo];
                        size_t at = 0;
                        for (int sbi = 0; sbi < WW; ++sbi) {
                            cptr[y * (size_t)WW + (size_t)sbi] = c_at[y] + (int64_t)at;
                            while (at < m && sb[at] == (uint16_t)sbi) ++at;
                        }
                    }
                });
                cptr[cand.size() * (size_t)WW] = c_at[cand.size()];
                t_lists += since(t_lvl);
                if (ext->pack_count_parts(prm.ingest->ctx, (int64_t)cand.size(), lst.data(), (int64_t)lst.size(), cptr.data(),
                                          pinfo.data(), dev_tables ? nullptr : psubs.data()) != 0) {
                    err = "build_schedule: the device packer's COUNT pass over the chunks failed";
                    return -1;
                }
            }
            t_count += since(t_lvl);
            // 3. leaves and cuts
            std::vector<Part> next;
            std::vector<uint8_t> is_leaf(level.size(), 0);
            for (size_t y = 0; y < cand.size(); ++y) {
                const Part& p = level[cand[y]];
                const PackCellInfo& ci = pinfo[y];
                if (ci.status == 0 && sched_bytes_for(geo, W, p.nu + p.ni, (int64_t)ci.n_steps) <= lim_s) {
                    Leaf lf;
                    lf.lo = p.lo;
                    lf.hi = p.hi;
                    lf.ci = ci;
                    lf.tab = (int64_t)(part_subs.size() / (size_t)WW);
                    if (!dev_tables)
                        part_subs.insert(part_subs.end(), psubs.begin() + (long)(y * (size_t)WW), psubs.begin() + (long)((y + 1) * (size_t)WW));
                    leaves.push_back(lf);
                    is_leaf[cand[y]] = 1;
                }
            }
            {
                std::mutex mu;
                on_all_threads(level.size(), 4, [&](size_t x0, size_t x1) {
                    thread_local std::vector<uint32_t> hold_idx;
                    thread_local std::vector<uint16_t> hold_sb;
                    std::vector<Part> mine;
                    for (size_t x = x0; x < x1 && !failed.load(); ++x) {
                        if (is_leaf[x]) continue;
                        const Part& p = level[x];
                        const int64_t m = p.hi - p.lo;
                        if (m <= 1) {
                            if (!failed.exchange(1)) fail_msg = "lds: a single rating does not fit the chunk limits";
                            break;
                        }
                        if (p.depth >= 62) {
                            if (!failed.exchange(1)) fail_msg = "build_schedule: a cell was cut more than 62 times";
                            break;
                        }
                        int64_t nl;
                        if (p.nu <= 1 && p.ni <= 1) {
                            nl = m / 2;  // the same (user, item) pair many times over: any cut of the sequence will do
                        } else {
                            // stable partition of the range in place: the lefts close up (a write never passes the
                            // read position), the rights wait in a buffer and follow
                            const int32_t* key_of = p.by_user ? u : i;
                            const uint32_t pivot = p.pivot;
                            uint32_t* idx = &part_ratings[(size_t)p.lo];
                            uint16_t* sb = &part_sbs[(size_t)p.lo];
                            hold_idx.clear();
                            hold_sb.clear();
                            nl = 0;
                            for (int64_t y = 0; y < m; ++y) {
                                if ((uint32_t)key_of[idx[y]] < pivot) {
                                    idx[nl] = idx[y];
                                    sb[nl] = sb[y];
                                    ++nl;
                                } else {
                                    hold_idx.push_back(idx[y]);
                                    hold_sb.push_back(sb[y]);
                                }
                            }
                            std::memcpy(idx + nl, hold_idx.data(), hold_idx.size() * sizeof(uint32_t));
                            std::memcpy(sb + nl, hold_sb.data(), hold_sb.size() * sizeof(uint16_t));
                        }
                        Part l, r2;
                        l.depth = r2.depth = p.depth + 1;
                        l.lo = p.lo;
                        l.hi = r2.lo = p.lo + nl;
                        r2.hi = p.hi;
                        mine.push_back(l);
                        mine.push_back(r2);
                    }
                    std::lock_guard<std::mutex> lk(mu);
                    next.insert(next.end(), mine.begin(), mine.end());
                });
            }
            level = std::move(next);
            root = false;
            t_cut += since(t_lvl);
        }
        if (trace)
            std::fprintf(stderr, "[schedule]     %d levels: rows + pivots %.3f s, candidate lists %.3f s, device COUNT %.3f s, leaves + cuts %.3f s\n",
                         n_levels, t_rows, t_lists, t_count, t_cut);
        if (!failed.load()) {
            // a cell's chunks in chain order = its leaves by position
            std::sort(leaves.begin(), leaves.end(), [](const Leaf& a, const Leaf& b) { return a.lo < b.lo; });
            size_t at = 0;
            for (size_t x = 0; x < todo.size() && !failed.load(); ++x) {
                size_t end = at;
                while (end < leaves.size() && leaves[end].lo < r_at[x + 1]) ++end;
                // (the leaves of a cell tile its range: first at its start, each at the end of the one before, last at its end)
                bool tiles = end > at && leaves[at].lo == r_at[x] && leaves[end - 1].hi == r_at[x + 1];
                for (size_t y = at + 1; tiles && y < end; ++y) tiles = leaves[y].lo == leaves[y - 1].hi;
                if (!tiles) {
                    if (!failed.exchange(1)) fail_msg = "build_schedule: internal error, a cut cell's chunks do not tile its ratings";
                    break;
                }
                const int64_t c = todo[x];
                auto chunk_of = [&](const Leaf& lf) {
                    CellOut o;
                    o.nu = lf.ci.nu;
                    o.ni = lf.ci.ni;
                    o.n_steps = lf.ci.n_steps;
                    o.crit = lf.ci.crit;
                    o.has_run = lf.ci.has_run != 0;
                    o.n_rows = lf.ci.nu + lf.ci.ni;
                    o.n_order = lf.hi - lf.lo;
                    o.dev_part = true;
                    o.part_lo = lf.lo;
                    o.part_tab = lf.tab;
                    return o;
                };
                co[(size_t)c] = chunk_of(leaves[at]);
                extra[(size_t)c].reserve(end - at - 1);
                for (size_t y = at + 1; y < end; ++y) extra[(size_t)c].push_back(chunk_of(leaves[y]));
                at = end;
            }
        }
    } else {
        std::atomic<int64_t> next_todo{0};
        run_parallel([&](Scratch& sc, std::vector<RawRat>& sel) {
            std::vector<CellOut> pieces;
            std::function<void(std::vector<RawRat>&)> cut = [&](std::vector<RawRat>& part) {
                if (failed.load()) return;
                int nu, ni;
                distinct_rows(part, sc, nu, ni);
                const int nrows = nu + ni;
                if (addressable(nrows) && rows_bytes_for(geo, nrows) <= lim_r) {
                    CellOut o;
                    if (pack_chunk(part, o, sc) &&
                        sched_bytes_for(geo, W, nrows, (int64_t)o.n_steps) <= lim_s) {
                        pieces.push_back(std::move(o));
                        return;
                    }
                }
                if (part.size() <= 1) {
                    if (!failed.exchange(1)) fail_msg = "lds: a single rating does not fit the chunk limits";
                    return;
                }
                // (distinct_rows left the sorted distinct ids in sc.us / sc.is; pack_chunk may have
                // overwritten them, so recompute)
                distinct_rows(part, sc, nu, ni);
                if (nu <= 1 && ni <= 1) {
                    // the same (user, item) pair many times over: any cut of the sequence will do
                    std::vector<RawRat> left(part.begin(), part.begin() + (long)(part.size() / 2));
                    std::vector<RawRat> right(part.begin() + (long)(part.size() / 2), part.end());
                    std::vector<RawRat>().swap(part);
                    cut(left);
                    cut(right);
                    return;
                }
                const bool by_user = (nu >= ni && nu > 1) || ni <= 1;
                const std::vector<uint32_t>& ids = by_user ? sc.us : sc.is;
                const int nid = by_user ? nu : ni;
                std::vector<int64_t> cnt((size_t)nid, 0);
                for (const RawRat& a : part) {
                    const uint32_t key = by_user ? a.u : a.i;
                    cnt[(size_t)(std::lower_bound(ids.begin(), ids.begin() + nid, key) - ids.begin())]++;
                }
                int64_t half = (int64_t)part.size() / 2, acc = 0;
                int cutpos = 1;
                for (int x = 0; x < nid - 1; ++x) {
                    acc += cnt[(size_t)x];
                    cutpos = x + 1;
                    if (acc >= half) break;
                }
                const uint32_t pivot = ids[(size_t)cutpos];  // ids >= pivot go right
                std::vector<RawRat> left, right;
                for (const RawRat& a : part) ((by_user ? a.u : a.i) < pivot ? left : right).push_back(a);
                std::vector<RawRat>().swap(part);
                cut(left);
                cut(right);
            };
            for (;;) {
                const int64_t x = next_todo.fetch_add(1);
                if (x >= (int64_t)todo.size() || failed.load()) break;
                const int64_t c = todo[(size_t)x];
                load_cell(c, sel);
                pieces.clear();
                cut(sel);
                if (failed.load() || pieces.empty()) break;
                co[(size_t)c] = std::move(pieces[0]);
                extra[(size_t)c].assign(std::make_move_iterator(pieces.begin() + 1),
                                        std::make_move_iterator(pieces.end()));
            }
        });
    }
    if (failed.load()) {
        err = fail_msg;
        return -1;
    }
    lap("  chunking");

    // ---- concatenate: first chunks at their cell index, the rest behind B*B ---
    out = Schedule{};
    out.geo = geo;
    out.B = B;
    out.W = W;
    out.nnz = n;
    int64_t n_descs = ncell;
    for (int64_t c = 0; c < ncell; ++c) n_descs += (int64_t)extra[(size_t)c].size();
    if (n_descs > 0x7FFFFFFFll / WW) {
        err = "build_schedule: too many chunks";
        return -1;
    }
    reserve_huge(out.cells, (size_t)n_descs);
    out.cells.resize((size_t)n_descs);
    // (with the chunks packed on the device the sub-cell tables are assembled there: dev_tables above)
    const bool host_tables = !(dev_tables && dev_chunks);
    out.n_sub_recs = n_descs * WW + 2;
    if (host_tables) {
        reserve_huge(out.subs, (size_t)(n_descs * WW) + 2);
        out.subs.resize((size_t)(n_descs * WW) + 2);  // +16 B: the staging DMA reads whole 16-byte units
        out.subs[(size_t)(n_descs * WW)] = out.subs[(size_t)(n_descs * WW) + 1] = SubDesc{0, 0};
    }
    std::vector<const CellOut*> by_desc((size_t)n_descs, nullptr);
    int64_t tot_rows = 0, tot_steps = 0;
    int64_t sched_cap = 0, rows_cap = 0;
    int64_t n_dev_cells = 0, n_dev_parts = 0;
    // per cell, compact (the loops over rounds below walk the cells with a stride of B + 1: these stay in cache,
    // the CellOut records do not): ratings and critical steps of all its chunks; the cells that are more than one
    // device-packed chunk, whose pieces have to be walked one by one
    std::vector<int64_t> cell_nnz((size_t)ncell), cell_crit((size_t)ncell);
    std::vector<int64_t> walk_cells;
    {
        // pass 1, sequential and light: descriptor numbers, the chain of a cell's chunks, running offsets
        int64_t next_desc = ncell;
        auto place = [&](int64_t d, const CellOut& o, uint32_t next) -> bool {
            if (tot_rows > 0xFFFFFFFFll - (int64_t)o.n_rows || tot_steps > 0xFFFFFFFFll - o.n_steps) return false;
            CellDesc& cdsc = out.cells[(size_t)d];
            cdsc = CellDesc{};
            cdsc.row_off = (uint32_t)tot_rows;
            cdsc.ent_off = (uint32_t)tot_steps;
            cdsc.next = next;
            by_desc[(size_t)d] = &o;
            tot_rows += (int64_t)o.n_rows;
            tot_steps += o.n_steps;
            n_dev_cells += o.dev ? 1 : 0;
            n_dev_parts += o.dev_part ? 1 : 0;
            return true;
        };
        for (int64_t c = 0; c < ncell; ++c) {
            const std::vector<CellOut>& ex = extra[(size_t)c];
            bool ok = place(c, co[(size_t)c], ex.empty() ? 0u : (uint32_t)next_desc);
            int64_t nnz_c = co[(size_t)c].n_order, rows_c = (int64_t)co[(size_t)c].n_rows;
            int64_t crit_c = co[(size_t)c].crit;
            for (size_t x = 0; ok && x < ex.size(); ++x) {
                ok = place(next_desc, ex[x], x + 1 < ex.size() ? (uint32_t)(next_desc + 1) : 0u);
                ++next_desc;
                nnz_c += ex[x].n_order;
                rows_c = std::max(rows_c, (int64_t)ex[x].n_rows);
                crit_c += ex[x].crit;
            }
            if (!ok) {
                err = "build_schedule: schedule exceeds 32-bit offsets";
                return -1;
            }
            if (!ex.empty()) out.split_cells++;
            if (!ex.empty() || !co[(size_t)c].dev) walk_cells.push_back(c);
            cell_nnz[(size_t)c] = nnz_c;
            cell_crit[(size_t)c] = crit_c;
            out.max_cell_nnz = std::max(out.max_cell_nnz, nnz_c);
            out.max_cell_rows = std::max(out.max_cell_rows, rows_c);
            out.max_cell_steps = std::max(out.max_cell_steps, crit_c);
        }
        // pass 2, parallel over the descriptors: the rest of each descriptor, its sub-cell table, the LDS capacities
        std::atomic<int64_t> nx{0};
        std::mutex mx;
        auto fill = [&]() {
            int64_t my_sched = 0, my_rows = 0;
            for (;;) {
                const int64_t d0 = nx.fetch_add(4096);
                if (d0 >= n_descs) break;
                for (int64_t d = d0; d < std::min(n_descs, d0 + 4096); ++d) {
                    const CellOut& o = *by_desc[(size_t)d];
                    CellDesc& cdsc = out.cells[(size_t)d];
                    cdsc.n_steps = o.n_steps | (o.has_run ? kCellCritical : 0u);
                    cdsc.nu = (uint16_t)o.nu;
                    cdsc.ni = (uint16_t)o.ni;
                    o.desc = d;
                    if (!host_tables) {
                        // (the device assembles the table)
                    } else if (o.dev && o.a().subs.empty() && d < ncell)
                        std::memcpy(&out.subs[(size_t)(d * WW)], &dsubs[(size_t)(d * WW)], sizeof(SubDesc) * (size_t)WW);
                    else if (o.dev_part)
                        std::memcpy(&out.subs[(size_t)(d * WW)], &part_subs[(size_t)(o.part_tab * WW)], sizeof(SubDesc) * (size_t)WW);
                    else
                        for (int x = 0; x < WW; ++x)
                            out.subs[(size_t)(d * WW + x)] = o.a().subs.empty() ? SubDesc{0, 0} : o.a().subs[(size_t)x];
                    if (o.n_steps != 0) {
                        my_sched = std::max(my_sched, sched_bytes_for(geo, W, (int)(o.nu + o.ni), (int64_t)o.n_steps));
                        my_rows = std::max(my_rows, rows_bytes_for(geo, (int)(o.nu + o.ni)));
                    }
                }
            }
            std::lock_guard<std::mutex> lk(mx);
            sched_cap = std::max(sched_cap, my_sched);
            rows_cap = std::max(rows_cap, my_rows);
        };
        std::vector<std::thread> th;
        const int nt = n_descs >= 65536 ? nthreads : 1;
        for (int t = 1; t < nt; ++t) th.emplace_back(fill);
        fill();
        for (auto& t : th) t.join();
    }
    mark_lone_tiles(out.cells, B, geo, tile_items);
    lap("  offsets");
    {
        sched_cap = std::max(sched_cap, min_sched);
        rows_cap = std::max(rows_cap, min_rows);
        const int64_t need = 16 + 2 * sched_cap + rows_cap;
        if (need > prm.lds_budget) {
            err = "build_schedule: internal error, chunks need " + std::to_string(need) + " bytes of LDS (budget " +
                  std::to_string(prm.lds_budget) + ")";
            return -1;
        }
        out.lds_bytes = (int)((need + 15) & ~(int64_t)15);
        out.sched_cap = (int)sched_cap;
    }
    out.total_rows = tot_rows;
    out.total_steps = tot_steps;
    // canonical order: rounds, then blocks; a cell's ratings (all its chunks) are contiguous
    out.cell_ptr.assign((size_t)ncell + 1, 0);
    int64_t pos = 0;
    for (int rd = 0; rd < B; ++rd) {
        int64_t worst = 0;
        for (int b = 0; b < B; ++b) {
            const int64_t c = (int64_t)b * B + (b + rd) % B;
            out.cell_ptr[(size_t)((int64_t)rd * B + b)] = pos;
            pos += cell_nnz[(size_t)c];
            worst = std::max(worst, cell_crit[(size_t)c]);
        }
        out.sum_round_steps += worst;
    }
    out.cell_ptr[(size_t)ncell] = pos;
    if (pos != n) {
        err = "build_schedule: internal error, packed " + std::to_string(pos) + " of " + std::to_string(n);
        return -1;
    }
    if (n_dev_cells > 0 || n_dev_parts > 0) {
        // ---- mixed finish: the device writes its cells at their final places (EMIT pass); the chunks of the cells that
        // were cut are packed there too ([r3], from their rating lists), or -- round 2's form, MFSGD_HOST_CHUNKS -- what
        // the host packed is scattered behind it
        MixedPieces mp;
        std::vector<uint32_t> row_off((size_t)ncell, 0xFFFFFFFFu), ent_off((size_t)ncell, 0u);
        std::vector<int64_t> ord_off((size_t)ncell, 0);
        std::vector<uint32_t> p_ro, p_eo;
        std::vector<int64_t> p_cptr, p_oo, p_desc;  // (p_desc: the chunk descriptor of every part, for the device's final sub-cell table)
        // (device cells are first chunks, x < ncell, and most of all descriptors: they are skipped through the compact
        // list of the cells that are anything else)
        for (int64_t x = 0; x < ncell; ++x) {
            row_off[(size_t)x] = out.cells[(size_t)x].row_off;
            ent_off[(size_t)x] = out.cells[(size_t)x].ent_off;
        }
        for (int64_t x = 0; x < ncell; ++x) {
            const int64_t rd = x / B, b = x % B;
            ord_off[(size_t)(b * B + (b + rd) % B)] = out.cell_ptr[(size_t)x];
        }
        std::vector<int64_t> other_descs;
        for (const int64_t c : walk_cells) {
            if (!co[(size_t)c].dev) {
                row_off[(size_t)c] = 0xFFFFFFFFu;
                ent_off[(size_t)c] = 0u;
                other_descs.push_back(c);
            }
            for (const CellOut& o : extra[(size_t)c]) other_descs.push_back(o.desc);
        }
        std::sort(other_descs.begin(), other_descs.end());
        for (const int64_t x : other_descs) {
            const CellOut& o = *by_desc[(size_t)x];
            const CellDesc& d = out.cells[(size_t)x];
            if (o.dev || o.dev_part) continue;  // (the device's chunks: below, in the order of their rating lists)
            if (!o.a().rows.empty()) {
                mp.seg_rows.push_back({(uint64_t)d.row_off, (uint64_t)mp.rows.size(), (uint64_t)o.a().rows.size()});
                mp.rows.insert(mp.rows.end(), o.a().rows.begin(), o.a().rows.end());
            }
            if (!o.a().entries.empty()) {
                mp.seg_entries.push_back({(uint64_t)d.ent_off * G, (uint64_t)mp.entries.size(), (uint64_t)o.a().entries.size()});
                mp.entries.insert(mp.entries.end(), o.a().entries.begin(), o.a().entries.end());
            }
        }
        if (n_dev_parts > 0) {
            // The chunks the device packs, in the order of their ranges of part_ratings (cut cell after cut cell, a
            // cell's chunks in chain order): where each goes, and its W*W sub-cell starts -- independent per chunk.
            // A chunk's place in the canonical order is its cell's plus the ratings of the chunks before it, which is
            // its distance from the first chunk in part_ratings.
            std::vector<const CellOut*> parts;
            parts.reserve((size_t)n_dev_parts);
            std::vector<int64_t> part_cell;
            part_cell.reserve((size_t)n_dev_parts);
            for (const int64_t c : walk_cells) {
                if (co[(size_t)c].dev_part) {
                    parts.push_back(&co[(size_t)c]);
                    part_cell.push_back(c);
                }
                for (const CellOut& o : extra[(size_t)c])
                    if (o.dev_part) {
                        parts.push_back(&o);
                        part_cell.push_back(c);
                    }
            }
            bool in_order = (int64_t)parts.size() == n_dev_parts && !parts.empty() && parts[0]->part_lo == 0;
            for (size_t y = 1; in_order && y < parts.size(); ++y) in_order = parts[y]->part_lo == parts[y - 1]->part_lo + parts[y - 1]->n_order;
            if (!in_order || parts.back()->part_lo + parts.back()->n_order != (int64_t)part_ratings.size()) {
                err = "build_schedule: internal error, the device-packed chunks do not tile their rating list";
                return -1;
            }
            p_ro.resize(parts.size());
            p_eo.resize(parts.size());
            p_oo.resize(parts.size());
            p_desc.resize(parts.size());
            p_cptr.assign(parts.size() * (size_t)WW + 1, 0);
            std::atomic<size_t> nx{0};
            auto fill = [&]() {
                for (;;) {
                    const size_t y0 = nx.fetch_add(256);
                    if (y0 >= parts.size()) break;
                    for (size_t y = y0; y < std::min(parts.size(), y0 + 256); ++y) {
                        const CellOut& o = *parts[y];
                        const CellDesc& d = out.cells[(size_t)o.desc];
                        const int64_t c = part_cell[y];
                        p_ro[y] = d.row_off;
                        p_eo[y] = d.ent_off;
                        p_desc[y] = o.desc;
                        p_oo[y] = ord_off[(size_t)c] + (o.part_lo - co[(size_t)c].part_lo);
                        const uint16_t* sb = &part_sbs[(size_t)o.part_lo];
                        const size_t m = (size_t)o.n_order;
                        size_t at = 0;
                        for (int sbi = 0; sbi < WW; ++sbi) {
                            p_cptr[y * (size_t)WW + (size_t)sbi] = o.part_lo + (int64_t)at;
                            while (at < m && sb[at] == (uint16_t)sbi) ++at;
                        }
                    }
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < (parts.size() >= 4096 ? nthreads : 1); ++t) th.emplace_back(fill);
            fill();
            for (auto& t : th) t.join();
            p_cptr[parts.size() * (size_t)WW] = (int64_t)part_ratings.size();
        }
        for (const int64_t c : walk_cells) {
            int64_t at = ord_off[(size_t)c];
            auto piece = [&](const CellOut& o) {
                if (!o.dev && !o.dev_part && !o.a().order.empty()) {
                    mp.seg_order.push_back({(uint64_t)at, (uint64_t)mp.order.size(), (uint64_t)o.a().order.size()});
                    mp.order.insert(mp.order.end(), o.a().order.begin(), o.a().order.end());
                }
                at += o.n_order;
            };
            piece(co[(size_t)c]);
            for (const CellOut& o : extra[(size_t)c]) piece(o);
        }
        if (n_dev_parts > 0 || !host_tables) {
            if (!mp.rows.empty() || !mp.entries.empty() || !mp.order.empty()) {
                err = "build_schedule: internal error, host-packed pieces beside device-packed chunks";
                return -1;
            }
            lap("  mixed: lists of the chunks");
            if (ext->pack_emit_parts(prm.ingest->ctx, row_off.data(), ent_off.data(), ord_off.data(), tot_rows, tot_steps,
                                     (int64_t)p_ro.size(), part_ratings.data(), (int64_t)part_ratings.size(), p_cptr.data(), p_ro.data(),
                                     p_eo.data(), p_oo.data(), host_tables ? 0 : n_descs, p_desc.data(), &out.dev.buf) != 0) {
                err = "build_schedule: the device packer's EMIT pass (cells and chunks) failed";
                return -1;
            }
            lap("  mixed: emit (cells + chunks)");
            out.device_packed = true;
            out.dev_ops = ext;
            out.device_ingest = true;
            out.n_rows_words = tot_rows + 4;
            out.n_entry_recs = tot_steps * G;
            out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
            return 0;
        }
        lap("  mixed: staging of the host-packed pieces");
        if (ext->pack_emit_mixed(prm.ingest->ctx, row_off.data(), ent_off.data(), ord_off.data(), tot_rows, tot_steps, mp,
                                 &out.dev.buf) != 0) {
            err = "build_schedule: the device packer's EMIT pass failed";
            return -1;
        }
        lap("  mixed: emit + scatter");
        out.device_packed = true;
        out.dev_ops = ext;
        out.device_ingest = true;
        out.n_rows_words = tot_rows + 4;
        out.n_entry_recs = tot_steps * G;
        out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        return 0;
    }
    out.rows.resize_uninit((size_t)tot_rows + 4);  // +16 B: the staging DMA reads whole 16-byte units
    for (int x = 0; x < 4; ++x) out.rows[(size_t)tot_rows + (size_t)x] = 0u;
    out.entries.resize_uninit((size_t)(tot_steps * G));
    {
        std::atomic<int64_t> nc{0};
        auto copier = [&]() {
            for (;;) {
                const int64_t c = nc.fetch_add(64);
                if (c >= n_descs) break;
                for (int64_t x = c; x < std::min<int64_t>(c + 64, n_descs); ++x) {
                    const CellOut& o = *by_desc[(size_t)x];
                    const CellDesc& d = out.cells[(size_t)x];
                    if (!o.a().rows.empty())
                        std::memcpy(&out.rows[d.row_off], o.a().rows.data(), o.a().rows.size() * sizeof(uint32_t));
                    if (!o.a().entries.empty())
                        std::memcpy(&out.entries[(size_t)d.ent_off * G], o.a().entries.data(),
                                    o.a().entries.size() * sizeof(Entry));
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(copier);
        copier();
        for (auto& t : th) t.join();
    }
    lap("  copy rows/entries");
    out.order.resize_uninit((size_t)n);
    {
        // the copies, in parallel over (round, block) slots
        std::atomic<int64_t> nslot{0};
        auto copier = [&]() {
            for (;;) {
                const int64_t s0 = nslot.fetch_add(64);
                if (s0 >= ncell) break;
                for (int64_t x = s0; x < std::min<int64_t>(s0 + 64, ncell); ++x) {
                    const int64_t rd = x / B, b = x % B;
                    const int64_t c = b * B + (b + rd) % B;
                    int64_t at = out.cell_ptr[(size_t)x];
                    auto append = [&](const CellOut& o) {
                        if (!o.a().order.empty())
                            std::memcpy(&out.order[(size_t)at], o.a().order.data(), o.a().order.size() * sizeof(int64_t));
                        at += o.n_order;
                    };
                    append(co[(size_t)c]);
                    for (const CellOut& o : extra[(size_t)c]) append(o);
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(copier);
        copier();
        for (auto& t : th) t.join();
    }
    lap("concatenate + order");
    if (n >= (int64_t)1 << 20) {
        // Giving the per-cell buffers back (about 1 GB in a few hundred thousand pieces at 20 M
        // ratings) costs ~0.1 s even in parallel; nothing waits for it, so it happens on a
        // detached thread after the schedule has been handed over.
        std::thread([cells_done = std::move(co), chunks_done = std::move(extra), sorted_done = std::move(sorted),
                     bptr_done = std::move(bptr)]() mutable {
            cells_done.clear();
            chunks_done.clear();
        }).detach();
    }
    lap("  release");
    out.device_ingest = on_device;
    out.n_rows_words = (int64_t)out.rows.size();
    out.n_entry_recs = (int64_t)out.entries.size();
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
    if (prm.B <= 0) {
        // One workgroup per CU with the whole LDS (measured: sharing a CU between two or three
        // smaller workgroups loses more to the longer hand-off chain than it gains in overlap).
        // rows a cell may hold in LDS, leaving a tenth for step entries
        const double cap_rows = (double)prm.lds_budget * 0.9 / geo.rowbytes - 2.0 * geo.G;
        // a cell of x ratings touches at most 2x rows, typically ~1.1x
        const double nn = (double)std::max<int64_t>(n, 1);
        int64_t bb = (int64_t)std::ceil(std::sqrt(nn / std::max(16.0, 0.6 * cap_rows)));
        if (bb <= prm.n_cu) {
            bb = (bb + 7) / 8 * 8;
            if (bb > prm.n_cu) bb = prm.n_cu;
            // a few more blocks than strictly needed keeps every CU busy
            if (bb > prm.n_cu * 0.7) bb = prm.n_cu;
        } else {
            // Several cells per workgroup and round.  Cells that overflow are chunked, so B need not
            // grow until the largest cell fits: aim the typical cell at half the LDS rows and take a
            // multiple of half the CU count -- the lower one when it is within 15 % (measured on
            // k = 128 / 256 at 20 M ratings: fewer, fuller cells win until chunking hits the hot items).
            bb = (int64_t)std::ceil(std::sqrt(nn / std::max(16.0, 0.5 * cap_rows)));
            const int64_t half = std::max<int64_t>(1, prm.n_cu / 2);
            const int64_t lo = std::max<int64_t>(prm.n_cu, bb / half * half);
            bb = (double)bb <= 1.15 * (double)lo ? lo : lo + half;
        }
        const int64_t lim = std::max<int64_t>(1, minrows / W);
        if (bb > lim) bb = lim;
        if (bb < 1) bb = 1;
        prm.B = (int)bb;
    }
    if (prm.W <= 0 && W == 4 && prm.degu && prm.degi && n > 0) {
        // Two waves per workgroup instead of four when the epoch is bound by one row's chain of
        // dependent updates rather than by the amount of work: fewer sub-rounds (barriers) sit on
        // that chain then, and the other waves would only wait.  Measured (MI355X): ML-20M shape
        // k = 64 3.54 vs 3.64 ms, Netflix shape k = 128 / 20 M 10.1 vs 11.1 ms; but uniform
        // popularity 2.94 vs 2.49 ms and a 10 M-rating DSGD partition 4.89 vs 4.50 ms -- hence the
        // comparison of the chain (longest row x cycles per dependent step) with the per-workgroup
        // work at two waves (cycles per rating fitted on the uniform workload) plus the per-cell latency.
        const int64_t dmax = std::max(*std::max_element(prm.degu, prm.degu + prm.U),
                                      *std::max_element(prm.degi, prm.degi + prm.I));
        const double np = (double)std::min<int64_t>(prm.B, prm.n_cu);
        const double passes = std::ceil((double)prm.B / (double)prm.n_cu);
        const double t_chain = (double)dmax * (170.0 + 2.0 * geo.L);
        const double t_rest2 = 48.0 * (4.0 / geo.G) * (double)n / np + 12000.0 * (double)prm.B * passes;
        if (t_chain > 1.1 * t_rest2) W = 2;
    }
    prm.W = W;
    for (;;) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = build_schedule(prm, u, i, r, orig, n, out, err);
        if (std::getenv("MFSGD_SCHED_TRACE"))
            std::fprintf(stderr, "[schedule] build_schedule(B = %d, W = %d): %.3f s to its return statement, %.3f s with its clean-up\n", prm.B, prm.W,
                         rc == 0 ? out.build_seconds : 0.0, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        if (rc == 0 || err.compare(0, 4, "lds:") != 0 || prm.W <= 1) return rc;
        prm.W >>= 1;  // smaller sub-cell tables and step groups
    }
}

void dsgd_plan_users(const int64_t* degu, int32_t U, int32_t G, int32_t* user_begin) {
    // users: boundary g is the first user at which the running rating count reaches g/G of the
    // total, pushed right where needed so that no range is empty while users remain
    int64_t total = 0;
    for (int32_t x = 0; x < U; ++x) total += degu[x];
    user_begin[0] = 0;
    int64_t acc = 0;
    int32_t x = 0;
    for (int32_t g = 1; g < G; ++g) {
        const int64_t want = (int64_t)(((__int128)total * g + G - 1) / G);
        while (x < U && acc < want) acc += degu[x++];
        int32_t b = x;
        if (b <= user_begin[g - 1]) b = std::min<int32_t>(U, user_begin[g - 1] + 1);
        if (b > U - (G - g)) b = std::max<int32_t>(user_begin[g - 1], U - (G - g));  // leave one user each for the rest
        while (x < b) acc += degu[x++];
        user_begin[g] = b;
    }
    user_begin[G] = U;
}

// Items -> G partitions, balanced by rating count; optionally chain-aware (chain_crit > 0).
//
// What bounds a DSGD epoch.  A sub-epoch of a rank cannot end before the heaviest item of its partition has seen all
// of the rank's ratings of it, one dependent update after the other.  Two sums follow from that:
//   (a) a rank computes for at least the SUM over the partitions of their heaviest items' chains;
//   (b) a Q block is trained by one rank after the other, so block p needs world x t_p per epoch however the ranks
//       overlap: the RING's epoch is at least world x the SLOWEST partition.
// Plain LPT (chain_crit = 0, the default) deals the G heaviest items out one per partition: every partition carries a
// chain of about the same length, the partitions take the same time, and (b) -- the binding one on a real ring --
// is as small as the heaviest item allows (its own chain, which no partitioning shortens).  It is the worst case for
// (a).  The chain-aware mode minimises (a): items in descending order of rating count; an item is CHAIN-CRITICAL
// when its chain on one rank (count / world steps of ~(170 + 2L) cycles) reaches `chain_crit` of what the rank's
// sub-epoch takes when it is bound by work (the scheduler's own model); while critical items remain the next partition
// is filled SEQUENTIALLY from the sorted list up to an equal share of what is left and closed; the rest is dealt
// LPT over the remaining partitions.  Measured (round 3, one rank of the N = 8 bench job, tools/part_profile.py,
// chain_crit = 0.3): (a) 116 K -> 33 K steps and the rank's own compute 9.17 -> 8.30 ms per epoch, but the heavy
// partition takes 1.31 ms against 0.98 ms for the others and 1.14 ms for every LPT partition, so (b) goes from
// 8 x 1.14 = 9.2 ms to 8 x 1.31 = 10.5 ms: worse on the ring.  Hence off by default; kept for hosts that run the
// partitions of ONE device back to back (virtual devices, out-of-core Q), where (a) is what counts.
// Items nobody rated go to the partitions with the fewest rows.  Balance: every partition is within one item's count
// of an equal share of what was left when it was opened.
// info (nullable, 4 values): {sum over partitions of their heaviest item's count, chain-critical items,
// partitions filled sequentially, the criticality threshold (global count)}.
void dsgd_plan_items(const int64_t* degi, int32_t I, int32_t G, int32_t world, int32_t k, double chain_crit, int32_t* item_part,
                     int64_t* info) {
    if (world < 1) world = 1;
    std::vector<int32_t> idx;
    idx.reserve((size_t)I);
    int64_t total = 0;
    for (int32_t x = 0; x < I; ++x)
        if (degi[x] > 0) {
            idx.push_back(x);
            total += degi[x];
        }
    std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return degi[a] > degi[b]; });
    // 1. the threshold
    const Geometry geo = geometry_for_k(k > 0 ? k : 64);
    double crit = chain_crit;
    if (const char* e = std::getenv("MFSGD_PLAN_CRIT")) crit = std::atof(e);  // A/B measurements; <= 0: plain LPT
    int64_t thr = INT64_MAX;
    if (crit > 0 && G > 1 && total > 0) {
        const double n_rp = (double)total / G / world;  // ratings of one rank's sub-epoch
        const double c_r = 48.0 * 4.0 / geo.G, c_0 = 12000.0, c_s = 170.0 + 2.0 * geo.L;
        const double b = std::min(256.0, std::max(8.0, std::sqrt(n_rp * c_r / c_0)));
        const double t_rest = n_rp * c_r / b + c_0 * b;
        thr = std::max<int64_t>(1, (int64_t)(crit * t_rest / c_s * world));
    }
    int64_t n_crit = 0;
    while (n_crit < (int64_t)idx.size() && degi[idx[(size_t)n_crit]] >= thr) ++n_crit;
    // 2. sequential fill while critical items remain
    std::vector<int64_t> load((size_t)G, 0), heaviest((size_t)G, 0);
    size_t pos = 0;
    int32_t p = 0;
    int64_t left = total;
    while ((int64_t)pos < n_crit && p < G - 1) {
        const int64_t cap = (left + (G - p) - 1) / (G - p);
        while (pos < idx.size() && (load[(size_t)p] == 0 || load[(size_t)p] + degi[idx[pos]] <= cap)) {
            const int32_t x = idx[pos++];
            item_part[x] = p;
            load[(size_t)p] += degi[x];
            heaviest[(size_t)p] = std::max(heaviest[(size_t)p], degi[x]);
        }
        left -= load[(size_t)p];
        ++p;
    }
    const int32_t n_seq = p;
    // 3. LPT over the partitions still open
    {
        using Item = std::pair<int64_t, int32_t>;  // (load, partition): smallest load, then smallest index
        std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
        for (int32_t g = p; g < G; ++g) heap.push({0, g});
        for (; pos < idx.size(); ++pos) {
            const int32_t x = idx[pos];
            Item t = heap.top();
            heap.pop();
            item_part[x] = t.second;
            t.first += degi[x];
            load[(size_t)t.second] = t.first;
            heaviest[(size_t)t.second] = std::max(heaviest[(size_t)t.second], degi[x]);
            heap.push(t);
        }
    }
    // 4. unrated items even out the row counts
    std::vector<int64_t> rows((size_t)G, 0);
    for (int32_t x : idx) rows[(size_t)item_part[x]]++;
    {
        using Item = std::pair<int64_t, int32_t>;
        std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
        for (int32_t g = 0; g < G; ++g) heap.push({rows[(size_t)g], g});
        for (int32_t x = 0; x < I; ++x) {
            if (degi[x] > 0) continue;
            Item t = heap.top();
            heap.pop();
            item_part[x] = t.second;
            t.first++;
            heap.push(t);
        }
    }
    if (info) {
        info[0] = 0;
        for (int32_t g = 0; g < G; ++g) info[0] += heaviest[(size_t)g];
        info[1] = n_crit;
        info[2] = n_seq;
        info[3] = thr == INT64_MAX ? 0 : thr;
    }
}

void dsgd_plan(const int64_t* degu, const int64_t* degi, int32_t U, int32_t I, int32_t G, int32_t* user_begin,
               int32_t* item_part) {
    dsgd_plan_users(degu, U, G, user_begin);
    dsgd_plan_items(degi, I, G, G, 64, 0.0, item_part, nullptr);
}

}  // namespace mfsgd
