// records.hpp -- the device-facing schedule records (layout shared by the host packer, the device
// packer and the kernels).  No reference counterpart exists (/root/reference/README.md:1-2).
#pragma once

#include <cstdint>

namespace mfsgd {

// Device-facing records (layout shared with kernels.hip).
//
// A cell whose LDS image would not fit is cut into CHUNKS: disjoint subsets of its users
// (or items), each a complete little cell with its own row list, sub-cell table and steps.
// A workgroup runs the chunks of a cell back to back (rows are stored and gathered again
// between them); chunk order is part of the canonical order.  descs[c] for c < B*B is the
// first chunk of cell c; further chunks live behind B*B and are linked through `next`.
struct CellDesc {
    uint32_t row_off;  // first entry of this chunk in rows[]
    uint32_t ent_off;  // first step of this chunk (entries index = step * G + slot)
    uint32_t n_steps;  // steps over all sub-cells; bit 31: the chunk carries a register-resident run
    uint16_t nu;       // distinct users  -> LDS slots [0, nu)
    uint16_t ni;       // distinct items  -> LDS slots [nu, nu + ni)
    uint32_t next;     // index of the cell's next chunk, 0 = this is the last one
    uint32_t rsv[3];   // [0] bit 0 (kCellLoneTile): every cell of this cell's tile is ONE chunk holding ONE item row
                       // -- the persistent kernel hands that row on through the tile's mailbox (kernels.hip)
};
static_assert(sizeof(CellDesc) == 32, "CellDesc layout");
constexpr uint32_t kCellCritical = 0x80000000u;
constexpr uint32_t kCellLoneTile = 1u;

struct SubDesc {
    uint32_t off;  // first step, relative to the cell's first step (low 16 bits) | solo steps << 16
    uint32_t n;    // general steps | run steps << 16 (run steps follow the general ones, the solo
                   // records -- 16 bytes each, header first -- follow kSoloPad idle steps behind them)
};

constexpr int kSoloPad = 2;  // idle steps between a sub-cell's run steps and its solo records

struct Entry {
    // p-side LDS address | q-side LDS address << 16 | flag << 31; addresses in 16-byte
    // units.  General step: flag = forward, this slot's q row is the one it updated in the
    // previous step (take it from registers, not from LDS).  Run step: flag = idle slot.
    uint32_t slots;
    float r;    // the rating (RMSE pass)
    float lrr;  // lr * r, rounded once on the host (training: s = fma(-lr, dot, lrr))
    float ce;   // decay factor of this slot's rows: c = 1 - lr*lambda, or 1 for an idle run slot
};
static_assert(sizeof(Entry) == 16, "Entry layout");

// What the device packer (pack.hip) reports per cell after its COUNT pass.
struct PackCellInfo {
    uint32_t status;   // 0 ok; 1: the cell does not fit the kernel's arrays / counters (host packer needed); 2: it has
                       // more rows than the launch provided for (the device retries with more before giving up)
    uint32_t nu, ni;   // distinct users / items
    uint32_t n_steps;  // step units incl. the two trailing idle steps (0 for an empty cell)
    uint32_t has_run;
    uint32_t pad;
    long long crit;    // sum over sub-rounds of the slowest wave's step-equivalents
};

constexpr int kRunMin = 8;    // shortest item chain worth a register-resident run
constexpr int kSoloMin = 12;  // shortest chain worth a solo run (a second wave does the off-chain half)

}  // namespace mfsgd
