// run_asm.hpp -- hand-scheduled gfx950 rating loops: the run loop (up to G resident items, one wave)
// and, first described here, the loops of a SOLO run: one item's
// chain of dependent updates, split between the wave that owns the sub-cell (the CHAIN wave:
// dot -> s -> q', nothing else) and one copy wave of the same workgroup on another SIMD (the
// HELPER: re-runs the cheap q recurrence from the posted s and does everything that is not on
// the chain -- p' = fma(s, q, c p), its store, and the final store of q).
//
// A solo run of n steps is a compact stream of 16-byte entries in the LDS schedule image:
//     [header][entry 0] ... [entry n-1][terminator]
//     entry t   = { slots_{t+1},  mailbox_t,  lr * r_t,  r_t }      ([r3] order: {slots, mailbox} is an aligned 8-byte
//     header    = { slots_0,      0,          0,         0   }       unit, see the helper wave below)
//     slots_t   = p-row LDS address | q-row LDS address << 16 (16-byte units); slots_n (in entry
//                 n-1 and in the terminator) addresses an all-zero row
//     mailbox_t = 0xFFFFFFFF in the schedule; the chain wave overwrites it with the bits of s_t,
//                 which is how the helper learns that step t has happened and what s_t was.
// The decay factor is the same for every step (there are no idle slots in a solo run), so it is
// an operand (SGPR pair {c, c}), not part of the entry.  Arithmetic is, instruction for
// instruction, DESIGN.md section 3: the pair of waves produces the bits Cell::solo_generic does.
//
// [r2, late] s of an even and the following odd step are posted TOGETHER (one ds_write2_b32 behind the odd
// step; a last odd-numbered step is posted alone): the chain wave issues half an LDS write per step less.
// (A long run can be cut in two -- the chain wave stores q into its LDS row between the halves, a second copy wave
// follows the second half from there and the first helper leaves q alone, %[fin] = 0: tools/ubench3 mode 4.  Measured
// and not used by the product: the helper is the slower of the pair at 16 lanes per rating, but a third wave on the
// LDS slows the chain wave by what the second helper gains.)
//
// Shared with tools/ubench3.hip, which times the loops against each other in isolation.
#pragma once

// The loops below are a few 64-byte instruction-cache lines long and their cycle count per pass depends on
// where the loop head sits inside a 32-byte fetch window (tools/ubench3.hip built with -DMFSGD_PAD_*=n;
// period 8 instructions).  Cycles per step at 0 / 2 / 4 / 6 s_nops past a 64-byte boundary:
//     solo pair   L = 16: 136.1 137.5 133.7 138.8    L = 32: 165.4 168.6 174.3 178.0    L = 64: 173.4 175.4 170.3 176.4
//     run loop    L = 16: 145.6 143.6 149.6 151.6    L = 32: 174.5 174.5 182.4 180.4    L = 64: 177.8 177.7 181.7 183.7
// Every loop head is therefore placed explicitly (operand [pad], an assembly-time constant): the product
// runs the layout that was timed, whatever code the compiler puts in front of the loop.
constexpr int mfsgd_pad_run(int lanes) {
#ifdef MFSGD_PAD_RUN
    return MFSGD_PAD_RUN;
#else
    return 2;
#endif
}
constexpr int mfsgd_pad_gen(int lanes) {
#ifdef MFSGD_PAD_GEN
    return MFSGD_PAD_GEN;
#else
    return 0;
#endif
}
constexpr int mfsgd_pad_chain(int lanes) {
#ifdef MFSGD_PAD_CHAIN
    return MFSGD_PAD_CHAIN;
#else
    return lanes == 32 ? 2 : 0;  // (the loop that posts s in pairs; the table above is the earlier one-post-per-step loop)
#endif
}
constexpr int mfsgd_pad_helper(int lanes) {
#ifdef MFSGD_PAD_HELPER
    return MFSGD_PAD_HELPER;
#else
    return lanes == 16 ? 6 : 2;  // [r3] the loop that takes the steps in pairs (tools/ubench3, profiles/r03_ubench3.log)
#endif
}
#define MFSGD_LOOP_ALIGN ".p2align 6\n\t.rept %c[pad]\n\ts_nop 0\n\t.endr\n\t"

// ---- hand-scheduled run loop (gfx950) ---------------------------------------------------
// `pairs` x 2 run steps of one wave: resident q rows in v[100:103] / v[140:143] (alternating),
// one p row prefetched a step ahead, entry words fetched one / two steps ahead.  Per step:
// 3 dot ops, 4 DPP adds whose two required wait states are filled with the independent
// scale / address / LDS-issue instructions (the address is one v_mad_u32_u16: low 16 bits of
// the entry word x 16 + row base), 1 fma for s, 4 pk_fma, 1 store.  Arithmetic is
// instruction for instruction what Cell::apply's run_step does in C++ (which remains the
// reference for it and the RMSE path).  Fixed VGPRs v100..v143 are declared clobbered.
//   ea      : LDS byte address of this lane group's entry of run step 0 (entry stride `EST`)
//   rowbase : LDS byte address of row slot 0 plus this lane's 16-byte offset inside a row
// Register map (the text is a macro, which cannot carry comments line by line):
//   v138 entry pointer, v139 row base; v114 / v115 entry word `slots` of the even / odd step in
//   flight, v[116:117] / v[118:119] its {lr*r, ce}; v112 / v113 p-row address, v[104:107] / v[108:111]
//   p row (prefetched one step ahead); v[100:103] <-> v[140:143] resident q rows; v[120:121] chunk
//   products, v132 dot, v[122:125] ce*q, v[126:129] ce*p, v130 s, v[134:137] p'.
//   Step: wait for p and the entry words -> chunk dot (pk_mul, pk_fma, add) -> 4 DPP adds
//   (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) interleaved with the next p address, the
//   ce scalings, the p prefetch and the entry prefetches -> EXTRA: the xor-16 / xor-32 levels for
//   32 / 64 lanes per rating -> s = fma(-lr, dot, lr*r) -> q' = fma(s, p, ce*q), p' = fma(s, q, ce*p)
//   -> store p'.  The prologue loads step 0; a harmless rewrite of an entry word keeps "one LDS
//   operation behind the reads" so that every pass can use the same counted wait.
#define MFSGD_RUN_LOOP_ASM_TEXT(EXTRA, SFMA) \
        "v_mov_b32 v138, %[ea]\n\t" \
        "v_mov_b32 v131, %[lr]\n\t" \
        "v_mov_b32 v139, %[rb]\n\t" \
        "ds_read_b32 v114, v138\n\t" \
        "ds_read_b64 v[116:117], v138 offset:8\n\t" \
        "ds_read_b32 v115, v138 offset:%c[e1]\n\t" \
        "v_mov_b32 v100, %[q0]\n\t" \
        "v_mov_b32 v101, %[q1]\n\t" \
        "v_mov_b32 v102, %[q2]\n\t" \
        "v_mov_b32 v103, %[q3]\n\t" \
        "s_waitcnt lgkmcnt(2)\n\t" \
        "v_and_b32 v133, 0xffff, v114\n\t" \
        "v_lshl_add_u32 v112, v133, 4, v139\n\t" \
        "ds_read_b128 v[104:107], v112\n\t" \
        "ds_write_b32 v138, v114\n\t" \
        MFSGD_LOOP_ALIGN \
        "1:\n\t" \
        "s_waitcnt lgkmcnt(1)\n\t" \
        "v_pk_mul_f32 v[120:121], v[104:105], v[100:101]\n\t" \
        "v_pk_fma_f32 v[120:121], v[106:107], v[102:103], v[120:121]\n\t" \
        "v_add_f32 v132, v120, v121\n\t" \
        "v_mad_u32_u16 v113, v115, 16, v139\n\t" \
        "v_pk_mul_f32 v[122:123], v[116:117], v[100:101] op_sel:[1,0]\n\t" \
        "v_add_f32_dpp v132, v132, v132 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "ds_read_b128 v[108:111], v113\n\t" \
        "v_pk_mul_f32 v[124:125], v[116:117], v[102:103] op_sel:[1,0]\n\t" \
        "v_add_f32_dpp v132, v132, v132 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_pk_mul_f32 v[126:127], v[116:117], v[104:105] op_sel:[1,0]\n\t" \
        "v_pk_mul_f32 v[128:129], v[116:117], v[106:107] op_sel:[1,0]\n\t" \
        "v_add_f32_dpp v132, v132, v132 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "ds_read_b32 v114, v138 offset:%c[e2]\n\t" \
        "ds_read_b64 v[118:119], v138 offset:%c[e1p8]\n\t" \
        "v_add_f32_dpp v132, v132, v132 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        EXTRA \
        SFMA("116") \
        "v_pk_fma_f32 v[140:141], v[130:131], v[104:105], v[122:123] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[142:143], v[130:131], v[106:107], v[124:125] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[134:135], v[130:131], v[100:101], v[126:127] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[136:137], v[130:131], v[102:103], v[128:129] op_sel_hi:[0,1,1]\n\t" \
        "s_sub_u32 %[n], %[n], 1\n\t" \
        "ds_write_b128 v112, v[134:137]\n\t" \
        "s_waitcnt lgkmcnt(1)\n\t" \
        "v_pk_mul_f32 v[120:121], v[108:109], v[140:141]\n\t" \
        "v_pk_fma_f32 v[120:121], v[110:111], v[142:143], v[120:121]\n\t" \
        "v_add_f32 v132, v120, v121\n\t" \
        "v_mad_u32_u16 v112, v114, 16, v139\n\t" \
        "v_pk_mul_f32 v[122:123], v[118:119], v[140:141] op_sel:[1,0]\n\t" \
        "v_add_f32_dpp v132, v132, v132 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "ds_read_b128 v[104:107], v112\n\t" \
        "v_pk_mul_f32 v[124:125], v[118:119], v[142:143] op_sel:[1,0]\n\t" \
        "v_add_f32_dpp v132, v132, v132 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_pk_mul_f32 v[126:127], v[118:119], v[108:109] op_sel:[1,0]\n\t" \
        "v_pk_mul_f32 v[128:129], v[118:119], v[110:111] op_sel:[1,0]\n\t" \
        "v_add_f32_dpp v132, v132, v132 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "ds_read_b32 v115, v138 offset:%c[e3]\n\t" \
        "ds_read_b64 v[116:117], v138 offset:%c[e2p8]\n\t" \
        "v_add_f32_dpp v132, v132, v132 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        EXTRA \
        SFMA("118") \
        "v_pk_fma_f32 v[100:101], v[130:131], v[108:109], v[122:123] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[102:103], v[130:131], v[110:111], v[124:125] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[134:135], v[130:131], v[140:141], v[126:127] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[136:137], v[130:131], v[142:143], v[128:129] op_sel_hi:[0,1,1]\n\t" \
        "v_add_u32 v138, %c[e2], v138\n\t" \
        "s_cmp_lg_u32 %[n], 0\n\t" \
        "ds_write_b128 v113, v[134:137]\n\t" \
        "s_cbranch_scc1 1b\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_mov_b32 %[q0], v100\n\t" \
        "v_mov_b32 %[q1], v101\n\t" \
        "v_mov_b32 %[q2], v102\n\t" \
        "v_mov_b32 %[q3], v103\n\t"

// xor-16 / xor-32 levels of the dot reduction for L = 32 / 64 (see swap_add16 / swap_add32): v133 is free
#define MFSGD_SWAP_ADD16 "v_mov_b32 v133, v132\n\ts_nop 1\n\tv_permlane16_swap_b32 v132, v133\n\ts_nop 1\n\tv_add_f32 v132, v132, v133\n\t"
#define MFSGD_SWAP_ADD32 "v_mov_b32 v133, v132\n\ts_nop 1\n\tv_permlane32_swap_b32 v132, v133\n\ts_nop 1\n\tv_add_f32 v132, v132, v133\n\t"
// Both levels at once for L = 64 (one rating per wave, so the dot is wave-uniform): after the 16-lane
// reduction every lane of row r holds S_r; row_bcast:15 adds lane 15 of rows 0 / 2 into rows 1 / 3
// (S0+S1, S2+S3), row_bcast:31 adds lane 31 into rows 2 / 3, so lane 63 holds (S2+S3)+(S0+S1) -- the
// bits of the contract's tree, addition being commutative -- and a v_readlane broadcasts it.  Two DPP
// adds and two moves on the dependent chain instead of two five-instruction swap levels.
#define MFSGD_BCAST_ADD64 "s_nop 1\n\tv_add_f32_dpp v132, v132, v132 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\tv_add_f32_dpp v132, v132, v132 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0\n\tv_readlane_b32 vcc_lo, v132, 63\n\t"
// The same idea for the xor-16 level of a SOLO run at L = 32 (one rating per wave there too: both lane groups
// work on it): row_bcast:15 puts S0+S1 into row 1 (lane 31), a v_readlane hands it on as a scalar.
#define MFSGD_BCAST_ADD32 "s_nop 1\n\tv_add_f32_dpp v132, v132, v132 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 0\n\tv_readlane_b32 vcc_lo, v132, 31\n\t"
// s = fma(-lr, dot, lr*r): the dot in v132 (every lane of the group holds it), or -- after
// MFSGD_BCAST_ADD64 -- in vcc_lo, with lr in v131 (one scalar operand per VALU instruction on gfx9)
#define MFSGD_SFMA_V(LRR) "v_fma_f32 v130, -%[lr], v132, v" LRR "\n\t"
#define MFSGD_SFMA_S(LRR) "v_fma_f32 v130, -v131, vcc_lo, v" LRR "\n\t"
// the same with the destination named (the solo chain keeps s of an even and an odd step apart)
#define MFSGD_SFMA2_V(S, LRR) "v_fma_f32 v" S ", -%[lr], v132, v" LRR "\n\t"
#define MFSGD_SFMA2_S(S, LRR) "v_fma_f32 v" S ", -v131, vcc_lo, v" LRR "\n\t"
#define MFSGD_RUN_LOOP_ASM_OPERANDS                                                                                   \
    : [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [n] "+s"(pairs)                              \
    : [ea] "v"(ea), [rb] "v"(rowbase), [lr] "s"(lr), [e1] "n"(EST), [e2] "n"(2 * EST), [e3] "n"(3 * EST),              \
      [e1p8] "n"(EST + 8), [e2p8] "n"(2 * EST + 8), [pad] "n"(PADV)                                                                    \
    : "memory", "scc", "vcc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", \
      "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125",  \
      "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139",  \
      "v140", "v141", "v142", "v143"



// ---- hand-scheduled GENERAL step loop (gfx950) [r3] -------------------------------------------
// `n` general steps of one wave (kernels.hip Cell::apply, `step`): every lane group applies one rating per step --
// both rows come from LDS, both go back -- and consecutive steps are independent except where the packer flagged a
// q row as forwarded (bit 31 of the next entry's slots word: the same item in the same lane slot; its new row is taken
// from registers, the LDS copy read ahead of the write being stale).  The compiler's schedule of the C++ step waits
// twice per step for LDS reads it issued a few instructions earlier (~335 cycles per step of G ratings, measured
// in situ); here the rows of step t + 1 and the entry words of step t + 2 are fetched under the arithmetic of step
// t with counted waits, as in the run loop above: 24 VALU + 6 LDS instructions per step.
//   entry (16 bytes per slot and step): {slots = p address | q address << 16 | forward flag << 31 (16-byte units),
//   rating, lr * rating, decay}; the decay factor of a general step is the uniform c (operand c2 = {c, c}).
// Register map (two sets, A = even steps, B = odd steps):
//   v138 entry pointer, v139 row base + this lane's 16-byte offset; v114 / v115 slots word whose rows are fetched
//   during an odd / even step; v116 / v117 lr*r of the even / odd step; v112, v148 / v113, v149 p and q address of
//   set A / B; v[104:107], v[100:103] / v[108:111], v[140:143] p and q row of set A / B; v[144:147] the q row read
//   ahead (selected against the forwarded q'); v[120:121] chunk products, v132 dot, v[122:125] c*q, v[126:129] c*p,
//   v130 s, v[134:137] p'; q' is computed straight into the OTHER set's q registers.
// LDS operations of one step, in issue order: read p(t+1), read q(t+1), read slots(t+2), read lr*r(t+1), write p'(t),
// write q'(t).  LDS operations of a wave complete in order, so "all but the last four" has q(t+1) in registers (the
// select behind the writes) and "all but the last two" at the top of the next step has the two entry words.
// Hazards, as the packer guarantees (schedule.cpp, "Eligibility"): a p row never appears in two consecutive steps
// of a wave; a q row only in the same lane slot, flagged.  Arithmetic: instruction for instruction DESIGN.md section 3.
#define MFSGD_GEN_HALF(P0, P1, P2, P3, Q0, Q1, Q2, Q3, PADDR, QADDR, NP0, NP3, NQ0, NQ1, NQ2, NQ3, NPADDR, NQADDR, SLOT_NEXT, \
                       SLOT_NEXT2, OFF_SLOT2, LRR, LRR_NEXT, OFF_LRR1, EXTRA, SFMA)                                              \
        "s_waitcnt lgkmcnt(2)\n\t" \
        "v_pk_mul_f32 v[120:121], v[" P0 ":" P1 "], v[" Q0 ":" Q1 "]\n\t" \
        "v_pk_fma_f32 v[120:121], v[" P2 ":" P3 "], v[" Q2 ":" Q3 "], v[120:121]\n\t" \
        "v_add_f32 v132, v120, v121\n\t" \
        "v_mad_u32_u16 v" NPADDR ", v" SLOT_NEXT ", 16, v139\n\t" \
        "v_bfe_u32 v133, v" SLOT_NEXT ", 16, 15\n\t" \
        "v_add_f32_dpp v132, v132, v132 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "ds_read_b128 v[" NP0 ":" NP3 "], v" NPADDR "\n\t" \
        "v_lshl_add_u32 v" NQADDR ", v133, 4, v139\n\t" \
        "v_add_f32_dpp v132, v132, v132 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "ds_read_b128 v[144:147], v" NQADDR "\n\t" \
        "v_pk_mul_f32 v[122:123], v[" Q0 ":" Q1 "], %[c2]\n\t" \
        "v_add_f32_dpp v132, v132, v132 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_pk_mul_f32 v[124:125], v[" Q2 ":" Q3 "], %[c2]\n\t" \
        "v_pk_mul_f32 v[126:127], v[" P0 ":" P1 "], %[c2]\n\t" \
        "v_add_f32_dpp v132, v132, v132 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_pk_mul_f32 v[128:129], v[" P2 ":" P3 "], %[c2]\n\t" \
        "ds_read_b32 v" SLOT_NEXT2 ", v138 offset:" OFF_SLOT2 "\n\t" \
        "ds_read_b32 v" LRR_NEXT ", v138 offset:" OFF_LRR1 "\n\t" \
        EXTRA \
        SFMA(LRR) \
        "v_pk_fma_f32 v[" NQ0 ":" NQ1 "], v[130:131], v[" P0 ":" P1 "], v[122:123] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[" NQ2 ":" NQ3 "], v[130:131], v[" P2 ":" P3 "], v[124:125] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[134:135], v[130:131], v[" Q0 ":" Q1 "], v[126:127] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[136:137], v[130:131], v[" Q2 ":" Q3 "], v[128:129] op_sel_hi:[0,1,1]\n\t" \
        "s_sub_u32 %[n], %[n], 1\n\t" \
        "v_cmp_gt_i32 vcc, 0, v" SLOT_NEXT "\n\t" \
        "ds_write_b128 v" PADDR ", v[134:137]\n\t" \
        "ds_write_b128 v" QADDR ", v[" NQ0 ":" NQ3 "]\n\t" \
        "s_cmp_eq_u32 %[n], 0\n\t" \
        "s_waitcnt lgkmcnt(4)\n\t" \
        "v_cndmask_b32 v" NQ0 ", v144, v" NQ0 ", vcc\n\t" \
        "v_cndmask_b32 v" NQ1 ", v145, v" NQ1 ", vcc\n\t" \
        "v_cndmask_b32 v" NQ2 ", v146, v" NQ2 ", vcc\n\t" \
        "v_cndmask_b32 v" NQ3 ", v147, v" NQ3 ", vcc\n\t"

// `ea`: LDS byte address of this lane group's entry of general step 0 (entry stride EST), n >= 1 steps.
#define MFSGD_GEN_LOOP_ASM_TEXT(EXTRA, SFMA) \
        "v_mov_b32 v138, %[ea]\n\t" \
        "v_mov_b32 v131, %[lr]\n\t" \
        "v_mov_b32 v139, %[rb]\n\t" \
        "ds_read_b32 v114, v138\n\t" \
        "ds_read_b32 v116, v138 offset:8\n\t" \
        "ds_read_b32 v115, v138 offset:%c[e1]\n\t" \
        "s_waitcnt lgkmcnt(2)\n\t" \
        "v_mad_u32_u16 v112, v114, 16, v139\n\t" \
        "v_bfe_u32 v133, v114, 16, 15\n\t" \
        "v_lshl_add_u32 v148, v133, 4, v139\n\t" \
        "ds_read_b128 v[104:107], v112\n\t" \
        "ds_read_b128 v[100:103], v148\n\t" \
        "ds_write_b32 v138, v114\n\t" \
        "ds_write_b32 v138, v114\n\t" \
        MFSGD_LOOP_ALIGN \
        "1:\n\t" \
        MFSGD_GEN_HALF("104", "105", "106", "107", "100", "101", "102", "103", "112", "148", "108", "111", "140", "141", "142", "143", \
                       "113", "149", "115", "114", "%c[e2]", "116", "117", "%c[e1p8]", EXTRA, SFMA) \
        "s_cbranch_scc1 2f\n\t" \
        MFSGD_GEN_HALF("108", "109", "110", "111", "140", "141", "142", "143", "113", "149", "104", "107", "100", "101", "102", "103", \
                       "112", "148", "114", "115", "%c[e3]", "117", "116", "%c[e2p8]", EXTRA, SFMA) \
        "v_add_u32 v138, %c[e2], v138\n\t" \
        "s_cbranch_scc0 1b\n\t" \
        "2:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t"

#define MFSGD_GEN_LOOP_ASM_OPERANDS                                                                                    \
    : [n] "+s"(n)                                                                                                      \
    : [ea] "v"(ea), [rb] "v"(rowbase), [lr] "s"(lr), [c2] "s"(c2), [e1] "n"(EST), [e2] "n"(2 * EST), [e3] "n"(3 * EST), \
      [e1p8] "n"(EST + 8), [e2p8] "n"(2 * EST + 8), [pad] "n"(PADV)                                                    \
    : "memory", "scc", "vcc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", \
      "v112", "v113", "v114", "v115", "v116", "v117", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127",  \
      "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141",  \
      "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149"

// ---- chain wave -------------------------------------------------------------------------------
// v138 entry pointer (-> entry t at the top of half A), v139 row base + this lane's 16-byte offset;
// v[100:103] q (updated in place); v[104:107] / v[108:111] p row of the even / odd step (prefetched
// a step ahead); v[116:117] / v[118:119] {lr*r, next slots} of the even / odd step; v113 p address;
// v[120:121] chunk products, v132 dot, v[122:125] c*q, v130 s.
#define MFSGD_SOLO_CHAIN_HALF(P0, P1, P2, P3, N0, N1, N2, N3, ELRR, ESLOT, NEXTE, OFF_NEXT, WAIT, S, S1, EXTRA, SFMA) \
        "s_waitcnt lgkmcnt(" WAIT ")\n\t" \
        "v_pk_mul_f32 v[120:121], v[" P0 ":" P1 "], v[100:101]\n\t" \
        "v_pk_fma_f32 v[120:121], v[" P2 ":" P3 "], v[102:103], v[120:121]\n\t" \
        "v_add_f32 v132, v120, v121\n\t" \
        "v_mad_u32_u16 v113, v" ESLOT ", 16, v139\n\t" \
        "v_pk_mul_f32 v[122:123], v[100:101], %[c2]\n\t" \
        "v_add_f32_dpp v132, v132, v132 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "ds_read_b128 v[" N0 ":" N3 "], v113\n\t" \
        "v_pk_mul_f32 v[124:125], v[102:103], %[c2]\n\t" \
        "v_add_f32_dpp v132, v132, v132 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "ds_read2_b32 v[" NEXTE "], v138 " OFF_NEXT "\n\t" \
        "s_nop 0\n\t" \
        "v_add_f32_dpp v132, v132, v132 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "s_sub_u32 %[n], %[n], 1\n\t" \
        "s_cmp_eq_u32 %[n], 0\n\t" \
        "v_add_f32_dpp v132, v132, v132 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        EXTRA \
        SFMA(S, ELRR) \
        "v_pk_fma_f32 v[100:101], v[" S ":" S1 "], v[" P0 ":" P1 "], v[122:123] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[102:103], v[" S ":" S1 "], v[" P2 ":" P3 "], v[124:125] op_sel_hi:[0,1,1]\n\t"

// `ea` = LDS byte address of the header entry; n >= 1 steps.
#define MFSGD_SOLO_CHAIN_ASM_TEXT(EXTRA, SFMA) \
        "v_mov_b32 v138, %[ea]\n\t" \
        "v_mov_b32 v131, %[lr]\n\t" \
        "v_mov_b32 v139, %[rb]\n\t" \
        "ds_read_b32 v133, v138\n\t" \
        "ds_read2_b32 v[116:117], v138 offset0:6 offset1:4\n\t" \
        "v_mov_b32 v100, %[q0]\n\t" \
        "v_mov_b32 v101, %[q1]\n\t" \
        "v_mov_b32 v102, %[q2]\n\t" \
        "v_mov_b32 v103, %[q3]\n\t" \
        "s_waitcnt lgkmcnt(1)\n\t" \
        "v_mad_u32_u16 v113, v133, 16, v139\n\t" \
        "ds_read_b128 v[104:107], v113\n\t" \
        "ds_write_b32 v138, v133\n\t" \
        "s_nop 1\n\t" \
        "v_add_u32 v138, 16, v138\n\t" \
        MFSGD_LOOP_ALIGN \
        "1:\n\t" \
        MFSGD_SOLO_CHAIN_HALF("104", "105", "106", "107", "108", "109", "110", "111", "116", "117", "118:119", "offset0:6 offset1:4", "1", "130", "131", EXTRA, SFMA) \
        "s_cbranch_scc1 2f\n\t" \
        MFSGD_SOLO_CHAIN_HALF("108", "109", "110", "111", "104", "105", "106", "107", "118", "119", "116:117", "offset0:10 offset1:8", "0", "128", "129", EXTRA, SFMA) \
        "ds_write2_b32 v138, v130, v128 offset0:1 offset1:5\n\t" \
        "v_add_u32 v138, 32, v138\n\t" \
        "s_cbranch_scc0 1b\n\t" \
        "s_branch 3f\n\t" \
        "2:\n\t" \
        "ds_write_b32 v138, v130 offset:4\n\t" \
        "3:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_mov_b32 %[q0], v100\n\t" \
        "v_mov_b32 %[q1], v101\n\t" \
        "v_mov_b32 %[q2], v102\n\t" \
        "v_mov_b32 %[q3], v103\n\t"

#define MFSGD_SOLO_CHAIN_OPERANDS                                                                                      \
    : [n] "+s"(n), [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3])                                  \
    : [ea] "v"(ea), [rb] "v"(rowbase), [lr] "s"(lr), [c2] "s"(c2), [pad] "n"(PADV)                                     \
    : "memory", "scc", "vcc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", \
      "v113", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v128", "v129", "v130",  \
      "v131", "v132", "v133", "v138", "v139"

// ---- helper wave ------------------------------------------------------------------------------
// [r3] The helper takes the steps in PAIRS, as the chain wave posts them: ONE ds_read2_b64 fetches {slots_{t+1}, s_t}
// and {slots_{t+2}, s_{t+1}} -- bytes 4..11 of two consecutive entries; the address is 4 modulo 8, which the LDS of
// gfx9 takes in the unaligned mode the driver runs it in -- one poll test per pair (both mailboxes: v_max_u32 is all
// ones iff either is), and the loop counter per pair.  Round 2's loop read {s, slots} per step with a ds_read2_b32 and
// tested per step: 130 cycles per step at 16 lanes per rating against the chain wave's 120, so the pair ran at the
// helper's pace.  A last odd step (posted alone by the chain wave) is a tail of its own.
// v138 -> entry t + 4 at the top of a pair; v139 row base + lane offset; v[100:103] q_t; v[104:107] / v[108:111] p row
// of the pair's first / second step, v112 / v113 their addresses; v[116:119] <-> v[150:153] the pair's words
// {slots_{t+1}, s_t, slots_{t+2}, s_{t+1}} (s is the HIGH half of an aligned register pair: op_sel:[1,0,0]); v140 q
// address; v[122:125] c*q, v[126:129] c*p, v[134:137] p'.  %[spins]: polls left before giving up (the chain wave
// always arrives; the bound only keeps a broken schedule from hanging the GPU) -- on return %[spins] == 0 means it
// gave up and nothing further was written.
// LDS operations of a pair, in issue order: read p_{t+1}, read the next pair's words, write p'_t, read p_{t+2}, write
// p'_{t+1}: "all but the last one" at the top of the next pair has its words and its first p row.
// NOTHING of the rows is read before s_0 has been posted: the chain wave runs the general and run steps of its sub-cell
// before the solo run, and they may update the p rows the run is about to use (the helper starts with the sub-round).
// Once s_0 is there they are final -- only the helper writes them.  (A first version of this loop fetched p_0 in its
// prologue, in front of the poll: bit-exact in tools/ubench3 -- which has no general steps -- and wrong in the product
// whenever a user of step 0 also had a general step in the sub-cell.  tools/ubench3 now dirties p_0 first.)
#ifdef MFSGD_HELPER_PER_STEP  // round 2's per-step helper on the new record layout (A/B, bisecting)
#define MFSGD_SOLO_HELPER_HALF(TAG, P0, P1, P2, P3, N0, N3, MBOX, MSLOT, MPAIR, NEXTM, PADDR, NADDR, OFF0, OFF1) \
        "s_waitcnt lgkmcnt(1)\n\t" \
        "v_cmp_eq_u32 vcc, -1, v" MBOX "\n\t" \
        "v_mad_u32_u16 v" NADDR ", v" MSLOT ", 16, v139\n\t" \
        "v_pk_mul_f32 v[126:127], v[" P0 ":" P1 "], %[c2]\n\t" \
        "s_cbranch_vccnz 2" TAG "f\n\t" \
        "1" TAG ":\n\t" \
        "ds_read_b128 v[" N0 ":" N3 "], v" NADDR "\n\t" \
        "v_pk_mul_f32 v[128:129], v[" P2 ":" P3 "], %[c2]\n\t" \
        "ds_read2_b32 v[" NEXTM "], v138 " OFF0 "\n\t" \
        "v_pk_mul_f32 v[122:123], v[100:101], %[c2]\n\t" \
        "v_pk_mul_f32 v[124:125], v[102:103], %[c2]\n\t" \
        "v_pk_fma_f32 v[134:135], v[" MPAIR "], v[100:101], v[126:127] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[136:137], v[" MPAIR "], v[102:103], v[128:129] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[100:101], v[" MPAIR "], v[" P0 ":" P1 "], v[122:123] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 v[102:103], v[" MPAIR "], v[" P2 ":" P3 "], v[124:125] op_sel_hi:[0,1,1]\n\t" \
        "s_sub_u32 %[n], %[n], 1\n\t" \
        "ds_write_b128 v" PADDR ", v[134:137]\n\t" \
        "s_cmp_eq_u32 %[n], 0\n\t"

#define MFSGD_SOLO_HELPER_SLOW(TAG, MBOX, OFF1) \
        "2" TAG ":\n\t" \
        "s_sleep 1\n\t" \
        "ds_read_b32 v" MBOX ", v138 offset:" OFF1 "\n\t" \
        "s_sub_u32 %[spins], %[spins], 1\n\t" \
        "s_cmp_eq_u32 %[spins], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_cmp_eq_u32 vcc, -1, v" MBOX "\n\t" \
        "s_cbranch_vccnz 2" TAG "b\n\t" \
        "s_branch 1" TAG "b\n\t"

#define MFSGD_SOLO_HELPER_ASM_TEXT \
        "v_mov_b32 v138, %[ea]\n\t" \
        "v_mov_b32 v139, %[rb]\n\t" \
        "ds_read_b32 v133, v138\n\t" \
        "ds_read2_b32 v[116:117], v138 offset0:5 offset1:4\n\t" \
        "s_waitcnt lgkmcnt(1)\n\t" \
        "v_mad_u32_u16 v112, v133, 16, v139\n\t" \
        "v_bfe_u32 v140, v133, 16, 15\n\t" \
        "v_lshl_add_u32 v140, v140, 4, v139\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_cmp_eq_u32 vcc, -1, v116\n\t" \
        "s_cbranch_vccz 4f\n\t" \
        "3:\n\t" \
        "s_sleep 1\n\t" \
        "ds_read_b32 v116, v138 offset:20\n\t" \
        "s_sub_u32 %[spins], %[spins], 1\n\t" \
        "s_cmp_eq_u32 %[spins], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_cmp_eq_u32 vcc, -1, v116\n\t" \
        "s_cbranch_vccnz 3b\n\t" \
        "4:\n\t" \
        "ds_read_b128 v[100:103], v140\n\t" \
        "ds_read_b128 v[104:107], v112\n\t" \
        "ds_write_b32 v138, v133\n\t" \
        "s_nop 1\n\t" \
        "v_add_u32 v138, 16, v138\n\t" \
        MFSGD_LOOP_ALIGN \
        "5:\n\t" \
        MFSGD_SOLO_HELPER_HALF("0", "104", "105", "106", "107", "108", "111", "116", "117", "116:117", "118:119", "112", "113", "offset0:5 offset1:4", "4") \
        "s_cbranch_scc1 8f\n\t" \
        MFSGD_SOLO_HELPER_HALF("1", "108", "109", "110", "111", "104", "107", "118", "119", "118:119", "116:117", "113", "112", "offset0:9 offset1:8", "20") \
        "v_add_u32 v138, 32, v138\n\t" \
        "s_cbranch_scc0 5b\n\t" \
        "8:\n\t" \
        "s_cmp_eq_u32 %[fin], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "ds_write_b128 v140, v[100:103]\n\t" \
        "s_branch 9f\n\t" \
        MFSGD_SOLO_HELPER_SLOW("0", "116", "4") \
        MFSGD_SOLO_HELPER_SLOW("1", "118", "20") \
        "9:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t"

#else
#ifndef MFSGD_HV
#define MFSGD_HV 3
#endif
#define MFSGD_SOLO_HELPER_PAIR(TAG, M0, M1, M2, M3, N0, N3) \
        "s_waitcnt lgkmcnt(1)\n\t" \
        "v_max_u32 v133, v" M1 ", v" M3 "\n\t" \
        "v_mad_u32_u16 v113, v" M0 ", 16, v139\n\t" \
        "v_cmp_eq_u32 vcc, -1, v133\n\t" \
        "v_pk_mul_f32 v[126:127], v[104:105], %[c2]\n\t" \
        "s_cbranch_vccnz 7" TAG "f\n\t" \
        "6" TAG ":\n\t" \
        "ds_read_b128 v[108:111], v113\n\t" \
        "v_pk_mul_f32 v[128:129], v[106:107], %[c2]\n\t" \
        "ds_read2_b64 v[" N0 ":" N3 "], v138 offset0:4 offset1:6\n\t" \
        "v_pk_mul_f32 v[122:123], v[100:101], %[c2]\n\t" \
        "v_pk_mul_f32 v[124:125], v[102:103], %[c2]\n\t" \
        "v_pk_fma_f32 v[134:135], v[" M0 ":" M1 "], v[100:101], v[126:127] op_sel:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[136:137], v[" M0 ":" M1 "], v[102:103], v[128:129] op_sel:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[100:101], v[" M0 ":" M1 "], v[104:105], v[122:123] op_sel:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[102:103], v[" M0 ":" M1 "], v[106:107], v[124:125] op_sel:[1,0,0]\n\t" \
        "s_sub_u32 %[n], %[n], 2\n\t" \
        "ds_write_b128 v112, v[134:137]\n\t" \
        "v_mad_u32_u16 v112, v" M2 ", 16, v139\n\t" \
        "s_waitcnt lgkmcnt(2)\n\t" \
        "ds_read_b128 v[104:107], v112\n\t" \
        "v_pk_mul_f32 v[126:127], v[108:109], %[c2]\n\t" \
        "v_pk_mul_f32 v[128:129], v[110:111], %[c2]\n\t" \
        "v_pk_mul_f32 v[122:123], v[100:101], %[c2]\n\t" \
        "v_pk_mul_f32 v[124:125], v[102:103], %[c2]\n\t" \
        "v_pk_fma_f32 v[134:135], v[" M2 ":" M3 "], v[100:101], v[126:127] op_sel:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[136:137], v[" M2 ":" M3 "], v[102:103], v[128:129] op_sel:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[100:101], v[" M2 ":" M3 "], v[108:109], v[122:123] op_sel:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[102:103], v[" M2 ":" M3 "], v[110:111], v[124:125] op_sel:[1,0,0]\n\t" \
        "v_add_u32 v138, 32, v138\n\t" \
        "s_cmp_lt_u32 %[n], 2\n\t" \
        "ds_write_b128 v113, v[134:137]\n\t"

// re-poll of a pair whose s has not been posted yet (out of line)
#define MFSGD_SOLO_HELPER_SLOW(TAG, M0, M1, M3) \
        "7" TAG ":\n\t" \
        "s_sleep 1\n\t" \
        "ds_read2_b64 v[" M0 ":" M3 "], v138 offset1:2\n\t" \
        "s_sub_u32 %[spins], %[spins], 1\n\t" \
        "s_cmp_eq_u32 %[spins], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_max_u32 v133, v" M1 ", v" M3 "\n\t" \
        "v_cmp_eq_u32 vcc, -1, v133\n\t" \
        "s_cbranch_vccnz 7" TAG "b\n\t" \
        "s_branch 6" TAG "b\n\t"

// the last step of an odd-length run: its s is posted alone, behind the chain wave's loop
#define MFSGD_SOLO_HELPER_TAIL(TAG, M0, M1) \
        "8" TAG ":\n\t" \
        "s_cmp_eq_u32 %[n], 0\n\t" \
        "s_cbranch_scc1 40f\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_pk_mul_f32 v[126:127], v[104:105], %[c2]\n\t" \
        "v_pk_mul_f32 v[128:129], v[106:107], %[c2]\n\t" \
        "v_pk_mul_f32 v[122:123], v[100:101], %[c2]\n\t" \
        "v_pk_mul_f32 v[124:125], v[102:103], %[c2]\n\t" \
        "3" TAG ":\n\t" \
        "v_cmp_eq_u32 vcc, -1, v" M1 "\n\t" \
        "s_cbranch_vccz 2" TAG "f\n\t" \
        "s_sleep 1\n\t" \
        "ds_read_b32 v" M1 ", v138 offset:4\n\t" \
        "s_sub_u32 %[spins], %[spins], 1\n\t" \
        "s_cmp_eq_u32 %[spins], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "s_branch 3" TAG "b\n\t" \
        "2" TAG ":\n\t" \
        "v_pk_fma_f32 v[134:135], v[" M0 ":" M1 "], v[100:101], v[126:127] op_sel:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[136:137], v[" M0 ":" M1 "], v[102:103], v[128:129] op_sel:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[100:101], v[" M0 ":" M1 "], v[104:105], v[122:123] op_sel:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[102:103], v[" M0 ":" M1 "], v[106:107], v[124:125] op_sel:[1,0,0]\n\t" \
        "s_nop 0\n\t" \
        "ds_write_b128 v112, v[134:137]\n\t"

#define MFSGD_SOLO_HELPER_ASM_TEXT \
        "v_mov_b32 v141, %[ea]\n\t" \
        "v_mov_b32 v139, %[rb]\n\t" \
        "ds_read_b32 v133, v141\n\t" \
        "v_add_u32 v138, 16, v141\n\t" \
        "3:\n\t" \
        "ds_read2_b64 v[116:119], v138 offset1:2\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_cmp_eq_u32 vcc, -1, v117\n\t" \
        "s_cbranch_vccz 4f\n\t" \
        "s_sleep 1\n\t" \
        "s_sub_u32 %[spins], %[spins], 1\n\t" \
        "s_cmp_eq_u32 %[spins], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "s_branch 3b\n\t" \
        "4:\n\t" \
        "v_mad_u32_u16 v112, v133, 16, v139\n\t" \
        "v_bfe_u32 v140, v133, 16, 15\n\t" \
        "v_lshl_add_u32 v140, v140, 4, v139\n\t" \
        "ds_read_b128 v[100:103], v140\n\t" \
        "ds_read_b128 v[104:107], v112\n\t" \
        "ds_write_b32 v141, v133\n\t" \
        "s_cmp_lt_u32 %[n], 2\n\t" \
        "s_cbranch_scc1 80f\n\t" \
        MFSGD_LOOP_ALIGN \
        "5:\n\t" \
        MFSGD_SOLO_HELPER_PAIR("0", "116", "117", "118", "119", "150", "153") \
        "s_cbranch_scc1 81f\n\t" \
        MFSGD_SOLO_HELPER_PAIR("1", "150", "151", "152", "153", "116", "119") \
        "s_cbranch_scc0 5b\n\t" \
        MFSGD_SOLO_HELPER_TAIL("0", "116", "117") \
        "s_branch 40f\n\t" \
        MFSGD_SOLO_HELPER_TAIL("1", "150", "151") \
        "40:\n\t" \
        "s_cmp_eq_u32 %[fin], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "ds_write_b128 v140, v[100:103]\n\t" \
        "s_branch 9f\n\t" \
        MFSGD_SOLO_HELPER_SLOW("0", "116", "117", "119") \
        MFSGD_SOLO_HELPER_SLOW("1", "150", "151", "153") \
        "9:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t"

#endif
#define MFSGD_SOLO_HELPER_OPERANDS                                                                                     \
    : [n] "+s"(n), [spins] "+s"(spins)                                                                                 \
    : [ea] "v"(ea), [rb] "v"(rowbase), [c2] "s"(c2), [fin] "s"(fin), [pad] "n"(PADV)                                                                 \
    : "memory", "scc", "vcc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110",  \
      "v111", "v112", "v113", "v116", "v117", "v118", "v119", "v122", "v123", "v124", "v125", "v126", "v127", "v128",  \
      "v129", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v150", "v151", "v152", "v153"
