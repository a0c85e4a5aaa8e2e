// SUPERSEDED by ubench3.hip (round 2), which times the product's own loop texts (csrc/run_asm.hpp).  The
// body below is NOT the product's run loop: it copies the prefetched p row with four v_mov instead of
// alternating register sets, which is why it reads 184 cycles per step where the product loop takes 146.
// Kept for its knock-out variants (what a group of instructions costs), which are ratios, not absolutes.
//
// Microbenchmark of the hand-scheduled run step (one wave alone on its SIMD): cycles per
// step for the full body (V=0) and with groups of instructions knocked out (timing only).
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef V
#define V 0
#endif
#ifndef XMASK
#define XMASK 0xffffffffffffffffull
#endif
#define DPPI(c) "v_add_f32_dpp v132, v132, v132 " c " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#if V == 1
#define DPP(c) "s_nop 0\n\t"
#else
#define DPP(c) DPPI(c)
#endif
#if V == 2 || V == 5
#define LDS(x) ""
#else
#define LDS(x) x
#endif
#if V == 3 || V == 5
#define SCALE(x) ""
#elif V == 12
#define SCALE(x) x
#else
#define SCALE(x) x
#endif
#if V == 4 || V == 5 || V == 12
#define PUPD(x) ""
#else
#define PUPD(x) x
#endif
#if V == 5 || V == 6
#define MISC(x) ""
#else
#define MISC(x) x
#endif
__global__ void k(unsigned long long* cyc, float* out, int iters, float lr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int x = threadIdx.x; x < 8192; x += 64) ((float*)smem)[x] = 0.001f * (x & 255);
    for (int x = threadIdx.x; x < 4096; x += 64) ((unsigned*)(smem + 32768))[x] = (x % 4 == 0) ? ((x / 4 * 7) % 100) * 16 : 0x3c000000;
    __syncthreads();
    unsigned long long t0, t1;
    int n = iters;
    const unsigned lane16 = (threadIdx.x & 15) * 16, ea = 32768 + (threadIdx.x >> 4) * 16;
    float o;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile("s_mov_b64 exec, %0\n\ts_nop 4" ::"s"(XMASK));
    asm volatile(
        "v_mov_b32 v138, %[ea]\n\tv_mov_b32 v139, %[rb]\n\tv_mov_b32 v115, 0\n\tv_mov_b32 v112, %[rb]\n\tv_mov_b32 v113, %[rb]\n\t"
        "v_mov_b32 v100, 1.0\n\tv_mov_b32 v101, 1.0\n\tv_mov_b32 v102, 1.0\n\tv_mov_b32 v103, 1.0\n\t"
        "v_mov_b32 v116, 0\n\tv_mov_b32 v117, 1.0\n\tv_mov_b32 v131, 0\n\t"
        "v_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\tv_mov_b32 v124, 0\n\tv_mov_b32 v125, 0\n\tv_mov_b32 v126, 0\n\tv_mov_b32 v127, 0\n\tv_mov_b32 v128, 0\n\tv_mov_b32 v129, 0\n\t"
        "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t"
        "ds_read_b128 v[104:107], v112\n\tds_write_b32 v138, v115 offset:4\n\t"
        "1:\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_pk_mul_f32 v[120:121], v[104:105], v[100:101]\n\t"
        "v_pk_fma_f32 v[120:121], v[106:107], v[102:103], v[120:121]\n\t"
        MISC("v_and_b32 v133, 0xffff, v115\n\t")
        "v_add_f32 v132, v120, v121\n\t"
        MISC("v_lshl_add_u32 v113, v133, 4, v139\n\t")
        SCALE("v_pk_mul_f32 v[122:123], v[116:117], v[100:101] op_sel:[1,0]\n\t")
        DPP("quad_perm:[1,0,3,2]")
        LDS("ds_read_b128 v[108:111], v113\n\t")
        SCALE("v_pk_mul_f32 v[124:125], v[116:117], v[102:103] op_sel:[1,0]\n\t")
        DPP("quad_perm:[2,3,0,1]")
#if V != 12
        SCALE("v_pk_mul_f32 v[126:127], v[116:117], v[104:105] op_sel:[1,0]\n\t")
        SCALE("v_pk_mul_f32 v[128:129], v[116:117], v[106:107] op_sel:[1,0]\n\t")
#else
        "s_nop 1\n\t"
#endif
        DPP("row_half_mirror")
#if V == 7
        "ds_read_b128 v[144:147], v138 offset:64\n\t"
        "s_nop 0\n\t"
#elif V == 8
        "ds_read_b64 v[116:117], v138 offset:8\n\t"
        "s_nop 0\n\t"
#else
        LDS("ds_read_b32 v115, v138 offset:64\n\t")
        LDS("ds_read_b64 v[116:117], v138 offset:8\n\t")
#endif
        DPP("row_mirror")
        "v_fma_f32 v130, -%[lr], v132, v116\n\t"
        "v_pk_fma_f32 v[100:101], v[130:131], v[104:105], v[122:123] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[102:103], v[130:131], v[106:107], v[124:125] op_sel_hi:[0,1,1]\n\t"
        PUPD("v_pk_fma_f32 v[134:135], v[130:131], v[100:101], v[126:127] op_sel_hi:[0,1,1]\n\t")
        PUPD("v_pk_fma_f32 v[136:137], v[130:131], v[102:103], v[128:129] op_sel_hi:[0,1,1]\n\t")
#if V == 12
        "s_mov_b64 exec, %[mask]\n\t"
        "ds_write_b32 v138, v130 offset:2048\n\t"
        "s_mov_b64 exec, -1\n\t"
#endif
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        LDS("v_mov_b32 v104, v108\n\tv_mov_b32 v105, v109\n\tv_mov_b32 v106, v110\n\tv_mov_b32 v107, v111\n\t")
#if V == 9
        "ds_write_b64 v112, v[134:135]\n\tds_write_b64 v112, v[136:137] offset:8\n\t"
#elif V == 10
        "ds_write_b32 v112, v134\n\tds_write_b32 v112, v135 offset:4\n\tds_write_b32 v112, v136 offset:8\n\tds_write_b32 v112, v137 offset:12\n\t"
#elif V == 11
        "ds_write2_b64 v112, v[134:135], v[136:137] offset1:1\n\t"
#else
        LDS(PUPD("ds_write_b128 v112, v[134:137]\n\t"))
#endif
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32 %[o], v100\n\t"
        : [n] "+s"(n), [o] "=v"(o)
        : [ea] "v"(ea), [rb] "v"(lane16), [lr] "s"(lr), [mask] "s"(0x0001000100010001ull)
        : "memory", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v144", "v145", "v146", "v147");
    asm volatile("s_mov_b64 exec, -1\n\ts_nop 4");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[threadIdx.x] = o;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    unsigned long long* c; float* o; hipMalloc(&c, 8); hipMalloc(&o, 256);
    const int iters = 4096; unsigned long long h = 0, best = ~0ull;
    for (int r = 0; r < 5; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 65536, 0, c, o, iters, 0.01f); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); if (h < best) best = h; }
    const char* names[] = {"full step", "DPP adds -> s_nop", "no LDS ops (+4 movs gone)", "no scale pk_mul x4", "no p' update + store", "pure dependent chain", "no address calc", "entry: one b128 read", "entry: one b64 read", "store as 2 x b64", "store as 4 x b32", "store as write2_b64", "chain wave of a two-wave split"};
    printf("V=%d mask=%llx %-28s %7.1f cycles/step\n", V, (unsigned long long)XMASK, names[V], (double)best / iters);
    return 0;
}
