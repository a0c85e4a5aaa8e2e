// Microbenchmark + correctness check of the SOLO run loops (csrc/run_asm.hpp): one workgroup of
// two waves on two SIMDs, wave 0 = chain wave, wave 1 = helper.  Prints cycles per step of the pair
// (and of the chain wave alone, MODE=1: no helper, nothing stored) and compares every p row and the
// q row with a plain host restatement of DESIGN.md section 3.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../matrixfactorizationsgd.java_amd/csrc/run_asm.hpp"

#ifndef LG
#define LG 16  // lanes per rating
#endif
#define SWAP16 "v_mov_b32 v133, v132\n\ts_nop 1\n\tv_permlane16_swap_b32 v132, v133\n\ts_nop 1\n\tv_add_f32 v132, v132, v133\n\t"
#define SWAP32 "v_mov_b32 v133, v132\n\ts_nop 1\n\tv_permlane32_swap_b32 v132, v133\n\ts_nop 1\n\tv_add_f32 v132, v132, v133\n\t"
#define SFMA MFSGD_SFMA_V
#define SFMA2 MFSGD_SFMA2_V
#if LG == 16
#define EXTRA ""
#elif LG == 32
#define EXTRA SWAP16
#ifndef OLD32
#define EXTRA_SOLO MFSGD_BCAST_ADD32  // the solo chain: one rating per wave
#undef SFMA2
#define SFMA2 MFSGD_SFMA2_S
#endif
#else
#ifdef OLD64
#define EXTRA SWAP16 SWAP32
#else
#define EXTRA MFSGD_BCAST_ADD64
#undef SFMA
#define SFMA MFSGD_SFMA_S
#undef SFMA2
#define SFMA2 MFSGD_SFMA2_S
#endif
#endif

#ifndef EXTRA_SOLO
#define EXTRA_SOLO EXTRA
#endif

constexpr int ROWB = 16 * LG;
constexpr int NSTEP = (LG == 64 ? 120 : LG == 32 ? 200 : 300), NROWS = NSTEP + 2;  // p rows 0..NSTEP-1, q row NSTEP, zero row NSTEP+1
constexpr int ENT_OFF = NROWS * ROWB;          // entries behind the rows
constexpr int GS = 64 / LG;                    // slots per step of the (old) run loop
constexpr int RUN_OFF = ENT_OFF + (NSTEP + 2) * 16;  // mode 3: run-loop entries, NSTEP + 2 steps x GS x 16 bytes
constexpr int EXTRA_ZERO = NROWS;              // (mode 3 idle slots use the zero row too)

__global__ void __launch_bounds__(192) k(const float* rows_in, const uint32_t* ent_in, float* rows_out, uint32_t* ent_out,
                                         unsigned long long* cyc, int n_steps, float lr, float c, int mode,
                                         const uint32_t* run_in, int first_row) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int x = threadIdx.x; x < NROWS * ROWB / 4; x += blockDim.x) ((float*)smem)[x] = rows_in[x];
    for (int x = threadIdx.x; x < (NSTEP + 2) * 4; x += blockDim.x) ((uint32_t*)(smem + ENT_OFF))[x] = ent_in[x];
    for (int x = threadIdx.x; x < (NSTEP + 2) * GS * 4; x += blockDim.x) ((uint32_t*)(smem + RUN_OFF))[x] = run_in[x];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned rowbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem + (lane % LG) * 16;
    const unsigned ea = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(smem + ENT_OFF);
    const uint64_t c2 = ((uint64_t)__builtin_bit_cast(uint32_t, c) << 32) | __builtin_bit_cast(uint32_t, c);
    unsigned long long t0 = 0, t1 = 0;
    int n = n_steps;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (wave == 0 && mode == 3) {
        // the one-wave run loop (kernels.hip run_loop_asm): slot 0 carries the chain, the others idle
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int g = lane / LG;
        const unsigned qaddr = (g == 0 ? NSTEP : NSTEP + 1) * ROWB + (lane % LG) * 16;
        f4 q = *(const f4*)(smem + qaddr);
        const unsigned ea = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(smem + RUN_OFF) + g * 16;
        int pairs = n_steps / 2;
        constexpr int EST = GS * 16;
        constexpr int PADV = mfsgd_pad_run(LG);
        asm volatile(MFSGD_RUN_LOOP_ASM_TEXT(EXTRA, SFMA) MFSGD_RUN_LOOP_ASM_OPERANDS);
        if (g == 0) *(f4*)(smem + qaddr) = q;
    } else if (wave == 0 && mode == 4) {
        // cut run: the chain wave stores q into its row between the halves (kernels.hip, solo_split)
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 q = *(const f4*)(smem + NSTEP * ROWB + (lane % LG) * 16);
        constexpr int PADV = mfsgd_pad_chain(LG);
        const int m = (n_steps / 2) & ~1;
        const unsigned ea0 = ea;
        n = m;
        asm volatile(MFSGD_SOLO_CHAIN_ASM_TEXT(EXTRA_SOLO, SFMA2) MFSGD_SOLO_CHAIN_OPERANDS);
        *(f4*)(smem + NSTEP * ROWB + (lane % LG) * 16) = q;
        {
            const unsigned ea = ea0 + m * 16;
            n = n_steps - m;
            asm volatile(MFSGD_SOLO_CHAIN_ASM_TEXT(EXTRA_SOLO, SFMA2) MFSGD_SOLO_CHAIN_OPERANDS);
        }
    } else if (wave == 0 && mode != 2) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        if (mode == 5) {
            // (mode 5 = mode 0 plus this; correctness only, the delay is inside the timed region)
            // what the product's chain wave does in front of a solo run: general steps of its sub-cell, which may update
            // p rows the run is about to use -- here: p row of step 0 += 1, late enough for a helper that reads rows
            // before s_0 is posted to have read the old one (the host reference starts from the updated row)
            for (int d = 0; d < 40; ++d) __builtin_amdgcn_s_sleep(64);
            f4* p0 = (f4*)(smem + first_row * ROWB + (lane % LG) * 16);
            f4 v = *p0;
            v += 1.0f;
            *p0 = v;
        }
        f4 q = *(const f4*)(smem + NSTEP * ROWB + (lane % LG) * 16);
        constexpr int PADV = mfsgd_pad_chain(LG);
        asm volatile(MFSGD_SOLO_CHAIN_ASM_TEXT(EXTRA_SOLO, SFMA2) MFSGD_SOLO_CHAIN_OPERANDS);
    } else if (wave == 1 && (mode == 0 || mode == 2 || mode == 5)) {
        int spins = 1 << 20, fin = 1;
        asm volatile("" : "+s"(fin));
        constexpr int PADV = mfsgd_pad_helper(LG);
        asm volatile(MFSGD_SOLO_HELPER_ASM_TEXT MFSGD_SOLO_HELPER_OPERANDS);
        if (spins == 0 && lane == 0) cyc[2] = 1;
    } else if ((wave == 1 || wave == 2) && mode == 4) {
        // the two helpers of a cut run: wave 1 follows [0, m) and leaves q alone, wave 2 follows [m, n)
        const int m = (n_steps / 2) & ~1;
        int spins = 1 << 20, fin = wave == 2;
        fin = __builtin_amdgcn_readfirstlane(fin);
        asm volatile("" : "+s"(fin));
        const unsigned ea0 = ea;
        {
            const unsigned ea = wave == 1 ? ea0 : ea0 + m * 16;
            n = __builtin_amdgcn_readfirstlane(wave == 1 ? m : n_steps - m);
            constexpr int PADV = mfsgd_pad_helper(LG);
            asm volatile(MFSGD_SOLO_HELPER_ASM_TEXT MFSGD_SOLO_HELPER_OPERANDS);
        }
        if (spins == 0 && lane == 0) cyc[2] = 1;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) cyc[wave == 2 ? 3 : wave] = t1 - t0;
    __syncthreads();
    for (int x = threadIdx.x; x < NROWS * ROWB / 4; x += blockDim.x) rows_out[x] = ((float*)smem)[x];
    for (int x = threadIdx.x; x < (NSTEP + 2) * 4; x += blockDim.x) ent_out[x] = ((uint32_t*)(smem + ENT_OFF))[x];
}

static float ref_dot(const float* p, const float* q) {
    float s[64];
    for (int c = 0; c < LG; ++c) {
        float t0 = p[4 * c] * q[4 * c], t1 = p[4 * c + 1] * q[4 * c + 1];
        t0 = fmaf(p[4 * c + 2], q[4 * c + 2], t0);
        t1 = fmaf(p[4 * c + 3], q[4 * c + 3], t1);
        s[c] = t0 + t1;
    }
    for (int m = 1; m < LG; m <<= 1) {
        float t[64];
        for (int c = 0; c < LG; ++c) t[c] = s[c] + s[c ^ m];
        memcpy(s, t, sizeof(float) * LG);
    }
    return s[0];
}

int main() {
    const int KP = 4 * LG;
    std::vector<float> rows((size_t)NROWS * KP), ref;
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) / (float)(1 << 24); };
    for (int r = 0; r <= NSTEP; ++r)
        for (int f = 0; f < KP; ++f) rows[(size_t)r * KP + f] = rnd() * 0.25f;
    for (int f = 0; f < KP; ++f) rows[(size_t)(NSTEP + 1) * KP + f] = 0.f;
    const float lr = 0.01f, c = 1.0f - 0.01f * 0.05f;
    // steps visit the p rows in a scrambled order
    std::vector<int> prow(NSTEP);
    for (int t = 0; t < NSTEP; ++t) prow[t] = (t * 7) % NSTEP;
    std::vector<float> rr(NSTEP);
    std::vector<uint32_t> ent((NSTEP + 2) * 4, 0);
    auto slots = [&](int pr) { return (uint32_t)(pr * LG) | ((uint32_t)(NSTEP * LG) << 16); };
    // entry t = {slots_{t+1}, mailbox_t, lr * r_t, r_t}; header = {slots_0, 0, 0, 0} (run_asm.hpp)
    ent[0] = slots(prow[0]);
    for (int t = 0; t < NSTEP; ++t) {
        rr[t] = 1.0f + 4.0f * rnd();
        const float lrr = lr * rr[t];
        memcpy(&ent[(t + 1) * 4 + 2], &lrr, 4);
        ent[(t + 1) * 4 + 0] = t + 1 < NSTEP ? slots(prow[t + 1]) : slots(NSTEP + 1);
        ent[(t + 1) * 4 + 1] = 0xFFFFFFFFu;
        memcpy(&ent[(t + 1) * 4 + 3], &rr[t], 4);
    }
    ent[(NSTEP + 1) * 4 + 0] = slots(NSTEP + 1);
    ent[(NSTEP + 1) * 4 + 1] = 0xFFFFFFFFu;
    // host restatement
    ref = rows;
    float* q = &ref[(size_t)NSTEP * KP];
    for (int t = 0; t < NSTEP; ++t) {
        float* p = &ref[(size_t)prow[t] * KP];
        const float dot = ref_dot(p, q), s = fmaf(-lr, dot, lr * rr[t]);
        for (int f = 0; f < KP; ++f) {
            const float po = p[f], qo = q[f];
            p[f] = fmaf(s, qo, c * po);
            q[f] = fmaf(s, po, c * qo);
        }
    }
    float *d_in, *d_out;
    uint32_t *d_ent, *d_ent_out;
    unsigned long long* d_cyc;
    (void)hipMalloc(&d_in, rows.size() * 4);
    (void)hipMalloc(&d_out, rows.size() * 4);
    (void)hipMalloc(&d_ent, ent.size() * 4);
    (void)hipMalloc(&d_ent_out, ent.size() * 4);
    (void)hipMalloc(&d_cyc, 32);
    (void)hipMemcpy(d_in, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_ent, ent.data(), ent.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)NROWS * ROWB + (NSTEP + 2) * 16 + (size_t)(NSTEP + 2) * GS * 16;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // mode 2: mailboxes pre-filled with the reference's s_t -> the helper's own pace
    std::vector<uint32_t> ent2 = ent;
    {
        std::vector<float> r2 = rows;
        float* qq = &r2[(size_t)NSTEP * KP];
        for (int t = 0; t < NSTEP; ++t) {
            float* p = &r2[(size_t)prow[t] * KP];
            const float dot = ref_dot(p, qq), s = fmaf(-lr, dot, lr * rr[t]);
            memcpy(&ent2[(t + 1) * 4 + 1], &s, 4);
            for (int f = 0; f < KP; ++f) {
                const float po = p[f], qo = qq[f];
                p[f] = fmaf(s, qo, c * po);
                qq[f] = fmaf(s, po, c * qo);
            }
        }
    }
    // mode 3: the same chain as run-loop entries {slots, r, lr*r, ce}, slot 0 live, other slots idle
    std::vector<uint32_t> run((NSTEP + 2) * GS * 4, 0);
    for (int t = 0; t < NSTEP + 2; ++t)
        for (int g = 0; g < GS; ++g) {
            uint32_t* e = &run[(size_t)(t * GS + g) * 4];
            const float one = 1.0f;
            if (g == 0 && t < NSTEP) {
                const float lrr = lr * rr[t];
                e[0] = slots(prow[t]);
                memcpy(&e[1], &rr[t], 4);
                memcpy(&e[2], &lrr, 4);
                memcpy(&e[3], &c, 4);
            } else {
                e[0] = (uint32_t)((NSTEP + 1) * LG) | ((uint32_t)((g == 0 ? NSTEP : NSTEP + 1) * LG) << 16) | 0x80000000u;
                memcpy(&e[3], &one, 4);
            }
        }
    uint32_t* d_run;
    (void)hipMalloc(&d_run, run.size() * 4);
    (void)hipMemcpy(d_run, run.data(), run.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 6; ++mode)
        for (int n : {NSTEP, NSTEP - 1, 1, 2, 50, 51}) {
            if ((mode == 3 && (n & 1)) || (mode == 4 && n < 8)) continue;
            (void)hipMemcpy(d_ent, (mode == 2 ? ent2 : ent).data(), ent.size() * 4, hipMemcpyHostToDevice);
            unsigned long long best[2] = {~0ull, ~0ull}, h[4];
            std::vector<float> out(rows.size());
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipMemset(d_cyc, 0, 32);
                hipLaunchKernelGGL(k, dim3(1), dim3(mode == 4 ? 192 : 128), lds, 0, d_in, d_ent, d_out, d_ent_out, d_cyc, n, lr, c, mode, d_run, prow[0]);
                (void)hipMemcpy(h, d_cyc, 32, hipMemcpyDeviceToHost);
                if (h[2]) printf("helper gave up!\n");
                if (mode == 4 && h[3] > h[1]) h[1] = h[3];  // cut run: the later of the two helpers
                for (int w = 0; w < 2; ++w) best[w] = h[w] < best[w] ? h[w] : best[w];
            }
            (void)hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            if (mode != 1) {  // the reference for n steps (mode 0: p row of step 0 was updated in front of the run)
                std::vector<float> r2 = rows;
                if (mode == 5)
                    for (int f = 0; f < KP; ++f) r2[(size_t)prow[0] * KP + f] += 1.0f;
                float* qq = &r2[(size_t)NSTEP * KP];
                for (int t = 0; t < n; ++t) {
                    float* p = &r2[(size_t)prow[t] * KP];
                    const float dot = ref_dot(p, qq), s = fmaf(-lr, dot, lr * rr[t]);
                    for (int f = 0; f < KP; ++f) {
                        const float po = p[f], qo = qq[f];
                        p[f] = fmaf(s, qo, c * po);
                        qq[f] = fmaf(s, po, c * qo);
                    }
                }
                bad = memcmp(out.data(), r2.data(), out.size() * 4) != 0;
            }
            printf("L=%d mode=%d (%s) n=%d: chain %.1f cycles/step, helper %.1f cycles/step%s\n", LG, mode,
                   mode == 1 ? "chain alone" : mode == 4 ? "cut run: chain + two helpers" : mode == 2 ? "helper alone" : mode == 3 ? "one-wave run loop" : mode == 5 ? "chain + helper, p row of step 0 updated in front of the run" : "chain + helper", n, (double)best[0] / n, (double)best[1] / n,
                   mode == 1 ? "" : (bad ? "  MISMATCH" : "  bit-exact"));
        }
    return 0;
}
