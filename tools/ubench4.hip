// How long does a wave take to ISSUE a 16-byte-per-lane load -- from LDS (ds_read_b128) and from global memory
// (global_load_dwordx4, L2-resident rows) -- when nothing waits for the data?  12 loads back to back (below both
// counters' limits), s_memtime around the issue only; the wait for the data is outside the timed region.
// Background: DESIGN.md section 8 ("what comes next", 1).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__global__ void __launch_bounds__(64) k(const float* g, unsigned long long* out, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 16];
    for (int x = threadIdx.x; x < 64 * 4 * 16; x += 64) lds[x] = g[x];
    __syncthreads();
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (threadIdx.x % 16) * 16;
    const float* ga = g + (threadIdx.x % 16) * 4;
    unsigned long long t0, t1;
    float acc = 0.f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 r[12];
    for (int rep = 0; rep < 4; ++rep) {  // the last repetition is the one reported (instruction cache warm)
        if (mode == 0) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(r[j]) : "v"(la), "n"(1024 * 0 + 256 * 1) : "memory");
            asm volatile("s_memtime %0" : "=s"(t1)::"memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("global_load_dwordx4 %0, %1, off offset:%c2" : "=v"(r[j]) : "v"(ga), "n"(256) : "memory");
            asm volatile("s_memtime %0" : "=s"(t1)::"memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 12; ++j) acc += r[j][0] + r[j][3];
    }
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = (unsigned long long)acc;
    }
}

int main() {
    float* g;
    unsigned long long *o, h[2];
    (void)hipMalloc(&g, 64 * 4 * 16 * 4 + 4096);
    (void)hipMemset(g, 0, 64 * 4 * 16 * 4 + 4096);
    (void)hipMalloc(&o, 16);
    for (int mode = 0; mode < 2; ++mode) {
        unsigned long long best = ~0ull;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, mode);
            (void)hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
            if (h[0] < best) best = h[0];
        }
        printf("%s: %.1f cycles per instruction to issue (12 back to back, incl. one s_memtime)\n",
               mode == 0 ? "ds_read_b128      " : "global_load_dwordx4", (double)best / 12);
    }
    return 0;
}
