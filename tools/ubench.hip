// Microbenchmark: cycles per dependent VALU op for ONE wave alone on its SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dppmov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int N = 256;  // ops per timed region (unrolled)
template <int MODE> __global__ void k(float* out, unsigned long long* cyc, float a, float b) {
    float x = a + threadIdx.x, y = b;
    f2 px = {x, y}, pa = {a, b}, pb = {b, a};
    float z0 = x, z1 = y + 1, z2 = x + 2, z3 = y + 3;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define PIN(v) asm volatile("" : "+v"(v))
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (MODE == 0) { x = __builtin_fmaf(x, a, b); PIN(x); }
        if (MODE == 1) { px = __builtin_elementwise_fma(px, pa, pb); PIN(px); }
        if (MODE == 2) { x = x + dppmov<0xB1>(x); PIN(x); }
        if (MODE == 3) { z0 = __builtin_fmaf(z0, a, b); PIN(z0); z1 = __builtin_fmaf(z1, a, b); PIN(z1); z2 = __builtin_fmaf(z2, a, b); PIN(z2); z3 = __builtin_fmaf(z3, a, b); PIN(z3); }
        if (MODE == 4) { x = x + dppmov<0xB1>(x); PIN(x); x = x + dppmov<0x4E>(x); PIN(x); x = x + dppmov<0x141>(x); PIN(x); x = x + dppmov<0x140>(x); PIN(x); }
        if (MODE == 5) { x = x * a; PIN(x); }
        if (MODE == 6) { px = px * pa; PIN(px); }
        if (MODE == 7) { x = __builtin_fmaf(x, a, b); PIN(x); y = y * a; PIN(y); }
        if (MODE == 8) { px = __builtin_elementwise_fma(px, pa, pb); PIN(px); f2 q = {z0, z1}; q = __builtin_elementwise_fma(q, pa, pb); PIN(q); z0 = q.x; z1 = q.y; }
        if (MODE == 9) { x = (threadIdx.x & 1) ? x : y; PIN(x); y = y + x; PIN(y); }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[threadIdx.x] = x + y + px.x + px.y + z0 + z1 + z2 + z3;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int ops_per_iter) {
    float* o; unsigned long long* c; hipMalloc(&o, 256); hipMalloc(&c, 8);
    unsigned long long h = 0, best = ~0ull;
    for (int r = 0; r < 5; ++r) { k<MODE><<<1, 64>>>(o, c, 1.0001f, 0.5f); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); if (h < best) best = h; }
    printf("%-28s %6.2f cycles/op (%llu cycles / %d ops)\n", name, (double)best / (N * ops_per_iter), best, N * ops_per_iter);
    hipFree(o); hipFree(c);
}
int main() {
    run<0>("dependent v_fma_f32", 1);
    run<5>("dependent v_mul_f32", 1);
    run<1>("dependent v_pk_fma_f32", 1);
    run<6>("dependent v_pk_mul_f32", 1);
    run<2>("dependent v_add_f32_dpp", 1);
    run<4>("4-level dpp reduce (per op)", 4);
    run<3>("4 independent v_fma chains", 4);
    run<7>("1 dep fma + 1 indep mul", 2);
    run<8>("2 independent v_pk_fma chains", 2);
    run<9>("dependent cndmask + add", 2);
    return 0;
}
