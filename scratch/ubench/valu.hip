// microbenchmark: issue cost of common VALU ops on gfx950 at 1/2/4/8 waves per SIMD, and the sustained shader clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND> __global__ void k(float *out, long long *clk, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    long long t0 = __builtin_amdgcn_s_memtime();
    long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {  // 8 independent v_fma_f32
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
            a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
        } else if constexpr (KIND == 1) {  // 4 independent v_pk_fma_f32 (8 results)
            p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc);
            p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc);
        } else if constexpr (KIND == 2) {  // dependent chain of v_fma_f32
            a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c);
            a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c);
        } else if constexpr (KIND == 3) {  // cmp + cndmask pairs
            a0 = a0 > c ? a1 : a0 + 1.f; a1 = a1 > c ? a2 : a1 + 1.f; a2 = a2 > c ? a3 : a2 + 1.f; a3 = a3 > c ? a0 : a3 + 1.f;
        } else if constexpr (KIND == 4) {  // 8 x v_rsq_f32
            a0 = __builtin_amdgcn_rsqf(a0); a1 = __builtin_amdgcn_rsqf(a1); a2 = __builtin_amdgcn_rsqf(a2); a3 = __builtin_amdgcn_rsqf(a3);
            a4 = __builtin_amdgcn_rsqf(a4); a5 = __builtin_amdgcn_rsqf(a5); a6 = __builtin_amdgcn_rsqf(a6); a7 = __builtin_amdgcn_rsqf(a7);
        } else if constexpr (KIND == 5) {  // 8 x v_mul_lo_u32
            unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3);
            unsigned u4 = __float_as_uint(a4), u5 = __float_as_uint(a5), u6 = __float_as_uint(a6), u7 = __float_as_uint(a7);
            u0 *= 0x7feb352du; u1 *= 0x7feb352du; u2 *= 0x7feb352du; u3 *= 0x7feb352du; u4 *= 0x7feb352du; u5 *= 0x7feb352du; u6 *= 0x7feb352du; u7 *= 0x7feb352du;
            a0 = __uint_as_float(u0 | 1); a1 = __uint_as_float(u1 | 1); a2 = __uint_as_float(u2 | 1); a3 = __uint_as_float(u3 | 1);
            a4 = __uint_as_float(u4 | 1); a5 = __uint_as_float(u5 | 1); a6 = __uint_as_float(u6 | 1); a7 = __uint_as_float(u7 | 1);
        } else if constexpr (KIND == 7) {  // two dependent chains interleaved (a0 a1 a0 a1 ...)
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c);
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c);
        } else if constexpr (KIND == 8) {  // four dependent chains interleaved
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
        } else if constexpr (KIND == 9) {  // two chains, in pairs (a0 a0 a1 a1 ...): a dependent neighbour every other instruction
            a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a1 = __builtin_fmaf(a1, b, c);
            a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a1 = __builtin_fmaf(a1, b, c);
        } else if constexpr (KIND == 10) {  // ONE dependent chain of v_pk_fma_f32 (4 per iteration = 8 results)
            p0 = __builtin_elementwise_fma(p0, pb, pc); p0 = __builtin_elementwise_fma(p0, pb, pc);
            p0 = __builtin_elementwise_fma(p0, pb, pc); p0 = __builtin_elementwise_fma(p0, pb, pc);
        } else if constexpr (KIND == 11) {  // the same 8 results as two dependent chains of plain v_fma_f32 (x chain, y chain)
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c);
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c);
        } else if constexpr (KIND == 6) {  // 8 x v_mov_dpp + add
            a0 += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(a1), 0x39, 0xF, 0xF, true));
            a1 += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(a2), 0x39, 0xF, 0xF, true));
            a2 += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(a3), 0x39, 0xF, 0xF, true));
            a3 += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(a0), 0x39, 0xF, 0xF, true));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND> int run(const char *name, int ops_per_iter, float *out, long long *clk) {
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;  // 256 threads = 4 waves = 1 per SIMD per block; wps blocks per CU
        int iters = 20000;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, 100);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        double ghz = (double)h[0] / ((double)h[1] * 10.0);  // memrealtime = 100 MHz
        double cyc_per_inst = (double)h[0] / ((double)iters * ops_per_iter * wps);
        printf("%-26s waves/SIMD=%d  %.3f ms  shader clock %.2f GHz  SIMD cycles per wave-instruction %.2f\n", name, wps, ms, ghz, cyc_per_inst);
    }
    return 0;
}

int main() {
    float *out; long long *clk;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float))); CHECK(hipMalloc(&clk, 16));
    run<0>("v_fma_f32 x8 indep", 8, out, clk);
    run<1>("v_pk_fma_f32 x4 indep", 4, out, clk);
    run<2>("v_fma_f32 x8 dependent", 8, out, clk);
    run<3>("cmp+cndmask+add x4", 12, out, clk);
    run<4>("v_rsq_f32 x8", 8, out, clk);
    run<5>("v_mul_lo_u32 x8 (+or)", 16, out, clk);
    run<6>("mov_dpp+add x4", 8, out, clk);
    run<10>("v_pk_fma_f32 x4 ONE chain", 4, out, clk);
    run<11>("same 8 results: 2 fma chains", 8, out, clk);
    run<7>("v_fma_f32 x8, 2 chains abab", 8, out, clk);
    run<8>("v_fma_f32 x8, 4 chains abcd", 8, out, clk);
    run<9>("v_fma_f32 x8, 2 chains aabb", 8, out, clk);
    return 0;
}
