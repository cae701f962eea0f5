// microbenchmark: issue cost of the double-precision instructions the parity kernels are made of, per wavefront, with 1 and 4
// wavefronts resident per SIMD (1024 / 4096 wavefronts of 64 lanes, blocks of 256). Clocks are s_memtime ticks as in simd.hip.
// build: hipcc --offload-arch=gfx950 -O2 f64.hip -o f64
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(X) X X X X X X X X

template <int KIND> __global__ void __launch_bounds__(256) k(double *out, long long *rec, int iters) {
    double a = threadIdx.x * 1e-3 + 1.0, b = 0.99999, c = 0.5;
    float f = (float)a;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) asm volatile(REP8("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
        else if constexpr (KIND == 1) asm volatile(REP8("v_add_f64 %0, %0, %1\n\t") : "+v"(a) : "v"(c));
        else if constexpr (KIND == 2) asm volatile(REP8("v_mul_f64 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
        else if constexpr (KIND == 3) asm volatile(REP8("v_cvt_f32_f64 %0, %1\n\t") : "+v"(f) : "v"(a));
        else if constexpr (KIND == 4) asm volatile(REP8("v_rcp_f64 %0, %0\n\t") : "+v"(a));
        else if constexpr (KIND == 5) asm volatile(REP8("v_rsq_f64 %0, %0\n\t") : "+v"(a));
        else if constexpr (KIND == 6) asm volatile(REP8("v_fma_f32 %0, %0, %0, %0\n\t") : "+v"(f));
        else if constexpr (KIND == 7) { a = b / a; a = b / a; a = b / a; a = b / a; a = b / a; a = b / a; a = b / a; a = b / a; asm volatile("" : "+v"(a)); }   // C division (8 per iteration)
        else if constexpr (KIND == 8) { for (int j = 0; j < 8; ++j) { a = sqrt(a + 3.0); asm volatile("" : "+v"(a)); } }                                       // C sqrt + 1 add
        else if constexpr (KIND == 9) asm volatile(REP8("v_cmp_lt_f64 vcc, %0, %1\n\t") : : "v"(a), "v"(b) : "vcc");
        else if constexpr (KIND == 10) asm volatile(REP8("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t") : "+v"(f));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a + f;
    if ((threadIdx.x & 63) == 0) rec[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> int run(const char *name, double *out, long long *rec) {
    printf("%-40s", name);
    for (int wps : {1, 4}) {
        const int waves = 1024 * wps, blocks = waves / 4, iters = 2000;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, rec, 50);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, rec, iters);
        CHECK(hipDeviceSynchronize());
        std::vector<long long> h(waves); CHECK(hipMemcpy(h.data(), rec, waves * sizeof(long long), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("  %d wave(s)/SIMD: %7.2f clk per op per wave", wps, (double)h[waves / 2] / (iters * 8.0));
    }
    printf("\n");
    return 0;
}
int main() {
    double *out; long long *rec;
    CHECK(hipMalloc(&out, (size_t)4096 * 64 * sizeof(double))); CHECK(hipMalloc(&rec, 4096 * sizeof(long long)));
    run<6>("v_fma_f32 (dependent)", out, rec);
    run<0>("v_fma_f64 (dependent)", out, rec);
    run<1>("v_add_f64 (dependent)", out, rec);
    run<2>("v_mul_f64 (dependent)", out, rec);
    run<3>("v_cvt_f32_f64", out, rec);
    run<9>("v_cmp_lt_f64", out, rec);
    run<4>("v_rcp_f64 (dependent)", out, rec);
    run<5>("v_rsq_f64 (dependent)", out, rec);
    run<10>("v_mov_b32_dpp + s_nop 1", out, rec);
    run<7>("double division a / b (C)", out, rec);
    run<8>("double sqrt(a + 3) (C)", out, rec);
    return 0;
}
