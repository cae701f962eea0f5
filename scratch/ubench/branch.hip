// microbenchmark: what a wave-level decision costs on gfx950: v_cmp -> branch on VCC (taken / not taken), scalar-only
// branches, branch-free scalar selects. Reported as shader clocks per pattern for 1 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(X) X X X X X X X X
template <int KIND> __global__ void k(float *out, long long *clk, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, big = 1e30f, acc = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {
            asm volatile(REP8("v_add_f32 %0, %0, %1\n\t") : "+v"(acc) : "v"(a));
        } else if constexpr (KIND == 1) {  // v_cmp + branch on vcc, never taken
            asm volatile(REP8("v_cmp_gt_f32 vcc, %1, %2\n\ts_cbranch_vccnz 1f\n\tv_add_f32 %0, %0, %1\n\t1:\n\t") : "+v"(acc) : "v"(a), "v"(big) : "vcc");
        } else if constexpr (KIND == 2) {  // v_cmp + branch on vcc, always taken (skips one instruction)
            asm volatile(REP8("v_cmp_lt_f32 vcc, %1, %2\n\ts_cbranch_vccnz 1f\n\tv_add_f32 %0, %0, %1\n\t1:\n\tv_add_f32 %0, %0, %1\n\t") : "+v"(acc) : "v"(a), "v"(big) : "vcc");
        } else if constexpr (KIND == 3) {  // v_cmp + scalar mask logic + scc branch, never taken
            asm volatile(REP8("v_cmp_gt_f32 vcc, %1, %2\n\ts_and_b64 s[40:41], vcc, exec\n\ts_cmp_lg_u64 s[40:41], 0\n\ts_cbranch_scc1 1f\n\tv_add_f32 %0, %0, %1\n\t1:\n\t")
                         : "+v"(acc) : "v"(a), "v"(big) : "vcc", "scc", "s40", "s41");
        } else if constexpr (KIND == 4) {  // scalar compare + taken branch
            asm volatile(REP8("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\tv_add_f32 %0, %0, %1\n\t1:\n\tv_add_f32 %0, %0, %1\n\t") : "+v"(acc) : "v"(a) : "scc");
        } else if constexpr (KIND == 5) {  // v_cmp + branch-free scalar select
            asm volatile(REP8("v_cmp_gt_f32 vcc, %1, %2\n\ts_cmp_eq_u64 vcc, 0\n\ts_cselect_b32 s40, 16, 6\n\tv_add_f32 %0, %0, %1\n\t") : "+v"(acc) : "v"(a), "v"(big) : "vcc", "scc", "s40");
        } else if constexpr (KIND == 6) {  // scalar compare + not-taken branch
            asm volatile(REP8("s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 1f\n\tv_add_f32 %0, %0, %1\n\t1:\n\t") : "+v"(acc) : "v"(a) : "scc");
        } else if constexpr (KIND == 7) {  // 3 x v_cmp into SGPR pairs + one s_or + one vcc branch (a batched decision)
            asm volatile(REP8("v_cmp_gt_f32 s[40:41], %1, %2\n\tv_cmp_gt_f32 s[42:43], %1, %2\n\tv_cmp_gt_f32 vcc, %1, %2\n\ts_or_b64 s[40:41], s[40:41], s[42:43]\n\t"
                              "s_or_b64 vcc, vcc, s[40:41]\n\ts_cbranch_vccnz 1f\n\tv_add_f32 %0, %0, %1\n\t1:\n\t")
                         : "+v"(acc) : "v"(a), "v"(big) : "vcc", "scc", "s40", "s41", "s42", "s43");
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int KIND> int run(const char *name, float *out, long long *clk) {
    for (int wps : {1, 4}) {
        int blocks = 256 * wps, iters = 20000;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, 100);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
        CHECK(hipDeviceSynchronize());
        long long h; CHECK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
        printf("%-52s waves/SIMD=%d  clocks per pattern (one wave's view) %.1f\n", name, wps, (double)h / ((double)iters * 8));
    }
    return 0;
}

int main() {
    float *out; long long *clk;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float))); CHECK(hipMalloc(&clk, 16));
    run<0>("v_add (baseline, 1 instr)", out, clk);
    run<1>("v_cmp + s_cbranch_vccnz not taken + v_add", out, clk);
    run<2>("v_cmp + s_cbranch_vccnz TAKEN + v_add", out, clk);
    run<3>("v_cmp + s_and + s_cmp + s_cbranch_scc1 n.t. + v_add", out, clk);
    run<4>("s_cmp + s_cbranch_scc1 TAKEN + v_add", out, clk);
    run<5>("v_cmp + s_cmp_eq_u64 vcc + s_cselect + v_add", out, clk);
    run<6>("s_cmp + s_cbranch_scc1 not taken + v_add", out, clk);
    run<7>("3 v_cmp + 2 s_or + s_cbranch_vccnz n.t. + v_add", out, clk);
    return 0;
}
