// microbenchmark: does a wave64 vector instruction cost less on gfx950 when one 32-lane half of EXEC is empty?
// build: hipcc --offload-arch=gfx950 -O2 halfwave.hip -o halfwave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(X) X X X X X X X X
// MODE 0: all 64 lanes; 1: lanes 0..31 only; 2: lanes 32..63 only; 3: even lanes only (both halves half full)
template <int MODE, int KIND> __global__ void __launch_bounds__(256) k(float *out, long long *clk, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1.f, a1 = a0 + 1.f, b = 0.9999f, c = 0.5f;
    const int lane = threadIdx.x & 63;
    const bool on = MODE == 0 ? true : MODE == 1 ? lane < 32 : MODE == 2 ? lane >= 32 : (lane & 1) == 0;
    long long t0 = 0, t1 = 0;
    if (on) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            if constexpr (KIND == 0) asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(a0) : "v"(b), "v"(c));
            else {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 p0 = {a0, a1}, pb = {b, b}, pc = {c, c};
                asm volatile(REP8("v_pk_fma_f32 %0, %0, %1, %2\n\t") : "+v"(p0) : "v"(pb), "v"(pc));
                a0 = p0.x; a1 = p0.y;
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1;
    if (lane == (MODE == 2 ? 32 : 0)) clk[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE, int KIND> int run(const char *name, float *out, long long *clk) {
    for (int wps : {4, 8}) {
        const int blocks = 256 * wps, iters = 4000;
        hipLaunchKernelGGL((k<MODE, KIND>), dim3(blocks), dim3(256), 0, 0, out, clk, 50);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE, KIND>), dim3(blocks), dim3(256), 0, 0, out, clk, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> h(blocks * 4); CHECK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2] / (iters * 8.0);
        printf("%-40s waves/SIMD=%d  %.3f ms  clocks per instruction: per wavefront %.2f, per SIMD %.2f\n", name, wps, ms, med, med / wps);
    }
    return 0;
}
int main() {
    float *out; long long *clk;
    CHECK(hipMalloc(&out, (size_t)2048 * 256 * sizeof(float))); CHECK(hipMalloc(&clk, 2048 * 4 * 8));
    run<0, 0>("v_fma_f32, all 64 lanes", out, clk);
    run<1, 0>("v_fma_f32, lanes 0..31", out, clk);
    run<2, 0>("v_fma_f32, lanes 32..63", out, clk);
    run<3, 0>("v_fma_f32, even lanes", out, clk);
    run<0, 1>("v_pk_fma_f32, all 64 lanes", out, clk);
    run<1, 1>("v_pk_fma_f32, lanes 0..31", out, clk);
    run<3, 1>("v_pk_fma_f32, even lanes", out, clk);
    return 0;
}
