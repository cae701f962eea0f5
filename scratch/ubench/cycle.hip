// microbenchmark: the step kernel's quiet control cycle in isolation (one lane per mover, 4 lanes per env), variants of its
// instruction selection, at 4 and 8 wavefronts per SIMD. Reports shader clocks per cycle per wavefront and per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -I../../gymnasium-planar-robotics_amd/csrc -I../../include cycle.hip -o cycle
#include "gpr_device.h"
#include <chrono>
#include <type_traits>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace gpr;

// VAR 0: as in k_step (packed integrate, q, wall reach, DPP-sub pair sweep, min3/max3 accumulate)
// VAR 1: no pair sweep            VAR 2: no wall reach             VAR 3: integrate only
// VAR 4: pair sweep with v_mov_dpp + plain sub (compiler's choice)  VAR 5: non-packed integrate (separate x / y chains)
// VAR 6: pair sweep through ds_swizzle-free "bpermute"             VAR 7: VAR 0 with the two DPP asm blocks merged (one s_nop)
template <int VAR> __global__ void __launch_bounds__(256) k(float *out, long long *clk, int cycles, float dtx, float dty) {
    const float INF = __builtin_huge_valf();
    R2<float> dv{0.f, 0.f}, dp{0.f, 0.f}, v;
    const R2<float> v0{threadIdx.x * 1e-3f, threadIdx.x * -2e-3f}, q0{(threadIdx.x & 63) * 0.02f - 0.6f, (threadIdx.x >> 2) * 0.01f - 0.3f};
    R2<float> dtl{dtx * (1 + (threadIdx.x & 3)), dty};
    const float dt_l = 1e-3f;
    const CircleSize<float> cw{0.11f, 0.f, 0.f, 0.61f, 0.61f};
    float eacc = -INF, macc = INF;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < cycles; c += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (VAR == 5) {
                float dvx = dv.x + dtl.x, dvy = dv.y + dtl.y;
                asm volatile("" : "+v"(dvx), "+v"(dvy));
                dv.x = dvx; dv.y = dvy;
                float vx = v0.x + dv.x, vy = v0.y + dv.y;
                asm volatile("" : "+v"(vx), "+v"(vy));
                v.x = vx; v.y = vy;
                float px = __builtin_fmaf(dt_l, v.x, dp.x), py = __builtin_fmaf(dt_l, v.y, dp.y);
                asm volatile("" : "+v"(px), "+v"(py));
                dp.x = px; dp.y = py;
            } else {
                dv.x += dtl.x; dv.y += dtl.y;
                v.x = v0.x + dv.x; v.y = v0.y + dv.y;
                dp.x = __builtin_fmaf(dt_l, v.x, dp.x); dp.y = __builtin_fmaf(dt_l, v.y, dp.y);
            }
            if constexpr (VAR == 3) continue;
            if constexpr (VAR == 8) {
                // q, the wall reach and the four DPP subtracts as ONE block: the two reach instructions fill the two wait states a
                // DPP read needs after the VALU write of its source, so no s_nop
                float qx, qy, tt, e, dx1, dy1, dx2, dy2;
                const float kw = cw.wx - cw.wy;
                asm("v_add_f32 %0, %8, %10\n\tv_add_f32 %1, %9, %11\n\tv_add_f32_e64 %2, |%1|, %12\n\tv_max_f32_e64 %3, |%0|, %2\n\t"
                    "v_sub_f32_dpp %4, %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_sub_f32_dpp %5, %1, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_sub_f32_dpp %6, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_sub_f32_dpp %7, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                    : "=&v"(qx), "=&v"(qy), "=&v"(tt), "=&v"(e), "=&v"(dx1), "=&v"(dy1), "=&v"(dx2), "=&v"(dy2)
                    : "v"(q0.x), "v"(q0.y), "v"(dp.x), "v"(dp.y), "v"(kw));
                eacc = __builtin_fmaxf(eacc, e);
                macc = __builtin_fminf(macc, __builtin_fminf(__builtin_fmaf(dx1, dx1, dy1 * dy1), __builtin_fmaf(dx2, dx2, dy2 * dy2)));
                continue;
            }
            const R2<float> q{q0.x + dp.x, q0.y + dp.y};
            if constexpr (VAR != 2) eacc = __builtin_fmaxf(eacc, wall_reach_centred<float>(q.x, q.y, cw));
            if constexpr (VAR == 0 || VAR == 2 || VAR == 5) macc = __builtin_fminf(macc, pair_min_dist<float, 4>(q.x, q.y));
            if constexpr (VAR == 4) {
                const float dx1 = rot<4, 1>(q.x) - q.x, dy1 = rot<4, 1>(q.y) - q.y, dx2 = rot<4, 2>(q.x) - q.x, dy2 = rot<4, 2>(q.y) - q.y;
                macc = __builtin_fminf(macc, __builtin_fminf(__builtin_fmaf(dx1, dx1, dy1 * dy1), __builtin_fmaf(dx2, dx2, dy2 * dy2)));
            }
            if constexpr (VAR == 9) {  // partner coordinates through the LDS crossbar (ds_swizzle, quad-perm mode): no VALU DPP at all
                auto swz = [](float v, auto pat) { return __uint_as_float((unsigned)__builtin_amdgcn_ds_swizzle((int)__float_as_uint(v), decltype(pat)::value)); };
                const float x1 = swz(q.x, std::integral_constant<int, 0x8039>{}), y1 = swz(q.y, std::integral_constant<int, 0x8039>{});
                const float x2 = swz(q.x, std::integral_constant<int, 0x804E>{}), y2 = swz(q.y, std::integral_constant<int, 0x804E>{});
                const float dx1 = x1 - q.x, dy1 = y1 - q.y, dx2 = x2 - q.x, dy2 = y2 - q.y;
                macc = __builtin_fminf(macc, __builtin_fminf(__builtin_fmaf(dx1, dx1, dy1 * dy1), __builtin_fmaf(dx2, dx2, dy2 * dy2)));
            }
            if constexpr (VAR == 7) {
                float dx1, dy1, dx2, dy2;
                asm("s_nop 1\n\tv_sub_f32_dpp %0, %4, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_sub_f32_dpp %1, %5, %5 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_sub_f32_dpp %2, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_sub_f32_dpp %3, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                    : "=&v"(dx1), "=&v"(dy1), "=&v"(dx2), "=&v"(dy2) : "v"(q.x), "v"(q.y));
                macc = __builtin_fminf(macc, __builtin_fminf(__builtin_fmaf(dx1, dx1, dy1 * dy1), __builtin_fmaf(dx2, dx2, dy2 * dy2)));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = dp.x + dp.y + eacc + macc + dv.x;
    if ((threadIdx.x & 63) == 0) clk[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int VAR> int run(const char *name, float *out, long long *clk) {
    for (int wps : {4, 8}) {
        const int blocks = 256 * wps, cycles = 4000;
        hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), 0, 0, out, clk, 40, 1e-5f, 2e-5f);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), 0, 0, out, clk, cycles, 1e-5f, 2e-5f);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> h(blocks * 4); CHECK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2] / cycles;
        printf("%-44s waves/SIMD=%d  %.3f ms  clocks per cycle: per wavefront %.1f, per SIMD %.1f\n", name, wps, ms, med, med / wps);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    return 0;
}

int main() {
    auto T0 = std::chrono::steady_clock::now();
    float *out; long long *clk;
    CHECK(hipMalloc(&out, (size_t)2048 * 256 * sizeof(float))); CHECK(hipMalloc(&clk, 2048 * 4 * 8));
    printf("setup %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count());
    run<0>("quiet cycle as in k_step", out, clk);
    run<7>("same, one s_nop for the 4 DPP subs", out, clk);
    run<8>("q + reach + 4 DPP subs in one block, no s_nop", out, clk);
    run<9>("pair sweep via ds_swizzle (LDS crossbar)", out, clk);
    run<4>("pair sweep via compiler DPP movs", out, clk);
    run<1>("no pair sweep", out, clk);
    run<2>("no wall reach", out, clk);
    run<3>("integrate only (3 packed)", out, clk);
    run<5>("non-packed integrate + rest as k_step", out, clk);
    printf("total %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count());
    return 0;
}
