// microbenchmark: vector-issue rate of ONE SIMD of gfx950 as a function of (a) how many wavefronts are really resident on it
// and (b) how many independent dependency chains each wavefront offers. Every wavefront records where it ran
// (HW_REG_HW_ID, HW_REG_XCC_ID) and its own start / end shader clock, so that the rate is computed per SIMD from the
// wavefronts that were actually co-resident there -- not from an assumed placement ("weak" item 6).
//
// build: hipcc --offload-arch=gfx950 -O2 simd.hip -o simd
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Rec { unsigned hw_id, xcc_id; long long t0, t1; };

#define REP4(X) X X X X
#define REP8(X) X X X X X X X X
// KIND: 0 one dependent v_fma chain, 1 two chains, 2 four chains, 3 eight chains, 4 one dependent v_pk_fma chain,
//       5 two v_pk_fma chains, 6 four v_pk_fma chains, 7 dependent v_add with DPP source (quad_perm) + s_nop 1
//       8 v_cmp -> vcc -> s_cbranch (not taken) + 6 dependent fma, 9 same 7 VALU without the decision
//       (the cost of scalar-only decisions and branch-free selects: branch.hip)
template <int KIND> __global__ void __launch_bounds__(256) k(float *out, Rec *rec, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 0.9999f, c = 0.5f, big = 1e30f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {
            asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(a0) : "v"(b), "v"(c));
        } else if constexpr (KIND == 1) {
            asm volatile(REP4("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\t") : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
        } else if constexpr (KIND == 2) {
            asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
                         "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        } else if constexpr (KIND == 3) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 4) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, pb = {b, b}, pc = {c, c};
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %1, %2\n\t") : "+v"(p0) : "v"(pb), "v"(pc));
            a0 = p0.x; a1 = p0.y;
        } else if constexpr (KIND == 5) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, pb = {b, b}, pc = {c, c};
            asm volatile(REP4("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3\n\t") : "+v"(p0), "+v"(p1) : "v"(pb), "v"(pc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y;
        } else if constexpr (KIND == 6) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                         "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if constexpr (KIND == 7) {
            asm volatile(REP8("s_nop 1\n\tv_sub_f32_dpp %0, %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t") : "+v"(a0) : "v"(c));
        } else if constexpr (KIND == 8) {  // (iters counts 8-instruction groups: 1 cmp + 1 branch + 6 fma)
            asm volatile("v_cmp_gt_f32 vcc, %0, %3\n\ts_cbranch_vccnz 1f\n\t" REP4("v_fma_f32 %0, %0, %1, %2\n\t") "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t1:\n\t"
                         : "+v"(a0) : "v"(b), "v"(c), "v"(big) : "vcc");
        } else if constexpr (KIND == 9) {
            asm volatile("v_cmp_gt_f32 vcc, %0, %3\n\t" REP4("v_fma_f32 %0, %0, %1, %2\n\t") "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                         : "+v"(a0) : "v"(b), "v"(c), "v"(big) : "vcc");
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) {
        Rec r;
        r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        r.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        r.t0 = t0; r.t1 = t1;
        rec[(size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = r;
    }
}

static double median(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <int KIND> int run(const char *name, int valu_per_iter, float *out, Rec *rec, int block) {
    for (int wps : {1, 2, 4, 8}) {
        const int waves = 1024 * wps, blocks = waves * 64 / block, iters = 4000;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(block), 0, 0, out, rec, 50);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(block), 0, 0, out, rec, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<Rec> h(waves); CHECK(hipMemcpy(h.data(), rec, waves * sizeof(Rec), hipMemcpyDeviceToHost));
        // SIMD identity: xcc id + HW_ID without wave slot [3:0], pipe [7:6], tg_id.. upward (keep simd [5:4], cu [11:8], sh [12], se [15:13])
        std::map<unsigned, std::vector<int>> simd;
        for (int w = 0; w < waves; ++w) simd[((h[w].xcc_id & 0xF) << 16) | (h[w].hw_id & 0xFF30u)].push_back(w);
        std::map<int, int> hist;  // resident wavefronts per SIMD -> number of SIMDs
        std::map<int, std::vector<double>> rate, own;
        for (auto &kv : simd) {
            long long lo = h[kv.second[0]].t0, hi = h[kv.second[0]].t1;
            for (int w : kv.second) { lo = std::min(lo, h[w].t0); hi = std::max(hi, h[w].t1); }
            // wavefronts of this SIMD that overlap the busiest instant: count at the median start time
            const int n = (int)kv.second.size();
            hist[n]++;
            rate[n].push_back((double)(hi - lo) / ((double)n * iters * valu_per_iter));
            for (int w : kv.second) own[n].push_back((double)(h[w].t1 - h[w].t0) / ((double)iters * valu_per_iter));
        }
        printf("%-34s block %4d asked waves/SIMD=%d  %.3f ms  SIMDs used %4zu |", name, block, wps, ms, simd.size());
        for (auto &kv : hist)
            printf("  %d/SIMD x%d: %.2f clk per VALU per SIMD, %.2f per wave", kv.first, kv.second, median(rate[kv.first]), median(own[kv.first]));
        printf("\n");
    }
    return 0;
}

int main(int argc, char **argv) {
    const bool extras_only = argc > 1;
    float *out; Rec *rec;
    CHECK(hipMalloc(&out, (size_t)8192 * 64 * sizeof(float))); CHECK(hipMalloc(&rec, 8192 * sizeof(Rec)));
    for (int block : {256, 64}) {
        if (extras_only) {
            if (block == 64) continue;
            run<7>("s_nop 1 + v_sub_f32_dpp, 1 chain", 8, out, rec, block);
            run<9>("cmp + 6 fma (no decision)", 7, out, rec, block);
            run<8>("cmp + vcc branch n.t. + 6 fma", 7, out, rec, block);
            continue;
        }
        run<0>("v_fma_f32, 1 chain", 8, out, rec, block);
        run<1>("v_fma_f32, 2 chains", 8, out, rec, block);
        run<2>("v_fma_f32, 4 chains", 8, out, rec, block);
        run<3>("v_fma_f32, 8 chains", 8, out, rec, block);
        if (block == 64) continue;
        run<4>("v_pk_fma_f32, 1 chain", 8, out, rec, block);
        run<5>("v_pk_fma_f32, 2 chains", 8, out, rec, block);
        run<6>("v_pk_fma_f32, 4 chains", 8, out, rec, block);
        run<7>("s_nop 1 + v_sub_f32_dpp, 1 chain", 8, out, rec, block);
        run<9>("cmp + 6 fma (no decision)", 7, out, rec, block);
        run<8>("cmp + vcc branch n.t. + 6 fma", 7, out, rec, block);
    }
    return 0;
}
