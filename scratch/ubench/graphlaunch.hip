// Host cost of issuing S small kernels per iteration, steady state: (a) S hipLaunchKernelGGL on S streams, (b) ONE hipGraphLaunch of a graph
// with S parallel kernel nodes (captured from S streams forked from one), (c) S launches from S host threads (one per stream).
// build: hipcc --offload-arch=gfx950 -O2 graphlaunch.hip -o graphlaunch -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
struct Big { float pad[120]; int *out; int spin; };  // ~500 B of kernel arguments, like KParams
__global__ void work(const Big b) {
    long long t0 = clock64();
    while (clock64() - t0 < b.spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) b.out[0] += 1;
}
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    int *out; CK(hipMalloc(&out, 64 * sizeof(int))); CK(hipMemset(out, 0, 64 * sizeof(int)));
    for (int spin : {2000, 20000}) {  // ~1 us and ~8 us kernels
        for (int S : {1, 2, 4, 8}) {
            std::vector<hipStream_t> st(S);
            for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            Big b{}; b.spin = spin;
            auto now = [] { return std::chrono::steady_clock::now(); };
            auto us = [](auto a, auto c) { return std::chrono::duration<double, std::micro>(c - a).count(); };
            // (a) plain launches
            for (int w = 0; w < 200; ++w) for (int k = 0; k < S; ++k) { b.out = out + k; hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, st[k], b); }
            CK(hipDeviceSynchronize());
            auto t0 = now();
            for (int i = 0; i < iters; ++i) for (int k = 0; k < S; ++k) { b.out = out + k; hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, st[k], b); }
            auto t1 = now(); CK(hipDeviceSynchronize()); auto t2 = now();
            const double a_issue = us(t0, t1) / iters, a_done = us(t0, t2) / iters;
            // (b) graph: fork S branches from st[0]
            hipGraph_t g; hipGraphExec_t ge;
            std::vector<hipEvent_t> ev(S + 1);
            for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
            CK(hipEventRecord(ev[S], st[0]));
            for (int k = 1; k < S; ++k) CK(hipStreamWaitEvent(st[k], ev[S], 0));
            for (int k = 0; k < S; ++k) { b.out = out + k; hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, st[k], b); }
            for (int k = 1; k < S; ++k) { CK(hipEventRecord(ev[k], st[k])); CK(hipStreamWaitEvent(st[0], ev[k], 0)); }
            CK(hipStreamEndCapture(st[0], &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 200; ++w) CK(hipGraphLaunch(ge, st[0]));
            CK(hipDeviceSynchronize());
            t0 = now();
            for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(ge, st[0]));
            t1 = now(); CK(hipDeviceSynchronize()); t2 = now();
            const double g_issue = us(t0, t1) / iters, g_done = us(t0, t2) / iters;
            // (c) one host thread per stream
            std::atomic<int> go{0};
            std::vector<std::thread> th;
            auto t3 = now();
            for (int k = 0; k < S; ++k) th.emplace_back([&, k] { Big bb{}; bb.spin = spin; bb.out = out + k; for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, st[k], bb); });
            for (auto &t : th) t.join();
            auto t4 = now(); CK(hipDeviceSynchronize()); auto t5 = now();
            printf("spin %5d S=%d | plain: issue %6.2f done %6.2f | graph: issue %6.2f done %6.2f | threads: issue %6.2f done %6.2f  (us per iteration of S kernels)\n",
                   spin, S, a_issue, a_done, g_issue, g_done, us(t3, t4) / iters, us(t3, t5) / iters);
            fflush(stdout);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            for (auto &s : st) CK(hipStreamDestroy(s));
        }
    }
    return 0;
}
