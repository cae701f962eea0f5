// microbenchmark: the floor of one step-kernel launch on MI355X: back-to-back launches of (a) an empty kernel, (b) a kernel that
// only moves the step kernel's HBM traffic (65 536 envs x 4 movers: 4 float2 loads, 5 float2 stores, flags), for several block sizes
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_empty(int n) {}
__global__ void k_move(const float2 *a, float2 *p, float2 *v, const float2 *g, float2 *o, float2 *ag, float2 *dg, float *r, unsigned char *f, int *el, long long n) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float2 A = a[t], P = p[t], V = v[t], G = g[t];
    int e = el[t >> 2];
    V.x += 1e-3f * A.x; V.y += 1e-3f * A.y; P.x += 1e-3f * V.x; P.y += 1e-3f * V.y;
    p[t] = P; v[t] = V; o[t] = V; ag[t] = P; dg[t] = G;
    if ((t & 3) == 0) { r[t >> 2] = P.x; f[t >> 2] = 1; f[n / 4 + (t >> 2)] = 0; el[t >> 2] = e + 1; }
}
int main() {
    const long long n = 65536LL * 4;
    float2 *buf[7]; for (auto &b : buf) CHECK(hipMalloc(&b, n * sizeof(float2)));
    float *r; unsigned char *f; int *el; CHECK(hipMalloc(&r, n)); CHECK(hipMalloc(&f, n)); CHECK(hipMalloc(&el, n)); CHECK(hipMemset(el, 0, n));
    for (auto &b : buf) CHECK(hipMemset(b, 0, n * sizeof(float2)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int bs : {64, 128, 256, 512, 1024}) {
        for (int kind = 0; kind < 2; ++kind) {
            int blocks = (int)((n + bs - 1) / bs), iters = 3000;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i) {
                    if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(bs), 0, 0, 0);
                    else hipLaunchKernelGGL(k_move, dim3(blocks), dim3(bs), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], r, f, el, n);
                }
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            }
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("block %4d  %-6s  %.2f us per launch (back to back)\n", bs, kind ? "move" : "empty", ms * 1e3 / iters);
        }
    }
    return 0;
}
