// does gfx950 implement the GFX9 wave_ror DPP control, and in which direction does it move data?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *o) {
    float v = (float)threadIdx.x, r = -1.f;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v));
    o[threadIdx.x] = r;
}
int main() {
    float *d, h[64];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%g ", h[i]);
    printf("\n");
    return 0;
}
