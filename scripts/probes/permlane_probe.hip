// Probe: semantics of v_permlane16_swap / v_permlane32_swap vs __shfl_xor (build and run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, float* o) {
    const float v = x[threadIdx.x];
    float a = v, b = v, c = v, d = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    o[threadIdx.x] = a;
    o[64 + threadIdx.x] = b;
    o[128 + threadIdx.x] = c;
    o[192 + threadIdx.x] = d;
}
int main() {
    float h[64], r[256], *dx, *dout;
    for (int i = 0; i < 64; ++i) h[i] = (float)i;
    hipMalloc(&dx, sizeof(h)); hipMalloc(&dout, sizeof(r));
    hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout);
    hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
    const char* names[4] = {"p16[0]", "p16[1]", "p32[0]", "p32[1]"};
    for (int j = 0; j < 4; ++j) {
        printf("%s:", names[j]);
        for (int i = 0; i < 64; ++i) printf(" %d", (int)r[j * 64 + i]);
        printf("\n");
    }
    return 0;
}
