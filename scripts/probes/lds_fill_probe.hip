// Probe: how fast can a CU fill LDS from L2-resident global memory?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, what the GEMM kernels use)
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
// Every workgroup streams `iters` tiles of TILE bytes from a small (L2-resident) buffer; 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

template <int MODE, int THREADS, int TILE>
__global__ __launch_bounds__(THREADS) void fill(const char* __restrict__ src, int iters, int64_t span, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER_IT = THREADS * 16;           // bytes per pass of the whole workgroup
    constexpr int PASSES = TILE / PER_IT;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        char* dst = smem + (it & 1) * TILE;
        const char* s = src + ((int64_t)blockIdx.x * 4096 + (int64_t)it * TILE) % span;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int off = ps * PER_IT + wave * 1024;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((gptr_t)(s + off + lane * 16), (lds_ptr_t)(dst + off), 16, 0, 0);
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(s + off + lane * 16);
                *reinterpret_cast<uint4*>(dst + off + lane * 16) = v;
            }
        }
        __syncthreads();
        acc += *reinterpret_cast<const float*>(smem + (it & 1) * TILE + ((tid * 4) & (TILE - 1)));   // keep the data live
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int THREADS, int TILE>
static void run(const char* name, const char* buf, int64_t span, float* sink) {
    const int iters = 2000, blocks = 512;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fill<MODE, THREADS, TILE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((fill<MODE, THREADS, TILE>), dim3(blocks), dim3(THREADS), 2 * TILE, 0, buf, 10, span, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL((fill<MODE, THREADS, TILE>), dim3(blocks), dim3(THREADS), 2 * TILE, 0, buf, iters, span, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * iters * TILE;
    printf("%-46s threads %4d tile %3d KB: %7.2f TB/s chip, %6.1f GB/s per CU\n", name, THREADS, TILE / 1024, bytes / ms / 1e9,
           bytes / ms / 1e6 / 256);
}

int main() {
    const int64_t span = 16 << 20;     // 16 MiB: L2 (4 MiB/XCD) + MALL resident
    char* buf; float* sink;
    hipMalloc(&buf, span + (1 << 20)); hipMemset(buf, 1, span + (1 << 20)); hipMalloc(&sink, 4);
    run<0, 256, 32768>("LDS-DMA (global_load_lds_dwordx4)", buf, span, sink);
    run<1, 256, 32768>("global_load_dwordx4 + ds_write_b128", buf, span, sink);
    run<0, 512, 40960>("LDS-DMA (global_load_lds_dwordx4)", buf, span, sink);
    run<1, 512, 40960>("global_load_dwordx4 + ds_write_b128", buf, span, sink);
    run<0, 512, 32768>("LDS-DMA (global_load_lds_dwordx4)", buf, span, sink);
    run<1, 512, 32768>("global_load_dwordx4 + ds_write_b128", buf, span, sink);
    const int64_t small = 2 << 20;     // 2 MiB: L2 resident on every XCD
    run<0, 512, 32768>("LDS-DMA, 2 MiB working set", buf, small, sink);
    run<1, 512, 32768>("load+ds_write, 2 MiB working set", buf, small, sink);
    return 0;
}
