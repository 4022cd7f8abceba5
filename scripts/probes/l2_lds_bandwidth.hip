// Probe: how fast can all 256 CUs pull data through L2 into their LDS with global_load_lds_dwordx4 (the path of every
// k-tile of gemm8p_kernel and conv3x3_x3_kernel)?
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/l2_lds_bandwidth.hip -o /tmp/l2_lds && /tmp/l2_lds
// One 512-thread workgroup per CU (as the GEMM), each issues stages of 16 - 64 KiB (2 - 8 DMA instructions per wave, 1 KiB each) into a
// 128 KiB LDS ring with DEPTH - 1 stages in flight, from a working set of a given size that every workgroup walks with its own
// offset -- small sets live in L2 (4 MiB per XCD), larger ones in the 256 MiB Infinity Cache, 1 GiB comes from HBM.  Rows are
// 128 B (64 bf16 of a K-contiguous operand) at a pitch of `pitch` bytes, 8 rows per instruction: the GEMM's access shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int DEPTH, int INST, int AUX = 0>
__global__ __launch_bounds__(512, 2) void pull(const char* __restrict__ buf, size_t set_bytes, int pitch, int stages, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // stage s of workgroup b: 512 rows of 128 B starting at row (b * 977 + s * 512) of the set, wrapped
    const size_t rows_in_set = set_bytes / pitch;
    size_t row0 = ((size_t)blockIdx.x * 9973) % rows_in_set;
    auto issue = [&](int s) {
        char* dst = smem + (s % DEPTH) * (INST * 8192) + wave * (INST * 1024);
#pragma unroll
        for (int it = 0; it < INST; ++it) {
            size_t r = (row0 + (size_t)s * (INST * 64) + (wave * INST + it) * 8 + (lane >> 3)) % rows_in_set;
            const char* src = buf + r * pitch + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(dst + it * 1024), 16, 0, AUX);
        }
    };
    for (int s = 0; s < DEPTH - 1; ++s) issue(s);
    for (int s = 0; s < stages; ++s) {
        if (s + DEPTH - 1 < stages) issue(s + DEPTH - 1);
        // wait until stage s has landed: DEPTH - 1 younger stages (8 instructions each) may stay in flight
        if (s + DEPTH - 1 < stages) {
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"((DEPTH - 1) * INST) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<float*>(smem);
}

// the same access shape with plain 16-byte loads into registers (no LDS): is the ceiling the LDS-DMA path or L2 -> CU?
__global__ __launch_bounds__(512, 2) void pull_regs(const char* __restrict__ buf, size_t set_bytes, int pitch, int stages, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t rows_in_set = set_bytes / pitch;
    size_t row0 = ((size_t)blockIdx.x * 9973) % rows_in_set;
    uint4 acc = {0, 0, 0, 0};
    for (int s = 0; s < stages; ++s) {
        uint4 v[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            size_t r = (row0 + (size_t)s * 512 + (wave * 8 + it) * 8 + (lane >> 3)) % rows_in_set;
            v[it] = *reinterpret_cast<const uint4*>(buf + r * pitch + (lane & 7) * 16);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) { acc.x ^= v[it].x; acc.y ^= v[it].y; acc.z ^= v[it].z; acc.w ^= v[it].w; }
    }
    if (sink && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = 1.f;
}
static void run_regs(const char* buf, size_t set_bytes, int pitch, float* sink) {
    const int stages = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(pull_regs, dim3(256), dim3(512), 0, 0, buf, set_bytes, pitch, 64, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(pull_regs, dim3(256), dim3(512), 0, 0, buf, set_bytes, pitch, stages, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 256.0 * stages * 65536;
    printf("  set %7.1f MiB  pitch %5d B  global_load_dwordx4 into registers, 8 per lane in flight:          %6.2f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n",
           set_bytes / 1048576.0, pitch, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9);
}

template <int DEPTH, int INST, int AUX = 0>
static void run(const char* buf, size_t set_bytes, int pitch, float* sink) {
    const int stages = 4096;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pull<DEPTH, INST, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * INST * 8192));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((pull<DEPTH, INST, AUX>), dim3(256), dim3(512), DEPTH * INST * 8192, 0, buf, set_bytes, pitch, 64, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((pull<DEPTH, INST, AUX>), dim3(256), dim3(512), DEPTH * INST * 8192, 0, buf, set_bytes, pitch, stages, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 256.0 * stages * INST * 8192;
    printf("  cpol %2d  set %7.1f MiB  pitch %5d B  %3d KiB in flight per CU (%d stages of %d KiB): %6.2f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n", AUX, set_bytes / 1048576.0, pitch,
           (DEPTH - 1) * INST * 8, DEPTH - 1, INST * 8, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9);
}

int main() {
    const size_t maxb = (size_t)1 << 30;
    char* buf; float* sink;
    CHECK(hipMalloc(&buf, maxb)); CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(buf, 1, maxb));
    printf("global_load_lds_dwordx4, 256 workgroups x 512 threads, 64 KiB stages:\n");
    for (size_t mb : {2, 16, 64, 192, 1024})
        for (int pitch : {128, 3072}) {
            run<2, 8>(buf, mb << 20, pitch, sink);      //  64 KiB in flight
            run<4, 4>(buf, mb << 20, pitch, sink);      //  96 KiB in flight (the eight-phase GEMM keeps 80)
            run<8, 2>(buf, mb << 20, pitch, sink);      // 112 KiB in flight
            run_regs(buf, mb << 20, pitch, sink);
        }
    // cache-policy bits of the DMA instruction (1 = sc0, 2 = nt, 16 = sc1): does any of them move the ceiling?
    printf("cache policy of the DMA instruction, 16 MiB set, pitch 3072, 96 KiB in flight:\n");
    run<4, 4, 0>(buf, (size_t)16 << 20, 3072, sink);
    run<4, 4, 1>(buf, (size_t)16 << 20, 3072, sink);
    run<4, 4, 2>(buf, (size_t)16 << 20, 3072, sink);
    run<4, 4, 3>(buf, (size_t)16 << 20, 3072, sink);
    run<4, 4, 16>(buf, (size_t)16 << 20, 3072, sink);
    run<4, 4, 17>(buf, (size_t)16 << 20, 3072, sink);
    run<4, 4, 18>(buf, (size_t)16 << 20, 3072, sink);
    return 0;
}
