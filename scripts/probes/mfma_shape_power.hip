// Probe: sustained matrix rate of the whole chip under its power cap for the two bf16 MFMA shapes of equal nominal throughput --
// v_mfma_f32_16x16x32_bf16 (what gemm8p_kernel issues: 8 Ki MACs, 8 operand + 4 accumulator VGPRs per instruction) against
// v_mfma_f32_32x32x16_bf16 (16 Ki MACs, 8 operand + 16 accumulator VGPRs: half the operand reads and half the instructions per flop).
// 256 workgroups x 8 waves (2 per SIMD, as gemm8p), register operands only (no LDS, no memory): what differs is energy per flop, i.e. the
// clock the power management settles on.  Random (non-zero) operand bits: a zero operand would gate the datapath.
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_power scripts/probes/mfma_shape_power.hip && /tmp/mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512) void burn(const uint4* seed, float* out, int iters) {
    const uint4 s0 = seed[threadIdx.x & 63], s1 = seed[64 + (threadIdx.x & 63)];
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) { uint4 t = s0; t.x ^= 0x00010001u * (i + 1); a[i] = __builtin_bit_cast(bf16x8, t); }
    for (int i = 0; i < 2; ++i) { uint4 t = s1; t.y ^= 0x00010001u * (i + 1); b[i] = __builtin_bit_cast(bf16x8, t); }
    float r = 0.f;
    if constexpr (SHAPE == 16) {                   // a 64 x 32 patch per k-step of 32: 4 x 2 blocks of 16 x 16 -> 8 MFMAs = 64 Ki MACs
        f32x4 acc[4][2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) r += acc[i][j][0] + acc[i][j][3];
    } else {                                       // the same MACs: a 64 x 64 patch, 2 x 2 blocks of 32 x 32 per k-step of 16, two k-steps -> 8 MFMAs = 128 Ki MACs
        f32x16 acc[2][2] = {};                     // (four independent accumulators: two were 13 % slower still -- back-to-back dependent issue)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * k], b[(j + k) & 1], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) r += acc[i][j][0] + acc[i][j][15];
    }
    if (r == 12345.678f) out[0] = r;               // keeps the accumulators live
}

int main() {
    uint4 h[128];
    srand(7);
    for (auto& v : h) {                             // bf16 values in (0.5, 2): exponent 126 / 127, random mantissas, random signs
        auto w = [] { auto e = [] { return (unsigned)((rand() & 0x807f) | ((126 + (rand() & 1)) << 7)); }; return e() | (e() << 16); };
        v = uint4{w(), w(), w(), w()};
    }
    uint4* d; float* o;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 400000;                       // per wave: iters x 4 x 64 Ki MACs
    const double flop = 2.0 * 256 * 8 * (double)iters * 4 * 65536;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 4; ++round)
        for (int shape : {16, 32}) {
            hipEventRecord(e0);
            if (shape == 16) hipLaunchKernelGGL(burn<16>, dim3(256), dim3(512), 0, 0, d, o, iters);
            else hipLaunchKernelGGL(burn<32>, dim3(256), dim3(512), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("round %d  %s: %8.2f ms  %7.1f TFLOP/s\n", round, shape == 16 ? "16x16x32" : "32x32x16", ms, flop / ms / 1e9);
        }
    return 0;
}
