// Probe: operand layout and scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3) on gfx950.
// Hypothesis: lane l holds row (A) / column (B) l & 15 and the 32 consecutive k of block l >> 4, byte e of the 8 VGPRs = k 32*(l>>4)+e;
// scale operand: E8M0 in byte `opsel` of the lane's VGPR, applied to that lane's 32-element block; C/D as the 16x16 bf16 form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned char* A /*[16][128]*/, const unsigned char* B /*[16][128] = B^T rows n*/, float* C, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int w = 0; w < 8; ++w) {
        a[w] = *reinterpret_cast<const int*>(A + (l & 15) * 128 + (l >> 4) * 32 + w * 4);
        b[w] = *reinterpret_cast<const int*>(B + (l & 15) * 128 + (l >> 4) * 32 + w * 4);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa[l], 0, sb[l]);
    // operands as in the bf16 kernels: D[m][n] with lane: n = l & 15 (B's index), m = (l >> 4) * 4 + r (A's index)?  store raw
    for (int r = 0; r < 4; ++r) C[l * 4 + r] = acc[r];
}
// the unscaled form (both scale operands the constant 0: the compiler emits v_mfma_f32_16x16x128_f8f6f4 without the
// v_mfma_ld_scale prefix); expectation: identical to unit scales
__global__ void k0(const unsigned char* A, const unsigned char* B, float* C) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int w = 0; w < 8; ++w) {
        a[w] = *reinterpret_cast<const int*>(A + (l & 15) * 128 + (l >> 4) * 32 + w * 4);
        b[w] = *reinterpret_cast<const int*>(B + (l & 15) * 128 + (l >> 4) * 32 + w * 4);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[l * 4 + r] = acc[r];
}
// issue rate: 4 waves per SIMD-less block (one wave per SIMD), 8 independent accumulators, N iterations
template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, int iters, int s) {
    v8i a, b;
    for (int w = 0; w < 8; ++w) { a[w] = 0x38383838 + (int)threadIdx.x * 0; b[w] = 0x38383838; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, s, 0, s);
        }
    }
    const long long t1 = clock64();
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[100000] = (float)(t1 - t0) / (float)(iters * 8);
}
static float dec(unsigned char v) {   // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
    return s ? -x : x;
}
int main() {
    unsigned char hA[16 * 128], hB[16 * 128];
    const unsigned char vals[] = {0x00, 0x38, 0x40, 0x30, 0xB8, 0xC0, 0x28, 0x44};   // 0, 1, 2, .5, -1, -2, .25, 3
    srand(1);
    for (int i = 0; i < 16 * 128; ++i) { hA[i] = vals[rand() % 8]; hB[i] = vals[rand() % 8]; }
    int hsa[64], hsb[64];
    for (int l = 0; l < 64; ++l) { hsa[l] = 0x7F7F7F7F; hsb[l] = 0x7F7F7F7F; }
    // second experiment: lane-dependent A scale: block (l >> 4) == 2 gets 2^1, row 5 block 0 gets 2^-2 (byte 0 selected by opsel 0)
    unsigned char *dA, *dB; float* dC; int *dsa, *dsb;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 64 * 4 * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int exp = 0; exp < 2; ++exp) {
        if (exp == 1) for (int l = 0; l < 64; ++l) { if ((l >> 4) == 2) hsa[l] = 0x7F7F7F80; if (l == 5) hsa[l] = 0x7F7F7F7D; }
        hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dsa, dsb);
        float hC[256]; hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        // reference under the hypothesis, both orientations of the output
        double e1 = 0, e2 = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int x = l & 15, y = (l >> 4) * 4 + r;
            double s1 = 0, s2 = 0;     // s1: D[m = y (A row)][n = x (B row)] ; s2: D[m = x][n = y]
            for (int kk = 0; kk < 128; ++kk) {
                auto sc = [&](int row) { int v = hsa[(kk >> 5) * 16 + row] & 0xff; return ldexp(1.0, v - 127); };
                s1 += dec(hA[y * 128 + kk]) * sc(y) * dec(hB[x * 128 + kk]);
                s2 += dec(hA[x * 128 + kk]) * sc(x) * dec(hB[y * 128 + kk]);
            }
            e1 += fabs(s1 - hC[l * 4 + r]); e2 += fabs(s2 - hC[l * 4 + r]);
        }
        printf("experiment %d: sum |err| with D[m = (l>>4)*4+r (A row)][n = l&15 (B row)] = %g ; transposed = %g\n", exp, e1, e2);
    }
    {
        hipLaunchKernelGGL(k0, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        float hC[256]; hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        double e1 = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int x = l & 15, y = (l >> 4) * 4 + r;
            double s1 = 0;
            for (int kk = 0; kk < 128; ++kk) s1 += dec(hA[y * 128 + kk]) * dec(hB[x * 128 + kk]);
            e1 += fabs(s1 - hC[l * 4 + r]);
        }
        printf("unscaled form (constant 0 scales): sum |err| vs scale = 1 reference = %g\n", e1);
    }
    {
        float* dout; hipMalloc(&dout, 100001 * 4);
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                const int iters = 20000, blocks = 256;
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, dout, iters, 0x7F7F7F7F);
                else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, dout, iters, 0x7F7F7F7F);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                float cyc; hipMemcpy(&cyc, dout + 100000, 4, hipMemcpyDeviceToHost);
                const double flop = 2.0 * 16 * 16 * 128 * 8.0 * iters * blocks * 4;
                printf("%s: %.3f ms, %.1f TFLOP/s, %.1f clock64 ticks per MFMA\n", mode ? "scaled (unit scales in a VGPR)" : "unscaled", ms, flop / ms * 1e-9, cyc);
            }
        }
    }
    return 0;
}
