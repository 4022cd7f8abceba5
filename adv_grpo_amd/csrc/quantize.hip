// quantize.hip -- per-row fp8 (OCP e4m3) quantisation of bf16 matrices for the fp8 Linears of gemm8p_fp8.hip (BASELINE
// config 5's "fp8 MFMA path"; the reference has no code for it, so the scheme is this build's and is stated here):
//
//   scale[r] = max_k |x[r,k]| / 448            (1 when the row is all zero; 448 = largest finite e4m3 value)
//   q[r,k]   = e4m3( clamp(x[r,k] * (1 / scale[r]), -448, 448) )     round-to-nearest-even (v_cvt_pk_fp8_f32)
//
// so that x[r,k] ~= scale[r] * q[r,k]: one f32 scale per token row of an activation, per output channel of a weight.
// HBM-bound: a row is read once with 16-byte lane loads, kept in registers for the max and the conversion, and written
// once with 8-byte lane stores -- 3 B of traffic per element.  One row per wave, four rows per workgroup, no LDS.
//
// split_period > 0 compacts the two row ranges of a joint [B, period, K] buffer (the image rows s < split_first and the text
// rows behind them of the joint attention output): input row r = b * period + s goes to output row b * split_first + s for
// s < split_first and to B * split_first + b * (period - split_first) + (s - split_first) otherwise -- the image-stream and
// text-stream Linears then read contiguous row ranges of ONE quantised buffer.
#include "common.hpp"

namespace advgrpo {

namespace {

constexpr int Q_MAX_CHUNKS = 32;   // 8-element chunks per lane => K <= 16384 (the widest Linear input: 4 x 2432 = 9728)

struct QuantParams {
    const bf16_t* x; int64_t ldx;
    uint8_t* q; int64_t ldq;
    float* scale;
    int M, K, split_first, split_period;
};

template <int MAXC>
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const QuantParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int nch = p.K >> 3;
    const bf16_t* xr = p.x + (int64_t)row * p.ldx;
    uint4 v[MAXC];
    uint32_t amax_bits = 0;     // |bf16| compares like its bit pattern: the max runs on integers, two values per dword
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        v[i] = uint4{0u, 0u, 0u, 0u};
        if (c < nch) v[i] = *reinterpret_cast<const uint4*>(xr + c * 8);
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            amax_bits = max(amax_bits, w[k] & 0x7fffu);
            amax_bits = max(amax_bits, (w[k] >> 16) & 0x7fffu);
        }
    }
    const float amax = wave_allreduce(bf2f((bf16_t)amax_bits), [](float a, float b) { return fmaxf(a, b); });
    // (a NaN / Inf input row gives a NaN / Inf scale and NaN codes: propagated, not hidden)
    const float scale = amax > 0.f ? fmaxf(amax * (1.0f / 448.0f), 1.17549435e-38f) : 1.0f;   // (never subnormal / zero: 1 / scale stays finite)
    const float inv = 1.0f / scale;
    int64_t orow = row;
    if (p.split_period > 0) {
        const int b = row / p.split_period, s = row - b * p.split_period, nb = p.M / p.split_period;
        orow = s < p.split_first ? (int64_t)b * p.split_first + s
                                 : (int64_t)nb * p.split_first + (int64_t)b * (p.split_period - p.split_first) + (s - p.split_first);
    }
    if (lane == 0) p.scale[orow] = scale;
    uint8_t* qr = p.q + orow * p.ldq;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c >= nch) continue;
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = fminf(fmaxf(__builtin_bit_cast(float, w[k] << 16) * inv, -448.0f), 448.0f);
            f[2 * k + 1] = fminf(fmaxf(__builtin_bit_cast(float, w[k] & 0xffff0000u) * inv, -448.0f), 448.0f);
        }
        uint32_t o[2];
        o[0] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
        o[0] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], (int)o[0], true);
        o[1] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
        o[1] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], (int)o[1], true);
        *reinterpret_cast<uint2*>(qr + c * 8) = uint2{o[0], o[1]};
    }
}

}  // namespace
}  // namespace advgrpo

using namespace advgrpo;

extern "C" int advgrpo_quant_fp8_rows(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int M, int K,
                                      int split_first, int split_period, void* stream) {
    ADVGRPO_CHECK(x && q && scale, "quant_fp8_rows: null pointer");
    ADVGRPO_CHECK(M > 0 && K > 0 && K % 8 == 0 && K <= Q_MAX_CHUNKS * 512, "quant_fp8_rows: need K %% 8 == 0, K <= %d (K=%d)",
                  Q_MAX_CHUNKS * 512, K);
    ADVGRPO_CHECK(ldx % 8 == 0 && ldq % 8 == 0 && ldq >= K && ldx >= K, "quant_fp8_rows: pitches must be multiples of 8 and >= K");
    ADVGRPO_CHECK(((reinterpret_cast<uintptr_t>(x) & 15) | (reinterpret_cast<uintptr_t>(q) & 7)) == 0, "quant_fp8_rows: misaligned buffer");
    ADVGRPO_CHECK(split_period == 0 || (split_period > 0 && split_first > 0 && split_first < split_period && M % split_period == 0),
                  "quant_fp8_rows: the split map needs 0 < split_first < split_period and M %% split_period == 0");
    const QuantParams p{(const bf16_t*)x, ldx, (uint8_t*)q, ldq, scale, M, K, split_first, split_period};
    const dim3 grid((M + 3) / 4), block(256);
    if (K <= 2048) hipLaunchKernelGGL(quant_fp8_rows_kernel<4>, grid, block, 0, as_stream(stream), p);
    else if (K <= 4096) hipLaunchKernelGGL(quant_fp8_rows_kernel<8>, grid, block, 0, as_stream(stream), p);
    else if (K <= 8192) hipLaunchKernelGGL(quant_fp8_rows_kernel<16>, grid, block, 0, as_stream(stream), p);
    else hipLaunchKernelGGL(quant_fp8_rows_kernel<32>, grid, block, 0, as_stream(stream), p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
