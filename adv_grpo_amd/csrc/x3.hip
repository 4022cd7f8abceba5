// x3.hip -- row kernels of the split-bf16 ("bf16x3") ViT towers: the fp32 PickScore scorer of rewards.py:561-574
// (PickScoreScorer(dtype=torch.float32): config 4's `pickscore` reward and every eval with it) without fp32 matrix math.
// An f32 value travels between two matrix products as f32 and enters a product as hi = bf16(v), lo = bf16(v - hi) laid out
// along K as [hi | hi | lo] (left operand, order 0) or [hi | lo | hi] (right operand, order 1): include/advgrpo.h, "bf16x3".
//   layernorm_x3       f32 rows -> LayerNorm (f32 affine) -> split rows                     (LayerNorm before a Linear)
//   split_act_x3       f32 rows (+ bias) -> activation -> split rows                        (fc1 -> GELU -> fc2)
//   softmax_x3_masked  f32 score rows * alpha -> softmax over the valid / causal keys -> split rows, zeros past them
// HBM-bound row kernels: one pass over the f32 row, 6 B written per element.
#include "common.hpp"

namespace advgrpo {
namespace {

__device__ __forceinline__ void x3_ld8(const float* q, float o[8]) {
    const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ uint4 x3_pack8(const float o[8]) {
    uint4 r;
    r.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
    r.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
    r.z = (uint32_t)f2bf(o[4]) | ((uint32_t)f2bf(o[5]) << 16);
    r.w = (uint32_t)f2bf(o[6]) | ((uint32_t)f2bf(o[7]) << 16);
    return r;
}
// row + c*8 of the three K thirds
__device__ __forceinline__ void x3_store(bf16_t* row, int K, int c8, const float v[8], int order) {
    float h[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        h[k] = round_bf16(v[k]);
        l[k] = v[k] - h[k];               // exact
    }
    const uint4 hi = x3_pack8(h), lo = x3_pack8(l);
    *reinterpret_cast<uint4*>(row + c8) = hi;
    *reinterpret_cast<uint4*>(row + K + c8) = order ? lo : hi;
    *reinterpret_cast<uint4*>(row + 2 * K + c8) = order ? hi : lo;
}

constexpr int X3_LN_CHUNKS = 4;   // 8-element chunks per lane: D <= 2048

__global__ __launch_bounds__(256) void layernorm_x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, bf16_t* __restrict__ out, int M, int D,
                                                           float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = D >> 3;
    const float* xr = x + (int64_t)row * D;
    float v[X3_LN_CHUNKS][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < X3_LN_CHUNKS; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            x3_ld8(xr + c * 8, v[i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[i][k];
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < X3_LN_CHUNKS; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float d = v[i][k] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    bf16_t* orow = out + (int64_t)row * 3 * D;
#pragma unroll
    for (int i = 0; i < X3_LN_CHUNKS; ++i) {
        const int c = lane + i * 64;
        if (c >= nch) continue;
        float ww[8], bb[8], o[8];
        x3_ld8(w + c * 8, ww);
        x3_ld8(b + c * 8, bb);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (v[i][k] - mean) * rstd * ww[k] + bb[k];
        x3_store(orow, D, c * 8, o, 0);
    }
}

__device__ __forceinline__ float x3_act(float x, int act) {
    if (act == 1) return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));          // GELU (exact), torch F.gelu
    if (act == 2) return x / (1.0f + expf(-1.702f * x));                             // quick_gelu: x sigmoid(1.702 x)
    return x;
}
__global__ __launch_bounds__(256) void split_act_x3_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                           bf16_t* __restrict__ out, int K, int order, int act, int64_t total8) {
    const int k8n = K >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int slice = i % k8n;
        const int64_t r = i / k8n;
        float v[8];
        x3_ld8(x + i * 8, v);
        if (bias) {
            float b8[8];
            x3_ld8(bias + slice * 8, b8);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += b8[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = x3_act(v[k], act);
        x3_store(out + r * 3 * K, K, slice * 8, v, order);
    }
}

// one workgroup per row; n (padded row length, multiple of 8) is a few hundred to a few thousand: three passes over a row in L1/L2
__global__ __launch_bounds__(256) void softmax_x3_masked_kernel(const float* __restrict__ s, bf16_t* __restrict__ out, int n,
                                                                int n_valid, int causal_period, float alpha) {
    __shared__ float red[4];
    const int64_t rowi = blockIdx.x;
    const float* r = s + rowi * n;
    bf16_t* o = out + rowi * 3 * n;
    int valid = n_valid;
    if (causal_period > 0) valid = min(n_valid, (int)(rowi % causal_period) + 1);
    const int n8 = n >> 3;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < n8; c += 256) {
        float v[8];
        x3_ld8(r + c * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (c * 8 + k < valid) mx = fmaxf(mx, v[k] * alpha);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x; c < n8; c += 256) {
        float v[8];
        x3_ld8(r + c * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (c * 8 + k < valid) sum += expf(v[k] * alpha - mx);
    }
    const float inv = 1.0f / block_sum<4>(sum, red);
    for (int c = threadIdx.x; c < n8; c += 256) {
        float v[8];
        x3_ld8(r + c * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = c * 8 + k < valid ? expf(v[k] * alpha - mx) * inv : 0.f;
        x3_store(o, n, c * 8, v, 0);
    }
}

}  // namespace
}  // namespace advgrpo

using namespace advgrpo;

extern "C" int advgrpo_layernorm_x3(const float* x, const float* w, const float* b, void* out3, int M, int D, float eps,
                                    void* stream) {
    ADVGRPO_CHECK(x && w && b && out3, "layernorm_x3: null pointer");
    ADVGRPO_CHECK(M > 0 && D > 0 && D % 8 == 0 && D <= X3_LN_CHUNKS * 512, "layernorm_x3: need D %% 8 == 0, D <= %d (D=%d)",
                  X3_LN_CHUNKS * 512, D);
    hipLaunchKernelGGL(layernorm_x3_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), x, w, b, (bf16_t*)out3, M, D, eps);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_split_act_bf16x3(const float* x, const float* bias, void* out3, int64_t rows, int K, int order, int act,
                                        void* stream) {
    ADVGRPO_CHECK(x && out3 && rows > 0 && K > 0 && K % 8 == 0, "split_act_bf16x3: need K %% 8 == 0 (K=%d)", K);
    ADVGRPO_CHECK((order == 0 || order == 1) && act >= 0 && act <= 2, "split_act_bf16x3: bad order / activation");
    const int64_t total8 = rows * (K / 8);
    int64_t blocks = (total8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_act_x3_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, bias, (bf16_t*)out3, K, order,
                       act, total8);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_softmax_rows_x3_masked(const float* s, void* out3, int64_t rows, int n, int n_valid, int causal_period,
                                              float alpha, void* stream) {
    ADVGRPO_CHECK(s && out3 && rows > 0 && rows < (1ll << 31) && n > 0 && n % 8 == 0 && n_valid > 0 && n_valid <= n &&
                      causal_period >= 0,
                  "softmax_rows_x3_masked: need n %% 8 == 0 and 0 < n_valid <= n (n=%d n_valid=%d)", n, n_valid);
    hipLaunchKernelGGL(softmax_x3_masked_kernel, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), s, (bf16_t*)out3, n,
                       n_valid, causal_period, alpha);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
