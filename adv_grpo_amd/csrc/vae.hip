// vae.hip -- HBM-bound kernels of the SD3 VAE decoder (NHWC bf16 activations) for gfx950.
//
// The reference decodes with diffusers' AutoencoderKL in fp32 (sd3_pipeline_with_logprob_fast.py:667-670,
// train_sd3_fast_pickscore.py:481).  Here the 3x3 convolutions run as implicit GEMMs on the bf16 MFMA
// kernel (gemm.hip, conv loader); this file holds what sits between them:
//   groupnorm_stats  per-(image, group) sum / sum-of-squares, f32 per thread, f64 atomics per block
//   groupnorm_apply  (x - mean) * rstd * w + b, optional SiLU, bf16 out  -- one read, one write
//   softmax_rows     in-place row softmax of the mid-block attention scores (4096 keys, 1 head)
//   latents_to_nhwc  z/scaling + shift, NCHW f32|bf16 -> NHWC bf16 padded to 64 channels
//   image_postprocess NHWC (3 of ldc channels) -> NCHW f32, (x/2 + 0.5).clamp(0, 1)
// Activations are as large as 1 GB (8 x 512 x 512 x 128 bf16), far beyond the 256 MB Infinity Cache:
// each kernel is shaped as a single streaming pass with 16-byte lane accesses along the contiguous
// channel axis.
#include "common.hpp"

namespace advgrpo {

__device__ inline void unpack8v(const uint4& r, float o[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[2 * k] = bf2f((bf16_t)(w[k] & 0xffffu));
        o[2 * k + 1] = bf2f((bf16_t)(w[k] >> 16));
    }
}
__device__ inline uint4 pack8v(const float o[8]) {
    uint4 r;
    r.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
    r.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
    r.z = (uint32_t)f2bf(o[4]) | ((uint32_t)f2bf(o[5]) << 16);
    r.w = (uint32_t)f2bf(o[6]) | ((uint32_t)f2bf(o[7]) << 16);
    return r;
}

// x: [B, HW, C] bf16.  grid (chunks, B); each block walks pixels [chunk*ppb, (chunk+1)*ppb).
// A thread owns one 8-channel slice (c8 = tid % (C/8)) => with C/G in {4, 8, 16} its 8 channels
// belong to at most two groups... we keep per-thread sums per 4-channel half and reduce in LDS.
// stats: [B, G, 2] f64 (sum, sumsq), zeroed by the caller.
template <typename T>
__device__ inline void load8(const T* __restrict__ q, float o[8]) {
    if constexpr (sizeof(T) == 2) {
        unpack8v(*reinterpret_cast<const uint4*>(q), o);
    } else {
        const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
}
// split-bf16 ("bf16x3") representation of an f32 value: hi = bf16(v), lo = bf16(v - hi) (the subtraction is exact);
// hi + lo carries 16 significand bits.  A product x*w is then formed on the bf16 MFMA as xh*wh + xh*wl + xl*wh (f32
// accumulate; the dropped xl*wl term is 2^-16 relative), by laying the K axis out three times: activations as
// [hi | hi | lo], weights as [hi | lo | hi].
__device__ inline void split8(const float v[8], uint4& hi, uint4& lo) {
    float h[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        h[k] = round_bf16(v[k]);
        l[k] = v[k] - h[k];
    }
    hi = pack8v(h);
    lo = pack8v(l);
}

// fp16 pair ("f16x2", conv_x3.hip F16): hi = f16(s v), lo = f16(s v - hi) with a power-of-two pre-scale s that keeps un-normalised
// activations inside the fp16 range (the convolution's alpha = 1 / s undoes it exactly); 22 significant bits
__device__ inline void split8_f16(const float v[8], float prescale, uint4& hi, uint4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        // (saturated at the fp16 range: beyond it hi would be inf and lo = a - inf NaN -- a silently NaN image; clamped, the pair
        //  carries +-65504 + its rounding residue, i.e. the value is CLIPPED at |v| = 65504 / prescale (1.05e6 for the un-normalised
        //  inputs of the upsamplers, 65504 after a GroupNorm: activations of a working decoder are O(10)))
        const float a = fminf(fmaxf(v[k] * prescale, -65504.f), 65504.f), b = fminf(fmaxf(v[k + 1] * prescale, -65504.f), 65504.f);
        const _Float16 ha = (_Float16)a, hb = (_Float16)b;
        const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
        h[k >> 1] = (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
        l[k >> 1] = (uint32_t)__builtin_bit_cast(uint16_t, la) | ((uint32_t)__builtin_bit_cast(uint16_t, lb) << 16);
    }
    hi = uint4{h[0], h[1], h[2], h[3]};
    lo = uint4{l[0], l[1], l[2], l[3]};
}

// Deterministic: every sum is formed in a fixed order (per thread over its pixels, per block over its threads through LDS,
// per image over the blocks in groupnorm_finalize_kernel) -- with float / f64 atomics the statistics differed in the last
// bit from run to run, and 30 layers of a decoder amplify that to the bf16 level (two decodes of the same latents differed by
// up to 2e-2 on the [0,1] image).  partial: [B, nchunks, G, 2] f64.
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const T* __restrict__ x, double* __restrict__ partial,
                                                              int HW, int C, int G, int ppb) {
    __shared__ float s_part[256][4];         // per thread: sum / sum of squares of its two 4-channel halves
    const int b = blockIdx.y;
    const int c8n = C >> 3;                  // 8-channel slices per pixel
    const int slice = threadIdx.x % c8n, prow = threadIdx.x / c8n, pstep = blockDim.x / c8n;
    const int cpg = C / G;                   // channels per group (4, 8, 16, ...)
    float sum[2] = {0.f, 0.f}, sq[2] = {0.f, 0.f};   // halves: channels [0,4) and [4,8) of the slice
    const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
    const T* xb = x + (int64_t)b * HW * C;
    // four pixels' loads are issued before the first is consumed (a 32-byte load per thread and trip left the kernel at
    // 2.8 TB/s: too few bytes in flight); the sums are still formed pixel by pixel, so the statistics keep their bits
    constexpr int U = 4;
    for (int p = p0 + prow; p < p1; p += U * pstep) {
        float v[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pp = p + u * pstep;
            if (pp < p1) load8(xb + (int64_t)pp * C + slice * 8, v[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u * pstep >= p1) break;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sum[k >> 2] += v[u][k];
                sq[k >> 2] += v[u][k] * v[u][k];
            }
        }
    }
    s_part[threadIdx.x][0] = sum[0]; s_part[threadIdx.x][1] = sq[0];
    s_part[threadIdx.x][2] = sum[1]; s_part[threadIdx.x][3] = sq[1];
    __syncthreads();
    if (threadIdx.x < G) {                   // group g = the 4-channel halves [g*cpg/4, (g+1)*cpg/4) of every pixel row
        const int g = threadIdx.x, h0 = g * cpg / 4, h1 = (g + 1) * cpg / 4;
        double ds = 0.0, dq = 0.0;
        for (int pr = 0; pr < pstep; ++pr)
            for (int h = h0; h < h1; ++h) {
                const float* e = s_part[pr * c8n + (h >> 1)] + (h & 1) * 2;
                ds += (double)e[0];
                dq += (double)e[1];
            }
        double* o = partial + (((int64_t)b * gridDim.x + blockIdx.x) * G + g) * 2;
        o[0] = ds;
        o[1] = dq;
    }
}

// Cancellation guard of the two finalize kernels below.  The partial sums they combine are f32 {sum, sum of squares} over a
// few dozen values each; variance = E[x^2] - E[x]^2 then carries the f32 rounding of the squares, 6e-8 E[x^2] / sqrt(partials) -- invisible
// while |mean| is of the order of the standard deviation (every activation of a working decoder), but relative to the variance it grows
// with (mean / std)^2: 9e-5 of the output at mean / std = 34, 7e-3 at 340 (tests/test_gpu_vae.py).  When the first estimate says
// mean^2 > 256 var the workgroup re-reads its (image, group) and sums the DEVIATIONS from that estimate in f64, in a fixed order
// (thread t: pixels t, t + NT, ...; xor tree; waves in order): exact to f64 whatever the mean.  Never taken on ordinary inputs, so
// their bits are what they were.
template <int NT>
__device__ inline void groupnorm_refine(const float* __restrict__ x, int b, int g, int C, int G, int HW, double cnt, double& mean, double& var,
                                        double* red /* [2 * NT / 64] shared */) {
    const int cpg = C / G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xg = x + (int64_t)b * HW * C + g * cpg;
    double s = 0.0, q = 0.0;
    for (int p = tid; p < HW; p += NT) {
        const float* e = xg + (int64_t)p * C;
        for (int c = 0; c < cpg; ++c) {
            const double d = (double)e[c] - mean;
            s += d;
            q += d * d;
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        s += __shfl_xor(s, m, 64);
        q += __shfl_xor(q, m, 64);
    }
    __syncthreads();                          // (red is reused by the caller)
    if (lane == 0) { red[2 * wave] = s; red[2 * wave + 1] = q; }
    __syncthreads();
    double ss = 0.0, qq = 0.0;
    for (int w = 0; w < NT / 64; ++w) { ss += red[2 * w]; qq += red[2 * w + 1]; }
    const double dm = ss / cnt;
    var = qq / cnt - dm * dm;
    mean += dm;
}

// per-(image, group) mean and 1/std in f32 from the blocks' f64 partial sums (added in block order), once: done per element
// group inside the apply kernel the f64 division (and two f64 loads) per 4 channels slowed a streaming kernel down
__global__ __launch_bounds__(64) void groupnorm_finalize_kernel(const double* __restrict__ partial, float* __restrict__ mr, int B,
                                                                int G, int nchunks, double cnt, float eps,
                                                                const float* __restrict__ x_f32 = nullptr, int C = 0, int HW = 0) {
    __shared__ double red1[2];
    // one wave per (image, group): lane l adds chunks l, l + 64, ... in order, then a fixed xor tree over the lanes -- the same
    // order on every run (the first version walked all chunks in ONE thread: 216 us for the 512 chunks of a 512^2 image)
    const int i = blockIdx.x, lane = threadIdx.x;
    const int b = i / G, g = i - b * G;
    double s = 0.0, q = 0.0;
    for (int c = lane; c < nchunks; c += 64) {
        const double* e = partial + (((int64_t)b * nchunks + c) * G + g) * 2;
        s += e[0];
        q += e[1];
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        s += __shfl_xor(s, m, 64);
        q += __shfl_xor(q, m, 64);
    }
    double mean = s / cnt, var = q / cnt - mean * mean;          // (every lane holds the totals after the xor tree)
    if (x_f32 && mean * mean > 256.0 * var) groupnorm_refine<64>(x_f32, b, g, C, G, HW, cnt, mean, var, red1);
    if (lane == 0) {
        mr[2 * i] = (float)mean;
        mr[2 * i + 1] = rsqrtf(fmaxf((float)var, 0.f) + eps);
    }
}

// The same from the block sums the f16x2 convolution's epilogue leaves (GemmParams::gn_partial: [B * HW / 16][C / 4] {sum, sum of
// squares} f32 over 16 pixels x 4 channels): one 256-thread workgroup per (image, group), thread t adds the image's blocks t, t + 256, ...
// in order, then the fixed xor tree inside each wave and the four wave sums in wave order -- nothing depends on where the image stands
// in the batch.  (One wave per (image, group) took up to 0.56 ms at 512^2: 128 waves on 256 CUs walking 256 strided blocks each.)
__global__ __launch_bounds__(256) void groupnorm_finalize_tiles_kernel(const float* __restrict__ partial, float* __restrict__ mr, int B,
                                                                       int G, int C, int HW, int tile_rows, double cnt, float eps,
                                                                       const float* __restrict__ x_f32) {
    __shared__ double red[8];
    __shared__ double est[2];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = i / G, g = i - b * G;
    const int nch = C >> 2, cpg4 = (C / G) >> 2;
    const int nblk = HW / tile_rows;
    const float* pb = partial + ((int64_t)b * nblk * nch + g * cpg4) * 2;
    double s = 0.0, q = 0.0;
    for (int t = tid; t < nblk; t += 256) {
        const float* e = pb + (int64_t)t * nch * 2;
        for (int c = 0; c < cpg4; ++c) {
            s += (double)e[2 * c];
            q += (double)e[2 * c + 1];
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        s += __shfl_xor(s, m, 64);
        q += __shfl_xor(q, m, 64);
    }
    if (lane == 0) { red[2 * wave] = s; red[2 * wave + 1] = q; }
    __syncthreads();
    if (tid == 0) {
        const double ss = ((red[0] + red[2]) + red[4]) + red[6], qq = ((red[1] + red[3]) + red[5]) + red[7];
        est[0] = ss / cnt;
        est[1] = qq / cnt - est[0] * est[0];
    }
    __syncthreads();
    double mean = est[0], var = est[1];
    if (x_f32 && mean * mean > 256.0 * var) groupnorm_refine<256>(x_f32, b, g, C, G, HW, cnt, mean, var, red);
    if (tid == 0) {
        mr[2 * i] = (float)mean;
        mr[2 * i + 1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
    }
}

// T = bf16_t: bf16 in / bf16 affine / bf16 out.  T = float (split-bf16 mode): f32 in, f32 affine, output row of 3C bf16
// = [hi | hi | lo] of the normalised value.
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const T* __restrict__ x, bf16_t* __restrict__ y,
                                                              const float* __restrict__ mr,
                                                              const T* __restrict__ w, const T* __restrict__ bb,
                                                              int HW, int C, int G, int silu, int pair_only, int64_t total8,
                                                              float prescale = 1.0f) {
    const int c8n = C >> 3, cpg = C / G;
    auto finish = [&](int64_t i, float (&v)[8]) __attribute__((always_inline)) {
        const int slice = i % c8n;
        const int64_t pix = i / c8n;
        const int b = pix / HW;
        float ww[8], bv[8];
        load8(w + slice * 8, ww);
        load8(bb + slice * 8, bv);
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int g = (slice * 8 + hlf * 4) / cpg;
            const float2 m2 = *reinterpret_cast<const float2*>(mr + ((int64_t)b * G + g) * 2);
            const float mu = m2.x, rstd = m2.y;
#pragma unroll
            for (int k = hlf * 4; k < hlf * 4 + 4; ++k) {
                float t = (v[k] - mu) * rstd * ww[k] + bv[k];
                if (silu) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));   // (v_rcp: the IEEE division costs ~10 more instructions per element)
                v[k] = t;
            }
        }
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<uint4*>(y + i * 8) = pack8v(v);
        } else {
            uint4 hi, lo;
            if (pair_only == 2) split8_f16(v, prescale, hi, lo);     // fp16 pair [hi | unwritten | lo]
            else split8(v, hi, lo);
            bf16_t* row = y + pix * 3 * C + slice * 8;
            *reinterpret_cast<uint4*>(row) = hi;
            if (!pair_only) *reinterpret_cast<uint4*>(row + C) = hi;
            *reinterpret_cast<uint4*>(row + 2 * C) = lo;
        }
    };
    // two slices per trip, both loads in flight before the arithmetic of the first
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += 2 * stride) {
        float va[8], vb[8];
        const int64_t j = i + stride;
        load8(x + i * 8, va);
        if (j < total8) load8(x + j * 8, vb);
        finish(i, va);
        if (j < total8) finish(j, vb);
    }
}

// f32 [rows, K] (+ bias[K]) -> bf16 [rows, 3K]: order 0 = [hi | hi | lo] (left operand: activations), 1 = [hi | lo | hi]
// (right operand: weights), 2 = [hi | unwritten | lo] (activations of the dedicated 3x3 kernel, conv_x3.hip, which reads the hi
// and lo thirds only)
__global__ __launch_bounds__(256) void split_x3_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                       bf16_t* __restrict__ out, int K, int order, int64_t total8, float prescale = 1.0f) {
    const int k8n = K >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int slice = i % k8n;
        const int64_t r = i / k8n;
        float v[8];
        load8(x + i * 8, v);
        if (bias) {
            float b8[8];
            load8(bias + slice * 8, b8);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += b8[k];
        }
        uint4 hi, lo;
        if (order == 3) split8_f16(v, prescale, hi, lo);             // fp16 pair [hi | unwritten | lo]
        else split8(v, hi, lo);
        bf16_t* row = out + r * 3 * K + slice * 8;
        *reinterpret_cast<uint4*>(row) = hi;
        if (order < 2) *reinterpret_cast<uint4*>(row + K) = order ? lo : hi;
        *reinterpret_cast<uint4*>(row + 2 * K) = order == 1 ? hi : lo;
    }
}

// y = a + b + bias[c] (f32): the attention block's output projection + bias + skip in the split-bf16 mode
__global__ __launch_bounds__(256) void add_rows_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ bias, float* __restrict__ y, int C,
                                                           int64_t total4) {
    const int c4n = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 u = reinterpret_cast<const float4*>(a)[i];
        if (b) {
            const float4 w = reinterpret_cast<const float4*>(b)[i];
            u.x += w.x; u.y += w.y; u.z += w.z; u.w += w.w;
        }
        if (bias) {
            const float4 w = reinterpret_cast<const float4*>(bias)[i % c4n];
            u.x += w.x; u.y += w.y; u.z += w.z; u.w += w.w;
        }
        reinterpret_cast<float4*>(y)[i] = u;
    }
}

// softmax over f32 rows [rows, n] written as the left operand of the P.V product in the split-bf16 mode:
// out [rows, 3n] bf16 = [hi | hi | lo].  One workgroup per row, three passes over a row that stays in L2.
__global__ __launch_bounds__(256) void softmax_rows_x3_kernel(const float* __restrict__ s, bf16_t* __restrict__ out, int n) {
    __shared__ float red[4];
    const float* r = s + (int64_t)blockIdx.x * n;
    bf16_t* o = out + (int64_t)blockIdx.x * 3 * n;
    const int n8 = n >> 3;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < n8; c += 256) {
        float v[8];
        load8(r + c * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x; c < n8; c += 256) {
        float v[8];
        load8(r + c * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += expf(v[k] - mx);
    }
    const float inv = 1.0f / block_sum<4>(sum, red);
    for (int c = threadIdx.x; c < n8; c += 256) {
        float v[8];
        load8(r + c * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = expf(v[k] - mx) * inv;
        uint4 hi, lo;
        split8(v, hi, lo);
        *reinterpret_cast<uint4*>(o + c * 8) = hi;
        *reinterpret_cast<uint4*>(o + n + c * 8) = hi;
        *reinterpret_cast<uint4*>(o + 2 * n + c * 8) = lo;
    }
}

// in-place softmax over rows of length n (n % 8 == 0, n <= 8192): one wave per row, row kept in registers
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* __restrict__ s, int64_t rows, int n) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    bf16_t* r = s + row * n;
    constexpr int MAXC = 16;
    float v[MAXC][8];
    const int n8 = n >> 3;  // 16-byte chunks in the row
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = i * 64 + lane;
        if (c < n8) {
            unpack8v(*reinterpret_cast<const uint4*>(r + c * 8), v[i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) mx = fmaxf(mx, v[i][k]);
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = i * 64 + lane;
        if (c < n8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[i][k] = __expf(v[i][k] - mx);
                sum += v[i][k];
            }
        }
    }
    const float inv = 1.0f / wave_sum(sum);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = i * 64 + lane;
        if (c < n8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[i][k] *= inv;
            *reinterpret_cast<uint4*>(r + c * 8) = pack8v(v[i]);
        }
    }
}

// z [B,C,H,W] (f32 or bf16) -> NHWC bf16 [B,H,W,Cpad], value z*inv_scale + shift in channels < C, else 0
template <bool X3>
__global__ void latents_to_nhwc_kernel(const void* __restrict__ z, int z_dt, bf16_t* __restrict__ out, int B, int C,
                                       int H, int W, int Cpad, float scaling, float shift) {
    const int64_t total = (int64_t)B * H * W * Cpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % Cpad;
        const int64_t pix = i / Cpad;
        const int w = pix % W, h = (pix / W) % H, b = pix / ((int64_t)W * H);
        float v = 0.f;
        if (c < C) {
            const int64_t src = (((int64_t)b * C + c) * H + h) * W + w;
            v = z_dt == ADVGRPO_BF16 ? bf2f(reinterpret_cast<const bf16_t*>(z)[src]) : reinterpret_cast<const float*>(z)[src];
            v = v / scaling + shift;   // latents / scaling_factor + shift_factor (PF:667), one f32 rounding each
        }
        if constexpr (X3) {          // [hi | hi | lo] along a 3*Cpad channel axis
            const float h = round_bf16(v);
            bf16_t* row = out + pix * 3 * Cpad + c;
            row[0] = f2bf(h); row[Cpad] = f2bf(h); row[2 * Cpad] = f2bf(v - h);
        } else
        out[i] = f2bf(v);
    }
}

// Qwen-Image VAE: de-normalise the latents per channel (z / inv_std + mean, the pipeline's arithmetic) and apply the 1x1x1
// post_quant_conv (y[o] = bias[o] + sum_c P[o][c] z'[c], f32, c ascending), NCHW -> NHWC padded to Cpad channels.  The latent
// tensor is 2 MB per 1024^2 image: one thread per output element, the C strided reads of a pixel hit L2.
template <bool X3>
__global__ void latents_mix_to_nhwc_kernel(const void* __restrict__ z, int z_dt, bf16_t* __restrict__ out, int B, int C, int H, int W,
                                           int Cpad, const float* __restrict__ inv_std, const float* __restrict__ mean,
                                           const float* __restrict__ P, const float* __restrict__ bias) {
    const int64_t total = (int64_t)B * H * W * Cpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = i % Cpad;
        const int64_t pix = i / Cpad;
        const int w = pix % W, h = (pix / W) % H, b = pix / ((int64_t)W * H);
        float v = 0.f;
        if (o < C) {
            v = bias[o];
            for (int c = 0; c < C; ++c) {
                const int64_t src = (((int64_t)b * C + c) * H + h) * W + w;
                float t = z_dt == ADVGRPO_BF16 ? bf2f(reinterpret_cast<const bf16_t*>(z)[src]) : reinterpret_cast<const float*>(z)[src];
                t = t / inv_std[c] + mean[c];
                v = __builtin_fmaf(P[o * C + c], t, v);
            }
        }
        if constexpr (X3) {
            const float hh = round_bf16(v);
            bf16_t* row = out + pix * 3 * Cpad + o;
            row[0] = f2bf(hh); row[Cpad] = f2bf(hh); row[2 * Cpad] = f2bf(v - hh);
        } else
        out[i] = f2bf(v);
    }
}

// y NHWC [B,H,W,ldc] (bf16 or f32; channels 0..2 used) -> image NCHW f32 [B,3,H,W] = clamp(y/2+0.5, 0, 1)
__global__ void image_postprocess_kernel(const void* __restrict__ y, int y_dt, int ldc, float* __restrict__ img, int B,
                                         int H, int W) {
    const int64_t total = (int64_t)B * 3 * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = i % W, h = (i / W) % H, c = (i / ((int64_t)W * H)) % 3, b = i / ((int64_t)W * H * 3);
        const int64_t src = (((int64_t)b * H + h) * W + w) * ldc + c;
        const float v = y_dt == ADVGRPO_BF16 ? bf2f(reinterpret_cast<const bf16_t*>(y)[src])
                                             : reinterpret_cast<const float*>(y)[src];
        img[i] = fminf(fmaxf(v / 2.0f + 0.5f, 0.f), 1.f);
    }
}

}  // namespace advgrpo

using namespace advgrpo;

// pixels per statistics block: 512, more for very large images so that an image has at most 512 blocks
static int advgrpo_groupnorm_ppb(int HW) { return HW <= 512 * 512 ? 512 : (HW + 511) / 512; }

extern "C" int64_t advgrpo_groupnorm_scratch_bytes(int B, int HW, int G) {
    const int ppb = advgrpo_groupnorm_ppb(HW), nchunks = (HW + ppb - 1) / ppb;
    return ((int64_t)B * nchunks * G * 2 + (int64_t)B * G) * 8;
}

extern "C" int advgrpo_groupnorm_nhwc(const void* x, void* y, double* stats, const void* weight, const void* bias, int B,
                                      int HW, int C, int G, float eps, int silu, void* stream) {
    ADVGRPO_CHECK(x && y && stats && weight && bias, "groupnorm: null pointer");
    ADVGRPO_CHECK(B > 0 && HW > 0 && C % 8 == 0 && G > 0 && G <= 64 && C % G == 0 && (C / G) % 4 == 0 && 256 % (C / 8) == 0,
                  "groupnorm: unsupported shape C=%d G=%d", C, G);
    hipStream_t s = as_stream(stream);
    const int ppb = advgrpo_groupnorm_ppb(HW), nchunks = (HW + ppb - 1) / ppb;
    hipLaunchKernelGGL(groupnorm_stats_kernel<bf16_t>, dim3(nchunks, B), dim3(256), 0, s, (const bf16_t*)x, stats, HW, C, G, ppb);
    ADVGRPO_LAUNCH_CHECK();
    // scratch layout: [B, nchunks, G, 2] f64 partial sums, then B*G (mean, 1/std) f32 pairs (advgrpo_groupnorm_scratch_bytes)
    float* mr = reinterpret_cast<float*>(stats + (size_t)B * nchunks * G * 2);
    hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(B * G), dim3(64), 0, s, stats, mr, B, G, nchunks,
                       (double)HW * (C / G), eps);
    ADVGRPO_LAUNCH_CHECK();
    const int64_t total8 = (int64_t)B * HW * (C / 8);
    int64_t blocks = (total8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(groupnorm_apply_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, mr,
                       (const bf16_t*)weight, (const bf16_t*)bias, HW, C, G, silu, 0, total8);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_groupnorm_nhwc_x3(const float* x, void* y3, double* stats, const float* weight, const float* bias,
                                         int B, int HW, int C, int G, float eps, int silu, int pair_only, void* stream) {
    ADVGRPO_CHECK(x && y3 && stats && weight && bias, "groupnorm_x3: null pointer");
    ADVGRPO_CHECK(B > 0 && HW > 0 && C % 8 == 0 && G > 0 && G <= 64 && C % G == 0 && (C / G) % 4 == 0 && 256 % (C / 8) == 0,
                  "groupnorm_x3: unsupported shape C=%d G=%d", C, G);
    hipStream_t s = as_stream(stream);
    const int ppb = advgrpo_groupnorm_ppb(HW), nchunks = (HW + ppb - 1) / ppb;
    hipLaunchKernelGGL(groupnorm_stats_kernel<float>, dim3(nchunks, B), dim3(256), 0, s, x, stats, HW, C, G, ppb);
    ADVGRPO_LAUNCH_CHECK();
    float* mr = reinterpret_cast<float*>(stats + (size_t)B * nchunks * G * 2);
    hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(B * G), dim3(64), 0, s, stats, mr, B, G, nchunks,
                       (double)HW * (C / G), eps, x, C, HW);
    ADVGRPO_LAUNCH_CHECK();
    const int64_t total8 = (int64_t)B * HW * (C / 8);
    int64_t blocks = (total8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(groupnorm_apply_kernel<float>, dim3((int)blocks), dim3(256), 0, s, x, (bf16_t*)y3, mr, weight, bias,
                       HW, C, G, silu, pair_only, total8);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_groupnorm_nhwc_f16x2(const float* x, void* y3, double* stats, const float* weight, const float* bias,
                                            int B, int HW, int C, int G, float eps, int silu, float prescale, const float* tile_partial,
                                            int tile_rows, void* stream) {
    ADVGRPO_CHECK(x && y3 && stats && weight && bias, "groupnorm_f16x2: null pointer");
    ADVGRPO_CHECK(B > 0 && HW > 0 && C % 8 == 0 && G > 0 && G <= 64 && C % G == 0 && (C / G) % 4 == 0 && 256 % (C / 8) == 0,
                  "groupnorm_f16x2: unsupported shape C=%d G=%d", C, G);
    hipStream_t s = as_stream(stream);
    const int ppb = advgrpo_groupnorm_ppb(HW), nchunks = (HW + ppb - 1) / ppb;
    float* mr = reinterpret_cast<float*>(stats + (size_t)B * nchunks * G * 2);
    if (tile_partial) {     // the statistics came out of the producing convolution's epilogue: no pass over x for them
        ADVGRPO_CHECK(tile_rows > 0 && HW % tile_rows == 0, "groupnorm_f16x2: tile_rows %d must divide HW = %d", tile_rows, HW);
        hipLaunchKernelGGL(groupnorm_finalize_tiles_kernel, dim3(B * G), dim3(256), 0, s, tile_partial, mr, B, G, C, HW, tile_rows,
                           (double)HW * (C / G), eps, x);
    } else {
        hipLaunchKernelGGL(groupnorm_stats_kernel<float>, dim3(nchunks, B), dim3(256), 0, s, x, stats, HW, C, G, ppb);
        ADVGRPO_LAUNCH_CHECK();
        hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(B * G), dim3(64), 0, s, stats, mr, B, G, nchunks,
                           (double)HW * (C / G), eps, x, C, HW);
    }
    ADVGRPO_LAUNCH_CHECK();
    const int64_t total8 = (int64_t)B * HW * (C / 8);
    int64_t blocks = (total8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(groupnorm_apply_kernel<float>, dim3((int)blocks), dim3(256), 0, s, x, (bf16_t*)y3, mr, weight, bias,
                       HW, C, G, silu, 2, total8, prescale);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

// Per-pixel RMS norm over the channel axis (the Qwen-Image VAE's QwenImageRMS_norm = F.normalize(x, dim=channels) * sqrt(C) * gamma),
// optional SiLU.  x: [pixels, C] f32 or bf16, gamma f32 [C] (zero for padding channels).  A pixel is held by L = 16 / 32 / 64 lanes
// (8 channels per lane; lanes past C / 8 idle), its sum of squares reduced by a fixed xor tree over those lanes: one read, one write.
// OUT 0: bf16 [pixels, C]; 1: split-bf16 [hi | hi | lo]; 2: [hi | unwritten | lo] (conv3x3_x3 with Cout >= 128).
template <typename T, int L>
__global__ __launch_bounds__(256) void rmsnorm_pixels_kernel(const T* __restrict__ x, bf16_t* __restrict__ y,
                                                             const float* __restrict__ gamma, int64_t pixels, int C, float mult,
                                                             int silu, int out_mode) {
    const int sub = threadIdx.x % L, c8n = C >> 3;
    const int64_t ppb = 256 / L;
    float g8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (sub < c8n) load8(gamma + sub * 8, g8);
#pragma unroll
    for (int k = 0; k < 8; ++k) g8[k] *= mult;
    for (int64_t pix = (int64_t)blockIdx.x * ppb + threadIdx.x / L; pix < pixels; pix += (int64_t)gridDim.x * ppb) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (sub < c8n) load8(x + pix * C + sub * 8, v);
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) ss = __builtin_fmaf(v[k], v[k], ss);
#pragma unroll
        for (int o = 1; o < L; o <<= 1) ss += __shfl_xor(ss, o, 64);
        const float inv = 1.0f / fmaxf(__builtin_sqrtf(ss), 1e-12f);
        if (sub >= c8n) continue;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = v[k] * inv * g8[k];
            if (silu) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
            v[k] = t;
        }
        if (out_mode == 0) {
            *reinterpret_cast<uint4*>(y + pix * C + sub * 8) = pack8v(v);
        } else {
            uint4 hi, lo;
            split8(v, hi, lo);
            bf16_t* row = y + pix * 3 * C + sub * 8;
            *reinterpret_cast<uint4*>(row) = hi;
            if (out_mode == 1) *reinterpret_cast<uint4*>(row + C) = hi;
            *reinterpret_cast<uint4*>(row + 2 * C) = lo;
        }
    }
}

extern "C" int advgrpo_rmsnorm_nhwc(const void* x, int x_dtype, void* y, const float* gamma, int64_t pixels, int C, float mult,
                                    int silu, int out_mode, void* stream) {
    ADVGRPO_CHECK(x && y && gamma && pixels > 0, "rmsnorm_nhwc: null pointer");
    ADVGRPO_CHECK(C % 8 == 0 && C > 0 && C <= 512 && out_mode >= 0 && out_mode <= 2 && (x_dtype == ADVGRPO_F32 || x_dtype == ADVGRPO_BF16),
                  "rmsnorm_nhwc: unsupported shape / mode (C=%d, out_mode=%d, dtype=%d)", C, out_mode, x_dtype);
    hipStream_t s = as_stream(stream);
    const int L = C <= 128 ? 16 : C <= 256 ? 32 : 64;
    int64_t blocks = (pixels + 256 / L - 1) / (256 / L);
    if (blocks > 16384) blocks = 16384;
#define ADVGRPO_RMS_LAUNCH(T, LL)                                                                                                  \
    hipLaunchKernelGGL((rmsnorm_pixels_kernel<T, LL>), dim3((int)blocks), dim3(256), 0, s, (const T*)x, (bf16_t*)y, gamma, pixels, C, \
                       mult, silu, out_mode)
    if (x_dtype == ADVGRPO_F32) {
        if (L == 16) ADVGRPO_RMS_LAUNCH(float, 16); else if (L == 32) ADVGRPO_RMS_LAUNCH(float, 32); else ADVGRPO_RMS_LAUNCH(float, 64);
    } else {
        if (L == 16) ADVGRPO_RMS_LAUNCH(bf16_t, 16); else if (L == 32) ADVGRPO_RMS_LAUNCH(bf16_t, 32); else ADVGRPO_RMS_LAUNCH(bf16_t, 64);
    }
#undef ADVGRPO_RMS_LAUNCH
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_split_f16x2(const float* x, const float* bias, void* out, int64_t rows, int K, float prescale, void* stream) {
    ADVGRPO_CHECK(x && out && rows > 0 && K > 0 && K % 8 == 0, "split_f16x2: need K %% 8 == 0 (K=%d)", K);
    const int64_t total8 = rows * (K / 8);
    int64_t blocks = (total8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_x3_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, bias, (bf16_t*)out, K, 3, total8, prescale);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_split_bf16x3(const float* x, const float* bias, void* out, int64_t rows, int K, int order,
                                    void* stream) {
    ADVGRPO_CHECK(x && out && rows > 0 && K > 0 && K % 8 == 0 && order >= 0 && order <= 2,
                  "split_bf16x3: need K %% 8 == 0 and order 0|1|2 (K=%d)", K);
    const int64_t total8 = rows * (K / 8);
    int64_t blocks = (total8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_x3_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, bias, (bf16_t*)out, K, order,
                       total8);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_add_rows_f32(const float* a, const float* b, const float* bias, float* y, int64_t rows, int C,
                                    void* stream) {
    ADVGRPO_CHECK(a && y && rows > 0 && C > 0 && C % 4 == 0, "add_rows_f32: need C %% 4 == 0 (C=%d)", C);
    const int64_t total4 = rows * (C / 4);
    int64_t blocks = (total4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(add_rows_f32_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), a, b, bias, y, C, total4);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_softmax_rows_x3(const float* s, void* out3, int64_t rows, int n, void* stream) {
    ADVGRPO_CHECK(s && out3 && rows > 0 && n > 0 && n % 8 == 0 && rows < (1ll << 31), "softmax_rows_x3: need n %% 8 == 0 (n=%d)", n);
    hipLaunchKernelGGL(softmax_rows_x3_kernel, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), s, (bf16_t*)out3, n);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

// long rows (n > 8192: the 16384 keys of the mid-block attention at 1024^2): one workgroup per row, three passes over a
// row that stays in L2 (32 KiB at n = 16384) instead of registers; same arithmetic
__global__ __launch_bounds__(256) void softmax_rows_long_kernel(bf16_t* __restrict__ s, int n) {
    __shared__ float red[4];
    bf16_t* r = s + (int64_t)blockIdx.x * n;
    const int n8 = n >> 3;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < n8; c += 256) {
        float v[8];
        unpack8v(*reinterpret_cast<const uint4*>(r + c * 8), v);
#pragma unroll
        for (int k = 0; k < 8; ++k) mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int c = threadIdx.x; c < n8; c += 256) {
        float v[8];
        unpack8v(*reinterpret_cast<const uint4*>(r + c * 8), v);
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += __expf(v[k] - mx);
    }
    const float inv = 1.0f / block_sum<4>(sum, red);
    for (int c = threadIdx.x; c < n8; c += 256) {
        float v[8];
        unpack8v(*reinterpret_cast<const uint4*>(r + c * 8), v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __expf(v[k] - mx) * inv;
        *reinterpret_cast<uint4*>(r + c * 8) = pack8v(v);
    }
}

extern "C" int advgrpo_softmax_rows(void* s, int64_t rows, int n, void* stream) {
    ADVGRPO_CHECK(s && rows > 0 && n > 0 && n % 8 == 0 && rows < (1ll << 31), "softmax_rows: need n %% 8 == 0 (n=%d)", n);
    if (n > 8192) hipLaunchKernelGGL(softmax_rows_long_kernel, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), (bf16_t*)s, n);
    else
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), (bf16_t*)s,
                       rows, n);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_latents_to_nhwc(const void* z, int z_dtype, void* out, int B, int C, int H, int W, int Cpad,
                                       float scaling_factor, float shift_factor, void* stream) {
    ADVGRPO_CHECK(z && out && B > 0 && C > 0 && Cpad >= C, "latents_to_nhwc: bad argument");
    hipLaunchKernelGGL(latents_to_nhwc_kernel<false>, dim3(1024), dim3(256), 0, as_stream(stream), z, z_dtype, (bf16_t*)out, B, C,
                       H, W, Cpad, scaling_factor, shift_factor);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_latents_to_nhwc_x3(const void* z, int z_dtype, void* out3, int B, int C, int H, int W, int Cpad,
                                          float scaling_factor, float shift_factor, void* stream) {
    ADVGRPO_CHECK(z && out3 && B > 0 && C > 0 && Cpad >= C, "latents_to_nhwc_x3: bad argument");
    hipLaunchKernelGGL(latents_to_nhwc_kernel<true>, dim3(1024), dim3(256), 0, as_stream(stream), z, z_dtype, (bf16_t*)out3, B,
                       C, H, W, Cpad, scaling_factor, shift_factor);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_latents_mix_to_nhwc(const void* z, int z_dtype, void* out, int x3, int B, int C, int H, int W, int Cpad,
                                           const float* inv_std, const float* mean, const float* P, const float* bias, void* stream) {
    ADVGRPO_CHECK(z && out && inv_std && mean && P && bias && B > 0 && C > 0 && Cpad >= C, "latents_mix_to_nhwc: bad argument");
    if (x3)
        hipLaunchKernelGGL(latents_mix_to_nhwc_kernel<true>, dim3(1024), dim3(256), 0, as_stream(stream), z, z_dtype, (bf16_t*)out, B, C,
                           H, W, Cpad, inv_std, mean, P, bias);
    else
        hipLaunchKernelGGL(latents_mix_to_nhwc_kernel<false>, dim3(1024), dim3(256), 0, as_stream(stream), z, z_dtype, (bf16_t*)out, B, C,
                           H, W, Cpad, inv_std, mean, P, bias);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_image_postprocess(const void* y, int y_dtype, int ldc, float* image, int B, int H, int W,
                                         void* stream) {
    ADVGRPO_CHECK(y && image && B > 0 && ldc >= 3, "image_postprocess: bad argument");
    hipLaunchKernelGGL(image_postprocess_kernel, dim3(2048), dim3(256), 0, as_stream(stream), y, y_dtype, ldc, image, B, H,
                       W);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
