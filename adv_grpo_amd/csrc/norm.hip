// norm.hip -- HBM-bound row kernels of the transformer / ViT blocks for gfx950.
//
//   layernorm_mod : out = (LN(x) [* w + b]) * (1 + scale[batch]) + shift[batch]   (one or two outputs)
//                   = diffusers AdaLayerNormZero / AdaLayerNormContinuous / nn.LayerNorm as reached
//                   through sd3_pipeline_with_logprob_fast.py:630-637 (MMDiT) and the ViT towers.
//   rmsnorm_heads : in-place per-head RMSNorm(64) * weight on the q and k slices of a packed QKV
//                   buffer (qk_norm="rms_norm" of the SD3.5 attention).
//   timestep_embedding, silu, patchify / unpatchify, l2norm rows.
//
// Shape of the work: every row is read once from HBM with 16-byte lane loads (8 bf16), kept in
// registers for the two-pass mean/variance, reduced with wave64 shuffles (one row per wave, four
// rows per workgroup), and written once -- 4 B of traffic per element against ~10 flops: purely
// bandwidth bound, so no LDS staging and no MFMA.
#include "common.hpp"

namespace advgrpo {

constexpr int LN_MAX_CHUNKS = 8;  // 8-element chunks per lane => D <= 4096 (kernel instantiated for 4 and 8)

__device__ inline void unpack8(const uint4& r, float o[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[2 * k] = bf2f((bf16_t)(w[k] & 0xffffu));
        o[2 * k + 1] = bf2f((bf16_t)(w[k] >> 16));
    }
}
__device__ inline uint4 pack8(const float o[8]) {
    uint4 r;
    r.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
    r.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
    r.z = (uint32_t)f2bf(o[4]) | ((uint32_t)f2bf(o[5]) << 16);
    r.w = (uint32_t)f2bf(o[6]) | ((uint32_t)f2bf(o[7]) << 16);
    return r;
}

struct LnParams {
    const bf16_t* x; int64_t ldx;
    bf16_t* out0; bf16_t* out1; int64_t ldo;
    const bf16_t* w; const bf16_t* b;           // optional affine [D]
    const bf16_t* scale0; const bf16_t* shift0; // optional modulation, row m uses [(m / rows_per_batch) * mod_stride + d]
    const bf16_t* scale1; const bf16_t* shift1; // second modulation for out1
    int64_t mod_stride; int rows_per_batch;
    int M, D; float eps;
    int rms;   // 1: RMSNorm (T5LayerNorm: no mean subtraction, no bias)
    // fp8 outputs (advgrpo_layernorm_mod_fp8): e4m3 codes + one f32 scale per row of the bf16-ROUNDED output, exactly what
    // advgrpo_quant_fp8_rows makes of out0 / out1 (quantize.hip) -- the Linear that follows reads these, and out0 / out1
    // themselves may then be omitted (null)
    uint8_t* q0; uint8_t* q1; float* qs0; float* qs1; int64_t ldq;
};

// e4m3 codes of 8 bf16 values (packed in a uint4) scaled by inv: the arithmetic of quant_fp8_rows_kernel
__device__ __forceinline__ uint2 ln_fp8_codes(const uint4& b, float inv) {
    const uint32_t w[4] = {b.x, b.y, b.z, b.w};
    float f[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[2 * k] = fminf(fmaxf(__builtin_bit_cast(float, w[k] << 16) * inv, -448.0f), 448.0f);
        f[2 * k + 1] = fminf(fmaxf(__builtin_bit_cast(float, w[k] & 0xffff0000u) * inv, -448.0f), 448.0f);
    }
    uint32_t o0 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
    o0 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], (int)o0, true);
    uint32_t o1 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
    o1 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], (int)o1, true);
    return uint2{o0, o1};
}
__device__ __forceinline__ uint32_t ln_amax_bits(const uint4& b, uint32_t m) {
    const uint32_t w[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m = max(m, w[k] & 0x7fffu);
        m = max(m, (w[k] >> 16) & 0x7fffu);
    }
    return m;
}

template <int MAXC, bool FP8>
__device__ __forceinline__ void layernorm_mod_row(const LnParams& p, int row, int lane);

template <int MAXC, bool FP8 = false>
__global__ __launch_bounds__(256) void layernorm_mod_kernel(const LnParams p) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    layernorm_mod_row<MAXC, FP8>(p, row, threadIdx.x & 63);
}

// Two independent problems in ONE launch (advgrpo_layernorm_mod_pair): blocks [0, blocks_a) run a, the rest b.  The MMDiT's
// text-stream norms (154 rows per sample) are launch-bound on their own (6 - 7 us for 7.5 MB) -- as the tail blocks of the image
// stream's launch they cost what their bytes cost.  Same per-row code as the single launch: bit-identical outputs.
struct LnPair { LnParams a, b; int blocks_a; };
template <int MAXC, bool FP8 = false>
__global__ __launch_bounds__(256) void layernorm_mod_pair_kernel(const LnPair pp) {
    const bool first = (int)blockIdx.x < pp.blocks_a;
    const LnParams& p = first ? pp.a : pp.b;
    const int row = ((int)blockIdx.x - (first ? 0 : pp.blocks_a)) * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    layernorm_mod_row<MAXC, FP8>(p, row, threadIdx.x & 63);
}

template <int MAXC, bool FP8>
__device__ __forceinline__ void layernorm_mod_row(const LnParams& p, const int row, const int lane) {
    const int nch = p.D >> 3;
    float v[MAXC][8];
    const bf16_t* xr = p.x + (int64_t)row * p.ldx;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            unpack8(*reinterpret_cast<const uint4*>(xr + c * 8), v[i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[i][k];
        }
    }
    const float mean = p.rms ? 0.f : wave_sum(sum) / (float)p.D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float d = v[i][k] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)p.D + p.eps);
    const int64_t mrow = p.rows_per_batch > 0 ? (int64_t)(row / p.rows_per_batch) * p.mod_stride : 0;
    [[maybe_unused]] uint4 ob0[MAXC], ob1[MAXC];          // the outputs, packed bf16 (kept for the fp8 pass)
    [[maybe_unused]] uint32_t amax0 = 0, amax1 = 0;       // |bf16| compares like its bit pattern
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c >= nch) continue;
        float n[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) n[k] = (v[i][k] - mean) * rstd;
        if (p.rms) {   // T5LayerNorm multiplies the weight into the value already cast to the weight dtype
#pragma unroll
            for (int k = 0; k < 8; ++k) n[k] = round_bf16(n[k]);
        }
        if (p.w) {
            float w[8], bb[8];
            unpack8(*reinterpret_cast<const uint4*>(p.w + c * 8), w);
            if (p.b) unpack8(*reinterpret_cast<const uint4*>(p.b + c * 8), bb);
#pragma unroll
            for (int k = 0; k < 8; ++k) n[k] = n[k] * w[k] + (p.b ? bb[k] : 0.f);
        }
        float o[8];
        if (p.scale0) {
            float sc[8], sh[8];
            unpack8(*reinterpret_cast<const uint4*>(p.scale0 + mrow + c * 8), sc);
            unpack8(*reinterpret_cast<const uint4*>(p.shift0 + mrow + c * 8), sh);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = n[k] * (1.0f + sc[k]) + sh[k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = n[k];
        }
        ob0[i] = pack8(o);
        if (p.out0) *reinterpret_cast<uint4*>(p.out0 + (int64_t)row * p.ldo + c * 8) = ob0[i];
        if (FP8) amax0 = ln_amax_bits(ob0[i], amax0);
        if (p.out1 || (FP8 && p.q1)) {
            float sc[8], sh[8];
            unpack8(*reinterpret_cast<const uint4*>(p.scale1 + mrow + c * 8), sc);
            unpack8(*reinterpret_cast<const uint4*>(p.shift1 + mrow + c * 8), sh);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = n[k] * (1.0f + sc[k]) + sh[k];
            ob1[i] = pack8(o);
            if (p.out1) *reinterpret_cast<uint4*>(p.out1 + (int64_t)row * p.ldo + c * 8) = ob1[i];
            if (FP8) amax1 = ln_amax_bits(ob1[i], amax1);
        }
    }
    if constexpr (FP8) {   // second pass over the packed outputs kept in registers: row maximum -> scale -> codes
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint8_t* q = h ? p.q1 : p.q0;
            if (!q) continue;
            const float amax = wave_allreduce(bf2f((bf16_t)(h ? amax1 : amax0)), [](float a, float b) { return fmaxf(a, b); });
            const float scale = amax > 0.f ? fmaxf(amax * (1.0f / 448.0f), 1.17549435e-38f) : 1.0f;   // (never subnormal / zero: 1 / scale stays finite)
            const float inv = 1.0f / scale;
            if (lane == 0) (h ? p.qs1 : p.qs0)[row] = scale;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = lane + i * 64;
                if (c >= nch) continue;
                *reinterpret_cast<uint2*>(q + (int64_t)row * p.ldq + c * 8) = ln_fp8_codes(h ? ob1[i] : ob0[i], inv);
            }
        }
    }
}

// in-place RMSNorm over 64-wide heads: buffer rows [M, ld], heads at columns [col0, col0 + nheads*64);
// head hh uses weight w[(hh / heads_per_weight) * 64 ...] (q heads then k heads => heads_per_weight = H)
__global__ __launch_bounds__(256) void rmsnorm_heads_kernel(bf16_t* __restrict__ buf, int64_t ld, int M, int col0,
                                                            int nheads, const bf16_t* __restrict__ w,
                                                            int heads_per_weight, float eps, int seg_rows,
                                                            int64_t seg_stride, int64_t seg_off, float* __restrict__ rs_out) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    int64_t row = m;
    if (seg_rows > 0) row = (int64_t)(m / seg_rows) * seg_stride + seg_off + (m % seg_rows);
    bf16_t* r = buf + row * ld + col0;
    const int sub = lane & 7;  // 8 lanes x 8 elements = one head
    for (int h0 = 0; h0 < nheads; h0 += 8) {
        const int hh = h0 + (lane >> 3);
        if (hh >= nheads) break;  // trailing lanes idle (nheads is a multiple of 8 for every model here)
        float v[8], ww[8];
        unpack8(*reinterpret_cast<const uint4*>(r + hh * 64 + sub * 8), v);
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sq += v[k] * v[k];
        sq += __shfl_xor(sq, 1, 64);
        sq += __shfl_xor(sq, 2, 64);
        sq += __shfl_xor(sq, 4, 64);
        const float rs = rsqrtf(sq * (1.0f / 64.0f) + eps);
        if (rs_out && sub == 0) rs_out[row * nheads + hh] = rs;   // saved for the backward
        unpack8(*reinterpret_cast<const uint4*>(w + (hh / heads_per_weight) * 64 + sub * 8), ww);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = round_bf16(v[k] * rs) * ww[k];  // x.to(bf16) * weight
        *reinterpret_cast<uint4*>(r + hh * 64 + sub * 8) = pack8(v);
    }
}


// Per-head RMSNorm (affine) + rotary embedding, in place, on the q | k column range of a packed joint QKV buffer: the
// attn.norm_q / norm_k / norm_added_q / norm_added_k + apply_rotary_emb_qwen steps of diffusers' QwenDoubleStreamAttnProcessor2_0
// (the Qwen-Image MMDiT of BASELINE config 5; head dim 128 -- a head spans two wave tiles of the eight-phase GEMM, so its
// QK-norm epilogue class, which owns one 64-wide head per wave tile, does not apply).  One wave per token row, HD / 8 lanes per
// head; the token's rotary row (cos, sin interleaved, f32) is loaded once and serves all of its heads.  Arithmetic in the
// reference's order: bf16(bf16(x * rsqrt(mean x^2 + eps)) * w), then the complex product with (cos + i sin) on adjacent
// (even, odd) pairs in f32, one bf16 rounding.
template <int HD>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(bf16_t* __restrict__ buf, int64_t ld, int rows, int S, int n_first,
                                                           int col0, int nheads, const bf16_t* __restrict__ w_first,
                                                           const bf16_t* __restrict__ w_rest, int heads_per_weight, float eps,
                                                           const float* __restrict__ rope, float* __restrict__ rs_out) {
    constexpr int LPH = HD / 8, HPP = 64 / LPH;          // lanes per head, heads per pass
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int s = row % S;
    const bf16_t* w = s < n_first ? w_first : w_rest;
    bf16_t* r = buf + (int64_t)row * ld + col0;
    const int sub = lane % LPH;
    float cs[8];                                          // (cos, sin) of this lane's four pairs
    if (rope) {
        const float4 a = *reinterpret_cast<const float4*>(rope + (int64_t)s * HD + sub * 8);
        const float4 b = *reinterpret_cast<const float4*>(rope + (int64_t)s * HD + sub * 8 + 4);
        cs[0] = a.x; cs[1] = a.y; cs[2] = a.z; cs[3] = a.w; cs[4] = b.x; cs[5] = b.y; cs[6] = b.z; cs[7] = b.w;
    }
    for (int h0 = 0; h0 < nheads; h0 += HPP) {
        const int hh = h0 + lane / LPH;
        if (hh >= nheads) break;
        float v[8], ww[8];
        unpack8(*reinterpret_cast<const uint4*>(r + hh * HD + sub * 8), v);
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sq += v[k] * v[k];
        sq = group8_sum(sq);
        if (LPH == 16) {                                  // the other half of the 16-lane row (DPP row_mirror)
            sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x140, 0xf, 0xf, true));
        }
        const float rs = rsqrtf(sq * (1.0f / HD) + eps);
        if (rs_out && sub == 0) rs_out[(int64_t)row * nheads + hh] = rs;
        unpack8(*reinterpret_cast<const uint4*>(w + (hh / heads_per_weight) * HD + sub * 8), ww);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = round_bf16(round_bf16(v[k] * rs) * ww[k]);
        if (rope) {
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const float a = v[k], b = v[k + 1];
                v[k] = a * cs[k] - b * cs[k + 1];
                v[k + 1] = a * cs[k + 1] + b * cs[k];
            }
        }
        *reinterpret_cast<uint4*>(r + hh * HD + sub * 8) = pack8(v);
    }
}

// Rotary embedding in the "rotate_half" form (pairs (i, i + HD / 2)) of the Qwen2.5-VL language model, in place on the nheads heads at
// columns [col0, col0 + nheads * HD) of token rows [rows, ld]: out[i] = x[i] cos_i - x[i + h] sin_i, out[i + h] = x[i + h] cos_i + x[i] sin_i
// (= q * cos + rotate_half(q) * sin with the duplicated-halves tables of Qwen2RotaryEmbedding); cs [T, HD / 2] holds (cos, sin) pairs f32,
// position = row % T.  One wave per token row, one lane per pair of a 128-wide head.
__global__ __launch_bounds__(256) void rope_half_kernel(bf16_t* __restrict__ buf, int64_t ld, int rows, int T, int col0, int nheads, int hd,
                                                        const float* __restrict__ cs) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int half = hd >> 1, t = row % T;
    bf16_t* r = buf + (int64_t)row * ld + col0;
    for (int i = lane; i < half; i += 64) {
        const float co = cs[((int64_t)t * half + i) * 2], si = cs[((int64_t)t * half + i) * 2 + 1];
        for (int h = 0; h < nheads; ++h) {
            const float a = bf2f(r[h * hd + i]), b = bf2f(r[h * hd + half + i]);
            r[h * hd + i] = f2bf(a * co - b * si);
            r[h * hd + half + i] = f2bf(b * co + a * si);
        }
    }
}

// softmax over the first (r % n) + 1 columns of score row r (causal attention over materialised scores, query index = r % n); the
// other columns are written as zeros.  One wave per row, n <= 512.
__global__ __launch_bounds__(256) void softmax_rows_causal_kernel(const float* __restrict__ sc, bf16_t* __restrict__ p16, int64_t rows, int n) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int last = (int)(r % n);
    float s[8], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        s[i] = c <= last ? sc[r * n + c] : -INFINITY;
        mx = fmaxf(mx, s[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = __expf(s[i] - mx); sum += s[i]; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        if (c < n) p16[r * n + c] = f2bf(s[i] * inv);
    }
}

// sinusoidal timestep embedding, diffusers get_timestep_embedding(t, 256, flip_sin_to_cos=True,
// downscale_freq_shift=0): [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(1e4) i / 128); optional SiLU-free.
__global__ void timestep_embedding_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int B, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= B * half) return;
    const int b = i / half, k = i % half;
    const float f = expf(-9.210340371976184f * (float)k / (float)half);
    const float a = t[b] * f;
    out[(int64_t)b * dim + k] = f2bf(cosf(a));
    out[(int64_t)b * dim + half + k] = f2bf(sinf(a));
}

// y = act(x [+ x2]) elementwise on bf16 (n % 8 == 0)
__global__ __launch_bounds__(256) void unary_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ x2,
                                                    bf16_t* __restrict__ y, int64_t n, int act) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n;
         i += (int64_t)gridDim.x * blockDim.x * 8) {
        float v[8], u[8];
        unpack8(*reinterpret_cast<const uint4*>(x + i), v);
        if (x2) {
            unpack8(*reinterpret_cast<const uint4*>(x2 + i), u);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += u[k];
        }
        if (act == 3) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = v[k] / (1.0f + __expf(-v[k]));
        }
        *reinterpret_cast<uint4*>(y + i) = pack8(v);
    }
}

// latents [B,C,H,W] (f32 or bf16) -> patch rows [B*(H/2)*(W/2), C*4] bf16, column = c*4 + py*2 + px
// (= Conv2d(C, D, k=2, s=2) weight flattened [D, C*2*2])
__global__ void patchify_kernel(const void* __restrict__ x, int x_dt, bf16_t* __restrict__ out, int B, int C, int H,
                                int W) {
    const int64_t total = (int64_t)B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = i % W, h = (i / W) % H, c = (i / ((int64_t)W * H)) % C, b = i / ((int64_t)W * H * C);
        const float v = x_dt == ADVGRPO_BF16 ? bf2f(reinterpret_cast<const bf16_t*>(x)[i])
                                             : reinterpret_cast<const float*>(x)[i];
        const int64_t tok = ((int64_t)b * (H / 2) + h / 2) * (W / 2) + w / 2;
        out[tok * (C * 4) + c * 4 + (h & 1) * 2 + (w & 1)] = f2bf(v);
    }
}
// tokens [B*(H/2)*(W/2), 4*C] (column = (py*2+px)*C + c) -> [B,C,H,W] in out_dt
__global__ void unpatchify_kernel(const bf16_t* __restrict__ tok, void* __restrict__ out, int out_dt, int B, int C,
                                  int H, int W) {
    const int64_t total = (int64_t)B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = i % W, h = (i / W) % H, c = (i / ((int64_t)W * H)) % C, b = i / ((int64_t)W * H * C);
        const int64_t t = ((int64_t)b * (H / 2) + h / 2) * (W / 2) + w / 2;
        const bf16_t v = tok[t * (C * 4) + ((h & 1) * 2 + (w & 1)) * C + c];
        if (out_dt == ADVGRPO_BF16) reinterpret_cast<bf16_t*>(out)[i] = v;
        else reinterpret_cast<float*>(out)[i] = bf2f(v);
    }
}

}  // namespace advgrpo

using namespace advgrpo;

extern "C" int advgrpo_layernorm_mod(const void* x, int64_t ldx, void* out0, void* out1, int64_t ldo, const void* w,
                                     const void* b, const void* scale0, const void* shift0, const void* scale1,
                                     const void* shift1, int64_t mod_stride, int rows_per_batch, int M, int D,
                                     float eps, void* stream) {
    ADVGRPO_CHECK(x && out0, "layernorm_mod: null pointer");
    ADVGRPO_CHECK(M > 0 && D > 0 && D % 8 == 0 && D <= LN_MAX_CHUNKS * 512, "layernorm_mod: need D %% 8 == 0, D <= %d (D=%d)",
                  LN_MAX_CHUNKS * 512, D);
    ADVGRPO_CHECK(ldx % 8 == 0 && ldo % 8 == 0 && mod_stride % 8 == 0, "layernorm_mod: pitches must be multiples of 8");
    ADVGRPO_CHECK((scale0 == nullptr) == (shift0 == nullptr), "layernorm_mod: scale0/shift0 come together");
    ADVGRPO_CHECK(!out1 || (scale1 && shift1), "layernorm_mod: out1 needs scale1/shift1");
    LnParams p{(const bf16_t*)x, ldx, (bf16_t*)out0, (bf16_t*)out1, ldo, (const bf16_t*)w, (const bf16_t*)b,
               (const bf16_t*)scale0, (const bf16_t*)shift0, (const bf16_t*)scale1, (const bf16_t*)shift1,
               mod_stride, rows_per_batch, M, D, eps, 0, nullptr, nullptr, nullptr, nullptr, 0};
    if (D <= 2048) hipLaunchKernelGGL(layernorm_mod_kernel<4>, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(layernorm_mod_kernel<8>, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

static int ln_from_desc(const advgrpo_ln_desc* d, LnParams& p, bool& fp8) {
    ADVGRPO_CHECK(d && d->x && (d->out0 || d->q0), "layernorm_mod_pair: null pointer");
    ADVGRPO_CHECK(d->M > 0 && d->D > 0 && d->D % 8 == 0 && d->D <= LN_MAX_CHUNKS * 512, "layernorm_mod_pair: need D %% 8 == 0, D <= %d (D=%d)",
                  LN_MAX_CHUNKS * 512, d->D);
    ADVGRPO_CHECK(d->ldx % 8 == 0 && d->ldo % 8 == 0 && d->mod_stride % 8 == 0 && d->ldq % 8 == 0, "layernorm_mod_pair: pitches must be multiples of 8");
    ADVGRPO_CHECK((d->scale0 == nullptr) == (d->shift0 == nullptr), "layernorm_mod_pair: scale0/shift0 come together");
    ADVGRPO_CHECK(!(d->out1 || d->q1) || (d->scale1 && d->shift1), "layernorm_mod_pair: a second output needs scale1/shift1");
    ADVGRPO_CHECK(!d->q0 || (d->qs0 && d->ldq >= d->D), "layernorm_mod_pair: q0 needs qs0 and ldq >= D");
    ADVGRPO_CHECK(!d->q1 || (d->q0 && d->qs1), "layernorm_mod_pair: q1 needs q0 and qs1");
    p = LnParams{(const bf16_t*)d->x, d->ldx, (bf16_t*)d->out0, (bf16_t*)d->out1, d->ldo, (const bf16_t*)d->w, (const bf16_t*)d->b,
                 (const bf16_t*)d->scale0, (const bf16_t*)d->shift0, (const bf16_t*)d->scale1, (const bf16_t*)d->shift1,
                 d->mod_stride, d->rows_per_batch, d->M, d->D, d->eps, 0, (uint8_t*)d->q0, (uint8_t*)d->q1, d->qs0, d->qs1, d->ldq};
    fp8 = d->q0 != nullptr;
    return 0;
}

/* Two advgrpo_layernorm_mod / _fp8 problems in one launch (the image-stream and text-stream norms of an MMDiT block). */
extern "C" int advgrpo_layernorm_mod_pair(const advgrpo_ln_desc* a, const advgrpo_ln_desc* b, void* stream) {
    LnPair pp{};
    bool fa = false, fb = false;
    if (int rc = ln_from_desc(a, pp.a, fa)) return rc;
    if (int rc = ln_from_desc(b, pp.b, fb)) return rc;
    ADVGRPO_CHECK(fa == fb, "layernorm_mod_pair: both problems bf16 or both fp8");
    pp.blocks_a = (pp.a.M + 3) / 4;
    const dim3 grid(pp.blocks_a + (pp.b.M + 3) / 4);
    const bool wide = pp.a.D > 2048 || pp.b.D > 2048;
    hipStream_t s = as_stream(stream);
    if (fa) {
        if (wide) hipLaunchKernelGGL((layernorm_mod_pair_kernel<8, true>), grid, dim3(256), 0, s, pp);
        else hipLaunchKernelGGL((layernorm_mod_pair_kernel<4, true>), grid, dim3(256), 0, s, pp);
    } else {
        if (wide) hipLaunchKernelGGL((layernorm_mod_pair_kernel<8, false>), grid, dim3(256), 0, s, pp);
        else hipLaunchKernelGGL((layernorm_mod_pair_kernel<4, false>), grid, dim3(256), 0, s, pp);
    }
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

/* layernorm_mod whose outputs also (or only: out0 / out1 may be null) leave as fp8 rows for the fp8 Linears: q0 / q1 [M, D]
 * e4m3 codes (pitch ldq bytes) and qs0 / qs1 [M] f32 scales, bit for bit what advgrpo_quant_fp8_rows makes of out0 / out1. */
extern "C" int advgrpo_layernorm_mod_fp8(const void* x, int64_t ldx, void* out0, void* out1, int64_t ldo, const void* w,
                                         const void* b, const void* scale0, const void* shift0, const void* scale1,
                                         const void* shift1, int64_t mod_stride, int rows_per_batch, int M, int D, float eps,
                                         void* q0, float* qs0, void* q1, float* qs1, int64_t ldq, void* stream) {
    ADVGRPO_CHECK(x && q0 && qs0, "layernorm_mod_fp8: null pointer");
    ADVGRPO_CHECK(M > 0 && D > 0 && D % 8 == 0 && D <= LN_MAX_CHUNKS * 512, "layernorm_mod_fp8: need D %% 8 == 0, D <= %d (D=%d)",
                  LN_MAX_CHUNKS * 512, D);
    ADVGRPO_CHECK(ldx % 8 == 0 && ldo % 8 == 0 && mod_stride % 8 == 0 && ldq % 8 == 0 && ldq >= D,
                  "layernorm_mod_fp8: pitches must be multiples of 8");
    ADVGRPO_CHECK((scale0 == nullptr) == (shift0 == nullptr), "layernorm_mod_fp8: scale0/shift0 come together");
    ADVGRPO_CHECK(!(out1 || q1) || (scale1 && shift1), "layernorm_mod_fp8: a second output needs scale1/shift1");
    ADVGRPO_CHECK(!q1 || qs1, "layernorm_mod_fp8: q1 needs qs1");
    LnParams p{(const bf16_t*)x, ldx, (bf16_t*)out0, (bf16_t*)out1, ldo, (const bf16_t*)w, (const bf16_t*)b,
               (const bf16_t*)scale0, (const bf16_t*)shift0, (const bf16_t*)scale1, (const bf16_t*)shift1,
               mod_stride, rows_per_batch, M, D, eps, 0, (uint8_t*)q0, (uint8_t*)q1, qs0, qs1, ldq};
    if (D <= 2048) hipLaunchKernelGGL((layernorm_mod_kernel<4, true>), dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL((layernorm_mod_kernel<8, true>), dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

/* T5LayerNorm (transformers T5: x * rsqrt(mean(x^2) + eps) * w, statistics in f32): every norm of the T5-XXL text
 * encoder behind encode_prompt (adv_grpo/diffusers_patch/train_dreambooth_lora_sd3.py:19-56). */
extern "C" int advgrpo_rmsnorm_rows(const void* x, int64_t ldx, void* out, int64_t ldo, const void* w, int M, int D, float eps,
                                    void* stream) {
    ADVGRPO_CHECK(x && out && w, "rmsnorm_rows: null pointer");
    ADVGRPO_CHECK(M > 0 && D > 0 && D % 8 == 0 && D <= LN_MAX_CHUNKS * 512, "rmsnorm_rows: need D %% 8 == 0, D <= %d (D=%d)",
                  LN_MAX_CHUNKS * 512, D);
    ADVGRPO_CHECK(ldx % 8 == 0 && ldo % 8 == 0, "rmsnorm_rows: pitches must be multiples of 8");
    LnParams p{(const bf16_t*)x, ldx, (bf16_t*)out, nullptr, ldo, (const bf16_t*)w, nullptr, nullptr, nullptr, nullptr, nullptr,
               0, 0, M, D, eps, 1};
    if (D <= 2048) hipLaunchKernelGGL(layernorm_mod_kernel<4>, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(layernorm_mod_kernel<8>, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_rmsnorm_heads(void* buf, int64_t ld, int M, int col0, int nheads, const void* weight,
                                     int heads_per_weight, float eps, int seg_rows, int64_t seg_stride,
                                     int64_t seg_off, float* rs_out, void* stream) {
    ADVGRPO_CHECK(buf && weight && M > 0 && nheads > 0 && heads_per_weight > 0, "rmsnorm_heads: bad argument");
    ADVGRPO_CHECK(ld % 8 == 0 && col0 % 8 == 0, "rmsnorm_heads: pitch/offset must be multiples of 8");
    hipLaunchKernelGGL(rmsnorm_heads_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), (bf16_t*)buf, ld, M,
                       col0, nheads, (const bf16_t*)weight, heads_per_weight, eps, seg_rows, seg_stride, seg_off, rs_out);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}


extern "C" int advgrpo_rope_half(void* buf, int64_t ld, int rows, int T, int col0, int nheads, int head_dim, const float* cos_sin, void* stream) {
    ADVGRPO_CHECK(buf && cos_sin && rows > 0 && T > 0 && nheads > 0 && head_dim > 0 && head_dim % 2 == 0, "rope_half: bad argument");
    hipLaunchKernelGGL(rope_half_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), (bf16_t*)buf, ld, rows, T, col0, nheads, head_dim,
                       cos_sin);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_softmax_rows_causal(const float* sc, void* p16, int64_t rows, int n, void* stream) {
    ADVGRPO_CHECK(sc && p16 && rows > 0 && n > 0 && n <= 512, "softmax_rows_causal: bad argument (n=%d)", n);
    hipLaunchKernelGGL(softmax_rows_causal_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), sc, (bf16_t*)p16, rows, n);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_qk_norm_rope(void* buf, int64_t ld, int rows, int S, int n_first, int col0, int nheads, int head_dim,
                                    const void* w_first, const void* w_rest, int heads_per_weight, float eps,
                                    const float* rope, float* rs_out, void* stream) {
    ADVGRPO_CHECK(buf && w_first && w_rest && rows > 0 && S > 0 && nheads > 0 && heads_per_weight > 0, "qk_norm_rope: bad argument");
    ADVGRPO_CHECK(head_dim == 64 || head_dim == 128, "qk_norm_rope: head_dim %d not supported (64, 128)", head_dim);
    ADVGRPO_CHECK(ld % 8 == 0 && col0 % 8 == 0, "qk_norm_rope: pitch/offset must be multiples of 8");
    ADVGRPO_CHECK(!rope || (reinterpret_cast<uintptr_t>(rope) & 15) == 0, "qk_norm_rope: the rotary table must be 16-byte aligned");
    const dim3 grid((rows + 3) / 4);
    if (head_dim == 128)
        hipLaunchKernelGGL(qk_norm_rope_kernel<128>, grid, dim3(256), 0, as_stream(stream), (bf16_t*)buf, ld, rows, S, n_first, col0,
                           nheads, (const bf16_t*)w_first, (const bf16_t*)w_rest, heads_per_weight, eps, rope, rs_out);
    else
        hipLaunchKernelGGL(qk_norm_rope_kernel<64>, grid, dim3(256), 0, as_stream(stream), (bf16_t*)buf, ld, rows, S, n_first, col0,
                           nheads, (const bf16_t*)w_first, (const bf16_t*)w_rest, heads_per_weight, eps, rope, rs_out);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_timestep_embedding(const float* t, void* out, int B, int dim, void* stream) {
    ADVGRPO_CHECK(t && out && B > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: bad argument");
    const int n = B * dim / 2;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), t,
                       (bf16_t*)out, B, dim);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_unary(const void* x, const void* x2, void* y, int64_t n, int act, void* stream) {
    ADVGRPO_CHECK(x && y && n > 0 && n % 8 == 0, "unary: need n %% 8 == 0");
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(unary_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                       (const bf16_t*)x2, (bf16_t*)y, n, act);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_patchify(const void* x, int x_dtype, void* out, int B, int C, int H, int W, void* stream) {
    ADVGRPO_CHECK(x && out && B > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, "patchify: bad argument");
    hipLaunchKernelGGL(patchify_kernel, dim3(1024), dim3(256), 0, as_stream(stream), x, x_dtype, (bf16_t*)out, B, C, H,
                       W);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_unpatchify(const void* tokens, void* out, int out_dtype, int B, int C, int H, int W,
                                  void* stream) {
    ADVGRPO_CHECK(tokens && out && B > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, "unpatchify: bad argument");
    hipLaunchKernelGGL(unpatchify_kernel, dim3(1024), dim3(256), 0, as_stream(stream), (const bf16_t*)tokens, out,
                       out_dtype, B, C, H, W);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
