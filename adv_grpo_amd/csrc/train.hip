// train.hip -- HBM-bound kernels of the G-step (backward of the row ops, optimiser, EMA) for gfx950.
//
// Reference sites: loss.backward() / clip_grad_norm_ / AdamW.step / ema.step,
// scripts/train_sd3_fast_pickscore.py:1165-1171,1186-1187, adv_grpo/ema.py:39-52; the row-op backwards are
// the autograd of diffusers' AdaLayerNormZero / RMSNorm / gated residuals inside the transformer call of
// compute_log_prob (TP:233-267).
//
//   transpose_bf16        [R,C] -> [C,R]  (activations / gradients for the LoRA weight-gradient GEMMs, whose
//                         contraction runs over the token axis)
//   layernorm_mod_bwd     dx = dres + LN'(x) applied to (dy0*(1+scale0) [+ dy1*(1+scale1)])
//   rmsnorm_heads_bwd     in place on the packed dq|dk gradient, from the normalised q|k and the saved 1/rms
//   gate_mul              y[m,:] = g[m / rows, :] * x[m,:]   (gated residual branches)
//   adamw / sumsq / ema   fused optimiser step on the flat LoRA parameter vector (f32 master + bf16 copy)
// All are single streaming passes with 16-byte lane accesses; reductions with wave64 shuffles.
#include "common.hpp"

namespace advgrpo {

__device__ inline void unpack8t(const uint4& r, float o[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[2 * k] = bf2f((bf16_t)(w[k] & 0xffffu));
        o[2 * k + 1] = bf2f((bf16_t)(w[k] >> 16));
    }
}
__device__ inline uint4 pack8t(const float o[8]) {
    uint4 r;
    r.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
    r.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
    r.z = (uint32_t)f2bf(o[4]) | ((uint32_t)f2bf(o[5]) << 16);
    r.w = (uint32_t)f2bf(o[6]) | ((uint32_t)f2bf(o[7]) << 16);
    return r;
}

// ---- 64x64 tile transpose through LDS (padded to 65 to dodge bank conflicts); in [R, ldi] -> out [C, ldo]
// input row r is read from (r / seg_rows) * seg_stride + seg_off + r % seg_rows when seg_rows > 0; output columns
// [R, Rpad) are written as zeros (the weight-gradient GEMMs contract over a multiple of 64 tokens).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R,
                                                             int C, int64_t ldi, int64_t ldo, int Rpad, int seg_rows,
                                                             int64_t seg_stride, int64_t seg_off) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        int64_t rr = r;
        if (seg_rows > 0) rr = (int64_t)(r / seg_rows) * seg_stride + seg_off + (r % seg_rows);
        tile[i][tx] = (r < R && c < C) ? in[rr * ldi + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < Rpad) out[(int64_t)c * ldo + r] = tile[tx][i];
    }
}

// ---- LayerNorm(no affine) + modulation backward.  One row per wave (D <= 4096).
struct LnBwdParams {
    const bf16_t* x; int64_t ldx;
    const bf16_t* dy0; const bf16_t* dy1; int64_t lddy;
    const bf16_t* scale0; const bf16_t* scale1; int64_t mod_stride; int rows_per_batch;
    const bf16_t* dres; bf16_t* dx; int64_t lddx;
    int M, D; float eps;
    // optional: up to two gated copies of the result, gout_k[m, :] = gate_k[m / rows_per_batch, :] * bf16(dx[m, :]) (dense rows of D) -- the
    // left operands of the data-gradient GEMMs of the gated projections that consume dx next (what gate_mul_kernel computes from the stored dx,
    // same bits, without reading it back)
    const bf16_t* gate_a; bf16_t* gout_a; const bf16_t* gate_b; bf16_t* gout_b; int64_t gate_stride;
};
template <int MAXC>     // 512-column chunks per row: 4 up to D = 2048 (SD3.5-medium), 8 up to 4096 (SD3.5-large: D = 2432)
__global__ __launch_bounds__(256) void layernorm_mod_bwd_kernel(const LnBwdParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int nch = p.D >> 3;
    const int64_t mrow = p.rows_per_batch > 0 ? (int64_t)(row / p.rows_per_batch) * p.mod_stride : 0;
    float xv[MAXC][8], gv[MAXC][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            unpack8t(*reinterpret_cast<const uint4*>(p.x + (int64_t)row * p.ldx + c * 8), xv[i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += xv[i][k];
        }
    }
    const float mean = wave_sum(sum) / (float)p.D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { xv[i][k] -= mean; sq += xv[i][k] * xv[i][k]; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)p.D + p.eps);
    // g = d xhat = dy0*(1+scale0) [+ dy1*(1+scale1)] ; dx = rstd*(g - mean(g) - xhat*mean(g*xhat))
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            float d0[8], s0[8];
            unpack8t(*reinterpret_cast<const uint4*>(p.dy0 + (int64_t)row * p.lddy + c * 8), d0);
            if (p.scale0) unpack8t(*reinterpret_cast<const uint4*>(p.scale0 + mrow + c * 8), s0);
#pragma unroll
            for (int k = 0; k < 8; ++k) gv[i][k] = d0[k] * (p.scale0 ? 1.0f + s0[k] : 1.0f);
            if (p.dy1) {
                unpack8t(*reinterpret_cast<const uint4*>(p.dy1 + (int64_t)row * p.lddy + c * 8), d0);
                unpack8t(*reinterpret_cast<const uint4*>(p.scale1 + mrow + c * 8), s0);
#pragma unroll
                for (int k = 0; k < 8; ++k) gv[i][k] += d0[k] * (1.0f + s0[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                xv[i][k] *= rstd;   // xhat
                sg += gv[i][k];
                sgx += gv[i][k] * xv[i][k];
            }
        }
    }
    const float mg = wave_sum(sg) / (float)p.D, mgx = wave_sum(sgx) / (float)p.D;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            float o[8], r[8];
            if (p.dres) unpack8t(*reinterpret_cast<const uint4*>(p.dres + (int64_t)row * p.lddx + c * 8), r);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = rstd * (gv[i][k] - mg - xv[i][k] * mgx) + (p.dres ? r[k] : 0.f);
            const uint4 packed = pack8t(o);
            *reinterpret_cast<uint4*>(p.dx + (int64_t)row * p.lddx + c * 8) = packed;
            if (p.gate_a) {
                const int64_t grow = (int64_t)(row / p.rows_per_batch) * p.gate_stride;
                float a[8], g[8];
                unpack8t(packed, o);                         // the ROUNDED gradient is what the gated branch multiplies
                unpack8t(*reinterpret_cast<const uint4*>(p.gate_a + grow + c * 8), g);
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = o[k] * g[k];
                *reinterpret_cast<uint4*>(p.gout_a + (int64_t)row * p.D + c * 8) = pack8t(a);
                if (p.gate_b) {
                    unpack8t(*reinterpret_cast<const uint4*>(p.gate_b + grow + c * 8), g);
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[k] = o[k] * g[k];
                    *reinterpret_cast<uint4*>(p.gout_b + (int64_t)row * p.D + c * 8) = pack8t(a);
                }
            }
        }
    }
}

// ---- RMSNorm(64)-with-weight backward, in place on dy (the gradient w.r.t. the normalised, weighted heads).
// y = bf16(x * rs) * w ; xhat = y / w ; g = dy * w ; dx = rs * (g - xhat * mean(g * xhat))
__global__ __launch_bounds__(256) void rmsnorm_heads_bwd_kernel(bf16_t* __restrict__ dy, int64_t lddy, const bf16_t* __restrict__ y,
                                                                int64_t ldy, const float* __restrict__ rs, int M, int col0,
                                                                int nheads, const bf16_t* __restrict__ w, int heads_per_weight,
                                                                int seg_rows, int64_t seg_stride, int64_t seg_off) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    int64_t row = m;
    if (seg_rows > 0) row = (int64_t)(m / seg_rows) * seg_stride + seg_off + (m % seg_rows);
    const int sub = lane & 7;
    for (int h0 = 0; h0 < nheads; h0 += 8) {
        const int hh = h0 + (lane >> 3);
        if (hh >= nheads) break;
        float d[8], yy[8], ww[8];
        bf16_t* dp = dy + row * lddy + col0 + hh * 64 + sub * 8;
        unpack8t(*reinterpret_cast<const uint4*>(dp), d);
        unpack8t(*reinterpret_cast<const uint4*>(y + row * ldy + col0 + hh * 64 + sub * 8), yy);
        unpack8t(*reinterpret_cast<const uint4*>(w + (hh / heads_per_weight) * 64 + sub * 8), ww);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            d[k] *= ww[k];            // g
            yy[k] = yy[k] / ww[k];    // xhat
            dot += d[k] * yy[k];
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        dot += __shfl_xor(dot, 4, 64);
        const float r = rs[row * nheads + hh], mdot = dot * (1.0f / 64.0f);
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = r * (d[k] - yy[k] * mdot);
        *reinterpret_cast<uint4*>(dp) = pack8t(d);
    }
}

// ---- backward of qk_norm_rope_kernel (norm.hip), in place on dy = the gradient w.r.t. the rotated, normalised, weighted q | k heads
// of a packed joint QKV buffer (the Qwen-Image MMDiT in the G-step).  y = that forward's saved output.  Per (token, head):
//   un-rotate both (the rotation is orthogonal: its transpose is the rotation by -angle):  g' = R^T dy,  y_n = R^T y
//   then the weighted RMSNorm backward of rmsnorm_heads_bwd_kernel:  xhat = y_n / w,  g = g' w,  dx = rs (g - xhat mean(g xhat)).
// One wave per token row, HD / 8 lanes per head, the token's (cos, sin) row loaded once for all heads.
template <int HD>
__global__ __launch_bounds__(256) void qk_norm_rope_bwd_kernel(bf16_t* __restrict__ dy, int64_t lddy, const bf16_t* __restrict__ y,
                                                               int64_t ldy, const float* __restrict__ rs, int rows, int S, int n_first,
                                                               int col0, int nheads, const bf16_t* __restrict__ w_first,
                                                               const bf16_t* __restrict__ w_rest, int heads_per_weight,
                                                               const float* __restrict__ rope) {
    constexpr int LPH = HD / 8, HPP = 64 / LPH;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int s = row % S;
    const bf16_t* w = s < n_first ? w_first : w_rest;
    const int sub = lane % LPH;
    float cs[8] = {1.f, 0.f, 1.f, 0.f, 1.f, 0.f, 1.f, 0.f};
    if (rope) {
        const float4 a = *reinterpret_cast<const float4*>(rope + (int64_t)s * HD + sub * 8);
        const float4 b = *reinterpret_cast<const float4*>(rope + (int64_t)s * HD + sub * 8 + 4);
        cs[0] = a.x; cs[1] = a.y; cs[2] = a.z; cs[3] = a.w; cs[4] = b.x; cs[5] = b.y; cs[6] = b.z; cs[7] = b.w;
    }
    for (int h0 = 0; h0 < nheads; h0 += HPP) {
        const int hh = h0 + lane / LPH;
        if (hh >= nheads) break;
        float d[8], yy[8], ww[8];
        bf16_t* dp = dy + (int64_t)row * lddy + col0 + hh * HD + sub * 8;
        unpack8t(*reinterpret_cast<const uint4*>(dp), d);
        unpack8t(*reinterpret_cast<const uint4*>(y + (int64_t)row * ldy + col0 + hh * HD + sub * 8), yy);
        unpack8t(*reinterpret_cast<const uint4*>(w + (hh / heads_per_weight) * HD + sub * 8), ww);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const float co = cs[k], si = cs[k + 1];
            const float ga = d[k] * co + d[k + 1] * si, gb = -d[k] * si + d[k + 1] * co;
            const float ya = yy[k] * co + yy[k + 1] * si, yb = -yy[k] * si + yy[k + 1] * co;
            d[k] = ga * ww[k]; d[k + 1] = gb * ww[k + 1];               // g
            yy[k] = ya / ww[k]; yy[k + 1] = yb / ww[k + 1];             // xhat
            dot += d[k] * yy[k] + d[k + 1] * yy[k + 1];
        }
        dot = group8_sum(dot);
        if (LPH == 16) dot += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dot), 0x140, 0xf, 0xf, true));
        const float r = rs[(int64_t)row * nheads + hh], mdot = dot * (1.0f / HD);
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = r * (d[k] - yy[k] * mdot);
        *reinterpret_cast<uint4*>(dp) = pack8t(d);
    }
}

// ---- y[m, :] = gate[(m / rows_per_batch) * gate_stride + :] * x[m, :]
__global__ __launch_bounds__(256) void gate_mul_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gate,
                                                       bf16_t* __restrict__ y, int64_t total8, int D8, int rows_per_batch,
                                                       int64_t gate_stride) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % D8;
        const int64_t m = i / D8;
        float a[8], g[8];
        unpack8t(*reinterpret_cast<const uint4*>(x + i * 8), a);
        unpack8t(*reinterpret_cast<const uint4*>(gate + (m / rows_per_batch) * gate_stride + c * 8), g);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] *= g[k];
        *reinterpret_cast<uint4*>(y + i * 8) = pack8t(a);
    }
}

// ---- sum of squares of an f32 vector -> out[0] += ...; fixed summation order (the clipping factor of every optimizer step
// hangs on it: with one atomicAdd per block the factor, and with it every LoRA weight, moved in the last bit from run to run).
// Stage 1: SUMSQ_BLOCKS blocks, each a fixed grid-stride slice -> partial[block]; stage 2: one block adds the partials.
constexpr int SUMSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += g[i] * g[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) out[0] += s;
}

// ---- AdamW (torch semantics: decoupled weight decay, bias correction) with gradient clipping folded in:
// g *= min(1, max_norm / (sqrt(sumsq) + 1e-6)) as torch.nn.utils.clip_grad_norm_; p32 master, p16 bf16 copy.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p32, bf16_t* __restrict__ p16, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    const float* __restrict__ sumsq, float max_norm, float grad_scale) {
    float clip = 1.0f;
    if (sumsq && max_norm > 0.f) {
        const float norm = sqrtf(sumsq[0]) * grad_scale;
        clip = fminf(1.0f, max_norm / (norm + 1e-6f));
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale * clip;
        float w = p32[i];
        w *= 1.0f - lr * wd;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        w -= (lr / bc1) * (mi / denom);
        p32[i] = w;
        if (p16) p16[i] = f2bf(w);
        g[i] = 0.f;   // optimizer.zero_grad()
    }
}

// ---- EMA: e += (1 - decay) * (p - e)   (adv_grpo/ema.py:45-46)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ e, const float* __restrict__ p, int64_t n,
                                                  float one_minus_decay) {
    // the reference rounds three times (sub, mul, add_): no fma, or the 40-step golden sequence drifts by an ulp
#pragma clang fp contract(off)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = p[i] - e[i];
        const float s = one_minus_decay * d;
        e[i] = e[i] + s;
    }
}

static int grid_for(int64_t n, int per) {
    int64_t b = (n + per - 1) / per;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace advgrpo

using namespace advgrpo;

extern "C" int advgrpo_transpose_bf16(const void* in, void* out, int R, int C, int64_t ldi, int64_t ldo, int Rpad,
                                      int seg_rows, int64_t seg_stride, int64_t seg_off, void* stream) {
    ADVGRPO_CHECK(in && out && R > 0 && C > 0, "transpose: bad argument");
    if (Rpad < R) Rpad = R;
    ADVGRPO_CHECK(Rpad <= ldo, "transpose: padded row count exceeds the output pitch");
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (Rpad + 63) / 64), dim3(256), 0, as_stream(stream),
                       (const bf16_t*)in, (bf16_t*)out, R, C, ldi, ldo, Rpad, seg_rows, seg_stride, seg_off);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

static int ln_mod_bwd_launch(const LnBwdParams& p, void* stream) {
    if (p.D <= 2048) hipLaunchKernelGGL(layernorm_mod_bwd_kernel<4>, dim3((p.M + 3) / 4), dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(layernorm_mod_bwd_kernel<8>, dim3((p.M + 3) / 4), dim3(256), 0, as_stream(stream), p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_layernorm_mod_bwd(const void* x, int64_t ldx, const void* dy0, const void* dy1, int64_t lddy,
                                         const void* scale0, const void* scale1, int64_t mod_stride, int rows_per_batch,
                                         const void* dres, void* dx, int64_t lddx, int M, int D, float eps, void* stream) {
    ADVGRPO_CHECK(x && dy0 && dx, "layernorm_mod_bwd: null pointer");
    ADVGRPO_CHECK(M > 0 && D % 8 == 0 && D <= 4096, "layernorm_mod_bwd: need D %% 8 == 0, D <= 4096 (D=%d)", D);
    ADVGRPO_CHECK(!dy1 || scale1, "layernorm_mod_bwd: dy1 needs scale1");
    LnBwdParams p{(const bf16_t*)x, ldx, (const bf16_t*)dy0, (const bf16_t*)dy1, lddy, (const bf16_t*)scale0,
                  (const bf16_t*)scale1, mod_stride, rows_per_batch, (const bf16_t*)dres, (bf16_t*)dx, lddx, M, D, eps,
                  nullptr, nullptr, nullptr, nullptr, 0};
    return ln_mod_bwd_launch(p, stream);
}

extern "C" int advgrpo_layernorm_mod_bwd_gated(const void* x, int64_t ldx, const void* dy0, const void* dy1, int64_t lddy,
                                               const void* scale0, const void* scale1, int64_t mod_stride, int rows_per_batch,
                                               const void* dres, void* dx, int64_t lddx, int M, int D, float eps,
                                               const void* gate_a, void* gout_a, const void* gate_b, void* gout_b,
                                               int64_t gate_stride, void* stream) {
    ADVGRPO_CHECK(x && dy0 && dx, "layernorm_mod_bwd_gated: null pointer");
    ADVGRPO_CHECK(M > 0 && D % 8 == 0 && D <= 4096, "layernorm_mod_bwd_gated: need D %% 8 == 0, D <= 4096 (D=%d)", D);
    ADVGRPO_CHECK(!dy1 || scale1, "layernorm_mod_bwd_gated: dy1 needs scale1");
    ADVGRPO_CHECK(gate_a && gout_a && rows_per_batch > 0, "layernorm_mod_bwd_gated: the first gate, its output and rows_per_batch are required");
    ADVGRPO_CHECK(!gate_b == !gout_b, "layernorm_mod_bwd_gated: the second gate and its output come together");
    LnBwdParams p{(const bf16_t*)x, ldx, (const bf16_t*)dy0, (const bf16_t*)dy1, lddy, (const bf16_t*)scale0,
                  (const bf16_t*)scale1, mod_stride, rows_per_batch, (const bf16_t*)dres, (bf16_t*)dx, lddx, M, D, eps,
                  (const bf16_t*)gate_a, (bf16_t*)gout_a, (const bf16_t*)gate_b, (bf16_t*)gout_b, gate_stride};
    return ln_mod_bwd_launch(p, stream);
}

extern "C" int advgrpo_rmsnorm_heads_bwd(void* dy, int64_t lddy, const void* y, int64_t ldy, const float* rs, int M,
                                         int col0, int nheads, const void* weight, int heads_per_weight, int seg_rows,
                                         int64_t seg_stride, int64_t seg_off, void* stream) {
    ADVGRPO_CHECK(dy && y && rs && weight && M > 0 && nheads > 0, "rmsnorm_heads_bwd: bad argument");
    hipLaunchKernelGGL(rmsnorm_heads_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), (bf16_t*)dy, lddy,
                       (const bf16_t*)y, ldy, rs, M, col0, nheads, (const bf16_t*)weight, heads_per_weight, seg_rows,
                       seg_stride, seg_off);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_qk_norm_rope_bwd(void* dy, int64_t lddy, const void* y, int64_t ldy, const float* rs, int rows, int S,
                                        int n_first, int col0, int nheads, int head_dim, const void* w_first, const void* w_rest,
                                        int heads_per_weight, const float* rope, void* stream) {
    ADVGRPO_CHECK(dy && y && rs && w_first && w_rest && rows > 0 && S > 0 && nheads > 0 && heads_per_weight > 0, "qk_norm_rope_bwd: bad argument");
    ADVGRPO_CHECK(head_dim == 64 || head_dim == 128, "qk_norm_rope_bwd: head_dim %d not supported (64, 128)", head_dim);
    ADVGRPO_CHECK(lddy % 8 == 0 && ldy % 8 == 0 && col0 % 8 == 0, "qk_norm_rope_bwd: pitch/offset must be multiples of 8");
    const dim3 grid((rows + 3) / 4);
    if (head_dim == 128)
        hipLaunchKernelGGL(qk_norm_rope_bwd_kernel<128>, grid, dim3(256), 0, as_stream(stream), (bf16_t*)dy, lddy, (const bf16_t*)y, ldy,
                           rs, rows, S, n_first, col0, nheads, (const bf16_t*)w_first, (const bf16_t*)w_rest, heads_per_weight, rope);
    else
        hipLaunchKernelGGL(qk_norm_rope_bwd_kernel<64>, grid, dim3(256), 0, as_stream(stream), (bf16_t*)dy, lddy, (const bf16_t*)y, ldy,
                           rs, rows, S, n_first, col0, nheads, (const bf16_t*)w_first, (const bf16_t*)w_rest, heads_per_weight, rope);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_gate_mul(const void* x, const void* gate, void* y, int M, int D, int rows_per_batch,
                                int64_t gate_stride, void* stream) {
    ADVGRPO_CHECK(x && gate && y && M > 0 && D % 8 == 0 && rows_per_batch > 0, "gate_mul: bad argument");
    const int64_t total8 = (int64_t)M * (D / 8);
    hipLaunchKernelGGL(gate_mul_kernel, dim3(grid_for(total8, 256)), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                       (const bf16_t*)gate, (bf16_t*)y, total8, D / 8, rows_per_batch, gate_stride);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t advgrpo_sumsq_workspace_bytes(void) { return SUMSQ_BLOCKS * sizeof(float); }

extern "C" int advgrpo_sumsq_f32(const float* g, int64_t n, float* out, float* workspace, void* stream) {
    ADVGRPO_CHECK(g && out && workspace && n > 0, "sumsq: bad argument");
    int64_t nb = (n + 1023) / 1024;
    if (nb > SUMSQ_BLOCKS) nb = SUMSQ_BLOCKS;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), g, n, workspace);
    ADVGRPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), workspace, (int)nb, out);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_adamw_step(float* param_f32, void* param_bf16, float* grad, float* exp_avg, float* exp_avg_sq,
                                  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                  const float* grad_sumsq, float max_grad_norm, float grad_scale, void* stream) {
    ADVGRPO_CHECK(param_f32 && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adamw: bad argument");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), param_f32, (bf16_t*)param_bf16,
                       grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_sumsq,
                       max_grad_norm, grad_scale);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_ema_step(float* ema, const float* param, int64_t n, float one_minus_decay, void* stream) {
    ADVGRPO_CHECK(ema && param && n > 0, "ema: bad argument");
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), ema, param, n, one_minus_decay);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ DINO discriminator head (D-step)
// train_dino, scripts/train_sd3_fast_dino_patch.py:156-232, on backbone features: rows are [cls, n sampled
// patches] per image, real images first then fake.  head = Linear(D,Hd) -> GELU(erf) -> Linear(Hd,1).
namespace advgrpo {

// rows[b*(1+n) + 0] = feats[b,0,:], rows[b*(1+n)+1+j] = feats[b, 1+idx[b,j], :]   (no normalisation)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ feats, const int64_t* __restrict__ idx,
                                                          bf16_t* __restrict__ out, int B, int T, int D, int n) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * (1 + n)) return;
    const int b = row / (1 + n), j = row % (1 + n);
    const int tok = j == 0 ? 0 : 1 + (int)idx[(int64_t)b * n + j - 1];
    const bf16_t* src = feats + ((int64_t)b * T + tok) * D;
    for (int d = lane * 8; d < D; d += 512)
        *reinterpret_cast<uint4*>(out + (int64_t)row * D + d) = *reinterpret_cast<const uint4*>(src + d);
}

// per row: logit = h . w2 + b2 ; hinge loss / accuracy / d loss / d logit.  One wave per row.
// stats[0] += loss, stats[1] += correct real cls, stats[2] += correct fake cls, stats[3] += sum(dl) (= d b2)
__global__ __launch_bounds__(256) void dino_head_loss_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ w2,
                                                             const bf16_t* __restrict__ b2, int R, int Hd, int n, int Breal,
                                                             int Btot, float patch_w, float* __restrict__ logits,
                                                             float* __restrict__ dl, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float acc = 0.f;
    for (int d = lane; d < Hd; d += 64) acc += bf2f(h[(int64_t)row * Hd + d]) * bf2f(w2[d]);
    acc = wave_sum(acc);
    if (lane != 0) return;
    const float logit = round_bf16(acc + bf2f(b2[0]));
    const int b = row / (1 + n), j = row % (1 + n);
    const bool real = b < Breal, cls = j == 0;
    const float y = real ? 1.f : -1.f;
    const int nb = real ? Breal : (Btot - Breal);
    // d_loss = 0.5*(mean_real relu(1-l) + mean_fake relu(1+l)) + patch_w * (same over patches)
    const float wgt = cls ? 0.5f / (float)nb : patch_w * 0.5f / ((float)nb * (float)n);
    const float margin = 1.0f - y * logit;
    const float g = margin > 0.f ? -y * wgt : 0.f;
    logits[row] = logit;
    dl[row] = g;
    atomicAdd(&stats[0], margin > 0.f ? margin * wgt : 0.f);
    if (cls) atomicAdd(&stats[real ? 1 : 2], (y * logit > 0.f) ? 1.f : 0.f);
    atomicAdd(&stats[3], g);
}

// dpre[r,c] = dl[r] * w2[c] * gelu_erf'(pre[r,c]) (bf16);  g_w2[c] += dl[r] * h[r,c];  g_b1[c] += dpre[r,c]
__global__ __launch_bounds__(256) void dino_head_dpre_kernel(const bf16_t* __restrict__ pre, const bf16_t* __restrict__ h,
                                                             const bf16_t* __restrict__ w2, const float* __restrict__ dl,
                                                             bf16_t* __restrict__ dpre, float* __restrict__ g_w2,
                                                             float* __restrict__ g_b1, int R, int Hd) {
    // block = 256 columns x 32 rows
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Hd) return;
    const int r0 = blockIdx.y * 32, r1 = min(R, r0 + 32);
    const float w = bf2f(w2[c]);
    float sw = 0.f, sb = 0.f;
    for (int r = r0; r < r1; ++r) {
        const float u = bf2f(pre[(int64_t)r * Hd + c]);
        const float d = dl[r];
        const float dg = 0.5f * (1.0f + erff(u * 0.7071067811865476f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
        const float v = d * w * dg;
        dpre[(int64_t)r * Hd + c] = f2bf(v);
        sb += v;
        sw += d * bf2f(h[(int64_t)r * Hd + c]);
    }
    atomicAdd(&g_w2[c], sw);
    atomicAdd(&g_b1[c], sb);
}

}  // namespace advgrpo

extern "C" int advgrpo_gather_rows(const void* feats, const int64_t* idx, void* out, int B, int T, int D, int n,
                                   void* stream) {
    ADVGRPO_CHECK(feats && out && (n == 0 || idx) && B > 0 && D % 8 == 0, "gather_rows: bad argument");
    const int rows = B * (1 + n);
    hipLaunchKernelGGL(advgrpo::gather_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, advgrpo::as_stream(stream),
                       (const advgrpo::bf16_t*)feats, idx, (advgrpo::bf16_t*)out, B, T, D, n);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_dino_head_loss(const void* hidden, const void* w2, const void* b2, int R, int Hd, int n, int B_real,
                                      int B_total, float patch_loss_weight, float* logits, float* dlogits, float* stats4,
                                      void* stream) {
    ADVGRPO_CHECK(hidden && w2 && b2 && logits && dlogits && stats4 && R == B_total * (1 + n) && B_real > 0 &&
                      B_total > B_real, "dino_head_loss: bad argument");
    hipStream_t s = advgrpo::as_stream(stream);
    if (hipMemsetAsync(stats4, 0, 4 * sizeof(float), s) != hipSuccess) { advgrpo::set_error("memset failed"); return -2; }
    hipLaunchKernelGGL(advgrpo::dino_head_loss_kernel, dim3((R + 3) / 4), dim3(256), 0, s, (const advgrpo::bf16_t*)hidden,
                       (const advgrpo::bf16_t*)w2, (const advgrpo::bf16_t*)b2, R, Hd, n, B_real, B_total, patch_loss_weight,
                       logits, dlogits, stats4);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_dino_head_dpre(const void* pre, const void* hidden, const void* w2, const float* dlogits, void* dpre,
                                      float* grad_w2, float* grad_b1, int R, int Hd, void* stream) {
    ADVGRPO_CHECK(pre && hidden && w2 && dlogits && dpre && grad_w2 && grad_b1 && R > 0 && Hd > 0, "dino_head_dpre: bad argument");
    hipLaunchKernelGGL(advgrpo::dino_head_dpre_kernel, dim3((Hd + 255) / 256, (R + 31) / 32), dim3(256), 0,
                       advgrpo::as_stream(stream), (const advgrpo::bf16_t*)pre, (const advgrpo::bf16_t*)hidden,
                       (const advgrpo::bf16_t*)w2, dlogits, (advgrpo::bf16_t*)dpre, grad_w2, grad_b1, R, Hd);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
