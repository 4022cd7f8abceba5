// sde_step.hip -- fused CFG combine + Flow-CPS SDE step + per-sample Gaussian log-prob (gfx950).
//
// Replaces sd3_pipeline_with_logprob_fast.py:640-655 and sd3_sde_with_logprob.py:77-139 of the
// reference (six torch elementwise kernels, a randn, a .mean and two host syncs per step) with
// one HBM pass: 8 B (bf16 v_u, v_t, x in; bf16 x' out) to 12 B per latent element.  The kernel
// is launch-latency bound at 512^2 (0.5 MB per sample), so it is shaped for few, wide
// wavefronts: 16 B per lane per load, one block-level reduction, a tiny finalize launch.
//
// Built with -ffp-contract=off: every reference torch op is one f32 rounding here too, so
// mean / next are bit-exact with the torch-CPU oracle; log_prob differs by summation order only.
#include "common.hpp"

namespace advgrpo {

constexpr int SDE_THREADS = 256;
constexpr int SDE_VEC = 8;                       // elements per lane per iteration
constexpr int SDE_CHUNK = SDE_THREADS * SDE_VEC;  // 2048 elements per block iteration
constexpr int SDE_MAX_BLOCKS = 64;               // per sample

__device__ inline void load8(const void* p, int dt, int64_t i, float o[8]) {
    if (dt == ADVGRPO_BF16) {
        const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p) + i);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[2 * k] = bf2f((bf16_t)(w[k] & 0xffffu));
            o[2 * k + 1] = bf2f((bf16_t)(w[k] >> 16));
        }
    } else {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
        const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
}
__device__ inline void store8(void* p, int dt, int64_t i, const float o[8]) {
    if (dt == ADVGRPO_BF16) {
        uint4 r;
        r.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
        r.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
        r.z = (uint32_t)f2bf(o[4]) | ((uint32_t)f2bf(o[5]) << 16);
        r.w = (uint32_t)f2bf(o[6]) | ((uint32_t)f2bf(o[7]) << 16);
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p) + i) = r;
    } else {
        float* f = reinterpret_cast<float*>(p) + i;
        *reinterpret_cast<float4*>(f) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(f + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
}

struct SdeCoef {  // per-sample scalars, each one f32 rounding as in the reference
    float sigma, one_m_sigma, one_m_sigma_prev, sq, std;
};
__device__ inline SdeCoef sde_coef(const float* sigma, const float* sigma_prev, int stride, int b,
                                   float sin_coeff) {
    SdeCoef c;
    const float s = sigma[(int64_t)b * stride], sp = sigma_prev[(int64_t)b * stride];
    c.sigma = s;
    c.std = sp * sin_coeff;               // sd3_sde_with_logprob.py:118
    c.one_m_sigma = 1.0f - s;             // :120
    c.one_m_sigma_prev = 1.0f - sp;       // :121
    c.sq = sqrtf(sp * sp - c.std * c.std);  // :121
    return c;
}
// v after CFG (:640-642 of the pipeline); bf16 inputs => the three torch bf16 ops, each rounded
__device__ inline float cfg_v(float vu, float vt, bool has_cfg, bool bf, float g) {
    if (!has_cfg) return vu;
    float d = vt - vu;
    if (bf) d = round_bf16(d);
    float m = g * d;
    if (bf) m = round_bf16(m);
    float r = vu + m;
    if (bf) r = round_bf16(r);
    return r;
}

__global__ __launch_bounds__(SDE_THREADS) void sde_step_kernel(
    const void* __restrict__ v_u, const void* __restrict__ v_t, int v_dt, float guidance,
    const void* __restrict__ x, int x_dt, const float* __restrict__ sigma,
    const float* __restrict__ sigma_prev, int sigma_stride, float sin_coeff, int mode,
    const float* __restrict__ eps, uint64_t seed, uint64_t offset,
    const void* __restrict__ prev, int prev_dt, float* __restrict__ out_next,
    void* __restrict__ out_cast, int cast_dt, float* __restrict__ out_mean,
    float* __restrict__ partial, float* __restrict__ out_std, int64_t n) {
    __shared__ float red[SDE_THREADS / 64];
    const int b = blockIdx.y;
    const SdeCoef c = sde_coef(sigma, sigma_prev, sigma_stride, b, sin_coeff);
    const bool has_cfg = v_t != nullptr, bf = v_dt == ADVGRPO_BF16;
    const int64_t base = (int64_t)b * n;
    const Philox ph(seed);
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * SDE_CHUNK + threadIdx.x * SDE_VEC; i < n;
         i += (int64_t)gridDim.x * SDE_CHUNK) {
        float vu[8], vt[8], xs[8], e[8], mean[8], nxt[8];
        load8(v_u, v_dt, base + i, vu);
        if (has_cfg) load8(v_t, v_dt, base + i, vt);
        load8(x, x_dt, base + i, xs);
        if (mode == ADVGRPO_SDE_EPS) {
            load8(eps, ADVGRPO_F32, base + i, e);
        } else if (mode == ADVGRPO_SDE_PHILOX) {
            philox_normal4(ph, offset + (uint64_t)((base + i) >> 2), e);
            philox_normal4(ph, offset + (uint64_t)((base + i) >> 2) + 1, e + 4);
        } else {
            load8(prev, prev_dt, base + i, e);  // e holds prev_sample
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = cfg_v(vu[k], vt[k], has_cfg, bf, guidance);
            const float x0 = xs[k] - c.sigma * v;              // :119
            const float x1 = xs[k] + v * c.one_m_sigma;         // :120
            mean[k] = x0 * c.one_m_sigma_prev + x1 * c.sq;     // :121
            nxt[k] = (mode == ADVGRPO_SDE_REPLAY) ? e[k] : mean[k] + c.std * e[k];  // :131
            const float d = nxt[k] - mean[k];
            acc += d * d;                                       // :134 (negated at finalize)
        }
        if (out_next && mode != ADVGRPO_SDE_REPLAY) store8(out_next, ADVGRPO_F32, base + i, nxt);
        if (out_cast && mode != ADVGRPO_SDE_REPLAY) store8(out_cast, cast_dt, base + i, nxt);
        if (out_mean) store8(out_mean, ADVGRPO_F32, base + i, mean);
    }
    const float tot = block_sum<SDE_THREADS / 64>(acc, red);
    if (threadIdx.x == 0) {
        partial[b * gridDim.x + blockIdx.x] = tot;
        if (blockIdx.x == 0 && out_std) out_std[b] = c.std;
    }
}

// log_prob[b] = -(sum of block partials)/n ; partials summed in fixed order in f64
__global__ void sde_finalize_kernel(const float* __restrict__ partial, int nblk, int64_t n,
                                    float* __restrict__ out_log_prob, int B, float sign = -1.0f) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += (double)partial[b * nblk + i];
    out_log_prob[b] = (float)((double)sign * (s / (double)n));
}

__global__ __launch_bounds__(SDE_THREADS) void sde_step_bwd_kernel(
    const void* __restrict__ v_u, const void* __restrict__ v_t, int v_dt, float guidance,
    const void* __restrict__ x, int x_dt, const float* __restrict__ sigma,
    const float* __restrict__ sigma_prev, int sigma_stride, float sin_coeff,
    const void* __restrict__ prev, int prev_dt, const float* __restrict__ grad_lp,
    void* __restrict__ g_u, void* __restrict__ g_t, int64_t n,
    const float* __restrict__ mean_ref, float kl_weight, int B, float* __restrict__ kl_partial) {
    __shared__ float red[SDE_THREADS / 64];
    const int b = blockIdx.y;
    const SdeCoef c = sde_coef(sigma, sigma_prev, sigma_stride, b, sin_coeff);
    const bool has_cfg = v_t != nullptr, bf = v_dt == ADVGRPO_BF16;
    const int64_t base = (int64_t)b * n;
    // d mean / d v, and d log_prob / d mean = 2 (prev - mean) / n
    const float dmu_dv = c.one_m_sigma * c.sq - c.sigma * c.one_m_sigma_prev;
    const float scale = grad_lp[b] * 2.0f / (float)n * dmu_dv;
    // KL term (TP:1105-1108,1126-1130): loss += beta * mean_b mean_elem (mean - mean_ref)^2, mean_ref = the same step's mean
    // under the adapter-free transformer; d/d mean = beta * 2 (mean - mean_ref) / (n B), then the same d mean / d v
    const float kl_scale = mean_ref ? kl_weight * 2.0f / ((float)n * (float)B) * dmu_dv : 0.f;
    float kl_acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * SDE_CHUNK + threadIdx.x * SDE_VEC; i < n;
         i += (int64_t)gridDim.x * SDE_CHUNK) {
        float vu[8], vt[8], xs[8], pv[8], gu[8], gt[8], mr[8];
        if (mean_ref) load8(mean_ref, ADVGRPO_F32, base + i, mr);
        load8(v_u, v_dt, base + i, vu);
        if (has_cfg) load8(v_t, v_dt, base + i, vt);
        load8(x, x_dt, base + i, xs);
        load8(prev, prev_dt, base + i, pv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = cfg_v(vu[k], vt[k], has_cfg, bf, guidance);
            const float x0 = xs[k] - c.sigma * v;
            const float x1 = xs[k] + v * c.one_m_sigma;
            const float mean = x0 * c.one_m_sigma_prev + x1 * c.sq;
            float gv = scale * (pv[k] - mean);
            if (mean_ref) {
                const float d = mean - mr[k];
                gv += kl_scale * d;
                kl_acc += d * d;
            }
            if (bf) gv = round_bf16(gv);  // autograd hands the (summed) f32 grad back through .float()
            gu[k] = has_cfg ? (1.0f - guidance) * gv : gv;
            gt[k] = guidance * gv;
        }
        store8(g_u, v_dt, base + i, gu);
        if (has_cfg) store8(g_t, v_dt, base + i, gt);
    }
    if (mean_ref) {      // per-sample KL value: block partials, summed in fixed order by sde_finalize_kernel
        const float tot = block_sum<SDE_THREADS / 64>(kl_acc, red);
        if (threadIdx.x == 0) kl_partial[b * gridDim.x + blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(256) void randn_kernel(void* __restrict__ out, int dt, int64_t n,
                                                    uint64_t seed, uint64_t offset) {
    const Philox ph(seed);
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g * 4 < n;
         g += (int64_t)gridDim.x * blockDim.x) {
        float z[4];
        philox_normal4(ph, offset + (uint64_t)g, z);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = g * 4 + k;
            if (i < n) {
                if (dt == ADVGRPO_BF16) reinterpret_cast<bf16_t*>(out)[i] = f2bf(z[k]);
                else reinterpret_cast<float*>(out)[i] = z[k];
            }
        }
    }
}

static int sde_blocks(int64_t n) {
    int64_t nb = (n + SDE_CHUNK - 1) / SDE_CHUNK;
    return (int)(nb < 1 ? 1 : (nb > SDE_MAX_BLOCKS ? SDE_MAX_BLOCKS : nb));
}

}  // namespace advgrpo

using namespace advgrpo;

extern "C" int64_t advgrpo_sde_step_workspace_bytes(int B, int64_t n) {
    return (int64_t)B * sde_blocks(n) * sizeof(float);
}

static int check_common(int v_dtype, int x_dtype, int B, int64_t n) {
    ADVGRPO_CHECK(v_dtype == ADVGRPO_F32 || v_dtype == ADVGRPO_BF16, "sde_step: bad v_dtype %d", v_dtype);
    ADVGRPO_CHECK(x_dtype == ADVGRPO_F32 || x_dtype == ADVGRPO_BF16, "sde_step: bad x_dtype %d", x_dtype);
    ADVGRPO_CHECK(B > 0 && n > 0 && n % SDE_VEC == 0, "sde_step: need B>0 and n %% 8 == 0 (B=%d n=%lld)", B,
                  (long long)n);
    return 0;
}

extern "C" int advgrpo_sde_step(const void* v_uncond, const void* v_text, int v_dtype, float guidance_scale,
                                const void* x, int x_dtype, const float* sigma, const float* sigma_prev,
                                int sigma_stride, float sin_coeff, int mode, const float* eps, uint64_t seed,
                                uint64_t offset, const void* prev_sample, int prev_dtype, float* out_next_f32,
                                void* out_next_cast, int out_cast_dtype, float* out_mean, float* out_log_prob,
                                float* out_std, void* workspace, int B, int64_t n, void* stream) {
    if (check_common(v_dtype, x_dtype, B, n)) return -1;
    ADVGRPO_CHECK(v_uncond && x && sigma && sigma_prev && out_log_prob && workspace, "sde_step: null argument");
    ADVGRPO_CHECK(mode >= 0 && mode <= 2, "sde_step: bad mode %d", mode);
    ADVGRPO_CHECK(mode != ADVGRPO_SDE_EPS || eps, "sde_step: mode EPS needs eps");
    ADVGRPO_CHECK(mode != ADVGRPO_SDE_REPLAY || prev_sample, "sde_step: mode REPLAY needs prev_sample");
    const int nb = sde_blocks(n);
    hipLaunchKernelGGL(sde_step_kernel, dim3(nb, B), dim3(SDE_THREADS), 0, as_stream(stream), v_uncond, v_text,
                       v_dtype, guidance_scale, x, x_dtype, sigma, sigma_prev, sigma_stride, sin_coeff, mode, eps,
                       seed, offset, prev_sample, prev_dtype, out_next_f32, out_next_cast, out_cast_dtype, out_mean,
                       (float*)workspace, out_std, n);
    ADVGRPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(sde_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream),
                       (const float*)workspace, nb, n, out_log_prob, B);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_sde_step_bwd(const void* v_uncond, const void* v_text, int v_dtype, float guidance_scale,
                                    const void* x, int x_dtype, const float* sigma, const float* sigma_prev,
                                    int sigma_stride, float sin_coeff, const void* prev_sample, int prev_dtype,
                                    const float* grad_log_prob, void* grad_v_uncond, void* grad_v_text, int B,
                                    int64_t n, void* stream) {
    if (check_common(v_dtype, x_dtype, B, n)) return -1;
    ADVGRPO_CHECK(v_uncond && x && sigma && sigma_prev && prev_sample && grad_log_prob && grad_v_uncond,
                  "sde_step_bwd: null argument");
    ADVGRPO_CHECK(!v_text || grad_v_text, "sde_step_bwd: CFG needs grad_v_text");
    hipLaunchKernelGGL(sde_step_bwd_kernel, dim3(sde_blocks(n), B), dim3(SDE_THREADS), 0, as_stream(stream),
                       v_uncond, v_text, v_dtype, guidance_scale, x, x_dtype, sigma, sigma_prev, sigma_stride,
                       sin_coeff, prev_sample, prev_dtype, grad_log_prob, grad_v_uncond, grad_v_text, n,
                       (const float*)nullptr, 0.f, B, (float*)nullptr);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_sde_step_bwd_kl(const void* v_uncond, const void* v_text, int v_dtype, float guidance_scale,
                                       const void* x, int x_dtype, const float* sigma, const float* sigma_prev,
                                       int sigma_stride, float sin_coeff, const void* prev_sample, int prev_dtype,
                                       const float* grad_log_prob, const float* mean_ref, float kl_weight,
                                       void* grad_v_uncond, void* grad_v_text, float* kl_out, void* workspace, int B,
                                       int64_t n, void* stream) {
    if (check_common(v_dtype, x_dtype, B, n)) return -1;
    ADVGRPO_CHECK(v_uncond && x && sigma && sigma_prev && prev_sample && grad_log_prob && grad_v_uncond && mean_ref &&
                      kl_out && workspace, "sde_step_bwd_kl: null argument");
    ADVGRPO_CHECK(!v_text || grad_v_text, "sde_step_bwd_kl: CFG needs grad_v_text");
    const int nb = sde_blocks(n);
    hipLaunchKernelGGL(sde_step_bwd_kernel, dim3(nb, B), dim3(SDE_THREADS), 0, as_stream(stream), v_uncond, v_text,
                       v_dtype, guidance_scale, x, x_dtype, sigma, sigma_prev, sigma_stride, sin_coeff, prev_sample,
                       prev_dtype, grad_log_prob, grad_v_uncond, grad_v_text, n, mean_ref, kl_weight, B,
                       (float*)workspace);
    ADVGRPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(sde_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream),
                       (const float*)workspace, nb, n, kl_out, B, 1.0f);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_randn(void* out, int out_dtype, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    ADVGRPO_CHECK(out && n >= 0, "randn: bad argument");
    ADVGRPO_CHECK(out_dtype == ADVGRPO_F32 || out_dtype == ADVGRPO_BF16, "randn: bad dtype %d", out_dtype);
    if (n == 0) return 0;
    int64_t groups = (n + 3) / 4;
    int blocks = (int)((groups + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(randn_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), out, out_dtype, n, seed, offset);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
