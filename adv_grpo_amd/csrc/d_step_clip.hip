// d_step_clip.hip -- kernels of the PickScore discriminator update (D-step, CLIP variant) for gfx950.
//
// train_pickscore (scripts/train_sd3_fast_pickscore.py:151-183) + CLIPCriterion (adv_grpo/pick_score_training.py:
// 89-224) with requires_grad only on vision_model.encoder.layers[tune_layer:] (TP:1016-1020; shipped: tune_layer = -1).
// Because CLIP pools the CLS token, the last encoder layer's loss gradient lives on ONE query per image:
//   * cls_attention fwd/bwd: per (image, head) a single query against all 257 keys -- a wave-sized problem; the
//     backward fills dq (CLS row only), dk, dv (all rows) of a packed dqkv buffer.  No flash machinery needed.
//   * clip_pair_loss: L2-normalise, 2-way CE on the diagonal text->image logits (= mean softplus(s(t.f1 - t.f0))),
//     gradient w.r.t. the un-normalised image embeddings in the same launch.
//   * column sums (bias gradients) and LayerNorm affine gradients as atomic f32 reductions.
// Everything else of the D-step (Linears, weight gradients, LayerNorm dx, Adam) reuses the GEMM / row kernels.
#include "common.hpp"

namespace advgrpo {

// qkv: [Bt, S, 3*H*hd] bf16 (q | k | v).  One workgroup (64 lanes) per (image, head).
// fwd: o_cls[b, h*hd + d], probs[b, h, key]
__global__ __launch_bounds__(64) void cls_attention_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o_cls,
                                                               float* __restrict__ probs, int S, int H, int hd, float scale) {
    extern __shared__ float sm[];          // q[hd] | p[S]
    float* q = sm;
    float* p = sm + hd;
    const int b = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
    const int D = H * hd;
    const bf16_t* base = qkv + (int64_t)b * S * 3 * D;
    for (int d = lane; d < hd; d += 64) q[d] = bf2f(base[h * hd + d]) * scale;      // CLS = token 0
    __syncthreads();
    float mx = -INFINITY;
    for (int key = lane; key < S; key += 64) {
        const bf16_t* kr = base + (int64_t)key * 3 * D + D + h * hd;
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s += q[d] * bf2f(kr[d]);
        p[key] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int key = lane; key < S; key += 64) {
        const float e = __expf(p[key] - mx);
        p[key] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    __syncthreads();
    for (int key = lane; key < S; key += 64) {
        p[key] *= inv;
        probs[((int64_t)b * H + h) * S + key] = p[key];
    }
    __syncthreads();
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        for (int key = 0; key < S; ++key) acc += p[key] * bf2f(base[(int64_t)key * 3 * D + 2 * D + h * hd + d]);
        o_cls[(int64_t)b * D + h * hd + d] = f2bf(acc);
    }
}

// bwd: do_cls [Bt, D] -> dqkv [Bt, S, 3D] (this (b,h) slice fully written: dq zero except the CLS row)
__global__ __launch_bounds__(64) void cls_attention_bwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ probs,
                                                               const bf16_t* __restrict__ do_cls, bf16_t* __restrict__ dqkv,
                                                               int S, int H, int hd, float scale) {
    extern __shared__ float sm[];          // q[hd] | do[hd] | ds[S] | p[S]
    float* q = sm;
    float* dO = sm + hd;
    float* ds = sm + 2 * hd;
    float* p = ds + S;
    const int b = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
    const int D = H * hd;
    const bf16_t* base = qkv + (int64_t)b * S * 3 * D;
    bf16_t* dbase = dqkv + (int64_t)b * S * 3 * D;
    for (int d = lane; d < hd; d += 64) {
        q[d] = bf2f(base[h * hd + d]);
        dO[d] = bf2f(do_cls[(int64_t)b * D + h * hd + d]);
    }
    __syncthreads();
    float part = 0.f;
    for (int key = lane; key < S; key += 64) {
        const bf16_t* vr = base + (int64_t)key * 3 * D + 2 * D + h * hd;
        float dp = 0.f;
        for (int d = 0; d < hd; ++d) dp += dO[d] * bf2f(vr[d]);
        const float pk = probs[((int64_t)b * H + h) * S + key];
        p[key] = pk;
        ds[key] = dp;
        part += pk * dp;
    }
    const float delta = wave_sum(part);
    __syncthreads();
    for (int key = lane; key < S; key += 64) ds[key] = p[key] * (ds[key] - delta);
    __syncthreads();
    for (int d = lane; d < hd; d += 64) {
        float dq = 0.f;
        for (int key = 0; key < S; ++key) {
            const int64_t row = (int64_t)key * 3 * D;
            dq += ds[key] * bf2f(base[row + D + h * hd + d]);
            dbase[row + D + h * hd + d] = f2bf(scale * ds[key] * q[d]);          // dk
            dbase[row + 2 * D + h * hd + d] = f2bf(p[key] * dO[d]);              // dv
            if (key > 0) dbase[row + h * hd + d] = 0;                            // dq of non-CLS rows
        }
        dbase[h * hd + d] = f2bf(scale * dq);                                    // dq of the CLS row
    }
}

// e: [2B, P] image embeddings (real rows first), t: [B, P] text embeddings, both un-normalised bf16.
// loss = mean_i softplus(s * (t^_i . e^1_i - t^_i . e^0_i)); de = d loss / d e (bf16).  One wave per pair.
// With per-pair labels (CLIPCriterion.calc_loss, pick_score_training.py:172-186, in_batch_negatives = False): z = s (cos1 - cos0),
// loss_i = label_0 softplus(z) + label_1 softplus(-z) + [label_0 == label_1] log(0.5); lab0 == nullptr means (1, 0), the only pair
// the reference's caller passes (TP:170-171).
__global__ __launch_bounds__(64) void clip_pair_loss_kernel(const bf16_t* __restrict__ e, const bf16_t* __restrict__ t, int B,
                                                            int P, float s, const float* __restrict__ lab0,
                                                            const float* __restrict__ lab1, float* __restrict__ loss,
                                                            bf16_t* __restrict__ de) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const bf16_t* e0 = e + (int64_t)i * P;
    const bf16_t* e1 = e + (int64_t)(B + i) * P;
    const bf16_t* ti = t + (int64_t)i * P;
    float n0 = 0.f, n1 = 0.f, nt = 0.f, d0 = 0.f, d1 = 0.f;
    for (int k = lane; k < P; k += 64) {
        const float a = bf2f(e0[k]), c = bf2f(e1[k]), u = bf2f(ti[k]);
        n0 += a * a; n1 += c * c; nt += u * u; d0 += a * u; d1 += c * u;
    }
    n0 = sqrtf(wave_sum(n0)); n1 = sqrtf(wave_sum(n1)); nt = sqrtf(wave_sum(nt));
    d0 = wave_sum(d0) / (n0 * nt);   // cos(t, e0)
    d1 = wave_sum(d1) / (n1 * nt);
    const float z = s * (d1 - d0);
    const float sp = z > 20.f ? z : log1pf(__expf(z));
    const float sig = 1.0f / (1.0f + __expf(-z));
    float li = sp, gz = sig;
    if (lab0) {
        const float l0 = lab0[i], l1 = lab1[i];
        const float spn = -z > 20.f ? -z : log1pf(__expf(-z));          // softplus(-z): cross entropy against image 1
        li = l0 * sp + l1 * spn + (l0 == l1 ? -0.6931471805599453f : 0.f);
        gz = l0 * sig - l1 * (1.0f - sig);
    }
    if (lane == 0) atomicAdd(loss, li / (float)B);
    const float g = gz * s / (float)B;               // d loss / d cos1 ; d loss / d cos0 = -g
    for (int k = lane; k < P; k += 64) {
        const float a = bf2f(e0[k]) / n0, c = bf2f(e1[k]) / n1, u = bf2f(ti[k]) / nt;
        // d cos / d e = (t^ - e^ cos) / |e|
        de[(int64_t)i * P + k] = f2bf(-g * (u - a * d0) / n0);
        de[(int64_t)(B + i) * P + k] = f2bf(g * (u - c * d1) / n1);
    }
}

// out[c] += sum_r x[r, c]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int64_t ld, int R, int C,
                                                     float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * 64, r1 = min(R, r0 + 64);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += bf2f(x[(int64_t)r * ld + c]);
    atomicAdd(&out[c], s);
}

// LayerNorm affine gradients: gw[c] += sum_r dy[r,c] * xhat[r,c];  gb[c] += sum_r dy[r,c].  One row per wave.
__global__ __launch_bounds__(256) void ln_affine_grads_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                              const bf16_t* __restrict__ dy, int64_t lddy, int M, int D,
                                                              float eps, float* __restrict__ gw, float* __restrict__ gb) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float sum = 0.f;
    for (int d = lane; d < D; d += 64) sum += bf2f(x[(int64_t)row * ldx + d]);
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = bf2f(x[(int64_t)row * ldx + d]) - mean; sq += v * v; }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    for (int d = lane; d < D; d += 64) {
        const float g = bf2f(dy[(int64_t)row * lddy + d]);
        atomicAdd(&gw[d], g * (bf2f(x[(int64_t)row * ldx + d]) - mean) * rstd);
        atomicAdd(&gb[d], g);
    }
}

}  // namespace advgrpo

using namespace advgrpo;

extern "C" int advgrpo_cls_attention_fwd(const void* qkv, void* o_cls, float* probs, int Bt, int S, int H, int head_dim,
                                         float scale, void* stream) {
    ADVGRPO_CHECK(qkv && o_cls && probs && Bt > 0 && S > 0 && H > 0 && head_dim > 0, "cls_attention_fwd: bad argument");
    const size_t sm = (size_t)(head_dim + S) * sizeof(float);
    hipLaunchKernelGGL(cls_attention_fwd_kernel, dim3(H, Bt), dim3(64), sm, as_stream(stream), (const bf16_t*)qkv,
                       (bf16_t*)o_cls, probs, S, H, head_dim, scale);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_cls_attention_bwd(const void* qkv, const float* probs, const void* do_cls, void* dqkv, int Bt, int S,
                                         int H, int head_dim, float scale, void* stream) {
    ADVGRPO_CHECK(qkv && probs && do_cls && dqkv && Bt > 0 && S > 0, "cls_attention_bwd: bad argument");
    const size_t sm = (size_t)(2 * head_dim + 2 * S) * sizeof(float);
    hipLaunchKernelGGL(cls_attention_bwd_kernel, dim3(H, Bt), dim3(64), sm, as_stream(stream), (const bf16_t*)qkv, probs,
                       (const bf16_t*)do_cls, (bf16_t*)dqkv, S, H, head_dim, scale);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_clip_pair_loss(const void* image_embs, const void* text_embs, int B, int P, float logit_scale_exp,
                                      float* loss, void* d_image_embs, void* stream) {
    ADVGRPO_CHECK(image_embs && text_embs && loss && d_image_embs && B > 0 && P > 0, "clip_pair_loss: bad argument");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(loss, 0, sizeof(float), s) != hipSuccess) { set_error("clip_pair_loss: memset failed"); return -2; }
    hipLaunchKernelGGL(clip_pair_loss_kernel, dim3(B), dim3(64), 0, s, (const bf16_t*)image_embs, (const bf16_t*)text_embs,
                       B, P, logit_scale_exp, (const float*)nullptr, (const float*)nullptr, loss, (bf16_t*)d_image_embs);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_clip_pair_loss_labels(const void* image_embs, const void* text_embs, int B, int P, float logit_scale_exp,
                                             const float* label_0, const float* label_1, float* loss, void* d_image_embs,
                                             void* stream) {
    ADVGRPO_CHECK(image_embs && text_embs && loss && d_image_embs && B > 0 && P > 0, "clip_pair_loss_labels: bad argument");
    ADVGRPO_CHECK((label_0 == nullptr) == (label_1 == nullptr), "clip_pair_loss_labels: label_0 and label_1 come as a pair");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(loss, 0, sizeof(float), s) != hipSuccess) { set_error("clip_pair_loss_labels: memset failed"); return -2; }
    hipLaunchKernelGGL(clip_pair_loss_kernel, dim3(B), dim3(64), 0, s, (const bf16_t*)image_embs, (const bf16_t*)text_embs,
                       B, P, logit_scale_exp, label_0, label_1, loss, (bf16_t*)d_image_embs);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

// Softmax forward + backward over materialised score rows (the general tune_layer path: 257 keys, one (image, head, query) per row):
//   P = softmax(sc[r, :n_valid]),  dS = scale * P (dP - sum_j P_j dP_j);  columns >= n_valid and rows whose query index (r % n) >= n_valid
// are written as zeros (the padding the batched GEMMs around this kernel contract over).  One wave per row, n <= 512.
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ sc, const float* __restrict__ dp,
                                                               bf16_t* __restrict__ p16, bf16_t* __restrict__ ds16, int64_t rows, int n,
                                                               int n_valid, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const bool pad_row = (int)(r % n) >= n_valid;
    float s[8], d[8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < n_valid && !pad_row;
        s[i] = ok ? sc[r * n + c] : -INFINITY;
        d[i] = ok ? dp[r * n + c] : 0.f;
        mx = fmaxf(mx, s[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = pad_row ? 0.f : __expf(s[i] - mx); sum += s[i]; }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] *= inv; dot += s[i] * d[i]; }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        if (c < n) {
            p16[r * n + c] = f2bf(s[i]);
            ds16[r * n + c] = f2bf(scale * s[i] * (d[i] - dot));
        }
    }
}

extern "C" int advgrpo_softmax_bwd_rows(const float* sc, const float* dp, void* p16, void* ds16, int64_t rows, int n, int n_valid,
                                        float scale, void* stream) {
    ADVGRPO_CHECK(sc && dp && p16 && ds16 && rows > 0 && n > 0 && n <= 512 && n_valid > 0 && n_valid <= n, "softmax_bwd_rows: bad argument (n=%d)", n);
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), sc, dp, (bf16_t*)p16,
                       (bf16_t*)ds16, rows, n, n_valid, scale);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_colsum_bf16(const void* x, int64_t ld, int R, int C, float* out, void* stream) {
    ADVGRPO_CHECK(x && out && R > 0 && C > 0, "colsum: bad argument");
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256, (R + 63) / 64), dim3(256), 0, as_stream(stream),
                       (const bf16_t*)x, ld, R, C, out);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_ln_affine_grads(const void* x, int64_t ldx, const void* dy, int64_t lddy, int M, int D, float eps,
                                       float* grad_w, float* grad_b, void* stream) {
    ADVGRPO_CHECK(x && dy && grad_w && grad_b && M > 0 && D > 0, "ln_affine_grads: bad argument");
    hipLaunchKernelGGL(ln_affine_grads_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), (const bf16_t*)x, ldx,
                       (const bf16_t*)dy, lddy, M, D, eps, grad_w, grad_b);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
