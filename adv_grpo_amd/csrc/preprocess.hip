// preprocess.hip -- on-device image preprocessing and reward epilogues (gfx950).
//
// The reference moves every generated image GPU -> CPU -> PIL -> CLIPProcessor -> GPU before PickScore
// (adv_grpo/rewards.py:567-571, adv_grpo/pickscore_scorer.py:21-27) and runs F.interpolate + normalise as
// separate torch kernels before DINOv2 (adv_grpo/rewards.py:379-391).  Here both preprocessors run on the
// device and write the ViT patch-embedding im2col matrix directly (bf16 [B*P, 640], 3*14*14 = 588 real
// columns + zero pad to the GEMM's K granule), so the image is read once and no resized image is stored.
//
//   * CLIP path: bit-exact emulation of PIL's 8-bit antialiased bicubic resample (Pillow Resample.c:
//     horizontal then vertical pass, 22-bit fixed-point coefficients computed on the host in float64,
//     uint8 intermediate) after the reference's (x*255).round().clamp() quantisation.
//   * DINO path: torch's upsample_bicubic2d (A = -0.75, align_corners=False, no antialias) in f32 on
//     bf16 inputs, rounded to bf16, then (x - mean)/std in f32, rounded to bf16.
//   * reward epilogues: gather + L2-normalise the CLS / sampled patch tokens, head second layer +
//     0.7/0.3 mix (rewards.py:399-421); PickScore exp(logit_scale) * cos / 26 (pickscore_scorer.py:40-52).
#include "common.hpp"

namespace advgrpo {

constexpr int PATCH = 14, PATCH_K = 3 * PATCH * PATCH, PATCH_KPAD = 640;

__device__ inline float ld_img(const void* img, int dt, int64_t i) {
    return dt == ADVGRPO_BF16 ? bf2f(reinterpret_cast<const bf16_t*>(img)[i]) : reinterpret_cast<const float*>(img)[i];
}

// ---- CLIP: quantise + horizontal PIL pass.  img [B,3,H,W] -> tmp u8 [B,3,H,OW]
__global__ void clip_resize_h_kernel(const void* __restrict__ img, int dt, uint8_t* __restrict__ tmp, int64_t rows, int W,
                                     int OW, const int* __restrict__ bounds, const int* __restrict__ coefs, int ksize,
                                     int trunc) {
    const int64_t total = rows * OW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = i % OW;
        const int64_t row = i / OW;
        const int xmin = bounds[2 * ox], cnt = bounds[2 * ox + 1];
        int ss = 1 << 21;  // 1 << (PRECISION_BITS - 1), PRECISION_BITS = 22
        for (int k = 0; k < cnt; ++k) {
            float v = ld_img(img, dt, row * W + xmin + k) * 255.0f;
            if (dt == ADVGRPO_BF16) v = round_bf16(v);   // the product is a bf16 torch op when images are bf16
            // (x*255).round().clamp(0,255).to(uint8) (rewards.py:567)  |  (x*255).astype(uint8) (tensor_to_pil_list, TD:135-149)
            const int q = (int)fminf(fmaxf(trunc ? truncf(v) : rintf(v), 0.f), 255.f);
            ss += q * coefs[ox * ksize + k];
        }
        ss >>= 22;
        tmp[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
}

// ---- CLIP: vertical PIL pass + rescale/normalise + im2col.  tmp u8 [B,3,H,OW] -> patches bf16 [B*P, 640]
// X3: the normalised pixel stays f32 and leaves as the split-bf16 left operand [hi | hi | lo], row pitch 3 * 640 (x3.hip)
template <bool X3>
__global__ void clip_resize_v_patches_kernel(const uint8_t* __restrict__ tmp, bf16_t* __restrict__ patches, int B, int H,
                                             int OH, int OW, const int* __restrict__ bounds,
                                             const int* __restrict__ coefs, int ksize, float3 mean, float3 stdv) {
    const int gw = OW / PATCH;
    const int P = (OH / PATCH) * gw;
    const int64_t total = (int64_t)B * P * PATCH_KPAD;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = i % PATCH_KPAD;
        const int64_t prow = i / PATCH_KPAD;
        bf16_t* o3 = patches + prow * (3 * PATCH_KPAD) + col;
        if (col >= PATCH_K) {
            if (X3) { o3[0] = 0; o3[PATCH_KPAD] = 0; o3[2 * PATCH_KPAD] = 0; }
            else patches[i] = 0;
            continue;
        }
        const int p = prow % P, b = prow / P;
        const int c = col / (PATCH * PATCH), iy = (col / PATCH) % PATCH, ix = col % PATCH;
        const int oy = (p / gw) * PATCH + iy, ox = (p % gw) * PATCH + ix;
        const int ymin = bounds[2 * oy], cnt = bounds[2 * oy + 1];
        const uint8_t* src = tmp + ((int64_t)(b * 3 + c) * H) * OW + ox;
        int ss = 1 << 21;
        for (int k = 0; k < cnt; ++k) ss += (int)src[(int64_t)(ymin + k) * OW] * coefs[oy * ksize + k];
        ss >>= 22;
        const int q = ss < 0 ? 0 : (ss > 255 ? 255 : ss);
        // CLIPImageProcessor: rescale (u8 * (1/255) in f32) then (x - mean) / std in f32
        const float m = c == 0 ? mean.x : (c == 1 ? mean.y : mean.z);
        const float s = c == 0 ? stdv.x : (c == 1 ? stdv.y : stdv.z);
        const float x = (float)q * 0.00392156862745098f;
        const float v = (x - m) / s;
        if (X3) {
            const bf16_t hi = f2bf(v);
            o3[0] = hi; o3[PATCH_KPAD] = hi; o3[2 * PATCH_KPAD] = f2bf(v - bf2f(hi));
        } else {
            patches[i] = f2bf(v);
        }
    }
}

// ---- DINO: bicubic (A=-0.75) resize in f32 from (bf16-rounded) inputs -> bf16 -> normalise -> im2col
__device__ inline void cubic_coeffs(float t, float c[4]) {
    const float A = -0.75f;
    const float x1 = t + 1.0f, x2 = t, x3 = 1.0f - t, x4 = 2.0f - t;
    c[0] = ((A * x1 - 5.0f * A) * x1 + 8.0f * A) * x1 - 4.0f * A;
    c[1] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[2] = ((A + 2.0f) * x3 - (A + 3.0f)) * x3 * x3 + 1.0f;
    c[3] = ((A * x4 - 5.0f * A) * x4 + 8.0f * A) * x4 - 4.0f * A;
}
// X3 (the fp32 tower of image_similarity_score, rewards.py:147-203): no bf16 rounding anywhere, the normalised pixel leaves as
// the split-bf16 left operand [hi | hi | lo], row pitch 3 * 640
template <bool X3>
__global__ void dino_preprocess_patches_kernel(const void* __restrict__ img, int dt, bf16_t* __restrict__ patches, int B,
                                               int H, int W, int OH, int OW, float3 mean, float3 stdv) {
    const int gw = OW / PATCH;
    const int P = (OH / PATCH) * gw;
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    const int64_t total = (int64_t)B * P * PATCH_KPAD;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = i % PATCH_KPAD;
        const int64_t prow = i / PATCH_KPAD;
        bf16_t* o3 = patches + prow * (3 * PATCH_KPAD) + col;
        if (col >= PATCH_K) {
            if (X3) { o3[0] = 0; o3[PATCH_KPAD] = 0; o3[2 * PATCH_KPAD] = 0; }
            else patches[i] = 0;
            continue;
        }
        const int p = prow % P, b = prow / P;
        const int c = col / (PATCH * PATCH), iy = (col / PATCH) % PATCH, ix = col % PATCH;
        const int oy = (p / gw) * PATCH + iy, ox = (p % gw) * PATCH + ix;
        const float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        float cy[4], cx[4];
        cubic_coeffs(fy - (float)y0, cy);
        cubic_coeffs(fx - (float)x0, cx);
        const int64_t base = (int64_t)(b * 3 + c) * H * W;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(y0 - 1 + a, 0), H - 1);
            float r = 0.f;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int xx = min(max(x0 - 1 + d, 0), W - 1);
                float v = ld_img(img, dt, base + (int64_t)yy * W + xx);
                if (!X3 && dt != ADVGRPO_BF16) v = round_bf16(v);   // the reference casts images to bf16 first (TP:816)
                r += cx[d] * v;
            }
            acc += cy[a] * r;
        }
        const float rb = X3 ? acc : round_bf16(acc);
        const float m = c == 0 ? mean.x : (c == 1 ? mean.y : mean.z);
        const float s = c == 0 ? stdv.x : (c == 1 ? stdv.y : stdv.z);
        const float v = (rb - m) / s;
        if (X3) {
            const bf16_t hi = f2bf(v);
            o3[0] = hi; o3[PATCH_KPAD] = hi; o3[2 * PATCH_KPAD] = f2bf(v - bf2f(hi));
        } else {
            patches[i] = f2bf(v);
        }
    }
}

// ---- rows: out[b*(1+n) + 0] = norm(feats[b,0]); out[b*(1+n)+1+j] = norm(feats[b, 1+idx[b,j]]);  x/(||x||+eps)
__global__ __launch_bounds__(256) void gather_l2norm_rows_kernel(const bf16_t* __restrict__ feats, const int64_t* __restrict__ idx,
                                                                 bf16_t* __restrict__ out, int B, int T, int D, int n,
                                                                 float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * (1 + n)) return;
    const int b = row / (1 + n), j = row % (1 + n);
    const int tok = j == 0 ? 0 : 1 + (int)idx[(int64_t)b * n + j - 1];
    const bf16_t* src = feats + ((int64_t)b * T + tok) * D;
    float sq = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = bf2f(src[d]); sq += v * v; }
    // torch: x / (x.norm() + 1e-6) on bf16 tensors: norm accumulates in f32 and rounds to bf16, add and divide in bf16
    const float nrm = round_bf16(round_bf16(sqrtf(wave_sum(sq))) + eps);
    for (int d = lane; d < D; d += 64) out[(int64_t)row * D + d] = f2bf(bf2f(src[d]) / nrm);
}

// ---- hidden [B*(1+n), Hd] bf16 (after Linear+GELU) . w2 + b2 -> cls/patch logits; hybrid = cw*cls + (1-cw)*mean
__global__ __launch_bounds__(256) void dino_head_combine_kernel(const bf16_t* __restrict__ hidden, const bf16_t* __restrict__ w2,
                                                                const bf16_t* __restrict__ b2, int Hd, int n, float cls_w,
                                                                float* __restrict__ hybrid, float* __restrict__ cls_out,
                                                                float* __restrict__ patch_out) {
    __shared__ float logits[1 + 256];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < 1 + n; r += 4) {
        const bf16_t* h = hidden + ((int64_t)b * (1 + n) + r) * Hd;
        float acc = 0.f;
        for (int d = lane; d < Hd; d += 64) acc += bf2f(h[d]) * bf2f(w2[d]);
        acc = wave_sum(acc);
        if (lane == 0) logits[r] = round_bf16(acc + bf2f(b2[0]));   // bf16 Linear output
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int j = 0; j < n; ++j) s += logits[1 + j];
        const float mean = round_bf16(s / (float)n);
        const float c = logits[0];
        // cls_weight * cls_score + (1 - cls_weight) * patch_score_mean, each op rounded to bf16
        hybrid[b] = round_bf16(round_bf16(cls_w * c) + round_bf16((float)(1.0 - (double)cls_w) * mean));
        cls_out[b] = c;
    }
    for (int j = threadIdx.x; j < n; j += blockDim.x) patch_out[(int64_t)b * n + j] = logits[1 + j];
}

// ---- PickScore: exp(logit_scale) * <t/|t|, i/|i|> / 26 per pair
__global__ __launch_bounds__(64) void pickscore_pairs_kernel(const bf16_t* __restrict__ img, const bf16_t* __restrict__ txt,
                                                             int P, float logit_scale_exp, float* __restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float ii = 0.f, tt = 0.f, it = 0.f;
    for (int d = lane; d < P; d += 64) {
        const float a = bf2f(img[(int64_t)b * P + d]), c = bf2f(txt[(int64_t)b * P + d]);
        ii += a * a; tt += c * c; it += a * c;
    }
    ii = wave_sum(ii); tt = wave_sum(tt); it = wave_sum(it);
    if (lane == 0) out[b] = logit_scale_exp * it / (sqrtf(ii) * sqrtf(tt)) / 26.0f;
}

}  // namespace advgrpo

using namespace advgrpo;

static int clip_preprocess_impl(bool x3, const void* image, int image_dtype, void* patches, uint8_t* tmp, int B, int H, int W,
                                int OH, int OW, const int* bounds_h, const int* coefs_h, int ksize_h, const int* bounds_v,
                                const int* coefs_v, int ksize_v, const float* mean3_host, const float* std3_host,
                                int quant_trunc, void* stream) {
    ADVGRPO_CHECK(image && patches && tmp && bounds_h && coefs_h && bounds_v && coefs_v && mean3_host && std3_host,
                  "clip_preprocess: null pointer");
    ADVGRPO_CHECK(OH % PATCH == 0 && OW % PATCH == 0 && B > 0, "clip_preprocess: output must be a multiple of 14");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(clip_resize_h_kernel, dim3(2048), dim3(256), 0, s, image, image_dtype, tmp, (int64_t)B * 3 * H, W,
                       OW, bounds_h, coefs_h, ksize_h, quant_trunc);
    ADVGRPO_LAUNCH_CHECK();
    const float3 m = make_float3(mean3_host[0], mean3_host[1], mean3_host[2]);
    const float3 sd = make_float3(std3_host[0], std3_host[1], std3_host[2]);
    if (x3)
        hipLaunchKernelGGL(clip_resize_v_patches_kernel<true>, dim3(2048), dim3(256), 0, s, tmp, (bf16_t*)patches, B, H, OH, OW,
                           bounds_v, coefs_v, ksize_v, m, sd);
    else
        hipLaunchKernelGGL(clip_resize_v_patches_kernel<false>, dim3(2048), dim3(256), 0, s, tmp, (bf16_t*)patches, B, H, OH, OW,
                           bounds_v, coefs_v, ksize_v, m, sd);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_clip_preprocess_patches(const void* image, int image_dtype, void* patches, uint8_t* tmp, int B,
                                               int H, int W, int OH, int OW, const int* bounds_h, const int* coefs_h,
                                               int ksize_h, const int* bounds_v, const int* coefs_v, int ksize_v,
                                               const float* mean3_host, const float* std3_host, int quant_trunc,
                                               void* stream) {
    return clip_preprocess_impl(false, image, image_dtype, patches, tmp, B, H, W, OH, OW, bounds_h, coefs_h, ksize_h, bounds_v,
                                coefs_v, ksize_v, mean3_host, std3_host, quant_trunc, stream);
}

/* the same with the normalised pixels kept in f32 and written as the split-bf16 left operand: patches3 [B*P, 3*640] */
extern "C" int advgrpo_clip_preprocess_patches_x3(const void* image, int image_dtype, void* patches3, uint8_t* tmp, int B,
                                                  int H, int W, int OH, int OW, const int* bounds_h, const int* coefs_h,
                                                  int ksize_h, const int* bounds_v, const int* coefs_v, int ksize_v,
                                                  const float* mean3_host, const float* std3_host, int quant_trunc,
                                                  void* stream) {
    return clip_preprocess_impl(true, image, image_dtype, patches3, tmp, B, H, W, OH, OW, bounds_h, coefs_h, ksize_h, bounds_v,
                                coefs_v, ksize_v, mean3_host, std3_host, quant_trunc, stream);
}

static int dino_preprocess_impl(bool x3, const void* image, int image_dtype, void* patches, int B, int H, int W, int OH, int OW,
                                const float* mean3_host, const float* std3_host, void* stream) {
    ADVGRPO_CHECK(image && patches && mean3_host && std3_host, "dino_preprocess: null pointer");
    ADVGRPO_CHECK(OH % PATCH == 0 && OW % PATCH == 0 && B > 0, "dino_preprocess: output must be a multiple of 14");
    const float3 m = make_float3(mean3_host[0], mean3_host[1], mean3_host[2]);
    const float3 sd = make_float3(std3_host[0], std3_host[1], std3_host[2]);
    if (x3)
        hipLaunchKernelGGL(dino_preprocess_patches_kernel<true>, dim3(4096), dim3(256), 0, as_stream(stream), image, image_dtype,
                           (bf16_t*)patches, B, H, W, OH, OW, m, sd);
    else
        hipLaunchKernelGGL(dino_preprocess_patches_kernel<false>, dim3(4096), dim3(256), 0, as_stream(stream), image, image_dtype,
                           (bf16_t*)patches, B, H, W, OH, OW, m, sd);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_dino_preprocess_patches(const void* image, int image_dtype, void* patches, int B, int H, int W,
                                               int OH, int OW, const float* mean3_host, const float* std3_host,
                                               void* stream) {
    return dino_preprocess_impl(false, image, image_dtype, patches, B, H, W, OH, OW, mean3_host, std3_host, stream);
}

/* the fp32 pipeline of image_similarity_score (rewards.py:147-203): f32 bicubic, no bf16 rounding; patches3 [B*P, 3*640] */
extern "C" int advgrpo_dino_preprocess_patches_x3(const void* image, int image_dtype, void* patches3, int B, int H, int W,
                                                  int OH, int OW, const float* mean3_host, const float* std3_host,
                                                  void* stream) {
    return dino_preprocess_impl(true, image, image_dtype, patches3, B, H, W, OH, OW, mean3_host, std3_host, stream);
}

extern "C" int advgrpo_gather_l2norm_rows(const void* feats, const int64_t* idx, void* out, int B, int T, int D, int n,
                                          float eps, void* stream) {
    ADVGRPO_CHECK(feats && out && (n == 0 || idx) && B > 0 && T > 0 && D > 0, "gather_l2norm_rows: bad argument");
    const int rows = B * (1 + n);
    hipLaunchKernelGGL(gather_l2norm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream),
                       (const bf16_t*)feats, idx, (bf16_t*)out, B, T, D, n, eps);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_dino_head_combine(const void* hidden, const void* w2, const void* b2, int B, int Hd, int n,
                                         float cls_weight, float* hybrid, float* cls_score, float* patch_scores,
                                         void* stream) {
    ADVGRPO_CHECK(hidden && w2 && b2 && hybrid && cls_score && patch_scores && n >= 1 && n <= 256 && B > 0,
                  "dino_head_combine: bad argument");
    hipLaunchKernelGGL(dino_head_combine_kernel, dim3(B), dim3(256), 0, as_stream(stream), (const bf16_t*)hidden,
                       (const bf16_t*)w2, (const bf16_t*)b2, Hd, n, cls_weight, hybrid, cls_score, patch_scores);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_pickscore_pairs(const void* image_embs, const void* text_embs, int B, int P, float logit_scale_exp,
                                       float* scores, void* stream) {
    ADVGRPO_CHECK(image_embs && text_embs && scores && B > 0 && P > 0, "pickscore_pairs: bad argument");
    hipLaunchKernelGGL(pickscore_pairs_kernel, dim3(B), dim3(64), 0, as_stream(stream), (const bf16_t*)image_embs,
                       (const bf16_t*)text_embs, P, logit_scale_exp, scores);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
