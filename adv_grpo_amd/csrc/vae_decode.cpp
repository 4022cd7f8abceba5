// vae_decode.cpp -- the SD3 / SD3.5 VAE decoder (fp32-equivalent arithmetic) behind ONE C-ABI entry ("advgrpo_vae_decode").
//
// Stands in for `pipeline.vae.decode(latents / scaling_factor + shift_factor)` + `image_processor.postprocess(image, "pt")` at
// adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:667-670 (the reference decodes in fp32: train_sd3_fast_pickscore.py:481).  Until
// round 6 the launch order of the decoder -- ~190 launches -- lived in Python (adv_grpo_amd/vae.py: _decode_x3_chain); this file is that order
// in C++: conv_in, the mid block (resnet, single-head attention over the h x w pixels, resnet), the up blocks (resnets, nearest x2 upsample
// folded into the next convolution's gather), GroupNorm + SiLU + conv_out, postprocess.  Every 3x3 convolution runs in the arithmetic its
// weight allows, chosen per tensor by the CALLER at load time (advgrpo_vae_conv.form): 1 = one fp16 piece, activations as an fp16 pair, two
// products ("f16x2": the released VAE is an fp16 checkpoint); 0 = split-bf16 operands, three products ("bf16x3").  Everything between two
// matrix products stays f32.  The fusions of the Python sequencing are kept one for one -- GroupNorm statistics from the producing
// convolution's epilogue (16-pixel x 4-channel block sums), a resnet's second convolution in front of an upsampler writing the upsampler's
// operand rows directly -- so the image is bit-identical to it (tests/test_gpu_vae.py).
// Host-only code: it only calls this library's own C entries.
#include <math.h>

#include <algorithm>

#include "common.hpp"

using namespace advgrpo;

namespace {

constexpr float RAW_PRESCALE = 0.0625f;     // 2^-4: un-normalised convolution inputs (the upsamplers') as fp16 pairs stay in range
constexpr int STAT_ROWS = 16;               // ADVGRPO_CONV_F16X2_STAT_ROWS: pixels per block sum of the convolution epilogue

int64_t piece(int64_t bytes) { return (bytes + 255) / 256 * 256; }

struct Plan {                                // workspace sizes from one walk over the architecture
    int64_t max_elems = 0;                   // largest f32 activation [B, H, W, C]
    int64_t max_rows = 0;                    // largest convolution INPUT [B, H, W, Cin] (operand rows: 6 bytes per element)
    int64_t max_pair = 0;                    // largest pair-row OUTPUT (a block's last resnet in front of an upsampler)
    int64_t gn_scratch = 0;
    // mid-block attention scratch: the score matrix goes into activation slot 0 (free during the attention) when it fits, everything else
    // into the span from behind the attention's output (slot 2) to the end of the pair rows, which no convolution uses at that moment;
    // what does not fit there is `attn_extra` bytes behind the GroupNorm scratch
    int64_t attn_s = 0, attn_rest = 0, attn_extra = 0;
    bool s_in_slot0 = false;
    int64_t slot_bytes() const { return piece(max_elems * 4); }
    int64_t rows_bytes() const { return piece(max_rows * 6); }
    int64_t pair_bytes() const { return piece(max_pair * 6 + 256); }
    int64_t part_bytes() const { return piece(max_elems / 32 * 4 + 256); }
    int64_t total() const { return 3 * slot_bytes() + rows_bytes() + pair_bytes() + 2 * part_bytes() + piece(gn_scratch) + attn_extra + 256; }
};

Plan plan_of(const advgrpo_vae_decoder_desc& d) {
    Plan p;
    const int B = d.B;
    int H = d.h, W = d.w;
    auto see = [&](int c) { p.max_elems = std::max<int64_t>(p.max_elems, (int64_t)B * H * W * c); };
    auto see_in = [&](int c) { p.max_rows = std::max<int64_t>(p.max_rows, (int64_t)B * H * W * c); };
    see(std::max(d.conv_in.cout, 64));
    see_in(std::max(d.conv_in.cout, 64));    // (latent rows padded to 64 channels; the mid resnets' inputs)
    p.gn_scratch = std::max<int64_t>(p.gn_scratch, advgrpo_groupnorm_scratch_bytes(B, H * W, d.groups));
    const int64_t T0 = (int64_t)H * W, C0 = d.conv_in.cout;
    for (int i = 0; i < d.n_up; ++i) {
        for (int j = 0; j < d.resnets_per_up; ++j) {
            const advgrpo_vae_resnet& r = d.up_resnets[i * d.resnets_per_up + j];
            see(r.conv1.cin); see(r.conv1.cout); see(r.conv2.cout);
            see_in(r.conv1.cin); see_in(r.conv2.cin);
            p.gn_scratch = std::max<int64_t>(p.gn_scratch, advgrpo_groupnorm_scratch_bytes(B, H * W, d.groups));
        }
        if (i < d.n_up - 1) {
            const advgrpo_vae_resnet& last = d.up_resnets[i * d.resnets_per_up + d.resnets_per_up - 1];
            p.max_pair = std::max<int64_t>(p.max_pair, (int64_t)B * H * W * last.conv2.cout);
            see_in(d.upsamplers[i].cin);
            H *= 2; W *= 2;
            see(d.upsamplers[i].cout);
        }
    }
    see_in(d.conv_out.cin);
    p.gn_scratch = std::max<int64_t>(p.gn_scratch, advgrpo_groupnorm_scratch_bytes(B, H * W, d.groups));
    // attention scratch (sizes at the mid block's resolution)
    p.attn_s = piece(B * T0 * T0 * 4);
    p.attn_rest = piece(B * T0 * 3 * C0 * 2) * 4          // h3, q3, k3, o3
                  + piece(B * T0 * C0 * 4) * 4            // q, k, o, y
                  + piece(B * C0 * T0 * 4) + piece(B * C0 * 3 * T0 * 2)   // vt, vt3
                  + piece(B * T0 * 3 * T0 * 2);           // p3
    p.s_in_slot0 = p.attn_s <= p.slot_bytes();
    const int64_t span = p.slot_bytes() - piece(B * T0 * C0 * 4) + p.rows_bytes() + p.pair_bytes();
    const int64_t need = p.attn_rest + (p.s_in_slot0 ? 0 : p.attn_s);
    p.attn_extra = need <= span ? 0 : need;
    return p;
}

}  // namespace

extern "C" int64_t advgrpo_vae_decode_workspace_bytes(const advgrpo_vae_decoder_desc* d) {
    if (!d || d->B <= 0 || d->h <= 0 || d->w <= 0 || d->n_up <= 0 || d->resnets_per_up <= 0 || !d->up_resnets || (d->n_up > 1 && !d->upsamplers)) return -1;
    return plan_of(*d).total();
}

extern "C" int advgrpo_vae_decode(const advgrpo_vae_decoder_desc* dsc, const void* latents, int latents_dtype, float* image, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    ADVGRPO_CHECK(dsc && latents && image && workspace, "vae_decode: null argument");
    const advgrpo_vae_decoder_desc& d = *dsc;
    ADVGRPO_CHECK(d.B > 0 && d.h > 0 && d.w > 0 && d.n_up > 0 && d.resnets_per_up > 0 && d.groups > 0 && d.up_resnets && (d.n_up == 1 || d.upsamplers) &&
                      d.zero_page && d.conv_in.w && d.conv_out.w && d.norm_out_w && d.norm_out_b && d.attn_q_w && d.attn_k_w && d.attn_v_w && d.attn_o_w,
                  "vae_decode: bad descriptor");
    ADVGRPO_CHECK(d.conv_in.form == 0 && d.conv_out.form == 0 && d.conv_in.cin <= 64, "vae_decode: conv_in / conv_out run on split-bf16 operands (form 0), conv_in from <= 64 latent channels");
    ADVGRPO_CHECK(workspace_bytes >= advgrpo_vae_decode_workspace_bytes(dsc) && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                  "vae_decode: workspace too small or not 256-byte aligned");
    const Plan plan = plan_of(d);
    const int B = d.B, G = d.groups;
    char* w = reinterpret_cast<char*>(workspace);
    auto take = [&](int64_t bytes) { char* p = w; w += piece(bytes); return p; };
    float* act[3] = {reinterpret_cast<float*>(take(plan.slot_bytes())), reinterpret_cast<float*>(take(plan.slot_bytes())),
                     reinterpret_cast<float*>(take(plan.slot_bytes()))};
    char* rows = take(plan.rows_bytes());                   // operand rows [.., 3 C] 16-bit of the convolution about to run
    char* pair = take(plan.pair_bytes());                   // operand rows written by a producing convolution's epilogue
    float* part[2] = {reinterpret_cast<float*>(take(plan.part_bytes())), reinterpret_cast<float*>(take(plan.part_bytes()))};
    double* gn = reinterpret_cast<double*>(take(plan.gn_scratch));
    char* attn_extra = plan.attn_extra ? take(plan.attn_extra) : nullptr;
    int rc;
    int H = d.h, W = d.w;
    // an f32 activation: where it lives and -- when an f16x2 convolution produced it -- the block sums its GroupNorm reads instead of the data
    struct Act { int slot; const float* stats; };
    auto other = [&](int a, int b) { for (int s = 0; s < 3; ++s) if (s != a && s != b) return s; return 0; };
    int part_next = 0;
    const float alpha_raw = 1.0f / RAW_PRESCALE;

    // 3x3 convolution of an f32 activation after GroupNorm + SiLU, in the arithmetic its weight allows (vae.py: _conv_auto with gn given)
    auto conv_gn = [&](const advgrpo_vae_conv& c, const float* nw, const float* nb, const Act& x, const float* residual, int out_slot, bool pair_out,
                       Act& out) -> int {
        const int HW = H * W;
        if (c.form == 1) {
            if ((rc = advgrpo_groupnorm_nhwc_f16x2(act[x.slot], rows, gn, nw, nb, B, HW, c.cin, G, 1e-6f, 1, 1.0f, x.stats, STAT_ROWS, stream)) != 0) return rc;
            if (pair_out) {
                out = Act{-1, nullptr};
                if (d.f16_single)
                    return advgrpo_conv3x3_nhwc_f16x1_pair(rows, c.w, pair, RAW_PRESCALE, B, H, W, 3 * c.cin, c.cout, 0, c.bias, 0, residual, d.zero_page, 1.0f, stream);
                return advgrpo_conv3x3_nhwc_f16x2_pair(rows, c.w, pair, RAW_PRESCALE, B, H, W, 3 * c.cin, c.cout, 0, c.bias, 0, residual, d.zero_page, 1.0f, stream);
            }
            float* st = (HW % STAT_ROWS == 0) ? part[part_next] : nullptr;
            if (st) part_next ^= 1;
            out = Act{out_slot, st};
            if (d.f16_single)
                return advgrpo_conv3x3_nhwc_f16x1(rows, c.w, act[out_slot], B, H, W, 3 * c.cin, c.cout, 0, c.bias, 0, residual, d.zero_page, 1.0f, st, stream);
            return advgrpo_conv3x3_nhwc_f16x2(rows, c.w, act[out_slot], B, H, W, 3 * c.cin, c.cout, 0, c.bias, 0, residual, d.zero_page, 1.0f, st, stream);
        }
        if ((rc = advgrpo_groupnorm_nhwc_x3(act[x.slot], rows, gn, nw, nb, B, HW, c.cin, G, 1e-6f, 1, c.cout >= 128 ? 1 : 0, stream)) != 0) return rc;
        out = Act{out_slot, nullptr};
        return advgrpo_conv3x3_nhwc_x3(rows, c.w, act[out_slot], B, H, W, 3 * c.cin, c.cout, 0, c.bias, 0, residual, d.zero_page, stream);
    };
    // ResnetBlock2D (vae.py: _res3): x -> conv1(SiLU(GN(x))) -> conv2(SiLU(GN(.))) + shortcut(x); pair_out: the result leaves as the fp16-pair
    // rows of the upsampler that reads it (and nothing else does)
    auto resnet = [&](const advgrpo_vae_resnet& r, Act x, bool pair_out, Act& out) -> int {
        const int hs = (x.slot + 1) % 3, ss = other(x.slot, hs);
        Act h;
        if ((rc = conv_gn(r.conv1, r.norm1_w, r.norm1_b, x, nullptr, hs, false, h)) != 0) return rc;
        const float* sc = act[x.slot];
        if (r.shortcut_w) {     // 1 x 1 convolution on split operands (its bias rides on conv2's)
            const int64_t M = (int64_t)B * H * W;
            if ((rc = advgrpo_split_bf16x3(act[x.slot], nullptr, rows, M, r.conv1.cin, 0, stream)) != 0) return rc;
            if ((rc = advgrpo_gemm_bf16(rows, 3 * r.conv1.cin, r.shortcut_w, 3 * r.conv1.cin, act[ss], r.conv2.cout, ADVGRPO_F32, (int)M, r.conv2.cout, 3 * r.conv1.cin,
                                        nullptr, 0, 1.0f, nullptr, 0, 0, nullptr, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, stream)) != 0)
                return rc;
            sc = act[ss];
        }
        // (the only form with a pair output: both this convolution and its reader on fp16 pieces)
        return conv_gn(r.conv2, r.norm2_w, r.norm2_b, h, sc, hs, pair_out, out);
    };

    // ---- conv_in
    if ((rc = advgrpo_latents_to_nhwc_x3(latents, latents_dtype, rows, B, d.latent_channels, H, W, 64, d.scaling_factor, d.shift_factor, stream)) != 0) return rc;
    if ((rc = advgrpo_conv3x3_nhwc_x3(rows, d.conv_in.w, act[0], B, H, W, 3 * 64, d.conv_in.cout, 0, d.conv_in.bias, 0, nullptr, d.zero_page, stream)) != 0) return rc;
    Act x{0, nullptr};
    // ---- mid block
    if ((rc = resnet(d.mid[0], x, false, x)) != 0) return rc;
    {   // single-head attention over the T = H W pixels, every product on split operands, f32 in between (vae.py: _attn3_core)
        const int C = d.conv_in.cout;
        const int64_t T = (int64_t)H * W, M = B * T;
        // the input sits in slot 1, the output goes to slot 2 (conv_in -> slot 0, the first mid resnet -> slot 1): slot 0 and everything from
        // behind the output to the end of the pair rows is free until the next convolution (Plan)
        ADVGRPO_CHECK(x.slot == 1, "vae_decode: internal slot order changed");
        char* a = attn_extra ? attn_extra : reinterpret_cast<char*>(act[2]) + piece(M * C * 4);
        auto tk = [&](int64_t bytes) { char* p = a; a += piece(bytes); return p; };
        char* h3 = tk(M * 3 * C * 2); char* q3 = tk(M * 3 * C * 2); char* k3 = tk(M * 3 * C * 2); char* o3 = tk(M * 3 * C * 2);
        float* q = reinterpret_cast<float*>(tk(M * C * 4)); float* k = reinterpret_cast<float*>(tk(M * C * 4));
        float* o = reinterpret_cast<float*>(tk(M * C * 4)); float* y = reinterpret_cast<float*>(tk(M * C * 4));
        float* vt = reinterpret_cast<float*>(tk((int64_t)B * C * T * 4)); char* vt3 = tk((int64_t)B * C * 3 * T * 2);
        float* s = plan.s_in_slot0 ? act[0] : reinterpret_cast<float*>(tk((int64_t)B * T * T * 4));
        char* p3 = tk((int64_t)B * T * 3 * T * 2);
        auto gemm_f32 = [&](const void* A, int64_t lda, const void* Wt, int64_t ldw, float* Cc, int64_t ldc, int Mm, int N, int K, float alpha, int batch,
                            int64_t sA, int64_t sW, int64_t sC) {
            return advgrpo_gemm_bf16(A, lda, Wt, ldw, Cc, ldc, ADVGRPO_F32, Mm, N, K, nullptr, 0, alpha, nullptr, 0, 0, nullptr, 0, 0, 0, 0, 0, 0, 0, batch, sA, sW, sC,
                                     stream);
        };
        if ((rc = advgrpo_groupnorm_nhwc_x3(act[x.slot], h3, gn, d.attn_norm_w, d.attn_norm_b, B, (int)T, C, G, 1e-6f, 0, 0, stream)) != 0) return rc;
        if ((rc = gemm_f32(h3, 3 * C, d.attn_q_w, 3 * C, q, C, (int)M, C, 3 * C, 1.0f, 1, 0, 0, 0)) != 0) return rc;
        if ((rc = advgrpo_split_bf16x3(q, d.attn_q_b, q3, M, C, 0, stream)) != 0) return rc;
        if ((rc = gemm_f32(h3, 3 * C, d.attn_k_w, 3 * C, k, C, (int)M, C, 3 * C, 1.0f, 1, 0, 0, 0)) != 0) return rc;
        if ((rc = advgrpo_split_bf16x3(k, d.attn_k_b, k3, M, C, 1, stream)) != 0) return rc;
        // V^T[b] = Wv . h[b]^T (the bias is added after P . V: rows of P sum to one); the weight is the [hi | lo | hi] side
        if ((rc = gemm_f32(d.attn_v_w, 3 * C, h3, 3 * C, vt, T, C, (int)T, 3 * C, 1.0f, B, 0, T * 3 * C, (int64_t)C * T)) != 0) return rc;
        if ((rc = advgrpo_split_bf16x3(vt, nullptr, vt3, (int64_t)B * C, (int)T, 1, stream)) != 0) return rc;
        if ((rc = gemm_f32(q3, 3 * C, k3, 3 * C, s, T, (int)T, (int)T, 3 * C, (float)pow((double)C, -0.5), B, T * 3 * C, T * 3 * C, T * T)) != 0) return rc;
        if ((rc = advgrpo_softmax_rows_x3(s, p3, M, (int)T, stream)) != 0) return rc;
        if ((rc = gemm_f32(p3, 3 * T, vt3, 3 * T, o, C, (int)T, C, (int)(3 * T), 1.0f, B, T * 3 * T, (int64_t)C * 3 * T, T * C)) != 0) return rc;
        if ((rc = advgrpo_split_bf16x3(o, d.attn_v_b, o3, M, C, 0, stream)) != 0) return rc;
        if ((rc = gemm_f32(o3, 3 * C, d.attn_o_w, 3 * C, y, C, (int)M, C, 3 * C, 1.0f, 1, 0, 0, 0)) != 0) return rc;
        const int ns = 2;
        if ((rc = advgrpo_add_rows_f32(y, act[x.slot], d.attn_o_b, act[ns], M, C, stream)) != 0) return rc;
        x = Act{ns, nullptr};
    }
    if ((rc = resnet(d.mid[1], x, false, x)) != 0) return rc;
    // ---- up blocks
    for (int i = 0; i < d.n_up; ++i) {
        const advgrpo_vae_conv* up = i < d.n_up - 1 ? &d.upsamplers[i] : nullptr;
        for (int j = 0; j < d.resnets_per_up; ++j) {
            const advgrpo_vae_resnet& r = d.up_resnets[i * d.resnets_per_up + j];
            // the block's last resnet feeds the upsampler's convolution and nothing else: its conv2 writes that operand directly when both are on fp16 pieces
            const bool pair_out = up && j == d.resnets_per_up - 1 && r.conv2.form == 1 && up->form == 1;
            if ((rc = resnet(r, x, pair_out, x)) != 0) return rc;
        }
        if (up) {
            const int Ho = 2 * H, Wo = 2 * W;
            if (x.slot < 0) {            // operand rows from the producing convolution's epilogue
                const int os = 0;
                float* st = ((Ho * Wo) % STAT_ROWS == 0) ? part[part_next] : nullptr;
                if (st) part_next ^= 1;
                rc = d.f16_single ? advgrpo_conv3x3_nhwc_f16x1(pair, up->w, act[os], B, Ho, Wo, 3 * up->cin, up->cout, 1, up->bias, 0, nullptr, d.zero_page, alpha_raw, st, stream)
                                  : advgrpo_conv3x3_nhwc_f16x2(pair, up->w, act[os], B, Ho, Wo, 3 * up->cin, up->cout, 1, up->bias, 0, nullptr, d.zero_page, alpha_raw, st, stream);
                if (rc) return rc;
                x = Act{os, st};
            } else if (up->form == 1) {
                const int os = (x.slot + 1) % 3;
                if ((rc = advgrpo_split_f16x2(act[x.slot], nullptr, rows, (int64_t)B * H * W, up->cin, RAW_PRESCALE, stream)) != 0) return rc;
                float* st = ((Ho * Wo) % STAT_ROWS == 0) ? part[part_next] : nullptr;
                if (st) part_next ^= 1;
                rc = d.f16_single ? advgrpo_conv3x3_nhwc_f16x1(rows, up->w, act[os], B, Ho, Wo, 3 * up->cin, up->cout, 1, up->bias, 0, nullptr, d.zero_page, alpha_raw, st, stream)
                                  : advgrpo_conv3x3_nhwc_f16x2(rows, up->w, act[os], B, Ho, Wo, 3 * up->cin, up->cout, 1, up->bias, 0, nullptr, d.zero_page, alpha_raw, st, stream);
                if (rc) return rc;
                x = Act{os, st};
            } else {
                const int os = (x.slot + 1) % 3;
                if ((rc = advgrpo_split_bf16x3(act[x.slot], nullptr, rows, (int64_t)B * H * W, up->cin, up->cout >= 128 ? 2 : 0, stream)) != 0) return rc;
                if ((rc = advgrpo_conv3x3_nhwc_x3(rows, up->w, act[os], B, Ho, Wo, 3 * up->cin, up->cout, 1, up->bias, 0, nullptr, d.zero_page, stream)) != 0) return rc;
                x = Act{os, nullptr};
            }
            H = Ho; W = Wo;
        }
    }
    // ---- conv_norm_out + SiLU + conv_out, postprocess
    const int cl = d.conv_out.cin;
    if ((rc = advgrpo_groupnorm_nhwc_x3(act[x.slot], rows, gn, d.norm_out_w, d.norm_out_b, B, H * W, cl, G, 1e-6f, 1, d.conv_out.cout >= 128 ? 1 : 0, stream)) != 0) return rc;
    const int ys = (x.slot + 1) % 3;
    if ((rc = advgrpo_conv3x3_nhwc_x3(rows, d.conv_out.w, act[ys], B, H, W, 3 * cl, d.conv_out.cout, 0, d.conv_out.bias, 0, nullptr, d.zero_page, stream)) != 0) return rc;
    return advgrpo_image_postprocess(act[ys], ADVGRPO_F32, d.conv_out.cout, image, B, H, W, stream);
}
