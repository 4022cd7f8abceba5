// vit_encoder.cpp -- the pre-LN transformer encoder stack of the reward towers behind ONE C-ABI entry ("advgrpo_vit_forward").
//
// BASELINE.json's north_star names the "CLIP/DINO ViT reward forward" as a unit behind the C-ABI; until round 6 the launch order of a tower
// lived in Python (adv_grpo_amd/vit.py: _Encoder.__call__).  This file is that order in C++ for the whole stack: CLIP ViT-H/14's 32 vision
// layers and 24 causal text layers (transformers' CLIPEncoderLayer behind CLIPModel.get_image_features / get_text_features,
// adv_grpo/pickscore_scorer.py:40-44, adv_grpo/pick_score_training.py:95-106) and DINOv2 ViT-B/14's 12 blocks with LayerScale (timm's
// forward_features, adv_grpo/rewards.py:397, scripts/train_sd3_fast_dino_patch.py:183-184).  Per layer, on the caller's stream:
//   1  h = LayerNorm(x) (affine)                     2  qkv = h Wqkv^T + b            (q | k | v packed along N)
//   3  o = softmax(q k^T / sqrt(d) [causal]) v       4  x += [ls1 *] (o Wo^T + b)      (residual -- and LayerScale as the gate operand -- in the epilogue)
//   5  h = LayerNorm(x)                              6  m = act(h W1^T + b)            7  x += [ls2 *] (m W2^T + b)
// The same kernels, in the same order, with the same epilogue fusions as the Python sequencing: bit-identical (tests/test_gpu_vit.py).
// Embedding (patch GEMM + position rows / token lookup), the final norm and the projection stay with the caller: they differ per tower.
// Host-only code: it only calls this library's own C entries.
#include "common.hpp"

using namespace advgrpo;

extern "C" int64_t advgrpo_vit_workspace_bytes(int B, int S, int D, int mlp) {
    const int64_t M = (int64_t)B * S;
    auto piece = [](int64_t elems) { return (elems * 2 + 255) / 256 * 256; };
    return piece(M * D) + piece(M * 3 * D) + piece(M * D) + piece(M * mlp);
}

extern "C" int advgrpo_vit_forward(const advgrpo_vit_desc* dsc, void* workspace, int64_t workspace_bytes, void* stream) {
    ADVGRPO_CHECK(dsc && workspace, "vit_forward: null argument");
    const advgrpo_vit_desc& d = *dsc;
    const int B = d.B, S = d.S, D = d.D, H = d.H, F = d.mlp;
    ADVGRPO_CHECK(B > 0 && S > 0 && D > 0 && H > 0 && F > 0 && D % H == 0 && d.n_layers >= 0 && d.x && (d.layers || d.n_layers == 0),
                  "vit_forward: bad descriptor (B=%d S=%d D=%d H=%d mlp=%d layers=%d)", B, S, D, H, F, d.n_layers);
    const int hd = D / H;
    ADVGRPO_CHECK(hd == 64 || hd == 80, "vit_forward: head dim %d (the towers on this path have 64: DINOv2-B, CLIP text; 80: CLIP ViT-H)", hd);
    ADVGRPO_CHECK(workspace_bytes >= advgrpo_vit_workspace_bytes(B, S, D, F) && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                  "vit_forward: workspace too small or not 256-byte aligned");
    const int M = B * S;
    char* w = reinterpret_cast<char*>(workspace);
    auto take = [&](int64_t elems) { char* p = w; w += (elems * 2 + 255) / 256 * 256; return p; };
    char* h = take((int64_t)M * D);
    char* qkv = take((int64_t)M * 3 * D);
    char* o = take((int64_t)M * D);
    char* m = take((int64_t)M * F);
    const float scale = 1.0f / sqrtf((float)hd);
    int rc;
    for (int l = 0; l < d.n_layers; ++l) {
        const advgrpo_vit_layer& L = d.layers[l];
        ADVGRPO_CHECK(L.ln1_w && L.ln1_b && L.qkv_w && L.out_w && L.ln2_w && L.ln2_b && L.fc1_w && L.fc2_w, "vit_forward: layer %d: a weight is missing", l);
        if ((rc = advgrpo_layernorm_mod(d.x, D, h, nullptr, D, L.ln1_w, L.ln1_b, nullptr, nullptr, nullptr, nullptr, 0, 0, M, D, d.eps, stream)) != 0) return rc;
        if ((rc = advgrpo_gemm_bf16(h, D, L.qkv_w, D, qkv, 3 * D, ADVGRPO_BF16, M, 3 * D, D, L.qkv_b, 0, 1.0f, nullptr, 0, 0, nullptr, 0, 0, 0, 0, 0, 0, 0, 1,
                                    0, 0, 0, stream)) != 0)
            return rc;
        if ((rc = advgrpo_attention_fwd(qkv, qkv + (int64_t)D * 2, qkv + (int64_t)2 * D * 2, o, 3 * D, 3 * D, 3 * D, D, (int64_t)S * 3 * D, (int64_t)S * 3 * D,
                                        (int64_t)S * 3 * D, (int64_t)S * D, B, H, S, S, hd, scale, d.causal, nullptr, stream)) != 0)
            return rc;
        if ((rc = advgrpo_gemm_bf16(o, D, L.out_w, D, d.x, D, ADVGRPO_BF16, M, D, D, L.out_b, 0, 1.0f, L.ls1, L.ls1 ? D : 0, L.ls1 ? M : 0, d.x, D, 0, 0, 0, 0, 0, 0,
                                    1, 0, 0, 0, stream)) != 0)
            return rc;
        if ((rc = advgrpo_layernorm_mod(d.x, D, h, nullptr, D, L.ln2_w, L.ln2_b, nullptr, nullptr, nullptr, nullptr, 0, 0, M, D, d.eps, stream)) != 0) return rc;
        if ((rc = advgrpo_gemm_bf16(h, D, L.fc1_w, D, m, F, ADVGRPO_BF16, M, F, D, L.fc1_b, d.act, 1.0f, nullptr, 0, 0, nullptr, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0,
                                    stream)) != 0)
            return rc;
        if ((rc = advgrpo_gemm_bf16(m, F, L.fc2_w, F, d.x, D, ADVGRPO_BF16, M, D, F, L.fc2_b, 0, 1.0f, L.ls2, L.ls2 ? D : 0, L.ls2 ? M : 0, d.x, D, 0, 0, 0, 0, 0, 0,
                                    1, 0, 0, 0, stream)) != 0)
            return rc;
    }
    return 0;
}
