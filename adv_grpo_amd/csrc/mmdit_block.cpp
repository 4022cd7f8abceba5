// mmdit_block.cpp -- one MMDiT block (diffusers JointTransformerBlock, SD3 / SD3.5 "MMDiT-X") behind ONE C-ABI entry.
//
// BASELINE.json's north_star names "the MMDiT block (fused QKV/attention/MLP)" as a unit behind the C-ABI; until round 5 the ABI stopped
// at GEMM / attention / norm granularity and the launch order of a block lived in Python (adv_grpo_amd/mmdit.py).  This file is that
// order in C++: a caller that is not Python hands over the block's weights, the two residual streams, the modulation rows and a
// workspace and gets the block's ~10 launches on its stream -- the same kernels, in the same order, with the same epilogue fusions, so
// the result is bit-identical to the Python-sequenced forward (tests/test_gpu_mmdit.py).  Reference: the transformer call at
// sd3_pipeline_with_logprob_fast.py:630-637 / train_sd3_fast_pickscore.py:235-255 (one of its num_layers blocks).
//
//   image: (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp [, shift_msa2, scale_msa2, gate_msa2]) = chunks of mods at mod_x
//   text : the same six at mod_c; the LAST block (context_pre_only) has (scale, shift) only
//   1  LayerNorm + modulate, both streams in one launch (second image output for the dual-attention blocks)
//   2  fused q|k|v projections of both streams in one launch, scattered into the joint [B, Ni + Nt, 3 D] buffer (image rows first),
//      per-head QK RMSNorm in the epilogue
//   3  joint attention (head dim 64)
//   4  output projections of both streams in one launch: x += gate_msa * (att_img Wo^T + b), c += c_gate_msa * (att_txt Wco^T + b)
//   5  dual blocks: q|k|v of the second (image-only) attention, its attention, x += gate_msa2 * (.. Wo2^T + b)
//   6  LayerNorm + modulate (MLP chunks), both streams in one launch
//   7  feed-forward 1 (+ GELU-tanh) of both streams in one launch, 8 feed-forward 2: x += gate_mlp * (..), c += c_gate_mlp * (..)
// Host-only code: it only calls this library's own C entries.
#include "common.hpp"

using namespace advgrpo;

namespace {

const char* bf16_at(const void* base, int64_t elems) { return reinterpret_cast<const char*>(base) + elems * 2; }

advgrpo_gemm_desc linear(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc, int M, int N, int K) {
    advgrpo_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.W = W; d.C = C; d.lda = lda; d.ldw = K; d.ldc = ldc;
    d.out_dtype = ADVGRPO_BF16; d.M = M; d.N = N; d.K = K; d.bias = bias; d.act = 0; d.alpha = 1.0f;
    return d;
}

}  // namespace

extern "C" int64_t advgrpo_mmdit_block_workspace_bytes(int B, int Ni, int Nt, int D, int dual) {
    const int64_t Mi = (int64_t)B * Ni, Mt = (int64_t)B * Nt, S = (int64_t)B * (Ni + Nt);
    int64_t e = Mi * D + Mt * D            // nx, nc
                + S * 3 * D + S * D        // qkv, att
                + Mi * 4 * D + Mt * 4 * D; // feed-forward hidden rows
    if (dual) e += Mi * D + Mi * 3 * D + Mi * D;      // nx2, qkv2, att2
    return e * 2 + 8 * 256;
}

extern "C" int advgrpo_mmdit_block_forward(const advgrpo_mmdit_block_desc* dsc, void* workspace, int64_t workspace_bytes, void* stream) {
    ADVGRPO_CHECK(dsc && workspace, "mmdit_block_forward: null argument");
    const advgrpo_mmdit_block_desc& d = *dsc;
    const int B = d.B, Ni = d.Ni, Nt = d.Nt, D = d.D, H = d.H;
    ADVGRPO_CHECK(B > 0 && Ni > 0 && Nt > 0 && D == H * 64, "mmdit_block_forward: needs head dim 64 (D = %d, H = %d)", D, H);
    ADVGRPO_CHECK(d.x && d.c && d.mods && d.qkv_w && d.cqkv_w && d.out_w && d.ff1_w && d.ff2_w && (d.last || (d.cout_w && d.cff1_w && d.cff2_w)) &&
                      (!d.dual || (d.qkv2_w && d.out2_w)),
                  "mmdit_block_forward: a weight of the block is missing");
    ADVGRPO_CHECK(workspace_bytes >= advgrpo_mmdit_block_workspace_bytes(B, Ni, Nt, D, d.dual) && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                  "mmdit_block_forward: workspace too small or not 256-byte aligned");
    const int S = Ni + Nt, Mi = B * Ni, Mt = B * Nt;
    // workspace carve-up (256-byte aligned pieces)
    char* w = reinterpret_cast<char*>(workspace);
    auto take = [&](int64_t elems) { char* p = w; w += (elems * 2 + 255) / 256 * 256; return p; };
    char* nx = take((int64_t)Mi * D);
    char* nc = take((int64_t)Mt * D);
    char* qkv = take((int64_t)B * S * 3 * D);
    char* att = take((int64_t)B * S * D);
    char* hx = take((int64_t)Mi * 4 * D);
    char* hc = take((int64_t)Mt * 4 * D);
    char *nx2 = nullptr, *qkv2 = nullptr, *att2 = nullptr;
    if (d.dual) { nx2 = take((int64_t)Mi * D); qkv2 = take((int64_t)Mi * 3 * D); att2 = take((int64_t)Mi * D); }
    auto mx = [&](int j) { return bf16_at(d.mods, d.mod_x + (int64_t)j * D); };
    auto mc = [&](int j) { return bf16_at(d.mods, d.mod_c + (int64_t)j * D); };
    const int cs = d.last ? 0 : 1, ch = d.last ? 1 : 0;                 // text-stream (scale, shift) chunks: AdaLayerNormContinuous in the last block
    int rc;
    // ---- 1: norms + modulation
    {
        advgrpo_ln_desc a, b;
        memset(&a, 0, sizeof(a)); memset(&b, 0, sizeof(b));
        a.x = d.x; a.ldx = D; a.out0 = nx; a.out1 = d.dual ? nx2 : nullptr; a.ldo = D; a.scale0 = mx(1); a.shift0 = mx(0);
        a.scale1 = d.dual ? mx(7) : nullptr; a.shift1 = d.dual ? mx(6) : nullptr; a.mod_stride = d.mod_stride; a.rows_per_batch = Ni;
        a.M = Mi; a.D = D; a.eps = 1e-6f;
        b.x = d.c; b.ldx = D; b.out0 = nc; b.ldo = D; b.scale0 = mc(cs); b.shift0 = mc(ch); b.mod_stride = d.mod_stride; b.rows_per_batch = Nt;
        b.M = Mt; b.D = D; b.eps = 1e-6f;
        if ((rc = advgrpo_layernorm_mod_pair(&a, &b, stream)) != 0) return rc;
    }
    // ---- 2: fused q|k|v of both streams into the joint buffer, QK RMSNorm in the epilogue
    {
        advgrpo_gemm_desc g[2] = {linear(nx, D, d.qkv_w, d.qkv_b, qkv, 3 * D, Mi, 3 * D, D), linear(nc, D, d.cqkv_w, d.cqkv_b, qkv, 3 * D, Mt, 3 * D, D)};
        g[0].seg_rows = Ni; g[0].seg_stride = S; g[0].seg_off = 0;
        g[1].seg_rows = Nt; g[1].seg_stride = S; g[1].seg_off = Ni;
        if (d.rms_x) { g[0].rms_weight = d.rms_x; g[0].rms_nheads = 2 * H; g[0].rms_heads_per_weight = H; g[0].rms_eps = 1e-6f; }
        if (d.rms_c) { g[1].rms_weight = d.rms_c; g[1].rms_nheads = 2 * H; g[1].rms_heads_per_weight = H; g[1].rms_eps = 1e-6f; }
        if ((rc = advgrpo_gemm_grouped(g, 2, stream)) != 0) return rc;
    }
    // ---- 3: joint attention over [image ; text] tokens
    if ((rc = advgrpo_attention_fwd(qkv, qkv + (int64_t)D * 2, qkv + (int64_t)2 * D * 2, att, 3 * D, 3 * D, 3 * D, D, (int64_t)S * 3 * D,
                                    (int64_t)S * 3 * D, (int64_t)S * 3 * D, (int64_t)S * D, B, H, S, S, 64, 0.125f, 0, nullptr, stream)) != 0)
        return rc;
    // ---- 4: output projections, gated, onto the residual streams
    {
        advgrpo_gemm_desc g[2] = {linear(att, D, d.out_w, d.out_b, d.x, D, Mi, D, D), linear(att, D, d.cout_w, d.cout_b, d.c, D, Mt, D, D)};
        g[0].a_seg_rows = Ni; g[0].a_seg_stride = S; g[0].a_seg_off = 0;
        g[0].gate = mx(2); g[0].gate_stride = d.mod_stride; g[0].gate_rows = Ni; g[0].residual = d.x; g[0].ldr = D;
        g[1].a_seg_rows = Nt; g[1].a_seg_stride = S; g[1].a_seg_off = Ni;
        g[1].gate = mc(2); g[1].gate_stride = d.mod_stride; g[1].gate_rows = Nt; g[1].residual = d.c; g[1].ldr = D;
        if ((rc = advgrpo_gemm_grouped(g, d.last ? 1 : 2, stream)) != 0) return rc;
    }
    // ---- 5: second, image-only attention of the dual blocks
    if (d.dual) {
        advgrpo_gemm_desc g = linear(nx2, D, d.qkv2_w, d.qkv2_b, qkv2, 3 * D, Mi, 3 * D, D);
        if (d.rms_2) { g.rms_weight = d.rms_2; g.rms_nheads = 2 * H; g.rms_heads_per_weight = H; g.rms_eps = 1e-6f; }
        if ((rc = advgrpo_gemm_grouped(&g, 1, stream)) != 0) return rc;
        if ((rc = advgrpo_attention_fwd(qkv2, qkv2 + (int64_t)D * 2, qkv2 + (int64_t)2 * D * 2, att2, 3 * D, 3 * D, 3 * D, D, (int64_t)Ni * 3 * D,
                                        (int64_t)Ni * 3 * D, (int64_t)Ni * 3 * D, (int64_t)Ni * D, B, H, Ni, Ni, 64, 0.125f, 0, nullptr, stream)) != 0)
            return rc;
        advgrpo_gemm_desc o = linear(att2, D, d.out2_w, d.out2_b, d.x, D, Mi, D, D);
        o.gate = mx(8); o.gate_stride = d.mod_stride; o.gate_rows = Ni; o.residual = d.x; o.ldr = D;
        if ((rc = advgrpo_gemm_grouped(&o, 1, stream)) != 0) return rc;
    }
    // ---- 6: norms + modulation of the feed-forwards
    if (!d.last) {
        advgrpo_ln_desc a, b;
        memset(&a, 0, sizeof(a)); memset(&b, 0, sizeof(b));
        a.x = d.x; a.ldx = D; a.out0 = nx; a.ldo = D; a.scale0 = mx(4); a.shift0 = mx(3); a.mod_stride = d.mod_stride; a.rows_per_batch = Ni;
        a.M = Mi; a.D = D; a.eps = 1e-6f;
        b.x = d.c; b.ldx = D; b.out0 = nc; b.ldo = D; b.scale0 = mc(4); b.shift0 = mc(3); b.mod_stride = d.mod_stride; b.rows_per_batch = Nt;
        b.M = Mt; b.D = D; b.eps = 1e-6f;
        if ((rc = advgrpo_layernorm_mod_pair(&a, &b, stream)) != 0) return rc;
    } else if ((rc = advgrpo_layernorm_mod(d.x, D, nx, nullptr, D, nullptr, nullptr, mx(4), mx(3), nullptr, nullptr, d.mod_stride, Ni, Mi, D, 1e-6f,
                                           stream)) != 0) {
        return rc;
    }
    // ---- 7, 8: feed-forwards
    {
        advgrpo_gemm_desc g[2] = {linear(nx, D, d.ff1_w, d.ff1_b, hx, 4 * D, Mi, 4 * D, D), linear(nc, D, d.cff1_w, d.cff1_b, hc, 4 * D, Mt, 4 * D, D)};
        g[0].act = 1; g[1].act = 1;                                     // GELU (tanh approximation)
        if ((rc = advgrpo_gemm_grouped(g, d.last ? 1 : 2, stream)) != 0) return rc;
        advgrpo_gemm_desc f[2] = {linear(hx, 4 * D, d.ff2_w, d.ff2_b, d.x, D, Mi, D, 4 * D), linear(hc, 4 * D, d.cff2_w, d.cff2_b, d.c, D, Mt, D, 4 * D)};
        f[0].gate = mx(5); f[0].gate_stride = d.mod_stride; f[0].gate_rows = Ni; f[0].residual = d.x; f[0].ldr = D;
        f[1].gate = mc(5); f[1].gate_stride = d.mod_stride; f[1].gate_rows = Nt; f[1].residual = d.c; f[1].ldr = D;
        if ((rc = advgrpo_gemm_grouped(f, d.last ? 1 : 2, stream)) != 0) return rc;
    }
    return 0;
}
