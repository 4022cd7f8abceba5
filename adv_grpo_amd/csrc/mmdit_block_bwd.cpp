// mmdit_block_bwd.cpp -- the data-gradient chain of one MMDiT block (diffusers JointTransformerBlock) behind ONE C-ABI entry.
//
// The twin of mmdit_block.cpp for the update half: autograd of the transformer call inside compute_log_prob
// (scripts/train_sd3_fast_pickscore.py:233-267) as reached from loss.backward() at :1165, for ONE of its num_layers blocks.  Until round 6
// this launch order lived in Python (adv_grpo_amd/mmdit_train.py: SD3TransformerLoRA.backward); a caller that is not Python hands over the
// block's TRANSPOSED weights (the data-gradient GEMMs read W^T in nn.Linear layout), what the forward saved, the modulation rows and the
// gradients arriving from block i + 1, and gets the block's ~14 launches on its stream -- the same kernels, in the same order, with the
// same epilogue fusions: bit-identical to the Python sequencing (tests/test_gpu_train.py).
//
//   in : dx, dc      gradients of the block's outputs (residual streams);  dyg = gate_mlp * dx, dcyg = c_gate_mlp * dc (made by block i + 1's
//                    first-norm backward, or by the final layer's: the pass that produces a stream gradient writes its gated copies)
//   1  d(ff hidden) = (dyg W2) * gelu'(pre)   both streams, one launch      2  d(mlp norm out) = d(ff hidden) W1        one launch
//   3  dx1 = dx + LNmod'(x_mid; d(mlp norm out)), + gated copies dyo = gate_msa * dx1 [, dy2 = gate_msa2 * dx1];  dc1, dyc likewise
//   4  dual blocks: datt2 = dy2 Wo2, attention backward, QK-norm backward, dnx2 = dqkv2 Wqkv2
//   5  datt = [dyo Wo ; dyc Wco] scattered into the joint rows (one launch), attention backward, QK-norm backward of both streams
//   6  dnx = dqkv[image rows] Wqkv, dnc = dqkv[text rows] Wcqkv (one launch)
//   7  dx_out = dx1 + LNmod'(x_in; dnx [, dnx2]), dc_out = dc1 + LNmod'(c_in; dnc), each with the gated copy block i - 1's step 1 reads
// Left for the caller: the LoRA adapter gradients.  They read (att, dyo / dyc) and (nx / nc, dqkv): dyo, dyc and dqkv are OUTPUTS of this
// entry for that reason (the caller runs advgrpo_gemm_tn_grouped on a side stream beside the next block's chain).
// Host-only code: it only calls this library's own C entries.
#include "common.hpp"

using namespace advgrpo;

namespace {

const char* bf16_at(const void* base, int64_t elems) { return reinterpret_cast<const char*>(base) + elems * 2; }

// one data-gradient Linear: C[M, N] = A[M, K] . WT[N, K]^T  (WT = the forward weight transposed, in nn.Linear layout)
advgrpo_gemm_desc dgrad(const void* A, int64_t lda, const void* WT, void* C, int64_t ldc, int M, int N, int K) {
    advgrpo_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.W = WT; d.C = C; d.lda = lda; d.ldw = K; d.ldc = ldc;
    d.out_dtype = ADVGRPO_BF16; d.M = M; d.N = N; d.K = K; d.act = 0; d.alpha = 1.0f;
    return d;
}

int64_t piece(int64_t bytes) { return (bytes + 255) / 256 * 256; }

}  // namespace

extern "C" int64_t advgrpo_mmdit_block_backward_workspace_bytes(int B, int Ni, int Nt, int D, int H, int dual) {
    const int64_t Mi = (int64_t)B * Ni, Mt = (int64_t)B * Nt, S = Ni + Nt;
    int64_t b = piece(Mi * 4 * D * 2) + piece(Mt * 4 * D * 2)        // d(ff hidden)
                + piece(Mi * D * 2) + piece(Mt * D * 2)              // d(mlp norm out)
                + piece(Mi * D * 2) + piece(Mt * D * 2)              // dx1, dc1
                + piece((int64_t)B * S * D * 2)                      // datt (joint rows)
                + piece(Mi * D * 2) + piece(Mt * D * 2)              // dnx, dnc
                + piece((int64_t)B * H * ((S + 31) / 32) * 64 * 4);  // attention backward scratch
    if (dual) b += 3 * piece(Mi * D * 2) + piece(Mi * 3 * D * 2);    // dy2, datt2, dnx2, dqkv2
    return b + 256;
}

extern "C" int advgrpo_mmdit_block_backward(const advgrpo_mmdit_block_bwd_desc* dsc, void* workspace, int64_t workspace_bytes, void* stream) {
    ADVGRPO_CHECK(dsc && workspace, "mmdit_block_backward: null argument");
    const advgrpo_mmdit_block_bwd_desc& d = *dsc;
    const int B = d.B, Ni = d.Ni, Nt = d.Nt, D = d.D, H = d.H;
    ADVGRPO_CHECK(B > 0 && Ni > 0 && Nt > 0 && D == H * 64, "mmdit_block_backward: needs head dim 64 (D = %d, H = %d)", D, H);
    ADVGRPO_CHECK(d.mods && d.ff2_wT && d.ff1_wT && d.out_wT && d.qkv_wT && d.cqkv_wT && (d.last || (d.cff2_wT && d.cff1_wT && d.cout_wT)) &&
                      (!d.dual || (d.out2_wT && d.qkv2_wT)),
                  "mmdit_block_backward: a transposed weight of the block is missing");
    ADVGRPO_CHECK(d.x_in && d.c_in && d.x_mid && d.pre && d.qkv && d.rs && d.att && d.lse && (d.last || (d.c_mid && d.cpre)) &&
                      (!d.dual || (d.qkv2 && d.rs2 && d.att2 && d.lse2)),
                  "mmdit_block_backward: a saved activation of the forward is missing");
    ADVGRPO_CHECK(d.dx && d.dyg && (d.last || (d.dc && d.dcyg)) && d.dx_out && d.dc_out && d.dyo && (d.last || d.dyc) && d.dqkv &&
                      (d.first || (d.dyg_prev && d.dcyg_prev)),
                  "mmdit_block_backward: a gradient buffer is missing");
    ADVGRPO_CHECK(workspace_bytes >= advgrpo_mmdit_block_backward_workspace_bytes(B, Ni, Nt, D, H, d.dual) &&
                      (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                  "mmdit_block_backward: workspace too small or not 256-byte aligned");
    const int S = Ni + Nt, Mi = B * Ni, Mt = B * Nt;
    const int64_t ld_att = d.ld_att ? d.ld_att : D;
    char* w = reinterpret_cast<char*>(workspace);
    auto take = [&](int64_t bytes) { char* p = w; w += piece(bytes); return p; };
    char* dpre = take((int64_t)Mi * 4 * D * 2);
    char* dcpre = take((int64_t)Mt * 4 * D * 2);
    char* dmid = take((int64_t)Mi * D * 2);
    char* dcmid = take((int64_t)Mt * D * 2);
    char* dx1 = take((int64_t)Mi * D * 2);
    char* dc1 = take((int64_t)Mt * D * 2);
    char* datt = take((int64_t)B * S * D * 2);
    char* dnx = take((int64_t)Mi * D * 2);
    char* dnc = take((int64_t)Mt * D * 2);
    float* work = reinterpret_cast<float*>(take((int64_t)B * H * ((S + 31) / 32) * 64 * 4));
    char *dy2 = nullptr, *datt2 = nullptr, *dnx2 = nullptr, *dqkv2 = nullptr;
    if (d.dual) { dy2 = take((int64_t)Mi * D * 2); datt2 = take((int64_t)Mi * D * 2); dnx2 = take((int64_t)Mi * D * 2); dqkv2 = take((int64_t)Mi * 3 * D * 2); }
    auto mx = [&](int j) { return bf16_at(d.mods, d.mod_x + (int64_t)j * D); };
    auto mc = [&](int j) { return bf16_at(d.mods, d.mod_c + (int64_t)j * D); };
    int rc;
    // ---- 1, 2: feed-forward data gradients (the text-stream GEMMs ride in the launches of their image-stream twins, as in the forward)
    {
        advgrpo_gemm_desc g[2] = {dgrad(d.dyg, D, d.ff2_wT, dpre, 4 * D, Mi, 4 * D, D), dgrad(d.dcyg, D, d.cff2_wT, dcpre, 4 * D, Mt, 4 * D, D)};
        g[0].act = 5; g[0].aux_in = d.pre; g[0].ld_aux = 4 * D;                  // * gelu_tanh'(pre-activation)
        g[1].act = 5; g[1].aux_in = d.cpre; g[1].ld_aux = 4 * D;
        if ((rc = advgrpo_gemm_grouped(g, d.last ? 1 : 2, stream)) != 0) return rc;
        advgrpo_gemm_desc f[2] = {dgrad(dpre, 4 * D, d.ff1_wT, dmid, D, Mi, D, 4 * D), dgrad(dcpre, 4 * D, d.cff1_wT, dcmid, D, Mt, D, 4 * D)};
        if ((rc = advgrpo_gemm_grouped(f, d.last ? 1 : 2, stream)) != 0) return rc;
    }
    // ---- 3: second norms (+ the gated copies the output-projection data gradients read)
    if ((rc = advgrpo_layernorm_mod_bwd_gated(d.x_mid, D, dmid, nullptr, D, mx(4), nullptr, d.mod_stride, Ni, d.dx, dx1, D, Mi, D, 1e-6f, mx(2), d.dyo,
                                              d.dual ? mx(8) : nullptr, d.dual ? dy2 : nullptr, d.mod_stride, stream)) != 0)
        return rc;
    if (!d.last && (rc = advgrpo_layernorm_mod_bwd_gated(d.c_mid, D, dcmid, nullptr, D, mc(4), nullptr, d.mod_stride, Nt, d.dc, dc1, D, Mt, D, 1e-6f, mc(2),
                                                          d.dyc, nullptr, nullptr, d.mod_stride, stream)) != 0)
        return rc;
    // ---- 4: second (image-only) attention of the dual blocks
    if (d.dual) {
        advgrpo_gemm_desc o = dgrad(dy2, D, d.out2_wT, datt2, D, Mi, D, D);
        if ((rc = advgrpo_gemm_grouped(&o, 1, stream)) != 0) return rc;
        const char* q2 = reinterpret_cast<const char*>(d.qkv2);
        if ((rc = advgrpo_attention_bwd(q2, q2 + (int64_t)D * 2, q2 + (int64_t)2 * D * 2, d.att2, datt2, reinterpret_cast<const float*>(d.lse2), work, dqkv2,
                                        dqkv2 + (int64_t)D * 2, dqkv2 + (int64_t)2 * D * 2, 3 * D, 3 * D, 3 * D, D, D, 3 * D, (int64_t)Ni * 3 * D,
                                        (int64_t)Ni * 3 * D, (int64_t)Ni * 3 * D, (int64_t)Ni * D, (int64_t)Ni * D, (int64_t)Ni * 3 * D, B, H, Ni, Ni, 64,
                                        0.125f, stream)) != 0)
            return rc;
        if (d.rms_2 && (rc = advgrpo_rmsnorm_heads_bwd(dqkv2, 3 * D, d.qkv2, 3 * D, reinterpret_cast<const float*>(d.rs2), Mi, 0, 2 * H, d.rms_2, H, 0, 0, 0,
                                                       stream)) != 0)
            return rc;
        advgrpo_gemm_desc q = dgrad(dqkv2, 3 * D, d.qkv2_wT, dnx2, D, Mi, D, 3 * D);
        if ((rc = advgrpo_gemm_grouped(&q, 1, stream)) != 0) return rc;
    }
    // ---- 5: joint attention
    if (d.last && hipMemsetAsync(datt, 0, (size_t)B * S * D * 2, reinterpret_cast<hipStream_t>(stream)) != hipSuccess) {   // no text-stream output projection: its rows get no gradient
        set_error("mmdit_block_backward: hipMemsetAsync failed");
        return -2;
    }
    {
        advgrpo_gemm_desc g[2] = {dgrad(d.dyo, D, d.out_wT, datt, D, Mi, D, D), dgrad(d.dyc, D, d.cout_wT, datt, D, Mt, D, D)};
        g[0].seg_rows = Ni; g[0].seg_stride = S; g[0].seg_off = 0;
        g[1].seg_rows = Nt; g[1].seg_stride = S; g[1].seg_off = Ni;
        if ((rc = advgrpo_gemm_grouped(g, d.last ? 1 : 2, stream)) != 0) return rc;
    }
    {
        const char* q = reinterpret_cast<const char*>(d.qkv);
        char* dq = reinterpret_cast<char*>(d.dqkv);
        if ((rc = advgrpo_attention_bwd(q, q + (int64_t)D * 2, q + (int64_t)2 * D * 2, d.att, datt, reinterpret_cast<const float*>(d.lse), work, dq,
                                        dq + (int64_t)D * 2, dq + (int64_t)2 * D * 2, 3 * D, 3 * D, 3 * D, ld_att, D, 3 * D, (int64_t)S * 3 * D, (int64_t)S * 3 * D,
                                        (int64_t)S * 3 * D, (int64_t)S * ld_att, (int64_t)S * D, (int64_t)S * 3 * D, B, H, S, S, 64, 0.125f, stream)) != 0)
            return rc;
        if (d.rms_x && (rc = advgrpo_rmsnorm_heads_bwd(dq, 3 * D, d.qkv, 3 * D, reinterpret_cast<const float*>(d.rs), Mi, 0, 2 * H, d.rms_x, H, Ni, S, 0, stream)) != 0)
            return rc;
        if (d.rms_c && (rc = advgrpo_rmsnorm_heads_bwd(dq, 3 * D, d.qkv, 3 * D, reinterpret_cast<const float*>(d.rs), Mt, 0, 2 * H, d.rms_c, H, Nt, S, Ni, stream)) != 0)
            return rc;
        // ---- 6: q | k | v data gradients of both streams
        advgrpo_gemm_desc g[2] = {dgrad(dq, 3 * D, d.qkv_wT, dnx, D, Mi, D, 3 * D), dgrad(dq, 3 * D, d.cqkv_wT, dnc, D, Mt, D, 3 * D)};
        g[0].a_seg_rows = Ni; g[0].a_seg_stride = S; g[0].a_seg_off = 0;
        g[1].a_seg_rows = Nt; g[1].a_seg_stride = S; g[1].a_seg_off = Ni;
        if ((rc = advgrpo_gemm_grouped(g, 2, stream)) != 0) return rc;
    }
    // ---- 7: first norms (the gated copies are for block i - 1's feed-forward output projections: gate chunk 5 of ITS modulation rows)
    const void* gxp = d.first ? nullptr : bf16_at(d.mods, d.mod_x_prev + (int64_t)5 * D);
    const void* gcp = d.first ? nullptr : bf16_at(d.mods, d.mod_c_prev + (int64_t)5 * D);
    const void* cscale = d.last ? mc(0) : mc(1);                     // the last block's text norm is AdaLayerNormContinuous: (scale, shift)
    const void* cres = d.last ? nullptr : dc1;
    if (d.first) {
        if ((rc = advgrpo_layernorm_mod_bwd(d.x_in, D, dnx, d.dual ? dnx2 : nullptr, D, mx(1), d.dual ? mx(7) : nullptr, d.mod_stride, Ni, dx1, d.dx_out, D, Mi, D,
                                            1e-6f, stream)) != 0)
            return rc;
        return advgrpo_layernorm_mod_bwd(d.c_in, D, dnc, nullptr, D, cscale, nullptr, d.mod_stride, Nt, cres, d.dc_out, D, Mt, D, 1e-6f, stream);
    }
    if ((rc = advgrpo_layernorm_mod_bwd_gated(d.x_in, D, dnx, d.dual ? dnx2 : nullptr, D, mx(1), d.dual ? mx(7) : nullptr, d.mod_stride, Ni, dx1, d.dx_out, D, Mi, D,
                                              1e-6f, gxp, d.dyg_prev, nullptr, nullptr, d.mod_stride, stream)) != 0)
        return rc;
    return advgrpo_layernorm_mod_bwd_gated(d.c_in, D, dnc, nullptr, D, cscale, nullptr, d.mod_stride, Nt, cres, d.dc_out, D, Mt, D, 1e-6f, gcp, d.dcyg_prev, nullptr,
                                           nullptr, d.mod_stride, stream);
}
