#!/usr/bin/env python3
"""One command that PINS the three "parity unpinned" oracle restatements (oracle/mmdit.py, oracle/vae.py,
oracle/scheduler.py) against the real third-party code -- for a machine where `diffusers` (0.33.1, the reference's pin,
setup.py:8-52) is installed.  Neither diffusers nor any checkpoint exists in the build image or on the GPU box, so this
script cannot run there; it is the hand-off DESIGN.md section 4 and INTEGRATION.md point to.

    python scripts/verify_against_diffusers.py [--model stabilityai/stable-diffusion-3.5-medium] [--device cuda] [--random]

--random  builds diffusers' modules from their configs with seeded random weights (no download): pins the ARCHITECTURE
          restatement (adaLN chunk orders, dual-attention blocks, position-embedding crop, context_pre_only last block,
          VAE block order, scheduler shift) -- which is everything the oracle restates.
default   loads the checkpoint's transformer / vae / scheduler and compares on its real weights.

Checks (fp32, same inputs):
  * FlowMatchEulerDiscreteScheduler: sigmas / timesteps for 4, 10, 40 steps, index_for_timestep     -> equal to 1e-6
  * SD3Transformer2DModel forward (CFG pair, 64x64 latents, 205 text tokens)                           -> rel L2 < 1e-4
  * AutoencoderKL.decode (1 x 16 x 32 x 32 latents) + VaeImageProcessor.postprocess("pt")              -> max abs < 1e-4
The oracle takes diffusers' own state_dict (weights are keyed by diffusers names on purpose), so nothing is converted.
  * QwenImageTransformer2DModel forward (diffusers >= 0.35; reduced depth with --random; non-square latents, two prompts of      -> rel L2 < 1e-4
    different lengths through the pipeline's padding mask) vs oracle/qwen_mmdit.py
  * AutoencoderKLQwenImage.decode on one frame vs oracle/qwen_vae.py                                    -> max abs < 1e-4
  * peft: an adapter written by `get_peft_model(...).save_pretrained` read by checkpoint.load_lora, and one written by
    checkpoint.save_lora read by `PeftModel.from_pretrained`                                            -> bit-identical tensors
Exit status 0 = the restatements are pinned on this machine (non-zero on any mismatch; sections whose package is missing are reported
and skipped); the printed lines are what to paste into DESIGN.md section 4.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="stabilityai/stable-diffusion-3.5-medium")
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--random", action="store_true")
    a = ap.parse_args()
    try:
        import diffusers
        from diffusers import AutoencoderKL, FlowMatchEulerDiscreteScheduler, SD3Transformer2DModel
        from diffusers.image_processor import VaeImageProcessor
    except ImportError:
        print("diffusers is not installed: the oracle restatements stay 'parity unpinned' on this machine "
              "(pip install diffusers==0.33.1 where there is network)")
        return 2
    from oracle import mmdit as o_m
    from oracle import vae as o_v
    from oracle.scheduler import FlowMatchEulerScheduler
    dev = a.device
    print(f"diffusers {diffusers.__version__}, torch {torch.__version__}, device {dev}, weights: {'seeded random' if a.random else a.model}")
    ok = True

    # ---------------------------------------------------------------- scheduler (PF:574, SDE:106-110)
    if a.random:
        ref_s = FlowMatchEulerDiscreteScheduler(shift=3.0)
    else:
        ref_s = FlowMatchEulerDiscreteScheduler.from_pretrained(a.model, subfolder="scheduler")
    for steps in (4, 10, 40):
        ref_s.set_timesteps(steps)
        mine = FlowMatchEulerScheduler(shift=float(ref_s.config.shift))
        mine.set_timesteps(steps)
        ds = (mine.sigmas.double().cpu() - ref_s.sigmas.double().cpu()).abs().max().item()
        dt = (mine.timesteps.double().cpu() - ref_s.timesteps.double().cpu()).abs().max().item()
        idx_ok = all(mine.index_for_timestep(t) == ref_s.index_for_timestep(t) for t in ref_s.timesteps)
        print(f"scheduler {steps:2d} steps: max |d sigma| {ds:.2e}  max |d t| {dt:.2e}  index_for_timestep {'ok' if idx_ok else 'DIFFERS'}")
        ok &= ds < 1e-6 and dt < 1e-3 and idx_ok

    # ---------------------------------------------------------------- transformer (PF:630-637, TP:235-255)
    torch.manual_seed(0)
    if a.random:
        cfg = o_m.MMDiTConfig()
        tr = SD3Transformer2DModel(sample_size=128, patch_size=2, in_channels=16, num_layers=cfg.num_layers, attention_head_dim=64,
                                   num_attention_heads=cfg.num_heads, joint_attention_dim=4096, caption_projection_dim=cfg.dim,
                                   pooled_projection_dim=2048, out_channels=16, pos_embed_max_size=cfg.pos_embed_max_size,
                                   dual_attention_layers=tuple(cfg.dual_attention_layers), qk_norm="rms_norm")
        for p_ in tr.parameters():
            torch.nn.init.normal_(p_, std=0.02)
    else:
        tr = SD3Transformer2DModel.from_pretrained(a.model, subfolder="transformer", torch_dtype=torch.float32)
        c = tr.config
        cfg = o_m.MMDiTConfig(num_layers=c.num_layers, num_heads=c.num_attention_heads, joint_attention_dim=c.joint_attention_dim,
                              pooled_projection_dim=c.pooled_projection_dim, pos_embed_max_size=c.pos_embed_max_size,
                              dual_attention_layers=tuple(c.dual_attention_layers))
    tr = tr.to(dev).float().eval()
    W = {k: v.detach() for k, v in tr.state_dict().items()}
    x = torch.randn(2, 16, 64, 64, device=dev)
    t = torch.tensor([913.35, 913.35], device=dev)
    ctx = torch.randn(2, 205, cfg.joint_attention_dim, device=dev)
    pooled = torch.randn(2, cfg.pooled_projection_dim, device=dev)
    with torch.no_grad():
        ref = tr(hidden_states=x, timestep=t, encoder_hidden_states=ctx, pooled_projections=pooled, return_dict=False)[0]
        out = o_m.mmdit_forward(W, cfg, x, t, ctx, pooled)
    rel = ((out - ref).norm() / ref.norm()).item()
    print(f"SD3Transformer2DModel ({cfg.num_layers} blocks, D={cfg.dim}): rel L2 {rel:.2e}")
    ok &= rel < 1e-4
    del tr, W

    # ---------------------------------------------------------------- VAE decoder + postprocess (PF:667-670)
    if a.random:
        vae = AutoencoderKL(in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512),
                            down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4, layers_per_block=2,
                            norm_num_groups=32, use_quant_conv=False, use_post_quant_conv=False, scaling_factor=1.5305, shift_factor=0.0609)
    else:
        vae = AutoencoderKL.from_pretrained(a.model, subfolder="vae", torch_dtype=torch.float32)
    vae = vae.to(dev).float().eval()
    vcfg = o_v.VaeConfig(scaling_factor=float(vae.config.scaling_factor), shift_factor=float(vae.config.shift_factor))
    Wv = {k: v.detach() for k, v in vae.state_dict().items() if k.startswith("decoder.")}      # oracle/vae.py reads "decoder.*"
    lat = torch.randn(1, 16, 32, 32, device=dev)
    with torch.no_grad():
        z = lat / vcfg.scaling_factor + vcfg.shift_factor
        ref_img = VaeImageProcessor(vae_scale_factor=8).postprocess(vae.decode(z, return_dict=False)[0], output_type="pt")
        img = o_v.postprocess(o_v.vae_decode(Wv, vcfg, z))
    dmax = (img - ref_img).abs().max().item()
    print(f"AutoencoderKL.decode + postprocess('pt'): max abs {dmax:.2e}")
    ok &= dmax < 1e-4
    # ---------------------------------------------------------------- Qwen-Image VAE decoder (BASELINE config 5; diffusers >= 0.35)
    try:
        from diffusers import AutoencoderKLQwenImage
    except ImportError:
        AutoencoderKLQwenImage = None
        print("AutoencoderKLQwenImage: not in this diffusers (needs >= 0.35) -- oracle/qwen_vae.py stays unpinned here")
    if AutoencoderKLQwenImage is not None:
        from oracle import qwen_vae as o_q
        qv = (AutoencoderKLQwenImage() if a.random else
              AutoencoderKLQwenImage.from_pretrained("Qwen/Qwen-Image", subfolder="vae", torch_dtype=torch.float32)).to(dev).float().eval()
        if a.random:
            torch.manual_seed(5)
            with torch.no_grad():
                for prm in qv.parameters():
                    prm.copy_(torch.randn_like(prm) * 0.05 + (1.0 if prm.dim() > 1 and prm.numel() == prm.shape[0] else 0.0))   # gammas near 1
        qcfg = o_q.QwenVaeConfig(latents_mean=tuple(qv.config.latents_mean), latents_std=tuple(qv.config.latents_std))
        Wq = {k: v.detach() for k, v in qv.state_dict().items() if k.startswith(("decoder.", "post_quant_conv.")) and "time_conv" not in k}
        latq = torch.randn(1, 16, 24, 24, device=dev)
        with torch.no_grad():
            zq = o_q.denormalise(qcfg, latq[:, :, None])
            ref_q = VaeImageProcessor(vae_scale_factor=8).postprocess(qv.decode(zq, return_dict=False)[0][:, :, 0], output_type="pt")
            img_q = o_q.decode_to_image(Wq, qcfg, latq)
        dq = (img_q - ref_q).abs().max().item()
        print(f"AutoencoderKLQwenImage.decode (one frame) + postprocess('pt'): max abs {dq:.2e}")
        ok &= dq < 1e-4
    # ---------------------------------------------------------------- Qwen-Image transformer (BASELINE config 5; diffusers >= 0.35)
    try:
        from diffusers import QwenImageTransformer2DModel
    except ImportError:
        QwenImageTransformer2DModel = None
        print("QwenImageTransformer2DModel: not in this diffusers (needs >= 0.35) -- oracle/qwen_mmdit.py stays unpinned here")
    if QwenImageTransformer2DModel is not None:
        from oracle import qwen_mmdit as o_qm
        torch.manual_seed(7)
        if a.random:
            # the released config with a reduced depth (the blocks are identical; 60 of them in fp32 are 82 GB): pins the chunk orders of
            # img_mod / txt_mod, the rotary layout (scale_rope index ranges, text offset), txt_norm, norm_out's (scale, shift) order, the
            # proj_out / unpack permutation
            qcfg = o_qm.QwenMMDiTConfig(num_layers=4)
            qt = QwenImageTransformer2DModel(patch_size=2, in_channels=64, out_channels=16, num_layers=qcfg.num_layers, attention_head_dim=128,
                                             num_attention_heads=24, joint_attention_dim=3584, guidance_embeds=False, axes_dims_rope=(16, 56, 56))
            with torch.no_grad():
                for n_, p_ in qt.named_parameters():
                    if p_.dim() > 1:
                        torch.nn.init.normal_(p_, std=0.02)
                    elif "norm" in n_ and n_.endswith("weight"):
                        p_.copy_(1 + 0.1 * torch.randn_like(p_))
                    else:
                        torch.nn.init.normal_(p_, std=0.1)
        else:
            qt = QwenImageTransformer2DModel.from_pretrained("Qwen/Qwen-Image", subfolder="transformer", torch_dtype=torch.float32)
            c = qt.config
            qcfg = o_qm.QwenMMDiTConfig(num_layers=c.num_layers, num_heads=c.num_attention_heads, head_dim=c.attention_head_dim,
                                        joint_attention_dim=c.joint_attention_dim, axes_dims_rope=tuple(c.axes_dims_rope))
        qt = qt.to(dev).float().eval()
        Wq = {k: v.detach() for k, v in qt.state_dict().items()}
        hq, wq = 48, 32                                             # latent height / width: 24 x 16 packed positions (not square on purpose)
        lat = torch.randn(2, 16, hq, wq, device=dev)
        # ragged text lengths as the pipeline produces them: padded to the longest, masked (the restatement has no mask: the shorter
        # prompt is compared on its own, unpadded, which is what the mask makes of it)
        lens = [37, 23]
        ctx = torch.randn(2, max(lens), 3584, device=dev)
        mask = torch.zeros(2, max(lens), dtype=torch.long, device=dev)
        for i, L in enumerate(lens):
            mask[i, :L] = 1
        sigma = torch.tensor([0.9133, 0.9133], device=dev)
        tok = o_qm.pack_latents(lat)
        with torch.no_grad():
            ref = qt(hidden_states=tok, timestep=sigma, encoder_hidden_states=ctx, encoder_hidden_states_mask=mask,
                     img_shapes=[[(1, hq // 2, wq // 2)]] * 2, txt_seq_lens=lens, return_dict=False)[0]
            ref = o_qm.unpack_latents(ref, hq, wq)
            out_full = o_qm.qwen_forward(Wq, qcfg, lat[:1], sigma[:1], ctx[:1, :lens[0]])
            out_short = o_qm.qwen_forward(Wq, qcfg, lat[1:], sigma[1:], ctx[1:, :lens[1]])
        r0 = ((out_full - ref[:1]).norm() / ref[:1].norm()).item()
        r1 = ((out_short - ref[1:]).norm() / ref[1:].norm()).item()
        print(f"QwenImageTransformer2DModel ({qcfg.num_layers} blocks, D={qcfg.dim}, {hq // 2}x{wq // 2} positions, text {lens}): rel L2 {r0:.2e} "
              f"(longest prompt), {r1:.2e} (shorter prompt: padded + masked there, unpadded here)")
        ok &= r0 < 1e-4 and r1 < 1e-4
        del qt, Wq

    # ---------------------------------------------------------------- PEFT's adapter files (TP:389-398 save_ckpt, TP:506-509 load)
    try:
        import peft
    except ImportError:
        peft = None
        print("peft is not installed: adv_grpo_amd/checkpoint.py's adapter_model.safetensors / adapter_config.json layout stays unpinned here")
    if peft is not None:
        import json
        import tempfile
        from adv_grpo_amd import checkpoint
        torch.manual_seed(11)
        tiny = SD3Transformer2DModel(sample_size=32, patch_size=2, in_channels=16, num_layers=2, attention_head_dim=64, num_attention_heads=2,
                                     joint_attention_dim=64, caption_projection_dim=128, pooled_projection_dim=32, out_channels=16,
                                     pos_embed_max_size=16, dual_attention_layers=(0,), qk_norm="rms_norm")
        targets = ["attn.add_k_proj", "attn.add_q_proj", "attn.add_v_proj", "attn.to_add_out", "attn.to_k", "attn.to_out.0", "attn.to_q", "attn.to_v"]
        pm = peft.get_peft_model(tiny, peft.LoraConfig(r=32, lora_alpha=64, init_lora_weights="gaussian", target_modules=targets))     # TP:490-505
        with torch.no_grad():
            for n_, p_ in pm.named_parameters():
                if "lora_B" in n_:
                    p_.copy_(torch.randn_like(p_) * 0.02)
        with tempfile.TemporaryDirectory() as d:
            pm.save_pretrained(d)                                                                                             # TP:389-398
            state, cfg_json = checkpoint.load_lora(d)
            want = {k.replace("base_model.model.", "").replace(".default", ""): v for k, v in pm.state_dict().items() if "lora_" in k}
            same_keys = set(state) == set(want)
            same_vals = same_keys and all(torch.equal(state[k].float().cpu(), want[k].float().cpu()) for k in want)
            print(f"peft {peft.__version__}: adapter written by save_pretrained -> checkpoint.load_lora: {len(state)} tensors, keys {'match' if same_keys else 'DIFFER'}, "
                  f"values {'bit-identical' if same_vals else 'DIFFER'}; r / alpha {cfg_json.get('r')} / {cfg_json.get('lora_alpha')}")
            ok &= same_vals and cfg_json.get("r") == 32 and cfg_json.get("lora_alpha") == 64
            # and the other direction: a directory written by checkpoint.save_lora loads into PEFT
            d2 = os.path.join(d, "ours")
            checkpoint.save_lora(d2, {k: v.clone() for k, v in want.items()}, r=32, lora_alpha=64)
            tiny2 = SD3Transformer2DModel(**{k: v for k, v in tiny.config.items() if not k.startswith("_")})
            pm2 = peft.PeftModel.from_pretrained(tiny2, d2)
            got = {k.replace("base_model.model.", "").replace(".default", ""): v for k, v in pm2.state_dict().items() if "lora_" in k}
            back = set(got) == set(want) and all(torch.equal(got[k].float().cpu(), want[k].float().cpu()) for k in want)
            print(f"checkpoint.save_lora -> peft.PeftModel.from_pretrained: {'bit-identical' if back else 'DIFFERS'}")
            ok &= back
    print("PINNED: every oracle restatement checked above reproduces the installed third-party code on this machine" if ok else "MISMATCH: see the lines above")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
