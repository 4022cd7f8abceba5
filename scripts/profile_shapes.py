"""Per-shape GEMM timing inside one real rollout step (HIP events around every launch)."""
import sys, torch, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from adv_grpo_amd import ops, synthetic, vit
from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
dev = torch.device("cuda")
pipe, clip = bench.build(dev)
pe, ppe, npe, nppe = (t.to(device=dev, dtype=torch.bfloat16) for t in synthetic.prompt_embeddings(7))
ids = synthetic.clip_input_ids(8, 3).to(dev)
def step(seed):
    image, lats, lps, tss = pipeline_with_logprob_random(pipe, prompt_embeds=pe, pooled_prompt_embeds=ppe, negative_prompt_embeds=npe,
        negative_pooled_prompt_embeds=nppe, num_inference_steps=10, guidance_scale=4.5, output_type="pt", height=512, width=512,
        noise_level=0.8, mini_num_image_per_prompt=8, train_num_steps=2, process_index=0, sample_num_steps=10, random_timestep=0, seed=seed)
    return vit.pickscore_scores(clip.get_image_features(images=image.to(torch.bfloat16)), clip.get_text_features(ids), clip.logit_scale)
step(0)
ops.PROFILE = []; ops.PROFILE_STRIDE = 1
torch.cuda.synchronize(); step(1); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, fl, s, e, shape in ops.PROFILE:
    a = agg[(name, shape)]; a[0] += 1; a[1] += fl; a[2] += s.elapsed_time(e) * 1e-3
tot = sum(a[2] for a in agg.values())
print(f"total GEMM time {tot*1e3:.1f} ms")
for (tile, shape), (n, fl, t) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:30]:
    print(f"{tile[:34]:34s} M={shape[0]:7d} N={shape[1]:5d} K={shape[2]:5d} b={shape[3]} conv={shape[4]}  n={n:4d}  {t*1e3:7.2f} ms  {fl/t/1e12:7.1f} TF  {100*t/tot:5.1f}%")
