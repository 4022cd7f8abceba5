"""Is the host ahead of the GPU in a rollout step?  Time for the step() call to RETURN (launches enqueued) against the time until
the GPU has finished it.  A hidden device->host synchronisation inside the step shows up as the two being equal."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from adv_grpo_amd import synthetic, vit
from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
dev = torch.device("cuda", 0)
pipe, clip = bench.build(dev)
pe, ppe, npe, nppe = (t.to(device=dev, dtype=torch.bfloat16) for t in synthetic.prompt_embeddings(7))
ids = synthetic.clip_input_ids(8, 3).to(dev)
def step(it):
    image, lats, lps, tss = pipeline_with_logprob_random(
        pipe, prompt_embeds=pe, pooled_prompt_embeds=ppe, negative_prompt_embeds=npe, negative_pooled_prompt_embeds=nppe,
        num_inference_steps=10, guidance_scale=4.5, output_type="pt", height=512, width=512, noise_level=0.8,
        mini_num_image_per_prompt=8, train_num_steps=2, process_index=0, sample_num_steps=10, random_timestep=0, seed=1234 + it)
    return vit.pickscore_scores(clip.get_image_features(images=image.to(torch.bfloat16)), clip.get_text_features(ids), clip.logit_scale)
for it in range(2):
    step(it)
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter(); step(10 + it); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"step call returned after {1e3 * (t1 - t0):7.1f} ms, GPU done after {1e3 * (t2 - t0):7.1f} ms")
# where the synchronisations are (torch ops only; the ctypes launches never synchronise)
import warnings
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step(99)
torch.cuda.set_sync_debug_mode("default")
import collections
c = collections.Counter((str(x.filename).split("/")[-1], x.lineno) for x in w if "synchroniz" in str(x.message).lower())
print("synchronising torch calls in one step:", dict(c))
