"""bench.py's epoch leg in a fresh process, optionally after the model set of the rollout leg was built / used / freed
(what bench.main does before it): where do its 104 ms micro-steps (92 in scripts/probes/gstep_in_situ.py) come from?
Usage: epoch_leg_alone.py [none|build|build+step]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
if mode != "none":
    pipe, clip = bench.build(device)
    if mode == "build+step":
        from adv_grpo_amd import synthetic
        from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
        pe, ppe, npe, nppe = (t.to(device=device, dtype=torch.bfloat16) for t in synthetic.prompt_embeddings(7))
        for it in range(2):
            pipeline_with_logprob_random(pipe, prompt_embeds=pe, pooled_prompt_embeds=ppe, negative_prompt_embeds=npe,
                                         negative_pooled_prompt_embeds=nppe, num_inference_steps=10, guidance_scale=4.5, height=512, width=512,
                                         noise_level=0.8, mini_num_image_per_prompt=8, train_num_steps=2, process_index=0, sample_num_steps=10,
                                         random_timestep=0, seed=it)
        torch.cuda.synchronize()
    del pipe, clip
    torch.cuda.empty_cache()
if os.environ.get("NO_OVERLAP"):
    from adv_grpo_amd import mmdit_train
    _init = mmdit_train.SD3TransformerLoRA.__init__
    def patched(self, *a, **k):
        _init(self, *a, **k)
        self.overlap_wgrad = False
    mmdit_train.SD3TransformerLoRA.__init__ = patched
ep = bench.full_epoch(device)
print(mode, ep["phases_s"], ep["g_step_inside"]["micro_step"])
