"""Same-process A/B of two builds of the library on the config-2 rollout (10 denoise steps of the CFG batch 16 + the VAE decode of 8 images; no
scorer): the pipeline is built once, `_lib._lib` is swapped between the builds, R rounds of N rollouts each, alternating order.  Repeats to ~0.1 %
(alternating PROCESSES carries a position effect of 1 - 3 %, LABNOTES.md 7).   Usage: rollout_ab_inprocess.py [base.so [new.so [N [R]]]]"""
import ctypes, os, sys, time, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import bench
from adv_grpo_amd import _lib, synthetic
from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
paths = [sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "adv_grpo_amd", "libadvgrpo_base.so"),
         sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "adv_grpo_amd", "libadvgrpo_hip.so")]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 3
R = int(sys.argv[4]) if len(sys.argv) > 4 else 5
libs = []
for p in paths:
    lib = ctypes.CDLL(p)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    libs.append(lib)
_lib._lib = libs[1]
device = torch.device("cuda:0")
pipe, _clip = bench.build(device)
del _clip
pe, ppe, npe, nppe = (t.to(device=device, dtype=torch.bfloat16) for t in synthetic.prompt_embeddings(7))


def rollouts(n, first):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(n):
        image, lats, lps, tss = pipeline_with_logprob_random(
            pipe, prompt_embeds=pe, pooled_prompt_embeds=ppe, negative_prompt_embeds=npe, negative_pooled_prompt_embeds=nppe,
            num_inference_steps=10, guidance_scale=4.5, height=512, width=512, noise_level=0.8, mini_num_image_per_prompt=8,
            train_num_steps=2, process_index=0, sample_num_steps=10, random_timestep=0, seed=1000 + first + it)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, image, torch.stack(lps)


rollouts(2, 0)
ts, outs = [[], []], [None, None]
for r in range(R):
    for i in ((0, 1) if r % 2 == 0 else (1, 0)):
        _lib._lib = libs[i]
        rollouts(1, 0)
        t, img, lp = rollouts(N, 0)
        ts[i].append(t)
        outs[i] = (img, lp)
same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
for i in (0, 1):
    t = sorted(ts[i])
    print(f"{os.path.basename(paths[i])}: median {t[len(t) // 2]:.2f} ms per rollout + decode   [{' '.join(f'{x:.2f}' for x in ts[i])}]")
print("images and log-probs of the two builds bit-identical:", same)
