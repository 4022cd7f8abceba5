"""The Adv-GRPO epoch loop on the gfx950 kernels: sample -> score -> gather -> advantage -> D-step | G-step.

Host loop standing in for ``main()`` of scripts/train_sd3_fast_{pickscore,dino_patch}.py (TP:709-1191, TD identical
unless noted); one process per GPU, torch.distributed (RCCL) only at the path's exchange steps:
  * packed all-gather of rewards + group ids once per epoch                       (TP:926-966)
  * all-reduce of the flat LoRA gradient vector once per optimizer step           (TP:1165, DeepSpeed/DDP)
  * all-reduce of the discriminator-head gradient vector once per D-step          (TD:749 DDP(head))
The D/G gate is computed from all-gathered data, so every rank takes the same branch (TP:1008-1025, TD:1091-1097).

Data enters through a ``DataProvider`` (prompt embeddings, CLIP ids and per-prompt reference images by dataset
index): the text encoders and the reference-image files (f4) are outside the accelerated path, so the
default provider is synthetic and seeded by the dataset index.
"""
import json
import os
import time

import torch

from . import distributed as D
from . import g_step, ops, rewards, stat_tracking
from .d_step import train_dino
from .d_step_pickscore import ClipLastLayerTrainable, ClipLayersTrainable, train_pickscore
from .diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
from .sampler import DistributedKRepeatSampler


def rollout_seed(config_seed, iteration, rank):
    """Philox key of one rollout call: a 64-bit mix (splitmix64 finaliser) of (config.seed, sampler iteration, rank), so
    that no two (iteration, rank) pairs share or neighbour a key whatever the world size -- the reference gets the same
    independence from set_seed(config.seed, device_specific=True) and each rank's own generator state (TP:444)."""
    m = (1 << 64) - 1
    x = ((int(config_seed) & 0xFFFFFFFF) << 32) ^ ((int(rank) & 0xFFFF) << 16) ^ ((int(iteration) * 0x9E3779B97F4A7C15) & m)
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & m
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & m
    x ^= x >> 31
    return x >> 1            # 63 bits: stays a positive int64 through ctypes / torch


class SyntheticData:
    """Seeded stand-in for the prompt dataset + text encoders + reference-image store."""

    def __init__(self, n_prompts=25432, n_tokens=205, ctx_dim=4096, pooled_dim=2048, resolution=512, device="cuda",
                 dtype=torch.bfloat16):
        self.n, self.nt, self.cd, self.pd, self.res = n_prompts, n_tokens, ctx_dim, pooled_dim, resolution
        self.device, self.dtype = device, dtype
        g = torch.Generator().manual_seed(99)
        self.neg = (torch.randn(1, n_tokens, ctx_dim, generator=g).to(dtype).to(device),
                    torch.randn(1, pooled_dim, generator=g).to(dtype).to(device))

    def __len__(self):
        return self.n

    def prompt(self, idx):
        g = torch.Generator().manual_seed(1_000_003 * idx + 7)
        return (torch.randn(1, self.nt, self.cd, generator=g).to(self.dtype).to(self.device),
                torch.randn(1, self.pd, generator=g).to(self.dtype).to(self.device))

    def clip_ids(self, idx, n):
        from .synthetic import clip_input_ids
        return clip_input_ids(1, seed=idx).repeat(n, 1).to(self.device)

    def prompt_text(self, idx):
        """A stand-in prompt string in the shape of dataset/ocr (a quoted target text, adv_grpo/ocr.py:32)."""
        return f'a sign that says "prompt {idx}" on a wall'

    def reference_images(self, idx, n):
        g = torch.Generator().manual_seed(2_000_003 * idx + 11)
        return torch.rand(n, 3, self.res, self.res, generator=g).to(self.device)


class PromptFileData:
    """The real data path with the interface of SyntheticData: prompts from ``<dataset>/<split>.txt`` (TextPromptDataset,
    TP:50-66), reference images from the ``json_path`` map + ``reference_image_path`` directory (TP:705-707,773-801,
    decoded ahead of time by reference_images.ReferenceImageStore), embeddings from a caller-supplied
    ``embed_fn(list[str]) -> (prompt_embeds [B,205,4096], pooled [B,2048])`` (tokenizers + text_encoders.encode_prompt)
    and ``clip_ids_fn(list[str]) -> ids [B,77]`` for the PickScore text tower.  Embeddings are cached per prompt."""

    def __init__(self, dataset_dir, json_path, reference_image_path, embed_fn, clip_ids_fn, split="train", resolution=512,
                 device="cuda", fallback_image=None):
        from .reference_images import ReferenceImageStore
        with open(os.path.join(dataset_dir, f"{split}.txt"), "r") as f:
            self.prompts = [line.strip() for line in f.readlines()]
        self.store = ReferenceImageStore(json_path, reference_image_path, resolution, device, fallback_image)
        self.embed_fn, self.clip_ids_fn, self.device = embed_fn, clip_ids_fn, device
        self._cache = {}
        self.neg = embed_fn([""])                                                          # TP:669 / TP:272

    def __len__(self):
        return len(self.prompts)

    def prompt(self, idx):
        if idx not in self._cache:
            self._cache[idx] = self.embed_fn([self.prompts[idx]])
        return self._cache[idx]

    def clip_ids(self, idx, n):
        return self.clip_ids_fn([self.prompts[idx]]).repeat(n, 1).to(self.device)

    def prompt_text(self, idx):
        return self.prompts[idx]

    def prefetch(self, idxs):
        self.store.prefetch([self.prompts[i] for i in idxs])

    def reference_images(self, idx, n):
        return self.store.get(self.prompts[idx], n)


class JsonlLogger:
    """wandb stand-in: one JSON object per log call (same metric names as TP:941-955,975-988,1132-1183)."""

    def __init__(self, path, enabled=True):
        self.path, self.enabled = path, enabled
        if enabled and path:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)

    def log(self, metrics, step):
        if not self.enabled or not self.path:
            return
        rec = {"step": step, "time": time.time()}
        rec.update({k: (v.item() if hasattr(v, "item") else v) for k, v in metrics.items()})
        with open(self.path, "a") as f:
            f.write(json.dumps(rec) + "\n")


def write_eval_images(images, folder, rank, batch_idx, node_id=0, size=512):
    """scripts/eval.py:258-266: [B,3,H,W] in [0,1] -> uint8 by truncation of x*255 (numpy astype) -> PIL resize to
    512x512 (PIL's default filter) -> node<n>_rank<r>_<batch:05d>_<i>.png.  Returns the file names."""
    from PIL import Image
    import numpy as np
    arr = (images.float().permute(0, 2, 3, 1).cpu().numpy() * 255).astype(np.uint8)
    names = []
    for i, a in enumerate(arr):
        name = f"node{node_id}_rank{rank}_{batch_idx:05d}_{i}.png"
        Image.fromarray(a).resize((size, size)).save(os.path.join(folder, name))
        names.append(name)
    return names


def write_prompt2img(local, folder, world=1, rank=0):
    """gather_dict + the rank-0 dump (scripts/eval.py:153-165,291-294): per-rank {prompt: [files]} maps merged in rank
    order (lists of a prompt seen on several ranks concatenate), written as prompt2img.json (indent 2, non-ASCII kept)."""
    gathered = [local]
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
    merged = {}
    for d in gathered:
        for k, v in (d or {}).items():
            merged.setdefault(k, []).extend(v)
    if rank == 0:
        with open(os.path.join(folder, "prompt2img.json"), "w", encoding="utf-8") as f:
            json.dump(merged, f, indent=2, ensure_ascii=False)
    return merged


class Trainer:
    def __init__(self, config, pipeline, data, scorer, head=None, rank=0, world=1, log_path=None):
        """pipeline.transformer: SD3TransformerLoRA; scorer: PickScoreScorer (pickscore variants) or vit.DinoV2 (DINO
        variants) with ``head`` a d_step.DinoHeadTrainable."""
        self.cfg, self.pipe, self.data, self.scorer, self.head = config, pipeline, data, scorer, head
        self.rank, self.world = rank, world
        self.device = pipeline.device
        c = config
        self.variant = "dino" if any(k.startswith("dino") for k in c.reward_fn.keys()) else "pickscore"
        self.reward_key = next(iter(c.reward_fn.keys()))
        self.reward_fn = rewards.multi_score(self.device, c.reward_fn.to_dict())
        k = c.sample.num_image_per_prompt // c.sample.mini_num_image_per_prompt          # TP:577
        self.sampler = DistributedKRepeatSampler(range(len(data)), c.sample.train_batch_size, k, world, rank, seed=c.seed)
        self.stat_tracker = stat_tracking.PerPromptStatTracker(c.sample.global_std, device=self.device)
        if c.get("linear_dtype", "bf16") == "fp8":       # BASELINE config 5's fp8 MFMA path (mmdit.enable_fp8): rollout and replay alike
            pipeline.transformer.enable_fp8()
        self.clip_trainable = None
        if self.variant == "pickscore" and c.get("train_d", False):
            if not isinstance(c.tune_layer, int) or c.tune_layer >= 0:
                raise NotImplementedError(f"tune_layer = {c.tune_layer!r}: TP:1016-1020 slices encoder.layers[tune_layer:]; a negative layer "
                                          "count is what that supports (the tuple values sit in DINO configs that never read the key)")
            if getattr(scorer, "compute_dtype", "bf16") != "bf16":
                raise ValueError("the co-trained PickScore scorer is the bf16 one (TP:514): build it with "
                                 "PickScoreScorer(dtype=torch.bfloat16, ...); dtype=torch.float32 is the frozen reward scorer")
            self.clip_trainable = (ClipLastLayerTrainable(scorer.model) if c.tune_layer == -1 else
                                   ClipLayersTrainable(scorer.model, c.tune_layer))       # TP:1016-1020
        if world > 1:
            # every trainable state starts identical on all ranks: rank 0's values are broadcast, as DDP / DeepSpeed do
            # at construction (TD:749, TP:554-561); afterwards only all-reduced gradients change them
            D.broadcast_state(self._trainable_state())
            if hasattr(pipeline.transformer, "refresh"):
                pipeline.transformer.ema.copy_(pipeline.transformer.params)
                pipeline.transformer.refresh()
            for obj in (head, self.clip_trainable):            # bf16 shadows / installed weights follow the broadcast masters
                if obj is not None and hasattr(obj, "p16"):
                    obj.p16.copy_(obj.params)
                if obj is not None and hasattr(obj, "sync_model"):
                    obj.sync_model()
        # reference rewards only feed the PickScore D/G gate and the D-steps (TP:1008-1037, TD:1091-1097)
        self.needs_reference = bool(c.get("train_d", False)) or self.variant == "dino"
        self.async_reward = bool(c.get("async_reward", True))
        self.profile_phases = True
        if self.async_reward:
            from concurrent.futures import ThreadPoolExecutor
            self._score_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="advgrpo-score")   # TP:668
            d = torch.device(self.device)
            self._dev_index = d.index if d.index is not None else torch.cuda.current_device()
            self._score_stream = torch.cuda.Stream(device=self._dev_index)
        # side streams that are chosen by measurement are chosen NOW, before a worker thread or a rollout exists (measured
        # lazily on the scoring worker, the device-wide synchronisations of the probe waited for every thread's kernels)
        if hasattr(scorer, "prepare_streams") and torch.cuda.is_available():
            scorer.prepare_streams(([self._score_stream] if self.async_reward else []) + [torch.cuda.current_stream(torch.device(self.device))])
        # the rollout workers' streams (config sample.groups_in_flight, default 2) exist from construction as well, and the decoder's side stream for
        # decodes on the launch stream (one group at a time, evaluate()) is one of them: they are idle whenever such a decode runs, and a fifth
        # live stream beside them cost the in-flight schedule 2.5 % (vae.py: set_side_streams)
        self._rollout_pool = None
        if torch.cuda.is_available() and torch.device(self.device).type == "cuda":
            self._make_rollout_streams(int(c.sample.get("groups_in_flight", 2)))
            vae = self.pipe.vae
            if hasattr(vae, "prepare_streams") and getattr(vae, "mode", None) == "bf16x3":
                main = torch.cuda.current_stream(torch.device(self.device))
                if self._rollout_pool is not None:
                    vae.set_side_streams(main, self._rollout_stream_list)
                else:
                    vae.prepare_streams([main], also=[self._score_stream] if self.async_reward else [])
            # ... and so is the G-step's adapter-gradient side stream (no group is in flight during the update half).  The model measured one of its
            # own when it was built, but a stream measured BEFORE other streams came to life does not stay concurrent with the launch stream
            # (LABNOTES 6 round 5; round 6: 110 instead of 87 ms per micro-step on the stream the model had chosen before this constructor ran)
            tr = self.pipe.transformer
            if getattr(tr, "_wgrad_stream", None) is not None:
                tr._wgrad_stream = self._rollout_stream_list[-1] if self._rollout_pool is not None else \
                    ops.concurrent_stream(torch.device(self.device), [torch.cuda.current_stream(torch.device(self.device))])
        self.epoch, self.global_step = 0, 0
        self.logger = JsonlLogger(log_path, enabled=(rank == 0))
        self.timers = {}

    def _make_rollout_streams(self, in_flight):
        """One worker thread + one HIP stream per prompt group in flight; the streams are chosen by measurement (ops.concurrent_stream: concurrent
        with the launch stream, with each other and with the scoring stream), which synchronises the device: construction time, or the first
        sample_epoch after groups_in_flight changed."""
        if in_flight < 2:
            return
        import threading
        from concurrent.futures import ThreadPoolExecutor
        if self._rollout_pool is not None:
            self._rollout_pool.shutdown(wait=True)
        self._rollout_pool = ThreadPoolExecutor(max_workers=in_flight, thread_name_prefix="advgrpo-rollout")
        d = torch.device(self.device)
        self._rollout_dev = d.index if d.index is not None else torch.cuda.current_device()
        self._rollout_streams = []
        for _ in range(in_flight):
            partners = [torch.cuda.current_stream(d)] + ([self._score_stream] if self.async_reward else []) + list(self._rollout_streams)
            self._rollout_streams.append(ops.concurrent_stream(d, partners))
        self._rollout_stream_list = list(self._rollout_streams)      # (the workers pop theirs from _rollout_streams)
        self._rollout_tls, self._rollout_lock = threading.local(), threading.Lock()

    def _trainable_state(self):
        """Flat f32 parameter vectors that training changes (LoRA, discriminator head / last CLIP layer)."""
        out = []
        tr = self.pipe.transformer
        if hasattr(tr, "params"):
            out.append(tr.params)
        if self.head is not None and hasattr(self.head, "params"):
            out.append(self.head.params)
        if self.clip_trainable is not None and hasattr(self.clip_trainable, "params"):
            out.append(self.clip_trainable.params)
        return out

    # ------------------------------------------------------------------ reward futures (TP:668,816-817,839-856)
    def _submit_score(self, images, ref, prompts, G):
        """Score generated (and reference) images on the worker thread, on its own stream, behind an event recorded on the
        caller's stream.  Returns a future of (rewards dict, reference rewards dict | None, completion event | None).
        One worker (the reference uses 8 for its PIL / HTTP scorers): the towers keep per-model workspaces, so calls are
        serialised; what overlaps is the scoring of group i with the rollout of group i + 1."""
        def score(stream_ctx):
            with stream_ctx:
                imgs = images.to(torch.bfloat16)
                r, _ = self.reward_fn(imgs, prompts, [{}] * G, scorer=self.scorer, head=self.head, only_strict=True)
                rr = None
                if ref is not None:
                    rr, _ = self.reward_fn(ref.to(torch.bfloat16), prompts, [{}] * G, scorer=self.scorer, head=self.head,
                                           only_strict=True)
            return r, rr
        if not self.async_reward:
            import contextlib
            from concurrent.futures import Future
            f = Future()
            try:
                f.set_result((*score(contextlib.nullcontext()), None))
            except Exception as e:      # surfaces at .result(), like the executor path
                f.set_exception(e)
            return f
        ready = torch.cuda.Event()
        ready.record()                                     # everything the scorer reads has been enqueued before this point

        def work():
            torch.cuda.set_device(self._dev_index)
            self._score_stream.wait_event(ready)
            r, rr = score(torch.cuda.stream(self._score_stream))
            done = torch.cuda.Event()
            done.record(self._score_stream)
            for t in (images, ref, getattr(prompts, "clip_ids", prompts)):   # consumed on the side stream: no early recycling
                if isinstance(t, torch.Tensor):
                    t.record_stream(self._score_stream)
            return r, rr, done
        return self._score_pool.submit(work)

    def _tick(self, name, t0):
        """Phase timers.  Only the calling thread's stream is drained (the scoring stream keeps running), and only when
        phase profiling is on: a device-wide synchronize here would serialise the reward futures against the rollouts."""
        if self.profile_phases:
            torch.cuda.current_stream().synchronize()
        self.timers[name] = self.timers.get(name, 0.0) + time.perf_counter() - t0

    # ------------------------------------------------------------------ hot loop 1: sampling + scoring
    def sample_epoch(self):
        """TP:727-856.  The prompt groups of an epoch are independent until the reward gather, so `groups_in_flight` of them
        (config `sample.groups_in_flight`, default 2) are rolled out at the same time, each on its own HIP stream from its own
        host thread: the kernels of one group run in the GEMM tails and epilogue bursts of the other (+10 % sampling
        throughput on one MI355X, `bench.py` -> `overlap`).  Seeds are a function of (config seed, batch index, rank), so the
        samples do not depend on the schedule."""
        c = self.cfg
        G, T = c.sample.mini_num_image_per_prompt, c.sample.train_num_steps
        nb = c.sample.num_batches_per_epoch
        neg_pe, neg_ppe = self.data.neg
        # host-side, sequential: the sampler and the data source are stateful
        plan = []
        for i in range(nb):
            self.sampler.set_epoch(self.epoch * nb + i)                                      # TP:729
            plan.append(next(iter(self.sampler))[0])
        if hasattr(self.data, "prefetch"):          # decode the reference images of the epoch's groups during sampling
            self.data.prefetch(plan[:2])

        def inputs(i):
            idx = plan[i]
            if hasattr(self.data, "prefetch") and i + 2 < nb:
                self.data.prefetch([plan[i + 2]])
            pe, ppe = self.data.prompt(idx)
            ref = self.data.reference_images(idx, G) if self.needs_reference else None     # TP:773-801
            return idx, pe, ppe, ref, self.data.clip_ids(idx, G)

        def rollout(i, idx, pe, ppe, ref, prompts):
            images, lats, lps, tss = pipeline_with_logprob_random(
                self.pipe, prompt_embeds=pe, pooled_prompt_embeds=ppe, negative_prompt_embeds=neg_pe,
                negative_pooled_prompt_embeds=neg_ppe, num_inference_steps=c.sample.num_steps,
                guidance_scale=c.sample.guidance_scale, output_type="pt", height=c.resolution, width=c.resolution,
                noise_level=c.sample.noise_level, mini_num_image_per_prompt=G, train_num_steps=T,
                process_index=self.rank, sample_num_steps=c.sample.num_steps, random_timestep=c.sample.random_timestep,
                seed=rollout_seed(c.seed, self.epoch * nb + i, self.rank))                  # TP:755-772
            first = int(self.pipe.last_random_timestep)
            score_prompts = prompts
            if hasattr(self.data, "prompt_text"):          # host scorers (ocr) read the strings, PickScore the ids
                score_prompts = rewards.PromptBatch([self.data.prompt_text(idx)] * G, clip_ids=prompts)
            # TP:816-817: rewards of the generated and of the reference images are submitted to the executor and the loop
            # goes on with the next group's rollout; they are waited for after the loop (TP:839-856)
            fut = self._submit_score(images, ref, score_prompts, G)
            lat = torch.stack(lats, dim=1)
            s = {"group": torch.full((G,), idx, dtype=torch.int32, device=self.device),
                 "prompt_embeds": pe.repeat(G, 1, 1), "pooled_prompt_embeds": ppe.repeat(G, 1),
                 "timesteps": torch.stack(tss, dim=1), "latents": lat[:, :-1], "next_latents": lat[:, 1:],
                 "log_probs": torch.stack(lps, dim=1), "images": images, "clip_ids": prompts, "_future": fut}
            if ref is not None:
                s["ref_images"] = ref
            return s, first

        self.pipe.cfg_two_streams = bool(c.sample.get("cfg_two_streams", False))     # the CFG halves of a forward on two HIP streams (same bits)
        in_flight = int(c.sample.get("groups_in_flight", 2))   # (the random SDE-window draw is per calling thread: pipeline.py)
        in_flight = max(1, min(in_flight, nb))
        # The decoder splits a batch into two half batches on two streams when it has the GPU to itself; beside another group's rollout that
        # split only adds streams taking turns CU by CU (round 6, same box, two alternations: 367.2 / 367.5 ms per step with one decode stream
        # against 376.1 / 376.9 with two; one group at a time: 386.2 / 386.7 against 384.6 / 385.0).  Every image has the same bits either way.
        vae_split = getattr(self.pipe.vae, "two_streams", None)
        if vae_split is not None:
            self.pipe.vae.two_streams = vae_split and in_flight == 1
        t0 = time.perf_counter()
        if in_flight == 1:
            done = [rollout(i, *inputs(i)) for i in range(nb)]
        else:
            if self._rollout_pool is None or self._rollout_pool._max_workers != in_flight:
                self._make_rollout_streams(in_flight)
            main = torch.cuda.current_stream()

            def work(i, args, ready):
                torch.cuda.set_device(self._rollout_dev)
                # one stream per WORKER THREAD (tasks are taken in completion order: an index-derived stream could be
                # shared by two running threads)
                st = getattr(self._rollout_tls, "stream", None)
                if st is None:
                    with self._rollout_lock:
                        st = self._rollout_tls.stream = self._rollout_streams.pop()
                st.wait_event(ready)                       # the inputs were produced on the caller's stream
                with torch.cuda.stream(st):
                    s, first = rollout(i, *args)
                    fin = torch.cuda.Event()
                    fin.record(st)
                return s, first, fin
            futs = []
            for i in range(nb):
                args = inputs(i)
                ready = torch.cuda.Event()
                ready.record(main)
                futs.append(self._rollout_pool.submit(work, i, args, ready))
            done = []
            for f in futs:
                s, first, fin = f.result()
                main.wait_event(fin)
                for t in s.values():                       # produced on a rollout stream, consumed on this one from here on
                    if isinstance(t, torch.Tensor):
                        t.record_stream(main)
                done.append((s, first))
        if vae_split is not None:
            self.pipe.vae.two_streams = vae_split
        self._tick("sample", t0)
        out = [s for s, _ in done]
        first_step = [f for _, f in done]
        t0 = time.perf_counter()
        for s in out:                                                                      # TP:839-856
            r, rr, done_ev = s.pop("_future").result()          # re-raises what the scorer raised
            if done_ev is not None:
                torch.cuda.current_stream().wait_event(done_ev)
            # produced on the scoring stream: copied on THIS stream (behind done_ev), so the scoring stream's allocator can
            # recycle its blocks whenever it likes
            s["rewards"] = torch.as_tensor(r["avg"], device=self.device).float().clone()
            if rr is not None:
                s["reference_rewards"] = torch.as_tensor(rr["avg"], device=self.device).float().clone()
        self._tick("score", t0)
        samples = {k: torch.cat([s[k] for s in out], dim=0) for k in out[0]}
        samples["first_step_index"] = first_step        # host ints, one per prompt group: no device -> host copy in the G-step
        return samples

    # ------------------------------------------------------------------ eval loop
    @torch.no_grad()
    def evaluate(self, n_prompts=None, eval_reward_fn=None, save_folder=None):
        """eval() (TP:269-382): EMA weights swapped in, `eval_num_steps` deterministic steps (noise_level = 0), one image
        per prompt, initial latents from a CPU generator seeded with 0 for every batch (TP:298-299), rewards gathered
        over ranks, means of the valid (!= -10) entries logged as eval_reward_<name> (TP:373-377).

        save_folder (scripts/eval.py:233-294): every image is also written as node0_rank<r>_<batch:05d>_<i>.png at 512x512
        (uint8 truncation of x*255, PIL's default resize) and rank 0 writes prompt2img.json = {prompt: [file, ...]} merged
        over ranks in rank order -- the file the trainers read back as `json_path`."""
        c = self.cfg
        model = self.pipe.transformer
        swapped = c.train.ema and getattr(model, "ema", None) is not None
        if swapped:                                                                         # ema.copy_ema_to(store_temp)
            live, model.params = model.params, model.ema
            model.params_bf16 = model.params.to(torch.bfloat16)
            model.refresh()
        spec = eval_reward_fn if eval_reward_fn is not None else c.eval_reward_fn
        fn = rewards.multi_score(self.device, spec.to_dict() if hasattr(spec, "to_dict") else dict(spec))
        bs = c.sample.test_batch_size
        n = n_prompts if n_prompts is not None else bs * self.world
        neg_pe, neg_ppe = self.data.neg
        acc, prompt2files, batch_idx = {}, {}, 0
        if save_folder is not None:
            os.makedirs(save_folder, exist_ok=True)
        try:
            for start in range(self.rank * bs, n, bs * self.world):                         # test set sharded over ranks
                idxs = list(range(start, min(start + bs, n)))
                pe = torch.cat([self.data.prompt(i)[0] for i in idxs])
                ppe = torch.cat([self.data.prompt(i)[1] for i in idxs])
                gen = torch.Generator().manual_seed(0)                                      # TP:298-299
                images, _, _, _ = pipeline_with_logprob_random(
                    self.pipe, prompt_embeds=pe, pooled_prompt_embeds=ppe,
                    negative_prompt_embeds=neg_pe.repeat(len(idxs), 1, 1), negative_pooled_prompt_embeds=neg_ppe.repeat(len(idxs), 1),
                    num_inference_steps=c.sample.eval_num_steps, guidance_scale=c.sample.guidance_scale, output_type="pt",
                    height=c.resolution, width=c.resolution, noise_level=0, mini_num_image_per_prompt=1,
                    process_index=self.rank, sample_num_steps=c.sample.num_steps, random_timestep=c.sample.random_timestep,
                    generator=gen)                                                          # TP:303-320
                if save_folder is not None:
                    names = write_eval_images(images, save_folder, self.rank, batch_idx)    # eval.py:258-266
                    for i, name in zip(idxs, names):
                        prompt2files[self.data.prompt_text(i)] = [name]
                batch_idx += 1
                prompts = torch.cat([self.data.clip_ids(i, 1) for i in idxs])
                ref = torch.cat([self.data.reference_images(i, 1) for i in idxs])
                r, _ = fn(images.to(torch.bfloat16), prompts, [{}] * len(idxs), scorer=self.scorer, head=self.head,
                          ref_images=ref, only_strict=True)                                 # TP:322
                for k, v in r.items():
                    acc.setdefault(k, []).append(torch.as_tensor(v, dtype=torch.float32, device=self.device).flatten())
        finally:
            if swapped:                                                                     # ema.copy_temp_to
                model.params = live
                model.params_bf16 = model.params.to(torch.bfloat16)
                model.refresh()
        if save_folder is not None:
            write_prompt2img(prompt2files, save_folder, self.world, self.rank)               # eval.py:291-294
        out = {}
        for k, chunks in acc.items():
            v = torch.cat(chunks)
            if self.world > 1:
                import torch.distributed as dist
                parts = [torch.empty_like(v) for _ in range(self.world)]
                dist.all_gather(parts, v)                                                   # accelerator.gather, TP:330
                v = torch.cat(parts)
            v = v[v != -10]
            out[f"eval_reward_{k}"] = v.mean().item() if v.numel() else float("nan")
        self.logger.log(out, self.global_step)
        return out

    def save_checkpoint(self):
        """save_ckpt (TP:389-398): rank 0 writes the (EMA) LoRA in PEFT layout under save_dir/checkpoints/checkpoint-<step>/lora."""
        from . import checkpoint
        if self.rank != 0:
            return None
        path = checkpoint.checkpoint_dir(self.cfg.save_dir, self.global_step)
        self.pipe.transformer.save_pretrained(path, use_ema=bool(self.cfg.train.ema))
        tensors, scalars = self.resume_state()
        checkpoint.save_resume_state(path, tensors, scalars)
        return path

    # ------------------------------------------------------------------ resume (what upstream cannot restore)
    def _stateful(self):
        """(name, object with .params / .exp_avg / .exp_avg_sq / .opt_step) of everything an optimizer updates."""
        out = [("lora", self.pipe.transformer)]
        if self.head is not None and hasattr(self.head, "exp_avg"):
            out.append(("head", self.head))                                                 # DINOHead, TD:592-603
        if self.clip_trainable is not None:
            out.append(("clip_last_layer", self.clip_trainable))                           # TP:1016-1020
        return out

    def resume_state(self):
        tensors, scalars = {}, {"global_step": self.global_step, "epoch": self.epoch}
        for name, obj in self._stateful():
            tensors[f"{name}.params"], tensors[f"{name}.exp_avg"] = obj.params, obj.exp_avg
            tensors[f"{name}.exp_avg_sq"] = obj.exp_avg_sq
            scalars[f"{name}.opt_step"] = int(obj.opt_step)
        tensors["lora.ema"] = self.pipe.transformer.ema
        return tensors, scalars

    def load_checkpoint(self, path):
        """Restore a checkpoint written by save_checkpoint.  With the resume file: the state is restored bit-exactly (live LoRA master
        weights, Adam moments and step counts, EMA, discriminator state, epoch / global_step -- the sampler and the noise
        streams are functions of those).  Without it (an adapter written upstream): PeftModel.from_pretrained semantics,
        TP:506-509."""
        from . import checkpoint
        tensors, scalars = checkpoint.load_resume_state(path)
        if tensors is None:
            state, _ = checkpoint.load_lora(path)
            self.pipe.transformer.load_lora_state(state)
            return False
        for name, obj in self._stateful():
            for field in ("params", "exp_avg", "exp_avg_sq"):
                getattr(obj, field).copy_(tensors[f"{name}.{field}"].to(obj.params.device))
            obj.opt_step = int(scalars[f"{name}.opt_step"])
            if hasattr(obj, "p16"):
                obj.p16.copy_(obj.params)
            if hasattr(obj, "sync_model"):
                obj.sync_model()
        tr = self.pipe.transformer
        tr.ema.copy_(tensors["lora.ema"].to(tr.params.device))
        tr.refresh()
        self.global_step, self.epoch = int(scalars["global_step"]), int(scalars["epoch"])
        return True

    # ------------------------------------------------------------------ one epoch
    def run_epoch(self):
        c = self.cfg
        T = c.sample.train_num_steps
        samples = self.sample_epoch()
        # ---- gather rewards + group ids (one packed all-gather), advantages on the device
        t0 = time.perf_counter()
        rew = samples["rewards"].unsqueeze(1).repeat(1, T)                                  # TP:926-928
        all_rew, all_gid = D.gather_rewards(rew, samples["group"])
        if c.per_prompt_stat_tracking:
            adv = self.stat_tracker.update(all_gid, all_rew)                                # TP:970
            group_stats = self.stat_tracker.last_group_stats                               # TP:975 (same launch)
            group_size, n_hist = self.stat_tracker.get_stats()
            self.stat_tracker.clear()                                                       # TP:989
        else:   # TP:991: numpy arrays upstream, i.e. float32 mean and POPULATION std (ddof = 0)
            adv = ((all_rew - all_rew.mean()) / (all_rew.std(unbiased=False) + 1e-4)).double()
            group_stats, group_size, n_hist = None, 0, 0
        samples["advantages"] = D.ungather(adv, self.world, self.rank).float()              # TP:995-999
        mean_gen = D.all_mean(samples["rewards"])                                           # TP:1008
        mean_ref = D.all_mean(samples["reference_rewards"]) if "reference_rewards" in samples else None   # TP:1011
        self._tick("gather+advantage", t0)
        metrics = {"epoch": self.epoch, "reward_avg": all_rew[:, 0].mean(), "group_size": group_size, "trained_prompt_num": n_hist}
        if mean_ref is not None:
            metrics["reference_reward_avg"] = mean_ref
        if group_stats is not None:                                                         # TP:977-988
            metrics["zero_std_ratio"], metrics["reward_std_mean"] = group_stats[0], group_stats[1]
        self.last_metrics = metrics
        self.logger.log(metrics, self.global_step)
        # ---- D or G (identical on every rank: derived from gathered data / the epoch counter)
        # (a config without a discriminator -- pickscore_sd3_fast, the multi-reward preset -- keeps the gate at G: SURVEY 8f)
        train_d = bool(c.get("train_d", False))
        if self.variant == "dino":
            do_d = train_d and (self.epoch + 1) % c.d_times != 0                            # TD:1097
        else:
            do_d = train_d and bool(mean_ref < mean_gen)                                    # TP:1025
        if do_d:
            t0 = time.perf_counter()
            info = self.d_step(samples)
            self._tick("d_step", t0)
            self.logger.log(info, self.global_step)
            self.global_step += 1
            self.epoch += 1
            return {"phase": "D", **info}
        t0 = time.perf_counter()
        info = self.g_step(samples)
        self._tick("g_step", t0)
        self.epoch += 1
        return {"phase": "G", **info}

    def d_step(self, samples):
        reduce = D.average_gradients if self.world > 1 else None
        if self.variant != "dino":
            d_loss = train_pickscore(self.clip_trainable, samples["clip_ids"], samples["ref_images"], samples["images"],
                                     lr=self.cfg.d_lr, all_reduce=reduce)                   # TP:1025-1037
            return {"train/d_loss": d_loss}
        d_loss, acc = train_dino(self.scorer, self.head, None, samples["ref_images"], samples["images"], lr=self.cfg.d_lr,
                                 all_reduce=reduce)
        return {"train/d_loss": d_loss, "train/acc": acc}

    def g_step(self, samples):
        c = self.cfg
        model = self.pipe.transformer
        G, T, nb = c.sample.mini_num_image_per_prompt, c.sample.train_num_steps, c.sample.num_batches_per_epoch
        GA = max(1, c.train.gradient_accumulation_steps)
        neg_pe, neg_ppe = self.data.neg
        # the sampling schedule (upstream: whatever the last rollout left on the pipeline's scheduler, PF:574): installed here as
        # well so that a trainer restored from a checkpoint, or one whose last rollout was an eval with another step count,
        # replays against the same sigma table (the tables are cached per step count: no allocation after the first time)
        self.pipe.scheduler.set_timesteps(c.sample.num_steps, device=self.device)
        agg = {}
        n_acc = 0
        # HIP-event pairs around every micro-step / optimizer step of this G-step (events on the launch stream: no host sync, no
        # cost): bench.py's epoch leg reads them after its final synchronize -- the in-situ micro-step time, not an isolated one
        ev = self.gstep_events = []

        def timed(kind, fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn()
            e1.record()
            ev.append((kind, e0, e1))
            return out
        for inner in range(c.train.num_inner_epochs):
            for i in range(nb):
                sl = slice(i * G, (i + 1) * G)
                s = {k: samples[k][sl] for k in ("latents", "next_latents", "timesteps", "log_probs", "advantages",
                                                 "prompt_embeds", "pooled_prompt_embeds")}
                embeds = torch.cat([neg_pe.repeat(G, 1, 1), s["prompt_embeds"]])            # TP:1084-1091
                pooled = torch.cat([neg_ppe.repeat(G, 1), s["pooled_prompt_embeds"]])
                for j in range(T):
                    info = timed("micro_step", lambda: g_step.micro_step(
                        model, self.pipe.scheduler, s, j, embeds, pooled, s["log_probs"][:, j], s["advantages"][:, j],
                        guidance_scale=c.sample.guidance_scale, noise_level=c.sample.noise_level, adv_clip_max=c.train.adv_clip_max,
                        clip_range=c.train.clip_range, loss_scale=1.0 / (GA * T), step_index=samples["first_step_index"][i] + j,
                        beta=c.train.beta))
                    for k in ("loss", "approx_kl", "clipfrac", "clipfrac_gt_one", "clipfrac_lt_one", "policy_loss") + \
                            (("kl_loss",) if c.train.beta > 0 else ()):                     # TP:1158-1160
                        agg[k] = agg.get(k, 0) + info[k]
                    n_acc += 1
                if (i + 1) % GA == 0:                                                       # sync_gradients, TP:1166-1185
                    timed("grad_all_reduce", lambda: D.average_gradients(model.grads))     # TP:1165 (DeepSpeed / DDP)
                    timed("optimizer_step", lambda: model.optimizer_step(
                        lr=c.train.learning_rate, betas=(c.train.adam_beta1, c.train.adam_beta2), eps=c.train.adam_epsilon,
                        weight_decay=c.train.adam_weight_decay, max_grad_norm=c.train.max_grad_norm))
                    out = D.reduce_mean({k: v / n_acc for k, v in agg.items()})                # TP:1179-1180 (mean over ranks)
                    out.update({"epoch": self.epoch, "inner_epoch": inner})
                    self.logger.log(out, self.global_step)
                    self.global_step += 1
                    agg, n_acc = {}, 0
                if c.train.ema:
                    model.ema_step(self.global_step)                                        # TP:1186-1187
        return {"global_step": self.global_step}
