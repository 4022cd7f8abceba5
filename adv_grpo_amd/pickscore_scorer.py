"""PickScore scorer on the gfx950 kernels.

Mirror of adv_grpo/pickscore_scorer.py:5-52 (class name, ``__call__(prompt, images) -> scores[N]``).
Differences that come with the platform, none of which change the number computed:
  * images may be a device tensor [N,3,H,W] in [0,1] (what the trainer has) -- the uint8 quantisation and
    PIL-exact CLIPProcessor resize run on the device instead of GPU->CPU->PIL->GPU;  a list of PIL images is
    still accepted (converted once);
  * prompts are either strings (needs a ``tokenizer`` callable; no tokenizer files exist on the GPU box) or
    ready ``input_ids`` [N,77];
  * weights come from a state dict (``model_sd``) because no checkpoint can be downloaded here;
  * ``dtype`` selects the arithmetic as in the reference: ``torch.bfloat16`` (the co-trained scorer, TP:514) runs the
    bf16 towers of vit.py; ``torch.float32`` (the ``pickscore`` reward factory, rewards.py:564,596) runs vit_x3.py --
    fp32-equivalent split-bf16 products (gfx950 has no fast fp32 matrix path: 1/16 of the bf16 rate), everything between
    two products in f32; both are bounded against the fp32 oracle in tests/test_gpu_vit.py.  ``compute_dtype`` says
    what runs ("bf16" or "bf16x3").
"""
import threading

import numpy as np
import torch

from . import ops, vit, vit_x3


class PickScoreScorer(torch.nn.Module):
    def __init__(self, device="cuda", dtype=torch.float32, model_sd=None, clip_cfg=None, tokenizer=None):
        super().__init__()
        if model_sd is None or clip_cfg is None:
            raise RuntimeError("PickScoreScorer needs model_sd + clip_cfg (no checkpoint download on this platform)")
        self.device = device
        self.dtype = dtype
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"PickScoreScorer: dtype must be torch.float32 or torch.bfloat16, got {dtype}")
        self.compute_dtype = "bf16x3" if dtype == torch.float32 else "bf16"
        self.tokenizer = tokenizer
        self.model = (vit_x3.CLIPModelX3 if dtype == torch.float32 else vit.CLIPModel)(model_sd, clip_cfg, device)
        self._text_streams = {}         # calling stream -> side stream of the text tower
        self._text_streams_lock = threading.Lock()

    def _images(self, images):
        if isinstance(images, torch.Tensor):
            return images.to(self.device)
        arr = np.stack([np.asarray(im.convert("RGB")) for im in images])        # PIL list
        return torch.from_numpy(arr).to(self.device).permute(0, 3, 1, 2).float() / 255.0

    def _ids(self, prompt):
        if isinstance(prompt, torch.Tensor):
            return prompt
        if getattr(prompt, "clip_ids", None) is not None:       # rewards.PromptBatch: strings that carry their token ids
            return prompt.clip_ids
        if self.tokenizer is None:
            raise RuntimeError("string prompts need a tokenizer; pass input_ids [N,77] instead")
        return self.tokenizer(prompt, padding="max_length", truncation=True, max_length=77, return_tensors="pt").input_ids

    def _text_features(self, prompt):
        """The text tower on the UNIQUE prompts of the call.  A GRPO group scores G images of ONE prompt (TP:813-817 repeat the
        prompt G times): when the prompts arrive as host strings (rewards.PromptBatch) equal strings are found on the host --
        no device comparison, no synchronisation -- and the tower runs once per distinct prompt; every row of its GEMMs is
        independent of the others, so the embeddings are bit for bit those of the full batch (tests/test_gpu_vit.py)."""
        ids = self._ids(prompt)
        if isinstance(prompt, (list, tuple)) and len(prompt) == ids.shape[0] and all(isinstance(t, str) for t in prompt):
            first, inverse = {}, []
            for i, t in enumerate(prompt):
                inverse.append(first.setdefault(t, len(first)))
            if len(first) < len(prompt):
                rows = [0] * len(first)
                for i, u in reversed(list(enumerate(inverse))):
                    rows[u] = i
                ids = ids.to(self.device)
                uniq = self.model.get_text_features(ids.index_select(0, torch.tensor(rows, device=ids.device)))
                return uniq.index_select(0, torch.tensor(inverse, device=uniq.device))
        return self.model.get_text_features(ids)

    def prepare_streams(self, mains):
        """Choose -- by measurement (ops.concurrent_stream: device-wide synchronisations, micro-bursts) -- the text tower's side stream for
        every stream the scorer will be called on.  Construction-time work: the Trainer calls it from __init__ for its scoring stream and
        the launch stream, before a worker thread or a rollout exists (measured lazily from the scoring worker the bursts waited on
        everyone's kernels and timed other threads' load)."""
        for main in mains:
            with self._text_streams_lock:
                if main.cuda_stream not in self._text_streams:
                    self._text_streams[main.cuda_stream] = ops.concurrent_stream(self.device, [main])

    def _side_stream(self, main):
        """The text tower's stream for calls made on `main`: the towers are independent until the logits, and the text tower of
        one prompt is 77 rows -- 24 layers of GEMMs that give 8 - 32 workgroups to 256 CUs, 3.2 ms of latency that hides
        completely beside the image tower.  One stream per calling stream (scorer calls may come from worker threads).  A calling
        stream that was not announced through prepare_streams gets a plain new stream, unmeasured: no device-wide synchronisation on
        a hot path (it may share the caller's hardware queue and then simply runs behind it)."""
        key = main.cuda_stream
        with self._text_streams_lock:
            st = self._text_streams.get(key)
            if st is None:
                st = self._text_streams[key] = torch.cuda.Stream(device=self.device)
        return st

    @torch.no_grad()
    def __call__(self, prompt, images):
        dev = torch.device(self.device)
        if dev.type != "cuda":
            image_embs = self.model.get_image_features(images=self._images(images))
            text_embs = self._text_features(prompt)
        else:
            main = torch.cuda.current_stream(dev)
            side = self._side_stream(main)
            ready = torch.cuda.Event()
            ready.record(main)                           # token ids the caller produced on its stream
            # image tower FIRST: its launches queue up on the device (8 ms of kernels for 3 - 4 ms of host time); the text tower's
            # launches then reach a device that is still busy and run beside it.  The other order overlaps nothing: 77-row kernels
            # finish as fast as the host can issue them.
            image_embs = self.model.get_image_features(images=self._images(images))
            side.wait_event(ready)
            with torch.cuda.stream(side):
                text_embs = self._text_features(prompt)
            main.wait_stream(side)
            text_embs.record_stream(main)
        if self.dtype == torch.float32:
            return vit_x3.pickscore_scores_f32(image_embs, text_embs, self.model.logit_scale)
        return vit.pickscore_scores(image_embs, text_embs, self.model.logit_scale)
