"""Reference-image data path: the ``json_path`` map {prompt: [file, ...]} + ``reference_image_path``
directory the trainers read inside the sampling loop (scripts/train_sd3_fast_pickscore.py:705-707,773-801):

    Image.open(path).convert("RGB") -> transforms.Resize((512, 512)) -> transforms.ToTensor() -> stack -> device, f32

Upstream this is synchronous file I/O + PIL decode between two rollouts.  Here the same arithmetic (PIL's antialiased
bilinear resize, /255) runs on a small thread pool that decodes the NEXT prompts while the GPU samples the current
one (``prefetch``); ``get`` then only waits for the future and does one host-to-device copy.  Host code, no kernels."""
import json
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def load_image(path, resolution=512):
    """One file -> [3,R,R] float32 in [0,1] exactly as Resize((R,R)) + ToTensor() on a PIL image
    (torchvision's Resize on PIL input = Image.resize(..., BILINEAR), which antialiases)."""
    from PIL import Image
    img = Image.open(path).convert("RGB")                                              # TP:779
    if img.size != (resolution, resolution):
        img = img.resize((resolution, resolution), Image.BILINEAR)                     # TP:793
    a = np.asarray(img, dtype=np.uint8)
    return torch.from_numpy(a).permute(2, 0, 1).contiguous().float().div_(255.0)      # ToTensor, TP:794


class ReferenceImageStore:
    def __init__(self, json_path, image_dir, resolution=512, device="cuda", fallback_path=None, workers=4):
        with open(json_path, "r", encoding="utf-8") as f:                              # TP:705-707
            self.map = json.load(f)
        self.dir, self.res, self.device, self.fallback = image_dir, resolution, device, fallback_path
        self.pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="refimg")
        self._pending, self._lock = {}, threading.Lock()

    def __contains__(self, prompt):
        return prompt in self.map

    def _load(self, prompt):
        out = []
        for fname in self.map[prompt]:
            path = os.path.join(self.dir, fname)
            try:
                out.append(load_image(path, self.res))
            except Exception as e:                                                     # TP:781-785: fall back to a default image
                if self.fallback is None:
                    raise FileNotFoundError(f"reference image {path}: {e} (and no fallback_path configured)") from e
                out.append(load_image(self.fallback, self.res))
        return torch.stack(out, dim=0)                                                 # [n,3,R,R], TP:798-799

    def prefetch(self, prompts):
        """Start decoding the images of `prompts` in the background (idempotent)."""
        with self._lock:
            for p in prompts:
                if p in self.map and p not in self._pending:
                    self._pending[p] = self.pool.submit(self._load, p)

    def get(self, prompt, n=None):
        """-> [n,3,R,R] float32 on the device (first n files of the prompt's list; all of them when n is None)."""
        if prompt not in self.map:
            # upstream only prints a warning and silently reuses the previous prompt's images (TP:786-790): refuse instead
            raise KeyError(f"no reference images for prompt {prompt!r}")
        with self._lock:
            fut = self._pending.pop(prompt, None)
        imgs = fut.result() if fut is not None else self._load(prompt)
        if n is not None:
            if imgs.shape[0] < n:
                raise ValueError(f"prompt {prompt!r} has {imgs.shape[0]} reference images, {n} requested")
            imgs = imgs[:n]
        return imgs.pin_memory().to(self.device, non_blocking=True) if str(self.device).startswith("cuda") else imgs

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)
