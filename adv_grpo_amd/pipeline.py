"""The pipeline object the rollout function takes as ``self`` (what StableDiffusion3Pipeline is to the
reference, scripts/train_sd3_fast_pickscore.py:447-486): transformer + scheduler + VAE decoder."""
import threading

import torch

from . import _lib
from .scheduler import FlowMatchEulerDiscreteScheduler


class SD3Pipeline:
    default_sample_size = 128
    vae_scale_factor = 8

    def __init__(self, transformer, vae, device="cuda"):
        self.transformer = transformer
        self.vae = vae
        self.device = torch.device(device)
        self.scheduler = FlowMatchEulerDiscreteScheduler(device=self.device)
        self._guidance_scale = 1.0
        self._tls = threading.local()

    @property
    def last_random_timestep(self):
        """First recorded scheduler index of the calling thread's latest rollout (prompt groups are rolled out from
        several host threads at once; each reads back its own draw)."""
        return getattr(self._tls, "last_random_timestep", None)

    @last_random_timestep.setter
    def last_random_timestep(self, v):
        self._tls.last_random_timestep = v

    @property
    def _execution_device(self):
        return self.device

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    def prepare_latents(self, batch_size, channels, height, width, dtype, device, seed):
        """N(0,1) latents [B,C,h/8,w/8] from the in-kernel Philox stream (reference: randn_tensor, PF:559-568)."""
        lib = _lib.load()
        out = torch.empty(batch_size, channels, height // 8, width // 8, dtype=dtype, device=device)
        _lib.check(lib.advgrpo_randn(out.data_ptr(), _lib.dtype_code(dtype), out.numel(), int(seed), 0, _lib.stream_ptr()))
        return out
