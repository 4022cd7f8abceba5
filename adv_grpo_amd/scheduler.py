"""Flow-matching Euler schedule (host mirror of the diffusers object the reference passes around).

The reference reaches ``scheduler.sigmas`` / ``index_for_timestep`` / ``set_timesteps`` at
adv_grpo/diffusers_patch/sd3_sde_with_logprob.py:106-110 and
sd3_pipeline_with_logprob_fast.py:574; SD3.5-medium config: 1000 train steps, shift 3.0, static
shift applied on the inference grid as well.  Unlike the reference's per-sample
``(timesteps == t).nonzero().item()`` (one host sync per sample per step) the index is resolved
on the host copy of the schedule; only the 11-float sigma table lives on the device.
"""
import threading

import numpy as np
import torch


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, shift=3.0, device=None):
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        self.device = device
        sig = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1] / np.float32(
            num_train_timesteps)
        sig = (np.float32(shift) * sig / (1 + (np.float32(shift) - 1) * sig)).astype(np.float32)
        self.sigma_max = float(sig[0])
        self.sigma_min = float(sig[-1])
        self._lock = threading.Lock()    # first fill of a table: two rollout threads may miss at the same time
        self._tables = {}          # (n_steps, device) -> (sigmas_host, timesteps_host, sigmas, timesteps): immutable, never freed
        self._install(sig, append_zero=False)

    def _install(self, sig, append_zero):
        sig = np.asarray(sig, dtype=np.float32)
        ts = sig * np.float32(self.num_train_timesteps)
        if append_zero:
            sig = np.concatenate([sig, np.zeros(1, dtype=np.float32)])
        self._sigmas_host = sig
        self._timesteps_host = ts
        self.sigmas = torch.from_numpy(sig.copy()).to(self.device) if self.device else torch.from_numpy(sig.copy())
        self.timesteps = torch.from_numpy(ts.copy()).to(self.device) if self.device else torch.from_numpy(ts.copy())

    def set_timesteps(self, num_inference_steps, device=None):
        """Installs the n-step table.  The device tensors of a given (n, device) are built ONCE and kept for the life of the
        scheduler: concurrent rollouts (two prompt groups in flight on two streams, trainer.sample_epoch) each re-install the
        same objects, so no kernel queued on another stream can find its sigma table freed and recycled under it."""
        if device is not None:
            self.device = device
        key = (int(num_inference_steps), str(self.device))
        hit = self._tables.get(key)
        if hit is not None:
            self._sigmas_host, self._timesteps_host, self.sigmas, self.timesteps = hit
            return
        with self._lock:
            self._build(key, num_inference_steps)

    def _build(self, key, num_inference_steps):
        hit = self._tables.get(key)         # (filled by the thread that held the lock before us)
        if hit is not None:
            self._sigmas_host, self._timesteps_host, self.sigmas, self.timesteps = hit
            return
        t = np.linspace(self.sigma_max * self.num_train_timesteps, self.sigma_min * self.num_train_timesteps,
                        num_inference_steps)
        sig = t / self.num_train_timesteps
        sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        self._install(sig.astype(np.float32), append_zero=True)
        if self.sigmas.is_cuda:
            torch.cuda.current_stream(self.sigmas.device).synchronize()     # one-time: the tables are complete before any stream reads them
        self._tables[key] = (self._sigmas_host, self._timesteps_host, self.sigmas, self.timesteps)

    def index_for_timestep(self, timestep):
        t = float(timestep)
        hits = np.nonzero(self._timesteps_host == np.float32(t))[0]
        if len(hits) == 0:
            raise ValueError(f"timestep {t} is not on the schedule")
        return int(hits[1] if len(hits) > 1 else hits[0])


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, **kwargs):
    scheduler.set_timesteps(num_inference_steps, device=device)
    return scheduler.timesteps, num_inference_steps
