"""FlowMatchEulerDiscreteScheduler restatement (test infrastructure).

PARITY UNPINNED: diffusers==0.33.1 (setup.py:8-52 of the reference) is not in
/root/reference nor in this image.  This restates its published
``FlowMatchEulerDiscreteScheduler`` for the SD3.5-medium scheduler config
(num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=False), which the
reference reaches at sd3_pipeline_with_logprob_fast.py:574 (retrieve_timesteps)
and sd3_sde_with_logprob.py:106-110 (sigmas / index_for_timestep).
"""
import numpy as np
import torch


class FlowMatchEulerScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, shift=3.0):
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        t = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = torch.from_numpy(t).to(torch.float32) / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigma_max = sig[0].item()
        self.sigma_min = sig[-1].item()
        self.timesteps = sig * num_train_timesteps
        self.sigmas = sig
        self._step_index = None

    def _sigma_to_t(self, s):
        return s * self.num_train_timesteps

    def set_timesteps(self, num_inference_steps, device=None):
        device = device if device is not None else getattr(self, "device", None)
        t = np.linspace(self._sigma_to_t(self.sigma_max), self._sigma_to_t(self.sigma_min),
                        num_inference_steps)
        sig = t / self.num_train_timesteps
        # the (static) shift is applied a second time on the inference grid
        sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        sig = torch.from_numpy(sig).to(dtype=torch.float32, device=device)
        self.timesteps = sig * self.num_train_timesteps
        self.sigmas = torch.cat([sig, torch.zeros(1, device=sig.device)])
        self._step_index = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        if schedule_timesteps is None:
            schedule_timesteps = self.timesteps
        indices = (schedule_timesteps == timestep).nonzero()
        pos = 1 if len(indices) > 1 else 0
        return indices[pos].item()


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, **kwargs):
    """diffusers pipeline helper used at sd3_pipeline_with_logprob_fast.py:574."""
    scheduler.set_timesteps(num_inference_steps, device=device)
    return scheduler.timesteps, num_inference_steps
