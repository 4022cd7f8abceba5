"""EMA of the trainable parameters on the device (mirror of adv_grpo/ema.py:10-77, ``EMAModuleWrapper``).

Same class name, constructor and methods as the reference: ``step(parameters, optimization_step)``,
``copy_ema_to(parameters, store_temp=True)``, ``copy_temp_to(parameters)``, ``state_dict`` / ``load_state_dict``.
``parameters`` is an iterable of f32 device tensors (the trainer hands over its one flat LoRA vector).  The update
``e += (1 - decay) * (p - e)`` runs in ``advgrpo_ema_step`` with the reference's two roundings (multiply, then add),
so the 40-step golden sequence made from the reference class is reproduced bit for bit (tests/test_gpu_leaf_kernels.py).
"""
import torch

from . import _lib


class EMAModuleWrapper:
    def __init__(self, parameters, decay=0.9999, update_step_interval=1, device=None):
        parameters = list(parameters)
        self.ema_parameters = [p.clone().detach().to(device) if device is not None else p.clone().detach() for p in parameters]
        self.temp_stored_parameters = None
        self.decay = decay
        self.update_step_interval = update_step_interval
        self.device = device

    def get_current_decay(self, optimization_step):
        return min((1 + optimization_step) / (10 + optimization_step), self.decay)      # ema.py:33-37

    @torch.no_grad()
    def step(self, parameters, optimization_step):
        parameters = list(parameters)
        one_minus_decay = 1 - self.get_current_decay(optimization_step)
        if (optimization_step + 1) % self.update_step_interval != 0:                    # ema.py:44
            return
        lib = _lib.load()
        if len(parameters) != len(self.ema_parameters):
            raise ValueError("EMAModuleWrapper.step: parameter list length changed")     # zip(strict=True) upstream
        for e, p in zip(self.ema_parameters, parameters):
            if e.dtype != torch.float32 or p.dtype != torch.float32 or e.numel() != p.numel():
                raise _lib.AdvGrpoError("EMA runs on f32 tensors of equal size")
            pc = p.contiguous()
            _lib.check(lib.advgrpo_ema_step(_lib.ptr(e), _lib.ptr(pc), e.numel(), float(one_minus_decay),
                                            _lib.stream_ptr()))

    def copy_ema_to(self, parameters, store_temp=True):
        parameters = list(parameters)
        if store_temp:
            self.temp_stored_parameters = [p.detach().clone() for p in parameters]     # (upstream parks them on the CPU)
        for e, p in zip(self.ema_parameters, parameters):
            p.data.copy_(e.to(p.device).data)

    def copy_temp_to(self, parameters):
        for t, p in zip(self.temp_stored_parameters, list(parameters)):
            p.data.copy_(t.data)
        self.temp_stored_parameters = None

    def load_state_dict(self, state_dict):
        self.decay = self.decay if self.decay else state_dict.get("decay", self.decay)
        loaded = list(state_dict.get("ema_parameters"))
        if len(loaded) == len(self.ema_parameters) and all(a.shape == b.shape for a, b in zip(loaded, self.ema_parameters)):
            for e, p in zip(self.ema_parameters, loaded):      # in place: callers alias these tensors (model.ema)
                e.copy_(p.to(e.device))
        else:
            self.ema_parameters = [p.to(self.device) if self.device is not None else p for p in loaded]

    def state_dict(self):
        return {"decay": self.decay, "ema_parameters": self.ema_parameters}
