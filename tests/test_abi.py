"""CPU: the C-ABI library loads and exports every symbol include/advgrpo.h declares (no compute
calls), and the host logic that needs no device (sampler, scheduler) matches the goldens."""
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "advgrpo.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(advgrpo_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from adv_grpo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/advgrpo.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in adv_grpo_amd/_lib.py"
    assert lib.advgrpo_abi_version() == 1
    assert set(_lib.SIGNATURES) == set(names)


def test_missing_library_fails_loudly(monkeypatch):
    from adv_grpo_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libadvgrpo_hip.so")
    with pytest.raises(_lib.AdvGrpoError):
        _lib.load()


def test_cpu_tensor_is_rejected():
    import torch
    from adv_grpo_amd import _lib
    with pytest.raises(_lib.AdvGrpoError):
        _lib.ptr(torch.zeros(4))


def test_sampler_matches_reference_goldens():
    from adv_grpo_amd.sampler import DistributedKRepeatSampler
    for c in json.load(open(os.path.join(ROOT, "tests", "golden", "sampler.json"))):
        for r in range(c["n"]):
            s = DistributedKRepeatSampler(range(c["dataset_len"]), c["b"], c["k"], c["n"], r, seed=c["seed"])
            s.set_epoch(c["epoch"])
            assert next(iter(s)) == c["per_rank"][r]


def test_scheduler_matches_oracle():
    from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from oracle.scheduler import FlowMatchEulerScheduler
    for n in (4, 10, 40):
        a = FlowMatchEulerDiscreteScheduler(); a.set_timesteps(n)
        b = FlowMatchEulerScheduler(); b.set_timesteps(n)
        assert np.array_equal(a.sigmas.numpy(), b.sigmas.numpy())
        assert np.array_equal(a.timesteps.numpy(), b.timesteps.numpy())
        for i in range(n):
            assert a.index_for_timestep(a.timesteps[i]) == i
