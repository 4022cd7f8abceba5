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


def test_header_is_plain_c_and_struct_layouts_match_the_ctypes_mirrors(tmp_path):
    """include/advgrpo.h is what a foreign binding (cgo / JNI / ctypes) compiles against: it must compile as C (gcc, no C++), and
    every descriptor struct the Python side mirrors in adv_grpo_amd/_lib.py must have the same size and the same field offsets, name by name."""
    import ctypes, re, shutil, subprocess
    from adv_grpo_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "advgrpo.h")).read()
    mirrors = {"advgrpo_gemm_desc": _lib.GemmDesc, "advgrpo_ln_desc": _lib.LnDesc, "advgrpo_mmdit_block_desc": _lib.MMDiTBlockDesc,
               "advgrpo_lora_merge_item": _lib.LoraMergeItem, "advgrpo_tn_desc": _lib.TnDesc, "advgrpo_fp8_scales": _lib.Fp8Scales,
               "advgrpo_vit_layer": _lib.VitLayer, "advgrpo_vit_desc": _lib.VitDesc, "advgrpo_mmdit_block_bwd_desc": _lib.MMDiTBlockBwdDesc,
               "advgrpo_vae_conv": _lib.VaeConv, "advgrpo_vae_resnet": _lib.VaeResnet, "advgrpo_vae_decoder_desc": _lib.VaeDecoderDesc}
    declared = set(re.findall(r"^\} (advgrpo_\w+);", header, flags=re.M)) | set(re.findall(r"typedef struct \w+ \{[^}]*\} (advgrpo_\w+);", header))
    assert declared == set(mirrors), (declared, set(mirrors))          # a struct added to the header needs its mirror (and this test)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "advgrpo.h"', 'int main(void) {']
    for cname, cls in mirrors.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        cname, fname, val = ln.split()
        got[(cname, fname)] = int(val)
    for cname, cls in mirrors.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), (cname, got[(cname, "size")], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_gemm_dispatch_counts_workgroup_slots_for_a_reward_towers_rows():
    """advgrpo_gemm_variant is host logic (no launch): the wide Linears of the MMDiT take the eight-phase kernel (30), a full round of
    128 x 128 tiles stays (15), and between 1024 and 8192 rows a launch under half a round of the 2 x 256 workgroup slots becomes
    128 x 64 tiles (1), one that would need a ragged second round becomes 192 x 128 tiles (26) -- CLIP ViT-H at 8 x 257 rows."""
    from adv_grpo_amd import _lib
    lib = _lib.load()
    v = lambda M, N, K: lib.advgrpo_gemm_variant(M, N, K, 1, 0)
    assert v(19664, 1536, 1536) == 30 and v(16384, 6144, 1536) == 30
    assert v(2056, 3840, 1280) == 15          # QKV: 510 tiles, one round
    assert v(2056, 1280, 1280) == 1           # out-proj: 170 tiles -> 340 of 128 x 64
    assert v(2056, 1280, 5120) == 1           # FC2
    assert v(2056, 5120, 1280) == 26          # FC1: 680 tiles (1.33 rounds) -> 440 of 192 x 128
    assert v(77, 1024, 1024) == 15 and v(1232, 768, 768) == 15      # below the rule's range: unchanged
    assert v(512, 32, 1536) == 1 and v(32, 1536, 1536) == 2         # skinny shapes keep their narrow tiles
