#!/usr/bin/env python3
"""The experimental 4-wave / two-workgroups-per-CU GEMM (csrc/gemm4w.hip, ADVGRPO_GEMM_FORCE=31) against the eight-phase kernel:
the four rollout epilogues at the config-2 sizes, bit for bit.  Usage: check_gemm4w.py  (re-runs itself once per variant)."""
import os, sys, subprocess, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
B, NI, NT, D = 16, 1024, 205, 1536
g = torch.Generator(device="cuda").manual_seed(0)
def rnd(*s): return torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
x, c = rnd(B * NI, D), rnd(B * NT, D)
gate = rnd(B, D)
S = NI + NT
w1, b1, w2, b2 = rnd(3 * D, D) * 0.02, rnd(3 * D), rnd(3 * D, D) * 0.02, rnd(3 * D)
rmsw = rnd(2, 64)
wf1, bf1 = rnd(4 * D, D) * 0.02, rnd(4 * D)
wo, bo = rnd(D, D) * 0.02, rnd(D)
res = rnd(B * NI, D)
out = {}
qkv = torch.zeros(B * S, 3 * D, dtype=torch.bfloat16, device="cuda")
ops.gemm_grouped([ops.gemm_desc(x, w1, bias=b1, out=qkv, seg=(NI, S, 0), rms=(rmsw, 48, 24, 1e-6, None)),
                  ops.gemm_desc(c, w2, bias=b2, out=qkv, seg=(NT, S, NI), rms=(rmsw, 48, 24, 1e-6, None))])
out["qkv"] = qkv
out["ff1"] = ops.gemm(x, wf1, bias=bf1, act="gelu_tanh")
out["out"] = ops.gemm(x, wo, bias=bo, gate=gate, gate_rows=NI, residual=res)
out["plain"] = ops.gemm(x, wo, bias=bo)
torch.cuda.synchronize()
if len(sys.argv) > 1:
    torch.save({k: v.cpu() for k, v in out.items()}, sys.argv[1])
    sys.exit(0)
torch.save({k: v.cpu() for k, v in out.items()}, "/tmp/gemm_ref.pt")
exp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adv_grpo_amd", "libadvgrpo_experiments.so")
assert os.path.exists(exp), "build the experiments library first: make -C adv_grpo_amd/csrc EXPERIMENTS=1"
env = dict(os.environ, ADVGRPO_GEMM_FORCE="31", ADVGRPO_LIB=exp)
subprocess.check_call([sys.executable, os.path.abspath(__file__), "/tmp/gemm_4w.pt"], env=env)
a, b = torch.load("/tmp/gemm_ref.pt"), torch.load("/tmp/gemm_4w.pt")
ok = True
for k in a:
    same = torch.equal(a[k], b[k])
    ok &= same
    print(k, "bit-identical" if same else f"max diff {(a[k].float() - b[k].float()).abs().max().item():.3e}")
sys.exit(0 if ok else 1)
