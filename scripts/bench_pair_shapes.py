"""The four paired Linears of a joint MMDiT block at the config-2 sizes, with their real epilogues, timed per shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
B, NI, NT, D = 16, 1024, 205, 1536
dev = "cuda"
def rnd(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
x, c = rnd(B * NI, D), rnd(B * NT, D)
h_x, h_c = rnd(B * NI, 4 * D), rnd(B * NT, 4 * D)
gate = rnd(B, D)
S = NI + NT
qkv = torch.empty(B * S, 3 * D, dtype=torch.bfloat16, device=dev)
rmsw = rnd(2, 64)
def lin(n, k): return rnd(n, k) * 0.02, rnd(n)
cases = {}
w1, b1 = lin(3 * D, D); w2, b2 = lin(3 * D, D)
def qkv_fn():
    d0 = ops.gemm_desc(x, w1, bias=b1, out=qkv, seg=(NI, S, 0), rms=(rmsw, 48, 24, 1e-6, None))
    d1 = ops.gemm_desc(c, w2, bias=b2, out=qkv, seg=(NT, S, NI), rms=(rmsw, 48, 24, 1e-6, None))
    ops.gemm_grouped([d0, d1])
cases["qkv (bias, qk-norm, scatter)"] = (qkv_fn, 2 * B * S * 3 * D * D)
att = rnd(B * S, D); wo, bo = lin(D, D); wo2, bo2 = lin(D, D)
def out_fn():
    d0 = ops.gemm_desc(att, wo, bias=bo, gate=gate, gate_rows=NI, residual=x, out=x, a_seg=(NI, S, 0), M=B * NI)
    d1 = ops.gemm_desc(att, wo2, bias=bo2, gate=gate, gate_rows=NT, residual=c, out=c, a_seg=(NT, S, NI), M=B * NT)
    ops.gemm_grouped([d0, d1])
cases["out-proj (bias, gate, residual, gather)"] = (out_fn, 2 * B * S * D * D)
wf1, bf1 = lin(4 * D, D); wf1c, bf1c = lin(4 * D, D)
def ff1_fn():
    ops.gemm_grouped([ops.gemm_desc(x, wf1, bias=bf1, act="gelu_tanh", out=h_x), ops.gemm_desc(c, wf1c, bias=bf1c, act="gelu_tanh", out=h_c)])
cases["ff1 (bias, gelu)"] = (ff1_fn, 2 * B * S * 4 * D * D)
wf2, bf2 = lin(D, 4 * D); wf2c, bf2c = lin(D, 4 * D)
def ff2_fn():
    ops.gemm_grouped([ops.gemm_desc(h_x, wf2, bias=bf2, gate=gate, gate_rows=NI, residual=x, out=x),
                      ops.gemm_desc(h_c, wf2c, bias=bf2c, gate=gate, gate_rows=NT, residual=c, out=c)])
cases["ff2 (bias, gate, residual)"] = (ff2_fn, 2 * B * S * 4 * D * D)
tot_t = tot_f = 0
for name, (fn, fl) in cases.items():
    t = timeit(fn); tot_t += t; tot_f += fl
    print(f"{name:42s} {t:8.1f} us  {fl / t / 1e6:7.0f} TFLOP/s")
print(f"{'block total':42s} {tot_t:8.1f} us  {tot_f / tot_t / 1e6:7.0f} TFLOP/s")
