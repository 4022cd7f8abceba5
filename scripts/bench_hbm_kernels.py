"""HBM-bound kernels of the path at the config-2 sizes: achieved TB/s of algorithmic bytes (reported
separately from the MFMA roofline, against 8 TB/s peak / ~6.3 TB/s achievable)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops, stat_tracking
from adv_grpo_amd.diffusers_patch.sd3_sde_with_logprob import sde_step_cfg
from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
dev = "cuda"
R = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
def timeit(fn, iters=30):
    for _ in range(3): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
rows = []
def rec(name, us, byt): rows.append((name, us, byt / us / 1e6))
M, D = 16384, 1536
x = R(M, D); mods = R(16, 4 * D)
rec("layernorm_mod (image stream, 1 output)", timeit(lambda: ops.layernorm_mod(x, scale=mods[:, :D], shift=mods[:, D:2*D], rows_per_batch=1024)), M * D * 4)
rec("layernorm_mod (dual: 2 outputs)", timeit(lambda: ops.layernorm_mod(x, scale=mods[:, :D], shift=mods[:, D:2*D], scale2=mods[:, 2*D:3*D], shift2=mods[:, 3*D:], rows_per_batch=1024)), M * D * 6)
dy = R(M, D)
rec("layernorm_mod_bwd (x, dy, dres in; dx out)", timeit(lambda: ops.layernorm_mod_bwd(x, dy, scale0=mods[:, :D], dres=dy, rows_per_batch=1024)), M * D * 8)
rec("gate_mul", timeit(lambda: ops.gate_mul(x, mods[:, :D], 1024)), M * D * 4)
xt = R(128, 4096); wt = R(4096)
rec("rmsnorm_rows (T5, 128 x 4096)", timeit(lambda: ops.rmsnorm_rows(xt, wt)), 128 * 4096 * 4)
sch = FlowMatchEulerDiscreteScheduler(device=dev); sch.set_timesteps(10)
vu, vt, lat = R(8, 16, 64, 64), R(8, 16, 64, 64), R(8, 16, 64, 64)
rec("sde_step (CFG + step + log-prob, 8 x 16x64x64)", timeit(lambda: sde_step_cfg(sch, vu, vt, 4.5, None, lat, 0.8, seed=1, out_dtype=torch.bfloat16, want_mean=False, step_index=2)), 8 * 65536 * 8)
act = R(8, 512, 512, 128); gw = R(128); gb = R(128)
rec("groupnorm_nhwc + SiLU (8 x 512 x 512 x 128)", timeit(lambda: ops.groupnorm_nhwc(act, gw, gb, 32, 1e-6, True), 10), act.numel() * 2 * 3)
P = R(M, D); Q = R(M, 64); acc = torch.zeros(D, 64, device=dev)
rec("gemm_tn (LoRA weight gradient, 16384 x 1536 x 64)", timeit(lambda: ops.gemm_tn(P, Q, acc)), M * D * 2 + M * 128)
rew = torch.randn(768, 2, device=dev); gid = (torch.arange(768, device=dev) // 8).int()
rec("group_advantage (768 x 2, f64)", timeit(lambda: stat_tracking.group_advantage(rew, gid, True)), 768 * 2 * 12)
n = 18_782_208
pf = torch.zeros(n, device=dev); pb = pf.to(torch.bfloat16); gr = torch.randn(n, device=dev); m1 = torch.zeros(n, device=dev); m2 = torch.zeros(n, device=dev)
from adv_grpo_amd import _lib
lib = _lib.load(); ss = torch.ones(1, device=dev)
def adam():
    _lib.check(lib.advgrpo_adamw_step(pf.data_ptr(), pb.data_ptr(), gr.data_ptr(), m1.data_ptr(), m2.data_ptr(), n, 3e-4, 0.9, 0.999, 1e-8, 1e-4, 1, ss.data_ptr(), 1.0, 1.0, _lib.stream_ptr()))
rec("adamw_step (18.8 M LoRA parameters, f32 + bf16 copy)", timeit(adam, 10), n * (4 * 4 * 2 + 2 - 4))
print("| kernel | us | TB/s (algorithmic bytes) | of 8 TB/s |\n|---|---|---|---|")
for name, us, tbs in rows:
    print(f"| {name} | {us:.1f} | {tbs:.2f} | {100 * tbs / 8:.0f} % |")
