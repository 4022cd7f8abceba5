"""Token-contracted GEMMs of the LoRA weight gradients at the config-2 shapes (CFG batch 16: 16384 image rows, 3280 text rows, D = 1536):
the grouped launches of round 5 (one per adapter group) beside the per-adapter launches they replace.  Bytes = one pass over P and Q."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
D, Mi, Mt, S, Ni, Nt = 1536, 16384, 3280, 1229, 1024, 205
rnd = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
def timed(fn, iters=50):
    for _ in range(5): fn()
    ts = []
    for _ in range(5):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return min(ts)
dqkv = rnd(16 * S, 3 * D); nx = rnd(Mi, D); nc = rnd(Mt, D)
t_i, t_t, u_i, u_t = rnd(Mi, 192), rnd(Mt, 192), rnd(Mi, 192), rnd(Mt, 192)
gB = [torch.zeros(D, 64, dtype=torch.float32, device="cuda") for _ in range(6)]
gA = [torch.zeros(64, D, dtype=torch.float32, device="cuda") for _ in range(6)]
seg_i, seg_t = (Ni, S, 0), (Nt, S, Ni)
def dB_grouped():
    ops.gemm_tn_grouped([ops.tn_desc(dqkv[:, j * D:(j + 1) * D], t_i[:, 64 * j:64 * j + 64], gB[j], alpha=2.0, M=Mi, p_seg=seg_i) for j in range(3)] +
                        [ops.tn_desc(dqkv[:, j * D:(j + 1) * D], t_t[:, 64 * j:64 * j + 64], gB[3 + j], alpha=2.0, M=Mt, p_seg=seg_t) for j in range(3)])
def dB_single():
    for j in range(3):
        ops.gemm_tn(dqkv[:, j * D:(j + 1) * D], t_i[:, 64 * j:64 * j + 64], gB[j], alpha=2.0, M=Mi, p_seg=seg_i)
        ops.gemm_tn(dqkv[:, j * D:(j + 1) * D], t_t[:, 64 * j:64 * j + 64], gB[3 + j], alpha=2.0, M=Mt, p_seg=seg_t)
def dA_grouped():
    ops.gemm_tn_grouped([ops.tn_desc(nx, u_i[:, 64 * j:64 * j + 64], gA[j], alpha=2.0, M=Mi, transpose_out=True) for j in range(3)] +
                        [ops.tn_desc(nc, u_t[:, 64 * j:64 * j + 64], gA[3 + j], alpha=2.0, M=Mt, transpose_out=True) for j in range(3)])
def dA_single():
    for j in range(3):
        ops.gemm_tn(nx, u_i[:, 64 * j:64 * j + 64], gA[j], alpha=2.0, M=Mi, transpose_out=True)
        ops.gemm_tn(nc, u_t[:, 64 * j:64 * j + 64], gA[3 + j], alpha=2.0, M=Mt, transpose_out=True)
def all_narrow():          # dB and dA of the group as twelve problems in ONE launch (X is read three times, twice of them from L2 / the Infinity Cache)
    ops.gemm_tn_grouped([ops.tn_desc(dqkv[:, j * D:(j + 1) * D], t_i[:, 64 * j:64 * j + 64], gB[j], alpha=2.0, M=Mi, p_seg=seg_i) for j in range(3)] +
                        [ops.tn_desc(dqkv[:, j * D:(j + 1) * D], t_t[:, 64 * j:64 * j + 64], gB[3 + j], alpha=2.0, M=Mt, p_seg=seg_t) for j in range(3)] +
                        [ops.tn_desc(nx, u_i[:, 64 * j:64 * j + 64], gA[j], alpha=2.0, M=Mi, transpose_out=True) for j in range(3)] +
                        [ops.tn_desc(nc, u_t[:, 64 * j:64 * j + 64], gA[3 + j], alpha=2.0, M=Mt, transpose_out=True) for j in range(3)])
bytes_dB = 3 * (Mi + Mt) * (D + 64) * 2
for name, fn, b in (("dB q|k|v, both streams: 6 problems, one grouped launch", dB_grouped, bytes_dB), ("dB: 6 launches + 6 reduces (round 4)", dB_single, bytes_dB),
                    ("dA q|k|v, both streams: 6 problems (X shared by three), one launch", dA_grouped, 3 * (Mi + Mt) * (D + 64) * 2), ("dA: 6 launches + 6 reduces (round 4)", dA_single, 3 * (Mi + Mt) * (D + 64) * 2),
                    ("dB + dA of the group: twelve problems, ONE launch (the product's)", all_narrow, 2 * bytes_dB)):
    us = timed(fn)
    print(f"{name:62s} {us:8.1f} us   {b / us / 1e6:5.2f} TB/s of its own operand bytes")
