"""Same-process A/B of the head-dim-64 attention forward of two builds (default: libadvgrpo_base.so vs libadvgrpo_hip.so): both libraries are
loaded into ONE process and alternate, R rounds of N launches each.  (Alternating PROCESSES, scripts/ab.sh, carries a position effect: identical
builds read 164.9 / 163.6 / 164.7 / 163.3 us in four consecutive processes.)   Usage: attention_ab_inprocess.py [base.so [new.so]]"""
import ctypes, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from adv_grpo_amd import _lib, ops
paths = [sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "adv_grpo_amd", "libadvgrpo_base.so"),
         sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "adv_grpo_amd", "libadvgrpo_hip.so")]
libs = []
for p in paths:
    lib = ctypes.CDLL(p)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    libs.append(lib)
N, R = 100, 6


def bench(B, H, S):
    D = 64
    qkv = torch.randn(B, S, 3 * H * D, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
    outs = [torch.empty(B, S, H * D, dtype=torch.bfloat16, device="cuda") for _ in libs]
    ts = [[], []]
    for r in range(R):
        for i in ((0, 1) if r % 2 == 0 else (1, 0)):
            _lib._lib = libs[i]
            for _ in range(10):
                ops.attention(q, k, v, H, out=outs[i])
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(N):
                ops.attention(q, k, v, H, out=outs[i])
            e.record(); torch.cuda.synchronize()
            ts[i].append(s.elapsed_time(e) / N * 1e3)
    assert torch.equal(outs[0], outs[1])
    for i in (0, 1):
        t = sorted(ts[i])
        print(f"B={B} H={H} S={S} {os.path.basename(paths[i])}: min {t[0]:.1f} median {t[len(t) // 2]:.1f} us   [{' '.join(f'{x:.1f}' for x in ts[i])}]")


bench(16, 24, 1229); bench(16, 24, 1024); bench(8, 12, 1370); bench(16, 24, 4301)
