#!/usr/bin/env python3
"""Tile choice for the mid-sized GEMMs of the scorers (CLIP-H vision 2056 rows, text 616 rows, ...): every two-stage tile
variant forced in turn through the experiments library (make EXPERIMENTS=1), one child process per variant."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(2056, 1280, 5120), (2056, 5120, 1280), (2056, 3840, 1280), (2056, 1280, 1280), (616, 1024, 4096), (616, 4096, 1024),
          (616, 3072, 1024), (616, 1024, 1024), (3280, 1536, 4096), (4112, 1280, 5120), (4112, 5120, 1280)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from adv_grpo_amd import ops
    for M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        bias = torch.randn(N, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        for _ in range(5):
            ops.gemm(a, w, bias=bias, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.gemm(a, w, bias=bias, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"{M} {N} {K} {us:.1f}")
else:
    res = {}
    for v in (15, 0, 1, 2, 26):
        env = dict(os.environ, ADVGRPO_LIB=os.path.join(ROOT, "adv_grpo_amd", "libadvgrpo_experiments.so"), ADVGRPO_GEMM_FORCE=str(v))
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            p = line.split()
            if len(p) == 4 and p[0].isdigit():
                res.setdefault(tuple(int(x) for x in p[:3]), {})[v] = float(p[3])
    print("shape                 " + "".join(f"{'v' + str(v):>9s}" for v in (15, 0, 1, 2, 26)) + "   (us; 15 = 128x128 8 waves [default], 0 = 128x128 4 waves, 1 = 128x64, 2 = 64x128, 26 = 192x128)")
    for s, r in res.items():
        print(f"{str(s):22s}" + "".join(f"{r.get(v, float('nan')):9.1f}" for v in (15, 0, 1, 2, 26)) + f"   best {min(r, key=r.get)}  {2e-6 * s[0] * s[1] * s[2] / min(r.values()):.0f} TF")
