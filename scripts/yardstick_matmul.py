"""Yardstick only (NOT part of the product path): torch.matmul (hipBLASLt) on the MMDiT GEMM shapes, to know how far
the hand-written kernels are from the vendor library on the same box."""
import torch
def bench(M, N, K, iters=20):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(3): torch.matmul(a, w.t())
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): torch.matmul(a, w.t())
    e.record(); torch.cuda.synchronize()
    return 2 * M * N * K / (s.elapsed_time(e) / iters) / 1e9
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (16384, 1536, 1536), (16384, 4608, 1536), (16384, 6144, 1536), (16384, 1536, 6144), (3280, 4608, 1536)]
print("hipBLASLt via torch.matmul:", " ".join(f"{bench(*s):7.0f}" for s in shapes))
