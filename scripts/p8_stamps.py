"""Where a gemm8p tile spends its time: s_memtime stamps of wave 0 of every workgroup (ADVGRPO_P8_STAMPS=1).
stamps: 0 tile start, 1 first items landed, 2 k loop done, 3 next tile located + first k-tile requested, 4 epilogue done,
5 barrier after the epilogue."""
import ctypes, os, sys
import numpy as np, torch
os.environ["ADVGRPO_P8_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import _lib, ops
lib = _lib.load()
lib.advgrpo_dbg_p8_stamps.restype = ctypes.c_int
lib.advgrpo_dbg_p8_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
def rnd(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 6144, 1536)
mode = sys.argv[4] if len(sys.argv) > 4 else "gelu"
a, w, b = rnd(M, K), rnd(N, K) * 0.02, rnd(N)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
kw = dict(bias=b)
if mode == "gelu": kw["act"] = "gelu_tanh"
if mode == "gateres": kw.update(gate=rnd(16, N), gate_rows=M // 16, residual=rnd(M, N))
if mode == "plain": kw = {}
if mode == "rms":      # the fused-QKV class: bias + per-head RMSNorm on the first 2/3 of the columns, row-segment scatter
    H = N // 3 // 64
    rw = rnd(2, 64)
    big = torch.empty(M // 1024 * 1229, N, dtype=torch.bfloat16, device="cuda")
    run = lambda: ops.gemm_grouped([ops.gemm_desc(a, w, bias=b, out=big, seg=(1024, 1229, 0), rms=(rw, 2 * H, H, 1e-6, None))])
else:
    run = lambda: ops.gemm(a, w, out=out, **kw)
for _ in range(3): run()
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
assert lib.advgrpo_dbg_p8_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(256, 8, 8).astype(np.int64)
names = ["wait first items", "k loop", "locate + request next", "epilogue", "barrier"]
for tile in range(8):
    s = st[:, tile, :]
    ok = s[:, 4] > 0
    if not ok.any(): break
    d = np.diff(s[ok][:, :6], axis=1)
    nxt = st[ok][:, tile + 1, 0] - s[ok][:, 0] if tile + 1 < 8 else None
    line = "  ".join(f"{n} {np.median(d[:, i]):8.0f}" for i, n in enumerate(names))
    print(f"tile {tile}: n={ok.sum():3d}  {line}  | tile-to-tile {np.median(nxt[nxt > 0]) if nxt is not None and (nxt > 0).any() else float('nan'):8.0f} ticks")
