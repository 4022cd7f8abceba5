"""Where a workgroup of attention_fwd_pipe_kernel spends its life (ATT_ABL=256 build: scripts/ablate_attention.sh build 256).
Seven s_memrealtime stamps (100 MHz) per workgroup: entry, Q arrived, first K/V tiles arrived, reference maximum done, main loop done,
tail done, stores issued; plus HW_ID / XCC_ID.  Prints phase durations and how many workgroups are alive over the launch."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import ops
for S in [int(a) for a in sys.argv[1:]] or [1229, 1024]:
    B, H, D = 16, 24, 64
    qkv = torch.randn(B, S, 3 * H * D, device='cuda').to(torch.bfloat16)
    q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
    lse = torch.zeros(B, H, S, dtype=torch.float32, device='cuda')
    for _ in range(3):
        ops.attention(q, k, v, H, lse=lse)
    nwg = B * H * ((S + 127) // 128)
    lse2, lse3 = torch.zeros_like(lse), torch.zeros_like(lse)
    ops.attention(q, k, v, H, lse=lse)
    ops.attention(q, k, v, H, lse=lse2)                  # right behind the last one: the gap between two kernels of one stream
    torch.cuda.synchronize()
    r1 = lse.view(-1)[:nwg * 16].view(torch.int64).view(nwg, 8).cpu().numpy()
    r2 = lse2.view(-1)[:nwg * 16].view(torch.int64).view(nwg, 8).cpu().numpy()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attention(q, k, v, H, lse=lse3)
    e1.record(); torch.cuda.synchronize()
    print(f"S={S}: back-to-back {e0.elapsed_time(e1) * 100:.1f} us per launch (this build, stamps on); last end of launch 1 -> first entry of "
          f"launch 2: {(r2[:, 0].min() - r1[:, 6].max()) / 100.0:.2f} us; first entry 1 -> first entry 2: {(r2[:, 0].min() - r1[:, 0].min()) / 100.0:.1f} us")
    raw = lse.view(-1)[:nwg * 16].view(torch.int64).view(nwg, 8).cpu().numpy()
    t = (raw[:, :7] - raw[:, 0].min()) / 100.0          # us since the first workgroup's entry
    hw = raw[:, 7] & 0xffffffff; xcc = raw[:, 7] >> 32
    cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | ((xcc & 0xf) << 8)     # CU_ID, SE_ID, XCC
    names = ["Q load", "first K/V", "S0 + max", "main loop", "tail", "check+epilogue"]
    print(f"   {nwg} workgroups on {len(np.unique(cu))} CUs, launch span {t[:, 6].max():.1f} us")
    d = np.diff(t, axis=1)
    for i, n in enumerate(names):
        print(f"   {n:16s} mean {d[:, i].mean():6.2f} us   p10 {np.percentile(d[:, i], 10):6.2f}   p90 {np.percentile(d[:, i], 90):6.2f}")
    life = t[:, 6] - t[:, 0]
    print(f"   lifetime         mean {life.mean():6.2f} us   p10 {np.percentile(life, 10):6.2f}   p90 {np.percentile(life, 90):6.2f}; tiles {(S + 63) // 64}")
    # by start order: rounds
    order = np.argsort(t[:, 0])
    per = len(order) // 5 if S == 1229 else len(order) // 4
    for r in range(0, len(order), per):
        sel = order[r:r + per]
        print(f"   start-order {r:5d}..: start {t[sel, 0].mean():7.2f} (p10 {np.percentile(t[sel, 0], 10):7.2f} p90 {np.percentile(t[sel, 0], 90):7.2f})  life {life[sel].mean():6.2f}  main {d[sel, 3].mean():6.2f}  pre {(t[sel, 3] - t[sel, 0]).mean():5.2f}  post {(t[sel, 6] - t[sel, 4]).mean():5.2f}")
    # alive count
    grid = np.arange(0, t[:, 6].max(), 5.0)
    alive = [(int(((t[:, 0] <= g) & (t[:, 6] > g)).sum()), int(((t[:, 3] <= g) & (t[:, 4] > g)).sum())) for g in grid]
    print("   alive / in main loop every 5 us:", " ".join(f"{a}/{m}" for a, m in alive))
    # gap between a workgroup's end and the next start on the same CU slot: per CU, sort events
    gaps = []
    for c in np.unique(cu)[:64]:
        sel = np.where(cu == c)[0]
        ends = np.sort(t[sel, 6]); starts = np.sort(t[sel, 0])
        # k-th end vs (k+3)-th start (three resident workgroups)
        for kk in range(len(sel) - 3):
            gaps.append(starts[kk + 3] - ends[kk])
    gaps = np.array(gaps)
    print(f"   end -> next start on the CU: mean {gaps.mean():.2f} us, p10 {np.percentile(gaps, 10):.2f}, p90 {np.percentile(gaps, 90):.2f}")
