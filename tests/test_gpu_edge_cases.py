"""GPU: degenerate and ragged sizes through the C ABI -- one row / one token / one group / non-multiples of every tile,
and the error behaviour of bad arguments (an AdvGrpoError with the C-side message, never a silent fallback)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gemm_single_row_and_ragged_everything():
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    for (M, N, K) in [(1, 8, 64), (1, 1536, 1536), (7, 24, 128), (193, 136, 192), (16385, 1544, 64)]:
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
        b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
        out = ops.gemm(a, w, bias=b, act="silu")
        ref = torch.nn.functional.silu(a.float() @ w.float().t() + b.float())
        assert out.shape == (M, N)
        assert (out.float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item()), (M, N, K)


def test_gemm_rejects_bad_arguments_loudly():
    from adv_grpo_amd import _lib, ops
    a = torch.zeros(4, 96, dtype=torch.bfloat16, device="cuda")          # K not a multiple of 64
    w = torch.zeros(8, 96, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(_lib.AdvGrpoError, match="K % 64"):
        ops.gemm(a, w)
    with pytest.raises(_lib.AdvGrpoError):
        ops.gemm_tn(torch.zeros(64, 100, dtype=torch.bfloat16, device="cuda"), torch.zeros(64, 64, dtype=torch.bfloat16, device="cuda"),
                    torch.zeros(100, 64, device="cuda"))                  # N1 not a multiple of 128


def test_attention_one_query_one_key_and_prime_lengths():
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    for (B, H, Sq, Skv, causal) in [(1, 1, 1, 1, False), (2, 3, 1, 67, False), (1, 2, 131, 1, False), (2, 2, 67, 67, True), (1, 4, 257, 129, False)]:
        q = torch.randn(B, Sq, H * 64, device="cuda", generator=g).to(torch.bfloat16)
        k = torch.randn(B, Skv, H * 64, device="cuda", generator=g).to(torch.bfloat16)
        v = torch.randn(B, Skv, H * 64, device="cuda", generator=g).to(torch.bfloat16)
        out = ops.attention(q, k, v, H, causal=causal)
        qh, kh, vh = (t.float().view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
        ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh, is_causal=causal).transpose(1, 2).reshape(B, Sq, H * 64)
        assert ((out.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-3)).item() < 2e-2, (B, H, Sq, Skv, causal)


def test_group_advantage_single_sample_single_group_and_constant_rewards():
    from adv_grpo_amd import stat_tracking
    from oracle import grouping as og
    for rewards, gids, gs in [(np.array([[1.5]]), [0], True), (np.array([[2.0, 3.0]] * 4), [7, 7, 7, 7], False),
                              (np.arange(6, dtype=np.float64).reshape(6, 1), [0, 1, 0, 2, 1, 0], True)]:
        want = og.group_advantages(np.array(gids), rewards.astype(np.float32), gs)
        got = stat_tracking.group_advantage(torch.from_numpy(rewards.astype(np.float32)).cuda(), torch.tensor(gids).cuda(), gs)
        assert np.array_equal(got.cpu().numpy(), want)            # bit-exact, including the zero-std group (-> 0 / 1e-4)


def test_sde_step_single_sample_and_noise_level_zero():
    from adv_grpo_amd.diffusers_patch.sd3_sde_with_logprob import sde_step_cfg
    from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
    sch = FlowMatchEulerDiscreteScheduler(device="cuda"); sch.set_timesteps(10)
    g = torch.Generator().manual_seed(3)
    vu, vt, x = (torch.randn(1, 16, 8, 8, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    nxt, cast, lp, _, _ = sde_step_cfg(sch, vu, vt, 4.5, None, x, 0.0, seed=5, out_dtype=torch.bfloat16, want_mean=False, step_index=3)
    # noise level 0: the step is the deterministic mean (eval loop, TP:303-320); the log-prob is not finite and is ignored upstream
    v = vu.float() + 4.5 * (vt.to(torch.bfloat16).float() - vu.float())
    assert torch.isfinite(nxt).all() and nxt.shape == x.shape and cast.dtype == torch.bfloat16
    nxt2, _, _, _, _ = sde_step_cfg(sch, vu, vt, 4.5, None, x, 0.0, seed=99, out_dtype=torch.bfloat16, want_mean=False, step_index=3)
    assert torch.equal(nxt, nxt2)                                 # no dependence on the noise stream
