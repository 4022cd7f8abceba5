"""Seeded stand-in networks used to pin control flow (test infrastructure).

The reference's rollout / replay functions take the transformer, VAE, scorer and head as
objects, so their control flow, cast order and RNG draw order can be pinned with small
deterministic stand-ins.  The same stand-ins are used by
tests/golden/make_golden.py (driving the reference functions) and by the tests (driving the
oracle and the HIP path)."""
import torch


class StandinVelocity(torch.nn.Module):
    """A tiny velocity field v(x, t, ctx, pooled) with the SD3 transformer call signature
    (sd3_pipeline_with_logprob_fast.py:630-637).  Smooth, batch-independent, dtype-following."""

    def __init__(self, channels=16, ctx_dim=32, pooled_dim=16, seed=7):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.mix = torch.nn.Parameter(torch.randn(channels, channels, generator=g) * 0.3)
        self.ctx = torch.nn.Parameter(torch.randn(ctx_dim, channels, generator=g) * 0.1)
        self.pool = torch.nn.Parameter(torch.randn(pooled_dim, channels, generator=g) * 0.1)

    def forward(self, hidden_states, timestep, encoder_hidden_states, pooled_projections,
                joint_attention_kwargs=None, return_dict=False):
        dt = hidden_states.dtype
        x = hidden_states
        h = torch.einsum("bchw,cd->bdhw", x, self.mix.to(dt))
        c = (encoder_hidden_states.mean(dim=1) @ self.ctx.to(dt)) + pooled_projections @ self.pool.to(dt)
        tt = (timestep.to(dt) / 1000.0).view(-1, 1, 1, 1)
        v = torch.tanh(h) * (0.5 + tt) + c[:, :, None, None] - 0.25 * x.roll(1, dims=-1)
        return (v,)


def standin_vae_decode(z):
    """[B,16,h,w] -> [B,3,8h,8w] in roughly [-1,1] (stand-in for AutoencoderKL.decode, PF:669)."""
    x = torch.nn.functional.interpolate(z[:, :3].float(), scale_factor=8, mode="nearest")
    return torch.tanh(x).to(z.dtype)
