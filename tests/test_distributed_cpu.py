"""CPU, world size 2 over gloo: the N>1 host path -- prompt sharding by the k-repeat sampler, the packed reward /
group-id all-gather (rank-major), the un-gather slice and the rank-identical gate scalar.  The advantage arithmetic
itself is the GPU kernel's job (tests/test_gpu_leaf_kernels.py); here the oracle stands in for it."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adv_grpo_amd import distributed as D
    from adv_grpo_amd.sampler import DistributedKRepeatSampler
    from oracle import grouping
    G, T, nb = 4, 2, 3                      # images per rank-batch, train steps, batches per epoch
    k = 2                                   # a prompt group spans k = 2 ranks (num_image_per_prompt = 2 * mini)
    sampler = DistributedKRepeatSampler(range(1000), 1, k, world, rank, seed=42)
    rewards, gids = [], []
    for i in range(nb):
        sampler.set_epoch(i)
        idx = next(iter(sampler))[0]
        g = torch.Generator().manual_seed(1000 * idx + rank)          # rank-dependent rewards
        rewards.append(torch.rand(G, generator=g).unsqueeze(1).repeat(1, T))
        gids.append(torch.full((G,), idx, dtype=torch.int32))
    rewards, gids = torch.cat(rewards), torch.cat(gids)
    all_r, all_g = D.gather_rewards(rewards, gids)
    assert all_r.shape == (world * nb * G, T) and all_g.dtype == torch.int32
    # rank-major: my rows sit at [rank*N_loc, (rank+1)*N_loc)
    assert torch.equal(all_r[rank * nb * G:(rank + 1) * nb * G], rewards)
    assert torch.equal(all_g[rank * nb * G:(rank + 1) * nb * G], gids)
    adv = grouping.group_advantages(all_g.numpy(), all_r.numpy(), True)
    mine = D.ungather(torch.from_numpy(adv))
    assert mine.shape == (nb * G, T)
    gate = D.all_mean(rewards[:, 0])
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.concatenate([all_r.numpy().ravel(), all_g.numpy().ravel(),
                                                                   adv.ravel(), [gate.item()]]))
    np.save(os.path.join(out_dir, f"mine{rank}.npy"), mine.numpy())
    dist.destroy_process_group()


def test_two_rank_gather_advantage_ungather(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(a, b)             # every rank holds identical gathered data, advantages and gate scalar
    m0, m1 = np.load(tmp_path / "mine0.npy"), np.load(tmp_path / "mine1.npy")
    n_all = 2 * 3 * 4
    adv = a[n_all * 2 + n_all: n_all * 2 + n_all + n_all * 2].reshape(n_all, 2)
    assert np.array_equal(np.concatenate([m0, m1]), adv)
    gids = a[n_all * 2: n_all * 2 + n_all].reshape(2, 3, 4)[:, :, 0]
    assert np.array_equal(gids[0], gids[1])  # k = 2: both ranks draw the same prompt each iteration -> groups of 8
    assert len(set(gids[0])) == 3


def _worker_update(rank, world, port, out_dir):
    """The update half's exchange: ranks hold DIFFERENT gradients (their own prompt groups); after the
    all-reduce + / world every rank feeds the optimizer the same vector, so identical Adam steps keep the replicated
    LoRA / head parameters identical without ever broadcasting them again.  The trainable state itself starts from rank 0's
    values (broadcast at construction)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adv_grpo_amd import distributed as D
    g = torch.Generator().manual_seed(100 + rank)                 # rank-dependent initial state and gradients
    params = torch.randn(1000, generator=g)
    head = torch.randn(77, generator=g)
    D.broadcast_state([params, head])
    lora_grad = torch.randn(1000, generator=g)
    head_grad = torch.randn(77, generator=g)
    mine = lora_grad.clone()
    D.average_gradients(lora_grad)                                # G-step: trainer.g_step before optimizer_step
    D.average_gradients(head_grad)                                # D-step: the closure handed to train_dino / train_pickscore
    # a plain Adam step on the averaged gradient: identical inputs -> identical parameters on both ranks
    opt = torch.optim.AdamW([torch.nn.Parameter(params.clone())], lr=3e-4, weight_decay=1e-4)
    opt.param_groups[0]["params"][0].grad = lora_grad
    opt.step()
    np.save(os.path.join(out_dir, f"u{rank}.npy"), np.concatenate([params.numpy(), head.numpy(), lora_grad.numpy(), head_grad.numpy(),
                                                                   opt.param_groups[0]["params"][0].detach().numpy()]))
    np.save(os.path.join(out_dir, f"g{rank}.npy"), mine.numpy())
    dist.destroy_process_group()


def test_two_rank_gradient_average_and_state_broadcast(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker_update, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "u0.npy"), np.load(tmp_path / "u1.npy")
    assert np.array_equal(a, b)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert not np.array_equal(g0, g1)
    np.testing.assert_allclose(a[1077:2077], (g0 + g1) / 2, rtol=1e-6)     # the averaged LoRA gradient
    # rank 0's initial state everywhere
    g = torch.Generator().manual_seed(100)
    np.testing.assert_array_equal(a[:1000], torch.randn(1000, generator=g).numpy())


def _worker_info(rank, world, port, out_dir):
    """TP:1179-1183: the diagnostics of an optimizer step are averaged over ranks (accelerator.reduce(info, "mean")) before rank 0 logs them."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adv_grpo_amd import distributed as D
    info = {"approx_kl": torch.tensor(0.25 * (rank + 1)), "clipfrac": torch.tensor(float(rank)), "policy_loss": -1.5 + rank, "loss": torch.tensor(2.0)}
    out = D.reduce_mean(info)
    assert sorted(out) == sorted(info)
    np.save(os.path.join(out_dir, f"info{rank}.npy"), np.array([float(out[k]) for k in sorted(out)]))
    dist.destroy_process_group()


def test_two_rank_training_diagnostics_are_rank_means(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker_info, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "info0.npy"), np.load(tmp_path / "info1.npy")
    assert np.array_equal(a, b)                                   # every rank holds the node's mean
    # keys sorted: approx_kl, clipfrac, loss, policy_loss
    assert np.allclose(a, [0.375, 0.5, 2.0, -1.0])


def test_reduce_mean_is_the_identity_without_a_process_group():
    from adv_grpo_amd import distributed as D
    info = {"a": torch.tensor(1.0), "b": 2.0}
    out = D.reduce_mean(info)
    assert out == info and out is not info
