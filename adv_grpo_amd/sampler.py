"""Prompt sharding across ranks: the k-repeat sampler.

Mirror of DistributedKRepeatSampler, scripts/train_sd3_fast_pickscore.py:87-129: every rank
draws the same seeded permutation (torch CPU generator, so the index stream is identical to
the reference's), each of the m = n*b/k prompts is repeated k times, shuffled, and rank r
takes slice r.  Pure host index arithmetic; nothing to accelerate."""
import torch


class DistributedKRepeatSampler(torch.utils.data.Sampler):
    def __init__(self, dataset, batch_size, k, num_replicas, rank, seed=0):
        self.dataset = dataset
        self.batch_size = batch_size
        self.k = k
        self.num_replicas = num_replicas
        self.rank = rank
        self.seed = seed
        self.total_samples = num_replicas * batch_size
        assert self.total_samples % k == 0, \
            f"k can not divide n*b, k{k}-num_replicas{num_replicas}-batch_size{batch_size}"
        self.m = self.total_samples // k
        self.epoch = 0

    def all_ranks(self):
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        picked = torch.randperm(len(self.dataset), generator=g)[:self.m].tolist()
        repeated = [i for i in picked for _ in range(self.k)]
        order = torch.randperm(len(repeated), generator=g).tolist()
        flat = [repeated[i] for i in order]
        b = self.batch_size
        return [flat[r * b:(r + 1) * b] for r in range(self.num_replicas)]

    def __iter__(self):
        while True:
            yield self.all_ranks()[self.rank]

    def set_epoch(self, epoch):
        self.epoch = epoch
