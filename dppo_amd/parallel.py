"""Data parallelism over env shards: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI).

New in this build -- the reference is single-process.  Design: sampling, chain capture, value /
log-prob precompute, GAE and the rollout buffer are per-env and never communicate.  The PPO update is data parallel
with ONE all-reduce per optimiser step over a single flat bucket [actor_ft grads | critic grads | statistics].
To stay equal to the single-process algorithm:
  * advantages are normalised with the moments of the GLOBAL minibatch (diffusion_ppo.py:129-130): the per-rank
    (sum, sum of squares, count) of every minibatch of an epoch are pooled in one tiny all-reduce up front and
    handed to the loss kernel (``global_moments``), which then scales every mean by the global count;
  * gradients / statistics are SUM-reduced (each rank already divided by the global count);
  * the KL early-stop (train_ppo_diffusion_agent.py:379-383) reads the all-reduced KL, so every rank breaks together;
  * gradient clipping uses the norm of the reduced gradient (identical on every rank).
The collective payload is 2.75 MB (hopper): latency-, not bandwidth-bound on xGMI, hence one bucket.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

STATS_SLOTS = 8


def pool_minibatch_moments(adv_k: torch.Tensor, minibatches: Sequence[torch.Tensor], Kft: int,
                           group=None) -> torch.Tensor:
    """(len(minibatches), 3) float64: sum adv, sum adv^2, count of every minibatch, summed over ranks."""
    rows = []
    for inds in minibatches:
        a = adv_k[torch.div(inds, Kft, rounding_mode="floor")].double()
        rows.append(torch.stack([a.sum(), (a * a).sum(), torch.tensor(float(a.numel()), dtype=torch.float64,
                                                                       device=a.device)]))
    m = torch.stack(rows).contiguous()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)
    return m


def allreduce_bucket(bucket: torch.Tensor, group=None) -> torch.Tensor:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    return bucket


class DataParallel:
    """Wires a PPODiffusion's flat gradient buffers into one all-reduce bucket."""

    def __init__(self, model, world_size: int, group=None):
        self.model, self.world, self.group = model, world_size, group
        self.bucket: Optional[torch.Tensor] = None
        if world_size > 1:
            a, c = model.actor_ft, model.critic
            na, nc = a.flat_params().numel(), c.flat_params().numel()
            self.na, self.nc = na, nc
            self.bucket = torch.zeros(na + nc + 2 * STATS_SLOTS, dtype=torch.float32, device=a.flat_params().device)
            object.__setattr__(a, "_flat_grad", self.bucket[:na])
            object.__setattr__(c, "_flat_grad", self.bucket[na:na + nc])
            # identical start on every rank (same seed already; broadcast makes it unconditional)
            dist.broadcast(a.flat_params(), src=0, group=group)
            dist.broadcast(c.flat_params(), src=0, group=group)
            dist.broadcast(model.actor.flat_params(), src=0, group=group)
            for m in (a, c, model.actor):
                m.mark_updated()

    def minibatch_moments(self, adv_k, minibatches, Kft) -> Optional[List[torch.Tensor]]:
        if self.world == 1:
            return None
        m = pool_minibatch_moments(adv_k, minibatches, Kft, self.group)
        return [m[i] for i in range(m.shape[0])]

    def allreduce_grads(self):
        """SUM-reduce [actor grads | critic grads | stats]; afterwards model._stats holds the global statistics."""
        if self.world == 1:
            return
        st = self.model._stats
        tail = self.bucket[self.na + self.nc:]
        # float64 statistics travel as (hi, lo) float32 pairs so the KL / loss values keep their precision
        if st.is_cuda:  # one launch each way (a dozen elementwise torch kernels cost ~50 us of a 0.6 ms step)
            from dppo_amd import hip
            lib = hip.load()
            hip.check(lib.dppo_stats_split(st.data_ptr(), tail.data_ptr(), hip.stream()), "dppo_stats_split")
            allreduce_bucket(self.bucket, self.group)
            hip.check(lib.dppo_stats_merge(tail.data_ptr(), st.data_ptr(), self.world, hip.stream()), "dppo_stats_merge")
            return
        hi = st.float()
        tail[:STATS_SLOTS].copy_(hi)
        tail[STATS_SLOTS:].copy_((st - hi.double()).float())
        allreduce_bucket(self.bucket, self.group)
        st.copy_(tail[:STATS_SLOTS].double() + tail[STATS_SLOTS:].double())
        st[5:7] /= self.world  # adv mean / std are global values every rank wrote, not partial sums
