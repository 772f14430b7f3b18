"""Data parallelism over env shards: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI).

New in this build -- the reference is single-process.  Design: sampling, chain capture, value /
log-prob precompute, GAE and the rollout buffer are per-env and never communicate.  The PPO update is data parallel over ONE
flat bucket [critic grads | actor_ft grads | statistics], all-reduced (SUM) once per optimiser step in TWO slices:
  * [critic grads] as soon as the critic pipeline's launches are queued -- it runs ahead of the actor's on the library's side
    stream (csrc/api.hip, ppo_impl), so the library calls back (``critic_hook``; include/dppo_hip.h, dppo_ppo_loss_fwd_bwd_dp)
    with that stream and the collective is queued THERE: it travels while the actor's forward / backward still compute;
  * [actor grads | statistics] behind the actor's last gradient kernel on the caller's stream.
torch.distributed's NCCL collectives are stream ordered (the communicator's stream waits for the current stream, the current
stream waits for the collective): no host synchronisation anywhere.  A SUM all-reduce is elementwise, so the two slices give
the bits one whole-bucket all-reduce would (tests/test_parallel_gloo.py holds them to each other).
To stay equal to the single-process algorithm:
  * advantages are normalised with the moments of the GLOBAL minibatch (diffusion_ppo.py:129-130): the per-rank
    (sum, sum of squares, count) of every minibatch of an epoch are pooled in one tiny all-reduce up front and
    handed to the loss kernel (``global_moments``), which then scales every mean by the global count;
  * gradients / statistics are SUM-reduced (each rank already divided by the global count);
  * the KL early-stop (train_ppo_diffusion_agent.py:379-383) reads the all-reduced KL, so every rank breaks together;
  * gradient clipping uses the norm of the reduced gradient (identical on every rank).
The collective payload is 2.75 MB (hopper: 0.54 critic + 2.21 actor): latency-, not bandwidth-bound on xGMI, hence two
slices and no finer buckets.
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
    """Wires a PPODiffusion's flat gradient buffers into one all-reduce bucket, reduced in two slices (module docstring).

    One optimiser step:   model.ppo_update(..., critic_hook=dp.critic_hook)   # critic slice queued from inside the call
                          dp.allreduce_grads()                                # actor slice + statistics (+ the critic slice
                                                                              #  if no hook ran: graphs, conv actors, CPU)
    ``split=False`` keeps the single whole-bucket all-reduce (the A/B reference of the tests and of bench.py)."""

    def __init__(self, model, world_size: int, group=None, split: bool = True):
        self.model, self.world, self.group, self.split = model, world_size, group, split
        self.bucket: Optional[torch.Tensor] = None
        self._critic_done = False
        self.skip_collectives = False  # bench.py: time the step without its all-reduces (what they cost when exposed)
        if world_size > 1:
            a, c = model.actor_ft, model.critic
            na, nc = a.flat_params().numel(), c.flat_params().numel()
            self.na, self.nc = na, nc
            self.bucket = torch.zeros(nc + na + 2 * STATS_SLOTS, dtype=torch.float32, device=a.flat_params().device)
            object.__setattr__(c, "_flat_grad", self.bucket[:nc])
            object.__setattr__(a, "_flat_grad", self.bucket[nc:nc + na])
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

    # ------------------------------------------------------------------ the critic slice, queued from inside the library call
    @property
    def critic_hook(self):
        """What to pass as ``ppo_update(critic_hook=...)``: None with one rank or ``split=False``."""
        return self._on_critic_grads if (self.world > 1 and self.split) else None

    def _on_critic_grads(self, side_stream: int):
        """Called by the library with the stream its critic pipeline was queued on: the slice's all-reduce goes behind it
        there (stream-ordered through torch.distributed: nothing blocks the host), the library then joins that stream into
        the caller's with the rest of the critic pipeline."""
        sl = self.bucket[:self.nc]
        if not self.skip_collectives:
            if sl.is_cuda and side_stream:
                with torch.cuda.stream(torch.cuda.ExternalStream(side_stream, device=sl.device)):
                    allreduce_bucket(sl, self.group)
            else:
                allreduce_bucket(sl, self.group)
        self._critic_done = True

    # ------------------------------------------------------------------ the rest, behind the actor's gradients
    def allreduce_grads(self):
        """SUM-reduce what is still local of [critic grads | actor grads | stats]; afterwards model._stats holds the global
        statistics.  The critic slice is skipped if ``critic_hook`` already sent it in this step."""
        if self.world == 1:
            return
        st = self.model._stats
        tail = self.bucket[self.nc + self.na:]
        rest = self.bucket[self.nc:] if self._critic_done else self.bucket
        self._critic_done = False
        # slots from ADV_MEAN on that every rank wrote in full (divide by world after the SUM): the two advantage statistics,
        # plus the entropy of a Gaussian head (a function of sigma only: gaussian_ppo.PPO_Gaussian.dp_avg_stats = 3)
        n_avg = int(getattr(self.model, "dp_avg_stats", 2))

        def reduce():
            if not self.skip_collectives:
                allreduce_bucket(rest, self.group)
        # float64 statistics travel as (hi, lo) float32 pairs so the KL / loss values keep their precision
        if st.is_cuda:  # one launch each way (a dozen elementwise torch kernels cost ~50 us of a 0.6 ms step)
            from dppo_amd import hip
            lib = hip.load()
            hip.check(lib.dppo_stats_split(st.data_ptr(), tail.data_ptr(), hip.stream()), "dppo_stats_split")
            reduce()
            hip.check(lib.dppo_stats_merge_n(tail.data_ptr(), st.data_ptr(), 1 if self.skip_collectives else self.world, n_avg,
                                             hip.stream()), "dppo_stats_merge_n")
            return
        head = st[:STATS_SLOTS]
        hi = head.float()
        tail[:STATS_SLOTS].copy_(hi)
        tail[STATS_SLOTS:].copy_((head - hi.double()).float())
        reduce()
        head.copy_(tail[:STATS_SLOTS].double() + tail[STATS_SLOTS:].double())
        if not self.skip_collectives:
            st[5:5 + n_avg] /= self.world  # global values every rank wrote, not partial sums
