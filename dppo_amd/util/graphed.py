"""One PPO minibatch update (fused loss forward/backward -> [gradient all-reduce] -> 2 x AdamW -> repack) captured as
hipGraphs and replayed: the ~38 dependent launches of a step cost ~5 us of dispatch each when issued one by one and
~1.7 us inside a graph (measured on MI355X / ROCm 7).  Reference loop: agent/finetune/train_ppo_diffusion_agent.py:306-383.

Everything a replay needs is static: the rollout buffers, workspaces, parameters, gradients and optimiser state keep
their addresses; the minibatch indices (and, under data parallelism, the pooled advantage moments) are copied into fixed
buffers before each replay; AdamW reads its step count and learning rate from device memory.  With more than one rank the
all-reduce stays outside the graphs (one graph up to the gradients, one from the optimiser on).
"""
from __future__ import annotations

from typing import Optional

import torch

from dppo_amd.util.optim import FlatAdamW, step_and_repack


class GraphedUpdate:
    def __init__(self, model, actor_opt: FlatAdamW, critic_opt: FlatAdamW, dp, rollout, batch_size: int,
                 reward_horizon: int, update_actor: bool = True, max_norm: Optional[float] = None,
                 n_time: Optional[int] = None, warmup: int = 2):
        """rollout = (obs_k, chains_k, returns_k, values_k, adv_k, logprobs_k): the device-resident buffers of
        ``PPODiffusion.ppo_update``; they must keep their storage for the lifetime of this object."""
        self.model, self.oa, self.oc, self.dp = model, actor_opt, critic_opt, dp
        self.rollout, self.rh = rollout, reward_horizon
        self.update_actor, self.max_norm, self.n_time = update_actor, max_norm, n_time
        dev = rollout[0].device
        self.inds = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        self.world = getattr(dp, "world", 1)
        self.moments = torch.zeros(3, dtype=torch.float64, device=dev) if self.world > 1 else None
        self._graphs = None
        self._warmup = warmup

    # the two halves of a step; eager and captured runs execute exactly this code
    def _grads(self):
        return self.model.ppo_update(*self.rollout, self.inds, reward_horizon=self.rh, global_moments=self.moments)

    def _apply(self):
        step_and_repack(self.model, self.oa, self.oc, update_actor=self.update_actor, max_norm=self.max_norm,
                        n_time=self.n_time)

    def _snapshot(self):
        nets = (self.model.actor_ft, self.model.critic)
        return ([n.flat_params().clone() for n in nets],
                [(o.exp_avg.clone(), o.exp_avg_sq.clone(), o._step_dev.clone(), o.step_count) for o in (self.oa, self.oc)])

    def _restore(self, snap):
        from dppo_amd.model.common.mlp import pack_pair
        params, opts = snap
        for n, p in zip((self.model.actor_ft, self.model.critic), params):
            n.flat_params().copy_(p)
            n.mark_updated()
        for o, (m, v, st, cnt) in zip((self.oa, self.oc), opts):
            o.exp_avg.copy_(m), o.exp_avg_sq.copy_(v), o._step_dev.copy_(st)
            o.step_count = cnt
        pack_pair(self.model.critic, 0, self.model.actor_ft,
                  self.model.denoising_steps if self.n_time is None else self.n_time, self.model.prec)

    def _capture(self):
        # warm-up runs (first-use allocations, kernel attributes, library side streams) are real updates: they run on a
        # snapshot -- parameters, AdamW moments and step counters are put back afterwards, so the first step() is ONE
        # optimiser step on its minibatch, like every later one
        snap = self._snapshot() if self._warmup > 0 else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self._warmup):
                self._grads()
                self.dp.allreduce_grads()
                self._apply()
            if snap is not None:
                self._restore(snap)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._captured_for = (self.update_actor, self.max_norm)
        if self.world == 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._grads()
                self._apply()
            self._graphs = (g,)
        else:
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self._grads()
            with torch.cuda.graph(g2):
                self._apply()
            self._graphs = (g1, g2)
        for o in (self.oa, self.oc):
            o.step_count -= 1  # the captured call counted itself but did not run

    def step(self, inds: torch.Tensor, global_moments: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One update on minibatch ``inds``; returns the model's device statistics tensor (see hip.STAT_*)."""
        self.inds.copy_(inds)
        if self.moments is not None:
            self.moments.copy_(global_moments)
        self.oa.sync_lr()
        self.oc.sync_lr()
        if self._graphs is not None and self._captured_for != (self.update_actor, self.max_norm):
            self._graphs = None  # e.g. the critic warm-up iterations ended: update_actor is frozen into a capture
        if self._graphs is None:
            self._capture()
        if len(self._graphs) == 1:
            self._graphs[0].replay()
        else:
            self._graphs[0].replay()
            self.dp.allreduce_grads()
            self._graphs[1].replay()
        for o in (self.oa, self.oc):
            o.step_count += 1
        return self.model._stats
