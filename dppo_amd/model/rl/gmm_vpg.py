"""Policy gradient with a mixture-of-Gaussians policy.  Mirrors ``dppo/model/rl/gmm_vpg.py:6-46`` (reference ``VPG_GMM``)."""
from __future__ import annotations

import ctypes as C
import math

import torch

from dppo_amd import hip
from dppo_amd.model.common.gmm import GMMModel


class VPG_GMM(GMMModel):
    def __init__(self, actor, critic, **kwargs):
        super().__init__(network=actor, **kwargs)
        self.actor_ft = actor
        self.critic = critic.to(self.device)

    @torch.no_grad()
    def get_logprobs(self, cond, actions):
        """(log p(a) (B,) summed over Ta*Da inside each component, entropy estimate, mean std) -- reference :33-43 through
        gmm.py:48-86.  Inference only; the differentiable evaluation is fused into ``PPO_GMM.loss``."""
        state = cond["state"]
        hip.require_gpu(state, "VPG_GMM.get_logprobs")
        net = self.actor_ft
        B, dev = state.shape[0], state.device
        AF = net.action_dim * net.horizon_steps
        obs = state.reshape(B, -1).contiguous().float()
        act = actions.reshape(B, AF).contiguous().float()
        out = torch.empty(B, device=dev)
        ws = self._workspace(net, None, B, dev)
        cfg = net.gmm_cfg()
        hip.check(hip.load().dppo_gmm_logprob(*self._net_args(net), C.byref(cfg), net.logvar_ptr(), obs.data_ptr(), act.data_ptr(), B,
                                              out.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream()), "dppo_gmm_logprob")
        return out, None, None  # entropy / std are statistics of PPO_GMM.loss (they need the mixture weights of every sample)

    def loss(self, obs, chains, reward):
        raise NotImplementedError
