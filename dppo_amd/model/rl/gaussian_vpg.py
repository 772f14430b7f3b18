"""Policy gradient with a Gaussian policy.  Mirrors ``dppo/model/rl/gaussian_vpg.py:12-68`` (reference ``VPG_Gaussian``):
``actor_ft`` is the trained network, ``actor`` a frozen copy, ``critic`` the value net."""
from __future__ import annotations

import copy
import ctypes as C

import torch

from dppo_amd import hip
from dppo_amd.model.common.gaussian import GaussianModel


class VPG_Gaussian(GaussianModel):
    def __init__(self, actor, critic, **kwargs):
        super().__init__(network=actor, **kwargs)
        self.critic = critic.to(self.device)
        self.actor_ft = actor
        self.actor = copy.deepcopy(actor)
        for p in self.actor.parameters():
            p.requires_grad = False

    @torch.no_grad()
    def forward(self, cond, deterministic=False, use_base_policy=False, noise=None):
        return super().forward(cond=cond, deterministic=deterministic,
                               network_override=self.actor if use_base_policy else None, noise=noise)

    @torch.no_grad()
    def get_logprobs(self, cond, actions, use_base_policy=False):
        """(log_prob (B,) = mean over Ta*Da of the element log-probs, entropy, mean std) -- reference :46-62.  Inference
        only (the rollout precompute); the differentiable evaluation is fused into ``PPO_Gaussian.loss``."""
        state = cond["state"]
        hip.require_gpu(state, "VPG_Gaussian.get_logprobs")
        net = self.actor if use_base_policy else self.actor_ft
        B, dev = state.shape[0], state.device
        AF = net.action_dim * net.horizon_steps
        obs = net.encode_obs(cond) if getattr(net, "is_vision", False) else state.reshape(B, -1).contiguous().float()
        act = actions.reshape(B, AF).contiguous().float()
        lib, d = hip.load(), net.net_desc()
        cfg = net.gaussian_cfg(randn_clip=self.randn_clip_value)
        out = torch.empty(B, device=dev)
        wsb = lib.dppo_gaussian_workspace_bytes(C.byref(d), None, self.prec, B)
        ws = self._ws_g.get(wsb, dev)
        hip.check(lib.dppo_gaussian_logprob(
            C.byref(d), self.prec, net.flat_params().data_ptr(), net.packed(self.prec, 0).data_ptr(), C.byref(cfg),
            net.logvar_ptr(), obs.data_ptr(), act.data_ptr(), B, out.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream()),
            "dppo_gaussian_logprob")
        entropy, std = self._entropy_and_std(net)
        return out, entropy, std

    @staticmethod
    def _entropy_and_std(net):
        """dist.entropy().mean() and dist.scale.mean(): functions of the (state-independent) sigma alone."""
        import math
        if net.learn_fixed_std:
            sigma = torch.exp(0.5 * torch.clamp(net.logvar.detach(), net.logvar_min, net.logvar_max))
            return (0.5 + 0.5 * math.log(2 * math.pi)) + torch.log(sigma).mean(), sigma.mean()
        s = torch.tensor(float(net.fixed_std))
        return (0.5 + 0.5 * math.log(2 * math.pi)) + torch.log(s), s

    def loss(self, obs, actions, reward):
        raise NotImplementedError
