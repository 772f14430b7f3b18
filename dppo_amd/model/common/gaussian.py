"""Gaussian policy parameterisation.  Mirrors ``dppo/model/common/gaussian.py:14-121`` (reference ``GaussianModel``):
constructor surface, checkpoint loading, ``forward`` (sampling with the draw clipped to +-randn_clip_value sigma).
Sampling is one trunk forward + one epilogue kernel (``dppo_gaussian_sample``); noise is drawn in the kernel (Philox keyed
from torch's CPU generator) unless a recorded tensor is passed (parity tests)."""
from __future__ import annotations

import ctypes as C
import logging

import torch

from dppo_amd import hip

log = logging.getLogger(__name__)


class GaussianModel(torch.nn.Module):
    def __init__(self, network, horizon_steps, network_path=None, device="cuda:0", randn_clip_value=10,
                 tanh_output=False, precision=None):
        super().__init__()
        if tanh_output:
            raise NotImplementedError("dppo_amd: tanh applied to the SAMPLED action (SAC / RLPD) is out of scope")
        self.device = device
        self.network = network.to(device)
        if network_path is not None:
            checkpoint = torch.load(network_path, map_location=self.device, weights_only=True)  # safe loader only
            self.load_state_dict(checkpoint["model"], strict=False)
            log.info("Loaded actor from %s", network_path)
        self.horizon_steps = horizon_steps
        self.randn_clip_value = randn_clip_value
        self.tanh_output = tanh_output
        self.prec = hip.PREC_BY_NAME[precision] if precision is not None else network.prec
        object.__setattr__(self, "_ws_g", hip.Workspace())

    def loss(self, true_action, cond, ent_coef):
        raise NotImplementedError("dppo_amd: supervised Gaussian pre-training (-log p - ent_coef * entropy) is out of scope")

    @torch.no_grad()
    def _sample(self, net, cond, deterministic, noise=None, want_mean=False):
        state = cond["state"]
        hip.require_gpu(state, type(self).__name__ + ".forward")
        B, dev = state.shape[0], state.device
        AF = net.action_dim * net.horizon_steps
        obs = net.encode_obs(cond) if getattr(net, "is_vision", False) else state.reshape(B, -1).contiguous().float()
        lib, d = hip.load(), net.net_desc()
        cfg = net.gaussian_cfg(deterministic=deterministic, randn_clip=self.randn_clip_value)
        if noise is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())  # CPU generator: no device sync
            cfg.seed_lo, cfg.seed_hi = seed & 0xFFFFFFFF, seed >> 32
        else:
            noise = noise.reshape(B, AF).contiguous().float()
        actions = torch.empty(B, AF, device=dev)
        mean = torch.empty(B, AF, device=dev) if want_mean else None
        wsb = lib.dppo_gaussian_workspace_bytes(C.byref(d), None, self.prec, B)
        ws = self._ws_g.get(wsb, dev)
        hip.check(lib.dppo_gaussian_sample(
            C.byref(d), self.prec, net.flat_params().data_ptr(), net.packed(self.prec, 0).data_ptr(), C.byref(cfg),
            net.logvar_ptr(), obs.data_ptr(), hip.ptr(noise), B, actions.data_ptr(), hip.ptr(mean), ws.data_ptr(), ws.numel(),
            hip.stream()), "dppo_gaussian_sample")
        return actions.view(B, self.horizon_steps, -1), mean

    def forward(self, cond, deterministic=False, network_override=None, reparameterize=False, get_logprob=False, noise=None):
        """cond {"state": (B,To,Do)} -> sampled action chunk (B,Ta,Da) (reference :94-121)."""
        if get_logprob or reparameterize:
            raise NotImplementedError("dppo_amd: get_logprob / reparameterize (SAC-style use) are out of scope")
        net = network_override if network_override is not None else self.network
        return self._sample(net, cond, deterministic, noise)[0]
