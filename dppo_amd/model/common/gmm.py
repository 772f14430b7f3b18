"""Mixture-of-Gaussians policy parameterisation.  Mirrors ``dppo/model/common/gmm.py:14-97`` (reference ``GMMModel``): constructor
surface, checkpoint loading, ``forward`` (draw a component from Categorical(logits), then the action from that component).  One
library call per sampling step (``dppo_gmm_sample``: both trunks + the epilogue); the draws are made in the kernel (Philox keyed
from torch's CPU generator) unless recorded ones are passed (parity tests)."""
from __future__ import annotations

import ctypes as C
import logging

import torch

from dppo_amd import hip

log = logging.getLogger(__name__)


class GMMModel(torch.nn.Module):
    def __init__(self, network, horizon_steps, network_path=None, device="cuda:0", precision=None, **kwargs):
        super().__init__()
        self.device = device
        self.network = network.to(device)
        if network_path is not None:
            checkpoint = torch.load(network_path, map_location=self.device, weights_only=True)  # safe loader only
            self.load_state_dict(checkpoint["model"], strict=False)
            self.network.mark_updated()
            log.info("Loaded actor from %s", network_path)
        self.horizon_steps = horizon_steps
        self.prec = hip.PREC_BY_NAME[precision] if precision is not None else network.prec
        object.__setattr__(self, "_ws_g", hip.Workspace())

    def loss(self, true_action, cond, **kwargs):
        raise NotImplementedError("dppo_amd: supervised GMM pre-training (-log p) is out of scope")

    def _net_args(self, net):
        m, w = net.mean_net, net.weights_net
        return (C.byref(m.net_desc()), C.byref(w.net_desc()), self.prec, m.flat_params().data_ptr(), m.packed(self.prec, 0).data_ptr(),
                w.flat_params().data_ptr(), w.packed(self.prec, 0).data_ptr())

    def _workspace(self, net, critic, N, dev):
        lib = hip.load()
        wsb = lib.dppo_gmm_workspace_bytes(C.byref(net.mean_net.net_desc()), C.byref(net.weights_net.net_desc()),
                                           C.byref(critic.net_desc()) if critic is not None else None, self.prec, N)
        if wsb < 0:
            hip.check(int(wsb), "dppo_gmm_workspace_bytes")
        return self._ws_g.get(wsb, dev)

    @torch.no_grad()
    def forward(self, cond, deterministic=False, modes=None, noise=None):
        """cond {"state": (B,To,Do)} -> sampled action chunk (B,Ta,Da) (reference :88-97; ``deterministic`` shrinks every
        component's std to 1e-4, the component is still drawn).  ``modes`` (B,) int64 / ``noise`` (B,Ta*Da): recorded draws."""
        state = cond["state"]
        hip.require_gpu(state, type(self).__name__ + ".forward")
        net = self.network if not hasattr(self, "actor_ft") else self.actor_ft
        B, dev = state.shape[0], state.device
        AF = net.action_dim * net.horizon_steps
        obs = state.reshape(B, -1).contiguous().float()
        cfg = net.gmm_cfg(deterministic=deterministic)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        cfg.seed_lo, cfg.seed_hi = seed & 0xFFFFFFFF, seed >> 32
        if modes is not None:
            modes = modes.reshape(B).to(torch.int64).contiguous()
        if noise is not None:
            noise = noise.reshape(B, AF).contiguous().float()
        actions = torch.empty(B, AF, device=dev)
        ws = self._workspace(net, None, B, dev)
        hip.check(hip.load().dppo_gmm_sample(*self._net_args(net), C.byref(cfg), net.logvar_ptr(), obs.data_ptr(), hip.ptr(modes),
                                             hip.ptr(noise), B, actions.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream()),
                  "dppo_gmm_sample")
        return actions.view(B, self.horizon_steps, -1)
