"""State-value critic.  Mirrors ``dppo/model/common/critic.py:15-54`` (reference ``CriticObs``)."""
from __future__ import annotations

import ctypes as C
from typing import Union

import torch

from dppo_amd import hip
from dppo_amd.model.common.mlp import MLP, HipNet, ResidualMLP
from dppo_amd.model.common.vit import VisionMixin


class CriticObs(HipNet):
    """V(s) = ResidualMLP([To*Do] + mlp_dims + [1]) on the flattened observation history."""

    def __init__(self, cond_dim, mlp_dims, activation_type="Mish", use_layernorm=False, residual_style=False,
                 precision="bf16", **kwargs):
        super().__init__()
        self.Q1 = (ResidualMLP if residual_style else MLP)([cond_dim] + list(mlp_dims) + [1], activation_type=activation_type,
                                                           out_activation_type="Identity", use_layernorm=use_layernorm)
        self.cond_dim = cond_dim
        self.prec = hip.PREC_BY_NAME[precision]
        object.__setattr__(self, "_ws", hip.Workspace())

    def net_desc(self) -> hip.NetDesc:
        d = self.__dict__.get("_desc_cache")  # architecture is fixed after construction
        if d is None:
            q = self.Q1
            d = hip.NetDesc(kind=1, in_dim=self.cond_dim, hidden=q.hidden, n_blocks=q.n_blocks, out_dim=1, act=q.act,
                            time_dim=0, act_flat=0, cond_dim=self.cond_dim, cond_hidden=0, cond_out=0,
                            use_layernorm=q.use_layernorm, plain=q.plain)
            object.__setattr__(self, "_desc_cache", d)
        return d

    @torch.no_grad()
    def forward(self, cond: Union[dict, torch.Tensor]) -> torch.Tensor:
        """cond: {"state": (B,To,Do)} or (B, To*Do) -> (B,1).  Inference only; the training forward/backward of
        the critic lives inside PPODiffusion.loss (fused)."""
        state = cond["state"] if isinstance(cond, dict) else cond
        hip.require_gpu(state, "CriticObs.forward")
        B = state.shape[0]
        state = state.reshape(B, -1).contiguous().float()
        lib, d = hip.load(), self.net_desc()
        flat, pk = self.flat_params(), self.packed(self.prec, 0)
        out = torch.empty(B, dtype=torch.float32, device=state.device)
        wsb = lib.dppo_mlp_forward_workspace_bytes(C.byref(d), self.prec, B)
        ws = self._ws.get(wsb, state.device)
        hip.check(lib.dppo_critic_forward(C.byref(d), self.prec, flat.data_ptr(), pk.data_ptr(), state.data_ptr(), B,
                                          out.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream()),
                  "dppo_critic_forward")
        return out.view(B, 1)


class ViTCritic(VisionMixin, CriticObs):
    """ViT backbone + SpatialEmb, then the state-value MLP on cat[feat, state].  Mirrors
    ``dppo/model/common/critic.py:116-206`` (Q1 is registered before the backbone there too)."""

    def __init__(self, backbone, cond_dim, img_cond_steps=1, spatial_emb=128, dropout=0, augment=False, num_img=1,
                 precision="bf16", **kwargs):
        CriticObs.__init__(self, cond_dim=spatial_emb * num_img + cond_dim, precision=precision, **kwargs)
        self._init_vision(backbone, cond_dim, img_cond_steps, spatial_emb, num_img, augment, dropout, precision)

    def trunk_parameters(self):
        return list(self.Q1.parameters())

    @torch.no_grad()
    def forward(self, cond, no_augment=False):
        return CriticObs.forward(self, self.encode_obs(cond, augment=self.augment and not no_augment))
