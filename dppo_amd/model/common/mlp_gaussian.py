"""Gaussian policy head on the observation trunk.  Mirrors ``dppo/model/common/mlp_gaussian.py:283-362`` (reference
``Gaussian_MLP``): mean = [tanh](ResidualMLP(state)); sigma fixed, or learned per action dimension (``logvar``, clamped
to [log std_min^2, log std_max^2]).  Parameter names match the reference state dict (``mlp_mean.layers.*``, ``logvar``,
``logvar_min``, ``logvar_max``).  The arithmetic is the HIP library's (``dppo_gaussian_*``, csrc/gaussian.hip around the
fused trunk kernels); this class owns parameters only.

Built: ``residual_style=True`` with ``fixed_std`` given -- the d3il / furniture fine-tuning cfgs (7 of the 10 shipped
PPO_Gaussian cfgs).  Not built (raises): the state-dependent ``mlp_logvar`` head on a plain ``MLP`` base (``fixed_std=None``,
the three gym ``scratch`` cfgs) and plain (non-residual) trunks.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from dppo_amd import hip
from dppo_amd.model.common.mlp import HipNet, ResidualMLP
from dppo_amd.model.common.vit import VisionMixin


class Gaussian_MLP(HipNet):
    def __init__(self, action_dim, horizon_steps, cond_dim, mlp_dims=[256, 256, 256], activation_type="Mish",
                 tanh_output=True, residual_style=False, use_layernorm=False, dropout=0.0, fixed_std=None,
                 learn_fixed_std=False, std_min=0.01, std_max=1, precision="bf16"):
        super().__init__()
        if fixed_std is None:
            raise NotImplementedError("dppo_amd: Gaussian_MLP with a state-dependent logvar head (fixed_std=None) is not built")
        if not residual_style:
            raise NotImplementedError("dppo_amd: Gaussian_MLP needs residual_style=True (plain MLP trunk not built)")
        if dropout:
            raise NotImplementedError("Dropout not implemented for residual MLP!")
        self.action_dim, self.horizon_steps, self.cond_dim = action_dim, horizon_steps, cond_dim
        out_dim = action_dim * horizon_steps
        self.mlp_mean = ResidualMLP([cond_dim] + list(mlp_dims) + [out_dim], activation_type=activation_type,
                                    out_activation_type="Identity", use_layernorm=use_layernorm)
        if learn_fixed_std:  # initialised to fixed_std (reference :331-336)
            self.logvar = nn.Parameter(torch.log(torch.tensor([fixed_std ** 2 for _ in range(action_dim)])),
                                       requires_grad=True)
        self.logvar_min = nn.Parameter(torch.log(torch.tensor(std_min ** 2)), requires_grad=False)
        self.logvar_max = nn.Parameter(torch.log(torch.tensor(std_max ** 2)), requires_grad=False)
        self.use_fixed_std, self.fixed_std, self.learn_fixed_std = True, fixed_std, learn_fixed_std
        self.tanh_output = tanh_output
        self.prec = hip.PREC_BY_NAME[precision]

    def trunk_parameters(self):
        return list(self.mlp_mean.parameters())

    def net_desc(self) -> hip.NetDesc:
        d = self.__dict__.get("_desc_cache")
        if d is None:
            m = self.mlp_mean
            d = hip.NetDesc(kind=1, in_dim=self.cond_dim, hidden=m.hidden, n_blocks=m.n_blocks, out_dim=m.out_dim, act=m.act,
                            time_dim=0, act_flat=0, cond_dim=self.cond_dim, cond_hidden=0, cond_out=0,
                            use_layernorm=m.use_layernorm, plain=m.plain)
            object.__setattr__(self, "_desc_cache", d)
        return d

    def gaussian_cfg(self, deterministic=False, randn_clip=10.0) -> hip.GaussianCfg:
        return hip.GaussianCfg(
            horizon_steps=self.horizon_steps, action_dim=self.action_dim, tanh_mean=int(bool(self.tanh_output)),
            std_mode=1 if self.learn_fixed_std else 0, norm_adv=1, has_vclip=0, deterministic=int(bool(deterministic)), pad=0,
            fixed_std=float(self.fixed_std), logvar_min=float(self.logvar_min), logvar_max=float(self.logvar_max),
            randn_clip=float(randn_clip), clip_ploss_coef=0.0, clip_vloss_coef=0.0, seed_lo=0, seed_hi=0)

    def logvar_ptr(self):
        return self.logvar.data_ptr() if self.learn_fixed_std else None

    @torch.no_grad()
    def forward(self, cond):
        """cond {"state": (B,To,Do)} -> (mean (B,Ta*Da), scale (B,Ta*Da)), like the reference's forward (inference only:
        the differentiable evaluation is fused into ``PPO_Gaussian.loss``)."""
        import ctypes as C
        state = cond["state"]
        hip.require_gpu(state, "Gaussian_MLP.forward")
        B = state.shape[0]
        obs = state.reshape(B, -1).contiguous().float()
        AF = self.action_dim * self.horizon_steps
        lib, d = hip.load(), self.net_desc()
        cfg = self.gaussian_cfg()
        zeros = torch.zeros(B, AF, device=state.device)
        actions = torch.empty(B, AF, device=state.device)
        mean = torch.empty(B, AF, device=state.device)
        wsb = lib.dppo_gaussian_workspace_bytes(C.byref(d), None, self.prec, B)
        ws = self.__dict__.setdefault("_ws", hip.Workspace()).get(wsb, state.device)
        hip.check(lib.dppo_gaussian_sample(C.byref(d), self.prec, self.flat_params().data_ptr(),
                                           self.packed(self.prec, 0).data_ptr(), C.byref(cfg), self.logvar_ptr(),
                                           obs.data_ptr(), zeros.data_ptr(), B, actions.data_ptr(), mean.data_ptr(),
                                           ws.data_ptr(), ws.numel(), hip.stream()), "dppo_gaussian_sample")
        if self.learn_fixed_std:
            lv = torch.clamp(self.logvar.detach(), self.logvar_min, self.logvar_max)
            scale = torch.exp(0.5 * lv).view(1, self.action_dim).repeat(B, self.horizon_steps)
        else:
            scale = torch.full_like(mean, self.fixed_std)
        return mean, scale


class Gaussian_VisionMLP(VisionMixin, Gaussian_MLP):
    """ViT backbone + SpatialEmb, then the Gaussian head's trunk on cat[feat, state]; the mean is always tanh-squashed.
    Mirrors ``dppo/model/common/mlp_gaussian.py:112-281``."""

    def __init__(self, backbone, action_dim, horizon_steps, cond_dim, img_cond_steps=1, mlp_dims=[256, 256, 256],
                 activation_type="Mish", residual_style=False, use_layernorm=False, fixed_std=None, learn_fixed_std=False,
                 std_min=0.01, std_max=1, spatial_emb=0, visual_feature_dim=128, dropout=0, num_img=1, augment=False,
                 precision="bf16"):
        Gaussian_MLP.__init__(self, action_dim, horizon_steps, cond_dim + spatial_emb * num_img, mlp_dims=mlp_dims,
                              activation_type=activation_type, tanh_output=True, residual_style=residual_style,
                              use_layernorm=use_layernorm, dropout=0.0, fixed_std=fixed_std, learn_fixed_std=learn_fixed_std,
                              std_min=std_min, std_max=std_max, precision=precision)
        self._init_vision(backbone, cond_dim, img_cond_steps, spatial_emb, num_img, augment, dropout, precision)
        self._vision_modules_first("backbone", "compress", "compress1", "compress2")

    @torch.no_grad()
    def forward(self, cond):
        return Gaussian_MLP.forward(self, {"state": self.encode_obs(cond)})
