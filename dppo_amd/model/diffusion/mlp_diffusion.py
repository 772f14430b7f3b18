"""Denoiser backbone.  Mirrors ``dppo/model/diffusion/mlp_diffusion.py:174-250`` (reference ``DiffusionMLP``)."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from dppo_amd import hip
from dppo_amd.model.common.mlp import MLP, HipNet, ResidualMLP
from dppo_amd.model.common.vit import VisionMixin


class _NoParams(nn.Module):
    """Placeholder for the parameter-free stages of the reference's nn.Sequential (keeps indices 1 and 3)."""


class _PlainMLP(nn.Module):
    """Parameter container with the reference MLP's names: ``moduleList.{i}.linear_1.{weight,bias}`` (mlp.py:46-74)."""

    def __init__(self, dims):
        super().__init__()
        self.moduleList = nn.ModuleList()
        for i in range(len(dims) - 1):
            layer = nn.Module()
            layer.linear_1 = nn.Linear(dims[i], dims[i + 1])
            self.moduleList.append(layer)


class DiffusionMLP(HipNet):
    """eps(x, t, state) = ResidualMLP(cat[x, time_mlp(sinusoid(t)), state]).

    Parameter names match the reference state dict: ``time_embedding.{1,3}.*``, ``mlp_mean.layers.*``.
    """

    def __init__(self, action_dim, horizon_steps, cond_dim, time_dim=16, mlp_dims=[256, 256], cond_mlp_dims=None,
                 activation_type="Mish", out_activation_type="Identity", use_layernorm=False, residual_style=False,
                 precision="bf16"):
        super().__init__()
        if cond_mlp_dims is not None and len(cond_mlp_dims) != 2:
            raise NotImplementedError("dppo_amd: cond_mlp is built for two layers (every shipped cfg: [hidden, out])")
        if not residual_style and (cond_mlp_dims is not None or use_layernorm):
            raise NotImplementedError("dppo_amd: a plain (residual_style=False) DiffusionMLP is built without cond_mlp / LayerNorm")
        self.time_embedding = nn.Sequential(_NoParams(), nn.Linear(time_dim, time_dim * 2), _NoParams(),
                                            nn.Linear(time_dim * 2, time_dim))
        out_dim = action_dim * horizon_steps
        self.cond_mlp_dims = list(cond_mlp_dims) if cond_mlp_dims is not None else None
        if cond_mlp_dims is not None:  # reference mlp_diffusion.py:201-207: MLP([cond_dim] + dims), same activation, Identity out
            if activation_type not in ("ReLU", "Mish"):
                raise NotImplementedError(f"dppo_amd: activation {activation_type!r} not built (ReLU, Mish are)")
            self.cond_mlp = _PlainMLP([cond_dim] + list(cond_mlp_dims))
        in_dim = time_dim + out_dim + (cond_mlp_dims[-1] if cond_mlp_dims is not None else cond_dim)
        self.mlp_mean = (ResidualMLP if residual_style else MLP)(
            [in_dim] + list(mlp_dims) + [out_dim], activation_type=activation_type, out_activation_type=out_activation_type,
            use_layernorm=use_layernorm)
        self.is_plain = not residual_style
        self.time_dim, self.action_dim, self.horizon_steps, self.cond_dim = time_dim, action_dim, horizon_steps, cond_dim
        self.prec = hip.PREC_BY_NAME[precision]
        self.n_time = 1000  # rows of the time-embedding table built for stand-alone forward() calls
        object.__setattr__(self, "_ws", hip.Workspace())

    def net_desc(self) -> hip.NetDesc:
        d = self.__dict__.get("_desc_cache")  # architecture is fixed after construction
        if d is not None:
            return d
        m = self.mlp_mean
        ch, co = self.cond_mlp_dims if self.cond_mlp_dims is not None else (0, 0)
        d = hip.NetDesc(kind=0, in_dim=m.in_dim, hidden=m.hidden, n_blocks=m.n_blocks, out_dim=m.out_dim, act=m.act,
                        time_dim=self.time_dim, act_flat=m.out_dim, cond_dim=self.cond_dim, cond_hidden=ch,
                        cond_out=co, use_layernorm=m.use_layernorm, plain=m.plain)
        object.__setattr__(self, "_desc_cache", d)
        return d

    @torch.no_grad()
    def forward(self, x, time, cond, **kwargs):
        """x (B,Ta,Da), time (B,) or int, cond {"state": (B,To,Do)} -> (B,Ta,Da).  Inference only."""
        hip.require_gpu(x, "DiffusionMLP.forward")
        B, Ta, Da = x.shape
        if not torch.is_tensor(time):
            time = torch.full((B,), int(time), device=x.device, dtype=torch.long)
        t = time.reshape(B).to(torch.long).contiguous()
        state = cond["state"].reshape(B, -1).contiguous().float()
        xf = x.reshape(B, -1).contiguous().float()
        lib, d = hip.load(), self.net_desc()
        flat, pk = self.flat_params(), self.packed(self.prec, self.n_time)
        out = torch.empty(B, Ta * Da, dtype=torch.float32, device=x.device)
        wsb = lib.dppo_mlp_forward_workspace_bytes(C.byref(d), self.prec, B)
        ws = self._ws.get(wsb, x.device)
        hip.check(lib.dppo_actor_forward(C.byref(d), self.prec, flat.data_ptr(), pk.data_ptr(), xf.data_ptr(),
                                         t.data_ptr(), state.data_ptr(), B, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                         hip.stream()), "dppo_actor_forward")
        return out.view(B, Ta, Da)


class VisionDiffusionMLP(VisionMixin, DiffusionMLP):
    """ViT backbone + SpatialEmb, then DiffusionMLP on cat[x, time_emb, feat, state].  Mirrors
    ``dppo/model/diffusion/mlp_diffusion.py:19-171``: the trunk is a DiffusionMLP whose observation vector is
    cat[feat, state] (same column order: :162-167), so every kernel of the state-observation path serves it."""

    def __init__(self, backbone, action_dim, horizon_steps, cond_dim, img_cond_steps=1, time_dim=16, mlp_dims=[256, 256],
                 activation_type="Mish", out_activation_type="Identity", use_layernorm=False, residual_style=False,
                 spatial_emb=0, visual_feature_dim=128, dropout=0, num_img=1, augment=False, precision="bf16"):
        vis_dim = spatial_emb * num_img
        DiffusionMLP.__init__(self, action_dim, horizon_steps, cond_dim + vis_dim, time_dim=time_dim, mlp_dims=mlp_dims,
                              activation_type=activation_type, out_activation_type=out_activation_type,
                              use_layernorm=use_layernorm, residual_style=residual_style, precision=precision)
        self._init_vision(backbone, cond_dim, img_cond_steps, spatial_emb, num_img, augment, dropout, precision)
        self._vision_modules_first("backbone", "compress", "compress1", "compress2")

    def trunk_parameters(self):
        skip = self._vision_parameter_ids()
        return [p for p in self.parameters() if id(p) not in skip]

    @torch.no_grad()
    def forward(self, x, time, cond, **kwargs):
        return DiffusionMLP.forward(self, x, time, {"state": self.encode_obs(cond)})
