"""Conv denoiser.  Mirrors ``dppo/model/diffusion/unet.py:27-327`` (reference ``ResidualBlock1D``, ``Unet1D``) and
``dppo/model/diffusion/modules.py:28-95`` (``Downsample1d``, ``Upsample1d``, ``Conv1dBlock``): same constructor, same
state-dict names and order (``time_mlp.{1,3}``, ``mid_modules.i.blocks.j.block.{0,2}``, ``...cond_encoder.{0,2,4}``,
``...residual_conv``, ``down_modules.i.{0,1,2}``, ``up_modules.i.{0,1,2}.conv``, ``final_conv.{0.block.{0,2},1}``).

The classes own parameters only (views of one flat fp32 buffer); the arithmetic is the HIP library's ``dppo_unet_*``
(csrc/unet.hip): every convolution an MFMA GEMM over a channel-last, time-padded activation image, GroupNorm + activation +
FiLM / residual in one epilogue kernel per block half.  ``forward``, the K-step sampler and the log-prob evaluation of
``VPGDiffusion``, ``PPODiffusion.loss`` / ``ppo_update`` and ``DiffusionModel.p_losses`` (forward with a tape + backward to every
parameter gradient) run on it.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from dppo_amd import hip
from dppo_amd.model.common.vit import VisionMixin
from dppo_amd.model.common.mlp import SUPPORTED_ACT, HipNet


class _Slot(nn.Module):
    """Parameter-free stage of one of the reference's nn.Sequential containers (keeps the state-dict indices)."""


class Conv1dBlock(nn.Module):
    """Conv1d -> GroupNorm -> act; ``block.0`` is the conv and ``block.2`` the norm (modules.py:73-92)."""

    def __init__(self, inp_channels, out_channels, kernel_size, n_groups=None, activation_type="Mish", eps=1e-5):
        super().__init__()
        if n_groups is None:
            raise NotImplementedError("dppo_amd: Conv1dBlock without GroupNorm (n_groups=None) is not built")
        self.block = nn.Sequential(nn.Conv1d(inp_channels, out_channels, kernel_size, padding=kernel_size // 2), _Slot(),
                                   nn.GroupNorm(n_groups, out_channels, eps=eps), _Slot(), _Slot())


class Downsample1d(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, 3, 2, 1)


class Upsample1d(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1)


class ResidualBlock1D(nn.Module):
    def __init__(self, in_channels, out_channels, cond_dim, kernel_size=5, n_groups=None, cond_predict_scale=False,
                 larger_encoder=False, activation_type="Mish", groupnorm_eps=1e-5):
        super().__init__()
        self.blocks = nn.ModuleList([
            Conv1dBlock(in_channels, out_channels, kernel_size, n_groups, activation_type, groupnorm_eps),
            Conv1dBlock(out_channels, out_channels, kernel_size, n_groups, activation_type, groupnorm_eps)])
        cc = out_channels * 2 if cond_predict_scale else out_channels
        if larger_encoder:
            self.cond_encoder = nn.Sequential(nn.Linear(cond_dim, cc), _Slot(), nn.Linear(cc, cc), _Slot(), nn.Linear(cc, cc),
                                              _Slot())
        else:
            self.cond_encoder = nn.Sequential(_Slot(), nn.Linear(cond_dim, cc), _Slot())
        self.residual_conv = nn.Conv1d(in_channels, out_channels, 1) if in_channels != out_channels else _Slot()


class Unet1D(HipNet):
    is_unet = True

    def __init__(self, action_dim, cond_dim=None, diffusion_step_embed_dim=32, dim=32, dim_mults=(1, 2, 4, 8),
                 smaller_encoder=False, cond_mlp_dims=None, kernel_size=5, n_groups=None, activation_type="Mish",
                 cond_predict_scale=False, groupnorm_eps=1e-5, horizon_steps=None, precision="bf16"):
        super().__init__()
        if cond_mlp_dims is not None:
            raise NotImplementedError("dppo_amd: Unet1D with an observation encoder (cond_mlp_dims) is not built (no shipped cfg)")
        if activation_type not in SUPPORTED_ACT:
            raise NotImplementedError(f"dppo_amd: activation {activation_type!r} not built (ReLU, Mish are)")
        dims = [action_dim] + [dim * m for m in dim_mults]
        in_out = list(zip(dims[:-1], dims[1:]))
        dsed = diffusion_step_embed_dim
        self.time_mlp = nn.Sequential(_Slot(), nn.Linear(dsed, dsed * 4), _Slot(), nn.Linear(dsed * 4, dsed))
        cbd = dsed + cond_dim
        larger = cond_mlp_dims is None and not smaller_encoder
        rb = lambda ci, co: ResidualBlock1D(ci, co, cond_dim=cbd, kernel_size=kernel_size, n_groups=n_groups,
                                            cond_predict_scale=cond_predict_scale, larger_encoder=larger,
                                            activation_type=activation_type, groupnorm_eps=groupnorm_eps)
        mid = dims[-1]
        self.mid_modules = nn.ModuleList([rb(mid, mid), rb(mid, mid)])
        self.down_modules = nn.ModuleList([])
        for ind, (di, do) in enumerate(in_out):
            last = ind >= len(in_out) - 1
            self.down_modules.append(nn.ModuleList([rb(di, do), rb(do, do), Downsample1d(do) if not last else _Slot()]))
        self.up_modules = nn.ModuleList([])
        for ind, (di, do) in enumerate(reversed(in_out[1:])):  # `is_last` is never true inside this loop (:219-222)
            self.up_modules.append(nn.ModuleList([rb(do * 2, di), rb(di, di), Upsample1d(di)]))
        self.final_conv = nn.Sequential(Conv1dBlock(dim, dim, kernel_size, n_groups, activation_type, groupnorm_eps),
                                        nn.Conv1d(dim, action_dim, 1))
        self.action_dim, self.cond_dim, self.time_dim, self.dim = action_dim, cond_dim, dsed, dim
        self.dim_mults, self.kernel_size, self.n_groups = list(dim_mults), kernel_size, n_groups
        self.larger_encoder, self.cond_predict_scale = larger, bool(cond_predict_scale)
        self.act, self.groupnorm_eps = SUPPORTED_ACT[activation_type], groupnorm_eps
        # the reference infers the chunk length from x at call time; the kernel workspace / descriptor want it up front --
        # DiffusionModel sets it from its own horizon_steps when it adopts the network, or pass it here
        self.horizon_steps = horizon_steps
        self.prec = hip.PREC_BY_NAME[precision]
        self.n_time = 1000

    def net_desc(self) -> hip.UnetDesc:
        assert self.horizon_steps is not None, "Unet1D.horizon_steps is not set (DiffusionModel sets it; or pass horizon_steps=)"
        key = ("_desc_cache", self.horizon_steps)
        d = self.__dict__.get("_desc_cache")
        if d is None or d[0] != key:
            m = (C.c_int32 * 4)(*(self.dim_mults + [0] * (4 - len(self.dim_mults))))
            desc = hip.UnetDesc(action_dim=self.action_dim, cond_dim=self.cond_dim, horizon_steps=self.horizon_steps,
                                time_dim=self.time_dim, dim=self.dim, n_levels=len(self.dim_mults), mults=m,
                                kernel_size=self.kernel_size, n_groups=self.n_groups, larger_encoder=int(self.larger_encoder),
                                cond_predict_scale=int(self.cond_predict_scale), act=self.act,
                                groupnorm_eps=float(self.groupnorm_eps))
            d = (key, desc)
            object.__setattr__(self, "_desc_cache", d)
        return d[1]

    def _abi_param_count(self) -> int:
        d = self.net_desc()
        return hip.load().dppo_unet_param_count(C.byref(d))

    def _abi_packed_bytes(self, prec: int, n_time: int) -> int:
        d = self.net_desc()
        return hip.load().dppo_unet_packed_bytes(C.byref(d), prec, n_time)

    def _abi_pack(self, prec: int, n_time: int, buf: torch.Tensor) -> None:
        d = self.net_desc()
        hip.check(hip.load().dppo_unet_pack(C.byref(d), prec, n_time, self.flat_params().data_ptr(), buf.data_ptr(),
                                            hip.stream()), "dppo_unet_pack")

    def workspace(self, rows: int, device, n_steps: int = 0) -> torch.Tensor:
        """n_steps > 0: sized for a sampling call of that many steps with its FiLM tables precomputed."""
        d = self.net_desc()
        lib = hip.load()
        wsb = (lib.dppo_unet_sample_workspace_bytes(C.byref(d), self.prec, rows, n_steps) if n_steps > 0 else
               lib.dppo_unet_workspace_bytes(C.byref(d), self.prec, rows))
        if wsb < 0:
            hip.check(int(wsb), "dppo_unet_workspace_bytes")
        return self.__dict__.setdefault("_ws", hip.Workspace()).get(wsb, device)

    @torch.no_grad()
    def forward(self, x, time, cond, **kwargs):
        """x (B,Ta,Da), time (B,) or int, cond {"state": (B,To,Do)} -> (B,Ta,Da).  Inference only."""
        hip.require_gpu(x, "Unet1D.forward")
        B, Ta, Da = x.shape
        if self.horizon_steps is None:
            self.horizon_steps = Ta
        assert Ta == self.horizon_steps and Da == self.action_dim
        if not torch.is_tensor(time):
            time = torch.full((B,), int(time), device=x.device, dtype=torch.long)
        t = time.reshape(-1).expand(B).to(torch.long).contiguous()
        state = cond["state"].reshape(B, -1).contiguous().float()
        xf = x.contiguous().float()
        out = torch.empty(B, Ta, Da, dtype=torch.float32, device=x.device)
        d = self.net_desc()
        ws = self.workspace(B, x.device)
        hip.check(hip.load().dppo_unet_forward(
            C.byref(d), self.prec, self.flat_params().data_ptr(), self.packed(self.prec, self.n_time).data_ptr(),
            xf.data_ptr(), t.data_ptr(), state.data_ptr(), B, out.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream()),
            "dppo_unet_forward")
        return out


class VisionUnet1D(VisionMixin, Unet1D):
    """ViT backbone + SpatialEmb, then Unet1D conditioned on cat[time_emb, feat, state].  Mirrors
    ``dppo/model/diffusion/unet.py:330-620`` (same column order: :581-593)."""

    def __init__(self, backbone, action_dim, img_cond_steps=1, cond_dim=None, diffusion_step_embed_dim=32, dim=32,
                 dim_mults=(1, 2, 4, 8), smaller_encoder=False, cond_mlp_dims=None, kernel_size=5, n_groups=None,
                 activation_type="Mish", cond_predict_scale=False, groupnorm_eps=1e-5, spatial_emb=0, dropout=0, num_img=1,
                 augment=False, horizon_steps=None, precision="bf16"):
        Unet1D.__init__(self, action_dim, cond_dim=cond_dim + spatial_emb * num_img,
                        diffusion_step_embed_dim=diffusion_step_embed_dim, dim=dim, dim_mults=dim_mults,
                        smaller_encoder=smaller_encoder, cond_mlp_dims=cond_mlp_dims, kernel_size=kernel_size,
                        n_groups=n_groups, activation_type=activation_type, cond_predict_scale=cond_predict_scale,
                        groupnorm_eps=groupnorm_eps, horizon_steps=horizon_steps, precision=precision)
        self._init_vision(backbone, cond_dim, img_cond_steps, spatial_emb, num_img, augment, dropout, precision)
        self._vision_modules_first("backbone", "compress", "compress1", "compress2")

    def trunk_parameters(self):
        skip = self._vision_parameter_ids()
        return [p for p in self.parameters() if id(p) not in skip]

    @torch.no_grad()
    def forward(self, x, time, cond, **kwargs):
        return Unet1D.forward(self, x, time, {"state": self.encode_obs(cond)})
