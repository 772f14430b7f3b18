"""ViT patch encoder: parameter container + the HIP visual encoder that runs it.

Mirrors ``dppo/model/common/vit.py`` (reference): ``VitEncoderConfig`` (:16-25), ``VitEncoder`` (:28-62), ``PatchEmbed2``
(:80-102), ``MultiHeadAttention`` (:105-127), ``TransformerLayer`` (:130-151), ``MinVit`` (:154-195); the parameter names
are the reference's (``vit.pos_embed``, ``vit.patch_embed.embed.{0,3}``, ``vit.net.{l}.{layer_norm1, mha.qkv_proj,
mha.out_proj, layer_norm2, linear1, linear2}``, ``vit.norm``).  These classes own parameters only; the arithmetic is
``dppo_vis_encode`` / ``dppo_vis_backward`` (csrc/vision.hip), reached through :class:`VisualEncoder`, which pairs a
backbone with the SpatialEmb head(s) of the network that owns both.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import List

import torch
from torch import nn

from dppo_amd import hip
from dppo_amd.model.common.mlp import HipNet


@dataclass
class VitEncoderConfig:
    patch_size: int = 8
    depth: int = 1
    embed_dim: int = 128
    num_heads: int = 4
    stride: int = -1
    embed_style: str = "embed2"
    embed_norm: int = 0


class _Slot(nn.Module):
    """Parameter-free stage of a reference nn.Sequential (keeps the indices of the stages around it)."""


class PatchEmbed2(nn.Module):
    def __init__(self, embed_dim, use_norm, num_channel=3, img_h=96, img_w=96):
        super().__init__()
        if use_norm:
            raise NotImplementedError("dppo_amd: PatchEmbed2 with embed_norm is not built (no shipped cfg sets it)")
        self.embed = nn.Sequential(nn.Conv2d(num_channel, embed_dim, kernel_size=8, stride=4), _Slot(), _Slot(),
                                   nn.Conv2d(embed_dim, embed_dim, kernel_size=3, stride=2))
        H1, W1 = math.ceil((img_h - 8) / 4) + 1, math.ceil((img_w - 8) / 4) + 1
        H2, W2 = math.ceil((H1 - 3) / 2) + 1, math.ceil((W1 - 3) / 2) + 1
        self.num_patch, self.patch_dim = H2 * W2, embed_dim


class MultiHeadAttention(nn.Module):
    def __init__(self, embed_dim, num_head):
        super().__init__()
        assert embed_dim % num_head == 0
        self.num_head = num_head
        self.qkv_proj = nn.Linear(embed_dim, 3 * embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)


class TransformerLayer(nn.Module):
    def __init__(self, embed_dim, num_head, dropout):
        super().__init__()
        if dropout:
            raise NotImplementedError("dppo_amd: dropout in the ViT is not built (the reference constructs it with 0)")
        self.layer_norm1 = nn.LayerNorm(embed_dim)
        self.mha = MultiHeadAttention(embed_dim, num_head)
        self.layer_norm2 = nn.LayerNorm(embed_dim)
        self.linear1 = nn.Linear(embed_dim, 4 * embed_dim)
        self.linear2 = nn.Linear(4 * embed_dim, embed_dim)


class MinVit(nn.Module):
    def __init__(self, embed_style, embed_dim, embed_norm, num_head, depth, num_channel=3, img_h=96, img_w=96):
        super().__init__()
        if embed_style != "embed2":
            raise NotImplementedError("dppo_amd: only embed_style='embed2' is built (every shipped cfg)")
        self.patch_embed = PatchEmbed2(embed_dim, use_norm=embed_norm, num_channel=num_channel, img_h=img_h, img_w=img_w)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patch, embed_dim))
        self.net = nn.Sequential(*[TransformerLayer(embed_dim, num_head, dropout=0) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.num_patches = self.patch_embed.num_patch
        nn.init.trunc_normal_(self.pos_embed, std=0.02)  # reference :187-189 (timm ViT initialisation)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)


class VitEncoder(nn.Module):
    def __init__(self, obs_shape: List[int], cfg: VitEncoderConfig, num_channel=3, img_h=96, img_w=96):
        super().__init__()
        if not isinstance(cfg, VitEncoderConfig):  # Hydra hands a DictConfig over
            cfg = VitEncoderConfig(**{k: cfg[k] for k in cfg})
        self.obs_shape, self.cfg = obs_shape, cfg
        self.vit = MinVit(embed_style=cfg.embed_style, embed_dim=cfg.embed_dim, embed_norm=cfg.embed_norm,
                          num_head=cfg.num_heads, depth=cfg.depth, num_channel=num_channel, img_h=img_h, img_w=img_w)
        self.img_h, self.img_w, self.num_channel = img_h, img_w, num_channel
        self.num_patch = self.vit.num_patches
        self.patch_repr_dim = cfg.embed_dim
        self.repr_dim = cfg.embed_dim * self.vit.num_patches

    def forward(self, obs, flatten=False):
        raise NotImplementedError("dppo_amd: the ViT runs inside dppo_vis_encode (VisualEncoder), together with SpatialEmb")


class VisualEncoder(HipNet):
    """backbone + SpatialEmb head(s) -> the observation vector cat[feat, state] (reference: the first half of
    VisionDiffusionMLP / VisionUnet1D / ViTCritic.forward).  Owns one flat fp32 buffer over exactly those parameters (in
    state-dict order) and the training tape's workspace.  Not a registered child of the network that builds it."""

    MAX_IMAGES = 1024  # per dppo_vis_encode call (the tape of one call is ~2.3 MB per 96x96 image in bf16)

    def __init__(self, backbone: VitEncoder, heads: List[nn.Module], prop_dim: int, spatial_emb: int, precision: str):
        super().__init__()
        object.__setattr__(self, "_mods", [backbone] + list(heads))
        c = backbone.cfg
        self.desc = hip.VisDesc(in_ch=backbone.num_channel, img_h=backbone.img_h, img_w=backbone.img_w, embed_dim=c.embed_dim,
                                num_heads=c.num_heads, depth=c.depth, embed_norm=int(c.embed_norm), prop_dim=prop_dim,
                                spatial_emb=spatial_emb, num_img=len(heads))
        self.prop_dim, self.feat_dim = prop_dim, spatial_emb * len(heads)
        self.prec = hip.PREC_BY_NAME[precision]
        object.__setattr__(self, "_ws", hip.Workspace())
        object.__setattr__(self, "_tape_rows", 0)

    def trunk_parameters(self):
        return [p for m in self._mods for p in m.parameters()]

    def _abi_param_count(self) -> int:
        return hip.load().dppo_vis_param_count(C.byref(self.desc))

    def _abi_packed_bytes(self, prec: int, n_time: int) -> int:
        return hip.load().dppo_vis_packed_bytes(C.byref(self.desc), prec)

    def _abi_pack(self, prec: int, n_time: int, buf: torch.Tensor) -> None:
        hip.check(hip.load().dppo_vis_pack(C.byref(self.desc), prec, self.flat_params().data_ptr(), buf.data_ptr(),
                                           hip.stream()), "dppo_vis_pack")

    @property
    def obs_dim(self) -> int:
        return self.feat_dim + self.prop_dim

    def _workspace(self, rows: int, train: bool, device) -> torch.Tensor:
        wsb = hip.load().dppo_vis_workspace_bytes(C.byref(self.desc), self.prec, rows, int(train))
        if wsb < 0:
            hip.check(int(wsb), "dppo_vis_workspace_bytes")
        return self._ws.get(wsb, device)

    @staticmethod
    def _images(cond, img_cond_steps: int):
        rgb = cond["rgb"]
        assert rgb.dim() == 5, "cond['rgb'] must be (B, T, C, H, W)"
        rgb = rgb[:, -img_cond_steps:]
        if rgb.dtype not in (torch.uint8, torch.float32):
            rgb = rgb.float()
        return rgb.contiguous()

    @torch.no_grad()
    def encode(self, cond, train: bool = False, out: torch.Tensor = None) -> torch.Tensor:
        """cond {"rgb": (B,T,C,H,W) uint8 or float in 0..255, "state": (B,To,Do)} -> (B, feat_dim + To*Do) fp32.
        ``train=True`` keeps the tape for :meth:`backward` (one call of at most MAX_IMAGES observations)."""
        d = self.desc
        rgb = self._images(cond, d.in_ch // 3)
        hip.require_gpu(rgb, "VisualEncoder.encode")
        B = rgb.shape[0]
        assert tuple(rgb.shape[1:]) == (d.in_ch // 3, 3 * d.num_img, d.img_h, d.img_w), tuple(rgb.shape)
        state = cond["state"].reshape(B, -1).contiguous().float()
        assert state.shape[1] == self.prop_dim
        if out is None:
            out = torch.empty(B, self.obs_dim, dtype=torch.float32, device=rgb.device)
        assert out.shape == (B, self.obs_dim) and out.is_contiguous()
        lib, flat, pk = hip.load(), self.flat_params(), self.packed(self.prec, 0)
        if train:
            assert B <= self.MAX_IMAGES, "one training call encodes at most MAX_IMAGES observations"
        step = B if train else min(B, self.MAX_IMAGES)
        ws = self._workspace(step, train, rgb.device)
        for b0 in range(0, B, step):
            n = min(step, B - b0)
            hip.check(lib.dppo_vis_encode(C.byref(d), self.prec, flat.data_ptr(), pk.data_ptr(), rgb[b0:].data_ptr(),
                                          int(rgb.dtype == torch.uint8), state[b0:].data_ptr(), n, out[b0:].data_ptr(),
                                          self.obs_dim, int(train), ws.data_ptr(), ws.numel(), hip.stream()), "dppo_vis_encode")
        object.__setattr__(self, "_tape_rows", B if train else 0)
        return out

    @torch.no_grad()
    def backward(self, d_obs: torch.Tensor) -> torch.Tensor:
        """d loss / d obs (B, >= feat_dim) of the observations of the last ``encode(train=True)`` -> the flat gradient
        (``flat_grads()``, overwritten)."""
        B = self._tape_rows
        assert B > 0 and d_obs.shape[0] == B and d_obs.is_contiguous() and d_obs.dtype == torch.float32
        g = self.flat_grads()
        ws = self._workspace(B, True, d_obs.device)
        hip.check(hip.load().dppo_vis_backward(C.byref(self.desc), self.prec, self.flat_params().data_ptr(),
                                               self.packed(self.prec, 0).data_ptr(), d_obs.data_ptr(), d_obs.shape[1], B,
                                               g.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream()), "dppo_vis_backward")
        return g


class VisionMixin:
    """What VisionDiffusionMLP / VisionUnet1D / ViTCritic share (reference mlp_diffusion.py:42-75, unet.py:355-383,
    critic.py:131-157): the backbone, one SpatialEmb per camera, and the encoder call.  The host class is the trunk on the
    observation vector cat[feat, state]; its ``trunk_parameters()`` excludes what is registered here."""

    is_vision = True

    def _init_vision(self, backbone, cond_dim, img_cond_steps, spatial_emb, num_img, augment, dropout, precision):
        from copy import deepcopy

        from dppo_amd.model.common.modules import RandomShiftsAug, SpatialEmb
        if not spatial_emb or spatial_emb <= 1:
            raise NotImplementedError("dppo_amd: the Linear `compress` of spatial_emb = 0 is not built (every shipped cfg uses SpatialEmb)")
        if num_img not in (1, 2):
            raise NotImplementedError("dppo_amd: num_img must be 1 or 2")
        self.backbone = backbone
        mk = lambda: SpatialEmb(num_patch=backbone.num_patch, patch_dim=backbone.patch_repr_dim, prop_dim=cond_dim,
                                proj_dim=spatial_emb, dropout=dropout)
        if num_img > 1:
            self.compress1 = mk()
            self.compress2 = deepcopy(self.compress1)
        else:
            self.compress = mk()
        self.num_img, self.img_cond_steps, self.augment = num_img, img_cond_steps, bool(augment)
        if augment:
            self.aug = RandomShiftsAug(pad=4)
        self.prop_dim, self.spatial_emb = cond_dim, spatial_emb
        self._vis_precision = precision

    def _vision_modules_first(self, *first):
        """Reorder the registered children so that the state dict reads like the reference's."""
        for name in [n for n in self._modules if n not in first]:
            self._modules[name] = self._modules.pop(name)  # re-insert at the end (a plain insertion-ordered dict)

    @property
    def vis(self) -> VisualEncoder:
        v = self.__dict__.get("_vis")
        if v is None:
            heads = [self.compress] if self.num_img == 1 else [self.compress1, self.compress2]
            v = VisualEncoder(self.backbone, heads, self.prop_dim, self.spatial_emb, self._vis_precision)
            object.__setattr__(self, "_vis", v)
        return v

    def _vision_parameter_ids(self):
        return {id(p) for m in self.vis._mods for p in m.parameters()}

    def grad_views_all(self):
        by_id = {id(p): g for p, g in zip(self.vis.trunk_parameters(), self.vis.grad_views())}
        by_id.update({id(p): g for p, g in zip(self.trunk_parameters(), self.grad_views())})
        return [by_id[id(p)] for p in self.parameters()]

    def mark_updated(self):
        HipNet.mark_updated(self)
        self.vis.mark_updated()

    def encode_obs(self, cond, train: bool = False, augment: bool = None):
        """cond {"rgb", "state"} -> (B, spatial_emb * num_img + To*Do): what the trunk observes."""
        use_aug = self.augment if augment is None else augment
        if use_aug:
            rgb = cond["rgb"][:, -self.img_cond_steps:]
            B, T, Cc, H, W = rgb.shape
            # the reference shifts each camera's (t c)-stacked image as one picture (mlp_diffusion.py:129-153)
            x = rgb.float().reshape(B, T, self.num_img, 3, H, W).permute(0, 2, 1, 3, 4, 5).reshape(B * self.num_img, T * 3, H, W)
            if self.num_img > 1:  # camera 1 of the whole batch is augmented first, then camera 2 (:139-142)
                x = x.reshape(B, self.num_img, T * 3, H, W).transpose(0, 1).reshape(B * self.num_img, T * 3, H, W)
                x = torch.cat([self.aug(x[:B]), self.aug(x[B:])], 0).reshape(self.num_img, B, T, 3, H, W).permute(1, 2, 0, 3, 4, 5)
            else:
                x = self.aug(x).reshape(B, 1, T, 3, H, W).permute(0, 2, 1, 3, 4, 5)
            cond = dict(cond, rgb=x.reshape(B, T, Cc, H, W).contiguous())
        return self.vis.encode(cond, train=train)
