"""Parameter containers for the reference's MLP family, backed by one flat fp32 device buffer.

Mirrors ``dppo/model/common/mlp.py`` (reference): ``ResidualMLP`` (:84-125) with
``TwoLayerPreActivationResNetLinear`` blocks (:128-154).  The modules own nn.Parameters under the
reference's state-dict names; the arithmetic is done by the HIP library on the flat image, so these
classes have no torch ``forward`` math at all.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from dppo_amd import hip

SUPPORTED_ACT = {"ReLU": hip.ACT_RELU, "Mish": hip.ACT_MISH}


class TwoLayerPreActivationResNetLinear(nn.Module):
    """h + l2(act(norm2(l1(act(norm1(h))))))  (reference mlp.py:128-154); the norms exist only with
    ``use_layernorm`` (nn.LayerNorm(H, eps=1e-6)), registered after l1/l2 like the reference's state dict."""

    def __init__(self, hidden_dim: int, use_layernorm: bool = False):
        super().__init__()
        self.l1 = nn.Linear(hidden_dim, hidden_dim)
        self.l2 = nn.Linear(hidden_dim, hidden_dim)
        if use_layernorm:
            self.norm1 = nn.LayerNorm(hidden_dim, eps=1e-6)
            self.norm2 = nn.LayerNorm(hidden_dim, eps=1e-6)


class ResidualMLP(nn.Module):
    """Linear(in,H) -> n x block -> Linear(H,out); ``layers.{i}`` names as in reference mlp.py:103-125."""

    def __init__(self, dim_list: List[int], activation_type: str = "Mish", out_activation_type: str = "Identity",
                 use_layernorm: bool = False, use_layernorm_final: bool = False, dropout: float = 0):
        super().__init__()
        if use_layernorm_final:
            raise NotImplementedError("dppo_amd: use_layernorm_final is not built (no shipped cfg sets it)")
        if use_layernorm and dim_list[1] not in (256, 512, 1024):
            raise NotImplementedError("dppo_amd: LayerNorm blocks need a hidden width of 256, 512 or 1024")
        if dropout:
            raise NotImplementedError("Dropout not implemented for residual MLP!")  # same as the reference
        if out_activation_type != "Identity":
            raise NotImplementedError("dppo_amd: only out_activation_type='Identity' is built")
        if activation_type not in SUPPORTED_ACT:
            raise NotImplementedError(f"dppo_amd: activation {activation_type!r} not built (ReLU, Mish are)")
        hidden = dim_list[1]
        n_hidden = len(dim_list) - 3
        assert n_hidden % 2 == 0
        assert all(d == hidden for d in dim_list[1:-1]), "residual MLP needs one hidden width"
        self.layers = nn.ModuleList([nn.Linear(dim_list[0], hidden)])
        self.layers.extend([TwoLayerPreActivationResNetLinear(hidden, use_layernorm) for _ in range(1, n_hidden, 2)])
        self.layers.append(nn.Linear(hidden, dim_list[-1]))
        self.hidden, self.n_blocks = hidden, n_hidden // 2
        self.in_dim, self.out_dim = dim_list[0], dim_list[-1]
        self.act = SUPPORTED_ACT[activation_type]
        self.use_layernorm = int(bool(use_layernorm))
        self.plain = 0


class MLP(nn.Module):
    """Plain trunk: Linear(in,H) -> act -> n x [Linear(H,H) -> act] -> Linear(H,out); ``moduleList.{i}.linear_1`` names as in
    reference mlp.py:27-81.  Built for one hidden width (a multiple of 64), at least two hidden layers, no LayerNorm / dropout /
    appended inputs; runs on the library's layered GEMM path (``NetDesc.plain = 1``)."""

    def __init__(self, dim_list: List[int], activation_type: str = "Tanh", out_activation_type: str = "Identity",
                 use_layernorm: bool = False, use_layernorm_final: bool = False, dropout: float = 0, append_dim: int = 0,
                 append_layers=None, use_drop_final: bool = False, verbose: bool = False):
        super().__init__()
        if use_layernorm or use_layernorm_final or dropout or append_dim:
            raise NotImplementedError("dppo_amd: plain MLP with LayerNorm / dropout / appended inputs is not built")
        if out_activation_type != "Identity":
            raise NotImplementedError("dppo_amd: only out_activation_type='Identity' is built")
        if activation_type not in SUPPORTED_ACT:
            raise NotImplementedError(f"dppo_amd: activation {activation_type!r} not built (ReLU, Mish are)")
        hidden = dim_list[1]
        if len(dim_list) < 4 or any(d != hidden for d in dim_list[1:-1]) or hidden % 64:
            raise NotImplementedError("dppo_amd: plain MLP needs >= 2 hidden layers of one width (a multiple of 64)")
        self.moduleList = nn.ModuleList()
        for i in range(len(dim_list) - 1):
            layer = nn.Module()
            layer.linear_1 = nn.Linear(dim_list[i], dim_list[i + 1])
            self.moduleList.append(layer)
        self.hidden, self.n_blocks = hidden, len(dim_list) - 3  # hidden-to-hidden layers
        self.in_dim, self.out_dim = dim_list[0], dim_list[-1]
        self.act, self.use_layernorm, self.plain = SUPPORTED_ACT[activation_type], 0, 1


class HipNet(nn.Module):
    """Base of DiffusionMLP / CriticObs: keeps every parameter a view of ONE flat fp32 buffer (state-dict
    order = the C ABI's flat layout) and caches the packed kernel image per (precision, n_time)."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_flat", None)
        object.__setattr__(self, "_flat_grad", None)
        object.__setattr__(self, "_packed", {})
        object.__setattr__(self, "_epoch", 0)  # bumped by optimiser kernels that write the flat buffer directly

    # subclasses fill this
    def net_desc(self) -> hip.NetDesc:
        raise NotImplementedError

    def trunk_parameters(self):
        """The parameters the C ABI's flat layout covers, in state-dict order (default: all of them; a module with extra
        parameters outside the trunk -- Gaussian_MLP's logvar -- narrows this)."""
        return list(self.parameters())

    def __deepcopy__(self, memo):
        # default Module deepcopy would alias the private caches; copy parameters only
        cls = self.__class__
        new = cls.__new__(cls)
        nn.Module.__init__(new)
        import copy
        for k, v in self.__dict__.items():
            if k in ("_flat", "_flat_grad", "_packed", "_epoch", "_plist", "_last_off", "_desc_cache", "_packed_bytes", "_vis"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        object.__setattr__(new, "_flat", None)
        object.__setattr__(new, "_flat_grad", None)
        object.__setattr__(new, "_packed", {})
        object.__setattr__(new, "_epoch", 0)
        return new

    def flat_params(self) -> torch.Tensor:
        # hot path (called several times per sampler / update call): the cached flat image is valid as long as the first
        # and the last parameter still sit at their offsets in it -- `.to()` / a re-assigned `.data` move all of them
        flat, ps = self._flat, self.__dict__.get("_plist")
        if flat is not None and ps is not None:
            first, last = ps[0], ps[-1]
            if first.data_ptr() == flat.data_ptr() and last.data_ptr() == flat.data_ptr() + 4 * self._last_off:
                return flat
        ps = self.trunk_parameters()
        object.__setattr__(self, "_plist", ps)
        object.__setattr__(self, "_last_off", sum(p.numel() for p in ps[:-1]))
        flat = self._flat
        ok = flat is not None and flat.device == ps[0].device
        if ok:
            off = 0
            base = flat.data_ptr()
            for p in ps:
                if p.data_ptr() != base + 4 * off or not p.is_contiguous():
                    ok = False
                    break
                off += p.numel()
        if not ok:
            total = sum(p.numel() for p in ps)
            flat = torch.empty(total, dtype=torch.float32, device=ps[0].device)
            off = 0
            for p in ps:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1).float())
                p.data = flat[off:off + n].view(p.shape)
                off += n
            object.__setattr__(self, "_flat", flat)
            self._packed.clear()
            if flat.is_cuda:
                want = self._abi_param_count()
                assert want == total, f"flat layout mismatch: python {total} vs C ABI {want}"
        return flat

    # the three ABI calls that depend on the network family (MLP trunks here; the conv denoiser overrides them)
    def _abi_param_count(self) -> int:
        d = self.net_desc()
        return hip.load().dppo_net_param_count(C.byref(d))

    def _abi_packed_bytes(self, prec: int, n_time: int) -> int:
        d = self.net_desc()
        return hip.load().dppo_packed_bytes(C.byref(d), prec, n_time)

    def _abi_pack(self, prec: int, n_time: int, buf: torch.Tensor) -> None:
        d = self.net_desc()
        hip.check(hip.load().dppo_pack_net(C.byref(d), prec, n_time, self.flat_params().data_ptr(), buf.data_ptr(),
                                           hip.stream()), "dppo_pack_net")

    def flat_grads(self) -> torch.Tensor:
        flat = self.flat_params()
        g = self._flat_grad
        if g is None or g.device != flat.device or g.numel() != flat.numel():
            g = torch.zeros_like(flat)
            object.__setattr__(self, "_flat_grad", g)
        return g

    def grad_views(self) -> List[torch.Tensor]:
        g = self.flat_grads()
        out, off = [], 0
        for p in self.trunk_parameters():
            out.append(g[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return out

    def grad_views_all(self) -> List[torch.Tensor]:
        """One gradient view per entry of ``self.parameters()`` (a pixel network adds its encoder's, see VisionMixin)."""
        return self.grad_views()

    def mark_updated(self):
        """Call after a kernel wrote the flat parameter buffer behind torch's back (fused AdamW)."""
        object.__setattr__(self, "_epoch", self._epoch + 1)

    def _packed_slot(self, prec: int, n_time: int):
        """(buffer, stamp, stale): the packed image's buffer for (prec, n_time), allocated if needed, and whether the
        parameters changed since it was last written."""
        flat = self.flat_params()
        hip.require_gpu(flat, type(self).__name__)
        key = (prec, n_time)
        # the Parameters are `.data` views of `flat`: in-place writes through them (torch.optim steps, load_state_dict)
        # advance THEIR version counters, never `flat._version` -- so the stamp folds every parameter's counter in
        ver = 0
        for p in self._plist:
            ver += p._version
        stamp = (flat.data_ptr(), flat._version, ver, self._epoch)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1], stamp, False
        nbytes = self.__dict__.setdefault("_packed_bytes", {}).get(key)
        if nbytes is None:
            nbytes = self._abi_packed_bytes(prec, n_time)
            if nbytes < 0:
                hip.check(int(nbytes), "dppo_packed_bytes")
            self._packed_bytes[key] = nbytes
        buf = hit[1] if hit is not None and hit[1].numel() == nbytes else torch.empty(
            nbytes, dtype=torch.uint8, device=flat.device)
        return buf, stamp, True

    def packed(self, prec: int, n_time: int) -> torch.Tensor:
        buf, stamp, stale = self._packed_slot(prec, n_time)
        if stale:
            self._abi_pack(prec, n_time, buf)
            self._packed[(prec, n_time)] = (stamp, buf)
        return buf


def pack_pair(net0: "HipNet", n_time0: int, net1: "HipNet", n_time1: int, prec: int) -> None:
    """Re-pack two networks after an optimiser step: when both images are stale, both composites and both images go out
    in two launches (dppo_pack_nets) instead of four."""
    if any(getattr(n, a, False) for n in (net0, net1) for a in ("is_unet", "is_composite")):  # own pack entry / two images
        net0.packed(prec, n_time0)
        net1.packed(prec, n_time1)
        return
    b0, s0, stale0 = net0._packed_slot(prec, n_time0)
    b1, s1, stale1 = net1._packed_slot(prec, n_time1)
    if stale0 and stale1:
        hip.check(hip.load().dppo_pack_nets(
            C.byref(net0.net_desc()), n_time0, net0.flat_params().data_ptr(), b0.data_ptr(),
            C.byref(net1.net_desc()), n_time1, net1.flat_params().data_ptr(), b1.data_ptr(), prec, hip.stream()),
            "dppo_pack_nets")
        net0._packed[(prec, n_time0)] = (s0, b0)
        net1._packed[(prec, n_time1)] = (s1, b1)
        return
    net0.packed(prec, n_time0)
    net1.packed(prec, n_time1)
