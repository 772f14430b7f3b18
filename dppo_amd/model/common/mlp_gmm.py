"""Mixture-of-Gaussians policy head.  Mirrors ``dppo/model/common/mlp_gmm.py:11-110`` (reference ``GMM_MLP``): component means
tanh(mlp_mean(s)) of shape (B, num_modes, Ta*Da), a fixed or learned per-(mode, action dimension) std, mixture logits
mlp_weights(s).  Parameter names match the reference state dict (``logvar``, ``logvar_min``, ``logvar_max``, ``mlp_mean.*``,
``mlp_weights.*``).  The two trunks are separate networks for the library (``dppo_gmm_*``, csrc/gmm.hip); their parameters sit
back to back in ONE flat fp32 buffer so that the agent's optimiser and data-parallel bucket see a single actor.

Built: ``fixed_std`` given (all 7 shipped ``ft_ppo_gmm_mlp`` cfgs).  Not built: the state-dependent ``mlp_logvar`` head."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from dppo_amd import hip
from dppo_amd.model.common.mlp import MLP, HipNet, ResidualMLP


class _Trunk(HipNet):
    """One of the two trunks as the library sees it: a kind-1 network on the observation whose flat buffer is a slice of the
    owner's."""

    def __init__(self, module, cond_dim):
        super().__init__()
        object.__setattr__(self, "_m", module)
        self.cond_dim = cond_dim

    def trunk_parameters(self):
        return list(self._m.parameters())

    def net_desc(self) -> hip.NetDesc:
        d = self.__dict__.get("_desc_cache")
        if d is None:
            m = self._m
            d = hip.NetDesc(kind=1, in_dim=self.cond_dim, hidden=m.hidden, n_blocks=m.n_blocks, out_dim=m.out_dim, act=m.act,
                            time_dim=0, act_flat=0, cond_dim=self.cond_dim, cond_hidden=0, cond_out=0,
                            use_layernorm=m.use_layernorm, plain=m.plain)
            object.__setattr__(self, "_desc_cache", d)
        return d


class GMM_MLP(HipNet):
    is_composite = True  # two kernel images: pack_pair() packs it through .packed()

    def __init__(self, action_dim, horizon_steps, cond_dim=None, mlp_dims=[256, 256, 256], num_modes=5, activation_type="Mish",
                 residual_style=False, use_layernorm=False, fixed_std=None, learn_fixed_std=False, std_min=0.01, std_max=1,
                 precision="bf16"):
        super().__init__()
        if fixed_std is None:
            raise NotImplementedError("dppo_amd: GMM_MLP with a state-dependent logvar head (fixed_std=None) is not built")
        if num_modes > 8:
            raise NotImplementedError("dppo_amd: GMM_MLP is built for up to 8 modes")
        self.action_dim, self.horizon_steps, self.cond_dim, self.num_modes = action_dim, horizon_steps, cond_dim, num_modes
        model = ResidualMLP if residual_style else MLP
        out_dim = action_dim * horizon_steps * num_modes
        self.mlp_mean = model([cond_dim] + list(mlp_dims) + [out_dim], activation_type=activation_type,
                              out_activation_type="Identity", use_layernorm=use_layernorm)
        if learn_fixed_std:  # separate for each action dimension and mode (reference :56-63)
            self.logvar = nn.Parameter(torch.log(torch.tensor([fixed_std ** 2 for _ in range(action_dim * num_modes)])),
                                       requires_grad=True)
        self.logvar_min = nn.Parameter(torch.log(torch.tensor(std_min ** 2)), requires_grad=False)
        self.logvar_max = nn.Parameter(torch.log(torch.tensor(std_max ** 2)), requires_grad=False)
        self.use_fixed_std, self.fixed_std, self.learn_fixed_std = True, fixed_std, learn_fixed_std
        self.mlp_weights = model([cond_dim] + list(mlp_dims) + [num_modes], activation_type=activation_type,
                                 out_activation_type="Identity", use_layernorm=use_layernorm)
        self.prec = hip.PREC_BY_NAME[precision]

    # ---- the flat buffer covers [mlp_mean | mlp_weights]; the two trunk views live inside it
    def trunk_parameters(self):
        return list(self.mlp_mean.parameters()) + list(self.mlp_weights.parameters())

    def _abi_param_count(self) -> int:
        lib = hip.load()
        return sum(lib.dppo_net_param_count(C.byref(t.net_desc())) for t in self._trunks(bind=False))

    def _trunks(self, bind=True):
        t = self.__dict__.get("_trunk_nets")
        if t is None:
            t = (_Trunk(self.mlp_mean, self.cond_dim), _Trunk(self.mlp_weights, self.cond_dim))
            object.__setattr__(self, "_trunk_nets", t)
        if bind:  # (re-)attach the slices: after .to(device) / a re-homed flat buffer
            flat, grads = self.flat_params(), self.flat_grads()
            n0 = sum(p.numel() for p in self.mlp_mean.parameters())
            for net, lo, hi in ((t[0], 0, n0), (t[1], n0, flat.numel())):
                if (net._flat is None or net._flat.data_ptr() != flat.data_ptr() + 4 * lo or
                        net._flat_grad.data_ptr() != grads.data_ptr() + 4 * lo):  # (a data-parallel bucket re-homes the gradients)
                    object.__setattr__(net, "_flat", flat[lo:hi])
                    object.__setattr__(net, "_flat_grad", grads[lo:hi])
                    net._packed.clear()
        return t

    @property
    def mean_net(self) -> _Trunk:
        return self._trunks()[0]

    @property
    def weights_net(self) -> _Trunk:
        return self._trunks()[1]

    def net_desc(self):
        raise TypeError("GMM_MLP is two networks: use mean_net / weights_net")

    def packed(self, prec: int, n_time: int = 0):
        a, b = self._trunks()
        return a.packed(prec, 0), b.packed(prec, 0)

    def mark_updated(self):
        HipNet.mark_updated(self)
        for t in self._trunks(bind=False):
            t.mark_updated()

    def gmm_cfg(self, deterministic=False, ent_coef=0.0) -> hip.GmmCfg:
        return hip.GmmCfg(horizon_steps=self.horizon_steps, action_dim=self.action_dim, num_modes=self.num_modes,
                          std_mode=1 if self.learn_fixed_std else 0, norm_adv=1, has_vclip=0, deterministic=int(bool(deterministic)),
                          pad=0, fixed_std=float(self.fixed_std), logvar_min=float(self.logvar_min), logvar_max=float(self.logvar_max),
                          ent_coef=float(ent_coef), clip_ploss_coef=0.0, clip_vloss_coef=0.0, seed_lo=0, seed_hi=0)

    def logvar_ptr(self):
        return self.logvar.data_ptr() if self.learn_fixed_std else None

    def forward(self, cond):
        raise NotImplementedError("dppo_amd: GMM_MLP's arithmetic runs inside dppo_gmm_* (GMMModel / VPG_GMM / PPO_GMM)")
