"""PPO for a Gaussian policy.  Mirrors ``dppo/model/rl/gaussian_ppo.py:19-128`` (reference ``PPO_Gaussian``): same
constructor, same ``loss`` signature and 8-tuple.  Forward AND backward are one library call
(``dppo_gaussian_ppo_loss_fwd_bwd``: trunk forward of actor_ft and critic on the fused kernels, the loss epilogue of
csrc/gaussian.hip, fused backward, grouped weight-gradient GEMMs); ``pg_loss`` / ``entropy_loss`` / ``v_loss`` come back
attached to autograd through a Function that hands out the gradients the kernels already produced."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from dppo_amd import hip
from dppo_amd.model.rl.gaussian_vpg import VPG_Gaussian


class _FusedGaussLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stats, a_grads, c_grads, lv_pg, lv_ent, n_a, n_c, *params):
        ctx.a_grads, ctx.c_grads, ctx.lv_pg, ctx.lv_ent, ctx.n = a_grads, c_grads, lv_pg, lv_ent, (n_a, n_c)
        return (stats[hip.STAT_PG_LOSS].float().clone(), -stats[hip.GAUSS_STAT_ENTROPY].float().clone(),
                stats[hip.STAT_V_LOSS].float().clone())

    @staticmethod
    def backward(ctx, g_pg, g_ent, g_v):
        ga = [g * g_pg for g in ctx.a_grads]
        gc = [g * g_v for g in ctx.c_grads]
        glv = [ctx.lv_pg * g_pg + ctx.lv_ent * g_ent] if ctx.lv_pg is not None else []
        return (None,) * 7 + (*ga, *gc, *glv)


class PPO_Gaussian(VPG_Gaussian):
    # data parallel (dppo_amd.parallel.DataParallel): statistics slots 5..7 are rank-global values (advantage mean / std and
    # the entropy, which depends on sigma only), averaged -- not summed -- over ranks
    dp_avg_stats = 3

    def __init__(self, clip_ploss_coef: float, clip_vloss_coef: Optional[float] = None, norm_adv: Optional[bool] = True,
                 **kwargs):
        super().__init__(**kwargs)
        self.norm_adv, self.clip_ploss_coef, self.clip_vloss_coef = norm_adv, clip_ploss_coef, clip_vloss_coef
        object.__setattr__(self, "_ws_ppo", hip.Workspace())
        object.__setattr__(self, "_stats", None)
        object.__setattr__(self, "_lv_grad", None)

    def _run(self, obs, actions, returns, oldvalues, adv, oldlogp, global_moments=None):
        """obs: (N, To*Do) tensor, or for pixel networks the cond dict {"rgb", "state"}: both encoders then run with a tape,
        the loss entry returns d loss / d observation and the encoders' backward fills their own flat gradients."""
        lib = hip.load()
        net = self.actor_ft
        vision = isinstance(obs, dict)
        if vision:
            cond = obs
            obs = net.encode_obs(cond, train=True)
            obs_c = self.critic.encode_obs(cond, train=True, augment=None)  # None: critic.augment decides, as in the reference's self.critic(obs)
        dev = obs.device
        da, dc = net.net_desc(), self.critic.net_desc()
        N = obs.shape[0]
        if self._stats is None or self._stats.device != dev:
            object.__setattr__(self, "_stats", torch.zeros(hip.GAUSS_STAT_COUNT, dtype=torch.float64, device=dev))
            object.__setattr__(self, "_lv_grad", torch.zeros(net.action_dim, device=dev))
        cfg = net.gaussian_cfg(randn_clip=self.randn_clip_value)
        cfg.norm_adv, cfg.has_vclip = int(bool(self.norm_adv)), int(self.clip_vloss_coef is not None)
        cfg.clip_ploss_coef, cfg.clip_vloss_coef = float(self.clip_ploss_coef), float(self.clip_vloss_coef or 0.0)
        wsb = lib.dppo_gaussian_workspace_bytes(C.byref(da), C.byref(dc), self.prec, N)
        if wsb < 0:
            hip.check(int(wsb), "dppo_gaussian_workspace_bytes")
        ws = self._ws_ppo.get(wsb, dev)
        args = (C.byref(da), C.byref(dc), self.prec, net.flat_params().data_ptr(), net.packed(self.prec, 0).data_ptr(),
                self.critic.flat_params().data_ptr(), self.critic.packed(self.prec, 0).data_ptr(), C.byref(cfg),
                net.logvar_ptr(), hip.ptr(obs), hip.ptr(actions), hip.ptr(returns), hip.ptr(oldvalues), hip.ptr(adv),
                hip.ptr(oldlogp), N, hip.ptr(global_moments), net.flat_grads().data_ptr(), self.critic.flat_grads().data_ptr(),
                self._lv_grad.data_ptr(), self._stats.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream())
        if vision:
            d_a, d_c = torch.empty_like(obs), torch.empty_like(obs_c)
            io = hip.ObsIO(obs_c.data_ptr(), d_a.data_ptr(), d_c.data_ptr())
            hip.check(lib.dppo_gaussian_ppo_loss_fwd_bwd_obs(*args, C.byref(io)), "dppo_gaussian_ppo_loss_fwd_bwd_obs")
            net.vis.backward(d_a)
            self.critic.vis.backward(d_c)
        else:
            hip.check(lib.dppo_gaussian_ppo_loss_fwd_bwd(*args), "dppo_gaussian_ppo_loss_fwd_bwd")
        return self._stats

    def ppo_update(self, obs, actions, returns, oldvalues, adv, oldlogp, global_moments=None):
        """One minibatch, no host sync: gradients land in the flat gradient buffers of ``actor_ft`` / ``critic`` (and
        ``_lv_grad`` for a learned std); returns the device statistics (float64[GAUSS_STAT_COUNT])."""
        hip.require_gpu(obs["state"] if isinstance(obs, dict) else obs, "PPO_Gaussian.ppo_update")
        return self._run(obs, actions, returns, oldvalues, adv, oldlogp, global_moments)

    def loss(self, obs, actions, returns, oldvalues, advantages, oldlogprobs, use_bc_loss=False):
        """Same arguments / 8-tuple as the reference: (pg_loss, entropy_loss, v_loss, clipfrac, approx_kl, ratio, bc_loss,
        std); the first three carry grad."""
        if use_bc_loss:
            raise NotImplementedError("dppo_amd: the BC term of PPO_Gaussian is not built (no shipped cfg enables it)")
        state = obs["state"]
        hip.require_gpu(state, "PPO_Gaussian.loss")
        N = state.shape[0]
        net = self.actor_ft
        AF = net.action_dim * net.horizon_steps
        f = lambda t, *shape: t.reshape(*shape).contiguous().float()
        vision = getattr(net, "is_vision", False)
        stats = self._run(obs if vision else f(state, N, -1), f(actions, N, AF), f(returns, N), f(oldvalues, N), f(advantages, N),
                          f(oldlogprobs, N))
        a_params, a_grads = net.trunk_parameters(), net.grad_views()
        if vision:  # + the encoder's parameters and the gradients its backward produced
            a_params, a_grads = net.vis.trunk_parameters() + a_params, net.vis.grad_views() + a_grads
        c_params = list(self.critic.parameters())
        lv_pg = lv_ent = None
        extra = []
        if net.learn_fixed_std:
            lv = net.logvar.detach()
            inside = ((lv >= net.logvar_min) & (lv <= net.logvar_max)).float()
            lv_pg = self._lv_grad.clone()
            lv_ent = -0.5 / net.action_dim * inside  # d(-entropy) / d logvar_j, through the clamp
            extra = [net.logvar]
        pg, ent, vl = _FusedGaussLoss.apply(stats, a_grads, self.critic.grad_views_all(), lv_pg, lv_ent, len(a_params),
                                            len(c_params), *a_params, *c_params, *extra)
        host = stats.tolist()  # one D2H sync, like the reference's .item() calls
        return (pg, ent, vl, host[hip.STAT_CLIPFRAC], host[hip.STAT_APPROX_KL], host[hip.STAT_RATIO], 0.0,
                host[hip.GAUSS_STAT_STD])
