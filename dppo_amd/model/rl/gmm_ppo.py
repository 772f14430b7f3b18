"""PPO for a mixture-of-Gaussians policy.  Mirrors ``dppo/model/rl/gmm_ppo.py:19-112`` (reference ``PPO_GMM``): same constructor,
same ``loss`` signature and 8-tuple.  Forward AND backward are one library call (``dppo_gmm_ppo_loss_fwd_bwd``: both actor trunks
and the critic forward, the mixture loss epilogue of csrc/gmm.hip, three backward passes).

Entropy term: the reference's agent adds ``entropy_loss * ent_coef`` to the loss; the mixture entropy depends on the logits trunk,
so its gradient must ride the same backward pass.  Set ``model.ent_coef`` (the agent does) and the gradients the call returns are
those of ``pg_loss + ent_coef * entropy_loss``; ``loss()`` attaches them to ``pg_loss`` and returns ``entropy_loss`` detached."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from dppo_amd import hip
from dppo_amd.model.rl.gmm_vpg import VPG_GMM


class _FusedGmmLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stats, a_grads, c_grads, lv_grad, n_a, n_c, *params):
        ctx.a_grads, ctx.c_grads, ctx.lv_grad = a_grads, c_grads, lv_grad
        return stats[hip.STAT_PG_LOSS].float().clone(), stats[hip.STAT_V_LOSS].float().clone()

    @staticmethod
    def backward(ctx, g_pg, g_v):
        ga = [g * g_pg for g in ctx.a_grads]
        gc = [g * g_v for g in ctx.c_grads]
        glv = [ctx.lv_grad * g_pg] if ctx.lv_grad is not None else []
        return (None,) * 6 + (*ga, *gc, *glv)


class PPO_GMM(VPG_GMM):
    entropy_in_kernel = True  # the agent must not add its own d entropy / d logvar term

    def __init__(self, clip_ploss_coef: float, clip_vloss_coef: Optional[float] = None, norm_adv: Optional[bool] = True, **kwargs):
        super().__init__(**kwargs)
        self.norm_adv, self.clip_ploss_coef, self.clip_vloss_coef = norm_adv, clip_ploss_coef, clip_vloss_coef
        self.ent_coef = 0.0
        object.__setattr__(self, "_ws_ppo", hip.Workspace())
        object.__setattr__(self, "_stats", None)
        object.__setattr__(self, "_lv_grad", None)

    def _run(self, obs, actions, returns, oldvalues, adv, oldlogp, global_moments=None):
        lib, dev = hip.load(), obs.device
        net, cr = self.actor_ft, self.critic
        N = obs.shape[0]
        if self._stats is None or self._stats.device != dev:
            object.__setattr__(self, "_stats", torch.zeros(hip.GAUSS_STAT_COUNT, dtype=torch.float64, device=dev))
            object.__setattr__(self, "_lv_grad", torch.zeros(net.action_dim * net.num_modes, device=dev))
        cfg = net.gmm_cfg(ent_coef=self.ent_coef)
        cfg.norm_adv, cfg.has_vclip = int(bool(self.norm_adv)), int(self.clip_vloss_coef is not None)
        cfg.clip_ploss_coef, cfg.clip_vloss_coef = float(self.clip_ploss_coef), float(self.clip_vloss_coef or 0.0)
        m, w = net.mean_net, net.weights_net
        wsb = lib.dppo_gmm_workspace_bytes(C.byref(m.net_desc()), C.byref(w.net_desc()), C.byref(cr.net_desc()), self.prec, N)
        if wsb < 0:
            hip.check(int(wsb), "dppo_gmm_workspace_bytes")
        ws = self._ws_ppo.get(wsb, dev)
        hip.check(lib.dppo_gmm_ppo_loss_fwd_bwd(
            C.byref(m.net_desc()), C.byref(w.net_desc()), C.byref(cr.net_desc()), self.prec, m.flat_params().data_ptr(),
            m.packed(self.prec, 0).data_ptr(), w.flat_params().data_ptr(), w.packed(self.prec, 0).data_ptr(),
            cr.flat_params().data_ptr(), cr.packed(self.prec, 0).data_ptr(), C.byref(cfg), net.logvar_ptr(), hip.ptr(obs),
            hip.ptr(actions), hip.ptr(returns), hip.ptr(oldvalues), hip.ptr(adv), hip.ptr(oldlogp), N, hip.ptr(global_moments),
            m.flat_grads().data_ptr(), w.flat_grads().data_ptr(), cr.flat_grads().data_ptr(), self._lv_grad.data_ptr(),
            self._stats.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream()), "dppo_gmm_ppo_loss_fwd_bwd")
        return self._stats

    def ppo_update(self, obs, actions, returns, oldvalues, adv, oldlogp, global_moments=None):
        """One minibatch, no host sync: gradients of pg_loss + ent_coef * entropy_loss land in ``actor_ft.flat_grads()`` (both
        trunks) and ``_lv_grad``, those of v_loss in ``critic.flat_grads()``; returns the device statistics."""
        hip.require_gpu(obs, "PPO_GMM.ppo_update")
        return self._run(obs, actions, returns, oldvalues, adv, oldlogp, global_moments)

    def loss(self, obs, actions, returns, oldvalues, advantages, oldlogprobs, **kwargs):
        """Same arguments / 8-tuple as the reference: (pg_loss, entropy_loss, v_loss, clipfrac, approx_kl, ratio, bc_loss, std)."""
        state = obs["state"]
        hip.require_gpu(state, "PPO_GMM.loss")
        N = state.shape[0]
        net = self.actor_ft
        AF = net.action_dim * net.horizon_steps
        f = lambda t, *shape: t.reshape(*shape).contiguous().float()
        stats = self._run(f(state, N, -1), f(actions, N, AF), f(returns, N), f(oldvalues, N), f(advantages, N), f(oldlogprobs, N))
        a_params, c_params = net.trunk_parameters(), list(self.critic.parameters())
        extra = [net.logvar] if net.learn_fixed_std else []
        pg, vl = _FusedGmmLoss.apply(stats, net.grad_views(), self.critic.grad_views_all(),
                                     self._lv_grad.clone() if net.learn_fixed_std else None, len(a_params), len(c_params),
                                     *a_params, *c_params, *extra)
        host = stats.tolist()
        return (pg, torch.tensor(-host[hip.GAUSS_STAT_ENTROPY], device=state.device), vl, host[hip.STAT_CLIPFRAC],
                host[hip.STAT_APPROX_KL], host[hip.STAT_RATIO], 0, host[hip.GAUSS_STAT_STD])
