"""DPPO: PPO on the denoising-chain MDP.

Mirrors ``dppo/model/diffusion/diffusion_ppo.py:24-199`` (reference ``PPODiffusion``).  ``loss`` keeps the
reference's signature and 8-tuple; its forward AND backward are one call into the HIP library
(``dppo_ppo_loss_fwd_bwd``: fused gather -> MFMA GEMM chain -> fused log-prob / ratio / clip / value epilogue
-> backward GEMMs).  ``pg_loss`` / ``v_loss`` come back attached to autograd through a Function whose backward
just hands out the gradients the kernel already produced, so ``loss.backward()`` + ``torch.optim.AdamW`` work
unchanged.  ``ppo_update`` is the fused-gather fast path used by dppo_amd's own agent.
"""
from __future__ import annotations

import ctypes as C
import logging
import math
from typing import Optional

import torch

from dppo_amd import hip
from dppo_amd.model.diffusion.diffusion_vpg import VPGDiffusion

log = logging.getLogger(__name__)


class _FusedPPOLoss(torch.autograd.Function):
    """(pg_loss, v_loss) with gradients precomputed by the HIP kernel.

    pg_loss depends on the actor_ft parameters only and v_loss on the critic parameters only, so the upstream
    gradients are two scalars and backward is a scale of the stored flat gradients.
    """

    @staticmethod
    def forward(ctx, stats, actor_grads, critic_grads, n_actor, *params):
        ctx.actor_grads, ctx.critic_grads, ctx.n_actor = actor_grads, critic_grads, n_actor
        pg = stats[hip.STAT_PG_LOSS].float().clone()
        vl = stats[hip.STAT_V_LOSS].float().clone()
        return pg, vl

    @staticmethod
    def backward(ctx, g_pg, g_v):
        ga = [g * g_pg for g in ctx.actor_grads]
        gc = [g * g_v for g in ctx.critic_grads]
        return (None, None, None, None, *ga, *gc)


class _FusedBCLoss(torch.autograd.Function):
    """bc_loss with the gradient w.r.t. the actor_ft parameters precomputed by the HIP kernel."""

    @staticmethod
    def forward(ctx, value, grads, *params):
        ctx.grads = grads
        return value.float().clone()

    @staticmethod
    def backward(ctx, g):
        return (None, None, *[x * g for x in ctx.grads])


class PPODiffusion(VPGDiffusion):
    def __init__(self, gamma_denoising: float, clip_ploss_coef: float, clip_ploss_coef_base: float = 1e-3,
                 clip_ploss_coef_rate: float = 3, clip_vloss_coef: Optional[float] = None,
                 clip_advantage_lower_quantile: float = 0, clip_advantage_upper_quantile: float = 1,
                 norm_adv: bool = True, **kwargs):
        super().__init__(**kwargs)
        self.norm_adv = norm_adv
        self.clip_ploss_coef = clip_ploss_coef
        self.clip_ploss_coef_base = clip_ploss_coef_base
        self.clip_ploss_coef_rate = clip_ploss_coef_rate
        self.clip_vloss_coef = clip_vloss_coef
        self.gamma_denoising = gamma_denoising
        self.clip_advantage_lower_quantile = clip_advantage_lower_quantile
        self.clip_advantage_upper_quantile = clip_advantage_upper_quantile
        object.__setattr__(self, "_ws_ppo", hip.Workspace())
        object.__setattr__(self, "_ws_bc", hip.Workspace())
        object.__setattr__(self, "_stats", None)
        object.__setattr__(self, "_bc_grad", None)
        object.__setattr__(self, "_bc_value", None)

    # ------------------------------------------------------------------ helpers
    def _ppo_cfg(self, reward_horizon: int, adv_gathered: Optional[torch.Tensor]) -> hip.PpoCfg:
        cfg = hip.PpoCfg(
            ft_denoising_steps=self.ft_denoising_steps, horizon_steps=self.horizon_steps, action_dim=self.action_dim,
            reward_horizon=int(reward_horizon), norm_adv=int(bool(self.norm_adv)), has_adv_clip=0,
            has_vclip=int(self.clip_vloss_coef is not None), pad=0, gamma_denoising=float(self.gamma_denoising),
            clip_ploss_coef=float(self.clip_ploss_coef), clip_ploss_coef_base=float(self.clip_ploss_coef_base),
            clip_ploss_coef_rate=float(self.clip_ploss_coef_rate),
            clip_vloss_coef=float(self.clip_vloss_coef or 0.0), adv_clip_lo=0.0, adv_clip_hi=0.0)
        lo_q, hi_q = self.clip_advantage_lower_quantile, self.clip_advantage_upper_quantile
        if (lo_q, hi_q) != (0, 1):
            # reference :129-135 -- quantiles of the (normalised) minibatch advantages; min/max (the default) is a no-op
            a = adv_gathered.float()
            if self.norm_adv:
                a = (a - a.mean()) / (a.std() + 1e-8)
            cfg.has_adv_clip = 1
            cfg.adv_clip_lo = float(torch.quantile(a, lo_q))
            cfg.adv_clip_hi = float(torch.quantile(a, hi_q))
        return cfg

    def _run_ppo(self, obs, chains, returns, values, adv, logprobs, inds, kinds, N, reward_horizon, adv_gathered,
                 global_moments=None, critic_hook=None):
        lib = hip.load()
        dev = obs.device
        da, dc = self.actor_ft.net_desc(), self.critic.net_desc()
        K = self.denoising_steps
        if self._stats is None or self._stats.device != dev:
            object.__setattr__(self, "_stats", torch.zeros(hip.STAT_COUNT, dtype=torch.float64, device=dev))
        pcfg = self._ppo_cfg(reward_horizon, adv_gathered)
        dcfg = self.diffusion_cfg()
        ks = self._logprob_schedule(dev)
        unet = getattr(self.actor_ft, "is_unet", False)  # conv denoiser: same call, dppo_unet_* entry points
        ws_bytes, entry = (lib.dppo_unet_ppo_workspace_bytes, lib.dppo_unet_ppo_loss_fwd_bwd) if unet else (
            lib.dppo_ppo_workspace_bytes, lib.dppo_ppo_loss_fwd_bwd)
        wsb = ws_bytes(C.byref(da), C.byref(dc), self.prec, N)
        if wsb < 0:
            hip.check(int(wsb), "dppo_ppo_workspace_bytes")
        ws = self._ws_ppo.get(wsb, dev)
        ga, gc = self.actor_ft.flat_grads(), self.critic.flat_grads()
        if critic_hook is not None and not unet:
            # data parallel: the library calls back once every launch that writes the critic's gradient is queued, with the
            # stream it queued them on (include/dppo_hip.h, dppo_ppo_loss_fwd_bwd_dp); an exception raised in the callback
            # surfaces here, after the call
            err = []

            def _cb(_user, side):
                try:
                    critic_hook(int(side or 0))
                except BaseException as e:  # noqa: BLE001 (ctypes would only print it)
                    err.append(e)
            hook = hip.DpHook(hip.DP_HOOK_FN(_cb), None)
            rc = lib.dppo_ppo_loss_fwd_bwd_dp(
                C.byref(da), C.byref(dc), self.prec, self.actor_ft.flat_params().data_ptr(),
                self.actor_ft.packed(self.prec, K).data_ptr(), self.critic.flat_params().data_ptr(),
                self.critic.packed(self.prec, 0).data_ptr(), C.byref(dcfg), C.byref(pcfg), ks.data_ptr(), hip.ptr(obs),
                hip.ptr(chains), hip.ptr(returns), hip.ptr(values), hip.ptr(adv), hip.ptr(logprobs), hip.ptr(inds),
                hip.ptr(kinds), N, hip.ptr(global_moments), ga.data_ptr(), gc.data_ptr(), self._stats.data_ptr(),
                ws.data_ptr(), ws.numel(), hip.stream(), C.byref(hook))
            if err:
                raise err[0]
            hip.check(rc, "dppo_ppo_loss_fwd_bwd_dp")
            return self._stats
        hip.check(entry(
            C.byref(da), C.byref(dc), self.prec, self.actor_ft.flat_params().data_ptr(),
            self.actor_ft.packed(self.prec, K).data_ptr(), self.critic.flat_params().data_ptr(),
            self.critic.packed(self.prec, 0).data_ptr(), C.byref(dcfg), C.byref(pcfg), ks.data_ptr(), hip.ptr(obs),
            hip.ptr(chains), hip.ptr(returns), hip.ptr(values), hip.ptr(adv), hip.ptr(logprobs), hip.ptr(inds),
            hip.ptr(kinds), N, hip.ptr(global_moments), ga.data_ptr(), gc.data_ptr(), self._stats.data_ptr(),
            ws.data_ptr(), ws.numel(), hip.stream()), "dppo_ppo_loss_fwd_bwd")
        return self._stats

    def _run_ppo_vision(self, cond, pairs, returns, values, adv, logprobs, kinds, N, reward_horizon, adv_gathered,
                        global_moments=None):
        """The PPO loss with pixel networks: both encoders run with a tape, the fused loss entry returns d loss / d obs for
        each of them next to the trunk gradients, and the encoders' backward turns those into their own flat gradients
        (``actor_ft.vis.flat_grads()`` / ``critic.vis.flat_grads()``)."""
        lib = hip.load()
        a, c = self.actor_ft, self.critic
        obs_a = a.encode_obs(cond, train=True)
        obs_c = c.encode_obs(cond, train=True, augment=False)
        dev = obs_a.device
        da, dc = a.net_desc(), c.net_desc()
        K = self.denoising_steps
        if self._stats is None or self._stats.device != dev:
            object.__setattr__(self, "_stats", torch.zeros(hip.STAT_COUNT, dtype=torch.float64, device=dev))
        pcfg = self._ppo_cfg(reward_horizon, adv_gathered)
        dcfg = self.diffusion_cfg()
        ks = self._logprob_schedule(dev)
        unet = getattr(a, "is_unet", False)
        ws_bytes, entry = (lib.dppo_unet_ppo_workspace_bytes, lib.dppo_unet_ppo_loss_fwd_bwd_obs) if unet else (
            lib.dppo_ppo_workspace_bytes, lib.dppo_ppo_loss_fwd_bwd_obs)
        wsb = ws_bytes(C.byref(da), C.byref(dc), self.prec, N)
        if wsb < 0:
            hip.check(int(wsb), "dppo_ppo_workspace_bytes")
        ws = self._ws_ppo.get(wsb, dev)
        d_a, d_c = torch.empty_like(obs_a), torch.empty_like(obs_c)
        io = hip.ObsIO(obs_c.data_ptr(), d_a.data_ptr(), d_c.data_ptr())
        ga, gc = a.flat_grads(), c.flat_grads()
        hip.check(entry(
            C.byref(da), C.byref(dc), self.prec, a.flat_params().data_ptr(), a.packed(self.prec, K).data_ptr(),
            c.flat_params().data_ptr(), c.packed(self.prec, 0).data_ptr(), C.byref(dcfg), C.byref(pcfg), ks.data_ptr(),
            hip.ptr(obs_a), hip.ptr(pairs), hip.ptr(returns), hip.ptr(values), hip.ptr(adv), hip.ptr(logprobs), hip.ptr(kinds), N,
            hip.ptr(global_moments), ga.data_ptr(), gc.data_ptr(), self._stats.data_ptr(), ws.data_ptr(), ws.numel(),
            hip.stream(), C.byref(io)), "dppo_ppo_loss_fwd_bwd_obs")
        a.vis.backward(d_a)
        c.vis.backward(d_c)
        return self._stats

    def _eta_mean(self) -> float:
        return self._eta_value(False) if self.use_ddim else 1.0

    # ------------------------------------------------------------------ behaviour-cloning term (reference :104-126)
    def bc_loss_and_grad(self, cond, noise=None):
        """-mean clamp(log p_ft(base policy's chains), -5, 2) and its gradient w.r.t. the flat ``actor_ft`` parameters.

        Samples the chains with the base policy (``forward(use_base_policy=True)``, stochastic, as the reference) and
        runs forward + backward of the fine-tuned network over the B*Kft rows in one library call.  Returns
        (device float64[1], flat fp32 gradient); both buffers are reused by the next call.
        """
        state = cond["state"]
        hip.require_gpu(state, "PPODiffusion.bc_loss_and_grad")
        samples = self.forward(cond=cond, deterministic=False, return_chain=True, use_base_policy=True, noise=noise)
        B, dev = state.shape[0], state.device
        Kft, AF = self.ft_denoising_steps, self.horizon_steps * self.action_dim
        net = self.actor_ft
        lib, d = hip.load(), net.net_desc()
        flat = net.flat_params()
        if self._bc_grad is None or self._bc_grad.numel() != flat.numel() or self._bc_grad.device != dev:
            object.__setattr__(self, "_bc_grad", torch.zeros_like(flat))
            object.__setattr__(self, "_bc_value", torch.zeros(1, dtype=torch.float64, device=dev))
        obs = state.reshape(B, -1).contiguous().float()
        ch = samples.chains.reshape(B, Kft + 1, AF).contiguous().float()
        ks = self._logprob_schedule(dev)
        cfg = self.diffusion_cfg()
        wsb = lib.dppo_bc_loss_workspace_bytes(C.byref(d), self.prec, B, Kft)
        if wsb < 0:
            hip.check(int(wsb), "dppo_bc_loss_workspace_bytes")
        ws = self._ws_bc.get(wsb, dev)
        hip.check(lib.dppo_bc_loss_fwd_bwd(
            C.byref(d), self.prec, flat.data_ptr(), net.packed(self.prec, self.denoising_steps).data_ptr(), C.byref(cfg),
            ks.data_ptr(), Kft, obs.data_ptr(), ch.data_ptr(), B, self._bc_grad.data_ptr(), self._bc_value.data_ptr(),
            ws.data_ptr(), ws.numel(), hip.stream()), "dppo_bc_loss_fwd_bwd")
        return self._bc_value, self._bc_grad

    def add_bc_gradient(self, cond, coeff: float, noise=None):
        """actor_ft.flat_grads() += coeff * d bc_loss / d theta (the ``bc_loss * bc_loss_coeff`` term of the agent's loss,
        train_ppo_diffusion_agent.py:351-357); returns the device bc_loss."""
        value, grad = self.bc_loss_and_grad(cond, noise)
        ga = self.actor_ft.flat_grads()
        hip.check(hip.load().dppo_axpy(ga.data_ptr(), grad.data_ptr(), float(coeff), ga.numel(), hip.stream()),
                  "dppo_axpy")
        return value

    # ------------------------------------------------------------------ drop-in loss (reference :57-199)
    def loss(self, obs, chains_prev, chains_next, denoising_inds, returns, oldvalues, advantages, oldlogprobs,
             use_bc_loss=False, reward_horizon=4):
        """Same arguments / 8-tuple as the reference.  pg_loss and v_loss carry grad."""
        state = obs["state"]
        hip.require_gpu(state, "PPODiffusion.loss")
        N = state.shape[0]
        AF = self.horizon_steps * self.action_dim
        f32 = dict(dtype=torch.float32)
        pairs = torch.stack([chains_prev.reshape(N, AF), chains_next.reshape(N, AF)], dim=1).contiguous().to(**f32)
        kinds = denoising_inds.reshape(N).to(torch.long).contiguous()
        adv = advantages.reshape(N).contiguous().to(**f32)
        if getattr(self.actor_ft, "is_vision", False):
            if use_bc_loss:
                raise NotImplementedError("dppo_amd: the BC term with pixel networks is not built")
            stats = self._run_ppo_vision(obs, pairs, returns.reshape(N).contiguous().to(**f32),
                                         oldvalues.reshape(N).contiguous().to(**f32), adv,
                                         oldlogprobs.reshape(N, AF).contiguous().to(**f32), kinds, N, reward_horizon, adv)
        else:
            obs_f = state.reshape(N, -1).contiguous().to(**f32)
            stats = self._run_ppo(obs_f, pairs, returns.reshape(N).contiguous().to(**f32),
                                  oldvalues.reshape(N).contiguous().to(**f32), adv,
                                  oldlogprobs.reshape(N, AF).contiguous().to(**f32), None, kinds, N, reward_horizon, adv)
        a_params = list(self.actor_ft.parameters())
        c_params = list(self.critic.parameters())
        pg_loss, v_loss = _FusedPPOLoss.apply(stats, self.actor_ft.grad_views_all(), self.critic.grad_views_all(),
                                              len(a_params), *a_params, *c_params)
        host = stats.tolist()  # one D2H sync, like the reference's .item() calls
        eta = self._eta_mean()
        entropy_loss = torch.tensor(-eta, device=state.device)
        bc_loss = 0
        if use_bc_loss:  # reference :104-126
            value, grad = self.bc_loss_and_grad(obs)
            views, off = [], 0
            for p in a_params:
                views.append(grad[off:off + p.numel()].view(p.shape).clone())
                off += p.numel()
            bc_loss = _FusedBCLoss.apply(value[0], views, *a_params)
        return (pg_loss, entropy_loss, v_loss, host[hip.STAT_CLIPFRAC], host[hip.STAT_APPROX_KL],
                host[hip.STAT_RATIO], bc_loss, eta)

    # ------------------------------------------------------------------ fused-gather fast path
    def ppo_update(self, obs_k, chains_k, returns_k, values_k, adv_k, logprobs_k, inds, reward_horizon=4,
                   global_moments=None, critic_hook=None):
        """One minibatch straight from the rollout buffer (R rows): gradients land in the flat grad buffers of
        ``actor_ft`` / ``critic``; returns the device stats tensor (float64[8], see hip.STAT_*).  No host sync.

        obs_k (R,To*Do), chains_k (R,Kft+1,Ta*Da), returns_k/values_k/adv_k (R,), logprobs_k (R,Kft,Ta*Da),
        inds (N,) int64 in [0, R*Kft)  -- the reference's minibatch assembly
        (agent/finetune/train_ppo_diffusion_agent.py:316-327) fused into the kernel's loader.
        critic_hook(side_stream_handle): data parallel only (``DataParallel.critic_hook``) -- called from inside the library
        call once the critic's gradient launches are queued, so its all-reduce overlaps the actor's backward.
        """
        hip.require_gpu(obs_k, "PPODiffusion.ppo_update")
        N = inds.numel()
        adv_g = None
        if (self.clip_advantage_lower_quantile, self.clip_advantage_upper_quantile) != (0, 1):
            adv_g = adv_k[torch.div(inds, self.ft_denoising_steps, rounding_mode="floor")]
        return self._run_ppo(obs_k, chains_k, returns_k, values_k, adv_k, logprobs_k, inds, None, N, reward_horizon,
                             adv_g, global_moments, critic_hook)
