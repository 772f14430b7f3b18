"""PPO fine-tuning of a Gaussian policy -- the comparison arm of the DPPO paper.

Mirrors the reference's ``TrainPPOGaussianAgent`` (agent/finetune/train_ppo_gaussian_agent.py:20-400): same cfg keys and
iteration structure as the diffusion agent (whose rollout, reward scaling, GAE, schedules, checkpointing and data-parallel
plumbing it inherits); what differs is the sample: one action chunk per step instead of a denoising chain, a scalar
log-prob per sample, minibatches over R = n_steps * n_envs rows.  The update is ``PPO_Gaussian.ppo_update`` (one library
call: forward, loss, backward) + fused AdamW; a learned per-dimension std (``logvar``) is stepped by its own tiny AdamW
with the actor's learning rate, like the reference's ``actor_optimizer`` over ``actor_ft.parameters()``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from dppo_amd import hip
from dppo_amd.agent.finetune.train_ppo_diffusion_agent import TrainPPODiffusionAgent
from dppo_amd.model.diffusion.diffusion import Sample
from dppo_amd.util.optim import FlatAdamW, step_and_repack
from dppo_amd.util.rollout import gae_device


class _OneShotPolicy:
    """Adapter for collect_rollout: the sampled action chunk is both the trajectory and a one-entry 'chain'."""

    def __init__(self, model):
        self.model, self.horizon_steps = model, model.horizon_steps

    def __call__(self, cond, deterministic=False, return_chain=True):
        a = self.model(cond=cond, deterministic=deterministic)
        return Sample(a, a.reshape(a.shape[0], 1, a.shape[1], a.shape[2]))


class TrainPPOGaussianAgent(TrainPPODiffusionAgent):
    def __init__(self, cfg, venv=None):
        super().__init__(cfg, venv)
        net = self.model.actor_ft
        self.logvar_optimizer = None
        if getattr(net, "learn_fixed_std", False):
            self.logvar_optimizer = FlatAdamW(net.logvar.data, lr=cfg.train.actor_lr, weight_decay=cfg.train.actor_weight_decay)
        if getattr(self.model, "entropy_in_kernel", False):  # PPO_GMM: the entropy term's gradient rides the fused backward
            self.model.ent_coef = float(self.ent_coef)

    def _policy(self):
        return _OneShotPolicy(self.model)

    def _update(self, obs_buf, chains_buf, values_buf, logp_buf, reward_trajs, terminated_trajs, firsts, last_obs, R, Kft):
        model, dev = self.model, self.device
        S, E = self.n_steps, self.n_envs
        To, Do, Ta, Da = self.n_cond_step, self.obs_dim, self.horizon_steps, self.action_dim
        samples = chains_buf.reshape(R, Ta * Da)
        logp = torch.empty(R, device=dev)
        for lo in range(0, R, self.logprob_batch_size):  # values and old log-probs over the buffer (:183-205)
            hi = min(R, lo + self.logprob_batch_size)
            st = {"state": obs_buf[lo:hi].reshape(hi - lo, To, Do)}
            values_buf[lo:hi] = model.critic(st).reshape(-1)
            logp[lo:hi] = model.get_logprobs(st, samples[lo:hi])[0]
        if self.reward_scale_running:
            reward_trajs = self.running_reward_scaler(reward=reward_trajs.T, first=firsts[:-1].T).T
        last_v = model.critic({"state": torch.from_numpy(last_obs["state"]).float().to(dev)}).reshape(-1)
        _, _, adv, ret = gae_device(torch.from_numpy(np.ascontiguousarray(reward_trajs)).to(dev), values_buf.reshape(S, E),
                                    torch.from_numpy(terminated_trajs).float().to(dev), last_v, self.gamma, self.gae_lambda,
                                    self.reward_scale_const)
        adv_k, ret_k = adv.reshape(-1).contiguous(), ret.reshape(-1).contiguous()
        num_batch = max(1, R // self.batch_size)
        clipfracs, stats, flag_break = [], None, False
        update_actor = self.itr >= self.n_critic_warmup_itr
        net = model.actor_ft
        for _ in range(self.update_epochs):
            perm = torch.randperm(R, device=dev)
            mbs = [perm[b * self.batch_size:(b + 1) * self.batch_size].contiguous() for b in range(num_batch)]
            moments = self.dp.minibatch_moments(adv_k, mbs, 1)
            for b, inds in enumerate(mbs):
                st = model.ppo_update(obs_buf[inds], samples[inds], ret_k[inds], values_buf[inds], adv_k[inds], logp[inds],
                                      global_moments=None if moments is None else moments[b])
                if self.vf_coef != 1:
                    model.critic.flat_grads().mul_(self.vf_coef)
                self.dp.allreduce_grads()
                g_lv = sq = None
                if self.logvar_optimizer is not None and update_actor:
                    # loss = pg + entropy_loss * ent_coef + ...: d(-entropy)/d logvar_j = -0.5 / Da inside the clamp range
                    lv = net.logvar.detach()
                    g_lv = model._lv_grad.clone()
                    if self.world > 1:
                        dist.all_reduce(g_lv)  # per-rank pg parts are already divided by the GLOBAL count: SUM gives the whole
                    if not getattr(model, "entropy_in_kernel", False):
                        g_lv -= self.ent_coef * 0.5 / Da * ((lv >= net.logvar_min) & (lv <= net.logvar_max)).float()
                    g_lv = g_lv.contiguous()
                    if self.max_grad_norm is not None:
                        # the reference clips actor_ft.parameters() as ONE group (train_ppo_gaussian_agent.py:319-322), and
                        # logvar is one of them: its gradient counts in the norm and is scaled with the trunk's
                        sq = self.actor_optimizer.sq_norm(net.flat_grads()).clone() + (g_lv.double() ** 2).sum()
                step_and_repack(model, self.actor_optimizer, self.critic_optimizer, update_actor=update_actor,
                                max_norm=self.max_grad_norm, n_time=0, actor_sq_norm=sq)
                if g_lv is not None:
                    self.logvar_optimizer.param_groups[0]["lr"] = self.actor_optimizer.param_groups[0]["lr"]
                    self.logvar_optimizer.step(g_lv, max_norm=self.max_grad_norm, sq_norm=sq)
                stats = st.tolist()
                clipfracs.append(stats[hip.STAT_CLIPFRAC])
                if self.target_kl is not None and stats[hip.STAT_APPROX_KL] > self.target_kl:
                    flag_break = True
                    break
            if flag_break:
                break
        y_pred, y_true = values_buf.cpu().numpy(), ret_k.cpu().numpy()
        var_y = np.var(y_true)
        pg, vl, ent = stats[hip.STAT_PG_LOSS], stats[hip.STAT_V_LOSS], stats[hip.GAUSS_STAT_ENTROPY]
        return {"loss": pg - ent * self.ent_coef + vl * self.vf_coef, "pg_loss": pg, "v_loss": vl,
                "approx_kl": stats[hip.STAT_APPROX_KL], "ratio": stats[hip.STAT_RATIO],
                "clipfrac": float(np.mean(clipfracs)), "std": stats[hip.GAUSS_STAT_STD], "entropy": ent,
                "explained_variance": float("nan") if var_y == 0 else float(1 - np.var(y_true - y_pred) / var_y),
                "actor_lr": self.actor_optimizer.param_groups[0]["lr"],
                "critic_lr": self.critic_optimizer.param_groups[0]["lr"]}
