"""PPO fine-tuning of a Gaussian policy from pixel observations.  Mirrors the reference's ``TrainPPOImgGaussianAgent``
(agent/finetune/train_ppo_gaussian_img_agent.py): the image agent's rollout / augmentation / gradient accumulation with the
Gaussian arm's sample (one action chunk per step, a scalar log-prob, minibatches over R = n_steps * n_envs rows) and a learned
per-dimension std stepped with the actor's learning rate."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from dppo_amd import hip
from dppo_amd.agent.finetune.train_ppo_diffusion_img_agent import TrainPPOImgDiffusionAgent
from dppo_amd.util.optim import FlatAdamW
from dppo_amd.util.rollout import gae_device


class TrainPPOImgGaussianAgent(TrainPPOImgDiffusionAgent):
    def __init__(self, cfg, venv=None):
        super().__init__(cfg, venv)
        net = self.model.actor_ft
        self.logvar_optimizer = None
        if getattr(net, "learn_fixed_std", False):
            self.logvar_optimizer = FlatAdamW(net.logvar.data, lr=cfg.train.actor_lr, weight_decay=cfg.train.actor_weight_decay)
            self._lv_acc = torch.zeros_like(net.logvar.data)

    def _sample(self, cond, eval_mode):
        a = self.model(cond=cond, deterministic=eval_mode)
        return a, a.reshape(a.shape[0], 1, a.shape[1], a.shape[2])

    def _update_img(self, bufs, chains_buf, reward_trajs, terminated_trajs, firsts, last_obs, R, Kft):
        model, dev = self.model, self.device
        S, E = self.n_steps, self.n_envs
        Ta, Da = self.horizon_steps, self.action_dim
        if self.augment:
            self._augment_buffer(bufs, R)
        samples = chains_buf.reshape(R, Ta * Da)
        values_buf, logp = torch.empty(R, device=dev), torch.empty(R, device=dev)
        for lo in range(0, R, self.logprob_batch_size):
            hi = min(R, lo + self.logprob_batch_size)
            cond = {k: bufs[k][lo:hi] for k in bufs}
            values_buf[lo:hi] = model.critic(cond, no_augment=True).reshape(-1)
            logp[lo:hi] = model.get_logprobs(cond, samples[lo:hi])[0]
        if self.reward_scale_running:
            reward_trajs = self.running_reward_scaler(reward=reward_trajs.T, first=firsts[:-1].T).T
        last_v = model.critic(self._cond(last_obs), no_augment=True).reshape(-1)
        _, _, adv, ret = gae_device(torch.from_numpy(np.ascontiguousarray(reward_trajs)).to(dev), values_buf.reshape(S, E),
                                    torch.from_numpy(terminated_trajs).float().to(dev), last_v, self.gamma, self.gae_lambda,
                                    self.reward_scale_const)
        adv_k, ret_k = adv.reshape(-1).contiguous(), ret.reshape(-1).contiguous()
        num_batch = max(1, R // self.batch_size)
        clipfracs, stats, flag_break, st = [], None, False, None
        update_actor = self.itr >= self.n_critic_warmup_itr
        net = model.actor_ft
        for _ in range(self.update_epochs):
            perm = torch.randperm(R, device=dev)
            mbs = [perm[b * self.batch_size:(b + 1) * self.batch_size].contiguous() for b in range(num_batch)]
            moments = self.dp.minibatch_moments(adv_k, mbs, 1)
            pending = 0
            for b, inds in enumerate(mbs):
                cond = {k: bufs[k][inds] for k in bufs}
                st = model.ppo_update(cond, samples[inds].contiguous(), ret_k[inds].contiguous(), values_buf[inds].contiguous(),
                                      adv_k[inds].contiguous(), logp[inds].contiguous(),
                                      global_moments=None if moments is None else moments[b])
                self._accumulate(first=pending == 0)
                if self.logvar_optimizer is not None:
                    self._lv_acc.copy_(model._lv_grad) if pending == 0 else self._lv_acc.add_(model._lv_grad)
                pending += 1
                if (b + 1) % self.grad_accumulate == 0:
                    self._optimizer_step(update_actor)
                    if self.logvar_optimizer is not None and update_actor:
                        lv = net.logvar.detach()
                        g = self._lv_acc.clone()
                        if self.world > 1:
                            dist.all_reduce(g)
                        g -= pending * self.ent_coef * 0.5 / Da * ((lv >= net.logvar_min) & (lv <= net.logvar_max)).float()
                        self.logvar_optimizer.param_groups[0]["lr"] = self.actor_optimizer.param_groups[0]["lr"]
                        self.logvar_optimizer.step(g.contiguous())
                    pending = 0
                    if self.world > 1:
                        dist.all_reduce(st)
                        st[5:7] /= self.world
                    stats = st.tolist()
                    clipfracs.append(stats[hip.STAT_CLIPFRAC])
                    if self.target_kl is not None and stats[hip.STAT_APPROX_KL] > self.target_kl and update_actor:
                        flag_break = True
                        break
            if flag_break:
                break
        if stats is None:
            stats = st.tolist()
            clipfracs.append(stats[hip.STAT_CLIPFRAC])
        y_pred, y_true = values_buf.cpu().numpy(), ret_k.cpu().numpy()
        var_y = np.var(y_true)
        pg, vl, ent = stats[hip.STAT_PG_LOSS], stats[hip.STAT_V_LOSS], stats[hip.GAUSS_STAT_ENTROPY]
        return {"loss": pg - ent * self.ent_coef + vl * self.vf_coef, "pg_loss": pg, "v_loss": vl,
                "approx_kl": stats[hip.STAT_APPROX_KL], "ratio": stats[hip.STAT_RATIO], "clipfrac": float(np.mean(clipfracs)),
                "std": stats[hip.GAUSS_STAT_STD], "entropy": ent,
                "explained_variance": float("nan") if var_y == 0 else float(1 - np.var(y_true - y_pred) / var_y),
                "actor_lr": self.actor_optimizer.param_groups[0]["lr"], "critic_lr": self.critic_optimizer.param_groups[0]["lr"]}
