"""DPPO fine-tuning from pixel observations.  Mirrors ``dppo/agent/finetune/train_ppo_diffusion_img_agent.py`` (reference
``TrainPPOImgDiffusionAgent``): the rollout keeps {"rgb", "state"} per env step, the whole buffer is augmented once per
iteration (random shifts, :188-200), values / log-probs are precomputed in ``logprob_batch_size`` splits, and the update
accumulates the gradients of ``grad_accumulate`` minibatches per optimiser step (:340-372).

What runs where: both ViT encoders, the denoiser, the critic, the loss and all gradients are the HIP library's
(``dppo_vis_*`` + the ``*_obs`` loss entries); each network has two flat parameter buffers (encoder, trunk) and the four
gradient accumulators of a step are slices of ONE bucket, which is also the data-parallel all-reduce payload.
"""
from __future__ import annotations

import logging
import math
import os
import pickle
import time

import numpy as np
import torch
import torch.distributed as dist

from dppo_amd import hip
from dppo_amd.agent.finetune.train_ppo_diffusion_agent import TrainPPODiffusionAgent
from dppo_amd.model.common.modules import RandomShiftsAug
from dppo_amd.util.optim import FlatAdamW, step_many
from dppo_amd.util.rollout import gae_device

log = logging.getLogger(__name__)


class TrainPPOImgDiffusionAgent(TrainPPODiffusionAgent):
    def __init__(self, cfg, venv=None):
        super().__init__(cfg, venv=venv)
        model = self.model
        if not getattr(model.actor_ft, "is_vision", False) or not getattr(model.critic, "is_vision", False):
            raise TypeError("TrainPPOImgDiffusionAgent needs pixel networks (VisionDiffusionMLP / VisionUnet1D + ViTCritic)")
        self.augment = cfg.train.get("augment", False)
        if self.augment:
            self.aug = RandomShiftsAug(pad=4)
        shape_meta = cfg.shape_meta
        self.obs_dims = {k: list(shape_meta.obs[k]["shape"]) for k in shape_meta.obs}
        self.grad_accumulate = cfg.train.get("grad_accumulate", 1)
        if self.use_bc_loss:
            raise NotImplementedError("dppo_amd: the BC term with pixel networks is not built")
        a, c = model.actor_ft, model.critic
        if self.world > 1:  # the parent broadcast the trunks; the encoders too
            for net in (a, c, model.actor):
                dist.broadcast(net.vis.flat_params(), src=0)
                net.vis.mark_updated()
        # the encoders step with their network's hyper-parameters and learning-rate schedule (one param group in the reference)
        self.actor_vis_optimizer = FlatAdamW(a.vis.flat_params(), lr=cfg.train.actor_lr, weight_decay=cfg.train.actor_weight_decay)
        self.critic_vis_optimizer = FlatAdamW(c.vis.flat_params(), lr=cfg.train.critic_lr, weight_decay=cfg.train.critic_weight_decay)
        self.actor_vis_optimizer.param_groups = self.actor_optimizer.param_groups
        self.critic_vis_optimizer.param_groups = self.critic_optimizer.param_groups
        # accumulators: [actor trunk | actor encoder | critic trunk | critic encoder] in one bucket
        sizes = [a.flat_params().numel(), a.vis.flat_params().numel(), c.flat_params().numel(), c.vis.flat_params().numel()]
        self._bucket = torch.zeros(sum(sizes), dtype=torch.float32, device=self.device)
        offs = np.cumsum([0] + sizes)
        self._acc = [self._bucket[offs[i]:offs[i + 1]] for i in range(4)]
        self._norm = torch.zeros(1, dtype=torch.float64, device=self.device)

    # ------------------------------------------------------------------------------------------------- rollout
    def _cond(self, obs):
        return {k: torch.from_numpy(obs[k]).to(self.device, non_blocking=True) for k in self.obs_dims}

    def run(self):
        model, dev = self.model, self.device
        Kft = getattr(model, "ft_denoising_steps", 0)  # 0: a one-shot (Gaussian) policy, the "chain" is the action itself
        Ta, Da = self.horizon_steps, self.action_dim
        AF = Ta * Da
        S, E = self.n_steps, self.n_envs
        R = S * E
        t_start = time.time()
        run_results, cnt_train_step, last_itr_eval = [], 0, False
        done_venv = np.zeros(E, dtype=bool)
        prev_obs = None
        bufs = None
        metrics = {}
        while self.itr < self.n_train_itr:
            eval_mode = self.itr % self.val_freq == 0 and not self.force_train
            model.eval() if eval_mode else model.train()
            firsts = np.zeros((S + 1, E))
            if self.reset_at_iteration or eval_mode or last_itr_eval or prev_obs is None:
                prev_obs = self.reset_env_all()
                firsts[0] = 1
            else:
                firsts[0] = done_venv
            last_itr_eval = eval_mode
            if bufs is None:  # device-resident rollout buffer, images in the dtype the env hands over (uint8 stays uint8)
                bufs = {k: torch.empty((R,) + tuple(prev_obs[k].shape[1:]), device=dev,
                                       dtype=torch.uint8 if prev_obs[k].dtype == np.uint8 else torch.float32)
                        for k in self.obs_dims}
                chains_buf = torch.empty(R, Kft + 1, AF, device=dev)
            reward_trajs, terminated_trajs = np.zeros((S, E)), np.zeros((S, E))
            for step in range(S):
                cond = self._cond(prev_obs)
                for k in bufs:
                    bufs[k][step * E:(step + 1) * E] = cond[k]
                traj, chain = self._sample(cond, eval_mode)
                chains_buf[step * E:(step + 1) * E] = chain.reshape(E, Kft + 1, AF)
                action = traj.cpu().numpy()[:, :self.act_steps]
                prev_obs, reward, terminated, truncated, _ = self.venv.step(action)
                if isinstance(prev_obs, list):
                    prev_obs = {k: np.stack([o[k] for o in prev_obs]) for k in prev_obs[0]}
                done_venv = terminated | truncated
                reward_trajs[step], terminated_trajs[step], firsts[step + 1] = reward, terminated, done_venv
            cnt_train_step += S * E * self.act_steps * self.world if not eval_mode else 0
            ep_rewards, ep_best = [], []
            for e in range(E):
                starts = np.where(firsts[:, e] == 1)[0]
                for i in range(len(starts) - 1):
                    a, b = starts[i], starts[i + 1]
                    if b - a > 1:
                        seg = reward_trajs[a:b, e]
                        ep_rewards.append(seg.sum())
                        ep_best.append(seg.max() / self.act_steps)
            n_ep = len(ep_rewards)
            avg_ep = float(np.mean(ep_rewards)) if n_ep else 0.0
            avg_best = float(np.mean(ep_best)) if n_ep else 0.0
            success = float(np.mean(np.array(ep_best) >= self.best_reward_threshold_for_success)) if n_ep else 0.0
            if not eval_mode:
                metrics = self._update_img(bufs, chains_buf, reward_trajs, terminated_trajs, firsts, prev_obs, R, Kft)
            if self.itr >= self.n_critic_warmup_itr:
                self.actor_lr_scheduler.step()
            self.critic_lr_scheduler.step()
            if hasattr(model, "step"):
                model.step()
            if self.itr % self.save_model_freq == 0 or self.itr == self.n_train_itr - 1:
                self.save_model()
            rec = {"itr": self.itr, "step": cnt_train_step}
            if self.itr % self.log_freq == 0 and self.rank == 0:
                rec["time"] = time.time() - t_start
                if eval_mode:
                    rec.update(eval_success_rate=success, eval_episode_reward=avg_ep, eval_best_reward=avg_best)
                    log.info("eval: success rate %8.4f | avg episode reward %8.4f | avg best reward %8.4f", success, avg_ep,
                             avg_best)
                else:
                    rec.update(train_episode_reward=avg_ep, **metrics)
                    log.info("%d: step %8d | loss %8.4f | pg loss %8.4f | value loss %8.4f | reward %8.4f | t:%8.4f", self.itr,
                             cnt_train_step, metrics.get("loss", float("nan")), metrics.get("pg_loss", float("nan")),
                             metrics.get("v_loss", float("nan")), avg_ep, rec["time"])
                run_results.append(rec)
                with open(self.result_path, "wb") as f:
                    pickle.dump(run_results, f)
            self.itr += 1
        return run_results

    def _sample(self, cond, eval_mode):
        """(trajectories (E,Ta,Da), chains (E,Kft+1,Ta,Da)) of one env step."""
        smp = self.model(cond=cond, deterministic=eval_mode, return_chain=True)
        return smp.trajectories, smp.chains

    # ------------------------------------------------------------------------------------------------- update
    def _augment_buffer(self, bufs, R):
        """One random shift per stored image, before anything reads the buffer (reference :188-200)."""
        rgb = bufs["rgb"]
        dt = rgb.dtype
        for lo in range(0, R, 4096):
            x = rgb[lo:lo + 4096]
            y = self.aug(x.reshape(-1, *x.shape[2:]))  # "(s e t) c h w": every frame of the history shifts on its own
            rgb[lo:lo + 4096] = y.reshape(x.shape).to(dt)

    def _grad_bufs(self):
        a, c = self.model.actor_ft, self.model.critic
        return [a.flat_grads(), a.vis.flat_grads(), c.flat_grads(), c.vis.flat_grads()]

    def _accumulate(self, first: bool):
        lib = hip.load()
        for acc, g, scale in zip(self._acc, self._grad_bufs(), (1.0, 1.0, self.vf_coef, self.vf_coef)):
            if first:
                acc.copy_(g)
                if scale != 1.0:
                    acc.mul_(scale)
            else:
                hip.check(lib.dppo_axpy(acc.data_ptr(), g.data_ptr(), float(scale), acc.numel(), hip.stream()), "dppo_axpy")

    def _optimizer_step(self, update_actor: bool):
        """One AdamW launch over the four accumulators; the actor's clip norm spans its trunk and its encoder
        (clip_grad_norm_ over actor_ft.parameters(), reference :347-351)."""
        model = self.model
        if self.world > 1:
            dist.all_reduce(self._bucket)
        slots = [self.critic_optimizer.slot(self._acc[2]), self.critic_vis_optimizer.slot(self._acc[3])]
        if update_actor:
            norm = None
            if self.max_grad_norm is not None:
                n1 = self.actor_optimizer.sq_norm(self._acc[0]).clone()
                n2 = self.actor_vis_optimizer.sq_norm(self._acc[1])
                self._norm.copy_(n1 + n2)
                norm = self._norm
            slots += [self.actor_optimizer.slot(self._acc[0], max_norm=self.max_grad_norm, sq_norm=norm),
                      self.actor_vis_optimizer.slot(self._acc[1], max_norm=self.max_grad_norm, sq_norm=norm)]
        step_many(slots)
        model.critic.mark_updated()
        if update_actor:
            model.actor_ft.mark_updated()

    def _update_img(self, bufs, chains_buf, reward_trajs, terminated_trajs, firsts, last_obs, R, Kft):
        model, dev = self.model, self.device
        S, E = self.n_steps, self.n_envs
        Ta, Da = self.horizon_steps, self.action_dim
        AF = Ta * Da
        if self.augment:
            self._augment_buffer(bufs, R)
        values_buf = torch.empty(R, device=dev)
        logp_buf = torch.empty(R, Kft, AF, device=dev)
        for lo in range(0, R, self.logprob_batch_size):
            hi = min(R, lo + self.logprob_batch_size)
            cond = {k: bufs[k][lo:hi] for k in bufs}
            values_buf[lo:hi] = model.critic(cond, no_augment=True).reshape(-1)
            logp_buf[lo:hi] = model.get_logprobs(cond, chains_buf[lo:hi].reshape(hi - lo, Kft + 1, Ta, Da)).reshape(hi - lo, Kft, AF)
        if self.reward_scale_running:
            reward_trajs = self.running_reward_scaler(reward=reward_trajs.T, first=firsts[:-1].T).T
        last_v = model.critic(self._cond(last_obs), no_augment=True).reshape(-1)
        _, _, adv, ret = gae_device(torch.from_numpy(np.ascontiguousarray(reward_trajs)).to(dev), values_buf.reshape(S, E),
                                    torch.from_numpy(terminated_trajs).float().to(dev), last_v, self.gamma, self.gae_lambda,
                                    self.reward_scale_const)
        adv_k, ret_k = adv.reshape(-1).contiguous(), ret.reshape(-1).contiguous()
        total = R * Kft
        num_batch = max(1, total // self.batch_size)
        clipfracs, stats, flag_break = [], None, False
        update_actor = self.itr >= self.n_critic_warmup_itr
        quant = (model.clip_advantage_lower_quantile, model.clip_advantage_upper_quantile) != (0, 1)
        for _ in range(self.update_epochs):
            perm = torch.randperm(total, device=dev)
            mbs = [perm[b * self.batch_size:(b + 1) * self.batch_size].contiguous() for b in range(num_batch)]
            moments = self.dp.minibatch_moments(adv_k, mbs, Kft)
            pending = 0
            for b, inds in enumerate(mbs):
                rows = torch.div(inds, Kft, rounding_mode="floor")
                kinds = inds - rows * Kft
                cond = {k: bufs[k][rows] for k in bufs}
                pairs = torch.stack([chains_buf[rows, kinds], chains_buf[rows, kinds + 1]], dim=1).contiguous()
                adv_b = adv_k[rows].contiguous()
                st = model._run_ppo_vision(cond, pairs, ret_k[rows].contiguous(), values_buf[rows].contiguous(), adv_b,
                                           logp_buf[rows, kinds].contiguous(), kinds.contiguous(), inds.numel(), self.reward_horizon,
                                           adv_b if quant else None, None if moments is None else moments[b])
                self._accumulate(first=pending == 0)
                pending += 1
                if (b + 1) % self.grad_accumulate == 0:
                    self._optimizer_step(update_actor)
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
        if stats is None:  # fewer minibatches than grad_accumulate: the reference never steps in that case either
            stats = st.tolist()
            clipfracs.append(stats[hip.STAT_CLIPFRAC])
        y_pred, y_true = values_buf.cpu().numpy(), ret_k.cpu().numpy()
        var_y = np.var(y_true)
        eta = model._eta_mean()
        pg, vl = stats[hip.STAT_PG_LOSS], stats[hip.STAT_V_LOSS]
        return {"loss": pg - eta * self.ent_coef + vl * self.vf_coef, "pg_loss": pg, "v_loss": vl,
                "approx_kl": stats[hip.STAT_APPROX_KL], "ratio": stats[hip.STAT_RATIO], "clipfrac": float(np.mean(clipfracs)),
                "eta": eta, "explained_variance": float("nan") if var_y == 0 else float(1 - np.var(y_true - y_pred) / var_y),
                "actor_lr": self.actor_optimizer.param_groups[0]["lr"], "critic_lr": self.critic_optimizer.param_groups[0]["lr"]}
