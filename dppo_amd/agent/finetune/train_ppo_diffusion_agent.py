"""DPPO fine-tuning agent (entry point named by the north star).

Mirrors the reference's ``TrainPPODiffusionAgent`` / ``TrainPPOAgent`` / ``TrainAgent``
(agent/finetune/train_ppo_diffusion_agent.py:21-483, train_ppo_agent.py:16-89, train_agent.py:19-145): same cfg
keys, same iteration structure (rollout -> value / log-prob precompute -> reward scaling -> GAE -> epochs of
minibatch PPO with KL early stop -> LR schedules -> checkpoint), same checkpoint format.  What changed is WHERE
things live: the rollout buffer is fp32 and device-resident (the reference keeps float64 numpy holders and copies
them back and forth, :78-93,198-303), sampling / log-probs / GAE / gather + loss + backward / AdamW are HIP kernels,
and with WORLD_SIZE > 1 every rank owns a shard of the envs and gradients are all-reduced once per step
(dppo_amd.parallel).  Envs stay on host CPU.
"""
from __future__ import annotations

import logging
import math
import os
import pickle
import random
import time

import numpy as np
import torch
import torch.distributed as dist

from dppo_amd.cfg.loader import instantiate
from dppo_amd.env.synthetic import make_venv
from dppo_amd.parallel import DataParallel
from dppo_amd.util.optim import FlatAdamW, step_and_repack
from dppo_amd.util.reward_scaling import RunningRewardScaler
from dppo_amd.util.rollout import collect_rollout, gae_device
from dppo_amd.util.scheduler import CosineAnnealingWarmupRestarts
from dppo_amd import hip

log = logging.getLogger(__name__)


class TrainPPODiffusionAgent:
    def __init__(self, cfg, venv=None):
        self.cfg = cfg
        self.device = cfg.device
        self.seed = cfg.get("seed", 42)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        random.seed(self.seed)
        np.random.seed(self.seed)
        torch.manual_seed(self.seed)  # identical on every rank -> identical initial weights

        # ---- TrainAgent (train_agent.py:21-120)
        self.use_wandb = cfg.get("wandb", None) is not None and self.rank == 0
        if self.use_wandb:
            try:
                import wandb
                wandb.init(entity=cfg.wandb.entity, project=cfg.wandb.project, name=cfg.wandb.run, config=dict(cfg))
                self._wandb = wandb
            except ImportError:
                log.warning("wandb is not installed; logging to the python logger and result.pkl only")
                self.use_wandb = False
        self.n_envs = cfg.env.n_envs  # per rank: env shards are independent
        self.venv = venv if venv is not None else make_venv(cfg)
        if hasattr(self.venv, "seed") and cfg.env.get("env_type", None) != "furniture":
            self.venv.seed([self.seed + self.rank * self.n_envs + i for i in range(self.n_envs)])
        self.n_cond_step, self.obs_dim, self.action_dim = cfg.cond_steps, cfg.obs_dim, cfg.action_dim
        self.act_steps, self.horizon_steps = cfg.act_steps, cfg.horizon_steps
        self.max_episode_steps = cfg.env.get("max_episode_steps", 1000)
        self.reset_at_iteration = cfg.env.get("reset_at_iteration", True)
        self.batch_size = cfg.train.batch_size
        self.model = instantiate(cfg.model)
        self.itr = 0
        self.n_train_itr, self.val_freq = cfg.train.n_train_itr, cfg.train.val_freq
        self.force_train = cfg.train.get("force_train", False)
        self.n_steps = cfg.train.n_steps
        self.best_reward_threshold_for_success = cfg.env.get("best_reward_threshold_for_success", 0)
        self.max_grad_norm = cfg.train.get("max_grad_norm", None)
        self.logdir = cfg.logdir
        self.checkpoint_dir = os.path.join(self.logdir, "checkpoint")
        self.result_path = os.path.join(self.logdir, "result.pkl")
        if self.rank == 0:
            os.makedirs(self.checkpoint_dir, exist_ok=True)
        self.log_freq = cfg.train.get("log_freq", 1)
        self.save_model_freq = cfg.train.save_model_freq

        # ---- TrainPPOAgent (train_ppo_agent.py:18-89)
        self.logprob_batch_size = cfg.train.get("logprob_batch_size", 10000)
        assert self.logprob_batch_size % self.n_envs == 0, "logprob_batch_size must be divisible by n_envs"
        self.gamma = cfg.train.gamma
        self.n_critic_warmup_itr = cfg.train.n_critic_warmup_itr
        self.dp = DataParallel(self.model, self.world)
        if self.world > 1:
            # the broadcast above made the weights identical; from here on every rank needs its OWN random stream
            # (sampler Philox key, minibatch permutation, BC noise): same seed everywhere would give env row i of
            # every shard the same exploration noise
            self.reseed(self.seed + self.rank)
        self.actor_optimizer = FlatAdamW(self.model.actor_ft.flat_params(), lr=cfg.train.actor_lr,
                                         weight_decay=cfg.train.actor_weight_decay)
        self.critic_optimizer = FlatAdamW(self.model.critic.flat_params(), lr=cfg.train.critic_lr,
                                          weight_decay=cfg.train.critic_weight_decay)
        sa, sc = cfg.train.actor_lr_scheduler, cfg.train.critic_lr_scheduler
        self.actor_lr_scheduler = CosineAnnealingWarmupRestarts(
            self.actor_optimizer, first_cycle_steps=sa.first_cycle_steps, cycle_mult=1.0, max_lr=cfg.train.actor_lr,
            min_lr=sa.min_lr, warmup_steps=sa.warmup_steps, gamma=1.0)
        self.critic_lr_scheduler = CosineAnnealingWarmupRestarts(
            self.critic_optimizer, first_cycle_steps=sc.first_cycle_steps, cycle_mult=1.0, max_lr=cfg.train.critic_lr,
            min_lr=sc.min_lr, warmup_steps=sc.warmup_steps, gamma=1.0)
        self.gae_lambda = cfg.train.get("gae_lambda", 0.95)
        self.target_kl = cfg.train.target_kl
        self.update_epochs = cfg.train.update_epochs
        self.ent_coef = cfg.train.get("ent_coef", 0)  # entropy of a fixed-variance chain is constant: no gradient
        self.vf_coef = cfg.train.get("vf_coef", 0)
        self.reward_scale_running = cfg.train.reward_scale_running
        if self.reward_scale_running:
            self.running_reward_scaler = RunningRewardScaler(self.n_envs, moments_hook=self._pool_return_moments)
        self.reward_scale_const = cfg.train.get("reward_scale_const", 1)
        self.use_bc_loss = cfg.train.get("use_bc_loss", False)
        self.bc_loss_coeff = cfg.train.get("bc_loss_coeff", 0)
        # ---- TrainPPODiffusionAgent (:22-45)
        self.reward_horizon = cfg.get("reward_horizon", self.act_steps)
        self.learn_eta = getattr(self.model, "learn_eta", False)

    @staticmethod
    def reseed(seed: int):
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)  # CPU generator (the sampler's Philox key is drawn from it) and the device generators

    # -------------------------------------------------------------------------------------------------
    def _pool_return_moments(self, mean, var, cnt):
        """Pool (mean, var, count) of the discounted returns over ranks (reward_scaling.py:52,65 pooled variance)."""
        if self.world == 1:
            return mean, var, cnt
        t = torch.tensor([cnt, cnt * mean, cnt * (var + mean * mean)], dtype=torch.float64, device=self.device)
        dist.all_reduce(t)
        n, s1, s2 = t.tolist()
        m = s1 / n
        return m, s2 / n - m * m, n

    def reset_env_all(self, options_venv=None):
        obs = self.venv.reset_arg(options_list=options_venv or [{} for _ in range(self.n_envs)])
        if isinstance(obs, list):
            obs = {k: np.stack([o[k] for o in obs]) for k in obs[0]}
        return obs

    def save_model(self):
        """checkpoint/state_{itr}.pt = {"itr", "model": state_dict} (train_agent.py:125-135)."""
        if self.rank != 0:
            return
        path = os.path.join(self.checkpoint_dir, f"state_{self.itr}.pt")
        torch.save({"itr": self.itr, "model": self.model.state_dict()}, path)
        log.info("Saved model to %s", path)

    def load(self, itr):
        data = torch.load(os.path.join(self.checkpoint_dir, f"state_{itr}.pt"), weights_only=True)
        self.itr = data["itr"]
        self.model.load_state_dict(data["model"])
        for net in (self.model.actor, self.model.actor_ft, self.model.critic):
            net.mark_updated()  # kernel images are rebuilt from the loaded weights on next use

    # -------------------------------------------------------------------------------------------------
    def run(self):
        model, dev = self.model, self.device
        Kft = getattr(model, "ft_denoising_steps", 0)  # 0: a one-shot (Gaussian) policy, the "chain" is the action itself
        AF = self.horizon_steps * self.action_dim
        S, E = self.n_steps, self.n_envs
        R = S * E
        t_start = time.time()
        run_results = []
        cnt_train_step = 0
        last_itr_eval = False
        done_venv = np.zeros(E, dtype=bool)
        prev_obs = None
        # device-resident rollout buffer (fp32), reused across iterations
        obs_buf = torch.empty(R, self.n_cond_step * self.obs_dim, device=dev)
        chains_buf = torch.empty(R, Kft + 1, AF, device=dev)
        values_buf = torch.empty(R, device=dev)
        logp_buf = torch.empty(R, Kft, AF, device=dev)
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
            # ---------------- rollout: sample on the GPU, step envs on the host (:101-151); pinned hand-off, env groups
            # (env.pipeline_groups) software-pipelined against the sampler (dppo_amd/util/rollout.py)
            reward_trajs, terminated_trajs, done_trajs, prev_obs = collect_rollout(
                self._policy(), self.venv, prev_obs, S, self.act_steps, obs_buf, chains_buf, deterministic=eval_mode)
            if hasattr(self.model, "check_sampler_health"):
                self.model.check_sampler_health()  # fail loudly if the split sampler's hand-over ever timed out
            firsts[1:] = done_trajs
            done_venv = done_trajs[-1].astype(bool)
            cnt_train_step += S * E * self.act_steps * self.world if not eval_mode else 0
            # ---------------- episode statistics (:153-193)
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
            # ---------------- update (:196-390)
            if not eval_mode:
                metrics = self._update(obs_buf, chains_buf, values_buf, logp_buf, reward_trajs, terminated_trajs,
                                       firsts, prev_obs, R, Kft)
            # ---------------- schedules, annealing, checkpoint, logging (:406-483)
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
                    log.info("eval: success rate %8.4f | avg episode reward %8.4f | avg best reward %8.4f", success,
                             avg_ep, avg_best)
                else:
                    rec.update(train_episode_reward=avg_ep, **metrics)
                    log.info("%d: step %8d | loss %8.4f | pg loss %8.4f | value loss %8.4f | reward %8.4f | t:%8.4f",
                             self.itr, cnt_train_step, metrics.get("loss", float("nan")),
                             metrics.get("pg_loss", float("nan")), metrics.get("v_loss", float("nan")), avg_ep,
                             rec["time"])
                if self.use_wandb:
                    self._wandb.log({k: v for k, v in rec.items() if k != "itr"}, step=self.itr)
                run_results.append(rec)
                with open(self.result_path, "wb") as f:
                    pickle.dump(run_results, f)
            self.itr += 1
        return run_results

    def _policy(self):
        """What collect_rollout calls per env step: (cond, deterministic, return_chain[, out]) -> Sample."""
        return self.model

    # -------------------------------------------------------------------------------------------------
    def _update(self, obs_buf, chains_buf, values_buf, logp_buf, reward_trajs, terminated_trajs, firsts, last_obs, R,
                Kft):
        model, dev = self.model, self.device
        S, E = self.n_steps, self.n_envs
        To, Do, Ta, Da = self.n_cond_step, self.obs_dim, self.horizon_steps, self.action_dim
        # values and old log-probs over the whole buffer, in logprob_batch_size splits (:203-240)
        for lo in range(0, R, self.logprob_batch_size):
            hi = min(R, lo + self.logprob_batch_size)
            st = {"state": obs_buf[lo:hi].reshape(hi - lo, To, Do)}
            values_buf[lo:hi] = model.critic(st).reshape(-1)
            logp_buf[lo:hi] = model.get_logprobs(st, chains_buf[lo:hi].reshape(hi - lo, Kft + 1, Ta, Da)).reshape(
                hi - lo, Kft, Ta * Da)
        # running reward scaling on the host, float64 (:243-247)
        if self.reward_scale_running:
            reward_trajs = self.running_reward_scaler(reward=reward_trajs.T, first=firsts[:-1].T).T
        # GAE on the device in float64 (:250-279)
        last_v = model.critic({"state": torch.from_numpy(last_obs["state"]).float().to(dev)}).reshape(-1)
        _, _, adv, ret = gae_device(torch.from_numpy(np.ascontiguousarray(reward_trajs)).to(dev),
                                    values_buf.reshape(S, E), torch.from_numpy(terminated_trajs).float().to(dev),
                                    last_v, self.gamma, self.gae_lambda, self.reward_scale_const)
        adv_k, ret_k = adv.reshape(-1).contiguous(), ret.reshape(-1).contiguous()
        # minibatch PPO (:306-383)
        total = R * Kft
        num_batch = max(1, total // self.batch_size)
        clipfracs, stats, flag_break = [], None, False
        update_actor = self.itr >= self.n_critic_warmup_itr
        for _ in range(self.update_epochs):
            perm = torch.randperm(total, device=dev)
            mbs = [perm[b * self.batch_size:(b + 1) * self.batch_size].contiguous() for b in range(num_batch)]
            moments = self.dp.minibatch_moments(adv_k, mbs, Kft)  # one tiny collective per epoch (None if 1 rank)
            for b, inds in enumerate(mbs):
                st = model.ppo_update(obs_buf, chains_buf, ret_k, values_buf, adv_k, logp_buf, inds,
                                      reward_horizon=self.reward_horizon,
                                      global_moments=None if moments is None else moments[b],
                                      critic_hook=self.dp.critic_hook)  # (ranks > 1: the critic slice's all-reduce rides the call)
                if self.use_bc_loss:  # + bc_loss * bc_loss_coeff on the minibatch's observations (reference :329-357)
                    rows = torch.div(inds, Kft, rounding_mode="floor")
                    model.add_bc_gradient({"state": obs_buf[rows].reshape(rows.numel(), self.n_cond_step, -1)},
                                          self.bc_loss_coeff / self.world)
                if self.vf_coef != 1:  # loss = pg + ... + v_loss * vf_coef: the critic sees vf_coef * d v_loss
                    model.critic.flat_grads().mul_(self.vf_coef)
                self.dp.allreduce_grads()
                step_and_repack(model, self.actor_optimizer, self.critic_optimizer, update_actor=update_actor,
                                max_norm=self.max_grad_norm)
                stats = st.tolist()  # D2H sync once per minibatch, as the reference's .item() calls
                clipfracs.append(stats[hip.STAT_CLIPFRAC])
                if self.target_kl is not None and stats[hip.STAT_APPROX_KL] > self.target_kl:
                    flag_break = True  # the KL is all-reduced, so every rank takes this branch together
                    break
            if flag_break:
                break
        y_pred, y_true = values_buf.cpu().numpy(), ret_k.cpu().numpy()
        var_y = np.var(y_true)
        eta = model._eta_mean()
        pg, vl = stats[hip.STAT_PG_LOSS], stats[hip.STAT_V_LOSS]
        return {"loss": pg - eta * self.ent_coef + vl * self.vf_coef, "pg_loss": pg, "v_loss": vl,
                "approx_kl": stats[hip.STAT_APPROX_KL], "ratio": stats[hip.STAT_RATIO],
                "clipfrac": float(np.mean(clipfracs)), "eta": eta,
                "explained_variance": float("nan") if var_y == 0 else float(1 - np.var(y_true - y_pred) / var_y),
                "actor_lr": self.actor_optimizer.param_groups[0]["lr"],
                "critic_lr": self.critic_optimizer.param_groups[0]["lr"]}
