"""Evaluation agents.  Mirror ``dppo/agent/eval/eval_agent.py`` (reference ``EvalAgent``) and its four subclasses
(``eval_diffusion_agent.py``, ``eval_diffusion_img_agent.py``, ``eval_gaussian_agent.py``, ``eval_gaussian_img_agent.py``):
build the vectorised env and the model from the cfg, roll ``n_steps`` deterministic action chunks, report the episodes that
finished inside the window (count, success rate, mean episode reward, mean best reward) and write them to ``result.npz``.
One class body serves the four: what differs is the observation dict handed to the model (state, or state + rgb) and whether the
model returns a ``Sample`` (diffusion) or the action chunk (Gaussian).  The sampling itself is the library's K-step sampler /
Gaussian head (and, for pixels, the visual encoder)."""
from __future__ import annotations

import logging
import os
import random
import time

import numpy as np
import torch

from dppo_amd.cfg.loader import instantiate
from dppo_amd.env.synthetic import make_venv

log = logging.getLogger(__name__)


class EvalAgent:
    obs_keys = ("state",)

    def __init__(self, cfg, venv=None):
        self.cfg = cfg
        self.device = cfg.device
        self.seed = cfg.get("seed", 42)
        random.seed(self.seed)
        np.random.seed(self.seed)
        torch.manual_seed(self.seed)
        self.n_envs = cfg.env.n_envs
        self.venv = venv if venv is not None else make_venv(cfg)
        if hasattr(self.venv, "seed") and cfg.env.get("env_type", None) != "furniture":
            self.venv.seed([self.seed + i for i in range(self.n_envs)])
        self.n_cond_step, self.obs_dim, self.action_dim = cfg.cond_steps, cfg.obs_dim, cfg.action_dim
        self.act_steps, self.horizon_steps = cfg.act_steps, cfg.horizon_steps
        self.max_episode_steps = cfg.env.get("max_episode_steps", 1000)
        self.furniture_sparse_reward = bool(cfg.env.specific.get("sparse_reward", False)) if "specific" in cfg.env else False
        self.model = instantiate(cfg.model)
        self.n_steps = cfg.n_steps
        self.best_reward_threshold_for_success = cfg.env.get("best_reward_threshold_for_success", 0)
        self.logdir = cfg.logdir
        os.makedirs(self.logdir, exist_ok=True)
        self.result_path = os.path.join(self.logdir, "result.npz")

    def reset_env_all(self, options_venv=None):
        obs = self.venv.reset_arg(options_list=options_venv or [{} for _ in range(self.n_envs)])
        if isinstance(obs, list):
            obs = {k: np.stack([o[k] for o in obs]) for k in obs[0]}
        return obs

    def _cond(self, obs):
        out = {}
        for k in self.obs_keys:
            t = torch.from_numpy(obs[k])
            out[k] = (t if t.dtype == torch.uint8 else t.float()).to(self.device)
        return out

    def _act(self, cond) -> torch.Tensor:
        """(n_envs, Ta, Da) deterministic action chunks."""
        out = self.model(cond=cond, deterministic=True)
        return out.trajectories if hasattr(out, "trajectories") else out

    @torch.no_grad()
    def run(self):
        t0 = time.time()
        self.model.eval()
        S, E = self.n_steps, self.n_envs
        firsts = np.zeros((S + 1, E))
        prev_obs = self.reset_env_all()
        firsts[0] = 1
        reward_trajs = np.zeros((S, E))
        for step in range(S):
            action = self._act(self._cond(prev_obs)).cpu().numpy()[:, :self.act_steps]
            prev_obs, reward, terminated, truncated, _ = self.venv.step(action)
            if isinstance(prev_obs, list):
                prev_obs = {k: np.stack([o[k] for o in prev_obs]) for k in prev_obs[0]}
            reward_trajs[step] = reward
            firsts[step + 1] = terminated | truncated
        ep_reward, ep_best = [], []
        for e in range(E):  # episodes that start and end inside the window (:82-120)
            starts = np.where(firsts[:, e] == 1)[0]
            for i in range(len(starts) - 1):
                a, b = starts[i], starts[i + 1]
                if b - a > 1:
                    seg = reward_trajs[a:b, e]
                    ep_reward.append(seg.sum())
                    ep_best.append(seg.sum() if self.furniture_sparse_reward else seg.max() / self.act_steps)
        n_ep = len(ep_reward)
        res = {"num_episode": n_ep, "eval_success_rate": float(np.mean(np.array(ep_best) >= self.best_reward_threshold_for_success)) if n_ep else 0.0,
               "eval_episode_reward": float(np.mean(ep_reward)) if n_ep else 0.0,
               "eval_best_reward": float(np.mean(ep_best)) if n_ep else 0.0, "time": time.time() - t0}
        if not n_ep:
            log.info("[WARNING] No episode completed within the iteration!")
        log.info("eval: num episode %4d | success rate %8.4f | avg episode reward %8.4f | avg best reward %8.4f", n_ep,
                 res["eval_success_rate"], res["eval_episode_reward"], res["eval_best_reward"])
        np.savez(self.result_path, **res)
        return res
