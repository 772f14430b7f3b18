"""Action-chunk stepping for a vectorised single-step simulator (reference env/gym_utils/wrapper/multi_step.py:82-221).

The reference wraps every environment in its own ``MultiStep`` (a Python object per env, stepped through a subprocess
pipe).  Here the same semantics run vectorised over all envs of a group in numpy -- one wrapper around an object that
steps n envs at once -- so that the rollout loop talks to the host simulators in whole (n_envs, act_steps, Da) chunks:

* the chunk's actions are applied one after the other; an env that terminates, or reaches ``max_episode_steps`` (counted
  like the reference's ``cnt``), stops being stepped for the rest of the chunk;
* the chunk reward is the SUM of the executed steps' rewards (``reward_agg_method="sum"``);
* the observation handed back is the last ``n_obs_steps`` observations, the earliest one repeated while the episode is
  younger than that (``stack_last_n_obs``);
* ``reset_within_step``: an env that ended inside the chunk is reset right away and its fresh observation returned
  (for a truncated env the pre-reset observation is kept in ``info["final_obs"]`` for bootstrapping).

Simulator protocol: ``reset(mask=None) -> obs (n, Do)`` (all envs, or only those in ``mask``; rows outside the mask are
ignored) and ``step(action (n, Da), active (n,) bool) -> obs (n, Do), reward (n,), done (n,)`` where rows with
``active == False`` must be left untouched.
"""
import numpy as np


class MultiStepVec:
    def __init__(self, sim, n_envs, n_obs_steps=1, n_action_steps=1, max_episode_steps=None, reset_within_step=True):
        self.sim, self.n_envs = sim, n_envs
        self.n_obs_steps, self.n_action_steps = n_obs_steps, n_action_steps
        self.max_episode_steps, self.reset_within_step = max_episode_steps, reset_within_step
        self.hist = None  # (n_envs, n_obs_steps, Do): last observations, oldest first
        self.cnt = np.zeros(n_envs, dtype=np.int64)

    def seed(self, seeds):
        if hasattr(self.sim, "seed"):
            self.sim.seed(seeds)

    def _restart(self, obs, mask):
        self.hist[mask] = obs[mask][:, None]  # a fresh episode: its first observation fills the whole window
        self.cnt[mask] = 0

    def reset_arg(self, options_list=None):
        obs = np.asarray(self.sim.reset(), dtype=np.float32)
        self.hist = np.repeat(obs[:, None], self.n_obs_steps, axis=1)
        self.cnt[:] = 0
        return {"state": self.hist.copy()}

    def step(self, action):
        action = np.asarray(action, dtype=np.float32).reshape(self.n_envs, -1, action.shape[-1])
        n = self.n_envs
        reward = np.zeros(n)
        terminated = np.zeros(n, dtype=bool)
        truncated = np.zeros(n, dtype=bool)
        for s in range(action.shape[1]):
            # the reference counts a chunk step before checking whether the episode already ended inside this chunk
            self.cnt += 1
            active = ~(terminated | truncated)
            if not active.any():
                continue
            obs, r, done = self.sim.step(action[:, s], active)
            obs = np.asarray(obs, dtype=np.float32)
            self.hist[active] = np.concatenate([self.hist[active][:, 1:], obs[active][:, None]], axis=1)
            reward[active] += np.asarray(r)[active]
            done = np.asarray(done, dtype=bool) & active
            terminated |= done
            if self.max_episode_steps is not None:
                truncated |= active & ~done & (self.cnt >= self.max_episode_steps)
        out = self.hist.copy()
        ended = terminated | truncated
        infos = [{} for _ in range(n)]
        if self.reset_within_step and ended.any():
            for i in np.where(truncated)[0]:
                infos[i]["final_obs"] = out[i].copy()
            fresh = np.asarray(self.sim.reset(ended), dtype=np.float32)
            self._restart(fresh, ended)
            out = self.hist.copy()
        return {"state": out}, reward, terminated, truncated, infos
