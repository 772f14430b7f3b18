"""Action-chunk wrapper around ONE environment (the form the worker processes of ``AsyncVectorEnv`` hold).

Semantics of the reference's ``MultiStep`` (``dppo/env/gym_utils/wrapper/multi_step.py:82-221``): a chunk of
``n_action_steps`` actions is applied one by one and stops at the first termination / truncation (``cnt`` counts every
slot of the chunk, executed or not, like there); the chunk reward is the sum of the executed steps; the observation is the
last ``n_obs_steps`` observations with the oldest repeated while the episode is younger than that; a ``TimeLimit.truncated``
entry in ``info`` takes precedence over ``max_episode_steps``; with ``reset_within_step`` an episode that ended inside the
chunk is reset at once (the pre-reset observation goes to ``info["final_obs"]`` when it was a truncation).
``dppo_amd.env.multi_step.MultiStepVec`` is the same rule vectorised over a batched simulator.
"""
from collections import defaultdict, deque

import numpy as np


def stack_last_n(items, n):
    """(n,) + shape: the last n entries, the earliest available one repeated in front when there are fewer."""
    items = list(items)[-n:]
    arr = np.stack([np.asarray(x) for x in items])
    if len(items) < n:
        arr = np.concatenate([np.repeat(arr[:1], n - len(items), axis=0), arr])
    return arr


class MultiStep:
    def __init__(self, env, n_obs_steps=1, n_action_steps=1, max_episode_steps=None, reward_agg_method="sum",
                 prev_action=True, reset_within_step=False, pass_full_observations=False, verbose=False, **kwargs):
        if reward_agg_method != "sum":
            raise NotImplementedError("only reward_agg_method='sum' is used by the reference's cfgs")
        self.env = env
        self.n_obs_steps, self.n_action_steps = n_obs_steps, n_action_steps
        self.max_episode_steps, self.reset_within_step = max_episode_steps, reset_within_step
        self.pass_full_observations, self.prev_action, self.verbose = pass_full_observations, prev_action, verbose
        self.observation_space = getattr(env, "observation_space", None)
        self.action_space = getattr(env, "action_space", None)

    def __getattr__(self, name):  # gym.Wrapper behaviour: unknown attributes are the wrapped env's
        if name in ("env", "__setstate__"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def seed(self, seed=None):
        return self.env.seed(seed) if hasattr(self.env, "seed") else None

    def reset(self, seed=None, return_info=False, options=None):
        try:
            obs = self.env.reset(seed=seed, options=options or {}, return_info=return_info)
        except TypeError:  # a bare simulator whose reset() takes nothing
            obs = self.env.reset()
        self.obs = deque([obs], maxlen=max(self.n_obs_steps + 1, self.n_action_steps))
        self.action = deque(maxlen=max(self.n_obs_steps, 1))
        if self.prev_action and hasattr(self.action_space, "sample"):  # the reference seeds the history with a random action
            self.action.append(self.action_space.sample())
        self.info = defaultdict(lambda: deque(maxlen=self.n_obs_steps + 1))
        self.cnt = 0
        return self._get_obs(self.n_obs_steps)

    def step(self, action):
        action = np.asarray(action)
        if action.ndim == 1:
            action = action[None]
        terminated = truncated = False
        rewards, dones, executed = [], [], 0
        for act in action:
            self.cnt += 1
            if terminated or truncated:
                break
            observation, reward, done, info = self.env.step(act)
            executed += 1
            self.obs.append(observation)
            self.action.append(act)
            rewards.append(reward)
            if "TimeLimit.truncated" not in info:
                if done:
                    terminated = True
                elif self.max_episode_steps is not None and self.cnt >= self.max_episode_steps:
                    truncated = True
            else:
                truncated, terminated = info["TimeLimit.truncated"], done
            dones.append(truncated or terminated)
            for k, v in info.items():
                self.info[k].append(v)
        observation = self._get_obs(self.n_obs_steps)
        reward = np.sum(rewards)
        info = {k: stack_last_n(v, self.n_obs_steps) for k, v in self.info.items()}
        if self.pass_full_observations:
            info["full_obs"] = self._get_obs(executed)
        if self.reset_within_step and dones[-1]:
            if truncated:
                info["final_obs"] = observation
            observation = self.reset()
        return observation, reward, terminated, truncated, info

    def _get_obs(self, n):
        if isinstance(self.obs[-1], dict):
            return {k: stack_last_n([o[k] for o in self.obs], n) for k in self.obs[-1]}
        return stack_last_n(self.obs, n)

    def get_prev_action(self, n_steps=None):
        n_steps = self.n_obs_steps - 1 if n_steps is None else n_steps
        assert len(self.action) > 0, "no action has been taken in this episode yet"
        return stack_last_n(self.action, n_steps)

    def render(self, **kwargs):
        return self.env.render(**kwargs)

    def close(self):
        if hasattr(self.env, "close"):
            self.env.close()
