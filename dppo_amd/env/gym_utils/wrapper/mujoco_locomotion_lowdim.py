"""State-observation wrapper for the Gym MuJoCo locomotion tasks.

Mirrors ``dppo/env/gym_utils/wrapper/mujoco_locomotion_lowdim.py:12-73`` (reference): observations are min-max normalised
to [-1, 1] with the dataset's statistics (``normalization.npz``: obs_min / obs_max / action_min / action_max) and returned
as ``{"state": obs}``; the policy's actions in [-1, 1] are mapped back to the simulator's range.  Works on anything with
``reset() -> obs`` and ``step(a) -> (obs, reward, done, info)`` (gym 0.22 API, as the reference uses); gym itself is only
needed for the simulator, not for this class.  ``normalize_obs`` / ``unnormalize_action`` are vectorised: the same object
serves a batched simulator (``dppo_amd.env.multi_step.MultiStepVec``).
"""
import numpy as np


class _Box:
    """What the callers read from ``observation_space["state"]`` / ``action_space`` when gym is absent."""

    def __init__(self, low, high):
        self.low, self.high, self.shape, self.dtype = low, high, low.shape, low.dtype

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)


class MujocoLocomotionLowdimWrapper:
    def __init__(self, env, normalization_path=None, normalization=None):
        self.env = env
        self.action_space = getattr(env, "action_space", None)
        stats = normalization if normalization is not None else np.load(normalization_path)  # .npz: arrays only
        self.obs_min, self.obs_max = np.asarray(stats["obs_min"]), np.asarray(stats["obs_max"])
        self.action_min, self.action_max = np.asarray(stats["action_min"]), np.asarray(stats["action_max"])
        example = np.asarray(self.env.reset())
        self.observation_space = {"state": _Box(np.full_like(example, -1), np.full_like(example, 1))}

    def seed(self, seed=None):
        np.random.seed(seed=seed) if seed is not None else np.random.seed()

    def reset(self, **kwargs):
        """Passed-in arguments other than ``options["seed"]`` are ignored, like the reference (:46-56)."""
        new_seed = (kwargs.get("options") or {}).get("seed", None)
        if new_seed is not None:
            self.seed(seed=new_seed)
        return {"state": self.normalize_obs(self.env.reset())}

    def normalize_obs(self, obs):
        return 2 * ((obs - self.obs_min) / (self.obs_max - self.obs_min + 1e-6) - 0.5)

    def unnormalize_action(self, action):
        action = (action + 1) / 2  # [-1, 1] -> [0, 1]
        return action * (self.action_max - self.action_min) + self.action_min

    def step(self, action):
        raw_obs, reward, done, info = self.env.step(self.unnormalize_action(action))
        return {"state": self.normalize_obs(raw_obs)}, reward, done, info

    def render(self, **kwargs):
        return self.env.render()

    def close(self):
        if hasattr(self.env, "close"):
            self.env.close()
