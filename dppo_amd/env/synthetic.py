"""A dependency-free vectorised environment with the interface the agent drives.

The env layer proper is dppo_amd/env/gym_utils (worker-pool AsyncVectorEnv, MultiStep, MujocoLocomotionLowdimWrapper:
SURVEY.md 8f row 1); gym / mujoco / d4rl are not installed here, so the agent's end-to-end tests need a simulator that is.
This stand-in exposes the same calls the agent makes -- ``reset_arg``, ``step`` on (n_envs, act_steps, act_dim)
action chunks returning multi-step-summed rewards, ``seed`` -- over a stable linear system, so the training loop
can be exercised end to end on synthetic data.
"""
import numpy as np


class SyntheticVecEnv:
    def __init__(self, n_envs, obs_dim, action_dim, n_obs_steps=1, n_action_steps=4, max_episode_steps=1000, seed=0):
        self.n_envs, self.obs_dim, self.action_dim = n_envs, obs_dim, action_dim
        self.n_obs_steps, self.n_action_steps, self.max_episode_steps = n_obs_steps, n_action_steps, max_episode_steps
        rs = np.random.RandomState(1234)
        q, _ = np.linalg.qr(rs.normal(size=(obs_dim, obs_dim)))
        self.A = 0.97 * q
        self.Bm = 0.2 * rs.normal(size=(obs_dim, action_dim))
        self.seed([seed + i for i in range(n_envs)])
        self.x = np.zeros((n_envs, obs_dim))
        self.t = np.zeros(n_envs, dtype=np.int64)

    def seed(self, seeds):
        self.rngs = [np.random.RandomState(s) for s in seeds]

    def _obs(self):
        o = np.clip(self.x, -1, 1).astype(np.float32)
        return {"state": np.repeat(o[:, None], self.n_obs_steps, axis=1)}

    def _reset(self, i):
        self.x[i] = self.rngs[i].uniform(-0.5, 0.5, size=self.obs_dim)
        self.t[i] = 0

    def reset_arg(self, options_list=None):
        for i in range(self.n_envs):
            self._reset(i)
        return self._obs()

    def step(self, action):
        action = np.asarray(action, dtype=np.float64).reshape(self.n_envs, -1, self.action_dim)
        reward = np.zeros(self.n_envs)
        for s in range(action.shape[1]):
            a = np.clip(action[:, s], -1, 1)
            self.x = self.x @ self.A.T + a @ self.Bm.T
            reward += -np.square(self.x).sum(-1) - 0.01 * np.square(a).sum(-1) + 1.0
            self.t += 1
        terminated = np.abs(self.x).max(-1) > 5.0
        truncated = self.t >= self.max_episode_steps
        for i in np.where(terminated | truncated)[0]:  # reset_within_step semantics of the reference MultiStep
            self._reset(i)
        return self._obs(), reward, terminated, truncated, [{} for _ in range(self.n_envs)]


class SyntheticPixelVecEnv(SyntheticVecEnv):
    """The same linear system observed through a camera: obs = {"state": the first ``state_dim`` coordinates, "rgb": uint8
    (n_envs, n_obs_steps, C, H, W)} -- a bright square per camera whose position follows two state coordinates over a fixed
    gradient background (C = 3 per camera).  Exercises the pixel path of the agent without robomimic / robosuite."""

    def __init__(self, n_envs, state_dim, action_dim, rgb_shape, n_obs_steps=1, n_action_steps=4, max_episode_steps=1000, seed=0):
        super().__init__(n_envs, max(state_dim, 4), action_dim, n_obs_steps, n_action_steps, max_episode_steps, seed)
        self.state_dim, self.rgb_shape = state_dim, tuple(rgb_shape)
        C, H, W = self.rgb_shape
        yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        self.bg = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx + yy) * 127 // max(H + W - 2, 1))] *
                           (C // 3)).astype(np.uint8)

    def _obs(self):
        C, H, W = self.rgb_shape
        o = np.clip(self.x, -1, 1).astype(np.float32)
        rgb = np.broadcast_to(self.bg, (self.n_envs, C, H, W)).copy()
        side = max(H // 8, 2)
        for cam in range(C // 3):
            cx = ((o[:, (2 * cam) % self.obs_dim] * 0.5 + 0.5) * (W - side)).astype(np.int64)
            cy = ((o[:, (2 * cam + 1) % self.obs_dim] * 0.5 + 0.5) * (H - side)).astype(np.int64)
            for i in range(self.n_envs):
                rgb[i, 3 * cam:3 * cam + 3, cy[i]:cy[i] + side, cx[i]:cx[i] + side] = 255
        return {"state": np.repeat(o[:, None, :self.state_dim], self.n_obs_steps, axis=1),
                "rgb": np.repeat(rgb[:, None], self.n_obs_steps, axis=1)}


def make_venv(cfg):
    """cfg.env.name == 'synthetic' -> SyntheticVecEnv; anything else needs the reference's env stack."""
    env = cfg.env
    groups = int(env.get("pipeline_groups", 1))  # > 1: env groups stepped on the host while the device samples the others
    assert env.n_envs % groups == 0, "env.n_envs must be divisible by env.pipeline_groups"
    if str(env.name).startswith("synthetic-img"):
        assert groups == 1, "the pixel stand-in is not pipelined over env groups"
        return SyntheticPixelVecEnv(env.n_envs, cfg.obs_dim, cfg.action_dim, list(cfg.shape_meta.obs.rgb.shape), cfg.cond_steps,
                                    cfg.act_steps, env.get("max_episode_steps", 1000), cfg.get("seed", 42))
    if str(env.name).startswith("synthetic"):
        mk = lambda g: SyntheticVecEnv(env.n_envs // groups, cfg.obs_dim, cfg.action_dim, cfg.cond_steps, cfg.act_steps,
                                       env.get("max_episode_steps", 1000), cfg.get("seed", 42) + g * (env.n_envs // groups))
        if groups == 1:
            return mk(0)
        from dppo_amd.util.rollout import GroupedVecEnv
        return GroupedVecEnv([mk(g) for g in range(groups)])
    # named simulators: this build's own worker-pool env layer (dppo_amd/env/gym_utils: AsyncVectorEnv + the cfg's wrappers),
    # which needs gym + the simulator stack of that env on the host (not installed in this image: ImportError says so)
    from dppo_amd.env.gym_utils import make_async
    mk = lambda: make_async(env.name, env_type=env.get("env_type", None), num_envs=env.n_envs // groups, asynchronous=True,
                            max_episode_steps=env.max_episode_steps, wrappers=env.get("wrappers", None),
                            obs_dim=cfg.obs_dim, action_dim=cfg.action_dim)
    if groups == 1:
        return mk()
    from dppo_amd.util.rollout import GroupedVecEnv
    venvs = [mk() for _ in range(groups)]
    for v in venvs:
        v.n_envs = env.n_envs // groups
    return GroupedVecEnv(venvs)
