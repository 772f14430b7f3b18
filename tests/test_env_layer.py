"""Host env layer: MujocoLocomotionLowdimWrapper, per-env MultiStep, and the worker-pool
AsyncVectorEnv against in-process stepping of the same environments (CPU only; a deterministic stand-in simulator with the
gym 0.22 API -- gym / MuJoCo are not installed here)."""
import numpy as np
import pytest

from dppo_amd.env.gym_utils import make_async
from dppo_amd.env.gym_utils.async_vector_env import AsyncVectorEnv, SyncVectorEnv
from dppo_amd.env.gym_utils.wrapper.mujoco_locomotion_lowdim import MujocoLocomotionLowdimWrapper
from dppo_amd.env.gym_utils.wrapper.multi_step import MultiStep

NORM = dict(obs_min=np.array([-2.0, -1.0, 0.0]), obs_max=np.array([2.0, 3.0, 1.0]),
            action_min=np.array([-0.5, -2.0]), action_max=np.array([0.5, 1.0]))


class ToySim:
    """x' = 0.9 x + B a + noise(seed); done when |x| > 3; obs (3,), action (2,).  gym 0.22 API."""

    def __init__(self, fail_at=None, dict_obs=False):
        self.dict_obs = dict_obs
        self.B = np.array([[0.5, 0.0], [0.1, -0.4], [0.0, 0.3]])
        self.rs = np.random.RandomState(0)
        self.fail_at, self.n = fail_at, 0
        self.x = np.zeros(3)

    def seed(self, seed=None):
        self.rs = np.random.RandomState(seed)

    def reset(self, **kwargs):
        self.x = self.rs.uniform(-1, 1, size=3)
        return self._o()

    def _o(self):
        return {"state": self.x.copy()} if self.dict_obs else self.x.copy()

    def step(self, a):
        self.n += 1
        if self.fail_at is not None and self.n >= self.fail_at:
            raise ValueError("simulator blew up")
        self.x = 0.9 * self.x + self.B @ np.asarray(a, dtype=np.float64) + 0.05 * self.rs.normal(size=3)
        return self._o(), float(1.0 - np.square(self.x).sum()), bool(np.abs(self.x).max() > 3.0), {"energy": float(np.square(a).sum())}

    def close(self):
        pass


def toy():
    return ToySim()


def toy_dict():
    return ToySim(dict_obs=True)


MS_ONLY = {"multi_step": {"n_obs_steps": 2, "n_action_steps": 4, "max_episode_steps": 10, "reset_within_step": True}}


WRAPPERS = {"mujoco_locomotion_lowdim": {"normalization": NORM},
            "multi_step": {"n_obs_steps": 2, "n_action_steps": 4, "max_episode_steps": 10, "reset_within_step": True}}


def test_lowdim_wrapper_normalises_like_the_reference():
    env = MujocoLocomotionLowdimWrapper(ToySim(), normalization=NORM)
    env.env.seed(3)
    raw = ToySim()
    raw.seed(3)
    o = env.reset()
    x = raw.reset()
    want = 2 * ((x - NORM["obs_min"]) / (NORM["obs_max"] - NORM["obs_min"] + 1e-6) - 0.5)
    np.testing.assert_array_equal(o["state"], want)
    a = np.array([0.2, -1.0])  # policy action in [-1, 1]
    raw_a = (a + 1) / 2 * (NORM["action_max"] - NORM["action_min"]) + NORM["action_min"]
    o2, r, d, info = env.step(a)
    x2, r2, d2, _ = raw.step(raw_a)
    np.testing.assert_array_equal(o2["state"], 2 * ((x2 - NORM["obs_min"]) / (NORM["obs_max"] - NORM["obs_min"] + 1e-6) - 0.5))
    assert r == r2 and d == d2
    np.testing.assert_allclose(env.unnormalize_action(np.array([-1.0, -1.0])), NORM["action_min"])
    np.testing.assert_allclose(env.unnormalize_action(np.array([1.0, 1.0])), NORM["action_max"])
    assert env.observation_space["state"].shape == (3,)


def test_multi_step_wrapper_chunk_semantics():
    class Count:  # obs = step index; done at step 6
        action_space = None

        def reset(self, **kw):
            self.t = 0
            return np.array([0.0])

        def step(self, a):
            self.t += 1
            return np.array([float(self.t)]), 1.0, self.t == 6, {}

    ms = MultiStep(Count(), n_obs_steps=3, n_action_steps=4, max_episode_steps=100, reset_within_step=False)
    o = ms.reset()
    np.testing.assert_array_equal(o[:, 0], [0, 0, 0])  # the first observation fills the window
    o, r, term, trunc, info = ms.step(np.zeros((4, 1)))
    np.testing.assert_array_equal(o[:, 0], [2, 3, 4])
    assert (r, term, trunc) == (4.0, False, False)
    o, r, term, trunc, info = ms.step(np.zeros((4, 1)))  # terminates at the chunk's 2nd step: the rest is not executed
    np.testing.assert_array_equal(o[:, 0], [4, 5, 6])
    assert (r, term, trunc) == (2.0, True, False) and ms.cnt == 7  # cnt counts the slot at which the loop broke (reference :147-150)
    # truncation by max_episode_steps, with a reset inside the step and the final observation kept
    ms = MultiStep(Count(), n_obs_steps=1, n_action_steps=4, max_episode_steps=4, reset_within_step=True)
    ms.reset()
    o, r, term, trunc, info = ms.step(np.zeros((4, 1)))
    assert (r, term, trunc) == (4.0, False, True)
    np.testing.assert_array_equal(info["final_obs"][:, 0], [4])
    np.testing.assert_array_equal(o[:, 0], [0])  # fresh episode


@pytest.mark.parametrize("n_workers", [1, 2, 5])
def test_worker_pool_equals_in_process_stepping(n_workers):
    n = 5
    # (the lowdim wrapper's seed() seeds numpy's GLOBAL generator, like the reference's: per-process state, so the
    # equality test wraps a simulator that owns its generator; the wrapper is exercised in the pool below)
    venv = make_async("toy", num_envs=n, asynchronous=True, wrappers=MS_ONLY, env_fn=toy_dict, n_workers=n_workers)
    ref = make_async("toy", num_envs=n, asynchronous=False, wrappers=MS_ONLY, env_fn=toy_dict)
    try:
        assert isinstance(venv, AsyncVectorEnv) and isinstance(ref, SyncVectorEnv) and venv.n_envs == n
        seeds = [100 + i for i in range(n)]
        venv.seed(seeds), ref.seed(seeds)  # the per-env form the agent uses (train_agent.py:58-61)
        for e, s in zip(ref.envs, seeds):
            assert e.env.rs.get_state()[1][0] == np.random.RandomState(s).get_state()[1][0]
        o1, o2 = venv.reset_arg(), ref.reset_arg()
        assert o1["state"].shape == (n, 2, 3)
        np.testing.assert_array_equal(o1["state"], o2["state"])
        rs = np.random.RandomState(0)
        n_trunc = 0
        for it in range(8):
            act = rs.uniform(-1, 1, size=(n, 4, 2))
            r1, r2 = venv.step(act), ref.step(act)
            np.testing.assert_array_equal(r1[0]["state"], r2[0]["state"])
            for a, b in zip(r1[1:4], r2[1:4]):
                np.testing.assert_array_equal(a, b)
            assert len(r1[4]) == n and set(r1[4][0]) == set(r2[4][0])
            np.testing.assert_array_equal(r1[4][2]["energy"], r2[4][2]["energy"])
            n_trunc += int(r1[3].sum())
        assert n_trunc >= n  # max_episode_steps = 10 with 4-step chunks: every env was truncated and reset inside a step
        assert venv.call("get_prev_action")[0].shape == (1, 2)
        one = venv.reset_one_arg(3, options={})
        two = ref.reset_one_arg(3, options={})
        np.testing.assert_array_equal(one["state"], two["state"])
        assert venv.get_attr("n_action_steps") == (4,) * n
        venv.set_attr("verbose", True)
        assert venv.get_attr("verbose") == (True,) * n
    finally:
        venv.close()
        ref.close()
    assert venv.closed and all(not p.is_alive() for p in venv.procs)


def failing():
    return MultiStep(ToySim(fail_at=2), n_action_steps=1)


def test_worker_exception_surfaces_in_the_parent_with_its_traceback():
    venv = AsyncVectorEnv([failing, failing, failing], n_workers=2)
    venv.reset_arg()
    venv.step(np.zeros((3, 1, 2)))
    with pytest.raises(RuntimeError, match="simulator blew up"):
        venv.step(np.zeros((3, 1, 2)))
    assert venv.closed


def test_rollout_collector_drives_the_worker_pool():
    """dppo_amd.util.rollout.collect_rollout over AsyncVectorEnv (host-only policy stand-in): the buffer rows hold the
    observations the pool returned, in (step, env) order."""
    import torch
    from dppo_amd.util.rollout import collect_rollout

    class Policy:  # the sampler's host-visible surface, on CPU tensors
        horizon_steps, action_dim, ft_denoising_steps = 4, 2, 2

        def __call__(self, cond, deterministic=False, return_chain=True):
            from dppo_amd.model.diffusion.diffusion import Sample
            B = cond["state"].shape[0]
            a = torch.tanh(cond["state"][:, -1, :2]).reshape(B, 1, 2).repeat(1, 4, 1)
            return Sample(a, a.reshape(B, 1, 4, 2).repeat(1, 3, 1, 1))

    n, S = 4, 3
    venv = make_async("toy", num_envs=n, asynchronous=True, wrappers=WRAPPERS, env_fn=toy, n_workers=2)
    try:
        venv.seed([7 + i for i in range(n)])
        obs0 = venv.reset_arg()
        obs_buf = torch.zeros(S * n, 2 * 3)
        chains_buf = torch.zeros(S * n, 3, 8)
        reward, term, done, last = collect_rollout(Policy(), venv, obs0, S, 4, obs_buf, chains_buf)
        assert reward.shape == (S, n) and np.isfinite(reward).all()
        np.testing.assert_allclose(obs_buf[:n].numpy(), obs0["state"].reshape(n, -1), rtol=1e-6)
        assert last["state"].shape == (n, 2, 3)
    finally:
        venv.close()
