"""Vectorised action-chunk stepping (dppo_amd/env/multi_step.py) against a per-env replay of the reference's MultiStep
rules (env/gym_utils/wrapper/multi_step.py:135-185): chunk reward = sum of executed steps, no stepping after the episode
ended inside the chunk, max_episode_steps counted per chunk step, last-n observation window with the first observation
repeated, reset within the step."""
import numpy as np

from dppo_amd.env.multi_step import MultiStepVec


class CounterSim:
    """n independent scalar envs: obs = [t, id]; reward = action sum + 0.1 t; env i terminates when t reaches term[i]."""

    def __init__(self, term):
        self.term = np.asarray(term)
        self.n = len(term)
        self.t = np.zeros(self.n)
        self.resets = np.zeros(self.n, dtype=int)

    def _obs(self):
        return np.stack([self.t, np.arange(self.n, dtype=float)], axis=1)

    def reset(self, mask=None):
        mask = np.ones(self.n, dtype=bool) if mask is None else mask
        self.t[mask] = 0
        self.resets[mask] += 1
        return self._obs()

    def step(self, action, active):
        self.t[active] += 1
        r = action.sum(-1) + 0.1 * self.t
        return self._obs(), r, self.t >= self.term


def replay_one(term, max_steps, n_obs, chunks):
    """One env through the reference's per-env loop (written out from its rules), reset_within_step=True."""
    t, cnt, hist, out = 0, 0, [np.array([0.0])], []
    for chunk in chunks:
        reward, terminated, truncated = 0.0, False, False
        for a in chunk:
            cnt += 1
            if terminated or truncated:
                break
            t += 1
            hist.append(np.array([float(t)]))
            reward += a.sum() + 0.1 * t
            if t >= term:
                terminated = True
            elif max_steps is not None and cnt >= max_steps:
                truncated = True
        window = hist[-n_obs:]
        window = [window[0]] * (n_obs - len(window)) + window
        if terminated or truncated:
            t, cnt, hist = 0, 0, [np.array([0.0])]
            window = [hist[0]] * n_obs
        out.append((np.array([w[0] for w in window]), reward, terminated, truncated))
    return out


def test_chunks_match_the_per_env_rules():
    term = [3, 100, 7, 5]
    rs = np.random.RandomState(0)
    chunks = rs.uniform(-1, 1, size=(6, 4, 4, 2))  # 6 chunks, 4 envs, 4 action steps, Da = 2
    for max_steps, n_obs in ((None, 1), (6, 3), (5, 2)):
        env = MultiStepVec(CounterSim(term), 4, n_obs_steps=n_obs, n_action_steps=4, max_episode_steps=max_steps)
        obs = env.reset_arg()["state"]
        assert obs.shape == (4, n_obs, 2) and (obs[..., 0] == 0).all()
        got = [env.step(chunks[c]) for c in range(6)]
        for i in range(4):
            want = replay_one(term[i], max_steps, n_obs, [chunks[c][i] for c in range(6)])
            for c in range(6):
                o, r, te, tr, info = got[c]
                w_obs, w_r, w_te, w_tr = want[c]
                np.testing.assert_array_equal(o["state"][i, :, 0], w_obs)
                assert o["state"][i, 0, 1] == i  # rows never mix
                assert abs(r[i] - w_r) < 1e-5
                assert bool(te[i]) == w_te and bool(tr[i]) == w_tr
                assert ("final_obs" in info[i]) == w_tr


def test_agent_protocol_shapes():
    env = MultiStepVec(CounterSim([4, 4]), 2, n_obs_steps=2, n_action_steps=3, max_episode_steps=10)
    obs = env.reset_arg()
    o, r, te, tr, info = env.step(np.zeros((2, 3, 1)))
    assert o["state"].shape == (2, 2, 2) and r.shape == (2,) and te.dtype == bool and len(info) == 2
    assert list(o["state"][0, :, 0]) == [2.0, 3.0]


def test_rollout_collector_over_multistep_groups():
    """The pieces compose: two MultiStepVec groups behind GroupedVecEnv, driven by collect_rollout with a stand-in policy."""
    from collections import namedtuple

    import torch

    from dppo_amd.util.rollout import GroupedVecEnv, collect_rollout
    Sample = namedtuple("Sample", "trajectories chains")

    class Policy:
        horizon_steps = 4

        def __call__(self, cond, deterministic=False, return_chain=True):
            st = cond["state"]
            traj = torch.tanh(st[:, -1, :1]).reshape(-1, 1, 1).repeat(1, 4, 1)
            return Sample(traj, torch.stack([traj * 0.5, traj], dim=1))

    groups = [MultiStepVec(CounterSim([5, 9]), 2, n_obs_steps=2, n_action_steps=3, max_episode_steps=8) for _ in range(2)]
    venv = GroupedVecEnv(groups)
    obs = venv.reset_arg()
    S, E = 4, 4
    obs_buf, chains_buf = torch.zeros(S * E, 2 * 2), torch.zeros(S * E, 2, 4)
    reward, term, done, last = collect_rollout(Policy(), venv, obs, S, 3, obs_buf, chains_buf)
    assert reward.shape == (S, E) and done.sum() > 0 and last["state"].shape == (E, 2, 2)
    assert (reward[:, 0] == reward[:, 2]).all() and (reward[:, 1] == reward[:, 3]).all()  # the two groups are twins
    assert torch.equal(obs_buf[:E].reshape(E, 2, 2), torch.from_numpy(obs["state"]).float())
