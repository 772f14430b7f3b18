"""Host side of the rollout loop (dppo_amd/util/rollout.py; reference agent/finetune/train_ppo_diffusion_agent.py:101-151):
buffer layout, reward / termination bookkeeping and the group pipeline, on CPU with a stand-in policy."""
from collections import namedtuple

import numpy as np
import torch

from dppo_amd.env.synthetic import SyntheticVecEnv
from dppo_amd.util.rollout import GroupedVecEnv, PinnedHandoff, collect_rollout

Sample = namedtuple("Sample", "trajectories chains")
E, DO, DA, TA, ACT, KFT, S = 8, 5, 3, 4, 2, 3, 6


class RowPolicy:
    """Deterministic per-row stand-in for PPODiffusion.forward: the action depends on that row's observation only."""
    horizon_steps = TA

    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self.W = torch.randn(DO, TA * DA, generator=g) * 0.3
        self.calls = []

    def __call__(self, cond, deterministic=False, return_chain=True):
        st = cond["state"]
        self.calls.append(st.shape[0])
        traj = torch.tanh(st[:, -1] @ self.W).reshape(-1, TA, DA)
        chains = torch.stack([traj * (k + 1) / (KFT + 1) for k in range(KFT + 1)], dim=1)
        return Sample(traj, chains)


def run(groups):
    n = E // groups
    venvs = [SyntheticVecEnv(n, DO, DA, 1, ACT, max_episode_steps=7, seed=100 + g * n) for g in range(groups)]
    venv = venvs[0] if groups == 1 else GroupedVecEnv(venvs)
    obs = venv.reset_arg()
    pol = RowPolicy()
    obs_buf, chains_buf = torch.zeros(S * E, DO), torch.zeros(S * E, KFT + 1, TA * DA)
    out = collect_rollout(pol, venv, obs, S, ACT, obs_buf, chains_buf)
    return out, obs_buf, chains_buf, pol, obs


def test_single_group_fills_buffers_like_the_reference_loop():
    (reward, term, done, last), obs_buf, chains_buf, pol, obs0 = run(1)
    assert pol.calls == [E] * S
    # replay by hand: same env, same policy, the reference's strictly alternating loop
    env = SyntheticVecEnv(E, DO, DA, 1, ACT, max_episode_steps=7, seed=100)
    o = env.reset_arg()
    ref = RowPolicy()
    for s in range(S):
        st = torch.from_numpy(o["state"]).float()
        smp = ref(cond={"state": st})
        np.testing.assert_array_equal(obs_buf[s * E:(s + 1) * E].numpy(), st.reshape(E, -1).numpy())
        np.testing.assert_array_equal(chains_buf[s * E:(s + 1) * E].numpy(), smp.chains.reshape(E, KFT + 1, -1).numpy())
        o, r, t, tr, _ = env.step(smp.trajectories[:, :ACT].numpy())
        np.testing.assert_array_equal(reward[s], r)
        np.testing.assert_array_equal(term[s], t)
        np.testing.assert_array_equal(done[s], t | tr)
    np.testing.assert_array_equal(last["state"], o["state"])
    assert done.sum() > 0  # max_episode_steps = 7 with 2 act steps per call: truncations happen inside the rollout


def test_two_pipelined_groups_equal_one_group():
    """Envs are independent and the stand-in policy is per-row, so splitting the env set into two pipelined groups must
    not change a single number: only the order of host / device work differs."""
    (r1, t1, d1, l1), o1, c1, _, _ = run(1)
    (r2, t2, d2, l2), o2, c2, pol, _ = run(2)
    assert pol.calls == [E // 2] * (2 * S)
    for a, b in ((r1, r2), (t1, t2), (d1, d2), (l1["state"], l2["state"])):
        np.testing.assert_array_equal(a, b)
    assert torch.equal(o1, o2) and torch.equal(c1, c2)


def test_grouped_env_presents_one_env():
    venv = GroupedVecEnv([SyntheticVecEnv(4, DO, DA, 1, ACT, seed=100 + 4 * g) for g in range(2)])
    one = SyntheticVecEnv(8, DO, DA, 1, ACT, seed=100)
    np.testing.assert_array_equal(venv.reset_arg()["state"], one.reset_arg()["state"])
    a = np.random.RandomState(0).uniform(-1, 1, size=(8, ACT, DA))
    og, rg, tg, ug, ig = venv.step(a)
    oo, ro, to, uo, io = one.step(a)
    np.testing.assert_array_equal(og["state"], oo["state"])
    np.testing.assert_array_equal(rg, ro)
    assert len(ig) == 8 and venv.n_envs == 8


def test_handoff_round_trip_without_a_gpu():
    h = PinnedHandoff(4, (1, DO), (ACT, DA), "cpu")
    x = np.random.RandomState(1).normal(size=(4, 1, DO)).astype(np.float32)
    d = h.obs_to_device(x)
    np.testing.assert_array_equal(d.numpy(), x)
    a = torch.arange(4 * ACT * DA, dtype=torch.float32).reshape(4, ACT, DA)
    np.testing.assert_array_equal(h.action_numpy(h.action_to_host_async(a)), a.numpy())
