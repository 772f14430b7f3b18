"""Device-resident sequence dataset for pre-training (reference agent/dataset/sequence.py:19-185).

Same sampling rule as the reference's ``StitchedSequenceDataset``: trajectories are stored back to back; a sample is a
window of ``horizon_steps`` actions that stays inside one trajectory, with the ``cond_steps`` most recent observations
(the first observation repeated at a trajectory's start).  Instead of one ``__getitem__`` per sample and a DataLoader,
whole minibatches are gathered on the device from two index vectors: the data never leaves HBM.
"""
from collections import namedtuple

import numpy as np
import torch

Batch = namedtuple("Batch", "actions conditions")


class StitchedSequenceDataset:
    def __init__(self, dataset_path=None, horizon_steps=4, cond_steps=1, max_n_episodes=10000, device="cuda:0",
                 states=None, actions=None, traj_lengths=None):
        if dataset_path is not None:
            if not str(dataset_path).endswith(".npz"):
                raise ValueError("dppo_amd loads .npz datasets only (numpy.load, allow_pickle=False)")
            data = np.load(dataset_path, allow_pickle=False)
            states, actions, traj_lengths = data["states"], data["actions"], data["traj_lengths"]
        traj_lengths = np.asarray(traj_lengths)[:max_n_episodes].astype(np.int64)
        total = int(traj_lengths.sum())
        self.horizon_steps, self.cond_steps, self.device = horizon_steps, cond_steps, device
        self.states = torch.as_tensor(np.asarray(states)[:total]).float().to(device)
        self.actions = torch.as_tensor(np.asarray(actions)[:total]).float().to(device)
        start, before = self.make_indices(traj_lengths, horizon_steps)
        self.start = torch.from_numpy(start).to(device)
        self.before = torch.from_numpy(before).to(device)

    @staticmethod
    def make_indices(traj_lengths, horizon_steps):
        """(window start, steps since the trajectory began) of every window that fits inside its trajectory (:174-187)."""
        starts, befores, cur = [], [], 0
        for n in traj_lengths:
            k = int(n) - horizon_steps + 1
            if k > 0:
                starts.append(np.arange(cur, cur + k))
                befores.append(np.arange(k))
            cur += int(n)
        if not starts:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
        return np.concatenate(starts).astype(np.int64), np.concatenate(befores).astype(np.int64)

    def __len__(self):
        return self.start.numel()

    def gather(self, idx: torch.Tensor) -> Batch:
        """Minibatch of samples ``idx`` (device int64): actions (B,Ta,Da), conditions {"state": (B,To,Do)}."""
        s, nb = self.start[idx], self.before[idx]
        act = self.actions[s[:, None] + torch.arange(self.horizon_steps, device=s.device)[None]]
        back = torch.arange(self.cond_steps - 1, -1, -1, device=s.device)[None]  # most recent observation last
        obs = self.states[s[:, None] - torch.minimum(back, nb[:, None])]
        return Batch(act, {"state": obs})

    def __getitem__(self, i):
        b = self.gather(torch.tensor([i], device=self.start.device))
        return Batch(b.actions[0], {"state": b.conditions["state"][0]})

    def epoch(self, batch_size: int, generator=None, drop_last: bool = False):
        """Shuffled minibatches covering the dataset once (the reference's DataLoader(shuffle=True))."""
        perm = torch.randperm(len(self), generator=generator).to(self.start.device)
        for i in range(0, len(self), batch_size):
            idx = perm[i:i + batch_size]
            if drop_last and idx.numel() < batch_size:
                break
            yield self.gather(idx)


def synthetic_dataset(obs_dim, action_dim, horizon_steps, cond_steps=1, n_traj=32, traj_len=64, seed=0, device="cuda:0"):
    """Trajectories of a smooth expert on a random linear system, normalised to [-1, 1]: a stand-in for the D4RL /
    robomimic files the reference downloads (no network here)."""
    rs = np.random.RandomState(seed)
    A = 0.95 * np.linalg.qr(rs.normal(size=(obs_dim, obs_dim)))[0]
    Kg = 0.3 * rs.normal(size=(action_dim, obs_dim))
    states, actions = [], []
    for _ in range(n_traj):
        x = rs.uniform(-0.5, 0.5, size=obs_dim)
        for _ in range(traj_len):
            a = np.tanh(Kg @ x + 0.05 * rs.normal(size=action_dim))
            states.append(np.clip(x, -1, 1))
            actions.append(a)
            x = A @ x + 0.1 * np.pad(a, (0, max(0, obs_dim - action_dim)))[:obs_dim]
    return StitchedSequenceDataset(horizon_steps=horizon_steps, cond_steps=cond_steps, device=device,
                                   states=np.array(states, dtype=np.float32), actions=np.array(actions, dtype=np.float32),
                                   traj_lengths=[traj_len] * n_traj)
