"""Device-side rollout post-processing: GAE (reference agent/finetune/train_ppo_diffusion_agent.py:255-279)."""
from __future__ import annotations

import torch

from dppo_amd import hip


def gae_device(reward: torch.Tensor, values: torch.Tensor, terminated: torch.Tensor, last_values: torch.Tensor,
               gamma: float, gae_lambda: float, reward_scale_const: float = 1.0):
    """reward (S,E) float64, values/terminated (S,E) fp32, last_values (E,) fp32, all on the GPU.

    Returns (adv64, ret64, adv32, ret32).  Float64 reverse scan, one lane per env (dppo_gae).
    """
    hip.require_gpu(reward, "gae_device")
    S, E = reward.shape
    reward = reward.contiguous().double()
    values, terminated = values.contiguous().float(), terminated.contiguous().float()
    last_values = last_values.reshape(E).contiguous().float()
    a64, r64 = torch.empty_like(reward), torch.empty_like(reward)
    a32, r32 = torch.empty_like(values), torch.empty_like(values)
    hip.check(hip.load().dppo_gae(reward.data_ptr(), values.data_ptr(), terminated.data_ptr(), last_values.data_ptr(),
                                  S, E, float(gamma), float(gae_lambda), float(reward_scale_const), a64.data_ptr(),
                                  r64.data_ptr(), a32.data_ptr(), r32.data_ptr(), hip.stream()), "dppo_gae")
    return a64, r64, a32, r32


# ----------------------------------------------------------------------------------------------------------------------
# Host side of the rollout loop (reference agent/finetune/train_ppo_diffusion_agent.py:101-151).
# The reference moves pageable numpy arrays with a blocking `.to(device)` and `.cpu().numpy()` on every env step and
# alternates strictly between the simulator (host) and the sampler (device).  Here the hand-off goes through pinned,
# double-buffered staging memory, and the env set may be split into groups that are software-pipelined: while the host
# steps group g's simulators the device samples the action chunks of group g+1.
# ----------------------------------------------------------------------------------------------------------------------
class PinnedHandoff:
    """obs numpy -> device tensor, action device tensor -> numpy, through two pinned staging buffers each way."""

    def __init__(self, n_envs: int, obs_shape, act_shape, device):
        self.device = torch.device(device)
        pin = self.device.type == "cuda"
        mk = lambda shape: [torch.empty((n_envs, *shape), dtype=torch.float32, pin_memory=pin) for _ in range(2)]
        self._obs, self._act = mk(obs_shape), mk(act_shape)
        self._ev = [torch.cuda.Event() if pin else None for _ in range(2)]
        self._i = 0

    def obs_to_device(self, obs_np, out: torch.Tensor = None) -> torch.Tensor:
        """Host copy into the pinned slot, then an asynchronous H2D on the current stream (into ``out`` when given)."""
        self._i ^= 1
        slot = self._obs[self._i]
        slot.copy_(torch.from_numpy(obs_np).reshape(slot.shape))
        if out is None:
            return slot.to(self.device, non_blocking=True)
        out.copy_(slot.reshape(out.shape), non_blocking=True)
        return out

    def action_to_host_async(self, action_dev: torch.Tensor) -> int:
        """Start the D2H of an action tensor; returns a ticket for ``action_numpy``.  Does not block the host."""
        i = self._i
        self._act[i].copy_(action_dev.reshape(self._act[i].shape), non_blocking=True)
        if self._ev[i] is not None:
            self._ev[i].record()
        return i

    def action_numpy(self, ticket: int):
        if self._ev[ticket] is not None:
            self._ev[ticket].synchronize()
        return self._act[ticket].numpy()


class GroupedVecEnv:
    """Several equally sized vectorised envs presented as one (reset_arg / step / seed over the concatenation), with the
    groups exposed to ``collect_rollout`` for pipelined stepping.  Group g owns envs [g * n, (g + 1) * n)."""

    def __init__(self, venvs):
        assert len(venvs) >= 1 and len({v.n_envs for v in venvs}) == 1
        self.groups = list(venvs)
        self.n_envs = sum(v.n_envs for v in venvs)

    def seed(self, seeds):
        n = self.groups[0].n_envs
        for g, v in enumerate(self.groups):
            v.seed(seeds[g * n:(g + 1) * n])

    @staticmethod
    def _cat_obs(obs_list):
        import numpy as np
        obs_list = [{k: np.stack([x[k] for x in o]) for k in o[0]} if isinstance(o, list) else o for o in obs_list]
        return {k: np.concatenate([o[k] for o in obs_list]) for k in obs_list[0]}

    def reset_arg(self, options_list=None):
        n = self.groups[0].n_envs
        return self._cat_obs([v.reset_arg(options_list=None if options_list is None else options_list[g * n:(g + 1) * n])
                              for g, v in enumerate(self.groups)])

    def step(self, action):
        import numpy as np
        n = self.groups[0].n_envs
        outs = [v.step(action[g * n:(g + 1) * n]) for g, v in enumerate(self.groups)]
        return (self._cat_obs([o[0] for o in outs]), np.concatenate([o[1] for o in outs]),
                np.concatenate([o[2] for o in outs]), np.concatenate([o[3] for o in outs]), sum((o[4] for o in outs), []))


def collect_rollout(model, venv, prev_obs, n_steps: int, act_steps: int, obs_buf, chains_buf, deterministic: bool = False,
                    handoffs=None):
    """``n_steps`` env steps of every env: sample on the device, step the simulators on the host, fill the
    device-resident rollout buffer (row = step * n_envs + env).

    venv: anything with ``step(action (n, act_steps, Da)) -> (obs dict, reward, terminated, truncated, info)``; a
    ``GroupedVecEnv`` is software-pipelined over its groups (the sampler call of a group is in flight while the host steps
    the other groups' simulators).  prev_obs: {"state": (n_envs, To, Do)} numpy.  Returns (reward (S,E), terminated (S,E),
    done (S,E), last obs dict) as numpy, like the reference's holders (:78-93).
    """
    import numpy as np
    groups = getattr(venv, "groups", [venv])
    G = len(groups)
    E = obs_buf.shape[0] // n_steps
    n = E // G
    assert n * G == E and obs_buf.shape[0] == n_steps * E
    dev = obs_buf.device
    state0 = prev_obs["state"]
    if handoffs is None:  # staging buffers live with the env object: allocated once, reused by every iteration
        key = (n, tuple(state0.shape[1:]), act_steps, chains_buf.shape[-1], str(dev))
        cache = getattr(venv, "_dppo_handoffs", None)
        if cache is None or cache[0] != key:
            Ta = model.horizon_steps  # whole trajectories cross PCIe (contiguous rows); the host keeps the first act_steps
            cache = (key, [PinnedHandoff(n, state0.shape[1:], (Ta, chains_buf.shape[-1] // Ta), dev) for _ in range(G)],
                     [torch.empty(n, chains_buf.shape[-1], device=dev) for _ in range(2 * G)])
            try:
                venv._dppo_handoffs = cache
            except AttributeError:
                pass
        handoffs, traj_bufs = cache[1], cache[2]
    else:
        traj_bufs = [torch.empty(n, chains_buf.shape[-1], device=dev) for _ in range(2 * G)]
    reward = np.zeros((n_steps, E))
    terminated = np.zeros((n_steps, E))
    done = np.zeros((n_steps, E))
    obs_g = [state0[g * n:(g + 1) * n] for g in range(G)]
    Kp1 = chains_buf.shape[1]

    To_Do = state0.shape[1:]
    direct = getattr(model, "supports_out", False)  # PPODiffusion: the kernel writes the buffer slices itself

    def launch(step, g):
        """H2D of group g's observations into its buffer rows, the sampler call writing the chain rows in place, the
        asynchronous D2H of the trajectories; returns the D2H ticket.  No allocation, no copy kernel."""
        r0 = step * E + g * n
        state = handoffs[g].obs_to_device(obs_g[g], out=obs_buf[r0:r0 + n]).view(n, *To_Do)
        if direct:
            traj = traj_bufs[2 * g + (step & 1)]
            model(cond={"state": state}, deterministic=deterministic, return_chain=True, out=(traj, chains_buf[r0:r0 + n]))
        else:
            smp = model(cond={"state": state}, deterministic=deterministic, return_chain=True)
            chains_buf[r0:r0 + n] = smp.chains.reshape(n, Kp1, -1)
            traj = smp.trajectories
        return handoffs[g].action_to_host_async(traj)

    tickets = [launch(0, g) for g in range(G)]
    for step in range(n_steps):
        for g in range(G):
            action = handoffs[g].action_numpy(tickets[g])[:, :act_steps]
            o, r, term, trunc, _ = groups[g].step(action)
            if isinstance(o, list):
                o = {k: np.stack([x[k] for x in o]) for k in o[0]}
            obs_g[g] = o["state"]
            sl = slice(g * n, (g + 1) * n)
            reward[step, sl], terminated[step, sl], done[step, sl] = r, term, term | trunc
            if step + 1 < n_steps:  # the next sampler call of this group runs while the host steps the other groups
                tickets[g] = launch(step + 1, g)
    return reward, terminated, done, {"state": np.concatenate(obs_g)}
