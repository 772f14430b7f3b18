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
