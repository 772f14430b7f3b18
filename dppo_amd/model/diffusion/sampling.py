"""Schedule helpers.  Mirrors ``dppo/model/diffusion/sampling.py:10-31`` (reference)."""
import numpy as np
import torch


def cosine_beta_schedule(timesteps, s=0.008, dtype=torch.float32):
    """Cosine abar schedule in float64 numpy, betas clipped to [0, 0.999], cast to fp32 (reference :10-20)."""
    n = timesteps + 1
    grid = np.linspace(0, n, n)
    abar = np.cos((grid / n + s) / (1 + s) * np.pi * 0.5) ** 2
    abar = abar / abar[0]
    return torch.tensor(np.clip(1 - abar[1:] / abar[:-1], 0, 0.999), dtype=dtype)


def extract(a, t, x_shape):
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def make_timesteps(batch_size, i, device):
    return torch.full((batch_size,), i, device=device, dtype=torch.long)
