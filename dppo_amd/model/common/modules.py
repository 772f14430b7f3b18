"""SpatialEmb parameter container and the random-shift augmentation.  Mirrors ``dppo/model/common/modules.py``
(reference): ``SpatialEmb`` (:10-41; its arithmetic is part of ``dppo_vis_encode``), ``RandomShiftsAug`` (:44-68)."""
from __future__ import annotations

import torch
from torch import nn


class _Slot(nn.Module):
    pass


class SpatialEmb(nn.Module):
    def __init__(self, num_patch, patch_dim, prop_dim, proj_dim, dropout):
        super().__init__()
        if dropout:
            raise NotImplementedError("dppo_amd: dropout in SpatialEmb is not built (every shipped cfg leaves it 0)")
        self.patch_dim, self.prop_dim = patch_dim, prop_dim
        self.weight = nn.Parameter(torch.zeros(1, patch_dim, proj_dim))  # registered first: leads the state dict (:25)
        self.input_proj = nn.Sequential(nn.Linear(num_patch + prop_dim, proj_dim), nn.LayerNorm(proj_dim), _Slot())
        # Module.__setattr__ files Parameters and child modules separately and state_dict() lists a module's own Parameters
        # first, so `weight` precedes input_proj.* whatever the assignment order -- as in the reference
        nn.init.normal_(self.weight)

    def extra_repr(self) -> str:
        return f"weight: nn.Parameter ({self.weight.size()})"


class RandomShiftsAug:
    """Pad by ``pad`` replicated pixels, then crop back at a random integer offset per image.  The reference builds the crop
    with grid_sample on a pixel-aligned grid (:56-68): the sample points are exact pixel centres, so bilinear sampling
    returns the pixel itself -- a gather.  Draws the offsets with torch.randint like the reference (one (x, y) pair per image)."""

    def __init__(self, pad):
        self.pad = pad

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        n, c, h, w = x.shape
        assert h == w
        p = self.pad
        xp = nn.functional.pad(x.float(), (p, p, p, p), "replicate")
        shift = torch.randint(0, 2 * p + 1, size=(n, 1, 1, 2), device=x.device, dtype=torch.float32)
        sx, sy = shift[:, 0, 0, 0].long(), shift[:, 0, 0, 1].long()
        ar = torch.arange(h, device=x.device)
        rows = (sy[:, None] + ar[None, :])[:, None, :, None].expand(n, c, h, w)
        cols = (sx[:, None] + ar[None, :])[:, None, None, :].expand(n, c, h, w)
        bi = torch.arange(n, device=x.device)[:, None, None, None]
        ci = torch.arange(c, device=x.device)[None, :, None, None]
        return xp[bi, ci, rows, cols]
