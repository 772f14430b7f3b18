"""Per-tensor relative error of the pixel networks' PPO-loss gradients against the golden fixture (debug aid)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.conftest import golden as _g  # noqa
from tests.golden.make_golden_cases import VIS_LOSS_CASES
from tests.test_vision import cuda_cond, hip_vision_model

T = torch.from_numpy
case = sys.argv[1] if len(sys.argv) > 1 else "vmlp_loss"
g = dict(np.load("tests/golden/g17_vision_loss.npz"))
name, N, kw, rh = VIS_LOSS_CASES[case]
m, v, trunk, cspec = hip_vision_model(name, 31, "fp32", kw)
d = lambda k: T(g[f"{case}_{k}"]).cuda()
res = m.loss(cuda_cond(g, case, u8=True), d("prev"), d("next"), d("kinds"), d("returns"), d("oldvalues"), d("adv"),
             d("oldlogprobs"), use_bc_loss=False, reward_horizon=rh)
(res[0] + 0.5 * res[2]).backward()
for who, net in (("gactor", m.actor_ft), ("gcritic", m.critic)):
    for k, p in net.named_parameters():
        x = p.grad.double().cpu().numpy().reshape(-1)
        key = f"{case}_{who}_{k}"
        if key in g:
            r = g[key].astype(np.float64).reshape(-1)
            xs = x
        else:
            r, xs = g[key + "__sub"].astype(np.float64), x[::61]
        print(f"{who:8s} {k:50s} |ref| {np.linalg.norm(r):.3e}  rel {np.linalg.norm(xs - r) / (np.linalg.norm(r) + 1e-30):.3e}")
