"""Conv denoiser (BASELINE configs[4]'s network): the oracle's Unet1D restatement against the
reference's golden vectors (CPU), and the HIP path (dppo_unet_* through the C ABI) against the same vectors (GPU)."""
import numpy as np
import pytest
import torch

from oracle import dppo_oracle as O
from tests.golden.make_golden_cases import UNET_CHAIN_CASES, UNET_SPECS
from tests.test_oracle_golden import make_cfg

T = torch.from_numpy


@pytest.mark.parametrize("name", sorted(UNET_SPECS))
def test_oracle_unet_forward_and_blocks(golden, name):
    g = golden("g13_unet")
    u = O.UnetSpec(**UNET_SPECS[name])
    p = O.unet_init_params(u, 81)
    with torch.no_grad():
        eps = O.unet_forward(p, u, T(g[f"{name}_x"]), T(g[f"{name}_t"]), T(g[f"{name}_state"]))
        by = O.residual_block1d(p, "down_modules.1.0", T(g[f"{name}_blk_x"]), T(g[f"{name}_blk_cond"]), u)
        bx = T(g[f"{name}_blk_x"])
        cy = O.conv1d_block(p, "final_conv.0", bx[:, :u.dim].repeat(1, 2, 1)[:, :u.dim], u)
    np.testing.assert_allclose(eps.numpy(), g[f"{name}_eps"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(by.numpy(), g[f"{name}_blk_y"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(cy.numpy(), g[f"{name}_cb_y"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("case", sorted(UNET_CHAIN_CASES))
def test_oracle_unet_chains_and_logprobs(golden, case):
    g = golden("g13_unet")
    sname, B, kw, det = UNET_CHAIN_CASES[case]
    u = O.UnetSpec(**UNET_SPECS[sname])
    cfg = make_cfg(u, kw)
    base, ft = O.unet_init_params(u, 21), O.unet_init_params(u, 22)
    state, noise = T(g[f"{case}_state"]), T(g[f"{case}_noise"])
    traj, chains = O.sample_chain(cfg, u, base, ft, state, noise, deterministic=det)
    np.testing.assert_allclose(chains.numpy(), g[f"{case}_chains"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(traj.numpy(), g[f"{case}_traj"], rtol=2e-4, atol=2e-4)
    with torch.no_grad():
        lp = O.chain_logprob(cfg, u, base, ft, state, T(g[f"{case}_chains"]))
    np.testing.assert_allclose(lp.numpy(), g[f"{case}_logprobs"], rtol=2e-4, atol=2e-4)
