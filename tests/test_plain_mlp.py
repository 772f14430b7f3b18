"""Plain (non-residual) MLP trunks, SURVEY §8 row A4 (model/common/mlp.py:27-81): the oracle against the reference's golden
vectors (CPU) and the HIP path -- layered gemm_nt forward / backward, host-looped sampler -- against the same vectors (GPU)."""
import numpy as np
import pytest
import torch

from oracle import dppo_oracle as O
from tests.golden.make_golden_cases import PLAIN_CASES
from tests.test_oracle_golden import check_grad, make_cfg

T = torch.from_numpy


@pytest.mark.parametrize("case", sorted(PLAIN_CASES))
def test_oracle_plain_mlp(golden, case):
    g = golden("g19_plain_mlp")
    sname, kw = PLAIN_CASES[case]
    a, c = O.named_specs(sname)
    cfg = make_cfg(a, dict(kw, gamma_denoising=0.99, randn_clip_value=3))
    base = O.init_params(a, 31)
    ft = {k: t.clone().requires_grad_(True) for k, t in O.init_params(a, 32).items()}
    cr = {k: t.clone().requires_grad_(True) for k, t in O.init_params(c, 33).items()}
    d = lambda k: T(g[f"{case}_{k}"])
    with torch.no_grad():
        eps = O.actor_forward(ft, a, d("x"), d("t"), d("state"))
        val = O.critic_forward(cr, c, d("state"))
        traj, chains = O.sample_chain(cfg, a, base, ft, d("state"), d("noise"))
        lp = O.chain_logprob(cfg, a, base, ft, d("state"), d("chains"))
    np.testing.assert_allclose(eps.numpy(), g[f"{case}_eps"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(val.numpy(), g[f"{case}_value"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(chains.numpy(), g[f"{case}_chains"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(lp.numpy(), g[f"{case}_logprobs"], rtol=2e-4, atol=2e-4)
    res = O.ppo_loss(cfg, a, c, base, ft, cr, d("state"), d("prev"), d("next"), d("kinds"), d("returns"), d("oldvalues"), d("adv"),
                     d("oldlogprobs"), reward_horizon=4)
    got = np.array([res[0].item(), float(res[1]), res[2].item(), res[3], res[4], res[5], float(res[6]), res[7]])
    np.testing.assert_allclose(got, g[f"{case}_stats"], rtol=5e-5, atol=5e-6)
    (res[0] + 0.5 * res[2]).backward()
    for k, p in ft.items():
        check_grad(g, f"{case}_gactor_{k}", p.grad, rtol=2e-3, atol=2e-6)
    for k, p in cr.items():
        check_grad(g, f"{case}_gcritic_{k}", p.grad, rtol=1e-4, atol=1e-6)
    prm = {k: t.clone().requires_grad_(True) for k, t in base.items()}
    loss = O.denoise_mse_loss(kw["denoising_steps"], a, prm, d("mse_x0"), d("state"), d("mse_t"), d("mse_noise"))
    assert float(loss.detach()) == pytest.approx(float(g[f"{case}_mse_loss"]), rel=1e-5)
    loss.backward()
    for k, t in prm.items():
        check_grad(g, f"{case}_mse_g_{k}", t.grad, 2e-3, 1e-6)


def build(case, prec):
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    from dppo_amd.model.diffusion.eta import EtaFixed
    from dppo_amd.model.diffusion.mlp_diffusion import DiffusionMLP
    sname, kw = PLAIN_CASES[case]
    a, c = O.named_specs(sname)
    actor = DiffusionMLP(action_dim=a.action_dim, horizon_steps=a.horizon_steps, cond_dim=a.cond_dim, time_dim=a.time_dim,
                         mlp_dims=list(a.mlp_dims), activation_type=a.activation, residual_style=False, precision=prec)
    critic = CriticObs(cond_dim=c.cond_dim, mlp_dims=list(c.mlp_dims), activation_type=c.activation, residual_style=False,
                       precision=prec)
    actor.load_state_dict(O.init_params(a, 31), strict=True)
    critic.load_state_dict(O.init_params(c, 33), strict=True)
    kw = dict(kw, gamma_denoising=0.99, randn_clip_value=3)
    if kw.get("use_ddim"):
        kw["eta"] = EtaFixed(base_eta=1.0)
    m = PPODiffusion(actor=actor, critic=critic, horizon_steps=a.horizon_steps, obs_dim=a.cond_dim, action_dim=a.action_dim,
                     device="cuda:0", precision=prec, **kw)
    m.actor_ft.load_state_dict({k: t.cuda() for k, t in O.init_params(a, 32).items()}, strict=True)
    m.actor_ft.mark_updated()
    return m, a, c


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", sorted(PLAIN_CASES))
def test_hip_plain_mlp_forward_chain_logprobs(golden, case, prec):
    g = golden("g19_plain_mlp")
    m, a, c = build(case, prec)
    d = lambda k: T(g[f"{case}_{k}"]).cuda()
    cond = {"state": d("state")}
    eps = m.actor_ft(d("x"), d("t"), cond).cpu().numpy()
    val = m.critic(cond).cpu().numpy()
    smp = m(cond=cond, deterministic=False, return_chain=True, noise=d("noise"))
    lp = m.get_logprobs(cond, d("chains")).cpu().numpy()
    if prec == "fp32":
        np.testing.assert_allclose(eps, g[f"{case}_eps"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(val, g[f"{case}_value"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(smp.chains.cpu().numpy(), g[f"{case}_chains"], rtol=5e-4, atol=5e-4)
        np.testing.assert_allclose(smp.trajectories.cpu().numpy(), g[f"{case}_traj"], rtol=5e-4, atol=5e-4)
        np.testing.assert_allclose(lp, g[f"{case}_logprobs"], rtol=1e-3, atol=1e-3)
    else:
        np.testing.assert_allclose(eps, g[f"{case}_eps"], rtol=5e-2, atol=5e-2 * float(np.abs(g[f"{case}_eps"]).max()))
        assert float(np.abs(smp.chains.cpu().numpy() - g[f"{case}_chains"]).mean()) < 0.05 and np.isfinite(lp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(PLAIN_CASES))
def test_hip_plain_mlp_losses_and_grads(golden, case):
    """PPODiffusion.loss and p_losses with plain trunks (fp32): statistics and every gradient against the reference."""
    from tests.test_unet import grad_report
    g = golden("g19_plain_mlp")
    m, a, c = build(case, "fp32")
    d = lambda k: T(g[f"{case}_{k}"]).cuda()
    res = m.loss({"state": d("state")}, d("prev"), d("next"), d("kinds"), d("returns"), d("oldvalues"), d("adv"), d("oldlogprobs"),
                 use_bc_loss=False, reward_horizon=4)
    got = np.array([res[0].item(), float(res[1]), res[2].item(), res[3], res[4], res[5], float(res[6]), res[7]])
    np.testing.assert_allclose(got, g[f"{case}_stats"], rtol=5e-4, atol=5e-5)
    (res[0] + 0.5 * res[2]).backward()
    (worst, e), nerr = grad_report(g, f"{case}_gactor", [(k, p.grad) for k, p in m.actor_ft.named_parameters()])
    assert e < 5e-3 and nerr < 1e-3, ("actor", worst, e, nerr)
    (worst, e), nerr = grad_report(g, f"{case}_gcritic", [(k, p.grad) for k, p in m.critic.named_parameters()])
    assert e < 5e-3 and nerr < 1e-3, ("critic", worst, e, nerr)
    net = m.network
    for p in net.parameters():
        p.requires_grad_(True)
    loss = m.p_losses(d("mse_x0"), {"state": d("state")}, d("mse_t"), noise=d("mse_noise"))
    assert float(loss.detach()) == pytest.approx(float(g[f"{case}_mse_loss"]), rel=2e-4)
    loss.backward()
    (worst, e), nerr = grad_report(g, f"{case}_mse_g", [(k, p.grad) for k, p in net.named_parameters()])
    assert e < 5e-3 and nerr < 1e-3, (worst, e, nerr)
