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


# ------------------------------------------------------------------ HIP path
def hip_unet(u, seed, prec, dev="cuda:0"):
    from dppo_amd.model.diffusion.unet import Unet1D
    m = Unet1D(action_dim=u.action_dim, cond_dim=u.cond_dim, diffusion_step_embed_dim=u.diffusion_step_embed_dim, dim=u.dim,
               dim_mults=list(u.dim_mults), smaller_encoder=u.smaller_encoder, kernel_size=u.kernel_size, n_groups=u.n_groups,
               activation_type=u.activation, cond_predict_scale=u.cond_predict_scale, groupnorm_eps=u.groupnorm_eps,
               horizon_steps=u.horizon_steps, precision=prec)
    m.load_state_dict(O.unet_init_params(u, seed), strict=True)
    return m.to(dev)


def test_state_dict_names_and_order_match_the_reference():
    for name, kw in UNET_SPECS.items():
        u = O.UnetSpec(**kw)
        m = hip_unet(u, 1, "fp32", dev="cpu")
        assert [k for k, _ in m.named_parameters()] == [n for n, _, _ in O.unet_param_shapes(u)], name
        assert list(m.state_dict()) == [n for n, _, _ in O.unet_param_shapes(u)], name


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("fp32", 2e-4), ("bf16", 6e-2)])
@pytest.mark.parametrize("name", sorted(UNET_SPECS))
def test_hip_unet_forward(golden, name, prec, tol):
    g = golden("g13_unet")
    u = O.UnetSpec(**UNET_SPECS[name])
    m = hip_unet(u, 81, prec)
    dev = "cuda:0"
    eps = m(T(g[f"{name}_x"]).to(dev), T(g[f"{name}_t"]).to(dev), {"state": T(g[f"{name}_state"]).to(dev)})
    np.testing.assert_allclose(eps.cpu().numpy(), g[f"{name}_eps"], rtol=tol, atol=tol)
    # ragged batch sizes (rows are independent: one workgroup per sample in the epilogues, GEMM row tiles of 128 / 256)
    ref = None
    for B in (1, 5, 130):
        gen = torch.Generator().manual_seed(B)
        x = torch.randn(B, u.horizon_steps, u.action_dim, generator=gen)
        t = torch.randint(0, 20, (B,), generator=gen)
        s = torch.rand(B, 1, u.cond_dim, generator=gen) * 2 - 1
        want = O.unet_forward(O.unet_init_params(u, 81), u, x, t, s)
        got = m(x.to(dev), t.to(dev), {"state": s.to(dev)})
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=tol, atol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", sorted(UNET_CHAIN_CASES))
def test_hip_unet_chains_and_logprobs(golden, case, prec):
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    from dppo_amd.model.diffusion.eta import EtaFixed
    g = golden("g13_unet")
    sname, B, kw, det = UNET_CHAIN_CASES[case]
    u = O.UnetSpec(**UNET_SPECS[sname])
    dev = "cuda:0"
    actor = hip_unet(u, 21, prec, dev="cpu")
    critic = CriticObs(cond_dim=u.cond_dim, mlp_dims=[256, 256, 256], residual_style=True, precision=prec)
    kw2 = dict(kw, eta=EtaFixed(base_eta=1.0)) if kw.get("use_ddim") else dict(kw)
    m = PPODiffusion(actor=actor, critic=critic, horizon_steps=u.horizon_steps, obs_dim=u.cond_dim, action_dim=u.action_dim,
                     device=dev, gamma_denoising=0.99, clip_ploss_coef=0.01, **kw2)
    m.actor_ft.load_state_dict(O.unet_init_params(u, 22), strict=True)
    state, noise = T(g[f"{case}_state"]).to(dev), T(g[f"{case}_noise"]).to(dev)
    smp = m(cond={"state": state}, deterministic=det, return_chain=True, noise=noise)
    assert tuple(smp.chains.shape) == g[f"{case}_chains"].shape
    ct = 5e-4 if prec == "fp32" else 8e-2  # GroupNorm renormalises every block: bf16 operand rounding does not shrink with depth
    np.testing.assert_allclose(smp.chains.cpu().numpy(), g[f"{case}_chains"], rtol=ct, atol=ct)
    np.testing.assert_allclose(smp.trajectories.cpu().numpy(), g[f"{case}_traj"], rtol=ct, atol=ct)
    lp = m.get_logprobs({"state": state}, T(g[f"{case}_chains"]).to(dev)).cpu().numpy()
    ref = g[f"{case}_logprobs"]
    sel = ref > -50
    lt = 2e-3 if prec == "fp32" else 1.0
    np.testing.assert_allclose(lp[sel], ref[sel], rtol=lt, atol=lt)
    assert np.abs(lp[sel] - ref[sel]).mean() <= (2e-4 if prec == "fp32" else 0.15)
    # in-kernel noise: reproducible
    torch.manual_seed(5)
    s1 = m(cond={"state": state})
    torch.manual_seed(5)
    assert torch.equal(s1.chains, m(cond={"state": state}).chains) and torch.isfinite(s1.chains).all()


# ------------------------------------------------------------------ G14: PPO loss / supervised loss with a conv actor
from tests.golden.make_golden_cases import UNET_LOSS_CASES, UNET_MSE_CASES  # noqa: E402
from tests.test_oracle_golden import check_grad  # noqa: E402

CRITIC = lambda u: O.NetSpec("critic", cond_dim=u.cond_dim, mlp_dims=[256, 256, 256], activation="Mish", residual=True)


@pytest.mark.parametrize("case", sorted(UNET_LOSS_CASES))
def test_oracle_unet_ppo_loss_and_grads(golden, case):
    g = golden("g14_unet_loss")
    sname, N, kw, rh = UNET_LOSS_CASES[case]
    u = O.UnetSpec(**UNET_SPECS[sname])
    c = CRITIC(u)
    cfg = make_cfg(u, dict(kw, gamma_denoising=0.99, randn_clip_value=3))
    base = O.unet_init_params(u, 31)
    ft = {k: v.clone().requires_grad_(True) for k, v in O.unet_init_params(u, 32).items()}
    cr = {k: v.clone().requires_grad_(True) for k, v in O.init_params(c, 33).items()}
    d = lambda k: T(g[f"{case}_{k}"])
    res = O.ppo_loss(cfg, u, c, base, ft, cr, d("state"), d("prev"), d("next"), d("kinds"), d("returns"), d("oldvalues"),
                     d("adv"), d("oldlogprobs"), reward_horizon=rh)
    got = np.array([res[0].item(), float(res[1]), res[2].item(), res[3], res[4], res[5], float(res[6]), res[7]])
    np.testing.assert_allclose(got, g[f"{case}_stats"], rtol=5e-5, atol=5e-6)
    (res[0] + 0.5 * res[2]).backward()
    for k, p in ft.items():
        check_grad(g, f"{case}_gactor_{k}", p.grad if p.grad is not None else torch.zeros_like(p), rtol=2e-3, atol=2e-6)
    for k, p in cr.items():
        check_grad(g, f"{case}_gcritic_{k}", p.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("case", sorted(UNET_MSE_CASES))
def test_oracle_unet_denoise_mse(golden, case):
    g = golden("g14_unet_loss")
    sname, K, N = UNET_MSE_CASES[case]
    u = O.UnetSpec(**UNET_SPECS[sname])
    prm = {k: v.clone().requires_grad_(True) for k, v in O.unet_init_params(u, 51).items()}
    d = lambda k: T(g[f"{case}_{k}"])
    loss = O.denoise_mse_loss(K, u, prm, d("x0"), d("state"), d("t"), d("noise"))
    assert float(loss.detach()) == pytest.approx(float(g[f"{case}_loss"]), rel=1e-5)
    loss.backward()
    for k, v in prm.items():
        check_grad(g, f"{case}_g_{k}", v.grad if v.grad is not None else torch.zeros_like(v), 2e-3, 1e-6)


def grad_report(g, prefix, named_grads):
    """(worst per-tensor relative L2 error over the tensors carrying >= 0.1 % of the gradient norm, whole-gradient norm ratio)"""
    worst, n_got, n_ref = ("", 0.0), 0.0, 0.0
    per = []
    for k, grad in named_grads:
        x = grad.double().cpu().numpy().reshape(-1)
        key = f"{prefix}_{k}"
        if key in g:
            r = g[key].astype(np.float64).reshape(-1)
            nr, xs = float(np.dot(r, r)), x
        else:
            r, xs = g[key + "__sub"].astype(np.float64), x[::61]
            nr = float(g[key + "__norm"]) ** 2
        per.append((k, nr, float(np.linalg.norm(xs - r) / (np.linalg.norm(r) + 1e-30))))
        n_got += float(np.dot(x, x))
        n_ref += nr
    for k, nr, e in per:
        if nr >= 1e-6 * n_ref and e > worst[1]:
            worst = (k, e)
    return worst, abs(np.sqrt(n_got / n_ref) - 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(UNET_LOSS_CASES))
def test_hip_unet_ppo_loss_and_grads(golden, case):
    """PPODiffusion.loss with a conv actor through dppo_unet_ppo_loss_fwd_bwd: fp32 statistics and EVERY gradient (time
    MLP, conv / GroupNorm / FiLM-encoder / residual-conv / down- and up-sampling parameters, critic) against the reference."""
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    from dppo_amd.model.diffusion.eta import EtaFixed
    g = golden("g14_unet_loss")
    sname, N, kw, rh = UNET_LOSS_CASES[case]
    u = O.UnetSpec(**UNET_SPECS[sname])
    c = CRITIC(u)
    dev = "cuda:0"
    actor = hip_unet(u, 31, "fp32", dev="cpu")
    critic = CriticObs(cond_dim=u.cond_dim, mlp_dims=[256, 256, 256], residual_style=True, precision="fp32")
    critic.load_state_dict(O.init_params(c, 33))
    kw2 = dict(kw, eta=EtaFixed(base_eta=1.0)) if kw.get("use_ddim") else dict(kw)
    m = PPODiffusion(actor=actor, critic=critic, horizon_steps=u.horizon_steps, obs_dim=u.cond_dim, action_dim=u.action_dim,
                     device=dev, gamma_denoising=0.99, randn_clip_value=3, **kw2)
    m.actor_ft.load_state_dict(O.unet_init_params(u, 32), strict=True)
    d = lambda k: T(g[f"{case}_{k}"]).to(dev)
    res = m.loss({"state": d("state")}, d("prev"), d("next"), d("kinds"), d("returns"), d("oldvalues"), d("adv"),
                 d("oldlogprobs"), use_bc_loss=False, reward_horizon=rh)
    got = np.array([res[0].item(), float(res[1]), res[2].item(), res[3], res[4], res[5], float(res[6]), res[7]])
    np.testing.assert_allclose(got, g[f"{case}_stats"], rtol=5e-4, atol=5e-5)
    (res[0] + 0.5 * res[2]).backward()
    worst, norm = grad_report(g, f"{case}_gactor", [(k, p.grad) for k, p in m.actor_ft.named_parameters()])
    assert worst[1] <= 5e-3 and norm <= 2e-3, (worst, norm)
    worst, norm = grad_report(g, f"{case}_gcritic", [(k, p.grad) for k, p in m.critic.named_parameters()])
    assert worst[1] <= 2e-3 and norm <= 1e-3, (worst, norm)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(UNET_MSE_CASES))
def test_hip_unet_denoise_mse_and_grads(golden, case):
    from dppo_amd.model.diffusion.diffusion import DiffusionModel
    g = golden("g14_unet_loss")
    sname, K, N = UNET_MSE_CASES[case]
    u = O.UnetSpec(**UNET_SPECS[sname])
    dev = "cuda:0"
    flat = {}
    for prec in ("fp32", "bf16"):
        net = hip_unet(u, 51, prec, dev="cpu")
        m = DiffusionModel(network=net, horizon_steps=u.horizon_steps, obs_dim=u.cond_dim, action_dim=u.action_dim, device=dev,
                           denoising_steps=K)
        d = lambda k: T(g[f"{case}_{k}"]).to(dev)
        loss = m.p_losses(d("x0"), {"state": d("state")}, d("t"), noise=d("noise"))
        loss.backward()
        flat[prec] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).double().cpu().numpy()
        if prec == "fp32":
            assert loss.item() == pytest.approx(float(g[f"{case}_loss"]), rel=2e-4)
            worst, norm = grad_report(g, f"{case}_g", [(k, p.grad) for k, p in net.named_parameters()])
            assert worst[1] <= 5e-3 and norm <= 2e-3, (worst, norm)
        else:
            assert loss.item() == pytest.approx(float(g[f"{case}_loss"]), rel=3e-2)
    x, y = flat["fp32"], flat["bf16"]
    assert float(np.dot(x, y) / (np.linalg.norm(x) * np.linalg.norm(y))) >= 0.98


@pytest.mark.gpu
def test_hip_unet_update_in_bf16_and_rollout_mode():
    """The benchmark-style path with a conv actor: ppo_update straight from a rollout buffer (rollout mode, bf16), ratio == 1
    against its own log-probs, finite gradients with the direction of the fp32 path."""
    from dppo_amd import hip
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    u = O.UnetSpec(**UNET_SPECS["unet_square"])
    dev = "cuda:0"
    out = {}
    for prec in ("fp32", "bf16"):
        actor = hip_unet(u, 41, prec, dev="cpu")
        critic = CriticObs(cond_dim=u.cond_dim, mlp_dims=[256, 256, 256], residual_style=True, precision=prec)
        critic.load_state_dict(O.init_params(CRITIC(u), 43))
        m = PPODiffusion(actor=actor, critic=critic, horizon_steps=u.horizon_steps, obs_dim=u.cond_dim, action_dim=u.action_dim,
                         device=dev, gamma_denoising=0.99, clip_ploss_coef=0.01, randn_clip_value=3, denoising_steps=20,
                         ft_denoising_steps=10)
        m.actor_ft.load_state_dict(O.unet_init_params(u, 42), strict=True)
        R, Kft, AF, N = 200, 10, u.horizon_steps * u.action_dim, 1000
        gen = torch.Generator().manual_seed(2)
        obs = (torch.rand(R, 1, u.cond_dim, generator=gen) * 2 - 1).to(dev)
        noise = torch.randn(21, R, AF, generator=gen).to(dev)
        chains = m(cond={"state": obs}, noise=noise).chains
        logp = m.get_logprobs({"state": obs}, chains).reshape(R, Kft, AF)
        val = m.critic({"state": obs}).reshape(R)
        ret, adv = val + torch.randn(R, generator=gen).to(dev), torch.randn(R, generator=gen).to(dev)
        inds = torch.randperm(R * Kft, generator=gen)[:N].to(dev).contiguous()
        st = m.ppo_update(obs.reshape(R, -1).contiguous(), chains.reshape(R, Kft + 1, AF).contiguous(), ret, val, adv, logp,
                          inds).cpu().numpy().copy()
        assert st[hip.STAT_RATIO] == pytest.approx(1.0, abs=1e-5 if prec == "fp32" else 1e-3)  # two forward kernels (inference / training)
        ga = m.actor_ft.flat_grads().double().cpu().numpy().copy()
        gc = m.critic.flat_grads().double().cpu().numpy().copy()
        assert np.isfinite(ga).all() and np.isfinite(gc).all() and np.linalg.norm(ga) > 0
        out[prec] = (ga, gc)
    cos = lambda x, y: float(np.dot(x, y) / (np.linalg.norm(x) * np.linalg.norm(y) + 1e-30))
    assert cos(out["fp32"][1], out["bf16"][1]) >= 0.99


# ------------------------------------------------------------------ G15: dim 40 (maps padded to 64 / 128 channels in their images)
from tests.golden.make_golden_cases import UNET40_CHAIN_CASES, UNET40_LOSS_CASES, UNET40_MSE_CASES, UNET40_SPECS  # noqa: E402


def test_oracle_dim40_fixtures(golden):
    g = golden("g15_unet_dim40")
    for name, kw in UNET40_SPECS.items():
        u = O.UnetSpec(**kw)
        with torch.no_grad():
            eps = O.unet_forward(O.unet_init_params(u, 81), u, T(g[f"{name}_x"]), T(g[f"{name}_t"]), T(g[f"{name}_state"]))
        np.testing.assert_allclose(eps.numpy(), g[f"{name}_eps"], rtol=2e-5, atol=2e-5)
    for case, (sname, N, kw, rh) in UNET40_LOSS_CASES.items():
        u = O.UnetSpec(**UNET40_SPECS[sname])
        c = CRITIC(u)
        cfg = make_cfg(u, dict(kw, gamma_denoising=0.99, randn_clip_value=3))
        ft = {k: v.clone().requires_grad_(True) for k, v in O.unet_init_params(u, 32).items()}
        cr = {k: v.clone().requires_grad_(True) for k, v in O.init_params(c, 33).items()}
        d = lambda k: T(g[f"{case}_{k}"])
        res = O.ppo_loss(cfg, u, c, O.unet_init_params(u, 31), ft, cr, d("state"), d("prev"), d("next"), d("kinds"), d("returns"),
                         d("oldvalues"), d("adv"), d("oldlogprobs"), reward_horizon=rh)
        got = np.array([res[0].item(), float(res[1]), res[2].item(), res[3], res[4], res[5], float(res[6]), res[7]])
        np.testing.assert_allclose(got, g[f"{case}_stats"], rtol=5e-5, atol=5e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("fp32", 2e-4), ("bf16", 6e-2)])
def test_hip_dim40_forward_and_chain(golden, prec, tol):
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    g = golden("g15_unet_dim40")
    dev = "cuda:0"
    for name, kw in UNET40_SPECS.items():
        u = O.UnetSpec(**kw)
        m = hip_unet(u, 81, prec)
        eps = m(T(g[f"{name}_x"]).to(dev), T(g[f"{name}_t"]).to(dev), {"state": T(g[f"{name}_state"]).to(dev)})
        np.testing.assert_allclose(eps.cpu().numpy(), g[f"{name}_eps"], rtol=tol, atol=tol)
    for case, (sname, B, kw, det) in UNET40_CHAIN_CASES.items():
        u = O.UnetSpec(**UNET40_SPECS[sname])
        actor = hip_unet(u, 21, prec, dev="cpu")
        critic = CriticObs(cond_dim=u.cond_dim, mlp_dims=[256, 256, 256], residual_style=True, precision=prec)
        m = PPODiffusion(actor=actor, critic=critic, horizon_steps=u.horizon_steps, obs_dim=u.cond_dim, action_dim=u.action_dim,
                         device=dev, gamma_denoising=0.99, clip_ploss_coef=0.01, **kw)
        m.actor_ft.load_state_dict(O.unet_init_params(u, 22), strict=True)
        smp = m(cond={"state": T(g[f"{case}_state"]).to(dev)}, deterministic=det, noise=T(g[f"{case}_noise"]).to(dev))
        ct = 5e-4 if prec == "fp32" else 8e-2
        np.testing.assert_allclose(smp.chains.cpu().numpy(), g[f"{case}_chains"], rtol=ct, atol=ct)
        lp = m.get_logprobs({"state": T(g[f"{case}_state"]).to(dev)}, T(g[f"{case}_chains"]).to(dev)).cpu().numpy()
        ref = g[f"{case}_logprobs"]
        sel = ref > -50
        assert np.abs(lp[sel] - ref[sel]).mean() <= (2e-4 if prec == "fp32" else 0.15)


@pytest.mark.gpu
def test_hip_dim40_losses_and_grads(golden):
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.diffusion import DiffusionModel
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    g = golden("g15_unet_dim40")
    dev = "cuda:0"
    for case, (sname, N, kw, rh) in UNET40_LOSS_CASES.items():
        u = O.UnetSpec(**UNET40_SPECS[sname])
        actor = hip_unet(u, 31, "fp32", dev="cpu")
        critic = CriticObs(cond_dim=u.cond_dim, mlp_dims=[256, 256, 256], residual_style=True, precision="fp32")
        critic.load_state_dict(O.init_params(CRITIC(u), 33))
        m = PPODiffusion(actor=actor, critic=critic, horizon_steps=u.horizon_steps, obs_dim=u.cond_dim, action_dim=u.action_dim,
                         device=dev, gamma_denoising=0.99, randn_clip_value=3, **kw)
        m.actor_ft.load_state_dict(O.unet_init_params(u, 32), strict=True)
        d = lambda k: T(g[f"{case}_{k}"]).to(dev)
        res = m.loss({"state": d("state")}, d("prev"), d("next"), d("kinds"), d("returns"), d("oldvalues"), d("adv"),
                     d("oldlogprobs"), use_bc_loss=False, reward_horizon=rh)
        got = np.array([res[0].item(), float(res[1]), res[2].item(), res[3], res[4], res[5], float(res[6]), res[7]])
        np.testing.assert_allclose(got, g[f"{case}_stats"], rtol=5e-4, atol=5e-5)
        (res[0] + 0.5 * res[2]).backward()
        worst, norm = grad_report(g, f"{case}_gactor", [(k, p.grad) for k, p in m.actor_ft.named_parameters()])
        assert worst[1] <= 5e-3 and norm <= 2e-3, (case, worst, norm)
    for case, (sname, K, N) in UNET40_MSE_CASES.items():
        u = O.UnetSpec(**UNET40_SPECS[sname])
        net = hip_unet(u, 51, "fp32", dev="cpu")
        m = DiffusionModel(network=net, horizon_steps=u.horizon_steps, obs_dim=u.cond_dim, action_dim=u.action_dim, device=dev,
                           denoising_steps=K)
        d = lambda k: T(g[f"{case}_{k}"]).to(dev)
        loss = m.p_losses(d("x0"), {"state": d("state")}, d("t"), noise=d("noise"))
        loss.backward()
        assert loss.item() == pytest.approx(float(g[f"{case}_loss"]), rel=2e-4)
        worst, norm = grad_report(g, f"{case}_g", [(k, p.grad) for k, p in net.named_parameters()])
        assert worst[1] <= 5e-3 and norm <= 2e-3, (case, worst, norm)
