"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz).

The fixtures were produced by tests/golden/make_golden.py, which imports the reference from
/root/reference in the build container.  Tolerances: fp32 <= 1e-5 rel / 1e-6 abs on tables,
network outputs, chains, log-probs and losses; <= 1e-4 rel on gradients.
"""
import numpy as np
import pytest
import torch

from oracle import dppo_oracle as O

torch.set_num_threads(4)
T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------ G1
def test_schedule_tables_bitexact(golden):
    g = golden("g1_tables")
    for K in (20, 100):
        tab = O.ddpm_tables(K)
        for k, v in tab.items():
            assert np.array_equal(v.numpy(), g[f"K{K}_{k}"]), (K, k)
    d = O.ddim_tables(100, 5)
    for k, v in d.items():
        assert np.array_equal(v.numpy(), g[f"ddim100_5_{k}"]), k
    assert O.eta_fixed_value(1.0) == pytest.approx(float(g["eta_fixed_base1"][0]), abs=0)
    assert O.eta_fixed_value(0.5) == pytest.approx(float(g["eta_fixed_base05"][0]), abs=0)


def test_survey_check_values():
    # SURVEY.md 8a row A1 check values dumped from the imported reference
    t = O.ddpm_tables(20)
    assert t["alphas_cumprod"][0].item() == pytest.approx(0.9920073, rel=1e-6)
    assert t["sqrt_recip_alphas_cumprod"][19].item() == pytest.approx(406.2368, rel=1e-6)
    d = O.ddim_tables(100, 5)
    assert d["ddim_t"].tolist() == [80, 60, 40, 20, 0]


# ------------------------------------------------------------------ G2
@pytest.mark.parametrize("name", ["hopper", "can", "halfcheetah", "furniture_like", "plain_mlp", "kitchen_like", "square_like", "furniture_256", "ln_relu",
                                  "transport", "furniture_one_leg", "can_relu"])
def test_network_forward(golden, name):
    g = golden("g2_forward")
    a, c = O.named_specs(name)
    pa, pc = O.init_params(a, 11), O.init_params(c, 12)
    eps = O.actor_forward(pa, a, T(g[f"{name}_x"]), T(g[f"{name}_t"]), T(g[f"{name}_state"]))
    close(eps, g[f"{name}_eps"], rtol=1e-5, atol=2e-6)
    close(O.critic_forward(pc, c, T(g[f"{name}_state"])), g[f"{name}_value"], rtol=1e-5, atol=2e-6)


def test_param_counts():
    a, c = O.named_specs("hopper")
    na = sum(int(np.prod(s)) for _, s, _ in O.param_shapes(a))
    nc = sum(int(np.prod(s)) for _, s, _ in O.param_shapes(c))
    assert (na, nc) == (553020, 134913)  # SURVEY.md section 8 header


# ------------------------------------------------------------------ G3 / G4
CHAIN_CASES = {
    "ddpm20_ft10": ("hopper", dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), False),
    "ddpm20_ft10_det": ("hopper", dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), True),
    "ddpm20_ft20": ("hopper", dict(denoising_steps=20, ft_denoising_steps=20, randn_clip_value=3,
                                   final_action_clip_value=1.0), False),
    "ddpm100_can": ("can", dict(denoising_steps=100, ft_denoising_steps=10, randn_clip_value=3,
                                min_sampling_denoising_std=0.08), False),
    "ddim100_5": ("hopper", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                 randn_clip_value=3, eps_clip_value=2.0), False),
    "ddim100_5_det": ("hopper", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                     randn_clip_value=3), True),
    "ddim100_10_ft4": ("halfcheetah", dict(denoising_steps=100, ft_denoising_steps=4, use_ddim=True,
                                           ddim_steps=10, randn_clip_value=3), False),
    "furniture_like": ("furniture_like", dict(denoising_steps=20, ft_denoising_steps=5, randn_clip_value=3), False),
    "kitchen_like": ("kitchen_like", dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), False),
    "square_like": ("square_like", dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), False),
    "furniture_256": ("furniture_256", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                            randn_clip_value=3), False),
    "ln_relu": ("ln_relu", dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), False),
    "transport": ("transport", dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3,
                                    min_sampling_denoising_std=0.1, min_logprob_denoising_std=0.1), False),
    "furniture_one_leg": ("furniture_one_leg", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                                    randn_clip_value=3, min_sampling_denoising_std=0.04), False),
    "ddpm100_can_relu": ("can_relu", dict(denoising_steps=100, ft_denoising_steps=10, randn_clip_value=3,
                                          min_sampling_denoising_std=0.08), False),
    "ddpm20_halfcheetah": ("halfcheetah", dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), False),
}


def make_cfg(a, kw):
    kw = dict(kw)
    if kw.get("use_ddim"):
        kw["eta"] = O.eta_fixed_value(1.0)
    return O.DiffusionCfg(horizon_steps=a.horizon_steps, action_dim=a.action_dim, **kw)


@pytest.mark.parametrize("case", sorted(CHAIN_CASES))
def test_sampling_chain_and_logprobs(golden, case):
    g = golden("g3_chains")
    sname, kw, det = CHAIN_CASES[case]
    a, _ = O.named_specs(sname)
    cfg = make_cfg(a, kw)
    base, ft = O.init_params(a, 21), O.init_params(a, 22)
    state, noise = T(g[f"{case}_state"]), T(g[f"{case}_noise"])
    traj, chains = O.sample_chain(cfg, a, base, ft, state, noise, deterministic=det)
    assert tuple(chains.shape) == g[f"{case}_chains"].shape
    # the early (t ~ K) steps amplify eps by sqrt(1/abar - 1) up to 400x before the x0 clamp: 1e-4 abs
    close(chains, g[f"{case}_chains"], rtol=1e-4, atol=1e-4)
    close(traj, g[f"{case}_traj"], rtol=1e-4, atol=1e-4)
    with torch.no_grad():
        lp = O.chain_logprob(cfg, a, base, ft, state, T(g[f"{case}_chains"]))
    close(lp, g[f"{case}_logprobs"], rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ G5
LOSS_CASES = {
    "default": ("hopper", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01,
                               clip_ploss_coef_base=0.001)),
    "vclip_nonorm": ("hopper", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.1,
                                    clip_ploss_coef_base=0.01, clip_vloss_coef=0.2, norm_adv=False)),
    "quantile_rh2": ("hopper", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01,
                                    clip_advantage_lower_quantile=0.05, clip_advantage_upper_quantile=0.95)),
    "can_k100": ("can", dict(denoising_steps=100, ft_denoising_steps=10, clip_ploss_coef=0.01)),
    "ddim": ("hopper", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                            clip_ploss_coef=0.01)),
    "furniture_like": ("furniture_like", dict(denoising_steps=20, ft_denoising_steps=5, clip_ploss_coef=0.01)),
    "kitchen_like": ("kitchen_like", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01)),
    "square_like": ("square_like", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01)),
    "furniture_256": ("furniture_256", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                            clip_ploss_coef=0.001)),
    "ln_relu": ("ln_relu", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01)),
    "transport": ("transport", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01,
                                    clip_ploss_coef_base=0.001, min_sampling_denoising_std=0.1,
                                    min_logprob_denoising_std=0.1)),
    "furniture_one_leg": ("furniture_one_leg", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                                    clip_ploss_coef=0.001, clip_ploss_coef_base=0.001,
                                                    min_sampling_denoising_std=0.04)),
    "can_relu_k100": ("can_relu", dict(denoising_steps=100, ft_denoising_steps=10, clip_ploss_coef=0.01)),
    "halfcheetah": ("halfcheetah", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01,
                                        clip_ploss_coef_base=0.01)),
}


def check_grad(g, key, grad, rtol, atol):
    grad = grad.detach().numpy()
    if key in g:
        np.testing.assert_allclose(grad, g[key], rtol=rtol, atol=atol)
    else:
        np.testing.assert_allclose(grad.reshape(-1)[::61], g[key + "__sub"], rtol=rtol, atol=atol)
        nrm = float(g[key + "__norm"])
        assert np.sqrt((grad.astype(np.float64) ** 2).sum()) == pytest.approx(nrm, rel=1e-4, abs=1e-7)


@pytest.mark.parametrize("case", sorted(LOSS_CASES))
def test_ppo_loss_and_grads(golden, case):
    g = golden("g5_loss")
    sname, kw = LOSS_CASES[case]
    a, c = O.named_specs(sname)
    cfg = make_cfg(a, dict(kw, gamma_denoising=0.99, randn_clip_value=3))
    base, ft, cr = O.init_params(a, 31), O.init_params(a, 32), O.init_params(c, 33)
    for p in list(ft.values()) + list(cr.values()):
        p.requires_grad_(True)
    res = O.ppo_loss(cfg, a, c, base, ft, cr, T(g[f"{case}_state"]), T(g[f"{case}_prev"]), T(g[f"{case}_next"]),
                     T(g[f"{case}_kinds"]), T(g[f"{case}_returns"]), T(g[f"{case}_oldvalues"]),
                     T(g[f"{case}_adv"]), T(g[f"{case}_oldlogprobs"]),
                     reward_horizon=int(g[f"{case}_reward_horizon"]))
    stats = g[f"{case}_stats"]
    got = np.array([res[0].item(), float(res[1]), res[2].item(), res[3], res[4], res[5], float(res[6]), res[7]])
    np.testing.assert_allclose(got, stats, rtol=2e-5, atol=2e-6)
    (res[0] + 0.5 * res[2]).backward()
    for k, p in ft.items():
        check_grad(g, f"{case}_gactor_{k}", p.grad, rtol=1e-4, atol=1e-6)
    for k, p in cr.items():
        check_grad(g, f"{case}_gcritic_{k}", p.grad, rtol=1e-4, atol=1e-6)


def test_discount_closed_form_matches_python_list(golden):
    g = golden("g5_loss")
    a, c = O.named_specs("hopper")
    cfg = make_cfg(a, dict(LOSS_CASES["default"][1], gamma_denoising=0.99))
    base, ft, cr = O.init_params(a, 31), O.init_params(a, 32), O.init_params(c, 33)
    args = [T(g[f"default_{k}"]) for k in ("state", "prev", "next", "kinds", "returns", "oldvalues", "adv",
                                           "oldlogprobs")]
    with torch.no_grad():
        r0 = O.ppo_loss(cfg, a, c, base, ft, cr, *args, python_list_discount=True)
        r1 = O.ppo_loss(cfg, a, c, base, ft, cr, *args, python_list_discount=False)
    assert r0[0].item() == pytest.approx(r1[0].item(), rel=1e-6)


# ------------------------------------------------------------------ G6
def test_reward_scaler(golden):
    g = golden("g6_reward_scaler")
    sc = O.RewardScalerOracle(4)
    for it in range(3):
        out = sc(g[f"it{it}_reward"], g[f"it{it}_first"])
        np.testing.assert_allclose(out, g[f"it{it}_scaled"], rtol=1e-12, atol=0)
        assert sc.var == pytest.approx(float(g[f"it{it}_var"]), rel=1e-12)


def test_gae_against_definition():
    """GAE lives in the un-importable agent loop (needs wandb/hydra); cross-check the recurrence
    (train_ppo_diffusion_agent.py:255-279) against the direct sum A_t = sum_l (g*lam)^l delta_{t+l}."""
    rs = np.random.RandomState(0)
    S, E, gm, lam = 16, 4, 0.99, 0.95
    r, v = rs.normal(size=(S, E)), rs.normal(size=(S, E))
    term = (rs.uniform(size=(S, E)) < 0.2).astype(np.float64)
    last = rs.normal(size=(E,))
    adv, ret = O.gae(r, v, term, last, gm, lam, reward_scale_const=0.7)
    vn = np.vstack([v[1:], last[None]])
    delta = 0.7 * r + gm * vn * (1 - term) - v
    ref = np.zeros_like(r)
    for t in range(S):
        w = np.ones(E)
        for l in range(t, S):
            ref[t] += w * delta[l]
            w = w * gm * lam * (1 - term[l])
    np.testing.assert_allclose(adv, ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ret, ref + v, rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------ G7
def test_adamw_and_clip(golden):
    g = golden("g7_adamw")
    p = T(g["p0"].copy())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i in range(3):
        O.adamw_step(p, T(g[f"g{i}"]), m, v, i + 1, lr=1e-3, weight_decay=0.01)
        close(p, g[f"p{i + 1}"], rtol=1e-6, atol=1e-7)
    gr = T(g["clip_in"].copy())
    tot = O.clip_grad_norm([gr], 1.5)
    assert tot == pytest.approx(float(g["clip_total"]), rel=1e-6)
    close(gr, g["clip_out"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------ G8 behaviour-cloning term
BC_CASES = {
    "bc_ddpm": ("hopper", dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3)),
    "bc_ddim_kitchen": ("kitchen_like", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                            randn_clip_value=3, min_sampling_denoising_std=0.08)),
}


@pytest.mark.parametrize("case", sorted(BC_CASES))
def test_bc_loss_matches_reference(golden, case):
    g = golden("g8_bc")
    sname, kw = BC_CASES[case]
    a, _ = O.named_specs(sname)
    cfg = make_cfg(a, kw)
    base = O.init_params(a, 41)
    ft = {k: v.clone().requires_grad_(True) for k, v in O.init_params(a, 42).items()}
    bc, chains = O.bc_loss(cfg, a, base, ft, T(g[f"{case}_state"]), T(g[f"{case}_noise"]))
    np.testing.assert_allclose(chains.numpy(), g[f"{case}_base_chains"], rtol=1e-5, atol=1e-5)
    assert float(bc.detach()) == pytest.approx(float(g[f"{case}_bc_loss"]), rel=1e-5, abs=1e-6)
    bc.backward()
    for k, v in ft.items():
        check_grad(g, f"{case}_gbc_{k}", v.grad if v.grad is not None else torch.zeros_like(v), 2e-4, 1e-7)



# ------------------------------------------------------------------ G9 supervised denoising loss (pre-training)
MSE_CASES = {"mse_hopper": ("hopper", 20), "mse_can_k100": ("can", 100), "mse_square_like": ("square_like", 20),
             "mse_ln_relu": ("ln_relu", 20), "mse_can_relu_k100": ("can_relu", 100)}


@pytest.mark.parametrize("case", sorted(MSE_CASES))
def test_denoise_mse_matches_reference(golden, case):
    """DiffusionModel.p_losses (reference model/diffusion/diffusion.py:325-363): q_sample, the loss and every gradient."""
    g = golden("g9_denoise_mse")
    sname, K = MSE_CASES[case]
    a, _ = O.named_specs(sname)
    prm = {k: v.clone().requires_grad_(True) for k, v in O.init_params(a, 51).items()}
    x0, state, t, noise = (T(g[f"{case}_{k}"]) for k in ("x0", "state", "t", "noise"))
    close(O.q_sample(K, x0, t, noise), g[f"{case}_xnoisy"], rtol=1e-6, atol=1e-7)
    loss = O.denoise_mse_loss(K, a, prm, x0, state, t, noise)
    assert float(loss.detach()) == pytest.approx(float(g[f"{case}_loss"]), rel=1e-5)
    loss.backward()
    for k, v in prm.items():
        check_grad(g, f"{case}_g_{k}", v.grad if v.grad is not None else torch.zeros_like(v), 2e-4, 1e-7)


# ------------------------------------------------------------------ G11 DiffusionEval
from tests.golden.make_golden_cases import EVAL_CASES  # noqa: E402


@pytest.mark.parametrize("case", sorted(EVAL_CASES))
def test_eval_sampling_matches_reference(golden, case):
    """DiffusionEval.forward (reference diffusion_eval.py + diffusion.py:261-316) == the oracle's deterministic chain with
    the checkpoint's base / fine-tuned weights (seed 61 / 62 of the recipe)."""
    g = golden("g11_eval")
    sname, B, kw, ft, kind = EVAL_CASES[case]
    a, _ = O.named_specs(sname)
    cfg = make_cfg(a, dict(kw, ft_denoising_steps=ft))
    base, ftw = O.init_params(a, 61), O.init_params(a, 62)
    traj, _ = O.sample_chain(cfg, a, base, ftw if kind == "rl" else base, T(g[f"{case}_state"]), T(g[f"{case}_noise"]),
                             deterministic=True)
    close(traj, g[f"{case}_traj"], rtol=1e-4, atol=1e-4)
