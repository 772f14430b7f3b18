"""Mixture-of-Gaussians policy PPO (second half): the oracle against the reference's golden vectors (CPU)
and the HIP path (dppo_gmm_* through the C ABI) against the same vectors (GPU)."""
import numpy as np
import pytest
import torch

from oracle import dppo_oracle as O
from tests.golden.make_golden_cases import GMM_CASES
from tests.test_oracle_golden import check_grad

T = torch.from_numpy


def setup(case):
    cond, tkw, Ta, Da, gkw, cdims = GMM_CASES[case]
    M = gkw["num_modes"]
    ms, ws = O.gmm_specs(cond, tkw["mlp_dims"], tkw["activation"], tkw["residual"], Da, Ta, M)
    c = O.NetSpec("critic", cond_dim=cond, mlp_dims=cdims, activation="Mish", residual=True)
    lv = None
    if gkw["learn_fixed_std"]:
        rs = np.random.RandomState(73)
        lv = T((np.log(gkw["fixed_std"] ** 2) + rs.uniform(-0.4, 0.4, size=Da * M)).astype(np.float32))
    return cond, tkw, Ta, Da, gkw, ms, ws, c, lv


@pytest.mark.parametrize("case", sorted(GMM_CASES))
def test_oracle_gmm(golden, case):
    g = golden("g20_gmm")
    cond, tkw, Ta, Da, gkw, ms, ws, c, lv = setup(case)
    gc = O.GmmCfg(**gkw)
    ft = {k: t.clone().requires_grad_(True) for k, t in O.gmm_init_params(ms, ws, 71).items()}
    cr = {k: t.clone().requires_grad_(True) for k, t in O.init_params(c, 72).items()}
    if lv is not None:
        lv = lv.clone().requires_grad_(True)
    d = lambda k: T(g[f"{case}_{k}"])
    with torch.no_grad():
        act = O.gmm_sample(gc, ms, ws, ft, lv, d("state"), d("modes"), d("noise"), Da, Ta)
        lp, _, _ = O.gmm_logprob(gc, ms, ws, ft, lv, d("state"), d("actions"), Da, Ta)
    np.testing.assert_allclose(act.numpy(), g[f"{case}_actions"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lp.numpy(), g[f"{case}_logprobs"], rtol=2e-4, atol=2e-4)
    res = O.gmm_ppo_loss(gc, ms, ws, c, ft, lv, cr, d("state"), d("actions"), d("returns"), d("oldvalues"), d("adv"),
                         d("oldlogprobs"), Da, Ta)
    got = np.array([res[0].item(), res[1].item(), res[2].item(), res[3], res[4], res[5], float(res[6]), res[7]])
    np.testing.assert_allclose(got, g[f"{case}_stats"], rtol=1e-4, atol=1e-5)
    (res[0] + 0.01 * res[1] + 0.5 * res[2]).backward()
    for k, p in ft.items():
        check_grad(g, f"{case}_gactor_{k}", p.grad, rtol=2e-3, atol=2e-6)
    for k, p in cr.items():
        check_grad(g, f"{case}_gcritic_{k}", p.grad, rtol=1e-4, atol=1e-6)
    if lv is not None:
        check_grad(g, f"{case}_gactor_logvar", lv.grad, rtol=1e-3, atol=1e-6)


def build(case, prec):
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.common.mlp_gmm import GMM_MLP
    from dppo_amd.model.rl.gmm_ppo import PPO_GMM
    cond, tkw, Ta, Da, gkw, ms, ws, c, lv = setup(case)
    actor = GMM_MLP(action_dim=Da, horizon_steps=Ta, cond_dim=cond, mlp_dims=list(tkw["mlp_dims"]), num_modes=gkw["num_modes"],
                    activation_type=tkw["activation"], residual_style=tkw["residual"], fixed_std=gkw["fixed_std"],
                    learn_fixed_std=gkw["learn_fixed_std"], std_min=gkw["std_min"], std_max=gkw["std_max"], precision=prec)
    sd = dict(O.gmm_init_params(ms, ws, 71))
    sd["logvar_min"], sd["logvar_max"] = actor.logvar_min.data.clone(), actor.logvar_max.data.clone()
    if lv is not None:
        sd["logvar"] = lv
    actor.load_state_dict(sd, strict=True)
    critic = CriticObs(cond_dim=cond, mlp_dims=list(c.mlp_dims), activation_type="Mish", residual_style=True, precision=prec)
    critic.load_state_dict(O.init_params(c, 72), strict=True)
    m = PPO_GMM(actor=actor, critic=critic, horizon_steps=Ta, device="cuda:0", clip_ploss_coef=gkw["clip_ploss_coef"],
                clip_vloss_coef=gkw.get("clip_vloss_coef"), norm_adv=True, precision=prec)
    return m


def test_gmm_state_dict_matches_the_reference_names():
    from dppo_amd.model.common.mlp_gmm import GMM_MLP
    m = GMM_MLP(7, 4, cond_dim=23, mlp_dims=[512, 512, 512], num_modes=5, residual_style=True, fixed_std=0.1, learn_fixed_std=True)
    ms, ws = O.gmm_specs(23, [512, 512, 512], "Mish", True, 7, 4, 5)
    assert list(m.state_dict()) == ["logvar", "logvar_min", "logvar_max"] + list(O.gmm_init_params(ms, ws, 1))
    assert m.flat_params().numel() == sum(t.numel() for t in O.gmm_init_params(ms, ws, 1).values())
    a, b = m.mean_net, m.weights_net  # the two trunks are slices of the one flat buffer
    assert a.flat_params().data_ptr() == m.flat_params().data_ptr()
    assert b.flat_params().data_ptr() == m.flat_params().data_ptr() + 4 * a.flat_params().numel()


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(GMM_CASES))
def test_hip_gmm(golden, case):
    """PPO_GMM through dppo_gmm_* (fp32): sampling with the recorded component / noise draws, log-probs, loss statistics and
    every gradient of pg + 0.01 entropy_loss + 0.5 v (both trunks, logvar, critic) against the reference."""
    from tests.test_unet import grad_report
    g = golden("g20_gmm")
    m = build(case, "fp32")
    d = lambda k: T(g[f"{case}_{k}"]).cuda()
    cond = {"state": d("state")}
    act = m(cond=cond, deterministic=False, modes=d("modes"), noise=d("noise"))
    np.testing.assert_allclose(act.cpu().numpy(), g[f"{case}_actions"], rtol=2e-4, atol=2e-4)
    lp, _, _ = m.get_logprobs(cond, d("actions"))
    np.testing.assert_allclose(lp.cpu().numpy(), g[f"{case}_logprobs"], rtol=2e-3, atol=2e-3)
    m.ent_coef = 0.01
    res = m.loss(cond, d("actions"), d("returns"), d("oldvalues"), d("adv"), d("oldlogprobs"))
    got = np.array([res[0].item(), res[1].item(), res[2].item(), res[3], res[4], res[5], float(res[6]), res[7]])
    np.testing.assert_allclose(got, g[f"{case}_stats"], rtol=2e-3, atol=2e-4)
    (res[0] + 0.5 * res[2]).backward()  # (the entropy term's gradient rides pg_loss: ent_coef above)
    named = [(k, p.grad) for k, p in m.actor_ft.named_parameters() if p.grad is not None]
    (worst, e), nerr = grad_report(g, f"{case}_gactor", named)
    assert e < 1e-2 and nerr < 2e-3, ("actor", worst, e, nerr)
    (worst, e), nerr = grad_report(g, f"{case}_gcritic", [(k, p.grad) for k, p in m.critic.named_parameters()])
    assert e < 5e-3 and nerr < 1e-3, ("critic", worst, e, nerr)


@pytest.mark.gpu
def test_hip_gmm_samples_follow_the_mixture():
    """In-kernel draws: the component frequencies follow softmax(logits) and the actions are finite (bf16)."""
    m = build("gmm_d3il", "bf16")
    torch.manual_seed(0)
    state = (torch.rand(1, 1, 4, device="cuda") * 2 - 1).repeat(20000, 1, 1)
    a = m(cond={"state": state}, deterministic=True).reshape(20000, -1)  # sigma = 1e-4: every action sits on a component's mean
    assert torch.isfinite(a).all()
    one = state[:1]
    means = torch.cat([m(cond={"state": one}, deterministic=True, modes=torch.tensor([k], device="cuda"),
                         noise=torch.zeros(1, 8, device="cuda")).reshape(1, -1) for k in range(5)])
    dist = torch.cdist(a, means)
    near, which = dist.min(dim=1)
    assert float(near.max()) < 5e-3  # every draw is one of the five component means (+- 1e-4 z)
    freq = torch.bincount(which, minlength=5).float() / 20000
    assert int((freq > 0.01).sum()) >= 2 and float(freq.max()) < 0.99  # the component is drawn, not fixed


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(GMM_CASES))
def test_hip_gmm_bf16(golden, case):
    """bf16 operands: log-probs within 0.15 absolute of the reference (a sum over Ta*Da = 8..28 element log-probs whose means carry
    bf16 rounding), value loss within 5e-2 relative, ratio statistics finite and close, critic gradient cosine >= 0.98."""
    g = golden("g20_gmm")
    m = build(case, "bf16")
    d = lambda k: T(g[f"{case}_{k}"]).cuda()
    cond = {"state": d("state")}
    lp, _, _ = m.get_logprobs(cond, d("actions"))
    assert float(np.abs(lp.cpu().numpy() - g[f"{case}_logprobs"]).max()) < 0.15
    res = m.loss(cond, d("actions"), d("returns"), d("oldvalues"), d("adv"), d("oldlogprobs"))
    ref = g[f"{case}_stats"]
    assert abs(res[2].item() - ref[2]) <= 5e-2 * abs(ref[2]) + 1e-3 and abs(res[5] - ref[5]) < 0.05 and np.isfinite(res[0].item())
    assert abs(res[7] - ref[7]) < 1e-3 and abs(float(res[1]) - ref[1]) < 2e-2
    (res[0] + 0.5 * res[2]).backward()
    num = a2 = b2 = 0.0
    for k, p in m.critic.named_parameters():
        x = p.grad.double().cpu().numpy().reshape(-1)
        key = f"{case}_gcritic_{k}"
        r, xs = (g[key].astype(np.float64).reshape(-1), x) if key in g else (g[key + "__sub"].astype(np.float64), x[::61])
        num, a2, b2 = num + float(xs @ r), a2 + float(xs @ xs), b2 + float(r @ r)
    assert num / np.sqrt(a2 * b2) > 0.98
