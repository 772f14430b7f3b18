"""Gaussian-policy PPO: the oracle against the reference's golden vectors (CPU), and the HIP path
(dppo_gaussian_* through the C ABI) against the same vectors on the GPU."""
import numpy as np
import pytest
import torch

from oracle import dppo_oracle as O
from tests.golden.make_golden_cases import GAUSS_CASES
from tests.test_oracle_golden import check_grad

T = torch.from_numpy


def logvar_of(a, kw, seed=73):
    if not kw["learn_fixed_std"]:
        return None
    rs = np.random.RandomState(seed)
    return T((np.log(kw["fixed_std"] ** 2) + rs.uniform(-0.4, 0.4, size=a.action_dim)).astype(np.float32))


def gcfg(kw):
    return O.GaussianCfg(**kw)


@pytest.mark.parametrize("case", sorted(GAUSS_CASES))
def test_oracle_matches_reference(golden, case):
    g = golden("g12_gaussian")
    sname, kw = GAUSS_CASES[case]
    a, c = O.named_specs(sname)
    gc = gcfg(kw)
    ft = {k: v.clone().requires_grad_(True) for k, v in O.init_params(a, 71).items()}
    cr = {k: v.clone().requires_grad_(True) for k, v in O.init_params(c, 72).items()}
    lv = logvar_of(a, kw)
    if lv is not None:
        lv.requires_grad_(True)
    state = T(g[f"{case}_state"])
    with torch.no_grad():
        act = O.gaussian_sample(gc, a, ft, lv, state, T(g[f"{case}_noise"]))
        det = O.gaussian_sample(gc, a, ft, lv, state, T(g[f"{case}_noise"]), deterministic=True)
        lp, ent, std = O.gaussian_logprob(gc, a, ft, lv, state, T(g[f"{case}_actions"]))
    np.testing.assert_allclose(act.numpy(), g[f"{case}_actions"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(det.numpy(), g[f"{case}_actions_det"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lp.numpy(), g[f"{case}_logprobs"], rtol=1e-4, atol=1e-4)
    assert float(ent) == pytest.approx(float(g[f"{case}_entropy"]), rel=1e-6) and float(std) == pytest.approx(float(g[f"{case}_std"]), rel=1e-6)
    res = O.gaussian_ppo_loss(gc, a, c, ft, lv, cr, state, T(g[f"{case}_actions"]), T(g[f"{case}_returns"]),
                              T(g[f"{case}_oldvalues"]), T(g[f"{case}_adv"]), T(g[f"{case}_oldlogprobs"]))
    got = np.array([res[0].item(), res[1].item(), res[2].item(), res[3], res[4], res[5], res[6], res[7]])
    np.testing.assert_allclose(got, g[f"{case}_stats"], rtol=2e-5, atol=2e-6)
    (res[0] + 0.01 * res[1] + 0.5 * res[2]).backward()
    for k, p in ft.items():
        check_grad(g, f"{case}_gactor_{k}", p.grad, rtol=1e-4, atol=1e-6)
    for k, p in cr.items():
        check_grad(g, f"{case}_gcritic_{k}", p.grad, rtol=1e-4, atol=1e-6)
    if lv is not None:
        check_grad(g, f"{case}_gactor_logvar", lv.grad, rtol=1e-4, atol=1e-7)


def build(case, prec, dev="cuda:0"):
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.common.mlp_gaussian import Gaussian_MLP
    from dppo_amd.model.rl.gaussian_ppo import PPO_Gaussian
    sname, kw = GAUSS_CASES[case]
    a, c = O.named_specs(sname)
    actor = Gaussian_MLP(action_dim=a.action_dim, horizon_steps=a.horizon_steps, cond_dim=a.cond_dim, mlp_dims=list(a.mlp_dims),
                         activation_type=a.activation, residual_style=True, fixed_std=kw["fixed_std"],
                         learn_fixed_std=kw["learn_fixed_std"], std_min=kw["std_min"], std_max=kw["std_max"], precision=prec)
    critic = CriticObs(cond_dim=c.cond_dim, mlp_dims=list(c.mlp_dims), activation_type=c.activation, residual_style=True,
                       precision=prec)
    sd = dict(O.init_params(a, 71))
    lv = logvar_of(a, kw)
    if lv is not None:
        sd["logvar"] = lv
    actor.load_state_dict(sd, strict=False)
    critic.load_state_dict(O.init_params(c, 72))
    m = PPO_Gaussian(actor=actor, critic=critic, horizon_steps=a.horizon_steps, device=dev,
                     clip_ploss_coef=kw["clip_ploss_coef"], clip_vloss_coef=kw.get("clip_vloss_coef"),
                     norm_adv=kw.get("norm_adv", True), randn_clip_value=kw["randn_clip_value"])
    return m, a, c


def test_state_dict_names_match_the_reference():
    m, a, c = build("gauss_furniture_learned", "fp32", dev="cpu")
    keys = set(m.state_dict())
    for pre in ("network.", "actor.", "actor_ft."):
        assert {pre + "logvar", pre + "logvar_min", pre + "logvar_max", pre + "mlp_mean.layers.0.weight",
                pre + "mlp_mean.layers.1.l1.weight", pre + "mlp_mean.layers.3.bias"} <= keys
    assert "critic.Q1.layers.0.weight" in keys
    assert m.actor_ft is m.network and m.actor is not m.actor_ft and not any(p.requires_grad for p in m.actor.parameters())
    # the flat kernel image covers the trunk only (logvar lives outside it)
    assert m.actor_ft.flat_params().numel() == sum(int(np.prod(s)) for _, s, _ in O.param_shapes(a))


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", sorted(GAUSS_CASES))
def test_hip_sampling_logprobs_loss_and_grads(golden, case, prec):
    from dppo_amd import hip
    g = golden("g12_gaussian")
    m, a, c = build(case, prec)
    dev = "cuda:0"
    f32 = prec == "fp32"
    state = T(g[f"{case}_state"]).to(dev)
    act = m(cond={"state": state}, noise=T(g[f"{case}_noise"]).to(dev))
    det = m(cond={"state": state}, deterministic=True, noise=T(g[f"{case}_noise"]).to(dev))
    tol = 2e-5 if f32 else 2e-2
    np.testing.assert_allclose(act.cpu().numpy(), g[f"{case}_actions"], rtol=tol, atol=tol)
    np.testing.assert_allclose(det.cpu().numpy(), g[f"{case}_actions_det"], rtol=tol, atol=tol)
    mean, scale = m.actor_ft({"state": state})
    assert tuple(mean.shape) == tuple(scale.shape) == (64, a.horizon_steps * a.action_dim)
    actions = T(g[f"{case}_actions"]).to(dev)
    lp, ent, std = m.get_logprobs({"state": state}, actions)
    # element log-probs amplify d mu by z / sigma; the chunk mean divides the error by sqrt(Ta Da)
    np.testing.assert_allclose(lp.cpu().numpy(), g[f"{case}_logprobs"], rtol=2e-4 if f32 else 0.3, atol=2e-4 if f32 else 0.3)
    assert float(ent) == pytest.approx(float(g[f"{case}_entropy"]), rel=1e-5) and float(std) == pytest.approx(float(g[f"{case}_std"]), rel=1e-5)
    # in-kernel noise: reproducible under torch.manual_seed, inside the clip range
    torch.manual_seed(3)
    s1 = m(cond={"state": state})
    torch.manual_seed(3)
    s2 = m(cond={"state": state})
    assert torch.equal(s1, s2) and not torch.equal(s1, m(cond={"state": state}))
    mu, sc = m.actor_ft({"state": state})
    assert ((s1.reshape(64, -1) - mu).abs() <= 3.0 * sc + 1e-6).all()
    if not f32:
        return  # the loss in bf16 is checked against the fp32 path below (the log-ratio error of bf16 exceeds the clip range)
    d = lambda k: T(g[f"{case}_{k}"]).to(dev)
    res = m.loss({"state": state}, actions, d("returns"), d("oldvalues"), d("adv"), d("oldlogprobs"))
    got = np.array([res[0].item(), res[1].item(), res[2].item(), res[3], res[4], res[5], float(res[6]), res[7]])
    np.testing.assert_allclose(got, g[f"{case}_stats"], rtol=2e-4, atol=2e-5)
    (res[0] + 0.01 * res[1] + 0.5 * res[2]).backward()
    for k, p in m.actor_ft.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, k
            check_grad(g, f"{case}_gactor_{k}", p.grad.cpu(), 2e-3, 2e-6)
    for k, p in m.critic.named_parameters():
        check_grad(g, f"{case}_gcritic_{k}", p.grad.cpu(), 2e-3, 2e-6)


@pytest.mark.gpu
def test_bf16_update_follows_fp32_and_pooled_shards_add_up():
    """bf16 loss statistics / gradient direction against the fp32 path on a 4,000-sample minibatch drawn by the policy
    itself, and the data-parallel contract: two shards evaluated with the pooled advantage moments sum to the whole."""
    from dppo_amd import hip
    dev = "cuda:0"
    out, args = {}, None
    for prec in ("fp32", "bf16"):
        m, a, c = build("gauss_furniture_learned", prec)
        N = 4000
        if args is None:  # the rollout comes from the fp32 policy; both precisions are updated on the same samples
            gen = torch.Generator().manual_seed(0)
            obs = (torch.rand(N, 1, a.cond_dim, generator=gen) * 2 - 1).to(dev)
            noise = torch.randn(N, a.horizon_steps * a.action_dim, generator=gen).to(dev)
            act = m(cond={"state": obs}, noise=noise)
            lp, _, _ = m.get_logprobs({"state": obs}, act)
            oldlp = lp + 0.01 * torch.randn(N, generator=gen).to(dev)
            val = m.critic({"state": obs}).reshape(-1)
            ret, adv = val + 0.5 * torch.randn(N, generator=gen).to(dev), torch.randn(N, generator=gen).to(dev)
            args = (obs.reshape(N, -1).contiguous(), act.reshape(N, -1).contiguous(), ret, val, adv, oldlp)
        st = m.ppo_update(*args).cpu().numpy().copy()
        ga, gc, gl = m.actor_ft.flat_grads().double().clone(), m.critic.flat_grads().double().clone(), m._lv_grad.double().clone()
        out[prec] = (st, ga.cpu().numpy(), gc.cpu().numpy())
        # shards
        am = adv.double()
        gm = torch.stack([am.sum(), (am * am).sum(), torch.tensor(float(N), dtype=torch.float64, device=dev)])
        sa, sc, sl, ss = torch.zeros_like(ga), torch.zeros_like(gc), torch.zeros_like(gl), np.zeros(5)
        for lo, hi in ((0, 1536), (1536, N)):
            sh = tuple(t[lo:hi].contiguous() for t in args)
            s2 = m.ppo_update(*sh, global_moments=gm).cpu().numpy()
            sa += m.actor_ft.flat_grads().double()
            sc += m.critic.flat_grads().double()
            sl += m._lv_grad.double()
            ss += s2[:5]  # each shard already divides by the global count
        tol = 2e-5 if prec == "fp32" else 2e-4
        assert (sa - ga).norm().item() <= tol * ga.norm().item() and (sc - gc).norm().item() <= tol * gc.norm().item()
        assert (sl - gl).norm().item() <= 1e-4 * gl.norm().item() + 1e-12
        np.testing.assert_allclose(ss, st[:5], rtol=1e-9, atol=1e-12)
    cos = lambda x, y: float(np.dot(x, y) / (np.linalg.norm(x) * np.linalg.norm(y) + 1e-30))
    (s32, a32, c32), (s16, a16, c16) = out["fp32"], out["bf16"]
    assert s16[hip.STAT_V_LOSS] == pytest.approx(s32[hip.STAT_V_LOSS], rel=5e-2)
    assert cos(a16, a32) >= 0.9 and cos(c16, c32) >= 0.99  # clip 0.01 vs the bf16 log-ratio error: direction only
