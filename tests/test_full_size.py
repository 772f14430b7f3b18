"""GPU parity at BASELINE.json's full sizes (configs[1]: hopper, n_envs = 512, K = 20, Kft = 10, minibatch N = 50,000).

The golden vectors and most oracle comparisons run at sizes the CPU restatement finishes in a blink; here the same HIP
entry points are driven at the benchmark's sizes and checked (a) against the oracle where it finishes in seconds (one
50,000-sample minibatch: ~2 s of torch CPU work) and (b) through size-independent properties of the domain:

* sampler -> log-prob round trip: a chain drawn with recorded noise z has log N(x_{k+1}; mu_k, sigma_k) =
  -z^2/2 - log sigma_k - log sqrt(2 pi) element by element (sampling and evaluation run in two different kernels);
* additivity of the update over shards of the minibatch under pooled advantage moments (what data parallelism relies on);
* GAE: closed form without terminations at gamma = lambda = 1, and linearity in (reward, values);
* AdamW on the actor's 553,020-parameter vector against torch.optim.AdamW on the same device.

Tolerances: fp32 path 1e-4-class (exact fp32 products, other summation order); bf16 path as in test_hip_parity.py.
"""
import math

import numpy as np
import pytest
import torch

from oracle import dppo_oracle as O
from tests.test_hip_parity import DEV, build_model, flat_of
from tests.test_oracle_golden import make_cfg

pytestmark = pytest.mark.gpu

N_ENVS, K, KFT, N_MB = 512, 20, 10, 50_000
KW = dict(denoising_steps=K, ft_denoising_steps=KFT, clip_ploss_coef=0.01, clip_ploss_coef_base=0.001,
          randn_clip_value=3, gamma_denoising=0.99)
cos = lambda x, y: float(np.dot(x, y) / (np.linalg.norm(x) * np.linalg.norm(y) + 1e-30))


def rollout(m, a, n_steps, seed):
    """n_steps sampling calls at n_envs = 512 -> a rollout buffer of R = n_steps * 512 rows, like the agent's."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    torch.manual_seed(seed)
    AF = a.horizon_steps * a.action_dim
    obs = (torch.rand(n_steps, N_ENVS, 1, a.cond_dim, generator=gen) * 2 - 1).to(DEV)
    chains = torch.stack([m(cond={"state": obs[s]}, deterministic=False, return_chain=True).chains for s in range(n_steps)])
    R = n_steps * N_ENVS
    obs, chains = obs.reshape(R, 1, -1), chains.reshape(R, KFT + 1, a.horizon_steps, a.action_dim)
    logp = m.get_logprobs({"state": obs}, chains).reshape(R, KFT, AF)
    logp = logp + 0.003 * torch.randn(logp.shape, generator=gen).to(DEV)  # ratio != 1: both surrogate branches taken
    values = m.critic({"state": obs}).reshape(R)
    returns = values + torch.randn(R, generator=gen).to(DEV) * 0.5
    adv = torch.randn(R, generator=gen).to(DEV) * 2 + 0.3
    inds = torch.randperm(R * KFT, generator=gen)[:N_MB].to(DEV).contiguous()
    return obs.reshape(R, -1).contiguous(), chains.reshape(R, KFT + 1, AF).contiguous(), returns, values, adv, logp, inds


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_sampler_logprob_round_trip_at_n_envs_512(prec):
    m, a, _ = build_model("hopper", KW, 61, prec)
    gen = torch.Generator(device="cpu").manual_seed(5)
    obs = (torch.rand(N_ENVS, 1, a.cond_dim, generator=gen) * 2 - 1).to(DEV)
    noise = torch.randn(K + 1, N_ENVS, a.horizon_steps, a.action_dim, generator=gen).to(DEV)
    smp = m(cond={"state": obs}, deterministic=False, return_chain=True, noise=noise)
    assert tuple(smp.chains.shape) == (N_ENVS, KFT + 1, a.horizon_steps, a.action_dim)
    assert torch.isfinite(smp.chains).all() and torch.equal(smp.chains[:, -1], smp.trajectories)
    again = m(cond={"state": obs}, deterministic=False, return_chain=True, noise=noise)
    assert torch.equal(again.chains, smp.chains)  # same inputs, same bits
    lp = m.get_logprobs({"state": obs}, smp.chains).reshape(N_ENVS, KFT, -1).double()
    z = noise[K - KFT + 1:].clamp(-3, 3).reshape(KFT, N_ENVS, -1).permute(1, 0, 2).double()  # the draw of fine-tuned step j
    const = lp + 0.5 * z * z  # = -log sigma_j - log sqrt(2 pi), the same for every env and action dimension of step j
    ref = const.median(dim=2).values.median(dim=0).values  # (Kft,)
    assert (ref <= -math.log(0.1) - 0.5 * math.log(2 * math.pi) + 1e-4).all()  # sigma_j >= min_logprob_denoising_std
    err = (const - ref[None, :, None]).abs()
    if prec == "fp32":
        assert err.max().item() <= 2e-3  # |z| <= 3 times the relative error of (x' - mu) / sigma at sigma >= 0.1
    else:
        assert err.mean().item() <= 0.06 and err.max().item() <= 1.5  # bf16 eps in two different kernels (see header)


def test_minibatch_of_50000_matches_oracle_fp32():
    from dppo_amd import hip
    m, a, c = build_model("hopper", KW, 41, "fp32")
    obs, chains, returns, values, adv, logp, inds = rollout(m, a, 12, 7)
    stats = m.ppo_update(obs, chains, returns, values, adv, logp, inds, reward_horizon=4).cpu().numpy()
    ga, gc = m.actor_ft.flat_grads().cpu().numpy().copy(), m.critic.flat_grads().cpu().numpy().copy()
    cfg = make_cfg(a, KW)
    base, ft, cr = O.init_params(a, 41), O.init_params(a, 42), O.init_params(c, 43)
    for p in list(ft.values()) + list(cr.values()):
        p.requires_grad_(True)
    R = obs.shape[0]
    b, k = (inds // KFT).cpu(), (inds % KFT).cpu()
    ch = chains.cpu().reshape(R, KFT + 1, a.horizon_steps, a.action_dim)
    res = O.ppo_loss(cfg, a, c, base, ft, cr, obs.cpu().reshape(R, 1, -1)[b], ch[b, k], ch[b, k + 1], k, returns.cpu()[b],
                     values.cpu()[b], adv.cpu()[b], logp.cpu().reshape(R, KFT, a.horizon_steps, a.action_dim)[b, k])
    (res[0] + res[2]).backward()
    ref_a, ref_c = flat_of({n: p.grad for n, p in ft.items()}, a), flat_of({n: p.grad for n, p in cr.items()}, c)
    assert stats[hip.STAT_PG_LOSS] == pytest.approx(res[0].item(), rel=1e-3, abs=1e-5)
    assert stats[hip.STAT_V_LOSS] == pytest.approx(res[2].item(), rel=2e-4)
    assert stats[hip.STAT_APPROX_KL] == pytest.approx(float(res[4]), rel=1e-2, abs=1e-7)
    assert stats[hip.STAT_CLIPFRAC] == pytest.approx(float(res[3]), abs=2e-3)  # a flipped branch needs |ratio - bound| < 1e-6
    assert stats[hip.STAT_RATIO] == pytest.approx(float(res[5]), rel=1e-5)
    assert 0.01 < stats[hip.STAT_CLIPFRAC] < 0.95  # the inputs exercise both branches of the clipped surrogate
    assert cos(ga, ref_a) >= 0.999 and cos(gc, ref_c) >= 0.9999
    assert np.linalg.norm(ga) == pytest.approx(np.linalg.norm(ref_a), rel=2e-3)
    assert np.linalg.norm(gc) == pytest.approx(np.linalg.norm(ref_c), rel=1e-3)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_minibatch_of_50000_is_the_sum_of_its_shards(prec):
    """Gradients and statistics of one 50,000-sample minibatch == the sum over two shards evaluated with the pooled
    advantage moments (every mean is scaled by the global count): the data-parallel contract, at the benchmark's size."""
    m, a, _ = build_model("hopper", KW, 43, prec)
    obs, chains, returns, values, adv, logp, inds = rollout(m, a, 12, 9)
    full = m.ppo_update(obs, chains, returns, values, adv, logp, inds, reward_horizon=4).clone()
    ga, gc = m.actor_ft.flat_grads().double().clone(), m.critic.flat_grads().double().clone()
    am = adv[inds // KFT].double()
    gm = torch.stack([am.sum(), (am * am).sum(), torch.tensor(float(N_MB), dtype=torch.float64, device=DEV)])
    sa, sc, ss = torch.zeros_like(ga), torch.zeros_like(gc), torch.zeros_like(full)
    cut = 20_032  # uneven shards, the second not a multiple of any tile height
    for shard in (inds[:cut].contiguous(), inds[cut:].contiguous()):
        st = m.ppo_update(obs, chains, returns, values, adv, logp, shard, reward_horizon=4, global_moments=gm)
        sa += m.actor_ft.flat_grads().double()
        sc += m.critic.flat_grads().double()
        ss += st
    tol = 2e-5 if prec == "fp32" else 2e-4  # same per-sample arithmetic; only the fp32 reduction order differs
    assert (sa - ga).norm().item() <= tol * ga.norm().item()
    assert (sc - gc).norm().item() <= tol * gc.norm().item()
    assert torch.allclose(ss[:5], full[:5], rtol=1e-9, atol=1e-12)  # pg, v, kl, clipfrac, ratio: float64 sums of the same terms


def test_gae_closed_form_and_linearity_at_500_by_512():
    from dppo_amd.util.rollout import gae_device
    S, E = 500, N_ENVS
    gen = torch.Generator(device="cpu").manual_seed(3)
    r = torch.randn(S, E, generator=gen, dtype=torch.float64).to(DEV)
    v = torch.randn(S, E, generator=gen).to(DEV)
    last = torch.randn(E, generator=gen).to(DEV)
    zero = torch.zeros(S, E, device=DEV)
    a64, r64, a32, r32 = gae_device(r, v, zero, last, 1.0, 1.0, 1.0)
    closed = torch.flip(torch.cumsum(torch.flip(r, [0]), 0), [0]) + last.double()[None] - v.double()
    assert torch.allclose(a64, closed, rtol=1e-11, atol=1e-9)
    assert torch.allclose(r64, a64 + v.double(), rtol=0, atol=1e-12)
    assert torch.equal(a32, a64.float()) and torch.equal(r32, r64.float())
    # with terminations: linear in (reward, values, last value) for a fixed termination pattern
    term = (torch.rand(S, E, generator=gen) < 0.002).float().to(DEV)
    r2 = torch.randn(S, E, generator=gen, dtype=torch.float64).to(DEV)
    v2, last2 = torch.randn(S, E, generator=gen).to(DEV), torch.randn(E, generator=gen).to(DEV)
    A = gae_device(r, v, term, last, 0.99, 0.95, 0.7)[0]
    B = gae_device(r2, v2, term, last2, 0.99, 0.95, 0.7)[0]
    AB = gae_device(r + r2, v + v2, term, last + last2, 0.99, 0.95, 0.7)[0]
    assert torch.allclose(AB, A + B, rtol=1e-6, atol=1e-5)  # v + v2 is rounded to fp32 before the scan
    # an env that terminates at step t does not see anything after t
    t_cut = 250
    term2 = torch.zeros(S, E, device=DEV)
    term2[t_cut] = 1
    C1 = gae_device(r, v, term2, last, 0.99, 0.95, 1.0)[0]
    r3 = r.clone()
    r3[t_cut + 1:] += 5.0
    C2 = gae_device(r3, v, term2, last + 1.0, 0.99, 0.95, 1.0)[0]
    assert torch.equal(C1[:t_cut + 1], C2[:t_cut + 1])


def test_adamw_on_the_actor_vector_matches_torch_on_device():
    from dppo_amd.util.optim import FlatAdamW, step_many
    n = 553_020  # actor_ft of the benchmark's network
    gen = torch.Generator(device="cpu").manual_seed(11)
    p0 = (torch.randn(n, generator=gen) * 0.05).to(DEV)
    ours, ref = p0.clone(), torch.nn.Parameter(p0.clone())
    opt = FlatAdamW(ours, lr=1e-4, weight_decay=1e-2)
    topt = torch.optim.AdamW([ref], lr=1e-4, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8)
    for it in range(5):
        g = (torch.randn(n, generator=gen) * (0.1 + it)).to(DEV)
        step_many([opt.slot(g)])
        ref.grad = g.clone()
        topt.step()
        assert torch.allclose(ours, ref.detach(), rtol=1e-6, atol=1e-8)
