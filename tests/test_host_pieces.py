"""The PRODUCT's host-side pieces against the reference's golden vectors (CPU; no HIP call is made): schedule tables
(A1), EtaFixed (A14), RunningRewardScaler (A13), CosineAnnealingWarmupRestarts, VPGDiffusion.step annealing (A15) and the
per-step coefficient tables the sampler / log-prob kernels consume (A6).  The oracle's twins of the first three are
pinned in test_oracle_golden.py; these tests pin what ships."""
import copy

import numpy as np
import pytest
import torch

from oracle import dppo_oracle as O


def product_model(use_ddim=False, **kw):
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    from dppo_amd.model.diffusion.eta import EtaFixed
    from dppo_amd.model.diffusion.mlp_diffusion import DiffusionMLP
    actor = DiffusionMLP(3, 4, 11, mlp_dims=[512, 512, 512], activation_type="ReLU", residual_style=True)
    critic = CriticObs(11, [256, 256, 256], residual_style=True)
    if use_ddim:
        kw = dict(kw, use_ddim=True, eta=EtaFixed(base_eta=kw.pop("base_eta", 1.0)))
    return PPODiffusion(actor=actor, critic=critic, horizon_steps=4, obs_dim=11, action_dim=3, device="cpu",
                        gamma_denoising=0.99, clip_ploss_coef=0.01, **kw)


def test_product_schedule_tables_are_bit_identical_to_the_reference(golden):
    g = golden("g1_tables")
    for K in (20, 100):
        m = product_model(denoising_steps=K, ft_denoising_steps=10)
        for k in ("betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "ddpm_var", "ddpm_logvar_clipped", "ddpm_mu_coef1", "ddpm_mu_coef2"):
            assert np.array_equal(getattr(m, k).numpy(), g[f"K{K}_{k}"]), (K, k)
    m = product_model(use_ddim=True, denoising_steps=100, ft_denoising_steps=5, ddim_steps=5)
    for k in ("ddim_t", "ddim_alphas", "ddim_alphas_sqrt", "ddim_alphas_prev", "ddim_sqrt_one_minus_alphas"):
        assert np.array_equal(getattr(m, k).numpy(), g[f"ddim100_5_{k}"]), k


def test_product_eta_fixed_matches_the_reference_and_is_cached(golden):
    from dppo_amd.model.diffusion.eta import EtaFixed
    g = golden("g1_tables")
    e1, e05 = EtaFixed(base_eta=1.0), EtaFixed(base_eta=0.5)
    assert e1.value() == float(g["eta_fixed_base1"][0]) and e05.value() == float(g["eta_fixed_base05"][0])
    out = e05({"state": torch.zeros(3, 1, 11)})
    assert tuple(out.shape) == (3, 1) and float(out[0, 0]) == pytest.approx(float(g["eta_fixed_base05"][0]), abs=1e-7)
    # the cached scalar follows a write of the parameter (load_state_dict), and only that
    v = e05.value()
    assert e05.value() is v or e05.value() == v
    e05.load_state_dict(e1.state_dict())
    assert e05.value() == e1.value()


def test_product_reward_scaler_matches_the_reference(golden):
    from dppo_amd.util.reward_scaling import RunningRewardScaler
    g = golden("g6_reward_scaler")
    sc = RunningRewardScaler(4)
    for it in range(3):
        out = sc(reward=g[f"it{it}_reward"], first=g[f"it{it}_first"])
        np.testing.assert_array_equal(out, g[f"it{it}_scaled"])
        assert float(sc.ret_rms.var) == float(g[f"it{it}_var"])


def test_reward_scaler_pooled_over_two_shards_equals_one_scaler_over_all_envs(golden):
    """The data-parallel hook: two scalers over env shards whose batch moments are pooled == one scaler over all envs."""
    from dppo_amd.util.reward_scaling import RunningRewardScaler
    g = golden("g6_reward_scaler")
    whole = RunningRewardScaler(4)
    shards = []

    def pooled(mean, var, cnt):  # what TrainPPODiffusionAgent._pool_return_moments does with an all-reduce
        n = sum(c for _, _, c in pending)
        s1 = sum(c * m for m, _, c in pending)
        s2 = sum(c * (v + m * m) for m, v, c in pending)
        return s1 / n, s2 / n - (s1 / n) ** 2, n

    for it in range(3):
        r, f = g[f"it{it}_reward"], g[f"it{it}_first"]
        want = whole(reward=r, first=f)
        if not shards:
            shards = [RunningRewardScaler(2, moments_hook=pooled) for _ in range(2)]
        # first pass: collect each shard's batch moments (the collective), second pass: the update itself
        from dppo_amd.util.reward_scaling import backward_discounted_sum
        pending = []
        for i, s in enumerate(shards):
            rets = backward_discounted_sum(s.ret, r[2 * i:2 * i + 2], f[2 * i:2 * i + 2], s.gamma).reshape(-1)
            pending.append((np.mean(rets), np.var(rets), rets.shape[0]))
        got = np.concatenate([s(reward=r[2 * i:2 * i + 2], first=f[2 * i:2 * i + 2]) for i, s in enumerate(shards)])
        np.testing.assert_allclose(got, want, rtol=1e-12)


def test_product_lr_scheduler_matches_the_reference_trace(golden):
    from dppo_amd.util.scheduler import CosineAnnealingWarmupRestarts
    from tests.golden.make_golden_cases import SCHED_CASES
    g = golden("g10_scheduler")

    class Opt:  # FlatAdamW's scheduler-facing surface
        def __init__(self, lr):
            self.param_groups = [{"lr": lr}]

    for name, (kw, n, lr0) in SCHED_CASES.items():
        opt = Opt(lr0)
        sch = CosineAnnealingWarmupRestarts(opt, **kw)
        trace = [opt.param_groups[0]["lr"]]
        for _ in range(n):
            sch.step()
            trace.append(opt.param_groups[0]["lr"])
        np.testing.assert_allclose(trace, g[name], rtol=1e-14, atol=0, err_msg=name)


def test_step_anneals_the_number_of_fine_tuned_steps_like_the_reference():
    """VPGDiffusion.step (reference diffusion_vpg.py:102-127): every ft_denoising_steps_t calls, ft_denoising_steps drops by
    ft_denoising_steps_d, the fine-tuned net becomes the new frozen base and a fresh trainable copy is made."""
    m = product_model(denoising_steps=20, ft_denoising_steps=10, ft_denoising_steps_d=3, ft_denoising_steps_t=2)
    with torch.no_grad():
        for p in m.actor_ft.parameters():
            p.add_(1.0)
    ft0 = copy.deepcopy(m.actor_ft.state_dict())
    m.step()
    assert m.ft_denoising_steps == 10 and m.ft_denoising_steps_cnt == 1
    old_ft = m.actor_ft
    m.step()
    assert m.ft_denoising_steps == 7
    assert m.actor is old_ft and m.actor_ft is not old_ft
    assert all(not p.requires_grad for p in m.actor.parameters()) and all(p.requires_grad for p in m.actor_ft.parameters())
    for k, v in m.actor_ft.state_dict().items():
        assert torch.equal(v, ft0[k]) and torch.equal(m.actor.state_dict()[k], ft0[k])
    for _ in range(6):
        m.step()
    assert m.ft_denoising_steps == 0  # max(0, ...)
    assert m.get_min_sampling_denoising_std() == 0.1


def test_sampling_and_logprob_step_tables_follow_the_reference_formulas():
    """The dppo_step tables (what the kernels consume) against the oracle's p_mean_var coefficients: DDPM K = 20 and
    DDIM 100 / 5, stochastic and deterministic."""
    from dppo_amd import hip
    for use_ddim in (False, True):
        kw = dict(denoising_steps=100, ft_denoising_steps=5, ddim_steps=5) if use_ddim else dict(
            denoising_steps=20, ft_denoising_steps=10)
        m = product_model(use_ddim=use_ddim, min_sampling_denoising_std=0.08, **kw)
        for det in (False, True):
            tab, n_steps, chain_len, init_slot = m._sampling_schedule(det, False, "cpu")
            rec = np.frombuffer(tab.numpy().tobytes(), dtype=hip.STEP_DTYPE)
            assert n_steps == (5 if use_ddim else 20) and chain_len == m.ft_denoising_steps + 1
            assert [int(r["net"]) for r in rec] == ([1] * 5 if use_ddim else [0] * 10 + [1] * 10)
            if not use_ddim:
                t = O.ddpm_tables(20)
                ts = list(reversed(range(20)))
                assert [int(r["t"]) for r in rec] == ts
                for r, tt in zip(rec, ts):
                    assert r["c0"] == t["sqrt_recip_alphas_cumprod"][tt].item()
                    assert r["c2"] == t["ddpm_mu_coef1"][tt].item() and r["c3"] == t["ddpm_mu_coef2"][tt].item()
                    std = float(torch.exp(0.5 * t["ddpm_logvar_clipped"][tt]))
                    want = (0.0 if tt == 0 else max(std, 1e-3)) if det else max(std, 0.08)
                    assert r["std"] == pytest.approx(want, rel=1e-7)
            else:
                assert [int(r["t"]) for r in rec] == [80, 60, 40, 20, 0]
                assert all(r["std"] == 0.0 for r in rec) if det else all(r["std"] >= np.float32(0.08) for r in rec)
        lp = np.frombuffer(m._logprob_schedule("cpu").numpy().tobytes(), dtype=hip.STEP_DTYPE)
        assert len(lp) == m.ft_denoising_steps and all(r["std"] >= np.float32(0.1) for r in lp)  # min_logprob_denoising_std
        assert [int(r["t"]) for r in lp] == ([80, 60, 40, 20, 0] if use_ddim else list(reversed(range(10))))


def test_split_sampler_block_id_mapping_is_a_bijection_with_tiles_on_one_residue_class():
    """csrc/sampler_split.hip maps block id b -> (tile, member) = ((b & 7) + 8 * (b >> 6), (b >> 3) & 7) on a grid padded to
    64 * ceil(tiles / 8) blocks (restated here): every (tile < tiles, member < 8) pair must be hit exactly once, all eight
    members of a tile must share b mod 8 (= one XCD under the observed round-robin placement) and be consecutive within
    that residue class (= resident together under in-order placement), and blocks of tiles >= `tiles` must be the ones that
    exit."""
    for tiles in range(1, 33):
        grid = (tiles + 7) // 8 * 64
        seen = {}
        for b in range(grid):
            tile, m = (b & 7) + 8 * (b >> 6), (b >> 3) & 7
            if tile < tiles:
                assert (tile, m) not in seen
                seen[(tile, m)] = b
        assert len(seen) == tiles * 8
        for t in range(tiles):
            ids = [seen[(t, m)] for m in range(8)]
            assert len({b & 7 for b in ids}) == 1
            ranks = sorted(b >> 3 for b in ids)
            assert ranks == list(range(ranks[0], ranks[0] + 8))
