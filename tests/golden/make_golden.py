"""Generate golden fixtures by running the REFERENCE (read-only, /root/reference) on CPU.

Run once in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Only data (inputs + expected outputs) is written, as small .npz files next to this script;
weights are never stored -- both sides rebuild them from the seeded recipe
``oracle.dppo_oracle.init_params``.  The reference cannot travel to the GPU box; these
fixtures can.
"""
import os
import sys
from contextlib import contextmanager

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from oracle import dppo_oracle as O  # noqa: E402  (seeded weight recipe + specs only)

from dppo.model.common.critic import CriticObs  # noqa: E402
from dppo.model.diffusion.diffusion_ppo import PPODiffusion  # noqa: E402
from dppo.model.diffusion.eta import EtaFixed  # noqa: E402
from dppo.model.diffusion.mlp_diffusion import DiffusionMLP  # noqa: E402
from dppo.util.reward_scaling import RunningRewardScaler  # noqa: E402

torch.set_num_threads(4)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"wrote {name}.npz: {len(out)} arrays, {sum(v.nbytes for v in out.values())} bytes raw")


GRAD_STRIDE = 61  # big gradient tensors are stored as flat[::61] plus their 2-norm and sum


def put_grad(out, key, g):
    g = g.detach().cpu().numpy()
    if g.size > 4096:
        out[key + "__sub"] = g.reshape(-1)[::GRAD_STRIDE].copy()
        out[key + "__norm"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        out[key + "__sum"] = np.float64(g.astype(np.float64).sum())
    else:
        out[key] = g


def ref_actor(spec: O.NetSpec, params):
    m = DiffusionMLP(action_dim=spec.action_dim, horizon_steps=spec.horizon_steps, cond_dim=spec.cond_dim,
                     time_dim=spec.time_dim, mlp_dims=list(spec.mlp_dims), cond_mlp_dims=spec.cond_mlp_dims,
                     activation_type=spec.activation, use_layernorm=spec.use_layernorm,
                     residual_style=spec.residual)
    m.load_state_dict(params, strict=True)
    return m


def ref_critic(spec: O.NetSpec, params):
    m = CriticObs(cond_dim=spec.cond_dim, mlp_dims=list(spec.mlp_dims), activation_type=spec.activation,
                  use_layernorm=spec.use_layernorm, residual_style=spec.residual)
    m.load_state_dict(params, strict=True)
    return m


def ref_model(aspec, cspec, seed, **kw):
    base = O.init_params(aspec, seed)
    ft = O.init_params(aspec, seed + 1)
    cr = O.init_params(cspec, seed + 2)
    model = PPODiffusion(actor=ref_actor(aspec, base), critic=ref_critic(cspec, cr),
                         horizon_steps=aspec.horizon_steps, obs_dim=aspec.cond_dim,
                         action_dim=aspec.action_dim, device="cpu", **kw)
    model.actor_ft.load_state_dict(ft, strict=True)
    return model


@contextmanager
def recorded_noise(noise):
    """Feed pre-drawn N(0,1) tensors to torch.randn / randn_like, in call order."""
    it = iter(noise)
    r0, r1 = torch.randn, torch.randn_like
    torch.randn = lambda *a, **k: next(it).clone()
    torch.randn_like = lambda *a, **k: next(it).clone()
    try:
        yield
    finally:
        torch.randn, torch.randn_like = r0, r1


specs = O.named_specs


# ---------------------------------------------------------------- G1 tables
def g1_tables():
    out = {}
    for K in (20, 100):
        a, c = specs("hopper")
        m = ref_model(a, c, 7, gamma_denoising=0.99, clip_ploss_coef=0.01, ft_denoising_steps=10,
                      denoising_steps=K)
        for k in ("betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "ddpm_var", "ddpm_logvar_clipped", "ddpm_mu_coef1",
                  "ddpm_mu_coef2"):
            out[f"K{K}_{k}"] = getattr(m, k)
    a, c = specs("hopper")
    m = ref_model(a, c, 7, gamma_denoising=0.99, clip_ploss_coef=0.01, ft_denoising_steps=5,
                  denoising_steps=100, use_ddim=True, ddim_steps=5, eta=EtaFixed(base_eta=1.0))
    for k in ("ddim_t", "ddim_alphas", "ddim_alphas_sqrt", "ddim_alphas_prev", "ddim_sqrt_one_minus_alphas"):
        out[f"ddim100_5_{k}"] = getattr(m, k)
    out["eta_fixed_base1"] = m.eta({"state": torch.zeros(1, 1, 11)}).reshape(-1)
    out["eta_fixed_base05"] = EtaFixed(base_eta=0.5)({"state": torch.zeros(1, 1, 11)}).reshape(-1)
    save("g1_tables", **out)


# ---------------------------------------------------------------- G2 network forwards
def g2_forward():
    rs = np.random.RandomState(100)
    out = {}
    for name in ("hopper", "can", "halfcheetah", "furniture_like", "plain_mlp", "kitchen_like", "square_like", "furniture_256", "ln_relu",
                 "transport", "furniture_one_leg", "can_relu"):
        a, c = specs(name)
        B = 8
        pa, pc = O.init_params(a, 11), O.init_params(c, 12)
        x = torch.from_numpy(rs.randn(B, a.horizon_steps, a.action_dim).astype(np.float32))
        t = torch.from_numpy(rs.randint(0, 20, size=(B,)).astype(np.int64))
        s = torch.from_numpy(rs.uniform(-1, 1, size=(B, 1, a.cond_dim)).astype(np.float32))
        with torch.no_grad():
            y = ref_actor(a, pa)(x, t, cond={"state": s})
            v = ref_critic(c, pc)({"state": s})
        out.update({f"{name}_x": x, f"{name}_t": t, f"{name}_state": s, f"{name}_eps": y, f"{name}_value": v})
    save("g2_forward", **out)


# ---------------------------------------------------------------- G3/G4 sampling chains + log-probs
def g3_g4_chains():
    cases = {
        # name: (spec, B, model kwargs, deterministic)
        "ddpm20_ft10": ("hopper", 6, dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), False),
        "ddpm20_ft10_det": ("hopper", 6, dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), True),
        "ddpm20_ft20": ("hopper", 4, dict(denoising_steps=20, ft_denoising_steps=20, randn_clip_value=3,
                                          final_action_clip_value=1.0), False),
        "ddpm100_can": ("can", 4, dict(denoising_steps=100, ft_denoising_steps=10, randn_clip_value=3,
                                       min_sampling_denoising_std=0.08), False),
        "ddim100_5": ("hopper", 5, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                        randn_clip_value=3, eps_clip_value=2.0,
                                        min_sampling_denoising_std=0.1), False),
        "ddim100_5_det": ("hopper", 5, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True,
                                            ddim_steps=5, randn_clip_value=3), True),
        "ddim100_10_ft4": ("halfcheetah", 3, dict(denoising_steps=100, ft_denoising_steps=4, use_ddim=True,
                                                  ddim_steps=10, randn_clip_value=3), False),
        "furniture_like": ("furniture_like", 4, dict(denoising_steps=20, ft_denoising_steps=5,
                                                     randn_clip_value=3), False),
        "kitchen_like": ("kitchen_like", 5, dict(denoising_steps=20, ft_denoising_steps=10,
                                                 randn_clip_value=3), False),
        "square_like": ("square_like", 3, dict(denoising_steps=20, ft_denoising_steps=10,
                                               randn_clip_value=3), False),
        "furniture_256": ("furniture_256", 4, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True,
                                                   ddim_steps=5, randn_clip_value=3), False),
        "ln_relu": ("ln_relu", 5, dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), False),
        "transport": ("transport", 3, dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3,
                                           min_sampling_denoising_std=0.1, min_logprob_denoising_std=0.1), False),
        "furniture_one_leg": ("furniture_one_leg", 3, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True,
                                                           ddim_steps=5, randn_clip_value=3,
                                                           min_sampling_denoising_std=0.04), False),
        # appended in round 2 (the RNG stream of the cases above is unchanged)
        "ddpm100_can_relu": ("can_relu", 4, dict(denoising_steps=100, ft_denoising_steps=10, randn_clip_value=3,
                                                 min_sampling_denoising_std=0.08), False),
        "ddpm20_halfcheetah": ("halfcheetah", 6, dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3),
                               False),  # BASELINE C4 as shipped: cfg/gym/finetune/halfcheetah-v2/ft_ppo_diffusion_mlp.yaml:16-22
    }
    out = {}
    rs = np.random.RandomState(200)
    for cname, (sname, B, kw, det) in cases.items():
        a, c = specs(sname)
        if kw.get("use_ddim"):
            kw = dict(kw, eta=EtaFixed(base_eta=1.0))
        m = ref_model(a, c, 21, gamma_denoising=0.99, clip_ploss_coef=0.01, **kw)
        n_steps = kw["ddim_steps"] if kw.get("use_ddim") else kw["denoising_steps"]
        state = torch.from_numpy(rs.uniform(-1, 1, size=(B, 1, a.cond_dim)).astype(np.float32))
        noise = torch.from_numpy(rs.randn(n_steps + 1, B, a.horizon_steps, a.action_dim).astype(np.float32))
        with recorded_noise(list(noise)):
            smp = m(cond={"state": state}, deterministic=det, return_chain=True)
        with torch.no_grad():
            lp = m.get_logprobs({"state": state}, smp.chains)
        out.update({f"{cname}_state": state, f"{cname}_noise": noise, f"{cname}_traj": smp.trajectories,
                    f"{cname}_chains": smp.chains, f"{cname}_logprobs": lp})
    save("g3_chains", **out)


# ---------------------------------------------------------------- G5 PPO loss + grads
def g5_loss():
    cases = {
        "default": ("hopper", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01,
                                   clip_ploss_coef_base=0.001), 4),
        "vclip_nonorm": ("hopper", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.1,
                                        clip_ploss_coef_base=0.01, clip_vloss_coef=0.2, norm_adv=False), 4),
        "quantile_rh2": ("hopper", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01,
                                        clip_advantage_lower_quantile=0.05, clip_advantage_upper_quantile=0.95), 2),
        "can_k100": ("can", dict(denoising_steps=100, ft_denoising_steps=10, clip_ploss_coef=0.01), 4),
        "ddim": ("hopper", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                clip_ploss_coef=0.01), 4),
        "furniture_like": ("furniture_like", dict(denoising_steps=20, ft_denoising_steps=5, clip_ploss_coef=0.01), 4),
        "kitchen_like": ("kitchen_like", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01), 4),
        "square_like": ("square_like", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01), 4),
        "furniture_256": ("furniture_256", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                                clip_ploss_coef=0.001), 4),
        "ln_relu": ("ln_relu", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01), 4),
        "transport": ("transport", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01,
                                        clip_ploss_coef_base=0.001, min_sampling_denoising_std=0.1,
                                        min_logprob_denoising_std=0.1), 8),
        "furniture_one_leg": ("furniture_one_leg", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True,
                                                        ddim_steps=5, clip_ploss_coef=0.001, clip_ploss_coef_base=0.001,
                                                        min_sampling_denoising_std=0.04), 8),
        # appended in round 2
        "can_relu_k100": ("can_relu", dict(denoising_steps=100, ft_denoising_steps=10, clip_ploss_coef=0.01), 4),
        "halfcheetah": ("halfcheetah", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01,
                                            clip_ploss_coef_base=0.01), 4),  # C4 as shipped (:77-80)
    }
    out = {}
    rs = np.random.RandomState(300)
    N = 64
    for cname, (sname, kw, rh) in cases.items():
        a, c = specs(sname)
        if kw.get("use_ddim"):
            kw = dict(kw, eta=EtaFixed(base_eta=1.0))
        m = ref_model(a, c, 31, gamma_denoising=0.99, randn_clip_value=3, **kw)
        Kft = kw["ft_denoising_steps"]
        n_steps = kw["ddim_steps"] if kw.get("use_ddim") else kw["denoising_steps"]
        # a realistic rollout: chains sampled by the policy itself, old log-probs from slightly different weights
        state = torch.from_numpy(rs.uniform(-1, 1, size=(N, 1, a.cond_dim)).astype(np.float32))
        noise = torch.from_numpy(rs.randn(n_steps + 1, N, a.horizon_steps, a.action_dim).astype(np.float32))
        with recorded_noise(list(noise)):
            chains = m(cond={"state": state}, deterministic=False, return_chain=True).chains
        kinds = torch.from_numpy(rs.randint(0, Kft, size=(N,)).astype(np.int64))
        rows = torch.arange(N)
        prev, nxt = chains[rows, kinds], chains[rows, kinds + 1]
        with torch.no_grad():
            oldlp_all = m.get_logprobs({"state": state}, chains).reshape(N, Kft, a.horizon_steps, a.action_dim)
            oldlp = oldlp_all[rows, kinds] + torch.from_numpy(
                rs.normal(0, 0.02, size=(N, a.horizon_steps, a.action_dim)).astype(np.float32))
            oldv = m.critic({"state": state}).view(-1) + torch.from_numpy(rs.normal(0, 0.3, N).astype(np.float32))
        ret = torch.from_numpy(rs.normal(0, 1, N).astype(np.float32))
        adv = torch.from_numpy(rs.normal(0.3, 2.0, N).astype(np.float32))
        res = m.loss({"state": state}, prev, nxt, kinds, ret, oldv, adv.clone(), oldlp,
                     use_bc_loss=False, reward_horizon=rh)
        pg, ent, vl = res[0], res[1], res[2]
        (pg + 0.5 * vl).backward()
        out.update({f"{cname}_state": state, f"{cname}_prev": prev, f"{cname}_next": nxt, f"{cname}_kinds": kinds,
                    f"{cname}_returns": ret, f"{cname}_oldvalues": oldv, f"{cname}_adv": adv,
                    f"{cname}_oldlogprobs": oldlp, f"{cname}_reward_horizon": rh,
                    f"{cname}_stats": np.array([pg.item(), float(ent), vl.item(), res[3], res[4], res[5],
                                                float(res[6]), res[7]], dtype=np.float64)})
        for k, p in m.actor_ft.named_parameters():
            put_grad(out, f"{cname}_gactor_{k}", p.grad)
        for k, p in m.critic.named_parameters():
            put_grad(out, f"{cname}_gcritic_{k}", p.grad)
    save("g5_loss", **out)


# ---------------------------------------------------------------- G8 behaviour-cloning term of the loss
BC_CASES = {
    "bc_ddpm": ("hopper", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01)),
    "bc_ddim_kitchen": ("kitchen_like", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                            clip_ploss_coef=0.01, min_sampling_denoising_std=0.08)),
}


def g8_bc():
    out = {}
    rs = np.random.RandomState(800)
    N = 16
    for cname, (sname, kw) in BC_CASES.items():
        a, c = specs(sname)
        if kw.get("use_ddim"):
            kw = dict(kw, eta=EtaFixed(base_eta=1.0))
        m = ref_model(a, c, 41, gamma_denoising=0.99, randn_clip_value=3, **kw)
        Kft = kw["ft_denoising_steps"]
        n_steps = kw["ddim_steps"] if kw.get("use_ddim") else kw["denoising_steps"]
        state = torch.from_numpy(rs.uniform(-1, 1, size=(N, 1, a.cond_dim)).astype(np.float32))
        noise = torch.from_numpy(rs.randn(n_steps + 1, N, a.horizon_steps, a.action_dim).astype(np.float32))
        with recorded_noise(list(noise)):
            base_chains = m(cond={"state": state}, deterministic=False, return_chain=True, use_base_policy=True).chains
        # the PPO inputs only have to be valid: the recorded outputs are the BC term and its gradient alone
        kinds = torch.from_numpy(rs.randint(0, Kft, size=(N,)).astype(np.int64))
        rows = torch.arange(N)
        prev, nxt = base_chains[rows, kinds], base_chains[rows, kinds + 1]
        oldlp = torch.from_numpy(rs.normal(-1, 0.5, size=(N, a.horizon_steps, a.action_dim)).astype(np.float32))
        ret = torch.from_numpy(rs.normal(0, 1, N).astype(np.float32))
        adv = torch.from_numpy(rs.normal(0.3, 2.0, N).astype(np.float32))
        with recorded_noise(list(noise)):
            res = m.loss({"state": state}, prev, nxt, kinds, ret, ret.clone(), adv, oldlp, use_bc_loss=True,
                         reward_horizon=4)
        bc = res[6]
        bc.backward()
        out.update({f"{cname}_state": state, f"{cname}_noise": noise, f"{cname}_base_chains": base_chains,
                    f"{cname}_bc_loss": np.float64(bc.item())})
        for k, p in m.actor_ft.named_parameters():
            put_grad(out, f"{cname}_gbc_{k}", p.grad)
    save("g8_bc", **out)


# ---------------------------------------------------------------- G9 supervised denoising loss (pre-training)
MSE_CASES = {"mse_hopper": ("hopper", 20), "mse_can_k100": ("can", 100), "mse_square_like": ("square_like", 20),
             "mse_ln_relu": ("ln_relu", 20), "mse_can_relu_k100": ("can_relu", 100)}


def g9_denoise_mse():
    """DiffusionModel.p_losses of the reference (model/diffusion/diffusion.py:325-349) on its ``network`` (= the base
    actor of a PPODiffusion, whose weights are the seeded recipe) with recorded t and noise, and every gradient."""
    out = {}
    rs = np.random.RandomState(900)
    N = 24
    for cname, (sname, K) in MSE_CASES.items():
        a, c = specs(sname)
        m = ref_model(a, c, 51, gamma_denoising=0.99, clip_ploss_coef=0.01, ft_denoising_steps=min(10, K),
                      denoising_steps=K)
        net = m.network
        for p in net.parameters():
            p.requires_grad_(True)
        x0 = torch.from_numpy(rs.uniform(-1, 1, size=(N, a.horizon_steps, a.action_dim)).astype(np.float32))
        state = torch.from_numpy(rs.uniform(-1, 1, size=(N, 1, a.cond_dim)).astype(np.float32))
        t = torch.from_numpy(rs.randint(0, K, size=(N,)).astype(np.int64))
        noise = torch.from_numpy(rs.randn(N, a.horizon_steps, a.action_dim).astype(np.float32))
        with recorded_noise([noise]):
            loss = m.p_losses(x0, {"state": state}, t)
        loss.backward()
        out.update({f"{cname}_x0": x0, f"{cname}_state": state, f"{cname}_t": t, f"{cname}_noise": noise,
                    f"{cname}_xnoisy": m.q_sample(x0, t, noise), f"{cname}_loss": np.float64(loss.item())})
        for k, p in net.named_parameters():
            put_grad(out, f"{cname}_g_{k}", p.grad)
    save("g9_denoise_mse", **out)


# ---------------------------------------------------------------- G6 reward scaler (GAE loop is not importable)
def g6_reward_scaler():
    rs = np.random.RandomState(400)
    n_envs, n_steps = 4, 16
    sc = RunningRewardScaler(n_envs)
    out = {}
    for it in range(3):
        r = rs.normal(1.0, 2.0, size=(n_envs, n_steps))
        f = (rs.uniform(size=(n_envs, n_steps)) < 0.15).astype(np.float64)
        out[f"it{it}_reward"] = r
        out[f"it{it}_first"] = f
        out[f"it{it}_scaled"] = sc(reward=r, first=f)
        out[f"it{it}_var"] = np.float64(sc.ret_rms.var)
    save("g6_reward_scaler", **out)


# ---------------------------------------------------------------- G7 one optimiser step (torch.optim.AdamW)
def g7_adamw():
    rs = np.random.RandomState(500)
    p0 = rs.normal(size=(257,)).astype(np.float32)
    out = {"p0": p0}
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=0.01)
    for i in range(3):
        g = rs.normal(size=(257,)).astype(np.float32)
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        out[f"g{i}"] = g
        out[f"p{i + 1}"] = p.detach().clone()
    g = torch.from_numpy(rs.normal(size=(257,)).astype(np.float32))
    q = torch.nn.Parameter(torch.zeros(257))
    q.grad = g.clone()
    out["clip_in"] = g
    out["clip_total"] = torch.nn.utils.clip_grad_norm_([q], 1.5)
    out["clip_out"] = q.grad
    save("g7_adamw", **out)


# ---------------------------------------------------------------- G11 DiffusionEval (checkpoint -> evaluation sampling)
from make_golden_cases import EVAL_CASES  # noqa: E402


def g11_eval():
    """The reference's DiffusionEval (model/diffusion/diffusion_eval.py:19-150) built from a checkpoint file written here
    from seeded weights (RL checkpoint: actor.* / actor_ft.* keys of a PPODiffusion; pre-training checkpoint: network.*
    keys), sampled with recorded noise.  Only inputs / outputs are stored; the test rebuilds the checkpoint itself."""
    import tempfile
    from dppo.model.diffusion.diffusion_eval import DiffusionEval
    out = {}
    rs = np.random.RandomState(1100)
    for cname, (sname, B, kw, ft, kind) in EVAL_CASES.items():
        a, c = specs(sname)
        src = ref_model(a, c, 61, gamma_denoising=0.99, clip_ploss_coef=0.01, ft_denoising_steps=max(ft, 1),
                        **({"eta": EtaFixed(base_eta=1.0)} if kw.get("use_ddim") else {}), **kw)
        sd = src.state_dict()
        if kind == "pretrain":  # what agent/pretrain/train_agent.py:146-168 saves: a DiffusionModel's `network.*`
            sd = {k: v for k, v in sd.items() if k.startswith("network.")}
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "state.pt")
            torch.save({"itr": 0, "model": sd}, path)
            ev = DiffusionEval(network_path=path, ft_denoising_steps=ft, network=ref_actor(a, O.init_params(a, 999)),
                               horizon_steps=a.horizon_steps, obs_dim=a.cond_dim, action_dim=a.action_dim, device="cpu",
                               **kw)
        n_steps = kw["ddim_steps"] if kw.get("use_ddim") else kw["denoising_steps"]
        state = torch.from_numpy(rs.uniform(-1, 1, size=(B, 1, a.cond_dim)).astype(np.float32))
        noise = torch.from_numpy(rs.randn(n_steps + 1, B, a.horizon_steps, a.action_dim).astype(np.float32))
        with recorded_noise(list(noise)):
            smp = ev(cond={"state": state}, deterministic=True)
        out.update({f"{cname}_state": state, f"{cname}_noise": noise, f"{cname}_traj": smp.trajectories})
    save("g11_eval", **out)


# ---------------------------------------------------------------- G12 Gaussian-policy PPO
from make_golden_cases import GAUSS_CASES  # noqa: E402


def gauss_logvar(spec, kw, seed):
    """Seeded per-dimension log-variance: the constructor's log(fixed_std^2) plus U(-0.4, 0.4) (some entries leave the
    clamp range of the `gauss_nonorm` case on purpose)."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy((np.log(kw["fixed_std"] ** 2) + rs.uniform(-0.4, 0.4, size=spec.action_dim)).astype(np.float32))


def g12_gaussian():
    """PPO_Gaussian (model/rl/gaussian_ppo.py) over Gaussian_MLP (model/common/mlp_gaussian.py:283-362) + CriticObs:
    sampling with recorded noise, get_logprobs, loss 8-tuple and every gradient (incl. the learned logvar's)."""
    from dppo.model.common.mlp_gaussian import Gaussian_MLP
    from dppo.model.rl.gaussian_ppo import PPO_Gaussian
    out = {}
    rs = np.random.RandomState(1200)
    N = 64
    for cname, (sname, kw) in GAUSS_CASES.items():
        a, c = specs(sname)
        actor = Gaussian_MLP(action_dim=a.action_dim, horizon_steps=a.horizon_steps, cond_dim=a.cond_dim,
                             mlp_dims=list(a.mlp_dims), activation_type=a.activation, residual_style=True,
                             fixed_std=kw["fixed_std"], learn_fixed_std=kw["learn_fixed_std"], std_min=kw["std_min"],
                             std_max=kw["std_max"])
        sd = dict(O.init_params(a, 71))
        sd["logvar_min"], sd["logvar_max"] = actor.logvar_min.data.clone(), actor.logvar_max.data.clone()
        if kw["learn_fixed_std"]:
            sd["logvar"] = gauss_logvar(a, kw, 73)
        actor.load_state_dict(sd, strict=True)
        m = PPO_Gaussian(actor=actor, critic=ref_critic(c, O.init_params(c, 72)), horizon_steps=a.horizon_steps, device="cpu",
                         clip_ploss_coef=kw["clip_ploss_coef"], clip_vloss_coef=kw.get("clip_vloss_coef"),
                         norm_adv=kw.get("norm_adv", True), randn_clip_value=kw["randn_clip_value"])
        state = torch.from_numpy(rs.uniform(-1, 1, size=(N, 1, a.cond_dim)).astype(np.float32))
        noise = torch.from_numpy(rs.randn(N, a.horizon_steps * a.action_dim).astype(np.float32))
        normal0 = torch.normal
        torch.normal = lambda loc, scale, *a_, **k_: loc + scale * noise  # dist.sample() -> torch.normal(loc, scale)
        try:
            actions = m(cond={"state": state}, deterministic=False)
            det = m(cond={"state": state}, deterministic=True)
        finally:
            torch.normal = normal0
        with torch.no_grad():
            lp, ent, std = m.get_logprobs({"state": state}, actions)
        oldlp = lp + torch.from_numpy(rs.normal(0, 0.02, N).astype(np.float32))
        with torch.no_grad():
            oldv = m.critic({"state": state}).view(-1) + torch.from_numpy(rs.normal(0, 0.3, N).astype(np.float32))
        ret = torch.from_numpy(rs.normal(0, 1, N).astype(np.float32))
        adv = torch.from_numpy(rs.normal(0.3, 2.0, N).astype(np.float32))
        res = m.loss({"state": state}, actions, ret, oldv, adv.clone(), oldlp, use_bc_loss=False)
        (res[0] + 0.01 * res[1] + 0.5 * res[2]).backward()
        out.update({f"{cname}_state": state, f"{cname}_noise": noise, f"{cname}_actions": actions, f"{cname}_actions_det": det,
                    f"{cname}_logprobs": lp, f"{cname}_entropy": np.float64(ent.item()), f"{cname}_std": np.float64(std.item()),
                    f"{cname}_oldlogprobs": oldlp, f"{cname}_oldvalues": oldv, f"{cname}_returns": ret, f"{cname}_adv": adv,
                    f"{cname}_stats": np.array([res[0].item(), res[1].item(), res[2].item(), res[3], res[4], res[5],
                                                float(res[6]), res[7]], dtype=np.float64)})
        for k, p in m.actor_ft.named_parameters():
            if p.grad is not None:
                put_grad(out, f"{cname}_gactor_{k}", p.grad)
        for k, p in m.critic.named_parameters():
            put_grad(out, f"{cname}_gcritic_{k}", p.grad)
    save("g12_gaussian", **out)


# ---------------------------------------------------------------- G13 conv denoiser (Unet1D)
from make_golden_cases import UNET_CHAIN_CASES, UNET_SPECS  # noqa: E402


def ref_unet(spec, params):
    from dppo.model.diffusion.unet import Unet1D
    m = Unet1D(action_dim=spec.action_dim, cond_dim=spec.cond_dim, diffusion_step_embed_dim=spec.diffusion_step_embed_dim,
               dim=spec.dim, dim_mults=list(spec.dim_mults), smaller_encoder=spec.smaller_encoder,
               kernel_size=spec.kernel_size, n_groups=spec.n_groups, activation_type=spec.activation,
               cond_predict_scale=spec.cond_predict_scale, groupnorm_eps=spec.groupnorm_eps)
    m.load_state_dict(params, strict=True)  # strict: the oracle's parameter list IS the reference's state dict
    return m


def g13_unet(UNET_SPECS=UNET_SPECS, UNET_CHAIN_CASES=UNET_CHAIN_CASES, seed=1300, fname="g13_unet", out=None):
    """Unet1D.forward (model/diffusion/unet.py:264-327), one ResidualBlock1D (:100-118), one Conv1dBlock
    (modules.py:50-95), and K-step chains + log-probs of a PPODiffusion whose actor is the UNet."""
    own = out is None
    out = {} if own else out
    rs = np.random.RandomState(seed)
    for name, kw in UNET_SPECS.items():
        u = O.UnetSpec(**kw)
        p = O.unet_init_params(u, 81)
        net = ref_unet(u, p)
        B = 6
        x = torch.from_numpy(rs.randn(B, u.horizon_steps, u.action_dim).astype(np.float32))
        t = torch.from_numpy(rs.randint(0, 20, size=(B,)).astype(np.int64))
        s = torch.from_numpy(rs.uniform(-1, 1, size=(B, 1, u.cond_dim)).astype(np.float32))
        with torch.no_grad():
            y = net(x, t, cond={"state": s})
            # pieces, fed with recorded inputs
            blk = net.down_modules[1][0]
            ci = blk.blocks[0].block[0].in_channels
            bx = torch.from_numpy(rs.randn(B, ci, u.horizon_steps // 2).astype(np.float32))
            bc = torch.from_numpy(rs.randn(B, u.cond_block_dim).astype(np.float32))
            by = blk(bx, bc)
            cy = net.final_conv[0](bx[:, :u.dim].repeat(1, 2, 1)[:, :u.dim])
        out.update({f"{name}_x": x, f"{name}_t": t, f"{name}_state": s, f"{name}_eps": y, f"{name}_blk_x": bx,
                    f"{name}_blk_cond": bc, f"{name}_blk_y": by, f"{name}_cb_y": cy})
    for cname, (sname, B, kw, det) in UNET_CHAIN_CASES.items():
        u = O.UnetSpec(**UNET_SPECS[sname])
        _, c = specs("hopper")
        c = O.NetSpec("critic", cond_dim=u.cond_dim, mlp_dims=[256, 256, 256], activation="Mish", residual=True)
        kw2 = dict(kw, eta=EtaFixed(base_eta=1.0)) if kw.get("use_ddim") else kw
        m = PPODiffusion(actor=ref_unet(u, O.unet_init_params(u, 21)), critic=ref_critic(c, O.init_params(c, 23)),
                         horizon_steps=u.horizon_steps, obs_dim=u.cond_dim, action_dim=u.action_dim, device="cpu",
                         gamma_denoising=0.99, clip_ploss_coef=0.01, **kw2)
        m.actor_ft.load_state_dict(O.unet_init_params(u, 22), strict=True)
        n_steps = kw["ddim_steps"] if kw.get("use_ddim") else kw["denoising_steps"]
        state = torch.from_numpy(rs.uniform(-1, 1, size=(B, 1, u.cond_dim)).astype(np.float32))
        noise = torch.from_numpy(rs.randn(n_steps + 1, B, u.horizon_steps, u.action_dim).astype(np.float32))
        with recorded_noise(list(noise)):
            smp = m(cond={"state": state}, deterministic=det, return_chain=True)
        with torch.no_grad():
            lp = m.get_logprobs({"state": state}, smp.chains)
        out.update({f"{cname}_state": state, f"{cname}_noise": noise, f"{cname}_traj": smp.trajectories,
                    f"{cname}_chains": smp.chains, f"{cname}_logprobs": lp})
    if own:
        save(fname, **out)


# ---------------------------------------------------------------- G14 conv denoiser: PPO loss and supervised loss with gradients
from make_golden_cases import UNET_LOSS_CASES, UNET_MSE_CASES  # noqa: E402


def unet_model(u, seed, **kw):
    c = O.NetSpec("critic", cond_dim=u.cond_dim, mlp_dims=[256, 256, 256], activation="Mish", residual=True)
    if kw.get("use_ddim"):
        kw = dict(kw, eta=EtaFixed(base_eta=1.0))
    m = PPODiffusion(actor=ref_unet(u, O.unet_init_params(u, seed)), critic=ref_critic(c, O.init_params(c, seed + 2)),
                     horizon_steps=u.horizon_steps, obs_dim=u.cond_dim, action_dim=u.action_dim, device="cpu",
                     gamma_denoising=0.99, randn_clip_value=3, **kw)
    m.actor_ft.load_state_dict(O.unet_init_params(u, seed + 1), strict=True)
    return m, c


def g14_unet_loss(UNET_SPECS=UNET_SPECS, UNET_LOSS_CASES=UNET_LOSS_CASES, UNET_MSE_CASES=UNET_MSE_CASES, seed=1400,
                  fname="g14_unet_loss", out=None):
    own = out is None
    out = {} if own else out
    rs = np.random.RandomState(seed)
    for cname, (sname, N, kw, rh) in UNET_LOSS_CASES.items():
        u = O.UnetSpec(**UNET_SPECS[sname])
        m, c = unet_model(u, 31, **kw)
        Kft = kw["ft_denoising_steps"]
        n_steps = kw["ddim_steps"] if kw.get("use_ddim") else kw["denoising_steps"]
        state = torch.from_numpy(rs.uniform(-1, 1, size=(N, 1, u.cond_dim)).astype(np.float32))
        noise = torch.from_numpy(rs.randn(n_steps + 1, N, u.horizon_steps, u.action_dim).astype(np.float32))
        with recorded_noise(list(noise)):
            chains = m(cond={"state": state}, deterministic=False, return_chain=True).chains
        kinds = torch.from_numpy(rs.randint(0, Kft, size=(N,)).astype(np.int64))
        rows = torch.arange(N)
        prev, nxt = chains[rows, kinds], chains[rows, kinds + 1]
        with torch.no_grad():
            oldlp_all = m.get_logprobs({"state": state}, chains).reshape(N, Kft, u.horizon_steps, u.action_dim)
            oldlp = oldlp_all[rows, kinds] + torch.from_numpy(
                rs.normal(0, 0.02, size=(N, u.horizon_steps, u.action_dim)).astype(np.float32))
            oldv = m.critic({"state": state}).view(-1) + torch.from_numpy(rs.normal(0, 0.3, N).astype(np.float32))
        ret = torch.from_numpy(rs.normal(0, 1, N).astype(np.float32))
        adv = torch.from_numpy(rs.normal(0.3, 2.0, N).astype(np.float32))
        res = m.loss({"state": state}, prev, nxt, kinds, ret, oldv, adv.clone(), oldlp, use_bc_loss=False, reward_horizon=rh)
        (res[0] + 0.5 * res[2]).backward()
        out.update({f"{cname}_state": state, f"{cname}_prev": prev, f"{cname}_next": nxt, f"{cname}_kinds": kinds,
                    f"{cname}_returns": ret, f"{cname}_oldvalues": oldv, f"{cname}_adv": adv, f"{cname}_oldlogprobs": oldlp,
                    f"{cname}_reward_horizon": rh,
                    f"{cname}_stats": np.array([res[0].item(), float(res[1]), res[2].item(), res[3], res[4], res[5],
                                                float(res[6]), res[7]], dtype=np.float64)})
        for k, p in m.actor_ft.named_parameters():
            put_grad(out, f"{cname}_gactor_{k}", p.grad)
        for k, p in m.critic.named_parameters():
            put_grad(out, f"{cname}_gcritic_{k}", p.grad)
    for cname, (sname, K, N) in UNET_MSE_CASES.items():
        u = O.UnetSpec(**UNET_SPECS[sname])
        m, _ = unet_model(u, 51, denoising_steps=K, ft_denoising_steps=min(10, K), clip_ploss_coef=0.01)
        net = m.network
        for p in net.parameters():
            p.requires_grad_(True)
        x0 = torch.from_numpy(rs.uniform(-1, 1, size=(N, u.horizon_steps, u.action_dim)).astype(np.float32))
        state = torch.from_numpy(rs.uniform(-1, 1, size=(N, 1, u.cond_dim)).astype(np.float32))
        t = torch.from_numpy(rs.randint(0, K, size=(N,)).astype(np.int64))
        noise = torch.from_numpy(rs.randn(N, u.horizon_steps, u.action_dim).astype(np.float32))
        with recorded_noise([noise]):
            loss = m.p_losses(x0, {"state": state}, t)
        loss.backward()
        out.update({f"{cname}_x0": x0, f"{cname}_state": state, f"{cname}_t": t, f"{cname}_noise": noise,
                    f"{cname}_loss": np.float64(loss.item())})
        for k, p in net.named_parameters():
            put_grad(out, f"{cname}_g_{k}", p.grad)
    if own:
        save(fname, **out)


def g15_unet_dim40():
    """The dim: 40 conv denoiser of the robomimic can / lift cfgs (cfg/robomimic/finetune/can/ft_ppo_diffusion_unet.yaml:93-103)
    and a three-level variant: forward, blocks, chain + log-probs, PPO loss and supervised loss with every gradient."""
    from make_golden_cases import UNET40_CHAIN_CASES, UNET40_LOSS_CASES, UNET40_MSE_CASES, UNET40_SPECS
    out = {}
    g13_unet(UNET40_SPECS, UNET40_CHAIN_CASES, 1500, None, out)
    g14_unet_loss(UNET40_SPECS, UNET40_LOSS_CASES, UNET40_MSE_CASES, 1501, None, out)
    save("g15_unet_dim40", **out)


# ---------------------------------------------------------------- G16 / G17 pixel observations: ViT + SpatialEmb networks
from make_golden_cases import (VIS_ALL_NETS, VIS_C5_CHAIN_CASES, VIS_C5_LOSS_CASES, VIS_C5_NETS,  # noqa: E402
                               VIS_CHAIN_CASES, VIS_FWD_BATCH, VIS_LOSS_CASES, VIS_MSE_CASES, VIS_NETS,
                               VIS_SPECS)


def vis_net_specs(name):
    """(VisSpec, trunk spec on cat[feat, state], critic trunk spec) of a VIS_NETS entry."""
    vname, kind, kw = VIS_ALL_NETS[name]
    v = O.VisSpec(**VIS_SPECS[vname])
    cd = v.feat_dim + v.prop_dim
    trunk = O.UnetSpec(cond_dim=cd, **kw) if kind == "unet" else O.NetSpec("actor", cond_dim=cd, residual=True, **kw)
    critic = O.NetSpec("critic", cond_dim=cd, mlp_dims=[256, 256, 256], activation="Mish", residual=True)
    return v, trunk, critic


def ref_vit(v):
    from dppo.model.common.vit import VitEncoder, VitEncoderConfig
    cfg = VitEncoderConfig(patch_size=8, depth=v.depth, embed_dim=v.embed_dim, num_heads=v.num_heads, embed_style="embed2",
                           embed_norm=0)
    return VitEncoder([v.in_ch, v.img_h, v.img_w], cfg, num_channel=v.in_ch, img_h=v.img_h, img_w=v.img_w)


def ref_vision_actor(v, trunk, params):
    common = dict(backbone=ref_vit(v), action_dim=trunk.action_dim, cond_dim=v.prop_dim, img_cond_steps=v.in_ch // 3,
                  spatial_emb=v.spatial_emb, num_img=v.num_img, augment=False)
    if trunk.kind == "unet":
        from dppo.model.diffusion.unet import VisionUnet1D
        m = VisionUnet1D(diffusion_step_embed_dim=trunk.diffusion_step_embed_dim, dim=trunk.dim,
                         dim_mults=list(trunk.dim_mults), smaller_encoder=trunk.smaller_encoder,
                         kernel_size=trunk.kernel_size, n_groups=trunk.n_groups, activation_type=trunk.activation,
                         cond_predict_scale=trunk.cond_predict_scale, groupnorm_eps=trunk.groupnorm_eps, **common)
    else:
        from dppo.model.diffusion.mlp_diffusion import VisionDiffusionMLP
        m = VisionDiffusionMLP(horizon_steps=trunk.horizon_steps, time_dim=trunk.time_dim, mlp_dims=list(trunk.mlp_dims),
                               activation_type=trunk.activation, residual_style=True, **common)
    m.load_state_dict(params, strict=True)  # strict: the oracle's parameter list IS the reference's state dict
    return m


def ref_vit_critic(v, cspec, params):
    from dppo.model.common.critic import ViTCritic
    m = ViTCritic(backbone=ref_vit(v), cond_dim=v.prop_dim, img_cond_steps=v.in_ch // 3, spatial_emb=v.spatial_emb,
                  num_img=v.num_img, augment=False, mlp_dims=list(cspec.mlp_dims), activation_type=cspec.activation,
                  residual_style=True)
    m.load_state_dict(params, strict=True)
    return m


def vis_inputs(rs, v, B):
    """Images with three bits per byte (they compress), (B, img_cond_steps, 3 * num_img, H, W) uint8; state in [-1, 1]."""
    T = v.in_ch // 3
    rgb = (rs.randint(0, 8, size=(B, T, 3 * v.num_img, v.img_h, v.img_w)) * 36).astype(np.uint8)
    state = rs.uniform(-1, 1, size=(B, 1, v.prop_dim)).astype(np.float32)
    return rgb, state


def vis_cond(rgb, state):
    return {"rgb": torch.from_numpy(rgb), "state": torch.from_numpy(state)}


def vision_model(name, seed, **kw):
    v, trunk, cspec = vis_net_specs(name)
    if kw.get("use_ddim"):
        kw = dict(kw, eta=EtaFixed(base_eta=1.0))
    actor = ref_vision_actor(v, trunk, O.vision_init_params(v, trunk, seed))
    critic = ref_vit_critic(v, cspec, O.vision_init_params(v, cspec, seed + 2))
    m = PPODiffusion(actor=actor, critic=critic, horizon_steps=trunk.horizon_steps, obs_dim=v.prop_dim,
                     action_dim=trunk.action_dim, device="cpu", gamma_denoising=0.99, **dict(dict(randn_clip_value=3), **kw))
    m.actor_ft.load_state_dict(O.vision_init_params(v, trunk, seed + 1), strict=True)
    return m, v, trunk, cspec


def g16_vision():
    """VitEncoder.forward (model/common/vit.py:55-61), SpatialEmb.forward (modules.py:31-41), VisionDiffusionMLP /
    VisionUnet1D / ViTCritic.forward (mlp_diffusion.py:101-171, unet.py:530-620, critic.py:159-206), and K-step chains +
    log-probs of a PPODiffusion whose actor observes pixels."""
    out = {}
    rs = np.random.RandomState(1600)
    for vname, kw in VIS_SPECS.items():
        v = O.VisSpec(**kw)
        p = O.vis_init_params(v, 61)
        B = VIS_FWD_BATCH[vname]
        rgb, state = vis_inputs(rs, v, B)
        enc = ref_vit(v)
        enc.load_state_dict({k[len("backbone."):]: t for k, t in p.items() if k.startswith("backbone.")}, strict=True)
        from dppo.model.common.modules import SpatialEmb
        img = torch.from_numpy(rgb).float()
        if v.num_img > 1:
            img = img.reshape(B, -1, v.num_img, 3, v.img_h, v.img_w).permute(0, 2, 1, 3, 4, 5).flatten(2, 3)[:, 0]
        else:
            img = img.flatten(1, 2)
        with torch.no_grad():
            feats = enc(img)
            c = v.compress_names()[0]
            se = SpatialEmb(num_patch=v.num_patch, patch_dim=v.embed_dim, prop_dim=v.prop_dim, proj_dim=v.spatial_emb, dropout=0)
            se.load_state_dict({k[len(c) + 1:]: t for k, t in p.items() if k.startswith(c + ".")}, strict=True)
            z = se(feats, torch.from_numpy(state).reshape(B, -1))
        out.update({f"{vname}_rgb": rgb, f"{vname}_state": state, f"{vname}_feats0": feats[0], f"{vname}_z": z})
    for name in VIS_NETS:
        v, trunk, cspec = vis_net_specs(name)
        B = 2 if v.img_h > 64 else 5
        rgb, state = vis_inputs(rs, v, B)
        net = ref_vision_actor(v, trunk, O.vision_init_params(v, trunk, 71))
        cr = ref_vit_critic(v, cspec, O.vision_init_params(v, cspec, 73))
        x = torch.from_numpy(rs.randn(B, trunk.horizon_steps, trunk.action_dim).astype(np.float32))
        t = torch.from_numpy(rs.randint(0, 20, size=(B,)).astype(np.int64))
        with torch.no_grad():
            y = net(x, t, cond=vis_cond(rgb, state))
            val = cr(vis_cond(rgb, state))
        out.update({f"{name}_rgb": rgb, f"{name}_state": state, f"{name}_x": x, f"{name}_t": t, f"{name}_eps": y,
                    f"{name}_value": val})
    for cname, (name, B, kw, det) in VIS_CHAIN_CASES.items():
        m, v, trunk, _ = vision_model(name, 21, clip_ploss_coef=0.01, **kw)
        n_steps = kw["ddim_steps"] if kw.get("use_ddim") else kw["denoising_steps"]
        rgb, state = vis_inputs(rs, v, B)
        noise = torch.from_numpy(rs.randn(n_steps + 1, B, trunk.horizon_steps, trunk.action_dim).astype(np.float32))
        with recorded_noise(list(noise)):
            smp = m(cond=vis_cond(rgb, state), deterministic=det, return_chain=True)
        with torch.no_grad():
            lp = m.get_logprobs(vis_cond(rgb, state), smp.chains)
        out.update({f"{cname}_rgb": rgb, f"{cname}_state": state, f"{cname}_noise": noise, f"{cname}_traj": smp.trajectories,
                    f"{cname}_chains": smp.chains, f"{cname}_logprobs": lp})
    save("g16_vision", **out)


def g17_vision_loss():
    """PPODiffusion.loss (diffusion_ppo.py:57-199) and the supervised loss (diffusion.py:325-349) with pixel networks: the
    gradients reach the ViT and SpatialEmb parameters of the fine-tuned actor and of the critic."""
    out = {}
    rs = np.random.RandomState(1700)
    for cname, (name, N, kw, rh) in VIS_LOSS_CASES.items():
        m, v, trunk, cspec = vision_model(name, 31, **kw)
        Kft = kw["ft_denoising_steps"]
        n_steps = kw["ddim_steps"] if kw.get("use_ddim") else kw["denoising_steps"]
        rgb, state = vis_inputs(rs, v, N)
        cond = vis_cond(rgb, state)
        Ta, Da = trunk.horizon_steps, trunk.action_dim
        noise = torch.from_numpy(rs.randn(n_steps + 1, N, Ta, Da).astype(np.float32))
        with recorded_noise(list(noise)):
            chains = m(cond=cond, deterministic=False, return_chain=True).chains
        kinds = torch.from_numpy(rs.randint(0, Kft, size=(N,)).astype(np.int64))
        rows = torch.arange(N)
        prev, nxt = chains[rows, kinds], chains[rows, kinds + 1]
        with torch.no_grad():
            oldlp_all = m.get_logprobs(cond, chains).reshape(N, Kft, Ta, Da)
            oldlp = oldlp_all[rows, kinds] + torch.from_numpy(rs.normal(0, 0.02, size=(N, Ta, Da)).astype(np.float32))
            oldv = m.critic(cond).view(-1) + torch.from_numpy(rs.normal(0, 0.3, N).astype(np.float32))
        ret = torch.from_numpy(rs.normal(0, 1, N).astype(np.float32))
        adv = torch.from_numpy(rs.normal(0.3, 2.0, N).astype(np.float32))
        res = m.loss(cond, prev, nxt, kinds, ret, oldv, adv.clone(), oldlp, use_bc_loss=False, reward_horizon=rh)
        (res[0] + 0.5 * res[2]).backward()
        out.update({f"{cname}_rgb": rgb, f"{cname}_state": state, f"{cname}_prev": prev, f"{cname}_next": nxt,
                    f"{cname}_kinds": kinds, f"{cname}_returns": ret, f"{cname}_oldvalues": oldv, f"{cname}_adv": adv,
                    f"{cname}_oldlogprobs": oldlp, f"{cname}_reward_horizon": rh,
                    f"{cname}_stats": np.array([res[0].item(), float(res[1]), res[2].item(), res[3], res[4], res[5],
                                                float(res[6]), res[7]], dtype=np.float64)})
        for k, p in m.actor_ft.named_parameters():
            put_grad(out, f"{cname}_gactor_{k}", p.grad)
        for k, p in m.critic.named_parameters():
            put_grad(out, f"{cname}_gcritic_{k}", p.grad)
    for cname, (name, K, N) in VIS_MSE_CASES.items():
        m, v, trunk, _ = vision_model(name, 51, denoising_steps=K, ft_denoising_steps=min(10, K), clip_ploss_coef=0.01)
        net = m.network
        for p in net.parameters():
            p.requires_grad_(True)
        Ta, Da = trunk.horizon_steps, trunk.action_dim
        x0 = torch.from_numpy(rs.uniform(-1, 1, size=(N, Ta, Da)).astype(np.float32))
        rgb, state = vis_inputs(rs, v, N)
        t = torch.from_numpy(rs.randint(0, K, size=(N,)).astype(np.int64))
        noise = torch.from_numpy(rs.randn(N, Ta, Da).astype(np.float32))
        with recorded_noise([noise]):
            loss = m.p_losses(x0, vis_cond(rgb, state), t)
        loss.backward()
        out.update({f"{cname}_x0": x0, f"{cname}_rgb": rgb, f"{cname}_state": state, f"{cname}_t": t, f"{cname}_noise": noise,
                    f"{cname}_loss": np.float64(loss.item())})
        for k, p in net.named_parameters():
            put_grad(out, f"{cname}_g_{k}", p.grad)
    save("g17_vision_loss", **out)


def g21_vision_c5():
    """BASELINE configs[4] end to end at the shipped shape (VisionUnet1D behind the 96 x 96 ViT + SpatialEmb, ViTCritic, DDIM
    100 -> 5): network forward, a K-step chain with recorded noise + its log-probs, and PPODiffusion.loss with EVERY gradient
    (both encoders, both trunks).  Same recipes as g16 / g17, which pin the pieces at smaller shapes."""
    out = {}
    rs = np.random.RandomState(2100)
    for name in VIS_C5_NETS:
        v, trunk, cspec = vis_net_specs(name)
        B = 2
        rgb, state = vis_inputs(rs, v, B)
        net = ref_vision_actor(v, trunk, O.vision_init_params(v, trunk, 71))
        cr = ref_vit_critic(v, cspec, O.vision_init_params(v, cspec, 73))
        x = torch.from_numpy(rs.randn(B, trunk.horizon_steps, trunk.action_dim).astype(np.float32))
        t = torch.from_numpy(rs.randint(0, 20, size=(B,)).astype(np.int64))
        with torch.no_grad():
            y = net(x, t, cond=vis_cond(rgb, state))
            val = cr(vis_cond(rgb, state))
        out.update({f"{name}_rgb": rgb, f"{name}_state": state, f"{name}_x": x, f"{name}_t": t, f"{name}_eps": y,
                    f"{name}_value": val})
    for cname, (name, B, kw, det) in VIS_C5_CHAIN_CASES.items():
        m, v, trunk, _ = vision_model(name, 21, clip_ploss_coef=0.01, **kw)
        n_steps = kw["ddim_steps"] if kw.get("use_ddim") else kw["denoising_steps"]
        rgb, state = vis_inputs(rs, v, B)
        noise = torch.from_numpy(rs.randn(n_steps + 1, B, trunk.horizon_steps, trunk.action_dim).astype(np.float32))
        with recorded_noise(list(noise)):
            smp = m(cond=vis_cond(rgb, state), deterministic=det, return_chain=True)
        with torch.no_grad():
            lp = m.get_logprobs(vis_cond(rgb, state), smp.chains)
        out.update({f"{cname}_rgb": rgb, f"{cname}_state": state, f"{cname}_noise": noise, f"{cname}_traj": smp.trajectories,
                    f"{cname}_chains": smp.chains, f"{cname}_logprobs": lp})
    for cname, (name, N, kw, rh) in VIS_C5_LOSS_CASES.items():
        m, v, trunk, cspec = vision_model(name, 31, **kw)
        Kft = kw["ft_denoising_steps"]
        n_steps = kw["ddim_steps"] if kw.get("use_ddim") else kw["denoising_steps"]
        rgb, state = vis_inputs(rs, v, N)
        cond = vis_cond(rgb, state)
        Ta, Da = trunk.horizon_steps, trunk.action_dim
        noise = torch.from_numpy(rs.randn(n_steps + 1, N, Ta, Da).astype(np.float32))
        with recorded_noise(list(noise)):
            chains = m(cond=cond, deterministic=False, return_chain=True).chains
        kinds = torch.from_numpy(rs.randint(0, Kft, size=(N,)).astype(np.int64))
        rows = torch.arange(N)
        prev, nxt = chains[rows, kinds], chains[rows, kinds + 1]
        with torch.no_grad():
            oldlp_all = m.get_logprobs(cond, chains).reshape(N, Kft, Ta, Da)
            oldlp = oldlp_all[rows, kinds] + torch.from_numpy(rs.normal(0, 0.02, size=(N, Ta, Da)).astype(np.float32))
            oldv = m.critic(cond).view(-1) + torch.from_numpy(rs.normal(0, 0.3, N).astype(np.float32))
        ret = torch.from_numpy(rs.normal(0, 1, N).astype(np.float32))
        adv = torch.from_numpy(rs.normal(0.3, 2.0, N).astype(np.float32))
        res = m.loss(cond, prev, nxt, kinds, ret, oldv, adv.clone(), oldlp, use_bc_loss=False, reward_horizon=rh)
        (res[0] + 0.5 * res[2]).backward()
        out.update({f"{cname}_rgb": rgb, f"{cname}_state": state, f"{cname}_prev": prev, f"{cname}_next": nxt,
                    f"{cname}_kinds": kinds, f"{cname}_returns": ret, f"{cname}_oldvalues": oldv, f"{cname}_adv": adv,
                    f"{cname}_oldlogprobs": oldlp, f"{cname}_reward_horizon": rh,
                    f"{cname}_stats": np.array([res[0].item(), float(res[1]), res[2].item(), res[3], res[4], res[5],
                                                float(res[6]), res[7]], dtype=np.float64)})
        for k, p in m.actor_ft.named_parameters():
            put_grad(out, f"{cname}_gactor_{k}", p.grad)
        for k, p in m.critic.named_parameters():
            put_grad(out, f"{cname}_gcritic_{k}", p.grad)
    save("g21_vision_c5", **out)


# ---------------------------------------------------------------- G18 Gaussian policy on pixel observations
from make_golden_cases import VIS_GAUSS_CASES  # noqa: E402


def vis_gauss_specs(cname):
    vname, tkw, kw = VIS_GAUSS_CASES[cname]
    v = O.VisSpec(**VIS_SPECS[vname])
    cd = v.feat_dim + v.prop_dim
    trunk = O.NetSpec("gaussian", cond_dim=cd, residual=True, **tkw)
    critic = O.NetSpec("critic", cond_dim=cd, mlp_dims=[256, 256, 256], activation="Mish", residual=True)
    return v, trunk, critic, kw


def g18_vision_gaussian():
    """PPO_Gaussian over Gaussian_VisionMLP (model/common/mlp_gaussian.py:112-281) + ViTCritic: sampling with recorded noise,
    get_logprobs, the loss 8-tuple and every gradient (both encoders, both trunks, the learned logvar)."""
    from dppo.model.common.mlp_gaussian import Gaussian_VisionMLP
    from dppo.model.rl.gaussian_ppo import PPO_Gaussian
    out = {}
    rs = np.random.RandomState(1800)
    N = 20
    for cname in VIS_GAUSS_CASES:
        v, a, c, kw = vis_gauss_specs(cname)
        actor = Gaussian_VisionMLP(backbone=ref_vit(v), action_dim=a.action_dim, horizon_steps=a.horizon_steps, cond_dim=v.prop_dim,
                                   img_cond_steps=v.in_ch // 3, mlp_dims=list(a.mlp_dims), activation_type=a.activation,
                                   residual_style=True, fixed_std=kw["fixed_std"], learn_fixed_std=kw["learn_fixed_std"],
                                   std_min=kw["std_min"], std_max=kw["std_max"], spatial_emb=v.spatial_emb, num_img=v.num_img,
                                   augment=False)
        sd = dict(O.vision_init_params(v, a, 71))
        sd["logvar_min"], sd["logvar_max"] = actor.logvar_min.data.clone(), actor.logvar_max.data.clone()
        if kw["learn_fixed_std"]:
            sd["logvar"] = gauss_logvar(a, kw, 73)
        actor.load_state_dict(sd, strict=True)
        m = PPO_Gaussian(actor=actor, critic=ref_vit_critic(v, c, O.vision_init_params(v, c, 72)), horizon_steps=a.horizon_steps,
                         device="cpu", clip_ploss_coef=kw["clip_ploss_coef"], clip_vloss_coef=kw.get("clip_vloss_coef"),
                         norm_adv=True, randn_clip_value=kw["randn_clip_value"])
        rgb, state = vis_inputs(rs, v, N)
        cond = vis_cond(rgb, state)
        noise = torch.from_numpy(rs.randn(N, a.horizon_steps * a.action_dim).astype(np.float32))
        normal0 = torch.normal
        torch.normal = lambda loc, scale, *a_, **k_: loc + scale * noise
        try:
            actions = m(cond=cond, deterministic=False)
        finally:
            torch.normal = normal0
        with torch.no_grad():
            lp, ent, std = m.get_logprobs(cond, actions)
            oldv = m.critic(cond).view(-1) + torch.from_numpy(rs.normal(0, 0.3, N).astype(np.float32))
        oldlp = lp + torch.from_numpy(rs.normal(0, 0.02, N).astype(np.float32))
        ret = torch.from_numpy(rs.normal(0, 1, N).astype(np.float32))
        adv = torch.from_numpy(rs.normal(0.3, 2.0, N).astype(np.float32))
        res = m.loss(cond, actions, ret, oldv, adv.clone(), oldlp, use_bc_loss=False)
        (res[0] + 0.01 * res[1] + 0.5 * res[2]).backward()
        out.update({f"{cname}_rgb": rgb, f"{cname}_state": state, f"{cname}_noise": noise, f"{cname}_actions": actions,
                    f"{cname}_logprobs": lp, f"{cname}_oldlogprobs": oldlp, f"{cname}_oldvalues": oldv, f"{cname}_returns": ret,
                    f"{cname}_adv": adv,
                    f"{cname}_stats": np.array([res[0].item(), res[1].item(), res[2].item(), res[3], res[4], res[5],
                                                float(res[6]), res[7]], dtype=np.float64)})
        for k, p in m.actor_ft.named_parameters():
            if p.grad is not None:
                put_grad(out, f"{cname}_gactor_{k}", p.grad)
        for k, p in m.critic.named_parameters():
            put_grad(out, f"{cname}_gcritic_{k}", p.grad)
    save("g18_vision_gaussian", **out)


# ---------------------------------------------------------------- G19 plain (non-residual) MLP trunks
from make_golden_cases import PLAIN_CASES  # noqa: E402


def g19_plain_mlp():
    """DiffusionMLP / CriticObs with residual_style=False (model/common/mlp.py:27-81): forward, a K-step chain + log-probs,
    PPODiffusion.loss with every gradient, and the supervised loss with every gradient."""
    out = {}
    rs = np.random.RandomState(1900)
    N = 48
    for cname, (sname, kw) in PLAIN_CASES.items():
        a, c = specs(sname)
        kw2 = dict(kw, eta=EtaFixed(base_eta=1.0)) if kw.get("use_ddim") else kw
        m = ref_model(a, c, 31, gamma_denoising=0.99, randn_clip_value=3, **kw2)
        Kft = kw["ft_denoising_steps"]
        n_steps = kw["ddim_steps"] if kw.get("use_ddim") else kw["denoising_steps"]
        state = torch.from_numpy(rs.uniform(-1, 1, size=(N, 1, a.cond_dim)).astype(np.float32))
        noise = torch.from_numpy(rs.randn(n_steps + 1, N, a.horizon_steps, a.action_dim).astype(np.float32))
        with recorded_noise(list(noise)):
            smp = m(cond={"state": state}, deterministic=False, return_chain=True)
        chains = smp.chains
        x = torch.from_numpy(rs.randn(N, a.horizon_steps, a.action_dim).astype(np.float32))
        t = torch.from_numpy(rs.randint(0, 20, size=(N,)).astype(np.int64))
        with torch.no_grad():
            eps = m.actor_ft(x, t, cond={"state": state})
            val = m.critic({"state": state})
            lp_all = m.get_logprobs({"state": state}, chains)
        kinds = torch.from_numpy(rs.randint(0, Kft, size=(N,)).astype(np.int64))
        rows = torch.arange(N)
        prev, nxt = chains[rows, kinds], chains[rows, kinds + 1]
        oldlp = lp_all.reshape(N, Kft, a.horizon_steps, a.action_dim)[rows, kinds] + torch.from_numpy(
            rs.normal(0, 0.02, size=(N, a.horizon_steps, a.action_dim)).astype(np.float32))
        oldv = val.view(-1) + torch.from_numpy(rs.normal(0, 0.3, N).astype(np.float32))
        ret = torch.from_numpy(rs.normal(0, 1, N).astype(np.float32))
        adv = torch.from_numpy(rs.normal(0.3, 2.0, N).astype(np.float32))
        res = m.loss({"state": state}, prev, nxt, kinds, ret, oldv, adv.clone(), oldlp, use_bc_loss=False, reward_horizon=4)
        (res[0] + 0.5 * res[2]).backward()
        out.update({f"{cname}_state": state, f"{cname}_noise": noise, f"{cname}_traj": smp.trajectories, f"{cname}_chains": chains,
                    f"{cname}_logprobs": lp_all, f"{cname}_x": x, f"{cname}_t": t, f"{cname}_eps": eps, f"{cname}_value": val,
                    f"{cname}_prev": prev, f"{cname}_next": nxt, f"{cname}_kinds": kinds, f"{cname}_returns": ret,
                    f"{cname}_oldvalues": oldv, f"{cname}_adv": adv, f"{cname}_oldlogprobs": oldlp,
                    f"{cname}_stats": np.array([res[0].item(), float(res[1]), res[2].item(), res[3], res[4], res[5],
                                                float(res[6]), res[7]], dtype=np.float64)})
        for k, p in m.actor_ft.named_parameters():
            put_grad(out, f"{cname}_gactor_{k}", p.grad)
        for k, p in m.critic.named_parameters():
            put_grad(out, f"{cname}_gcritic_{k}", p.grad)
        # supervised loss on the base network
        K = kw["denoising_steps"]
        net = m.network
        for p in net.parameters():
            p.requires_grad_(True)
        x0 = torch.from_numpy(rs.uniform(-1, 1, size=(N, a.horizon_steps, a.action_dim)).astype(np.float32))
        tm = torch.from_numpy(rs.randint(0, K, size=(N,)).astype(np.int64))
        nz = torch.from_numpy(rs.randn(N, a.horizon_steps, a.action_dim).astype(np.float32))
        with recorded_noise([nz]):
            loss = m.p_losses(x0, {"state": state}, tm)
        loss.backward()
        out.update({f"{cname}_mse_x0": x0, f"{cname}_mse_t": tm, f"{cname}_mse_noise": nz, f"{cname}_mse_loss": np.float64(loss.item())})
        for k, p in net.named_parameters():
            put_grad(out, f"{cname}_mse_g_{k}", p.grad)
    save("g19_plain_mlp", **out)


# ---------------------------------------------------------------- G20 mixture-of-Gaussians PPO
from make_golden_cases import GMM_CASES  # noqa: E402


def gmm_logvar(Da, M, fixed_std, seed):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((np.log(fixed_std ** 2) + rs.uniform(-0.4, 0.4, size=Da * M)).astype(np.float32))


def g20_gmm():
    """PPO_GMM (model/rl/gmm_ppo.py) over GMM_MLP (model/common/mlp_gmm.py) + CriticObs: sampling with recorded component and
    noise draws, get_logprobs, the loss 8-tuple and every gradient of pg + 0.01 entropy_loss + 0.5 v."""
    from dppo.model.common.mlp_gmm import GMM_MLP
    from dppo.model.rl.gmm_ppo import PPO_GMM
    out = {}
    rs = np.random.RandomState(2000)
    N = 40
    for cname, (cond, tkw, Ta, Da, gkw, cdims) in GMM_CASES.items():
        M = gkw["num_modes"]
        ms, ws = O.gmm_specs(cond, tkw["mlp_dims"], tkw["activation"], tkw["residual"], Da, Ta, M)
        c = O.NetSpec("critic", cond_dim=cond, mlp_dims=cdims, activation="Mish", residual=True)
        actor = GMM_MLP(action_dim=Da, horizon_steps=Ta, cond_dim=cond, mlp_dims=list(tkw["mlp_dims"]), num_modes=M,
                        activation_type=tkw["activation"], residual_style=tkw["residual"], fixed_std=gkw["fixed_std"],
                        learn_fixed_std=gkw["learn_fixed_std"], std_min=gkw["std_min"], std_max=gkw["std_max"])
        sd = dict(O.gmm_init_params(ms, ws, 71))
        sd["logvar_min"], sd["logvar_max"] = actor.logvar_min.data.clone(), actor.logvar_max.data.clone()
        if gkw["learn_fixed_std"]:
            sd["logvar"] = gmm_logvar(Da, M, gkw["fixed_std"], 73)
        actor.load_state_dict(sd, strict=True)
        m = PPO_GMM(actor=actor, critic=ref_critic(c, O.init_params(c, 72)), horizon_steps=Ta, device="cpu",
                    clip_ploss_coef=gkw["clip_ploss_coef"], clip_vloss_coef=gkw.get("clip_vloss_coef"), norm_adv=True)
        state = torch.from_numpy(rs.uniform(-1, 1, size=(N, 1, cond)).astype(np.float32))
        modes = torch.from_numpy(rs.randint(0, M, size=(N,)).astype(np.int64))
        noise_all = torch.from_numpy(rs.randn(N, M, Ta * Da).astype(np.float32))
        normal0, multi0 = torch.normal, torch.multinomial
        torch.normal = lambda loc, scale, *a_, **k_: loc + scale * noise_all
        torch.multinomial = lambda probs, n, repl=True, **k_: modes.view(-1, 1)
        try:
            actions = m(cond={"state": state}, deterministic=False)
        finally:
            torch.normal, torch.multinomial = normal0, multi0
        with torch.no_grad():
            lp, ent, std = m.get_logprobs({"state": state}, actions)
            oldv = m.critic({"state": state}).view(-1) + torch.from_numpy(rs.normal(0, 0.3, N).astype(np.float32))
        oldlp = lp + torch.from_numpy(rs.normal(0, 0.05, N).astype(np.float32))
        ret = torch.from_numpy(rs.normal(0, 1, N).astype(np.float32))
        adv = torch.from_numpy(rs.normal(0.3, 2.0, N).astype(np.float32))
        res = m.loss({"state": state}, actions, ret, oldv, adv.clone(), oldlp)
        (res[0] + 0.01 * res[1] + 0.5 * res[2]).backward()
        out.update({f"{cname}_state": state, f"{cname}_modes": modes, f"{cname}_noise": noise_all[torch.arange(N), modes],
                    f"{cname}_actions": actions, f"{cname}_logprobs": lp, f"{cname}_oldlogprobs": oldlp, f"{cname}_oldvalues": oldv,
                    f"{cname}_returns": ret, f"{cname}_adv": adv,
                    f"{cname}_stats": np.array([res[0].item(), res[1].item(), res[2].item(), res[3], res[4], res[5],
                                                float(res[6]), res[7]], dtype=np.float64)})
        for k, p in m.actor_ft.named_parameters():
            if p.grad is not None:
                put_grad(out, f"{cname}_gactor_{k}", p.grad)
        for k, p in m.critic.named_parameters():
            put_grad(out, f"{cname}_gcritic_{k}", p.grad)
    save("g20_gmm", **out)


# ---------------------------------------------------------------- G10 LR schedule trace
from make_golden_cases import SCHED_CASES  # noqa: E402


def g10_scheduler():
    """Learning rate after construction and after every step() of the reference's CosineAnnealingWarmupRestarts
    (util/scheduler.py:32-147) driving a torch.optim.AdamW, as the agents use it (step() without an epoch)."""
    from dppo.util.scheduler import CosineAnnealingWarmupRestarts
    out = {}
    for name, (kw, n, lr0) in SCHED_CASES.items():
        p = torch.nn.Parameter(torch.zeros(3))
        opt = torch.optim.AdamW([p], lr=lr0)
        sch = CosineAnnealingWarmupRestarts(opt, **kw)
        trace = [opt.param_groups[0]["lr"]]
        for _ in range(n):
            opt.step()
            sch.step()
            trace.append(opt.param_groups[0]["lr"])
        out[name] = np.array(trace, dtype=np.float64)
    save("g10_scheduler", **out)


if __name__ == "__main__":
    only = sys.argv[1:]  # e.g. `make_golden.py g8_bc` regenerates one file (each generator owns its RNG stream)
    for fn in (g1_tables, g2_forward, g3_g4_chains, g5_loss, g6_reward_scaler, g7_adamw, g8_bc, g9_denoise_mse, g10_scheduler, g11_eval, g12_gaussian, g13_unet, g14_unet_loss, g15_unet_dim40, g16_vision,
               g17_vision_loss, g18_vision_gaussian, g19_plain_mlp, g20_gmm, g21_vision_c5):
        if not only or fn.__name__ in only:
            fn()
