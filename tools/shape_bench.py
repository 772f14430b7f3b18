#!/usr/bin/env python3
"""Update-step and sampler time at the network shapes of the reference's other shipped cfgs (parity-test cases in
bench.py's terms, not bench lines): timing only, random weights, synthetic rollout.

    python tools/shape_bench.py [--shapes hopper,can,square,transport,furniture] [--prec bf16] [--batch 7500]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dppo_amd.model.common.critic import CriticObs  # noqa: E402
from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion  # noqa: E402
from dppo_amd.model.diffusion.eta import EtaFixed  # noqa: E402
from dppo_amd.model.diffusion.mlp_diffusion import DiffusionMLP  # noqa: E402
from dppo_amd.util.optim import FlatAdamW, step_and_repack  # noqa: E402

# (obs, act, Ta, actor dims, act fn, LN, cond_mlp, time_dim, critic dims, K, Kft, ddim, n_envs, batch) from cfg/*/finetune/*/ft_ppo_diffusion_mlp.yaml
SHAPES = {
    "hopper": dict(obs=11, act=3, ta=4, dims=[512] * 3, fn="ReLU", ln=False, cm=None, td=16, cdims=[256] * 3, K=20, Kft=10, ddim=False, envs=512, batch=50000),
    "halfcheetah": dict(obs=17, act=6, ta=4, dims=[512] * 3, fn="ReLU", ln=False, cm=None, td=16, cdims=[256] * 3, K=20, Kft=10, ddim=False, envs=512, batch=50000),
    "can": dict(obs=23, act=7, ta=4, dims=[512] * 3, fn="Mish", ln=False, cm=None, td=16, cdims=[256] * 3, K=20, Kft=10, ddim=False, envs=256, batch=7500),
    "square": dict(obs=23, act=7, ta=4, dims=[1024] * 3, fn="Mish", ln=False, cm=[512, 64], td=32, cdims=[256] * 3, K=20, Kft=10, ddim=False, envs=256, batch=10000),
    "transport": dict(obs=59, act=14, ta=8, dims=[1024] * 3, fn="Mish", ln=False, cm=None, td=32, cdims=[256] * 3, K=20, Kft=10, ddim=False, envs=256, batch=10000),
    "furniture": dict(obs=58, act=10, ta=8, dims=[1024] * 7, fn="Mish", ln=True, cm=[512, 64], td=32, cdims=[512] * 3, K=100, Kft=5, ddim=True, envs=1000, batch=17600),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--graph", action="store_true", help="replay the update as captured hipGraphs")
    ap.add_argument("--tune", default="", help="dppo_tune_set knobs, e.g. 1=0 (layered GEMMs instead of the fused row-tile kernels)")
    args = ap.parse_args()
    if args.tune:
        from dppo_amd import hip
        for kv in args.tune.split(","):
            k, v = kv.split("=")
            hip.check(hip.load().dppo_tune_set(int(k), int(v)), "dppo_tune_set")
    dev = torch.device("cuda", 0)
    for name in args.shapes.split(","):
        c = SHAPES[name]
        torch.manual_seed(0)
        actor = DiffusionMLP(action_dim=c["act"], horizon_steps=c["ta"], cond_dim=c["obs"], time_dim=c["td"], mlp_dims=c["dims"],
                             cond_mlp_dims=c["cm"], activation_type=c["fn"], use_layernorm=c["ln"], residual_style=True,
                             precision=args.prec)
        critic = CriticObs(cond_dim=c["obs"], mlp_dims=c["cdims"], activation_type="Mish", residual_style=True, precision=args.prec)
        kw = dict(use_ddim=True, ddim_steps=c["Kft"], eta=EtaFixed(base_eta=1.0)) if c["ddim"] else {}
        m = PPODiffusion(actor=actor, critic=critic, ft_denoising_steps=c["Kft"], horizon_steps=c["ta"], obs_dim=c["obs"],
                         action_dim=c["act"], denoising_steps=c["K"], device=str(dev), gamma_denoising=0.99, clip_ploss_coef=0.01,
                         randn_clip_value=3, min_sampling_denoising_std=0.1, min_logprob_denoising_std=0.1, **kw)
        E, Kft, AF, N = c["envs"], c["Kft"], c["ta"] * c["act"], c["batch"]
        S = max(1, (N + E * Kft - 1) // (E * Kft)) + 1
        R = S * E
        obs = torch.rand(R, 1, c["obs"], device=dev) * 2 - 1
        chains = torch.cat([m(cond={"state": obs[i * E:(i + 1) * E]}).chains for i in range(S)]).reshape(R, Kft + 1, AF)
        logp = torch.cat([m.get_logprobs({"state": obs[i * E:(i + 1) * E]}, chains[i * E:(i + 1) * E].reshape(E, Kft + 1, c["ta"], c["act"]))
                          for i in range(S)]).reshape(R, Kft, AF)
        val = torch.cat([m.critic({"state": obs[i * E:(i + 1) * E]}) for i in range(S)]).reshape(R)
        ret, adv = val + torch.randn(R, device=dev), torch.randn(R, device=dev)
        inds = torch.randperm(R * Kft, device=dev)[:N].contiguous()
        oa = FlatAdamW(m.actor_ft.flat_params(), lr=1e-5, weight_decay=0.0)
        oc = FlatAdamW(m.critic.flat_params(), lr=1e-4, weight_decay=0.0)

        graphed = None
        if args.graph:
            from dppo_amd.parallel import DataParallel
            from dppo_amd.util.graphed import GraphedUpdate
            graphed = GraphedUpdate(m, oa, oc, DataParallel(m, 1), (obs.reshape(R, -1), chains, ret, val, adv, logp), N, c["ta"])

        def update():
            if graphed is not None:
                graphed.step(inds)
                return
            m.ppo_update(obs.reshape(R, -1), chains, ret, val, adv, logp, inds, reward_horizon=c["ta"])
            step_and_repack(m, oa, oc)

        def sample():
            m(cond={"state": obs[:E]})

        out = []
        for fn in (update, sample):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / args.steps)
        npar = m.actor_ft.flat_params().numel()
        print(f"{name:10s} actor {npar / 1e6:5.2f} M params  update N={N:6d}: {out[0] * 1e3:7.3f} ms = {N / out[0] / 1e6:6.2f} M samples/s   "
              f"sampler B={E:5d} K={c['Kft'] if c['ddim'] else c['K']:3d}: {out[1] * 1e3:7.3f} ms = {E * c['ta'] / out[1] / 1e6:6.2f} M env-steps/s",
              flush=True)


if __name__ == "__main__":
    main()
