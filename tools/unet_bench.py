#!/usr/bin/env python3
"""Timing of the conv-denoiser path (state-obs Unet1D of the shipped robomimic / furniture cfgs): one forward, one K-step
sampling call, one log-prob pass; eager launches and the sampling call replayed from a hipGraph.
    python tools/unet_bench.py [--prec bf16] [--envs 256]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dppo_oracle as O  # noqa: E402  (seeded weights only)
from tests.golden.make_golden_cases import UNET_SPECS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--envs", type=int, default=256)
    args = ap.parse_args()
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    from dppo_amd.model.diffusion.eta import EtaFixed
    from dppo_amd.model.diffusion.unet import Unet1D
    dev = "cuda:0"
    for name, kw_model in (("unet_square", dict(denoising_steps=20, ft_denoising_steps=10)),
                           ("unet_furniture", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5))):
        u = O.UnetSpec(**UNET_SPECS[name])
        actor = Unet1D(action_dim=u.action_dim, cond_dim=u.cond_dim, diffusion_step_embed_dim=u.diffusion_step_embed_dim,
                       dim=u.dim, dim_mults=list(u.dim_mults), kernel_size=u.kernel_size, n_groups=u.n_groups,
                       cond_predict_scale=u.cond_predict_scale, horizon_steps=u.horizon_steps, precision=args.prec)
        actor.load_state_dict(O.unet_init_params(u, 1))
        critic = CriticObs(cond_dim=u.cond_dim, mlp_dims=[256, 256, 256], residual_style=True, precision=args.prec)
        if kw_model.get("use_ddim"):
            kw_model = dict(kw_model, eta=EtaFixed(base_eta=1.0))
        m = PPODiffusion(actor=actor, critic=critic, horizon_steps=u.horizon_steps, obs_dim=u.cond_dim,
                         action_dim=u.action_dim, device=dev, gamma_denoising=0.99, clip_ploss_coef=0.01, randn_clip_value=3,
                         **kw_model)
        B = args.envs
        obs = torch.rand(B, 1, u.cond_dim, device=dev) * 2 - 1
        x = torch.randn(B, u.horizon_steps, u.action_dim, device=dev)
        t = torch.randint(0, 20, (B,), device=dev)

        def timeit(fn, n=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        fwd = timeit(lambda: m.actor_ft(x, t, {"state": obs}))
        smp = timeit(lambda: m(cond={"state": obs}))
        chains = m(cond={"state": obs}).chains
        lp = timeit(lambda: m.get_logprobs({"state": obs}, chains))
        # the same sampling call captured once and replayed (the host loop's ~60 launches per denoising step become one graph)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m(cond={"state": obs})
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            out = m(cond={"state": obs})
        graph = timeit(g.replay)
        n_steps = kw_model.get("ddim_steps") or kw_model["denoising_steps"]
        print(f"{name:16s} {args.prec} B={B}: forward {fwd:.3f} ms | sampling call ({n_steps} steps) eager {smp:.2f} ms, "
              f"graph replay {graph:.2f} ms = {B * 4 / graph * 1e3 / 1e6:.3f} M env-steps/s (act_steps 4) | "
              f"log-probs ({B} x {m.ft_denoising_steps} rows) {lp:.2f} ms")


if __name__ == "__main__":
    main()
