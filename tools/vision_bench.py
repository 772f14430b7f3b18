#!/usr/bin/env python3
"""Timing of the pixel path at the shipped robomimic image shapes (cfg/robomimic/finetune/square/ft_ppo_diffusion_{mlp,unet}_img.yaml):
one 96x96 camera, ViT (embed 128, 4 heads, depth 1) + SpatialEmb 128, Ta 4, Da 7, DDIM 100 -> 5 steps, all fine-tuned.
  - rollout step: encode + 5-step sampling for n_envs observations
  - update minibatch: both encoders forward with a tape, the fused PPO loss forward/backward, both encoders backward
    python tools/vision_bench.py [--prec bf16] [--envs 50] [--batch 500] [--kind mlp|unet]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(kind, prec, dev):
    from dppo_amd.model.common.critic import ViTCritic
    from dppo_amd.model.common.vit import VitEncoder, VitEncoderConfig
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    from dppo_amd.model.diffusion.eta import EtaFixed
    from dppo_amd.model.diffusion.mlp_diffusion import VisionDiffusionMLP
    from dppo_amd.model.diffusion.unet import VisionUnet1D
    bb = lambda: VitEncoder([3, 96, 96], VitEncoderConfig(), num_channel=3, img_h=96, img_w=96)
    common = dict(backbone=bb(), action_dim=7, cond_dim=9, img_cond_steps=1, spatial_emb=128, num_img=1, precision=prec)
    if kind == "mlp":
        actor = VisionDiffusionMLP(horizon_steps=4, time_dim=32, mlp_dims=[768, 768, 768], residual_style=True, **common)
    else:
        actor = VisionUnet1D(diffusion_step_embed_dim=32, dim=64, dim_mults=[1, 2], kernel_size=5, n_groups=8,
                             cond_predict_scale=True, horizon_steps=4, **common)
    critic = ViTCritic(backbone=bb(), cond_dim=9, spatial_emb=128, mlp_dims=[256, 256, 256], residual_style=True, precision=prec)
    return PPODiffusion(actor=actor, critic=critic, horizon_steps=4, obs_dim=9, action_dim=7, device=dev, gamma_denoising=0.99,
                        clip_ploss_coef=0.01, clip_ploss_coef_base=0.001, randn_clip_value=3, min_sampling_denoising_std=0.1,
                        min_logprob_denoising_std=0.1, denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                        eta=EtaFixed(base_eta=1.0), precision=prec)


# ---- algorithmic FLOP of the shipped square image cfgs (necessary evaluations only; 2 FLOP per multiply-add) -------------------
def encoder_mflop():
    """One 96 x 96 frame through VitEncoder (PatchEmbed2: Conv2d(3->128, k8, s4) + ReLU + Conv2d(128->128, k3, s2) -> 121 tokens;
    one pre-norm layer of 4 heads, MLP 128 -> 512 -> 128) and SpatialEmb (128 channel rows x (121 + 9) -> 128)
    (reference model/common/vit.py:28-200, modules.py:10-41)."""
    conv1 = 23 * 23 * 128 * (3 * 8 * 8)
    conv2 = 121 * 128 * (128 * 9)
    layer = 121 * 128 * 384 + 2 * 121 * 121 * 128 + 121 * 128 * 128 + 2 * 121 * 128 * 512
    semb = 128 * 130 * 128
    return 2.0 * (conv1 + conv2 + layer + semb) / 1e6


def trunk_mflop(kind, cond=137, Ta=4, Da=7):
    """One evaluation of the denoiser trunk on cat[feat, state] (137 wide)."""
    if kind == "mlp":  # VisionDiffusionMLP: Linear(28 + 32 + 137, 768), one residual block, Linear(768, 28); time MLP 32 -> 64 -> 32
        H, AF, td = 768, Ta * Da, 32
        return 2.0 * ((AF + td + cond) * H + 2 * H * H + H * AF + 2 * td * 2 * td) / 1e6
    # VisionUnet1D (unet.py:330-618): dim 64, mults (1, 2), kernel 5, FiLM scale + bias from [step embedding 32 | cond]
    g = 32 + cond

    def res(ci, co, T):
        return T * co * ci * 5 + T * co * co * 5 + g * 2 * co + (T * co * ci if ci != co else 0)
    f = res(Da, 64, Ta) + res(64, 64, Ta) + (Ta // 2) * 64 * 64 * 3            # level 0 + Downsample1d
    f += res(64, 128, Ta // 2) + res(128, 128, Ta // 2) + 2 * res(128, 128, Ta // 2)  # level 1, mid
    f += res(256, 64, Ta // 2) + res(64, 64, Ta // 2) + Ta * 64 * 64 * 4        # up + Upsample1d (ConvTranspose1d k4)
    f += Ta * 64 * 64 * 5 + Ta * Da * 64 + 32 * 128 * 2                          # final block, 1x1 conv, time MLP
    return 2.0 * f / 1e6


def critic_mflop(cond=137):
    return 2.0 * (cond * 256 + 2 * 256 * 256 + 256) / 1e6


def flops(kind, batch, ms_upd, n_envs, ms_smp, peak_tflops, Kft=5):
    """bench.py's `pixel` block: FLOP per PPO sample updated (both encoders and both trunks, forward + backward = 3x) and per
    env observation sampled (ONE encoder pass + Kft trunk evaluations -- the reference re-encodes inside every denoising step and
    for both networks, unet.py:573, diffusion_vpg.py:148,162), and the fractions of the dense bf16 MFMA peak they amount to."""
    e = encoder_mflop()
    upd = 3.0 * (2 * e + trunk_mflop(kind) + critic_mflop())
    smp = e + Kft * trunk_mflop(kind)
    return {"mflop_per_sample": upd, "mflop_per_chunk": smp, "encoder_mflop_per_image": e, "trunk_mflop_per_eval": trunk_mflop(kind),
            "update_frac": batch / (ms_upd * 1e-3) * upd * 1e6 / (peak_tflops * 1e12),
            "sampler_frac": n_envs / (ms_smp * 1e-3) * smp * 1e6 / (peak_tflops * 1e12)}


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--envs", type=int, default=50)
    ap.add_argument("--batch", type=int, default=500)
    ap.add_argument("--kind", default="both")
    ap.add_argument("--once", action="store_true", help="one untimed pass of each piece (for rocprofv3)")
    args = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    for kind in (("mlp", "unet") if args.kind == "both" else (args.kind,)):
        m = build(kind, args.prec, dev)
        E, N, Kft, AF = args.envs, args.batch, 5, 28
        cond_e = {"rgb": torch.randint(0, 256, (E, 1, 3, 96, 96), device=dev, dtype=torch.uint8),
                  "state": torch.rand(E, 1, 9, device=dev) * 2 - 1}
        cond_n = {"rgb": torch.randint(0, 256, (N, 1, 3, 96, 96), device=dev, dtype=torch.uint8),
                  "state": torch.rand(N, 1, 9, device=dev) * 2 - 1}
        chains = m(cond=cond_n).chains
        kinds = torch.randint(0, Kft, (N,), device=dev)
        rows = torch.arange(N, device=dev)
        pairs = torch.stack([chains.reshape(N, Kft + 1, AF)[rows, kinds], chains.reshape(N, Kft + 1, AF)[rows, kinds + 1]], 1).contiguous()
        lp = m.get_logprobs(cond_n, chains).reshape(N, Kft, AF)[rows, kinds].contiguous()
        ret, val, adv = torch.randn(N, device=dev), torch.randn(N, device=dev), torch.randn(N, device=dev)
        pieces = {
            "encode (rollout, n_envs obs)": lambda: m.actor_ft.encode_obs(cond_e),
            "rollout step: encode + 5-step sampling": lambda: m(cond=cond_e),
            "log-probs of N obs (1 encode + N*Kft rows)": lambda: m.get_logprobs(cond_n, chains),
            "encode with tape (N obs)": lambda: m.actor_ft.encode_obs(cond_n, train=True),
            "encoder backward (N obs)": lambda: m.actor_ft.vis.backward(torch.ones(N, 137, device=dev)),
            "update minibatch (2 encoders fwd+bwd, loss fwd+bwd)": lambda: m._run_ppo_vision(cond_n, pairs, ret, val, adv, lp, kinds, N, 4, None),
        }
        if args.once:
            for fn in pieces.values():
                fn()
            torch.cuda.synchronize()
            continue
        print(f"== {kind}_img  prec {args.prec}  n_envs {E}  minibatch {N}")
        for name, fn in pieces.items():
            print(f"  {name:55s} {timeit(fn):8.3f} ms")
        ms = timeit(pieces["update minibatch (2 encoders fwd+bwd, loss fwd+bwd)"])
        print(f"  -> {N / ms * 1e3:,.0f} ppo-update samples/s (pixel obs)  |  {E * 4 / timeit(pieces['rollout step: encode + 5-step sampling']) * 1e3:,.0f} env-steps/s sampling side")


if __name__ == "__main__":
    main()
