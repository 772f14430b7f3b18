#!/usr/bin/env python3
"""Largest / mean difference between the chains of the split sampler (knob 27 = 1) and of the one-workgroup kernel (0) on the
same inputs, per case of tests/test_sampler_split.py -- the numbers behind that test's stated tolerance."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dppo_amd import hip  # noqa: E402
from tests.test_hip_parity import DEV, build_model  # noqa: E402
from tests.test_sampler_split import CASES  # noqa: E402

lib = hip.load()
for case in sorted(CASES):
    sname, kw = CASES[case]
    for B in (37, 512):
        m, a, _ = build_model(sname, kw, 5, "bf16")
        gen = torch.Generator(device="cpu").manual_seed(B + 1)
        st = (torch.rand(B, 1, a.cond_dim, generator=gen) * 2 - 1).to(DEV)
        n_steps = kw.get("ddim_steps", kw["denoising_steps"])
        noise = torch.randn(n_steps + 1, B, a.horizon_steps, a.action_dim, generator=gen).to(DEV)
        out = []
        for split in (0, 1):
            lib.dppo_tune_set(27, split)
            out.append(m(cond={"state": st}, noise=noise, return_chain=True).chains.clone())
        lib.dppo_tune_set(27, 1)
        d = (out[0] - out[1]).abs()
        print(f"{case:18s} B={B:4d}  max |d| = {d.max().item():.3e}  mean |d| = {d.mean().item():.3e}  "
              f"elements differing = {(d > 0).float().mean().item():.4f}")
