#!/usr/bin/env python3
"""Two (or more) PROCESSES sampling on the same GPU at once: the split sampler's members of one process wait for each other
while the other process's workgroups hold CUs.  Each process draws 300 chains of 512 envs, compares every one with its own quiet
reference bit for bit and reads the time-out word.  Run N copies concurrently:  python tools/sampler_two_processes.py <tag> &"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_hip_parity import DEV, build_model  # noqa: E402
from tests.test_sampler_split import DDPM, timeout_word  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "0"
m, a, _ = build_model("hopper", DDPM, 11, "bf16")
B = 512
st = torch.rand(B, 1, a.cond_dim, device=DEV) * 2 - 1
noise = torch.randn(21, B, a.horizon_steps, a.action_dim, device=DEV)
ref = m(cond={"state": st}, noise=noise).chains.clone()
torch.cuda.synchronize()
time.sleep(max(0.0, float(os.environ.get("START_AT", "0")) - time.time()))  # all copies start their loops together
t0 = time.perf_counter()
bad = 0
for rep in range(300):
    c = m(cond={"state": st}, noise=noise).chains
    if rep % 10 == 9:
        bad += int(not torch.equal(c, ref))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 300
print(f"process {tag}: {dt * 1e6:.1f} us per call, mismatching calls {bad}, time-out word {timeout_word(m)}")
assert bad == 0 and timeout_word(m) == 0
