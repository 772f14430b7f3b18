#!/usr/bin/env python3
"""Whole rollout loop (host + device) at the benchmark's shapes: env-steps/s of collect_rollout with one env group (the
reference's strictly alternating loop, with the pinned hand-off) vs two pipelined groups, for a given simulator cost.

    python tools/rollout_bench.py [--n-envs 512] [--n-steps 60] [--sim-us 1000]

--sim-us: host time one vectorised env.step call of ALL envs takes (busy wait added to the synthetic env, split evenly
over the groups): real simulators (MuJoCo through AsyncVectorEnv) cost milliseconds per call, the sampler 0.2-0.3 ms.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dppo_amd.env.synthetic import SyntheticVecEnv  # noqa: E402
from dppo_amd.util.rollout import GroupedVecEnv, collect_rollout  # noqa: E402


class SlowEnv(SyntheticVecEnv):
    busy_us = 0.0

    def step(self, action):
        t0 = time.perf_counter()
        out = super().step(action)
        while (time.perf_counter() - t0) * 1e6 < self.busy_us:
            pass
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-envs", type=int, default=512)
    ap.add_argument("--n-steps", type=int, default=60)
    ap.add_argument("--sim-us", type=float, nargs="*", default=[0, 500, 1000, 3000])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    m = bench.build_model(str(dev), "bf16")
    E, S = args.n_envs, args.n_steps
    AF = bench.TA * bench.ACT_DIM
    obs_buf = torch.zeros(S * E, bench.OBS_DIM, device=dev)
    chains_buf = torch.zeros(S * E, bench.KFT + 1, AF, device=dev)
    print(f"n_envs={E} n_steps={S} act_steps={bench.ACT_STEPS}; env-steps/s of the whole loop (host + device)")
    for sim in args.sim_us:
        row = []
        for G in (1, 2, 4):
            n = E // G
            envs = [SlowEnv(n, bench.OBS_DIM, bench.ACT_DIM, 1, bench.ACT_STEPS, seed=1 + g * n) for g in range(G)]
            for e in envs:
                e.busy_us = sim / G
            venv = envs[0] if G == 1 else GroupedVecEnv(envs)
            obs = venv.reset_arg()
            collect_rollout(m, venv, obs, 5, bench.ACT_STEPS, obs_buf[:5 * E], chains_buf[:5 * E])  # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            collect_rollout(m, venv, obs, S, bench.ACT_STEPS, obs_buf, chains_buf)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            row.append((G, S * E * bench.ACT_STEPS / dt, dt / S * 1e3))
        print(f"  simulator {sim:6.0f} us/step: " + "   ".join(f"{G} group(s) {v / 1e6:6.2f} M ({ms:5.2f} ms/step)" for G, v, ms in row))


if __name__ == "__main__":
    main()
