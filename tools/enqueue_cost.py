#!/usr/bin/env python3
"""Host-side cost of enqueueing one PPO update step (no device sync inside the loop): if it approaches the device time
per step, the GPU idles at step boundaries.    python tools/enqueue_cost.py [--steps 30]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dppo_amd import hip  # noqa: E402
from dppo_amd.util.optim import FlatAdamW  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--tune", action="append", default=[])
    args = ap.parse_args()
    lib = hip.load()
    for kv in args.tune:
        k, v = kv.split("=")
        lib.dppo_tune_set(int(k), int(v))
    dev = torch.device("cuda", 0)
    model = bench.build_model(str(dev), "bf16")
    gen = torch.Generator(device=dev).manual_seed(1)
    R = 5000
    obs_k, chains_k, ret_k, val_k, adv_k, logp_k = bench.make_rollout(model, R, 1, dev, gen)
    inds = torch.randperm(R * bench.KFT, device=dev, generator=gen)[:50000].contiguous()
    opt_a = FlatAdamW(model.actor_ft.flat_params(), lr=1e-4, weight_decay=0.0)
    opt_c = FlatAdamW(model.critic.flat_params(), lr=1e-3, weight_decay=0.0)

    def step():
        model.ppo_update(obs_k, chains_k, ret_k, val_k, adv_k, logp_k, inds, reward_horizon=bench.ACT_STEPS)
        opt_a.step(model.actor_ft.flat_grads())
        opt_c.step(model.critic.flat_grads())
        model.actor_ft.mark_updated()
        model.critic.mark_updated()
        model.actor_ft.packed(model.prec, bench.K)
        model.critic.packed(model.prec, 0)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    host = []
    t_all = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        step()
        host.append(time.perf_counter() - t0)
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    host.sort()
    print(f"host enqueue per step: median {host[len(host)//2]*1e6:.0f} us, min {host[0]*1e6:.0f}, max {host[-1]*1e6:.0f}; "
          f"all enqueued after {t_enq*1e3:.2f} ms, device done after {t_tot*1e3:.2f} ms "
          f"({t_tot/args.steps*1e6:.0f} us per step)")


if __name__ == "__main__":
    main()
