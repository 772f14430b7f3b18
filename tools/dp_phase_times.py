#!/usr/bin/env python3
"""Where a data-parallel update step spends its time, phase by phase: loss forward/backward (fork .. joins), the gradient
all-reduce, AdamW + repack -- bracketed with events on the caller's stream (the library's side streams join back into it
before `ppo_update` returns), for the four combinations of knob 2 (side streams) and knob 14 (joins issued behind the
actor's weight-gradient launch).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        tools/dp_phase_times.py --share-gpu --backend gloo --steps 10 > gpurun_out/dp2_phases.json

--share-gpu puts every rank on cuda:0: a REHEARSAL of the code path on a one-GPU box (two processes time-slice one
device and gloo moves the bucket through host memory), not a scaling measurement.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--n-steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=50000)
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", local)
    from dppo_amd import hip
    from dppo_amd.parallel import DataParallel
    from dppo_amd.util.optim import FlatAdamW, step_and_repack
    lib = hip.load()
    model = bench.build_model(str(dev), "bf16")
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    torch.manual_seed(42 + rank)
    dp = DataParallel(model, world)
    ro = bench.make_rollout(model, 512, args.n_steps, dev, gen)
    R = 512 * args.n_steps
    oa = FlatAdamW(model.actor_ft.flat_params(), lr=1e-4, weight_decay=0.0)
    oc = FlatAdamW(model.critic.flat_params(), lr=1e-3, weight_decay=0.0)
    n_total = args.steps + args.warmup
    perm = torch.randperm(R * bench.KFT, device=dev, generator=gen)
    n_mb = (R * bench.KFT) // args.batch
    mbs = [perm[(i % n_mb) * args.batch:(i % n_mb + 1) * args.batch].contiguous() for i in range(n_total)]
    moments = dp.minibatch_moments(ro[4], mbs, bench.KFT)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    results = []
    for k2, k14 in ((1, 1), (1, 0), (0, 1), (0, 0)):
        hip.check(lib.dppo_tune_set(2, k2), "tune")
        hip.check(lib.dppo_tune_set(14, k14), "tune")
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_total)]
        host = []
        for i in range(n_total):
            if i == args.warmup:
                barrier()
                t0 = time.perf_counter()
            h0 = time.perf_counter()
            ev[i][0].record()
            model.ppo_update(*ro, mbs[i], reward_horizon=bench.ACT_STEPS,
                             global_moments=None if moments is None else moments[i])
            ev[i][1].record()
            h1 = time.perf_counter()
            dp.allreduce_grads()
            ev[i][2].record()
            h2 = time.perf_counter()
            step_and_repack(model, oa, oc, n_time=bench.K)
            ev[i][3].record()
            host.append((h1 - h0, h2 - h1, time.perf_counter() - h2))
        barrier()
        wall = (time.perf_counter() - t0) / args.steps * 1e3
        sel = range(args.warmup, n_total)
        avg = lambda a, b: sum(ev[i][a].elapsed_time(ev[i][b]) for i in sel) / len(sel)
        havg = lambda j: sum(host[i][j] for i in sel) / len(sel) * 1e3
        results.append({"knob2_side_streams": k2, "knob14_early_joins": k14, "wall_ms_per_step": wall,
                        "device_ms": {"loss_fwd_bwd": avg(0, 1), "allreduce": avg(1, 2), "adamw_repack": avg(2, 3),
                                      "step": avg(0, 3)},
                        "host_enqueue_ms": {"loss_fwd_bwd": havg(0), "allreduce": havg(1), "adamw_repack": havg(2)}})
    hip.check(lib.dppo_tune_set(2, 1), "tune")
    hip.check(lib.dppo_tune_set(14, 1), "tune")
    if rank == 0:
        print(json.dumps({"world": world, "backend": args.backend if world > 1 else None, "share_gpu": args.share_gpu,
                          "batch_per_rank": args.batch, "steps": args.steps, "passes": results}, indent=1))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
