#!/usr/bin/env python3
"""Per-kernel timing of the fused row-tile MLP kernels at the PPO update's shapes (HIP-event probe inside the library).

    python tools/fused_bench.py [--rows 50000] [--prec bf16] [--iters 10]

Prints, for the hopper nets: inference forward (log-prob precompute, no activation stores), training forward and
backward (PPO update, serial streams so the probe sees each kernel alone), with the algorithmic TFLOP/s of each.
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dppo_amd import hip  # noqa: E402


def probe(lib, kid, fn, iters):
    fn()
    torch.cuda.synchronize()
    hip.check(lib.dppo_probe_arm(kid, 64 * iters), "arm")
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    ms, cnt, fl = C.c_double(), C.c_int(), C.c_double()
    hip.check(lib.dppo_probe_collect(C.byref(ms), C.byref(cnt), C.byref(fl)), "collect")
    n = max(cnt.value, 1)
    return ms.value / iters, cnt.value / iters, fl.value / iters / (ms.value / iters * 1e-3) / 1e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--stamps", action="store_true",
                    help="phase stamps of workgroup 0 (needs DPPO_HIP_LIB=.../libdppo_hip_stamps.so, built by "
                         "DPPO_STAMPS=1 dppo_amd/csrc/build.sh)")
    args = ap.parse_args()
    lib = hip.load()
    dev = torch.device("cuda", 0)
    model = bench.build_model(str(dev), args.prec)
    gen = torch.Generator(device=dev).manual_seed(1)
    KFT, AF = bench.KFT, bench.TA * bench.ACT_DIM
    R = args.rows // KFT
    obs_k, chains_k, ret_k, val_k, adv_k, logp_k = bench.make_rollout(model, R, 1, dev, gen)
    st = {"state": obs_k.reshape(R, 1, bench.OBS_DIM)}
    ch = chains_k.reshape(R, KFT + 1, bench.TA, bench.ACT_DIM)
    inds = torch.randperm(R * KFT, device=dev, generator=gen)[:args.rows].contiguous()

    def infer():
        model.get_logprobs(st, ch)

    def update():
        model.ppo_update(obs_k, chains_k, ret_k, val_k, adv_k, logp_k, inds, reward_horizon=bench.ACT_STEPS)

    def crit():
        model.critic(st)

    def stamps(title, fn):
        import numpy as np
        fn()
        torch.cuda.synchronize()
        buf = np.zeros((8, 32), dtype=np.uint64)
        raw = C.CDLL(hip.LIB_PATH)
        raw.dppo_debug_stamps.argtypes = [C.c_void_p]
        assert raw.dppo_debug_stamps(buf.ctypes.data) == 0
        fwd = ["start", "tile loaded", "L0 run", "L0 emit", "barrier", "l1 run", "l1 emit", "barrier", "l2 run",
               "l2 emit", "barrier", "out layer", "barrier", "reduce+store", "barrier"]
        bwd = ["start", "tile loaded", "dh run", "dh emit", "dh colsum", "barrier", "W2T run", "x act' + emit",
               "dz1 colsum", "barrier", "W1T run", "x act' + emit", "dh colsum", "barrier"]
        for label, first, names in (("forward kernel", 0, fwd), ("backward kernel", 16, bwd)):
            t = buf[:, first:first + len(names)].astype(np.int64)
            if t.max() == 0:
                continue
            t0 = t[:, 0].min()
            print(f"--- {title}, {label}: cycles since tile start, per wave (workgroup 0, its 2nd tile; last block only)")
            for i, n in enumerate(names):
                print(f"{n:14s} " + " ".join(f"{int(v - t0):7d}" for v in t[:, i]))

    lib.dppo_tune_set(2, 0)
    rows = R * KFT
    if args.stamps:
        stamps("inference fwd actor", infer)
        stamps("training fwd (last launch = actor_ft)", update)
        return
    print(f"rows={rows} prec={args.prec}")
    for name, kid, fn in (("inference fwd  (actor, get_logprobs)", 3, infer), ("inference fwd  (critic, R rows)", 3, crit),
                          ("training  fwd  (actor + critic)", 3, update), ("training  bwd  (actor + critic)", 4, update),
                          ("weight grads   (gemm_tn, 8 launches)", 2, update)):
        ms, launches, tf = probe(lib, kid, fn, args.iters)
        print(f"{name:40s} {ms * 1e3:8.1f} us/step  {launches:4.1f} launches/step  {tf:7.1f} TFLOP/s")
    lib.dppo_tune_set(2, 1)


if __name__ == "__main__":
    main()
