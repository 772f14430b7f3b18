#!/usr/bin/env python3
"""Phase stamps of the split sampler (csrc/sampler_split.hip): where one denoising step of workgroup 0 spends its cycles.

    DPPO_STAMPS=1 bash dppo_amd/csrc/build.sh
    DPPO_HIP_LIB=$PWD/dppo_amd/lib/libdppo_hip_stamps.so python tools/sampler_stamps.py [--n-envs 512]

Prints, per wave of workgroup 0 (tile 0, member 0) at step 5, the s_memtime deltas between the phase boundaries (100 MHz
ticks x 10 ns) and the number of polling passes the exchange took; then the kernel time from HIP events.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dppo_amd import hip  # noqa: E402

NAMES = ["top", "L0+emit", "barrier1", "l1+emit", "barrier2", "out+store", "poll", "posterior", "barrier3"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-envs", type=int, default=512)
    ap.add_argument("--tune", action="append", default=[])
    args = ap.parse_args()
    lib = hip.load()
    for kv in args.tune:
        k, v = kv.split("=")
        assert lib.dppo_tune_set(int(k), int(v)) == 0
    dev = torch.device("cuda", 0)
    m = bench.build_model(str(dev), "bf16")
    st = torch.rand(args.n_envs, 1, bench.OBS_DIM, device=dev) * 2 - 1
    for _ in range(3):
        m(cond={"state": st}, deterministic=False, return_chain=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        m(cond={"state": st}, deterministic=False, return_chain=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"sampler call (n_envs={args.n_envs}): {e0.elapsed_time(e1) / n * 1e3:.1f} us per call (stream time incl. the memset)")
    if hasattr(lib, "dppo_debug_split_stamps"):
        raw = C.CDLL(lib._name)
        buf = np.zeros((4, 16), dtype=np.uint64)
        raw.dppo_debug_split_stamps.argtypes = [C.c_void_p]
        assert raw.dppo_debug_split_stamps(buf.ctypes.data) == 0
        for w in range(4):
            t = buf[w].astype(np.int64)
            d = [int(t[k] - t[k - 1]) if t[k] and t[k - 1] else None for k in range(1, 9)]
            print(f"wave {w}: " + "  ".join(f"{NAMES[k]}={d[k - 1]}" for k in range(1, 9)) +
                  f"  step={int(t[8] - t[0])} ticks  polling passes={int(t[9])}"
                  f"  | prologue={int(t[11] - t[10])}  all steps={int(t[12] - t[11])}")


if __name__ == "__main__":
    main()
