#!/usr/bin/env python3
"""Micro-benchmark + correctness check of the bare MFMA GEMMs (dppo_gemm_nt_raw / dppo_gemm_tn_raw) at the PPO
update's shapes.  Variants are timed interleaved in ONE process on random data (cdna_hip_programming.md rule 24/25).

    python tools/gemm_bench.py [--M 50000] [--rounds 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dppo_amd import hip  # noqa: E402


def time_ms(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=50000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tn-only", action="store_true")
    ap.add_argument("--variants", default="", help="comma list of gemm_tn variants (tuning knob 5); default all")
    ap.add_argument("--prec", default="", help="bf16 or fp32 only")
    ap.add_argument("--shapes", default="512,256", help="comma list of square gemm_tn output sizes")
    args = ap.parse_args()
    lib = hip.load()
    dev = "cuda:0"
    M = args.M
    for prec, name, dt in ((hip.PREC_BF16, "bf16", torch.bfloat16), (hip.PREC_F32, "fp32", torch.float32)):
        if args.prec and args.prec != name:
            continue
        for (N, K) in (() if args.tn_only else ((512, 512), (256, 256), (512, 64))):
            X = torch.randn(M, K, device=dev).to(dt).contiguous()
            W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt).contiguous()
            b = torch.randn(N, device=dev)
            out = torch.empty(M, N, device=dev)
            oe = torch.empty(M, N, device=dev, dtype=dt)
            ref = X.float() @ W.float().t() + b

            def run():
                hip.check(lib.dppo_gemm_nt_raw(prec, X.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, out.data_ptr(),
                                               oe.data_ptr(), N, hip.ACT_RELU, hip.stream()), "gemm_nt_raw")
            res = {}
            for variant in (0, 1):
                lib.dppo_tune_set(0, variant)
                run()
                torch.cuda.synchronize()
                err = (out - ref).abs().max().item()
                err_e = (oe.float() - torch.relu(ref)).abs().max().item()
                res[variant] = [err, err_e, []]
            for _ in range(args.rounds):
                for variant in (0, 1):
                    lib.dppo_tune_set(0, variant)
                    res[variant][2].append(time_ms(run, args.iters))
            for variant in (0, 1):
                err, err_e, ts = res[variant]
                ts = sorted(ts)
                tf = 2.0 * M * N * K / (ts[len(ts) // 2] * 1e-3) / 1e12
                print(f"gemm_nt {name} M={M} N={N} K={K} staging={'dma' if variant else 'reg'}: median {ts[len(ts)//2]*1e3:8.1f} us "
                      f"min {ts[0]*1e3:8.1f} us  {tf:7.1f} TFLOP/s  max|err| f32-out {err:.2e} elem-out {err_e:.2e}", flush=True)
            lib.dppo_tune_set(0, 1)
        # weight-gradient GEMM
        for (N1, N2) in [(int(x), int(x)) for x in args.shapes.split(",")]:
            A = torch.randn(M, N1, device=dev).to(dt).contiguous()
            B = torch.randn(M, N2, device=dev).to(dt).contiguous()
            rps = (M // 32 + 63) // 64 * 64
            splits = (M + rps - 1) // rps
            slab = torch.empty(splits * N1 * N2, device=dev)
            Cc = torch.empty(N1, N2, device=dev)
            ref = A.float().t() @ B.float()

            def run_tn():
                hip.check(lib.dppo_gemm_tn_raw(prec, A.data_ptr(), N1, N1, B.data_ptr(), N2, N2, M, rps, slab.data_ptr(),
                                               Cc.data_ptr(), hip.stream()), "gemm_tn_raw")
            names = {0: "reg-staged 128x128", 1: "dma 128x128 x3", 2: "dma 128x128 x2", 3: "dma 256x128 x3", 4: "dma 128x128 x4",
                     5: "dma 256x128 w128x64 x2k1", 6: "dma 256x128 w128x64 x3k1", 7: "dma 256x256 8w x2",
                     8: "dma 256x128 w128x64 x2"}
            for variant in (sorted(names) if not args.variants else [int(v) for v in args.variants.split(",")]):
                lib.dppo_tune_set(5, variant)
                Cc.zero_()
                run_tn()
                torch.cuda.synchronize()
                err = ((Cc - ref).abs().max() / ref.abs().max()).item()
                ts = sorted(time_ms(run_tn, args.iters) for _ in range(args.rounds))
                tf = 2.0 * M * N1 * N2 / (ts[len(ts) // 2] * 1e-3) / 1e12
                print(f"gemm_tn {name} M={M} N1={N1} N2={N2} splits={splits} {names[variant]:26s}: median "
                      f"{ts[len(ts)//2]*1e3:8.1f} us  {tf:7.1f} TFLOP/s (incl. slab reduce)  rel err {err:.2e}", flush=True)
            lib.dppo_tune_set(5, 0)


if __name__ == "__main__":
    main()
