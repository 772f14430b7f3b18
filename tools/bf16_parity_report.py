#!/usr/bin/env python3
"""Measured error of the bf16 path (the benchmarked precision) against the reference's golden vectors: every G5 loss case
and both G8 behaviour-cloning cases, next to the fp32 path's error on the same vectors.  The tolerances stated in
tests/test_bf16_parity.py are these maxima with head-room.   python tools/bf16_parity_report.py > profiles/rNN_bf16_parity.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.conftest import load_golden  # noqa: E402
from tests.test_bf16_parity import bc_metrics, loss_metrics  # noqa: E402
from tests.test_hip_parity import HIP_SUPPORTED  # noqa: E402
from tests.test_oracle_golden import BC_CASES, LOSS_CASES  # noqa: E402

cache = {}


def golden(name):
    if name not in cache:
        cache[name] = load_golden(name)
    return cache[name]


cols = ("pg_abs", "v_rel", "kl_abs", "ratio_abs", "clipfrac_abs", "actor_tensor_cos", "actor_norm", "critic_tensor_cos",
        "critic_norm", "ref_pg", "ref_kl", "ref_clipfrac")
print("G5 PPODiffusion.loss vs reference goldens (N = 64 per case)")
print(f"{'case':20s} {'prec':5s} " + " ".join(f"{c:>17s}" for c in cols))
worst = {}
for case in sorted(k for k, v in LOSS_CASES.items() if v[0] in HIP_SUPPORTED):
    for prec in ("fp32", "bf16"):
        r = loss_metrics(golden, case, prec)
        print(f"{case:20s} {prec:5s} " + " ".join(f"{r[c]:17.4e}" for c in cols))
        if prec == "bf16":
            for c in cols[:9]:
                f = min if "cos" in c else max
                worst[c] = f(worst.get(c, r[c]), r[c])
print("worst bf16:", {k: float(f"{v:.4g}") for k, v in worst.items()})
print()
print("G8 BC term vs reference goldens (N = 16)")
for case in sorted(BC_CASES):
    for prec in ("fp32", "bf16"):
        r = bc_metrics(golden, case, prec)
        print(f"{case:20s} {prec:5s} " + " ".join(f"{k}={v:.4e}" for k, v in r.items()))
