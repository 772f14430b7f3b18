#!/usr/bin/env python3
"""Register / LDS / spill table of every kernel in one .hip source (hipcc -Rpass-analysis=kernel-resource-usage, device only).

    python tools/kernel_regs.py dppo_amd/csrc/fused.hip [filter-substring ...] [-- extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    src, filters = argv[0], argv[1:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", "--offload-device-only",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
    if os.path.basename(src) in ("sampler.hip", "sampler_split.hip", "ppo.hip", "gaussian.hip", "gmm.hip", "unet.hip"):
        cmd.insert(-4, "-ffp-contract=off")
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = txt.split("Function Name: ")[1:]
    names = [b.split()[0] for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    print(f"{'VGPR':>5} {'AGPR':>5} {'scratch':>8} {'occ':>4} {'LDS':>7} {'spill':>6}  kernel")
    for b, d in zip(blocks, dem):
        if filters and not any(f in d for f in filters):
            continue

        def g(k):
            m = re.search(re.escape(k) + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        print(f"{g('VGPRs'):5d} {g('AGPRs'):5d} {g('ScratchSize [bytes/lane]'):8d} {g('Occupancy [waves/SIMD]'):4d} "
              f"{g('LDS Size [bytes/block]'):7d} {g('VGPRs Spill'):6d}  {d[:170]}")


if __name__ == "__main__":
    main()
