"""Launcher: ``python -m dppo_amd.script.run --config-dir=DIR --config-name=NAME [key=value ...]``.

Same contract as the reference's ``script/run.py:43-87`` (resolve the cfg, ``cls = get_class(cfg._target_)``,
``cls(cfg).run()``) without hydra; checkpoint / dataset downloads (gdown) are not attempted -- there is no network.
Under ``torchrun`` (WORLD_SIZE > 1) it initialises RCCL and binds one process to one GPU.
"""
import argparse
import logging
import os
import sys

import torch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-dir", required=True)
    ap.add_argument("--config-name", required=True)
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(name)s] %(message)s")
    from dppo_amd.cfg.loader import get_class, load_config
    name = args.config_name if args.config_name.endswith((".yaml", ".yml")) else args.config_name + ".yaml"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    overrides = list(args.overrides)
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        overrides.append(f"device=cuda:{local}")
    cfg = load_config(os.path.join(args.config_dir, name), overrides)
    if "cuda" in str(cfg.device):
        torch.cuda.set_device(torch.device(cfg.device))
    agent = get_class(cfg._target_)(cfg)
    agent.run()


if __name__ == "__main__":
    main(sys.argv[1:])
