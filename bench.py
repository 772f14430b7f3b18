#!/usr/bin/env python3
"""DPPO hot-path benchmark on MI355X: hopper-medium-v2 shapes, K=20 denoising steps, n_envs=512 per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both forms work for every N.  Invoked plainly (no WORLD_SIZE in the environment) with N > 1, this process is a LAUNCHER:
before importing torch or touching a GPU it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
child, forwards rank 0's one JSON line to stdout (everything else to stderr) and exits with the child's exit code.

One "step" = one pass of the hot path over one batch of synthetic input:
  (a) one K=20 DDPM sampling call for 512 envs (obs -> action chunk + denoising chain), and
  (b) one PPO minibatch update of 50,000 samples drawn from the device-resident rollout buffer: fused gather,
      actor_ft + critic forward, loss, backward, [gradient all-reduce over ranks], AdamW on both networks and
      re-packing of the updated weights for the next step.
Each timed region is EXACTLY `steps` steps bracketed by barrier + torch.cuda.synchronize(), MAX over ranks.  The update leg
is timed `--passes` (10) times over, each pass its own bracketed region of the same `steps` steps: `value` / `ms_per_step`
are the MEDIAN pass, min / max are reported beside it (`passes`), so the number survives a noisy neighbour or a slow clock
ramp.  The sampler (a 0.06 ms call) runs ten passes of its `steps` calls inside one region.  After the `warmup` steps and
before the first timed region of a leg `--spin-up` (default 300, reported as `spin_up`) more untimed calls bring the GPU's
clocks up (same count on every rank).
`value` is the headline the north star puts the target on -- PPO-update samples/s, whole job -- and the sampler's
env-steps/s is reported beside it (BASELINE.json's metric names both).  Weak scaling: per-GPU load is fixed.

Synthetic data (seed 42): random-init networks of the reference architecture, obs ~ U(-1,1),
N(0,1) noise clipped at +-3, the rollout buffer is the sampler's own chains over 500 obs batches, rewards ~ N(0,1),
terminated ~ Bernoulli(0.002), values from the random critic.

The line also carries
  `roofline`     SURVEY.md 8(d)'s figure: the update path's algorithmic FLOP rate (samples/s x 4.11 MFLOP) against the
                 dense MFMA peak of the operand dtype, the sampler's (chunks/s x 22.06 MFLOP) beside it, and as
                 `dominant_kernel` the largest kernel of the step (`gemm_tn_group_kernel`: every weight gradient of one
                 network's backward pass in one launch) timed live with HIP events on its launch stream: its operand
                 bytes per second against the HBM peak and its own MFMA fraction;
  `fp32`         the same two legs with fp32 operands (the reference's arithmetic), secondary;
  `configs`      BASELINE configs[2] (robomimic can, K=100 DDPM, Ta=8, 256 envs, minibatch 7,500) and one GPU's share of
                 configs[3] (halfcheetah, 512 envs, minibatch 50,000): the same two legs, each with its 8(d)-style fraction;
  `allreduce_ms` the gradient bucket's all-reduce alone (N > 1), so an N-GPU run reads as compute + collective;
  `cpu_baseline` the CPU oracle = op-for-op restatement of the reference's PyTorch path, timed on this box's host
                 cores on a bounded sample (median of 10 sampling calls / 5 minibatch updates).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md chip table
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # MI355X_MICROARCH.md chip table (dense)

# Workloads = BASELINE.json configs.  hopper: reference cfg/gym/finetune/hopper-v2/ft_ppo_diffusion_mlp.yaml; halfcheetah:
# cfg/gym/finetune/halfcheetah-v2/ft_ppo_diffusion_mlp.yaml:16-22,88-102; can: the networks of
# cfg/robomimic/finetune/can/ft_ppo_diffusion_mlp.yaml:18-24,92-105 at BASELINE configs[2]'s K=100 DDPM / Ta=8 / 256 envs.
WORKLOADS = {
    "hopper": dict(obs=11, act=3, ta=4, K=20, kft=10, act_steps=4, fn="ReLU", envs=512, n_steps=500, batch=50000,
                   clip_base=0.01, label="hopper-medium-v2 ft_ppo_diffusion_mlp K=20 Kft=10 Ta=4 (BASELINE configs[1])"),
    "can": dict(obs=23, act=7, ta=8, K=100, kft=10, act_steps=8, fn="Mish", envs=256, n_steps=300, batch=7500,
                clip_base=0.001, label="robomimic can state-obs diffusion_mlp K=100 DDPM Kft=10 Ta=8 (BASELINE configs[2])"),
    "halfcheetah": dict(obs=17, act=6, ta=4, K=20, kft=10, act_steps=4, fn="ReLU", envs=512, n_steps=500, batch=50000,
                        clip_base=0.01, label="halfcheetah-medium-v2 ft_ppo_diffusion_mlp K=20 Kft=10 Ta=4, one GPU's "
                                              "share (512 of 4096 envs) of BASELINE configs[3]"),
}
TIME_DIM, ACTOR_H, CRITIC_H = 16, 512, 256
# the headline workload's shape constants (tests and tools import them)
OBS_DIM, ACT_DIM, TA, K, KFT, ACT_STEPS = (WORKLOADS["hopper"][k] for k in ("obs", "act", "ta", "K", "kft", "act_steps"))


def net_flops(wl):
    """SURVEY.md 8(d)'s count for a workload's networks: one forward of the actor (time MLP + Linear(in, H) + one residual
    block of two H x H layers + Linear(H, Ta.Da)) and of the critic, 2 FLOP per multiply-add.  hopper: 1.103 / 0.268 MFLOP."""
    af = wl["ta"] * wl["act"]
    actor = 2 * ((af + TIME_DIM + wl["obs"]) * ACTOR_H + 2 * ACTOR_H * ACTOR_H + ACTOR_H * af) + 2 * (2 * TIME_DIM * 2 * TIME_DIM)
    critic = 2 * (wl["obs"] * CRITIC_H + 2 * CRITIC_H * CRITIC_H + CRITIC_H)
    return actor, critic


def flop_per_sample(wl):
    """One PPO sample updated: forward + backward (2x) of actor_ft and critic -- necessary evaluations only."""
    a, c = net_flops(wl)
    return 3.0 * (a + c)


def flop_per_chunk(wl):
    """One action chunk sampled: K necessary network evaluations."""
    return float(wl["K"] * net_flops(wl)[0])


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prec", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--n-envs", type=int, default=512)
    ap.add_argument("--n-steps", type=int, default=500, help="rollout length behind the update buffer")
    ap.add_argument("--batch", type=int, default=50000)
    ap.add_argument("--passes", type=int, default=10,
                    help="timed passes of the update leg, each its own barrier-bracketed region of exactly --steps steps; the "
                         "line reports the median pass (min / max beside it)")
    ap.add_argument("--spin-up", type=int, default=300,
                    help="untimed calls in front of each leg's first timed region, beyond --warmup, so that the GPU's clocks have ramped")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the secondary fp32 (the reference's own precision) pass")
    ap.add_argument("--no-pixel", action="store_true", help="skip the secondary pixel-observation pass (BASELINE configs[4] shapes)")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary BASELINE configs[2] / configs[3] legs")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-gpu rehearses the data-parallel path with several ranks on ONE GPU")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--dp-mode", default="split", choices=["split", "single"],
                    help="N > 1: gradient bucket all-reduced in two slices, the critic's overlapped with the actor's backward "
                         "(default), or in one piece behind the update")
    ap.add_argument("--probe", type=int, default=2,
                    help="kernel timed live for `roofline.dominant_kernel`: 2 gemm_tn (weight grads), 3 fused fwd, "
                         "4 fused bwd, 5 sampler")
    ap.add_argument("--graph", action="store_true",
                    help="replay the update step as captured hipGraphs (dppo_amd.util.graphed) instead of issuing its "
                         "launches one by one; pays at small minibatches, not at this workload")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE",
                    help="dppo_tune_set knob (include/dppo_hip.h), e.g. 2=0: critic half on the main stream (serial kernels)")
    ap.add_argument("--master-port", type=int, default=0, help="launcher mode: rendezvous port (default: a free one)")
    # test hooks of the launcher path (tests/test_bench_launcher.py): ranks rendezvous over gloo on the CPU, report and exit
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launch-check-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """Launcher mode (plain `python bench.py --gpus N`, N > 1, no WORLD_SIZE): start N fresh rank processes through
    torch.distributed.run BEFORE this process imports torch or makes any GPU call, forward rank 0's JSON line, return the child's
    exit code.  (An `os.exec*` would do too here -- nothing has touched the GPU -- but a child keeps the one-line contract
    enforceable: whatever the ranks print besides the result line goes to stderr.)"""
    assert "torch" not in sys.modules, "the launcher must not import torch"
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["DPPO_BENCH_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench launcher] torch imported: {'torch' in sys.modules}; starting {args.gpus} ranks: {' '.join(cmd)}",
          file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    n_lines = 0
    for line in child.stdout:
        is_result = False
        if line.lstrip().startswith("{"):
            try:
                is_result = isinstance(json.loads(line), dict)
            except ValueError:
                pass
        if is_result and n_lines == 0:
            sys.stdout.write(line)
            sys.stdout.flush()
            n_lines += 1
        else:
            sys.stderr.write(line)
    rc = child.wait()
    if rc == 0 and n_lines != 1:
        print(f"[bench launcher] ranks exited 0 but printed {n_lines} result lines", file=sys.stderr)
        rc = 1
    return rc


def launch_check(args):
    """What the ranks do under --launch-check: rendezvous over gloo (CPU only), gather (RANK, WORLD_SIZE), rank 0 prints one
    JSON line.  Exercises the launcher path where no GPU exists."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    t = torch.zeros(world, dtype=torch.int64)
    t[rank] = rank + 1
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "ranks": (t - 1).tolist(), "gpus_arg": args.gpus,
                          "env_world_size": int(os.environ["WORLD_SIZE"]), "cuda_initialized": torch.cuda.is_initialized()}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if rank == args.launch_check_fail_rank:
        sys.exit(3)


def build_model(device, prec, wl=None):
    wl = wl or WORKLOADS["hopper"]
    import torch
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    from dppo_amd.model.diffusion.mlp_diffusion import DiffusionMLP
    torch.manual_seed(42)
    actor = DiffusionMLP(action_dim=wl["act"], horizon_steps=wl["ta"], cond_dim=wl["obs"], time_dim=TIME_DIM,
                         mlp_dims=[ACTOR_H] * 3, activation_type=wl["fn"], residual_style=True, precision=prec)
    critic = CriticObs(cond_dim=wl["obs"], mlp_dims=[CRITIC_H] * 3, activation_type="Mish", residual_style=True,
                       precision=prec)
    model = PPODiffusion(actor=actor, critic=critic, ft_denoising_steps=wl["kft"], horizon_steps=wl["ta"], obs_dim=wl["obs"],
                         action_dim=wl["act"], denoising_steps=wl["K"], device=device, gamma_denoising=0.99,
                         clip_ploss_coef=0.01, clip_ploss_coef_base=wl["clip_base"], clip_ploss_coef_rate=3, randn_clip_value=3,
                         min_sampling_denoising_std=0.1, min_logprob_denoising_std=0.1)
    return model


def make_rollout(model, n_envs, n_steps, device, gen, wl=None):
    wl = wl or WORKLOADS["hopper"]
    """Device-resident rollout buffer produced by the sampler itself (R = n_steps * n_envs rows)."""
    import torch
    from dppo_amd.util.rollout import gae_device
    OBS_DIM, KFT, TA, ACT_DIM = wl["obs"], wl["kft"], wl["ta"], wl["act"]
    AF = TA * ACT_DIM
    R = n_steps * n_envs
    obs = torch.empty(R, OBS_DIM, device=device)
    chains = torch.empty(R, KFT + 1, AF, device=device)
    for s in range(n_steps):
        o = torch.rand(n_envs, 1, OBS_DIM, device=device, generator=gen) * 2 - 1
        smp = model(cond={"state": o}, deterministic=False, return_chain=True)
        obs[s * n_envs:(s + 1) * n_envs] = o.reshape(n_envs, -1)
        chains[s * n_envs:(s + 1) * n_envs] = smp.chains.reshape(n_envs, KFT + 1, AF)
    values = torch.empty(R, device=device)
    logp = torch.empty(R, KFT, AF, device=device)
    split = 20 * n_envs  # logprob_batch_size must be a multiple of n_envs (reference train_ppo_agent.py:22-25)
    for lo in range(0, R, split):
        hi = min(R, lo + split)
        st = {"state": obs[lo:hi].reshape(hi - lo, 1, OBS_DIM)}
        values[lo:hi] = model.critic(st).reshape(-1)
        logp[lo:hi] = model.get_logprobs(st, chains[lo:hi].reshape(hi - lo, KFT + 1, TA, ACT_DIM)).reshape(
            hi - lo, KFT, AF)
    reward = torch.randn(n_steps, n_envs, device=device, generator=gen, dtype=torch.float64)
    term = (torch.rand(n_steps, n_envs, device=device, generator=gen) < 0.002).float()
    last_v = model.critic({"state": torch.rand(n_envs, 1, OBS_DIM, device=device, generator=gen) * 2 - 1}).reshape(-1)
    _, _, adv, ret = gae_device(reward, values.reshape(n_steps, n_envs), term, last_v, 0.99, 0.95, 1.0)
    return obs, chains, ret.reshape(-1).contiguous(), values, adv.reshape(-1).contiguous(), logp


def usable_cores():
    """CPU threads this process may really use: cgroup quota if set, else the affinity mask (os.cpu_count() reports
    the whole host, which oversubscribes a container that owns a slice of it)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 64)


def cpu_baseline(n_envs, batch, budget_s=40.0):
    """The CPU oracle (a validated op-for-op restatement of the reference's PyTorch path, incl. the discarded base-net
    pass and the Python-list discount) on this box's host cores.  BASELINE.md section 3's protocol: warm-up, then the median of
    10 sampling calls and of 5 minibatch updates (fewer updates only if `budget_s` runs out; the result says how many)."""
    import numpy as np
    import torch
    from oracle import dppo_oracle as O
    wl = WORKLOADS["hopper"]
    OBS_DIM, KFT, TA, ACT_DIM, K = wl["obs"], wl["kft"], wl["ta"], wl["act"], wl["K"]
    cores = usable_cores()
    torch.set_num_threads(cores)
    a, c = O.named_specs("hopper")
    cfg = O.DiffusionCfg(denoising_steps=K, ft_denoising_steps=KFT, horizon_steps=TA, action_dim=ACT_DIM,
                         randn_clip_value=3, gamma_denoising=0.99, clip_ploss_coef=0.01, clip_ploss_coef_base=0.01)
    base, ft, cr = O.init_params(a, 42), O.init_params(a, 43), O.init_params(c, 44)
    rs = np.random.RandomState(42)
    state = torch.from_numpy(rs.uniform(-1, 1, size=(n_envs, 1, OBS_DIM)).astype(np.float32))
    noise = torch.from_numpy(rs.randn(K + 1, n_envs, TA, ACT_DIM).astype(np.float32))
    for _ in range(2):
        O.sample_chain(cfg, a, base, ft, state, noise)  # warm-up
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        _, chains = O.sample_chain(cfg, a, base, ft, state, noise)
        ts.append(time.perf_counter() - t0)
    t_sample = float(np.median(ts))
    # one PPO minibatch: loss forward + backward + 2 x AdamW (torch.optim, like the reference agent)
    for p in list(ft.values()) + list(cr.values()):
        p.requires_grad_(True)
    opt_a = torch.optim.AdamW(list(ft.values()), lr=1e-4, weight_decay=0)
    opt_c = torch.optim.AdamW(list(cr.values()), lr=1e-3, weight_decay=0)
    kinds = torch.from_numpy(rs.randint(0, KFT, size=(batch,)).astype(np.int64))
    rows = torch.from_numpy(rs.randint(0, n_envs, size=(batch,)).astype(np.int64))
    obs_b, prev, nxt = state[rows], chains[rows, kinds], chains[rows, kinds + 1]
    with torch.no_grad():
        oldlp = O.chain_logprob(cfg, a, base, ft, state, chains).reshape(n_envs, KFT, TA, ACT_DIM)[rows, kinds]
    ret = torch.from_numpy(rs.normal(size=batch).astype(np.float32))
    adv = torch.from_numpy(rs.normal(size=batch).astype(np.float32))
    oldv = torch.zeros(batch)
    tu, n_done, t_begin = [], -1, time.perf_counter()  # the first update is the warm-up (autograd graph, allocator), not counted
    while n_done < 5 and (n_done < 1 or time.perf_counter() - t_begin < budget_s):
        t0 = time.perf_counter()
        res = O.ppo_loss(cfg, a, c, base, ft, cr, obs_b, prev, nxt, kinds, ret, oldv, adv, oldlp)
        opt_a.zero_grad()
        opt_c.zero_grad()
        (res[0] + 0.5 * res[2]).backward()
        opt_a.step()
        opt_c.step()
        if n_done >= 0:
            tu.append(time.perf_counter() - t0)
        n_done += 1
    t_update = float(np.median(tu))
    return {"value": batch / t_update, "unit": "PPO-update samples/s", "cores": cores, "kind": "port",
            "env_steps_per_sec": n_envs * wl["act_steps"] / t_sample,
            "sample": f"median of 10 sampling calls (B={n_envs}, K={K}) and of {n_done} minibatch updates "
                      f"(N={batch}: loss fwd + bwd + 2x AdamW) after warm-up, torch {torch.__version__} CPU, {cores} threads"}


def pixel_leg(n_envs=256, batch=500, reps=10):
    """Secondary fields: the pixel path at BASELINE configs[4]'s shapes (robomimic square image cfg: one 96x96 camera, ViT +
    SpatialEmb, VisionUnet1D denoiser and, beside it, the VisionDiffusionMLP of the same cfg family; DDIM 100 -> 5 steps all
    fine-tuned, Ta 4, Da 7, ViTCritic), one GPU's share: rollout step for n_envs observations (both encoders are NOT needed
    there: one ViT pass + the 5-step sampler) and one update minibatch of the cfg's batch_size (two encoders forward with a
    tape, fused loss forward / backward, two encoders backward).  Synthetic images, random-init weights."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("vision_bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools",
                                                                                 "vision_bench.py"))
    vb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vb)
    dev, out = "cuda:0", {"what": pixel_leg.__doc__.split("  Synthetic")[0].replace("\n    ", " "), "n_envs": n_envs,
                          "minibatch": batch, "dtype": "bf16", "data": "synthetic"}
    for kind in ("unet", "mlp"):
        m = vb.build(kind, "bf16", dev)
        Kft, AF = 5, 28
        cond_e = {"rgb": torch.randint(0, 256, (n_envs, 1, 3, 96, 96), device=dev, dtype=torch.uint8),
                  "state": torch.rand(n_envs, 1, 9, device=dev) * 2 - 1}
        cond_n = {"rgb": torch.randint(0, 256, (batch, 1, 3, 96, 96), device=dev, dtype=torch.uint8),
                  "state": torch.rand(batch, 1, 9, device=dev) * 2 - 1}
        chains = m(cond=cond_n).chains.reshape(batch, Kft + 1, AF)
        kinds = torch.randint(0, Kft, (batch,), device=dev)
        rows = torch.arange(batch, device=dev)
        pairs = torch.stack([chains[rows, kinds], chains[rows, kinds + 1]], 1).contiguous()
        lp = m.get_logprobs(cond_n, chains.reshape(batch, Kft + 1, 4, 7)).reshape(batch, Kft, AF)[rows, kinds].contiguous()
        ret, val, adv = (torch.randn(batch, device=dev) for _ in range(3))
        ms_upd = vb.timeit(lambda: m._run_ppo_vision(cond_n, pairs, ret, val, adv, lp, kinds, batch, 4, None), n=reps)
        ms_smp = vb.timeit(lambda: m(cond=cond_e), n=reps)
        out[kind + "_img"] = {"update_ms_per_minibatch": ms_upd, "update_samples_per_sec": batch / ms_upd * 1e3,
                              "rollout_ms_per_step": ms_smp, "env_steps_per_sec": n_envs * 4 / ms_smp * 1e3}
        if hasattr(vb, "flops"):
            out[kind + "_img"].update(vb.flops(kind, batch, ms_upd, n_envs, ms_smp, MFMA_PEAK_TFLOPS["bf16"]))
        del m
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:  # plain invocation for N > 1 GPUs: become the launcher (no torch, no GPU call)
        sys.exit(launch_ranks(args, argv))
    if args.launch_check:
        launch_check(args)
        return
    import ctypes as C
    import statistics

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:  # the process group is what it is: report it, do not die on it
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: running (and reporting n_gpus =) {world} ranks",
              file=sys.stderr, flush=True)
    if args.share_gpu:
        local = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    from dppo_amd import hip
    from dppo_amd.parallel import DataParallel
    from dppo_amd.util.graphed import GraphedUpdate
    from dppo_amd.util.optim import FlatAdamW, step_and_repack
    lib = hip.load()
    for kv in args.tune:
        k, v = kv.split("=")
        hip.check(lib.dppo_tune_set(int(k), int(v)), "dppo_tune_set")
    n_total = args.steps + args.warmup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def run_path(prec, nranks, probe_id, wl, n_envs, n_steps, batch, passes):
        """Both legs of a step at one operand precision: `steps` sampling calls and `steps` update steps, each timed
        between barriers (+ the serial roofline pass of the update when `probe_id` is set).  nranks = 1 runs the path
        rank-locally (no collective): the secondary passes."""
        KFT, K = wl["kft"], wl["K"]
        model = build_model(str(device), prec, wl)  # same seed on every rank => identical initial weights
        gen = torch.Generator(device=device).manual_seed(42 + rank)  # env shards differ per rank
        torch.manual_seed(42 + rank)
        dp = DataParallel(model, nranks, split=(args.dp_mode == "split"))
        ro = make_rollout(model, n_envs, n_steps, device, gen, wl)
        adv_k = ro[4]
        R = n_envs * n_steps
        opt_a = FlatAdamW(model.actor_ft.flat_params(), lr=1e-4, weight_decay=0.0)
        opt_c = FlatAdamW(model.critic.flat_params(), lr=1e-3, weight_decay=0.0)
        perm = torch.randperm(R * KFT, device=device, generator=gen)
        n_mb = (R * KFT) // batch
        minibatches = [perm[(i % n_mb) * batch:(i % n_mb + 1) * batch].contiguous() for i in range(n_total)]
        moments = dp.minibatch_moments(adv_k, minibatches, KFT)  # ONE small collective for all steps (None if 1 rank)
        obs_batches = [torch.rand(n_envs, 1, wl["obs"], device=device, generator=gen) * 2 - 1 for _ in range(4)]
        graphed = GraphedUpdate(model, opt_a, opt_c, dp, ro, batch, wl["act_steps"], n_time=K) if args.graph else None
        eager = [False]  # the roofline pass issues the launches one by one (per-launch HIP events cannot be captured)

        def update_step(i):
            if graphed is not None and not eager[0]:  # same work as below, launched as hipGraph replays
                graphed.step(minibatches[i], None if moments is None else moments[i])
                return
            model.ppo_update(*ro, minibatches[i], reward_horizon=wl["act_steps"],
                             global_moments=None if moments is None else moments[i], critic_hook=dp.critic_hook)
            # RCCL all-reduce of the bucket [critic grads | actor grads | stats] in two slices: the critic's was queued from
            # inside the call above, on the library's critic stream (it overlaps the actor's backward); the rest here.  No-op
            # with one rank
            dp.allreduce_grads()
            # 2 x AdamW, then re-pack so the next sampling / update call sees the new weights (part of the step's cost)
            step_and_repack(model, opt_a, opt_c, n_time=K)

        def sample_step(i):
            return model(cond={"state": obs_batches[i % 4]}, deterministic=False, return_chain=True)

        def allreduce_only(i):  # both slices back to back on the caller's stream, nothing to hide behind
            if dp.critic_hook is not None:
                dp.critic_hook(0)
            dp.allreduce_grads()

        def timed(fn, probe=False, reps=1, passes=1):
            """`passes` timed regions, each EXACTLY `steps` calls (x `reps` repeats of them inside the region) between
            barrier + synchronize brackets, MAX over ranks; returns the list of per-pass times of ONE run of the `steps` calls."""
            for i in range(args.warmup):
                fn(i)
            # Clock spin-up (untimed, same count on every rank): an idle MI355X sits at its lowest sclk level (531 MHz by rocm-smi)
            # and takes tens of milliseconds of load to ramp.  As the first GPU process on a fresh box the 20 timed update steps
            # (8 ms of work) ran 6x slower than in the next process (2.4 vs 0.40 ms per step, twice in two tries); `--spin-up`
            # more untimed calls of the same function in front of the timed region take the ramp out of the measurement.
            for i in range(args.spin_up):
                fn(i % n_total)
                if i % 32 == 31:
                    torch.cuda.synchronize()
            out = []
            for _ in range(passes):
                barrier()
                if probe:
                    hip.check(lib.dppo_probe_arm(probe_id, 16 * args.steps), "dppo_probe_arm")
                t0 = time.perf_counter()
                for _ in range(reps):
                    for i in range(args.warmup, n_total):
                        fn(i)
                barrier()
                out.append(max_over_ranks(time.perf_counter() - t0) / reps)
            return out

        r = {"model": model, "prec": prec, "dp_mode": None}
        if nranks > 1:
            # One untimed step of the data-parallel path before anything is timed.  The two-slice form (critic slice from a
            # callback inside the library call, on the library's critic stream) has run over gloo only -- RCCL refuses two ranks
            # on one device, and a builder's box has one GPU -- so if it raises under nccl EVERY rank falls back to the plain
            # one-bucket all-reduce behind the update (same bits: a SUM is elementwise) and the line says so.
            ok = torch.ones(1, device=device)
            try:
                update_step(0)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                ok.zero_()
                print(f"[bench rank {rank}] two-slice all-reduce failed ({type(e).__name__}: {e}); falling back to one bucket",
                      file=sys.stderr, flush=True)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() == 0 and dp.split:
                dp.split = False
                dp._critic_done = False
            r["dp_mode"] = "two slices (critic slice overlapped from inside the update call)" if dp.split else "one bucket"
        # the sampler leg (secondary): a call is ~0.07 ms, so `steps` calls are timed ten times over in one region (1.4 ms of
        # work gave 26-30 M env-steps/s from run to run); with the per-launch probe on it stays at one pass
        probe_sampler = rank == 0 and probe_id == 5
        r["dt_sample"] = timed(sample_step, probe=probe_sampler, reps=1 if probe_id == 5 else 10)[0]
        r["dt_update_passes"] = timed(update_step, passes=passes)
        r["dt_update"] = statistics.median(r["dt_update_passes"])
        # the collective alone, same bucket, same `steps`: lets an N-GPU run be read as compute + all-reduce
        r["dt_allreduce"] = timed(allreduce_only)[0] if nranks > 1 else None
        # Roofline pass: the same update steps once more with the library's side streams off (knob 2 = 0).  In the timed
        # region above the critic half and the gradient tails run beside the probed kernel, so an event-bracketed launch
        # duration there includes its co-runners' share of the chip; serial, it is the kernel's own (rocprofv3 on
        # `bench.py --tune 2=0` reports the same average).  `value` always comes from the default (overlapped) pass.
        r["dt_serial"] = None
        if probe_id not in (None, 5):
            overlap = next((int(kv.split("=")[1]) for kv in args.tune if kv.split("=")[0] == "2"), 1)
            hip.check(lib.dppo_tune_set(2, 0), "dppo_tune_set")
            eager[0] = True
            r["dt_serial"] = timed(update_step, probe=(rank == 0))[0]
            eager[0] = False
            hip.check(lib.dppo_tune_set(2, overlap), "dppo_tune_set")
        r["stats"] = model._stats.tolist()
        r["graphed"] = graphed is not None
        # ... and what of it the step actually waits for: the same update steps with the collectives skipped (gradients stay
        # rank-local: timing only, run last on a throw-away model state) -- exposed = step - step_without
        r["dt_update_nocomm"] = None
        if nranks > 1:
            dp.skip_collectives = True
            r["dt_update_nocomm"] = statistics.median(timed(update_step, passes=max(3, passes // 2)))
            dp.skip_collectives = False
        return r

    def passes_ms(r):
        ps = sorted(x / args.steps * 1e3 for x in r["dt_update_passes"])
        return {"n": len(ps), "steps_each": args.steps, "min_ms_per_step": ps[0], "median_ms_per_step": statistics.median(ps),
                "max_ms_per_step": ps[-1]}

    WL = WORKLOADS["hopper"]
    ACT_STEPS = WL["act_steps"]
    FLOP_PER_SAMPLE, FLOP_PER_CHUNK = flop_per_sample(WL), flop_per_chunk(WL)  # 4.11 MFLOP, 22.06 MFLOP
    main_run = run_path(args.prec, world, args.probe, WL, args.n_envs, args.n_steps, args.batch, args.passes)
    model = main_run["model"]
    dt_sample, dt_update, dt_serial = main_run["dt_sample"], main_run["dt_update"], main_run["dt_serial"]

    kernel_probe = None  # the dominant kernel of the update, timed live with HIP events on its launch stream
    if rank == 0:
        ms, cnt, fl, nb_lib = C.c_double(), C.c_int(), C.c_double(), C.c_double()
        hip.check(lib.dppo_probe_collect_bytes(C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(nb_lib)),
                  "dppo_probe_collect_bytes")
        if cnt.value > 0:
            avg_ms = ms.value / cnt.value
            tf = fl.value / cnt.value / (avg_ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.prec]
            grouped = nb_lib.value > 0  # knob 12: every weight gradient of a backward pass in one launch
            names = {1: "gemm_nt_kernel (H x H layers, layered path)",
                     2: "gemm_tn_group_kernel (all weight gradients of one network's backward pass, one launch)"
                     if grouped else "gemm_tn_kernel (H x H weight gradients)",
                     3: "fused_forward_kernel (actor_ft + critic)", 4: "fused_backward_kernel (actor_ft + critic)",
                     5: "sample_chain_split_kernel (a 16-row tile over eight workgroups; sample_chain_kernel with --tune 27=0)"}
            traffic = None  # HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py)
            tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_probe{args.probe}{'g' if grouped else ''}_{args.prec}.json")
            one_wg_sampler = args.probe == 5 and any(t.replace(" ", "") == "27=0" for t in args.tune)
            if os.path.exists(tpath) and not one_wg_sampler:  # (the probe-5 file was measured on the split sampler)
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            kernel_probe = {"kernel": f"{names[args.probe]}, {args.prec}", "avg_launch_ms": avg_ms, "launches": cnt.value,
                            "algorithmic_gflop_per_launch": fl.value / cnt.value / 1e9, "mfma_tflops": tf,
                            "mfma_frac": tf / peak, "traffic": traffic,
                            "measured": "HIP events around every launch of the kernel" + (
                                "" if dt_serial is None else
                                f", in a second pass of the same {args.steps} steps with side streams off "
                                f"({dt_serial / args.steps * 1e3:.3f} ms per step)")}
            if args.probe == 2:
                # dW[N1 x N2] = A[M x N1]^T . B[M x N2]: bytes = both operands once + the fp32 result (per GEMM of the
                # group, counted by the library at launch).  Intermediate ACTIVATION bytes, not SURVEY 8(d)'s algorithmic
                # bytes of the update (13 MB): this sub-entry says how well the launch streams what it has to read, the
                # top-level `frac` says what the path achieves.
                es = 2 if args.prec == "bf16" else 4
                shapes = [(model.actor_ft.mlp_mean.hidden, model.actor_ft.mlp_mean.n_blocks),
                          (model.critic.Q1.hidden, model.critic.Q1.n_blocks)]
                nbytes = sum(2 * nb * (args.batch * 2 * h * es + h * h * 4) for h, nb in shapes)
                b_per = nb_lib.value / cnt.value if grouped else nbytes / sum(2 * nb for _, nb in shapes)
                gbs = b_per / (avg_ms * 1e-3) / 1e9
                kernel_probe.update({"operand_mb_per_launch": b_per / 1e6, "operand_gb_per_s": gbs,
                                     "hbm_frac": gbs / HBM_PEAK_GBS, "flop_per_operand_byte": (fl.value / cnt.value) / b_per})
    del model
    main_run.pop("model")
    torch.cuda.empty_cache()

    fp32_run = None
    if world == 1 and args.prec != "fp32" and not args.no_fp32:  # the reference's own arithmetic, same steps, same shapes
        fp32_run = run_path("fp32", 1, None, WL, args.n_envs, args.n_steps, args.batch, 3)
        del fp32_run["model"]
        torch.cuda.empty_cache()

    other = None  # BASELINE configs[2] and one GPU's share of configs[3]: same legs, same clocks, secondary
    if world == 1 and not args.no_configs:
        other = {}
        for key, name in (("C3_can", "can"), ("C4_halfcheetah_per_gpu", "halfcheetah")):
            wl = WORKLOADS[name]
            r = run_path(args.prec, 1, None, wl, wl["envs"], wl["n_steps"], wl["batch"], max(3, args.passes // 2))
            del r["model"]
            torch.cuda.empty_cache()
            sps = wl["batch"] / (r["dt_update"] / args.steps)
            cps = wl["envs"] / (r["dt_sample"] / args.steps)
            peak = MFMA_PEAK_TFLOPS[args.prec]
            other[key] = {"workload": wl["label"], "n_envs": wl["envs"], "minibatch": wl["batch"], "dtype": args.prec,
                          "samples_per_sec": sps, "ms_per_step": r["dt_update"] / args.steps * 1e3, "passes": passes_ms(r),
                          "env_steps_per_sec": cps * wl["act_steps"], "sampler_ms_per_call": r["dt_sample"] / args.steps * 1e3,
                          "mflop_per_sample": flop_per_sample(wl) / 1e6, "mflop_per_chunk": flop_per_chunk(wl) / 1e6,
                          "update_frac": sps * flop_per_sample(wl) / 1e12 / peak,
                          "sampler_frac": cps * flop_per_chunk(wl) / 1e12 / peak}

    pixel = None
    if world == 1 and args.prec == "bf16" and not args.no_pixel:
        pixel = pixel_leg()
        torch.cuda.empty_cache()

    stats = main_run["stats"]
    if rank == 0:
        ms_update = dt_update / args.steps * 1e3
        ms_sample = dt_sample / args.steps * 1e3
        samples_per_s = args.batch * world / (dt_update / args.steps)
        chunks_per_s = args.n_envs * world / (dt_sample / args.steps)
        env_steps_per_s = chunks_per_s * ACT_STEPS
        peak_tf = MFMA_PEAK_TFLOPS[args.prec] * world
        upd_tf, smp_tf = samples_per_s * FLOP_PER_SAMPLE / 1e12, chunks_per_s * FLOP_PER_CHUNK / 1e12
        # the path is MFMA-bound (20 kFLOP per algorithmic byte); achieved = units/s x algorithmic FLOP per
        # unit (necessary network evaluations only), priced against the dense MFMA peak of the operand dtype.
        roofline = {"bound": "mfma", "achieved": upd_tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": upd_tf / peak_tf,
                    "traffic": None if kernel_probe is None else kernel_probe["traffic"],
                    "what": "PPO update path: samples/s x 4.11 MFLOP per sample (SURVEY 8d) / dense MFMA peak of the dtype",
                    "sampler": {"achieved": smp_tf, "frac": smp_tf / peak_tf,
                                "what": "chunks/s x 22.06 MFLOP per action chunk / the same peak"},
                    "dominant_kernel": kernel_probe,
                    "traffic_is": "HBM bytes per launch of dominant_kernel (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"}
        ar_ms = None if main_run["dt_allreduce"] is None else main_run["dt_allreduce"] / args.steps * 1e3
        out = {
            "metric": "PPO-update samples/sec (+ env-steps/sec of the K=20 sampler), hopper K=20 n_envs=512",
            "value": samples_per_s, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "spin_up": args.spin_up, "ms_per_step": ms_update, "passes": passes_ms(main_run),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.prec, "data": "synthetic",
            "launch": "hipGraph replay of the update step (captured once in warm-up)" if main_run["graphed"] else "eager",
            "config": {"workload": WL["label"],
                       "n_envs_per_gpu": args.n_envs, "minibatch_per_gpu": args.batch,
                       "rollout_rows_per_gpu": args.n_envs * args.n_steps,
                       "parallelism": f"dp{world} (env-sharded, RCCL grad all-reduce)"},
            "env_steps_per_sec": env_steps_per_s, "sampler_ms_per_call": ms_sample, "chunks_per_sec": chunks_per_s,
            "allreduce_ms": ar_ms, "dp_mode": main_run.get("dp_mode"),
            "allreduce_exposed_ms": None if main_run.get("dt_update_nocomm") is None else
            (dt_update - main_run["dt_update_nocomm"]) / args.steps * 1e3,
            "allreduce_is": "gradient bucket [critic | actor | stats] SUM-reduced in two slices; allreduce_ms = both alone, "
                            "allreduce_exposed_ms = ms_per_step minus the same step with the collectives skipped",
            "last_stats": {"pg_loss": stats[0], "v_loss": stats[1], "approx_kl": stats[2], "ratio": stats[4]},
            "roofline": roofline,
        }
        if fp32_run is not None:
            f_sps = args.batch / (fp32_run["dt_update"] / args.steps)
            f_cps = args.n_envs / (fp32_run["dt_sample"] / args.steps)
            out["fp32"] = {"what": "the same two legs with fp32 MFMA operands (the reference's own arithmetic); secondary",
                           "samples_per_sec": f_sps, "ms_per_step": fp32_run["dt_update"] / args.steps * 1e3,
                           "env_steps_per_sec": f_cps * ACT_STEPS, "sampler_ms_per_call": fp32_run["dt_sample"] / args.steps * 1e3,
                           "peak_tflops": MFMA_PEAK_TFLOPS["fp32"],
                           "update_frac": f_sps * FLOP_PER_SAMPLE / 1e12 / MFMA_PEAK_TFLOPS["fp32"],
                           "sampler_frac": f_cps * FLOP_PER_CHUNK / 1e12 / MFMA_PEAK_TFLOPS["fp32"]}
        if other is not None:
            out["configs"] = other
        if pixel is not None:
            out["pixel"] = pixel
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.n_envs, args.batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
