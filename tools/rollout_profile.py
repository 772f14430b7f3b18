"""Host / device time of one rollout step at 512 envs, split into its parts (obs H2D, sampler call, action D2H, wait, env)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from dppo_amd.env.synthetic import SyntheticVecEnv
from dppo_amd.util import rollout as R
dev = torch.device("cuda", 0)
m = bench.build_model(str(dev), "bf16")
E, S = 512, 200
AF = bench.TA * bench.ACT_DIM
obs_buf = torch.zeros(S * E, bench.OBS_DIM, device=dev); chains_buf = torch.zeros(S * E, bench.KFT + 1, AF, device=dev)
env = SyntheticVecEnv(E, bench.OBS_DIM, bench.ACT_DIM, 1, bench.ACT_STEPS, seed=1)
obs = env.reset_arg()
h = R.PinnedHandoff(E, obs["state"].shape[1:], (bench.TA, bench.ACT_DIM), dev)
traj = torch.empty(E, AF, device=dev)
T = dict(h2d=0.0, model=0.0, d2h=0.0, wait=0.0, env=0.0)
o = obs["state"]
for step in range(S):
    t0 = time.perf_counter()
    st = h.obs_to_device(o, out=obs_buf[step*E:(step+1)*E]).view(E, 1, -1)
    t1 = time.perf_counter()
    m(cond={"state": st}, deterministic=False, return_chain=True, out=(traj, chains_buf[step*E:(step+1)*E]))
    t2 = time.perf_counter()
    tk = h.action_to_host_async(traj)
    t3 = time.perf_counter()
    a = h.action_numpy(tk)[:, :bench.ACT_STEPS]
    t4 = time.perf_counter()
    ob, r, te, tr, _ = env.step(a)
    o = ob["state"]
    t5 = time.perf_counter()
    if step >= 20:
        T["h2d"] += t1-t0; T["model"] += t2-t1; T["d2h"] += t3-t2; T["wait"] += t4-t3; T["env"] += t5-t4
n = S - 20
print({k: round(v / n * 1e6, 1) for k, v in T.items()}, "us per step; total", round(sum(T.values()) / n * 1e6, 1))
