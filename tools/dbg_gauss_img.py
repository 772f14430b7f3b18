import os, sys, tempfile
sys.path.insert(0, ".")
import numpy as np, torch
os.environ["DPPO_LOG_DIR"] = tempfile.mkdtemp()
from tests.test_agent_gpu import GAUSS_IMG_YAML
from dppo_amd.cfg.loader import get_class, load_config
from dppo_amd import hip
p = os.path.join(os.environ["DPPO_LOG_DIR"], "c.yaml")
open(p, "w").write(GAUSS_IMG_YAML.replace("RGB_C", "3"))
cfg = load_config(p)
agent = get_class(cfg._target_)(cfg)
m = agent.model
obs = agent.reset_env_all()
cond = agent._cond(obs)
N = 4
a = m(cond=cond, deterministic=False)
lp, _, _ = m.get_logprobs(cond, a)
mean0, _ = m.actor_ft(cond)
obs0 = m.actor_ft.encode_obs(cond).clone()
st = m.ppo_update(cond, a.reshape(N, -1).contiguous(), torch.randn(N, device="cuda"), torch.zeros(N, device="cuda"),
                  torch.randn(N, device="cuda"), lp.contiguous())
print("grad norms: trunk", float(m.actor_ft.flat_grads().norm()), "enc", float(m.actor_ft.vis.flat_grads().norm()))
agent._accumulate(first=True)
p_tr, p_en = m.actor_ft.flat_params().clone(), m.actor_ft.vis.flat_params().clone()
agent._optimizer_step(True)
print("param delta: trunk max", float((m.actor_ft.flat_params() - p_tr).abs().max()), "enc max", float((m.actor_ft.vis.flat_params() - p_en).abs().max()))
mean1, _ = m.actor_ft(cond)
obs1 = m.actor_ft.encode_obs(cond)
print("obs' shift max", float((obs1 - obs0).abs().max()), "obs' scale", float(obs0.abs().max()))
print("mean shift max", float((mean1 - mean0).abs().max()))
lp1, _, _ = m.get_logprobs(cond, a)
print("logp before", lp.cpu().numpy(), "after", lp1.cpu().numpy())
