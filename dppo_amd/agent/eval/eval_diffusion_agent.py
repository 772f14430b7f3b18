"""``EvalDiffusionAgent`` of the reference (``dppo/agent/eval/eval_diffusion_agent.py``): see eval_agent.py."""
from dppo_amd.agent.eval.eval_agent import EvalAgent


class EvalDiffusionAgent(EvalAgent):
    obs_keys = ("state",)
