"""``EvalGaussianAgent`` of the reference (``dppo/agent/eval/eval_gaussian_agent.py``): see eval_agent.py."""
from dppo_amd.agent.eval.eval_agent import EvalAgent


class EvalGaussianAgent(EvalAgent):
    obs_keys = ("state",)
