"""``EvalImgGaussianAgent`` of the reference (``dppo/agent/eval/eval_gaussian_img_agent.py``): see eval_agent.py."""
from dppo_amd.agent.eval.eval_agent import EvalAgent


class EvalImgGaussianAgent(EvalAgent):
    obs_keys = ("rgb", "state")
