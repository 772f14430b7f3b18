"""Alias: 26 of the reference's eval cfgs name ``dppo.model.diffusion.diffusion_eval_ft.DiffusionEval`` (e.g.
``cfg/gym/eval/hopper-v2/eval_diffusion_mlp.yaml``), a module path the reference itself no longer ships; here it resolves."""
from dppo_amd.model.diffusion.diffusion_eval import DiffusionEval  # noqa: F401
