"""Evaluation of a pre-trained or RL-fine-tuned diffusion policy from a checkpoint.

Mirrors ``dppo/model/diffusion/diffusion_eval.py:19-150`` (reference ``DiffusionEval``): the frozen base policy on the
early denoising steps, the fine-tuned policy on the last ``ft_denoising_steps``; weights come from an RL checkpoint's
``actor.*`` / ``actor_ft.*`` keys, or -- with ``ft_denoising_steps = 0`` -- from a pre-training checkpoint's
``network.*`` keys.  Sampling is ``DiffusionModel.forward``: the same one-launch K-step kernel the fine-tuning
sampler uses, driven by the deterministic step table (reference ``diffusion.py:261-316`` through the overridden
``p_mean_var`` :70-150).
"""
from __future__ import annotations

import copy
import logging

import torch

from dppo_amd.model.diffusion.diffusion import DiffusionModel

log = logging.getLogger(__name__)


class DiffusionEval(DiffusionModel):
    def __init__(self, network_path, ft_denoising_steps, use_ddim=False, **kwargs):
        super().__init__(use_ddim=use_ddim, network_path=None, **kwargs)  # the base class must not load the checkpoint
        assert ft_denoising_steps <= (self.ddim_steps if use_ddim else self.denoising_steps)
        self.ft_denoising_steps = ft_denoising_steps
        checkpoint = torch.load(network_path, map_location=self.device, weights_only=True)  # safe loader only
        state = checkpoint["model"]
        self.actor = self.network

        def sub(prefix):
            return {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}

        base_weights = sub("actor.")
        if base_weights:
            self.actor.load_state_dict(base_weights, strict=True)
            use_ft = True
        else:  # a pre-training checkpoint: one network, every step runs it
            assert ft_denoising_steps == 0, "If no base policy weights are found, ft_denoising_steps must be 0"
            log.info("Actor weights not found. Using pre-trained weights!")
            self.actor.load_state_dict(sub("network."), strict=True)
            use_ft = False
        self.actor.mark_updated()
        log.info("Loaded base policy weights from %s", network_path)
        if use_ft:
            self.actor_ft = copy.deepcopy(self.network)
            self.actor_ft.load_state_dict(sub("actor_ft."), strict=True)
            self.actor_ft.mark_updated()
            log.info("Loaded fine-tuned policy weights from %s", network_path)
        for p in self.parameters():
            p.requires_grad = False
