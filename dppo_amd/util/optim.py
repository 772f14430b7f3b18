"""Fused AdamW on a flat fp32 buffer (torch.optim.AdamW + clip_grad_norm_ semantics; reference
agent/finetune/train_ppo_agent.py:34-62, train_ppo_diffusion_agent.py:360-373)."""
from __future__ import annotations

from typing import Optional

import torch

from dppo_amd import hip


class FlatAdamW:
    """One kernel per step over the whole parameter image; optional global-norm clipping without a host sync."""

    def __init__(self, flat_params: torch.Tensor, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        hip.require_gpu(flat_params, "FlatAdamW")
        assert flat_params.dtype == torch.float32 and flat_params.is_contiguous()
        self.p = flat_params
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat_params)
        self.exp_avg_sq = torch.zeros_like(flat_params)
        self.step_count = 0
        # step count and learning rate also live on the device (dppo_adamw_step_dev): a captured graph of the update
        # replays without any argument changing from step to step
        self._step_dev = torch.zeros(2, dtype=torch.int32, device=flat_params.device)  # {steps taken, scratch}
        self._lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=flat_params.device)
        self._lr_host = float(lr)
        self._norm = torch.zeros(1, dtype=torch.float64, device=flat_params.device)
        self._scratch = torch.zeros(1024, dtype=torch.float64, device=flat_params.device)
        # torch-optimizer look-alike for LR schedulers
        self.param_groups = [{"lr": lr}]

    def sq_norm(self, grad: torch.Tensor) -> torch.Tensor:
        hip.check(hip.load().dppo_grad_sq_norm(grad.data_ptr(), grad.numel(), self._scratch.data_ptr(),
                                               self._norm.data_ptr(), hip.stream()), "dppo_grad_sq_norm")
        return self._norm

    def step(self, grad: torch.Tensor, max_norm: Optional[float] = None, sq_norm: Optional[torch.Tensor] = None):
        """grad: flat fp32 like the params.  max_norm: clip_grad_norm_ threshold (norm computed on device unless
        ``sq_norm`` -- e.g. an all-reduced value -- is given)."""
        assert grad.numel() == self.p.numel() and grad.is_contiguous()
        self.step_count += 1
        self.sync_lr()
        norm_ptr = None
        if max_norm is not None:
            norm_ptr = (sq_norm if sq_norm is not None else self.sq_norm(grad)).data_ptr()
        hip.check(hip.load().dppo_adamw_step_dev(
            self.p.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.p.numel(),
            self._step_dev.data_ptr(), self._lr_dev.data_ptr(), float(self.betas[0]), float(self.betas[1]), float(self.eps),
            float(self.weight_decay), norm_ptr, float(max_norm or 0.0), hip.stream()), "dppo_adamw_step_dev")

    def slot(self, grad: torch.Tensor, max_norm: Optional[float] = None,
             sq_norm: Optional[torch.Tensor] = None) -> "hip.AdamwSlot":
        """This optimiser's step as one slot of ``dppo_adamw_step_multi`` (see ``step_many``)."""
        assert grad.numel() == self.p.numel() and grad.is_contiguous()
        self.step_count += 1
        self.sync_lr()
        norm_ptr = None
        if max_norm is not None:
            norm_ptr = (sq_norm if sq_norm is not None else self.sq_norm(grad)).data_ptr()
        return hip.AdamwSlot(self.p.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                             self.p.numel(), self._step_dev.data_ptr(), self._lr_dev.data_ptr(), float(self.betas[0]),
                             float(self.betas[1]), float(self.eps), float(self.weight_decay), norm_ptr,
                             float(max_norm or 0.0))

    def sync_lr(self):
        """Push a changed ``param_groups[0]['lr']`` (LR schedulers) to the device copy.  Never inside a graph capture: a
        captured fill would pin the old value; callers that replay graphs call this before the replay."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host and not torch.cuda.is_current_stream_capturing():
            self._lr_dev.fill_(lr)
            self._lr_host = lr


def step_many(slots) -> None:
    """One launch for the AdamW steps of several optimisers (``FlatAdamW.slot`` each)."""
    import ctypes as C
    arr = (hip.AdamwSlot * len(slots))(*slots)
    hip.check(hip.load().dppo_adamw_step_multi(C.cast(arr, C.c_void_p), len(slots), hip.stream()),
              "dppo_adamw_step_multi")


def step_and_repack(model, actor_opt: "FlatAdamW", critic_opt: "FlatAdamW", update_actor: bool = True,
                    max_norm: Optional[float] = None, n_time: Optional[int] = None,
                    actor_sq_norm: Optional[torch.Tensor] = None):
    """Optimiser step + kernel-image repack of both networks (reference train_ppo_diffusion_agent.py:360-373).

    The tail of an update is a chain of launch-latency-bound kernels, so it is kept short: one AdamW launch for both
    networks, then both networks' weight images in two launches, all on the caller's stream (a side stream for the critic's chain costs
    a cross-stream hop of 10-20 us, more than the two launches it hides).
    """
    n_time = model.denoising_steps if n_time is None else n_time
    slots = [critic_opt.slot(model.critic.flat_grads())]
    if update_actor:
        # actor_sq_norm: a clip norm that spans more than the trunk (a Gaussian head's logvar: train_ppo_gaussian_agent.py)
        slots.append(actor_opt.slot(model.actor_ft.flat_grads(), max_norm=max_norm, sq_norm=actor_sq_norm))
    step_many(slots)
    model.critic.mark_updated()
    if update_actor:
        from dppo_amd.model.common.mlp import pack_pair
        model.actor_ft.mark_updated()
        pack_pair(model.critic, 0, model.actor_ft, n_time, model.prec)  # both composites, then both images: 2 launches
    else:
        model.critic.packed(model.prec, 0)
