"""Pre-training of a diffusion policy (reference agent/pretrain/train_agent.py:58-168, train_diffusion_agent.py:18-88).

Same schedule as the reference -- AdamW, cosine LR with warm-up per epoch, EMA copy updated every ``update_ema_freq``
batches once ``epoch_start_ema`` is reached, checkpoints {"epoch", "model", "ema"} that ``DiffusionModel(network_path=)``
and the fine-tuning agents load -- on the HIP path: the supervised loss and all its gradients come from
``DiffusionModel.loss`` (dppo_denoise_mse_fwd_bwd), the optimiser is the flat fused AdamW, the EMA is one fused
multiply-add over the flat parameter image, minibatches are gathered on the device.
"""
from __future__ import annotations

import logging
import os
import random
import time

import numpy as np
import torch

from dppo_amd.cfg.loader import instantiate
from dppo_amd.util.optim import FlatAdamW, step_many
from dppo_amd.util.scheduler import CosineAnnealingWarmupRestarts

log = logging.getLogger(__name__)


class TrainDiffusionAgent:
    def __init__(self, cfg, dataset=None):
        self.cfg = cfg
        self.seed = cfg.get("seed", 42)
        random.seed(self.seed)
        np.random.seed(self.seed)
        torch.manual_seed(self.seed)
        self.model = instantiate(cfg.model)
        self.net = self.model.network
        for p in self.net.parameters():
            p.requires_grad_(True)
        self.ema_decay = cfg.ema.decay
        self.ema_flat = self.net.flat_params().clone()  # reset_parameters(): the EMA starts as a copy of the model
        self.n_epochs, self.batch_size = cfg.train.n_epochs, cfg.train.batch_size
        self.epoch_start_ema = cfg.train.get("epoch_start_ema", 20)
        self.update_ema_freq = cfg.train.get("update_ema_freq", 10)
        self.logdir = cfg.logdir
        self.checkpoint_dir = os.path.join(self.logdir, "checkpoint")
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        self.log_freq = cfg.train.get("log_freq", 1)
        self.save_model_freq = cfg.train.save_model_freq
        self.dataset_train = dataset if dataset is not None else instantiate(cfg.train_dataset)
        self.optimizer = FlatAdamW(self.net.flat_params(), lr=cfg.train.learning_rate,
                                   weight_decay=cfg.train.weight_decay)
        sch = cfg.train.lr_scheduler
        self.lr_scheduler = CosineAnnealingWarmupRestarts(
            self.optimizer, first_cycle_steps=sch.first_cycle_steps, cycle_mult=1.0, max_lr=cfg.train.learning_rate,
            min_lr=sch.min_lr, warmup_steps=sch.warmup_steps, gamma=1.0)
        self.epoch = 1

    # ---- EMA (train_agent.py:36-56, :137-144)
    def step_ema(self):
        p = self.net.flat_params()
        if self.epoch < self.epoch_start_ema:
            self.ema_flat.copy_(p)
        else:
            self.ema_flat.mul_(self.ema_decay).add_(p, alpha=1.0 - self.ema_decay)

    def _state_dict_of(self, flat):
        """state_dict of the whole DiffusionModel with the network's parameters taken from ``flat``."""
        sd = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        off = 0
        for name, p in self.net.named_parameters():
            sd["network." + name] = flat[off:off + p.numel()].view(p.shape).detach().clone()
            off += p.numel()
        return sd

    def save_model(self):
        path = os.path.join(self.checkpoint_dir, f"state_{self.epoch}.pt")
        torch.save({"epoch": self.epoch, "model": self._state_dict_of(self.net.flat_params()),
                    "ema": self._state_dict_of(self.ema_flat)}, path)
        log.info("Saved model to %s", path)
        return path

    def load(self, epoch):
        data = torch.load(os.path.join(self.checkpoint_dir, f"state_{epoch}.pt"), weights_only=True)
        self.epoch = data["epoch"]
        self.model.load_state_dict(data["model"])
        self.net.mark_updated()
        off, flat = 0, self.ema_flat
        for name, p in self.net.named_parameters():
            flat[off:off + p.numel()].copy_(data["ema"]["network." + name].reshape(-1))
            off += p.numel()

    def run(self):
        t0 = time.time()
        cnt_batch = 0
        gen = torch.Generator().manual_seed(self.seed)
        history = []
        for _ in range(self.n_epochs):
            losses = []
            for batch in self.dataset_train.epoch(self.batch_size, generator=gen):
                # value + every gradient in one library call; `loss.backward()` would hand each parameter its slice of the
                # same flat image (that is what a torch optimiser needs) -- the flat AdamW reads it directly
                loss = self.model.loss(batch.actions, batch.conditions)
                step_many([self.optimizer.slot(self.model.last_loss_grad)])
                self.net.mark_updated()
                losses.append(loss.detach())
                if cnt_batch % self.update_ema_freq == 0:
                    self.step_ema()
                cnt_batch += 1
            loss_train = float(torch.stack(losses).mean()) if losses else float("nan")
            self.lr_scheduler.step()
            if self.epoch % self.save_model_freq == 0 or self.epoch == self.n_epochs:
                self.save_model()
            if self.epoch % self.log_freq == 0:
                log.info("%d: train loss %8.4f | t:%8.4f", self.epoch, loss_train, time.time() - t0)
            history.append({"epoch": self.epoch, "loss": loss_train})
            self.epoch += 1
        self.epoch -= 1
        return history
