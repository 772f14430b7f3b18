"""Diffusion policy with a frozen base net and a fine-tuned copy.

Mirrors ``dppo/model/diffusion/diffusion_vpg.py:27-461`` (reference ``VPGDiffusion``): constructor surface,
``forward`` (K-step sampling with chain capture), ``get_logprobs``, ``get_logprobs_subsample``, ``step``.
The K-step loop, both networks and the posterior arithmetic run in ONE persistent HIP kernel
(``dppo_sample_chain``); the host only prepares the per-step coefficient table, in fp32 torch ops that
repeat the reference's own expressions so the coefficients are bit-identical.
"""
from __future__ import annotations

import copy
import ctypes as C
import logging

import numpy as np
import torch

from dppo_amd import hip
from dppo_amd.model.diffusion.diffusion import DiffusionModel, Sample

log = logging.getLogger(__name__)


class VPGDiffusion(DiffusionModel):
    supports_out = True  # forward(..., out=(trajectories, chains)) writes caller-owned buffers (util/rollout.py)

    def __init__(self, actor, critic, ft_denoising_steps, ft_denoising_steps_d=0, ft_denoising_steps_t=0,
                 network_path=None, min_sampling_denoising_std=0.1, min_logprob_denoising_std=0.1, eta=None,
                 learn_eta=False, **kwargs):
        super().__init__(network=actor, network_path=network_path, **kwargs)
        assert ft_denoising_steps <= self.denoising_steps
        assert ft_denoising_steps <= self.ddim_steps if self.use_ddim else True
        assert not (learn_eta and not self.use_ddim), "Cannot learn eta with DDPM."
        if learn_eta:
            raise NotImplementedError("dppo_amd: learn_eta=True is out of scope (all shipped cfgs keep eta fixed)")
        self.ft_denoising_steps = ft_denoising_steps
        self.ft_denoising_steps_d = ft_denoising_steps_d
        self.ft_denoising_steps_t = ft_denoising_steps_t
        self.ft_denoising_steps_cnt = 0
        self.min_sampling_denoising_std = min_sampling_denoising_std
        self.min_logprob_denoising_std = min_logprob_denoising_std
        self.learn_eta = learn_eta
        if eta is not None:
            self.eta = eta.to(self.device)
            for p in self.eta.parameters():
                p.requires_grad = False
        self.actor = self.network
        self.actor_ft = copy.deepcopy(self.actor)
        for p in self.actor.parameters():
            p.requires_grad = False
        self.critic = critic.to(self.device)
        if network_path is not None:
            checkpoint = torch.load(network_path, map_location=self.device, weights_only=True)
            if "ema" not in checkpoint:
                self.load_state_dict(checkpoint["model"], strict=False)
        object.__setattr__(self, "_sched_cache", {})
        object.__setattr__(self, "_ws_logprob", hip.Workspace())
        object.__setattr__(self, "_ws_sample", hip.Workspace())

    # ------------------------------------------------------------------ annealing (reference :102-136)
    def step(self):
        if type(self.min_sampling_denoising_std) is not float:
            self.min_sampling_denoising_std.step()
        self.ft_denoising_steps_cnt += 1
        if (self.ft_denoising_steps_d > 0 and self.ft_denoising_steps_t > 0
                and self.ft_denoising_steps_cnt % self.ft_denoising_steps_t == 0):
            self.ft_denoising_steps = max(0, self.ft_denoising_steps - self.ft_denoising_steps_d)
            self.actor = self.actor_ft
            self.actor_ft = copy.deepcopy(self.actor)
            for p in self.actor.parameters():
                p.requires_grad = False
            self._sched_cache.clear()

    def get_min_sampling_denoising_std(self):
        if type(self.min_sampling_denoising_std) is float:
            return self.min_sampling_denoising_std
        return self.min_sampling_denoising_std()

    # ------------------------------------------------------------------ per-step coefficient tables
    # (_eta_value, _ddim_coefs, _ddpm_coefs, _sampling_schedule and the sampler launch live in DiffusionModel)
    def _logprob_schedule(self, device):
        """dppo_step table of chain position k = 0..Kft-1 (reference :351-370, :388-389)."""
        key = ("logprob", self.ft_denoising_steps, float(self.min_logprob_denoising_std), str(device),
               self._eta_value(False))
        hit = self._sched_cache.get(key)
        if hit is not None:
            return hit
        Kft = self.ft_denoising_steps
        tab = np.zeros(Kft, dtype=hip.STEP_DTYPE)
        for k in range(Kft):
            if self.use_ddim:
                i = self.ddim_steps - Kft + k
                t = int(self.ddim_t[i])
                c0, c1, c2, c3, std = self._ddim_coefs(i, self._eta_value(False))
            else:
                t = Kft - 1 - k
                c0, c1, c2, c3, std = self._ddpm_coefs(t)
            std = torch.clip(std, min=self.min_logprob_denoising_std)
            tab[k] = (1, t, -1, 0, c0, c1, c2, c3, float(std), 0.0)
        out = torch.from_numpy(tab.view(np.uint8)).to(device)
        self._sched_cache[key] = out
        return out

    # ------------------------------------------------------------------ sampling (reference :227-315)
    @torch.no_grad()
    def forward(self, cond, deterministic=False, return_chain=True, use_base_policy=False, noise=None, out=None):
        """cond {"state": (B,To,Do)} -> Sample(trajectories (B,Ta,Da), chains (B,Kft+1,Ta,Da)): the K-step loop of the
        reference's ``VPGDiffusion.forward`` as ONE launch (``DiffusionModel._run_sampler``).

        ``out = (trajectories, chains)``: contiguous fp32 device tensors of B*Ta*Da and B*(Kft+1)*Ta*Da elements the
        kernel writes straight into (the rollout loop passes slices of its buffer: no allocation, no copy kernel).

        ``noise`` (n_steps+1,B,Ta,Da) replaces the internal draw (parity tests): noise[0] is the initial x, noise[i+1]
        the draw of step i.  Without it the kernel draws N(0,1) itself (Philox keyed by a 64-bit value taken from
        torch's CPU generator, so ``torch.manual_seed`` reproduces a run) -- the reference's ``torch.randn`` /
        ``randn_like`` (:271, :303) in distribution, without a separate noise launch and tensor.
        """
        return self._run_sampler(cond, deterministic, return_chain, use_base_policy, noise, out, "VPGDiffusion.forward")

    # ------------------------------------------------------------------ log-probs (reference :319-396)
    @torch.no_grad()
    def get_logprobs(self, cond, chains, get_ent: bool = False, use_base_policy: bool = False):
        """chains (B,Kft+1,Ta,Da) -> log N(x_{k+1}; mu(x_k), sigma_k) elementwise, (B*Kft,Ta,Da).

        Inference only (the rollout precompute); the differentiable evaluation happens fused inside
        ``PPODiffusion.loss``.
        """
        state = cond["state"]
        hip.require_gpu(state, "VPGDiffusion.get_logprobs")
        B, dev = chains.shape[0], chains.device
        Kft = self.ft_denoising_steps
        AF = self.horizon_steps * self.action_dim
        net = self.actor if use_base_policy else self.actor_ft
        lib, d = hip.load(), net.net_desc()
        ch = chains.reshape(B, Kft + 1, AF).contiguous().float()
        # pixel networks: ONE encoder pass per observation serves its Kft chain steps (the reference repeats the images Kft
        # times and encodes every copy, diffusion_vpg.py:340-345)
        obs = net.encode_obs(cond) if getattr(net, "is_vision", False) else state.reshape(B, -1).contiguous().float()
        out = torch.empty((B * Kft, AF), device=dev, dtype=torch.float32)
        ks = self._logprob_schedule(dev)
        cfg = self.diffusion_cfg()
        if getattr(net, "is_unet", False):  # conv denoiser
            ws = net.workspace(B * Kft, dev)
            hip.check(lib.dppo_unet_chain_logprob(
                C.byref(d), self.prec, net.flat_params().data_ptr(), net.packed(self.prec, self.denoising_steps).data_ptr(),
                C.byref(cfg), ks.data_ptr(), None, Kft, obs.data_ptr(), ch.data_ptr(), B, out.data_ptr(), ws.data_ptr(),
                ws.numel(), hip.stream()), "dppo_unet_chain_logprob")
        else:
            wsb = lib.dppo_chain_logprob_workspace_bytes(C.byref(d), self.prec, B, Kft)
            ws = self._ws_logprob.get(wsb, dev)
            hip.check(lib.dppo_chain_logprob(
                C.byref(d), self.prec, net.flat_params().data_ptr(), net.packed(self.prec, self.denoising_steps).data_ptr(),
                C.byref(cfg), ks.data_ptr(), Kft, obs.data_ptr(), ch.data_ptr(), B, out.data_ptr(), ws.data_ptr(),
                ws.numel(), hip.stream()), "dppo_chain_logprob")
        out = out.view(B * Kft, self.horizon_steps, self.action_dim)
        if get_ent:
            eta = torch.full_like(out, 1.0) if not self.use_ddim else torch.full(
                (B * Kft, 1, 1), self._eta_value(False), device=dev)
            return out, eta
        return out

    @torch.no_grad()
    def get_logprobs_subsample(self, cond, chains_prev, chains_next, denoising_inds, get_ent: bool = False,
                               use_base_policy: bool = False):
        """log N(chains_next; mu(chains_prev, t_k, s), sigma_k) for one random denoising step per sample, (B,Ta,Da)
        (reference diffusion_vpg.py:398-461; k = denoising_inds[b] indexes the chain position like there).

        Inference only: inside ``PPODiffusion.loss`` the same evaluation runs fused and differentiable.  Here the
        samples are grouped by k and each group goes through the chain kernel as a one-step chain.
        """
        if getattr(self.actor_ft, "is_unet", False):
            raise NotImplementedError("dppo_amd: get_logprobs_subsample with a conv denoiser is not built yet (use get_logprobs)")
        state = cond["state"]
        hip.require_gpu(state, "VPGDiffusion.get_logprobs_subsample")
        B, dev = chains_prev.shape[0], chains_prev.device
        AF = self.horizon_steps * self.action_dim
        net = self.actor if use_base_policy else self.actor_ft
        lib, d = hip.load(), net.net_desc()
        obs = net.encode_obs(cond) if getattr(net, "is_vision", False) else state.reshape(B, -1).contiguous().float()
        pairs = torch.stack([chains_prev.reshape(B, AF), chains_next.reshape(B, AF)], dim=1).contiguous().float()
        kinds = denoising_inds.reshape(B).to(torch.long)
        ks = self._logprob_schedule(dev)
        step_bytes = hip.STEP_DTYPE.itemsize
        cfg = self.diffusion_cfg()
        out = torch.empty((B, AF), device=dev, dtype=torch.float32)
        packed = net.packed(self.prec, self.denoising_steps)
        for k in torch.unique(kinds).tolist():
            rows = torch.nonzero(kinds == k).reshape(-1)
            n = rows.numel()
            ob, ch = obs[rows].contiguous(), pairs[rows].contiguous()
            lp = torch.empty((n, AF), device=dev, dtype=torch.float32)
            ks_k = ks[k * step_bytes:(k + 1) * step_bytes].contiguous()
            wsb = lib.dppo_chain_logprob_workspace_bytes(C.byref(d), self.prec, n, 1)
            ws = self._ws_logprob.get(wsb, dev)
            hip.check(lib.dppo_chain_logprob(
                C.byref(d), self.prec, net.flat_params().data_ptr(), packed.data_ptr(), C.byref(cfg), ks_k.data_ptr(), 1,
                ob.data_ptr(), ch.data_ptr(), n, lp.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream()),
                "dppo_chain_logprob")
            out[rows] = lp
        out = out.view(B, self.horizon_steps, self.action_dim)
        if get_ent:
            eta = torch.full_like(out, 1.0) if not self.use_ddim else torch.full(
                (B, 1, 1), self._eta_value(False), device=dev)
            return out, eta
        return out

