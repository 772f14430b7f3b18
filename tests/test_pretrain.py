"""Pre-training path: device-resident sequence dataset (CPU), and on the GPU a short supervised
run through dppo_denoise_mse_fwd_bwd whose checkpoint the fine-tuning model loads (reference agent/pretrain/*,
agent/dataset/sequence.py, model/diffusion/diffusion.py:77-86)."""
import os
import textwrap

import numpy as np
import pytest
import torch

from dppo_amd.agent.dataset.sequence import StitchedSequenceDataset, synthetic_dataset


def test_sequence_dataset_windows_and_history():
    lens = [5, 3, 9]
    T = sum(lens)
    states = np.arange(T, dtype=np.float32)[:, None] * np.ones((1, 2), dtype=np.float32)
    actions = 100 + np.arange(T, dtype=np.float32)[:, None]
    ds = StitchedSequenceDataset(horizon_steps=4, cond_steps=3, device="cpu", states=states, actions=actions,
                                 traj_lengths=lens)
    # the reference's rule (agent/dataset/sequence.py:174-187): every start whose window stays inside its trajectory
    expect, cur = [], 0
    for n in lens:
        expect += [(i, i - cur) for i in range(cur, cur + n - 4 + 1)]
        cur += n
    assert len(ds) == len(expect) == 2 + 0 + 6
    got = list(zip(ds.start.tolist(), ds.before.tolist()))
    assert got == expect
    for i, (start, before) in enumerate(expect):
        b = ds[i]
        np.testing.assert_array_equal(b.actions[:, 0].numpy(), 100 + np.arange(start, start + 4))
        hist = [start - min(t, before) for t in (2, 1, 0)]  # first observation repeated at a trajectory's start (:141-143)
        np.testing.assert_array_equal(b.conditions["state"][:, 0].numpy(), np.array(hist, dtype=np.float32))
    seen = torch.cat([b.actions[:, 0, 0] for b in ds.epoch(3, generator=torch.Generator().manual_seed(0))])
    assert sorted(seen.tolist()) == sorted(100.0 + s for s, _ in expect)  # one epoch = every window once


def test_npz_loader_refuses_pickles(tmp_path):
    p = tmp_path / "train.npz"
    np.savez(p, states=np.zeros((6, 2), np.float32), actions=np.zeros((6, 1), np.float32), traj_lengths=np.array([6]))
    assert len(StitchedSequenceDataset(str(p), horizon_steps=4, device="cpu")) == 3
    with pytest.raises(ValueError):
        StitchedSequenceDataset(str(tmp_path / "train.pkl"), horizon_steps=4, device="cpu")


YAML = textwrap.dedent("""
    _target_: dppo.agent.pretrain.train_diffusion_agent.TrainDiffusionAgent
    logdir: ${oc.env:DPPO_LOG_DIR}/pretrain
    seed: 42
    device: cuda:0
    obs_dim: 11
    action_dim: 3
    denoising_steps: 20
    horizon_steps: 4
    cond_steps: 1
    wandb: null
    train:
      n_epochs: 6
      batch_size: 128
      learning_rate: 1e-3
      weight_decay: 1e-6
      lr_scheduler: {first_cycle_steps: 200, warmup_steps: 1, min_lr: 1e-4}
      save_model_freq: 100
      epoch_start_ema: 2
      update_ema_freq: 2
    model:
      _target_: dppo.model.diffusion.diffusion.DiffusionModel
      predict_epsilon: True
      denoised_clip_value: 1.0
      network:
        _target_: dppo.model.diffusion.mlp_diffusion.DiffusionMLP
        horizon_steps: ${horizon_steps}
        action_dim: ${action_dim}
        cond_dim: ${eval:'${obs_dim} * ${cond_steps}'}
        time_dim: 16
        mlp_dims: [512, 512, 512]
        activation_type: ReLU
        out_activation_type: Identity
        use_layernorm: False
        residual_style: True
      horizon_steps: ${horizon_steps}
      obs_dim: ${obs_dim}
      action_dim: ${action_dim}
      denoising_steps: ${denoising_steps}
      device: ${device}
    ema:
      decay: 0.9
""")


@pytest.mark.gpu
def test_pretraining_runs_and_its_checkpoint_feeds_fine_tuning(tmp_path, monkeypatch):
    from dppo_amd.cfg.loader import get_class, load_config
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    from dppo_amd.model.diffusion.mlp_diffusion import DiffusionMLP
    monkeypatch.setenv("DPPO_LOG_DIR", str(tmp_path))
    p = tmp_path / "pre.yaml"
    p.write_text(YAML)
    cfg = load_config(str(p))
    ds = synthetic_dataset(11, 3, 4, cond_steps=1, n_traj=24, traj_len=48, seed=1, device="cuda:0")
    agent = get_class(cfg._target_)(cfg, dataset=ds)
    w0 = agent.net.flat_params().clone()
    hist = agent.run()
    assert len(hist) == 6 and all(np.isfinite(h["loss"]) for h in hist)
    assert hist[-1]["loss"] < 0.8 * hist[0]["loss"], hist  # eps-prediction on a smooth expert: the loss falls quickly
    assert not torch.equal(agent.net.flat_params(), w0)
    assert not torch.equal(agent.ema_flat, agent.net.flat_params())  # the EMA lags the model
    ck = os.path.join(str(tmp_path), "pretrain", "checkpoint", "state_6.pt")
    data = torch.load(ck, weights_only=True)
    assert data["epoch"] == 6 and set(data) == {"epoch", "model", "ema"}
    assert "network.mlp_mean.layers.1.l1.weight" in data["ema"]
    # fine-tuning loads the EMA weights into both the frozen base policy and the fine-tuned copy (diffusion.py:77-86)
    mk = lambda: DiffusionMLP(action_dim=3, horizon_steps=4, cond_dim=11, time_dim=16, mlp_dims=[512, 512, 512],
                              activation_type="ReLU", residual_style=True)
    ft = PPODiffusion(actor=mk(), critic=CriticObs(cond_dim=11, mlp_dims=[256, 256, 256], residual_style=True),
                      horizon_steps=4, obs_dim=11, action_dim=3, device="cuda:0", denoising_steps=20,
                      ft_denoising_steps=10, gamma_denoising=0.99, clip_ploss_coef=0.01, network_path=ck)
    assert torch.equal(ft.actor.flat_params(), agent.ema_flat)
    assert torch.equal(ft.actor_ft.flat_params(), agent.ema_flat)
    smp = ft(cond={"state": ds.states[:8, None]}, deterministic=True)
    assert torch.isfinite(smp.trajectories).all()
