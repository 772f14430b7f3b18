"""The YAML surface: a cfg written in the reference's style (same keys as
cfg/gym/finetune/hopper-v2/ft_ppo_diffusion_mlp.yaml, typed in here -- not a copy of the file) resolves and
instantiates the dppo_amd classes through `_target_` strings that still name the reference package."""
import os
import textwrap

import pytest

from dppo_amd.cfg.loader import get_class, instantiate, load_config

YAML = textwrap.dedent("""
    defaults:
      - _self_
    _target_: dppo.agent.finetune.train_ppo_diffusion_agent.TrainPPODiffusionAgent
    name: ${env_name}_ppo_diffusion_mlp_ta${horizon_steps}_td${denoising_steps}_tdf${ft_denoising_steps}
    logdir: ${oc.env:DPPO_LOG_DIR}/gym-finetune/${name}/${now:%Y-%m-%d}_${seed}
    seed: 42
    device: cuda:0
    env_name: hopper-medium-v2
    obs_dim: 11
    action_dim: 3
    denoising_steps: 20
    ft_denoising_steps: 10
    cond_steps: 1
    horizon_steps: 4
    act_steps: 4
    train:
      n_steps: 500
      batch_size: 50000
      actor_lr: 1e-4
    model:
      _target_: dppo.model.diffusion.diffusion_ppo.PPODiffusion
      gamma_denoising: 0.99
      clip_ploss_coef: 0.01
      randn_clip_value: 3
      network_path: null
      actor:
        _target_: dppo.model.diffusion.mlp_diffusion.DiffusionMLP
        time_dim: 16
        mlp_dims: [512, 512, 512]
        activation_type: ReLU
        residual_style: True
        cond_dim: ${eval:'${obs_dim} * ${cond_steps}'}
        horizon_steps: ${horizon_steps}
        action_dim: ${action_dim}
      critic:
        _target_: dppo.model.common.critic.CriticObs
        cond_dim: ${eval:'${obs_dim} * ${cond_steps}'}
        mlp_dims: [256, 256, 256]
        activation_type: Mish
        residual_style: True
      ft_denoising_steps: ${ft_denoising_steps}
      horizon_steps: ${horizon_steps}
      obs_dim: ${obs_dim}
      action_dim: ${action_dim}
      denoising_steps: ${denoising_steps}
      device: ${device}
""")


def test_resolve_and_override(tmp_path, monkeypatch):
    monkeypatch.setenv("DPPO_LOG_DIR", "/tmp/logs")
    p = tmp_path / "cfg.yaml"
    p.write_text(YAML)
    cfg = load_config(str(p), overrides=["train.n_steps=7", "device=cpu", "cond_steps=2"])
    assert cfg.name == "hopper-medium-v2_ppo_diffusion_mlp_ta4_td20_tdf10"
    assert cfg.logdir.startswith("/tmp/logs/gym-finetune/hopper-medium-v2_ppo") and cfg.logdir.endswith("_42")
    assert cfg.train.n_steps == 7 and cfg.train.get("gae_lambda", 0.95) == 0.95
    assert cfg.model.actor.cond_dim == 22 and isinstance(cfg.model.actor.cond_dim, int)
    assert cfg.train.actor_lr == pytest.approx(1e-4)
    assert cfg.model.network_path is None


def test_missing_env_var_is_an_error(tmp_path, monkeypatch):
    monkeypatch.delenv("DPPO_LOG_DIR", raising=False)
    p = tmp_path / "cfg.yaml"
    p.write_text(YAML)
    with pytest.raises(KeyError):
        load_config(str(p))


def test_targets_map_onto_this_package(tmp_path, monkeypatch):
    monkeypatch.setenv("DPPO_LOG_DIR", "/tmp/logs")
    p = tmp_path / "cfg.yaml"
    p.write_text(YAML)
    cfg = load_config(str(p), overrides=["device=cpu"])
    model = instantiate(cfg.model)  # parameters only; no kernel runs on CPU
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    assert isinstance(model, PPODiffusion)
    assert sum(p.numel() for p in model.actor_ft.parameters()) == 553020
    keys = list(model.state_dict())
    assert keys[0] == "network.time_embedding.1.weight" and "actor_ft.mlp_mean.layers.1.l2.bias" in keys
    assert "critic.Q1.layers.2.weight" in keys
    assert get_class(cfg._target_).__name__ == "TrainPPODiffusionAgent"


REF_CFG = "/root/reference/dppo/cfg"


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference checkout not present (GPU box)")
def test_every_shipped_diffusion_mlp_cfg_builds():
    """Each ft_ppo_diffusion_mlp.yaml the reference ships resolves through the loader, its actor / critic build as
    dppo_amd containers, and the C ABI accepts both descriptors (parameter count == the containers' flat size)."""
    import ctypes as C
    import glob

    from dppo_amd import hip
    os.environ.setdefault("DPPO_LOG_DIR", "/tmp/log")
    os.environ.setdefault("DPPO_DATA_DIR", "/tmp/data")
    os.environ.setdefault("DPPO_WANDB_ENTITY", "none")
    paths = sorted(glob.glob(os.path.join(REF_CFG, "*", "finetune", "*", "ft_ppo_diffusion_mlp.yaml")))
    assert len(paths) >= 19
    lib = hip.load()
    for p in paths:
        cfg = load_config(p)
        actor = instantiate(cfg.model.actor)
        critic = instantiate(cfg.model.critic)
        for net in (actor, critic):
            n = lib.dppo_net_param_count(C.byref(net.net_desc()))
            assert n == net.flat_params().numel(), (p, hip.last_error())
            for prec in (hip.PREC_F32, hip.PREC_BF16):
                assert lib.dppo_packed_bytes(C.byref(net.net_desc()), prec, int(cfg.denoising_steps)) > 0, p


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference checkout not present (GPU box)")
def test_every_shipped_pretraining_cfg_builds():
    """Each pre_diffusion_mlp.yaml the reference ships resolves, maps onto dppo_amd's pre-training agent and model, and
    its network is one the C ABI accepts (the supervised loss runs on the update's kernels)."""
    import ctypes as C
    import glob

    from dppo_amd import hip
    from dppo_amd.agent.pretrain.train_diffusion_agent import TrainDiffusionAgent
    from dppo_amd.model.diffusion.diffusion import DiffusionModel
    os.environ.setdefault("DPPO_LOG_DIR", "/tmp/log")
    os.environ.setdefault("DPPO_DATA_DIR", "/tmp/data")
    os.environ.setdefault("DPPO_WANDB_ENTITY", "none")
    paths = sorted(glob.glob(os.path.join(REF_CFG, "*", "pretrain", "*", "pre_diffusion_mlp.yaml")))
    assert len(paths) >= 19
    lib = hip.load()
    built = 0
    for p in paths:
        cfg = load_config(p, overrides=["device=cpu"])
        assert get_class(cfg._target_) is TrainDiffusionAgent, p
        assert get_class(cfg.model._target_) is DiffusionModel, p
        net = instantiate(cfg.model.network)
        assert lib.dppo_net_param_count(C.byref(net.net_desc())) == net.flat_params().numel(), (p, hip.last_error())
        assert cfg.train.batch_size >= 1 and cfg.ema.decay < 1
        built += 1
    assert built == len(paths)


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference checkout not present (GPU box)")
def test_every_shipped_state_unet_gaussian_and_eval_cfg_builds():
    """Round 2's rows through the same loader: every ft_ppo_diffusion_unet.yaml / pre_diffusion_unet.yaml (conv actor: model
    builds on CPU, state-dict order = the C ABI's flat layout), every ft_ppo_gaussian_mlp.yaml with a fixed-std residual trunk,
    and every eval_diffusion_{mlp,unet}.yaml's model target (DiffusionEval, incl. the `diffusion_eval_ft` module path)."""
    import ctypes as C
    import glob

    from dppo_amd import hip
    from dppo_amd.model.diffusion.diffusion_eval import DiffusionEval
    from dppo_amd.model.diffusion.unet import Unet1D
    from dppo_amd.model.rl.gaussian_ppo import PPO_Gaussian
    os.environ.setdefault("DPPO_LOG_DIR", "/tmp/log")
    os.environ.setdefault("DPPO_DATA_DIR", "/tmp/data")
    os.environ.setdefault("DPPO_WANDB_ENTITY", "none")
    lib = hip.load()
    n_unet = 0
    for pat in ("*/finetune/*/ft_ppo_diffusion_unet.yaml", "*/pretrain/*/pre_diffusion_unet.yaml"):
        for p in sorted(glob.glob(os.path.join(REF_CFG, pat))):
            cfg = load_config(p, overrides=["device=cpu"])
            node = cfg.model.actor if "actor" in cfg.model else cfg.model.network
            net = instantiate(node)
            assert isinstance(net, Unet1D), p
            net.horizon_steps = int(cfg.horizon_steps)
            d = net.net_desc()
            assert lib.dppo_unet_param_count(C.byref(d)) == sum(q.numel() for q in net.parameters()), (p, hip.load().dppo_last_error())
            assert lib.dppo_unet_workspace_bytes(C.byref(d), hip.PREC_BF16, 64) > 0
            assert lib.dppo_unet_ppo_workspace_bytes(C.byref(d), None, hip.PREC_BF16, 64) == -1  # needs a critic descriptor
            n_unet += 1
    assert n_unet >= 20  # incl. the dim: 40 cfgs (robomimic can / lift): maps padded to 64 / 128 channels in their images
    n_gauss = 0
    for p in sorted(glob.glob(os.path.join(REF_CFG, "*", "finetune", "*", "ft_ppo_gaussian_mlp.yaml"))):
        cfg = load_config(p, overrides=["device=cpu"])
        if cfg.model.actor.get("fixed_std", None) is None or not cfg.model.actor.get("residual_style", False):
            with pytest.raises(NotImplementedError):
                instantiate(cfg.model.actor)
            continue
        cfg.model.network_path = None  # (the pre-trained checkpoint the cfg points at is a download)
        model = instantiate(cfg.model)
        assert isinstance(model, PPO_Gaussian) and get_class(cfg._target_).__name__ == "TrainPPOGaussianAgent", p
        n_gauss += 1
    assert n_gauss >= 7
    n_eval = 0
    for pat in ("*/eval/*/eval_diffusion_mlp.yaml", "*/eval/*/eval_diffusion_unet.yaml"):
        for p in sorted(glob.glob(os.path.join(REF_CFG, pat))):
            cfg = load_config(p, overrides=["device=cpu"])
            assert get_class(cfg.model._target_) is DiffusionEval, p
            n_eval += 1
    assert n_eval >= 10


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference checkout not present (GPU box)")
def test_every_shipped_pixel_diffusion_cfg_builds():
    """The image cfgs (ft_ppo_diffusion_{mlp,unet}_img.yaml, pre_diffusion_{mlp,unet}_img.yaml, eval_diffusion_*_img.yaml):
    the pixel networks build on CPU, their state dict splits into encoder | trunk in the C ABI's two flat layouts, and the
    fine-tuning cfgs resolve to the image agent with a ViTCritic."""
    import ctypes as C
    import glob

    from dppo_amd import hip
    from dppo_amd.model.common.critic import ViTCritic
    os.environ.setdefault("DPPO_LOG_DIR", "/tmp/log")
    os.environ.setdefault("DPPO_DATA_DIR", "/tmp/data")
    os.environ.setdefault("DPPO_WANDB_ENTITY", "none")
    lib = hip.load()
    n = 0
    for pat in ("*/finetune/*/ft_ppo_diffusion_mlp_img.yaml", "*/finetune/*/ft_ppo_diffusion_unet_img.yaml",
                "*/pretrain/*/pre_diffusion_mlp_img.yaml", "*/pretrain/*/pre_diffusion_unet_img.yaml",
                "*/eval/*/eval_diffusion_mlp_img.yaml", "*/eval/*/eval_diffusion_unet_img.yaml"):
        for p in sorted(glob.glob(os.path.join(REF_CFG, pat))):
            cfg = load_config(p, overrides=["device=cpu"])
            node = cfg.model.actor if "actor" in cfg.model else cfg.model.network
            net = instantiate(node)
            assert getattr(net, "is_vision", False), p
            if getattr(net, "is_unet", False):
                net.horizon_steps = int(cfg.horizon_steps)
            vis = net.vis
            assert lib.dppo_vis_param_count(C.byref(vis.desc)) == sum(q.numel() for q in vis.trunk_parameters()), (p, lib.dppo_last_error())
            assert lib.dppo_vis_workspace_bytes(C.byref(vis.desc), hip.PREC_BF16, 64, 1) > 0
            assert vis.desc.num_img == (2 if "transport" in p else 1)
            assert net._abi_param_count() == sum(q.numel() for q in net.trunk_parameters()), p
            assert len(list(net.parameters())) == len(vis.trunk_parameters()) + len(net.trunk_parameters())
            if "finetune" in p:
                assert get_class(cfg._target_).__name__ == "TrainPPOImgDiffusionAgent", p
                assert isinstance(instantiate(cfg.model.critic), ViTCritic), p
            n += 1
    assert n >= 20
    n_g = 0
    for p in sorted(glob.glob(os.path.join(REF_CFG, "*", "finetune", "*", "ft_ppo_gaussian_mlp_img.yaml"))):
        cfg = load_config(p, overrides=["device=cpu"])
        net = instantiate(cfg.model.actor)
        assert getattr(net, "is_vision", False) and net.learn_fixed_std and net.tanh_output, p
        assert get_class(cfg._target_).__name__ == "TrainPPOImgGaussianAgent", p
        assert lib.dppo_vis_param_count(C.byref(net.vis.desc)) == sum(q.numel() for q in net.vis.trunk_parameters())
        n_g += 1
    assert n_g == 4


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference checkout not present (GPU box)")
def test_every_shipped_gmm_mlp_cfg_builds():
    """ft_ppo_gmm_mlp.yaml (d3il: plain 256 x 2 trunks; robomimic: residual trunks with 140 / 560 mean outputs): the model builds,
    the two trunk descriptors pass the C ABI's checks, and the cfg resolves to the Gaussian agent like in the reference."""
    import ctypes as C
    import glob

    from dppo_amd import hip
    from dppo_amd.model.rl.gmm_ppo import PPO_GMM
    os.environ.setdefault("DPPO_LOG_DIR", "/tmp/log")
    os.environ.setdefault("DPPO_DATA_DIR", "/tmp/data")
    os.environ.setdefault("DPPO_WANDB_ENTITY", "none")
    lib = hip.load()
    n = 0
    for p in sorted(glob.glob(os.path.join(REF_CFG, "*", "finetune", "*", "ft_ppo_gmm_mlp.yaml"))):
        cfg = load_config(p, overrides=["device=cpu"])
        cfg.model.network_path = None
        model = instantiate(cfg.model)
        assert isinstance(model, PPO_GMM) and get_class(cfg._target_).__name__ == "TrainPPOGaussianAgent", p
        net = model.actor_ft
        for t in (net.mean_net, net.weights_net):
            assert lib.dppo_net_param_count(C.byref(t.net_desc())) == t.flat_params().numel(), (p, lib.dppo_last_error())
        assert net.mean_net.net_desc().out_dim == int(cfg.action_dim) * int(cfg.horizon_steps) * int(cfg.num_modes)
        assert lib.dppo_gmm_workspace_bytes(C.byref(net.mean_net.net_desc()), C.byref(net.weights_net.net_desc()),
                                            C.byref(model.critic.net_desc()), hip.PREC_BF16, 256) > 0
        n += 1
    assert n == 7
