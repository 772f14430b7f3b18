"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the header declares,
argument validation works without a GPU, and the product path refuses to run on CPU (no silent fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from dppo_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "dppo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dppo_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = hip.load()
    names = header_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"libdppo_hip.so does not export {n}"
    assert sorted(hip.SYMBOLS) == names, "ctypes table and header disagree"
    assert lib.dppo_version() == 1


def hopper_desc():
    return hip.NetDesc(kind=0, in_dim=39, hidden=512, n_blocks=1, out_dim=12, act=hip.ACT_RELU, time_dim=16,
                       act_flat=12, cond_dim=11, cond_hidden=0, cond_out=0)


def test_param_count_matches_reference_sizes():
    lib = hip.load()
    assert lib.dppo_net_param_count(C.byref(hopper_desc())) == 553020
    critic = hip.NetDesc(kind=1, in_dim=11, hidden=256, n_blocks=1, out_dim=1, act=hip.ACT_MISH, time_dim=0,
                         act_flat=0, cond_dim=11, cond_hidden=0, cond_out=0)
    assert lib.dppo_net_param_count(C.byref(critic)) == 134913
    for prec in (hip.PREC_F32, hip.PREC_BF16):
        assert lib.dppo_packed_bytes(C.byref(hopper_desc()), prec, 20) > 553020
        assert lib.dppo_ppo_workspace_bytes(C.byref(hopper_desc()), C.byref(critic), prec, 50000) > 0


def test_bad_arguments_are_rejected_with_a_message():
    lib = hip.load()
    d = hopper_desc()
    d.hidden = 100
    assert lib.dppo_net_param_count(C.byref(d)) == -1
    assert b"hidden" in lib.dppo_last_error()
    d = hopper_desc()
    d.in_dim = 40
    assert lib.dppo_packed_bytes(C.byref(d), hip.PREC_BF16, 20) == -1
    assert lib.dppo_packed_bytes(C.byref(hopper_desc()), 7, 20) == -1
    assert lib.dppo_gae(None, None, None, None, 1, 1, 0.99, 0.95, 1.0, None, None, None, None, None) == -1


def test_python_layout_matches_c_abi_and_reference_names():
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.mlp_diffusion import DiffusionMLP
    a = DiffusionMLP(3, 4, 11, mlp_dims=[512, 512, 512], activation_type="ReLU", residual_style=True)
    c = CriticObs(11, [256, 256, 256], residual_style=True)
    assert list(dict(a.named_parameters())) == [
        "time_embedding.1.weight", "time_embedding.1.bias", "time_embedding.3.weight", "time_embedding.3.bias",
        "mlp_mean.layers.0.weight", "mlp_mean.layers.0.bias", "mlp_mean.layers.1.l1.weight",
        "mlp_mean.layers.1.l1.bias", "mlp_mean.layers.1.l2.weight", "mlp_mean.layers.1.l2.bias",
        "mlp_mean.layers.2.weight", "mlp_mean.layers.2.bias"]
    assert list(dict(c.named_parameters()))[:2] == ["Q1.layers.0.weight", "Q1.layers.0.bias"]
    lib = hip.load()
    assert a.flat_params().numel() == lib.dppo_net_param_count(C.byref(a.net_desc()))
    assert c.flat_params().numel() == lib.dppo_net_param_count(C.byref(c.net_desc()))
    # parameters are views of the flat image, in state-dict order
    f = a.flat_params()
    off = 0
    for p in a.parameters():
        assert p.data_ptr() == f.data_ptr() + 4 * off
        off += p.numel()


def test_no_cpu_fallback():
    from dppo_amd.model.common.critic import CriticObs
    c = CriticObs(11, [256, 256, 256], residual_style=True)
    with pytest.raises(hip.DppoHipError):
        c({"state": torch.zeros(2, 1, 11)})


def test_unsupported_variants_fail_loudly():
    from dppo_amd.model.diffusion.mlp_diffusion import DiffusionMLP
    with pytest.raises(NotImplementedError):  # LayerNorm blocks exist only on the fused kernels' widths
        DiffusionMLP(3, 4, 11, mlp_dims=[384, 384, 384], residual_style=True, use_layernorm=True)
    with pytest.raises(NotImplementedError):  # plain trunks: one hidden width, no LayerNorm / cond_mlp
        DiffusionMLP(3, 4, 11, mlp_dims=[512, 512, 512], residual_style=False, use_layernorm=True)
    with pytest.raises(NotImplementedError):
        DiffusionMLP(3, 4, 11, mlp_dims=[512, 256], residual_style=False)
    d = hopper_desc()
    d.use_layernorm, d.hidden = 1, 384
    assert hip.load().dppo_net_param_count(C.byref(d)) < 0


def test_plain_mlp_layout_matches_reference_names():
    """residual_style=False: moduleList.{i}.linear_1 names (reference mlp.py:46-74) and the C ABI's flat size."""
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.mlp_diffusion import DiffusionMLP
    a = DiffusionMLP(3, 4, 11, mlp_dims=[64, 64], residual_style=False)
    names = [n for n in dict(a.named_parameters()) if n.startswith("mlp_mean.")]
    assert names == [f"mlp_mean.moduleList.{i}.linear_1.{w}" for i in range(3) for w in ("weight", "bias")]
    d = a.net_desc()
    assert d.plain == 1 and d.n_blocks == 1 and d.hidden == 64
    assert a.flat_params().numel() == hip.load().dppo_net_param_count(C.byref(d))
    c = CriticObs(cond_dim=11, mlp_dims=[64, 64], residual_style=False)
    assert c.flat_params().numel() == hip.load().dppo_net_param_count(C.byref(c.net_desc())) == 11 * 64 + 64 + 64 * 64 + 64 + 64 + 1


def test_layernorm_layout_matches_reference_names():
    from dppo_amd.model.diffusion.mlp_diffusion import DiffusionMLP
    a = DiffusionMLP(3, 4, 11, mlp_dims=[512, 512, 512], residual_style=True, use_layernorm=True)
    names = [n for n in dict(a.named_parameters()) if n.startswith("mlp_mean.layers.1.")]
    assert names == [f"mlp_mean.layers.1.{m}.{w}" for m in ("l1", "l2", "norm1", "norm2") for w in ("weight", "bias")]
    assert a.flat_params().numel() == hip.load().dppo_net_param_count(C.byref(a.net_desc())) == 553020 + 4 * 512


def test_cond_mlp_layout_matches_reference_names():
    from dppo_amd.model.diffusion.mlp_diffusion import DiffusionMLP
    a = DiffusionMLP(9, 4, 60, mlp_dims=[256, 256, 256], residual_style=True, cond_mlp_dims=[128, 32])
    names = list(dict(a.named_parameters()))
    assert names[4:8] == ["cond_mlp.moduleList.0.linear_1.weight", "cond_mlp.moduleList.0.linear_1.bias",
                          "cond_mlp.moduleList.1.linear_1.weight", "cond_mlp.moduleList.1.linear_1.bias"]
    assert names[8] == "mlp_mean.layers.0.weight" and tuple(a.mlp_mean.layers[0].weight.shape) == (256, 36 + 16 + 32)
    lib = hip.load()
    assert a.flat_params().numel() == lib.dppo_net_param_count(C.byref(a.net_desc()))


def test_new_entry_points_reject_bad_descriptors_with_a_message():
    """Argument validation of the round-2 entries runs on the host (no GPU needed): every refusal leaves its reason in
    dppo_last_error()."""
    lib = hip.load()
    v = hip.VisDesc(in_ch=3, img_h=96, img_w=96, embed_dim=128, num_heads=4, depth=1, embed_norm=0, prop_dim=9, spatial_emb=128,
                    num_img=1)
    assert lib.dppo_vis_param_count(C.byref(v)) == 419712
    for field, value, word in (("img_h", 97, b"image"), ("embed_dim", 100, b"embed_dim"), ("num_heads", 3, b"num_heads"),
                               ("depth", 9, b"depth"), ("num_img", 3, b"num_img"), ("embed_norm", 1, b"embed_norm"),
                               ("img_w", 512, b"patches")):
        bad = hip.VisDesc.from_buffer_copy(v)
        setattr(bad, field, value)
        assert lib.dppo_vis_param_count(C.byref(bad)) == -1, field
        assert word in lib.dppo_last_error(), (field, lib.dppo_last_error())
    assert lib.dppo_vis_workspace_bytes(C.byref(v), hip.PREC_BF16, 0, 1) == -1
    # plain trunks: multiples of 64, no LayerNorm / cond_mlp; wide outputs only off the denoiser
    d = hopper_desc()
    d.plain, d.hidden, d.in_dim = 1, 64, 39
    assert lib.dppo_net_param_count(C.byref(d)) == 2 * 16 * 16 + 2 * 16 + 2 * 16 * 16 + 16 + 39 * 64 + 64 + 64 * 64 + 64 + 12 * 64 + 12
    d.use_layernorm = 1
    assert lib.dppo_net_param_count(C.byref(d)) == -1 and b"plain" in lib.dppo_last_error()
    d = hopper_desc()
    d.out_dim = d.act_flat = 140
    d.in_dim = 140 + 16 + 11
    assert lib.dppo_net_param_count(C.byref(d)) == -1 and b"out_dim" in lib.dppo_last_error()
    wide = hip.NetDesc(kind=1, in_dim=23, hidden=512, n_blocks=1, out_dim=140, act=hip.ACT_MISH, time_dim=0, act_flat=0,
                       cond_dim=23, cond_hidden=0, cond_out=0)
    assert lib.dppo_net_param_count(C.byref(wide)) == 23 * 512 + 512 + 2 * (512 * 512 + 512) + 140 * 512 + 140
    assert lib.dppo_sample_chain_workspace_bytes(C.byref(d), hip.PREC_BF16, 16) == -1
    # GMM: the two trunks must agree with the cfg
    wts = hip.NetDesc(kind=1, in_dim=23, hidden=512, n_blocks=1, out_dim=5, act=hip.ACT_MISH, time_dim=0, act_flat=0, cond_dim=23,
                      cond_hidden=0, cond_out=0)
    crit = hip.NetDesc(kind=1, in_dim=23, hidden=256, n_blocks=1, out_dim=1, act=hip.ACT_MISH, time_dim=0, act_flat=0, cond_dim=23,
                       cond_hidden=0, cond_out=0)
    assert lib.dppo_gmm_workspace_bytes(C.byref(wide), C.byref(wts), C.byref(crit), hip.PREC_BF16, 64) > 0
    cfg = hip.GmmCfg(horizon_steps=4, action_dim=7, num_modes=4, std_mode=0, fixed_std=0.1)
    assert lib.dppo_gmm_logprob(C.byref(wide), C.byref(wts), hip.PREC_BF16, None, None, None, None, C.byref(cfg), None, None, None,
                                4, None, None, 0, None) == -1
    assert b"out_dim" in lib.dppo_last_error()
