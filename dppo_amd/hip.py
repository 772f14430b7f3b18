"""ctypes binding of libdppo_hip.so (C ABI: include/dppo_hip.h).

The shared library is built in-tree by ``dppo_amd/csrc/build.sh`` (hipcc, gfx950) into
``dppo_amd/lib/``.  There is NO fallback: if the library cannot be loaded every compute entry point
raises, by design -- the product path is the HIP path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

# DPPO_HIP_LIB names another build of the same library (debug variants such as libdppo_hip_stamps.so); never a CPU path
LIB_PATH = os.environ.get("DPPO_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                          "libdppo_hip.so")

PREC_F32, PREC_BF16 = 0, 1
ACT_RELU, ACT_MISH = 0, 1
STAT_PG_LOSS, STAT_V_LOSS, STAT_APPROX_KL, STAT_CLIPFRAC, STAT_RATIO, STAT_ADV_MEAN, STAT_ADV_STD = range(7)
STAT_COUNT = 8

PREC_BY_NAME = {"fp32": PREC_F32, "f32": PREC_F32, "float32": PREC_F32, "bf16": PREC_BF16, "bfloat16": PREC_BF16}


class NetDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("kind", "in_dim", "hidden", "n_blocks", "out_dim", "act", "time_dim", "act_flat", "cond_dim",
                 "cond_hidden", "cond_out", "use_layernorm", "plain")]


class DiffusionCfg(C.Structure):
    _fields_ = [("use_ddim", C.c_int32), ("has_denoised_clip", C.c_int32), ("has_eps_clip", C.c_int32),
                ("has_final_clip", C.c_int32), ("denoised_clip", C.c_float), ("eps_clip", C.c_float),
                ("randn_clip", C.c_float), ("final_clip", C.c_float), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32)]


class AdamwSlot(C.Structure):  # struct dppo_adamw_slot
    _fields_ = [("params", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("step_dev", C.c_void_p), ("lr_dev", C.c_void_p), ("beta1", C.c_double),
                ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double), ("sq_norm", C.c_void_p),
                ("max_norm", C.c_double)]


class PpoCfg(C.Structure):
    _fields_ = [("ft_denoising_steps", C.c_int32), ("horizon_steps", C.c_int32), ("action_dim", C.c_int32),
                ("reward_horizon", C.c_int32), ("norm_adv", C.c_int32), ("has_adv_clip", C.c_int32),
                ("has_vclip", C.c_int32), ("pad", C.c_int32),
                ("gamma_denoising", C.c_double), ("clip_ploss_coef", C.c_double),
                ("clip_ploss_coef_base", C.c_double), ("clip_ploss_coef_rate", C.c_double),
                ("clip_vloss_coef", C.c_double), ("adv_clip_lo", C.c_float), ("adv_clip_hi", C.c_float)]


class GaussianCfg(C.Structure):  # struct dppo_gaussian_cfg
    _fields_ = [("horizon_steps", C.c_int32), ("action_dim", C.c_int32), ("tanh_mean", C.c_int32), ("std_mode", C.c_int32),
                ("norm_adv", C.c_int32), ("has_vclip", C.c_int32), ("deterministic", C.c_int32), ("pad", C.c_int32),
                ("fixed_std", C.c_float), ("logvar_min", C.c_float), ("logvar_max", C.c_float), ("randn_clip", C.c_float),
                ("clip_ploss_coef", C.c_double), ("clip_vloss_coef", C.c_double), ("seed_lo", C.c_uint32),
                ("seed_hi", C.c_uint32)]


GAUSS_STAT_ENTROPY, GAUSS_STAT_STD, GAUSS_STAT_COUNT = 7, 8, 9


class UnetDesc(C.Structure):  # struct dppo_unet_desc
    _fields_ = [("action_dim", C.c_int32), ("cond_dim", C.c_int32), ("horizon_steps", C.c_int32), ("time_dim", C.c_int32),
                ("dim", C.c_int32), ("n_levels", C.c_int32), ("mults", C.c_int32 * 4), ("kernel_size", C.c_int32),
                ("n_groups", C.c_int32), ("larger_encoder", C.c_int32), ("cond_predict_scale", C.c_int32),
                ("act", C.c_int32), ("groupnorm_eps", C.c_float)]

class VisDesc(C.Structure):  # struct dppo_vis_desc
    _fields_ = [("in_ch", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32), ("embed_dim", C.c_int32),
                ("num_heads", C.c_int32), ("depth", C.c_int32), ("embed_norm", C.c_int32), ("prop_dim", C.c_int32),
                ("spatial_emb", C.c_int32), ("num_img", C.c_int32)]


class GmmCfg(C.Structure):  # struct dppo_gmm_cfg
    _fields_ = [("horizon_steps", C.c_int32), ("action_dim", C.c_int32), ("num_modes", C.c_int32), ("std_mode", C.c_int32),
                ("norm_adv", C.c_int32), ("has_vclip", C.c_int32), ("deterministic", C.c_int32), ("pad", C.c_int32),
                ("fixed_std", C.c_float), ("logvar_min", C.c_float), ("logvar_max", C.c_float), ("ent_coef", C.c_float),
                ("clip_ploss_coef", C.c_double), ("clip_vloss_coef", C.c_double), ("seed_lo", C.c_uint32),
                ("seed_hi", C.c_uint32)]


DP_HOOK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)  # void (*)(void* user, dppo_stream_t side)


class DpHook(C.Structure):  # struct dppo_dp_hook
    _fields_ = [("critic_grads_enqueued", DP_HOOK_FN), ("user", C.c_void_p)]


class ObsIO(C.Structure):  # struct dppo_obs_io
    _fields_ = [("obs_critic", C.c_void_p), ("d_obs_actor", C.c_void_p), ("d_obs_critic", C.c_void_p)]


# numpy mirror of `dppo_step` (40 bytes) so schedules are built vectorised on the host
STEP_DTYPE = np.dtype([("net", "<i4"), ("t", "<i4"), ("chain_slot", "<i4"), ("final_clip", "<i4"),
                       ("c0", "<f4"), ("c1", "<f4"), ("c2", "<f4"), ("c3", "<f4"), ("std", "<f4"), ("pad", "<f4")])
assert STEP_DTYPE.itemsize == 40

# every symbol include/dppo_hip.h declares: name -> (restype, argtypes)
_P, _I, _L, _D = C.c_void_p, C.c_int, C.c_int64, C.c_double
_ND = C.POINTER(NetDesc)
SYMBOLS = {
    "dppo_version": (C.c_int, []),
    "dppo_last_error": (C.c_char_p, []),
    "dppo_net_param_count": (_L, [_ND]),
    "dppo_packed_bytes": (_L, [_ND, _I, _I]),
    "dppo_pack_net": (_I, [_ND, _I, _I, _P, _P, _P]),
    "dppo_mlp_forward_workspace_bytes": (_L, [_ND, _I, _L]),
    "dppo_actor_forward": (_I, [_ND, _I, _P, _P, _P, _P, _P, _L, _P, _P, _L, _P]),
    "dppo_critic_forward": (_I, [_ND, _I, _P, _P, _P, _L, _P, _P, _L, _P]),
    "dppo_sample_chain_workspace_bytes": (_L, [_ND, _I, _L]),
    "dppo_sample_chain_exchange_bytes": (_L, [_ND, _I, _L]),
    "dppo_sample_chain": (_I, [_ND, _I, _P, _P, _P, _P, C.POINTER(DiffusionCfg), _P, _I, _P, _P, _L, _P, _P, _I, _I,
                               _P, _L, _P]),
    "dppo_plain_sample_workspace_bytes": (_L, [_ND, _I, _L]),
    "dppo_plain_sample_chain": (_I, [_ND, _I, _P, _P, _P, _P, C.POINTER(DiffusionCfg), _P, _I, _P, _P, _L, _P, _P, _I, _I, _P,
                                     _L, _P]),
    "dppo_chain_logprob_workspace_bytes": (_L, [_ND, _I, _L, _I]),
    "dppo_chain_logprob": (_I, [_ND, _I, _P, _P, C.POINTER(DiffusionCfg), _P, _I, _P, _P, _L, _P, _P, _L, _P]),
    "dppo_bc_loss_workspace_bytes": (_L, [_ND, _I, _L, _I]),
    "dppo_bc_loss_fwd_bwd": (_I, [_ND, _I, _P, _P, C.POINTER(DiffusionCfg), _P, _I, _P, _P, _L, _P, _P, _P, _L, _P]),
    "dppo_axpy": (_I, [_P, _P, _D, _L, _P]),
    "dppo_denoise_mse_workspace_bytes": (_L, [_ND, _I, _L]),
    "dppo_denoise_mse_fwd_bwd": (_I, [_ND, _I, _P, _P, _P, _I, _P, _P, _P, _L, _P, _P, _P, _L, _P]),
    "dppo_gae": (_I, [_P, _P, _P, _P, _I, _I, _D, _D, _D, _P, _P, _P, _P, _P]),
    "dppo_ppo_workspace_bytes": (_L, [_ND, _ND, _I, _L]),
    "dppo_ppo_loss_fwd_bwd": (_I, [_ND, _ND, _I, _P, _P, _P, _P, C.POINTER(DiffusionCfg), C.POINTER(PpoCfg), _P,
                                   _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _L, _P]),
    "dppo_ppo_loss_fwd_bwd_dp": (_I, [_ND, _ND, _I, _P, _P, _P, _P, C.POINTER(DiffusionCfg), C.POINTER(PpoCfg), _P,
                                      _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _L, _P, C.POINTER(DpHook)]),
    "dppo_gaussian_workspace_bytes": (_L, [_ND, _ND, _I, _L]),
    "dppo_gaussian_sample": (_I, [_ND, _I, _P, _P, C.POINTER(GaussianCfg), _P, _P, _P, _L, _P, _P, _P, _L, _P]),
    "dppo_gaussian_logprob": (_I, [_ND, _I, _P, _P, C.POINTER(GaussianCfg), _P, _P, _P, _L, _P, _P, _L, _P]),
    "dppo_gaussian_ppo_loss_fwd_bwd": (_I, [_ND, _ND, _I, _P, _P, _P, _P, C.POINTER(GaussianCfg), _P, _P, _P, _P, _P, _P,
                                            _P, _L, _P, _P, _P, _P, _P, _P, _L, _P]),
    "dppo_unet_param_count": (_L, [C.POINTER(UnetDesc)]),
    "dppo_unet_packed_bytes": (_L, [C.POINTER(UnetDesc), _I, _I]),
    "dppo_gmm_workspace_bytes": (_L, [_ND, _ND, _ND, _I, _L]),
    "dppo_gmm_sample": (_I, [_ND, _ND, _I, _P, _P, _P, _P, C.POINTER(GmmCfg), _P, _P, _P, _P, _L, _P, _P, _L, _P]),
    "dppo_gmm_logprob": (_I, [_ND, _ND, _I, _P, _P, _P, _P, C.POINTER(GmmCfg), _P, _P, _P, _L, _P, _P, _L, _P]),
    "dppo_gmm_ppo_loss_fwd_bwd": (_I, [_ND, _ND, _ND, _I, _P, _P, _P, _P, _P, _P, C.POINTER(GmmCfg), _P, _P, _P, _P, _P, _P, _P,
                                       _L, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    # the *_obs entries: pre-gathered mode only (no `inds`), + dppo_obs_io* / d_obs
    "dppo_ppo_loss_fwd_bwd_obs": (_I, [_ND, _ND, _I, _P, _P, _P, _P, C.POINTER(DiffusionCfg), C.POINTER(PpoCfg), _P,
                                       _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _L, _P, C.POINTER(ObsIO)]),
    "dppo_unet_ppo_loss_fwd_bwd_obs": (_I, [C.POINTER(UnetDesc), _ND, _I, _P, _P, _P, _P, C.POINTER(DiffusionCfg),
                                            C.POINTER(PpoCfg), _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _L, _P,
                                            C.POINTER(ObsIO)]),
    "dppo_gaussian_ppo_loss_fwd_bwd_obs": (_I, [_ND, _ND, _I, _P, _P, _P, _P, C.POINTER(GaussianCfg), _P, _P, _P, _P, _P, _P,
                                            _P, _L, _P, _P, _P, _P, _P, _P, _L, _P, C.POINTER(ObsIO)]),
    "dppo_denoise_mse_fwd_bwd_obs": (_I, [_ND, _I, _P, _P, _P, _I, _P, _P, _P, _L, _P, _P, _P, _L, _P, _P]),
    "dppo_unet_denoise_mse_fwd_bwd_obs": (_I, [C.POINTER(UnetDesc), _I, _P, _P, _P, _I, _P, _P, _P, _L, _P, _P, _P, _L, _P, _P]),
    "dppo_vis_param_count": (_L, [C.POINTER(VisDesc)]),
    "dppo_vis_packed_bytes": (_L, [C.POINTER(VisDesc), _I]),
    "dppo_vis_pack": (_I, [C.POINTER(VisDesc), _I, _P, _P, _P]),
    "dppo_vis_workspace_bytes": (_L, [C.POINTER(VisDesc), _I, _L, _I]),
    "dppo_vis_encode": (_I, [C.POINTER(VisDesc), _I, _P, _P, _P, _I, _P, _L, _P, _I, _I, _P, _L, _P]),
    "dppo_vis_backward": (_I, [C.POINTER(VisDesc), _I, _P, _P, _P, _I, _L, _P, _P, _L, _P]),
    "dppo_unet_pack": (_I, [C.POINTER(UnetDesc), _I, _I, _P, _P, _P]),
    "dppo_unet_workspace_bytes": (_L, [C.POINTER(UnetDesc), _I, _L]),
    "dppo_unet_sample_workspace_bytes": (_L, [C.POINTER(UnetDesc), _I, _L, _I]),
    "dppo_unet_forward": (_I, [C.POINTER(UnetDesc), _I, _P, _P, _P, _P, _P, _L, _P, _P, _L, _P]),
    "dppo_unet_sample_chain": (_I, [C.POINTER(UnetDesc), _I, _P, _P, _P, _P, C.POINTER(DiffusionCfg), _P, _I, _P, _P, _L, _P,
                                    _P, _I, _I, _P, _L, _P]),
    "dppo_unet_chain_logprob": (_I, [C.POINTER(UnetDesc), _I, _P, _P, C.POINTER(DiffusionCfg), _P, _P, _I, _P, _P, _L, _P, _P,
                                     _L, _P]),
    "dppo_unet_ppo_workspace_bytes": (_L, [C.POINTER(UnetDesc), _ND, _I, _L]),
    "dppo_unet_ppo_loss_fwd_bwd": (_I, [C.POINTER(UnetDesc), _ND, _I, _P, _P, _P, _P, C.POINTER(DiffusionCfg),
                                        C.POINTER(PpoCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _L, _P]),
    "dppo_unet_denoise_mse_workspace_bytes": (_L, [C.POINTER(UnetDesc), _I, _L]),
    "dppo_unet_denoise_mse_fwd_bwd": (_I, [C.POINTER(UnetDesc), _I, _P, _P, _P, _I, _P, _P, _P, _L, _P, _P, _P, _L, _P]),
    "dppo_grad_sq_norm": (_I, [_P, _L, _P, _P, _P]),
    "dppo_adamw_step": (_I, [_P, _P, _P, _P, _L, _I, _D, _D, _D, _D, _D, _P, _D, _P]),
    "dppo_adamw_step_dev": (_I, [_P, _P, _P, _P, _L, _P, _P, _D, _D, _D, _D, _P, _D, _P]),
    "dppo_adamw_step_multi": (_I, [_P, _I, _P]),
    "dppo_pack_nets": (_I, [_ND, _I, _P, _P, _ND, _I, _P, _P, _I, _P]),
    "dppo_stats_split": (_I, [_P, _P, _P]),
    "dppo_stats_merge": (_I, [_P, _P, _I, _P]),
    "dppo_stats_merge_n": (_I, [_P, _P, _I, _I, _P]),
    "dppo_probe_arm": (_I, [_I, _I]),
    "dppo_probe_collect": (_I, [C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "dppo_probe_collect_bytes": (_I, [C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double)]),
    "dppo_tune_set": (_I, [_I, _I]),
    "dppo_gemm_nt_raw": (_I, [_I, _P, _P, _P, _L, _I, _I, _P, _P, _I, _I, _P]),
    "dppo_gemm_tn_raw": (_I, [_I, _P, _I, _I, _P, _I, _I, _L, _I, _P, _P, _P]),
}

_lib: Optional[C.CDLL] = None


class DppoHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises DppoHipError when it is missing -- there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DppoHipError(
                f"{LIB_PATH} not found: build it with dppo_amd/csrc/build.sh (or __graft_entry__.build()). "
                "dppo_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        for kv in filter(None, os.environ.get("DPPO_TUNE", "").split(",")):  # e.g. DPPO_TUNE=2=0,7=1 (dppo_tune_set knobs)
            k, v = kv.split("=")
            if lib.dppo_tune_set(int(k), int(v)) != 0:
                raise DppoHipError(f"DPPO_TUNE: bad knob {kv!r}: {lib.dppo_last_error().decode()}")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().dppo_last_error().decode()
        raise DppoHipError(f"{what} failed (rc={rc}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_contiguous(), "dppo_amd kernels take contiguous tensors"
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> int:
    """hipStream_t of torch's current stream on the current device (called once per library call: the raw getter costs
    0.3 us where ``torch.cuda.current_stream().cuda_stream`` costs 10)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise DppoHipError(f"{what}: tensor is on {t.device}; the DPPO hot path runs on an MI355X only "
                           "(no CPU fallback)")


class Workspace:
    """Grow-only device scratch buffer (one per purpose, reused across calls)."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            # zero-filled: a sampling workspace begins with the sticky time-out word (include/dppo_hip.h, dppo_sample_chain)
            self.buf = torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        return self.buf
