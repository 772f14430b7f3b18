"""Case tables shared by make_golden.py (generator, needs the reference) and the tests (no reference needed)."""
SCHED_CASES = {
    # name: (kwargs, n_steps, optimizer lr)
    "hopper_actor": (dict(first_cycle_steps=1000, cycle_mult=1.0, max_lr=1e-4, min_lr=1e-4, warmup_steps=10, gamma=1.0), 15, 1e-4),
    "warm5_gamma": (dict(first_cycle_steps=20, cycle_mult=1.0, max_lr=1e-3, min_lr=1e-5, warmup_steps=5, gamma=0.5), 50, 1e-3),
    "nowarm_mult2": (dict(first_cycle_steps=10, cycle_mult=2.0, max_lr=3e-4, min_lr=1e-6, warmup_steps=0, gamma=0.9), 45, 3e-4),
    "pretrain_like": (dict(first_cycle_steps=200, cycle_mult=1.0, max_lr=1e-3, min_lr=1e-4, warmup_steps=1, gamma=1.0), 30, 1e-3),
}

EVAL_CASES = {
    # name: (spec, B, model kwargs, ft_denoising_steps, checkpoint kind)
    "eval_ddpm_ft10": ("hopper", 6, dict(denoising_steps=20, randn_clip_value=3), 10, "rl"),
    "eval_ddim_ft5": ("hopper", 5, dict(denoising_steps=100, use_ddim=True, ddim_steps=5, randn_clip_value=3), 5, "rl"),
    "eval_pretrained_can": ("can", 4, dict(denoising_steps=20, randn_clip_value=3, final_action_clip_value=1.0), 0, "pretrain"),
}

GAUSS_CASES = {
    # name: (spec, GaussianCfg kwargs)
    "gauss_d3il_fixed": ("gauss_d3il", dict(fixed_std=0.1, learn_fixed_std=False, std_min=0.01, std_max=1.0,
                                            clip_ploss_coef=0.1, randn_clip_value=3)),
    "gauss_furniture_learned": ("gauss_furniture", dict(fixed_std=0.04, learn_fixed_std=True, std_min=0.01, std_max=0.2,
                                                        clip_ploss_coef=0.01, clip_vloss_coef=0.2, randn_clip_value=3)),
    "gauss_nonorm": ("gauss_d3il", dict(fixed_std=0.1, learn_fixed_std=True, std_min=0.05, std_max=0.12,
                                        clip_ploss_coef=0.02, norm_adv=False, randn_clip_value=3)),
}

UNET_SPECS = {
    # name: UnetSpec kwargs (+ horizon_steps)
    # shipped robomimic state cfgs (cfg/robomimic/finetune/square/ft_ppo_diffusion_unet.yaml:93-103): Ta 4, two levels
    "unet_square": dict(action_dim=7, cond_dim=23, horizon_steps=4, diffusion_step_embed_dim=16, dim=64, dim_mults=(1, 2),
                        kernel_size=5, n_groups=8, smaller_encoder=False, cond_predict_scale=True),
    # shipped furniture cfgs (cfg/furniture/finetune/one_leg_low/ft_ppo_diffusion_unet.yaml:99-109): three levels, Ta 8
    "unet_furniture": dict(action_dim=10, cond_dim=58, horizon_steps=8, diffusion_step_embed_dim=16, dim=64,
                           dim_mults=(1, 2, 4), kernel_size=5, n_groups=8, smaller_encoder=False, cond_predict_scale=True),
    # the other constructor branches: one-layer encoder, additive conditioning, ReLU, kernel 3, eps 1e-4
    "unet_small": dict(action_dim=3, cond_dim=11, horizon_steps=4, diffusion_step_embed_dim=16, dim=64, dim_mults=(1, 2),
                       kernel_size=3, n_groups=4, smaller_encoder=True, cond_predict_scale=False, activation="ReLU",
                       groupnorm_eps=1e-4),
}
UNET_CHAIN_CASES = {
    # name: (spec name, B, model kwargs, deterministic)
    "unet_ddpm20_ft10": ("unet_square", 5, dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), False),
    "unet_ddim100_5": ("unet_furniture", 3, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                                 randn_clip_value=3, min_sampling_denoising_std=0.04), False),
    "unet_small_det": ("unet_small", 4, dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), True),
}

UNET_LOSS_CASES = {
    # name: (spec name, N, model kwargs, reward_horizon)
    "unet_loss_square": ("unet_square", 32, dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01,
                                                 clip_ploss_coef_base=0.001), 4),
    "unet_loss_small": ("unet_small", 32, dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.1,
                                               clip_ploss_coef_base=0.01, clip_vloss_coef=0.2), 4),
    "unet_loss_furniture_ddim": ("unet_furniture", 16, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True,
                                                            ddim_steps=5, clip_ploss_coef=0.01,
                                                            min_sampling_denoising_std=0.04), 8),
}
UNET_MSE_CASES = {"unet_mse_square": ("unet_square", 20, 24), "unet_mse_furniture": ("unet_furniture", 100, 12)}

# the robomimic can / lift cfgs ship dim: 40 (channel counts 40 / 80, not multiples of the 64-element GEMM k-step: every map
# is padded to 64 / 128 channels in its image) -- its own fixture file, g15_unet_dim40.npz
UNET40_SPECS = {"unet_dim40": dict(action_dim=7, cond_dim=23, horizon_steps=4, diffusion_step_embed_dim=16, dim=40,
                                   dim_mults=(1, 2), kernel_size=5, n_groups=8, smaller_encoder=False, cond_predict_scale=True),
                "unet_dim40_l3": dict(action_dim=7, cond_dim=19, horizon_steps=8, diffusion_step_embed_dim=16, dim=40,
                                      dim_mults=(1, 2, 4), kernel_size=5, n_groups=8, smaller_encoder=False,
                                      cond_predict_scale=True)}
UNET40_CHAIN_CASES = {"unet40_ddpm20_ft10": ("unet_dim40", 5, dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), False)}
UNET40_LOSS_CASES = {"unet40_loss": ("unet_dim40", 32, dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01,
                                                           clip_ploss_coef_base=0.001), 4),
                     "unet40_l3_loss": ("unet_dim40_l3", 16, dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01), 8)}
UNET40_MSE_CASES = {"unet40_mse": ("unet_dim40", 20, 24)}

# ---- pixel observations: ViT encoder + SpatialEmb (model/common/vit.py, modules.py) in front of either denoiser / the critic
VIS_SPECS = {
    # shipped robomimic image cfgs (cfg/robomimic/finetune/square/ft_ppo_diffusion_mlp_img.yaml:49-56,118-140): one 96x96 camera
    "vis_square": dict(in_ch=3, img_h=96, img_w=96, prop_dim=9),
    # small maps (16 patches) with a two-frame image history (img_cond_steps = 2 -> six input channels)
    "vis_small": dict(in_ch=6, img_h=40, img_w=40, prop_dim=5),
    # two cameras sharing the backbone, one SpatialEmb each (transport: .../transport/ft_ppo_diffusion_unet_img.yaml:52-59,140)
    "vis_two": dict(in_ch=3, img_h=40, img_w=40, prop_dim=6, num_img=2),
}
# trunk behind the encoder: name -> (vis spec, kind, kwargs); cond_dim is filled in as feat_dim + prop_dim
VIS_NETS = {
    "vmlp_square": ("vis_square", "mlp", dict(mlp_dims=[768, 768, 768], activation="Mish", action_dim=7, horizon_steps=4, time_dim=32)),
    "vmlp_small": ("vis_small", "mlp", dict(mlp_dims=[256, 256, 256], activation="Mish", action_dim=3, horizon_steps=4, time_dim=16)),
    "vunet_small": ("vis_small", "unet", dict(action_dim=3, horizon_steps=4, diffusion_step_embed_dim=32, dim=64, dim_mults=(1, 2),
                                               kernel_size=5, n_groups=8, smaller_encoder=False, cond_predict_scale=True)),
    "vunet_two": ("vis_two", "unet", dict(action_dim=4, horizon_steps=8, diffusion_step_embed_dim=32, dim=64, dim_mults=(1, 2),
                                           kernel_size=5, n_groups=8, smaller_encoder=False, cond_predict_scale=True)),
}
VIS_FWD_BATCH = {"vis_square": 2, "vis_small": 6, "vis_two": 5}
VIS_CHAIN_CASES = {
    # name: (net, B, model kwargs, deterministic)
    "vmlp_ddim100_5": ("vmlp_small", 5, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                            randn_clip_value=3, min_sampling_denoising_std=0.1), False),
    "vunet_ddim100_5": ("vunet_small", 4, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                              randn_clip_value=3, min_sampling_denoising_std=0.1), False),
    "vunet_two_ddpm20_ft10": ("vunet_two", 3, dict(denoising_steps=20, ft_denoising_steps=10, randn_clip_value=3), False),
    # the shipped square image actor: hidden 768 (cfg/robomimic/finetune/square/ft_ppo_diffusion_mlp_img.yaml:134-140)
    "vmlp_square_ddim100_5": ("vmlp_square", 2, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                                    randn_clip_value=3, min_sampling_denoising_std=0.1), False),
}
VIS_LOSS_CASES = {
    # name: (net, N, model kwargs, reward_horizon)
    "vmlp_loss": ("vmlp_small", 24, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                        clip_ploss_coef=0.01, clip_ploss_coef_base=0.001, min_sampling_denoising_std=0.1), 4),
    "vunet_loss": ("vunet_small", 16, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                          clip_ploss_coef=0.01, clip_ploss_coef_base=0.001, min_sampling_denoising_std=0.1), 4),
    "vunet_two_loss": ("vunet_two", 12, dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01, clip_vloss_coef=0.2), 8),
    "vmlp_square_loss": ("vmlp_square", 6, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                               clip_ploss_coef=0.01, clip_ploss_coef_base=0.001, min_sampling_denoising_std=0.1), 4),
}
VIS_MSE_CASES = {"vmlp_mse": ("vmlp_small", 20, 16), "vunet_mse": ("vunet_small", 20, 12)}

# BASELINE configs[4] END TO END at its shipped shape (cfg/robomimic/finetune/square/ft_ppo_diffusion_unet_img.yaml:17-25,
# 97-139): one 96 x 96 camera, ViT (patch 8, depth 1, 128 wide, 4 heads) + SpatialEmb 128, VisionUnet1D dim 64, mults (1, 2),
# kernel 5, 8 groups, step embedding 32, Da 7, Ta 4, DDIM 100 -> 5 with EtaFixed, ViTCritic.  Its own fixture file
# (g21_vision_c5.npz: the entries above keep their files and random streams).
VIS_C5_NETS = {
    "vunet_square": ("vis_square", "unet", dict(action_dim=7, horizon_steps=4, diffusion_step_embed_dim=32, dim=64, dim_mults=(1, 2),
                                                kernel_size=5, n_groups=8, smaller_encoder=False, cond_predict_scale=True)),
}
VIS_C5_CHAIN_CASES = {
    "vunet_square_ddim100_5": ("vunet_square", 2, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                                      randn_clip_value=3, min_sampling_denoising_std=0.1,
                                                      min_logprob_denoising_std=0.1), False),
}
VIS_C5_LOSS_CASES = {
    "vunet_square_loss": ("vunet_square", 4, dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5,
                                                 clip_ploss_coef=0.01, clip_ploss_coef_base=0.001, clip_ploss_coef_rate=3,
                                                 min_sampling_denoising_std=0.1, min_logprob_denoising_std=0.1), 4),
}
VIS_ALL_NETS = dict(VIS_NETS, **VIS_C5_NETS)

# Gaussian policy on pixels (Gaussian_VisionMLP + ViTCritic): name -> (vis spec, trunk kwargs, model kwargs)
VIS_GAUSS_CASES = {
    # cfg/robomimic/finetune/can/ft_ppo_gaussian_mlp_img.yaml:98-124 at small maps: learned std from 0.1, residual 512 trunk
    "vgauss_small": ("vis_small", dict(mlp_dims=[512, 512, 512], activation="Mish", action_dim=3, horizon_steps=4),
                     dict(fixed_std=0.1, learn_fixed_std=True, std_min=0.01, std_max=0.2, clip_ploss_coef=0.01,
                          randn_clip_value=3)),
    "vgauss_two_fixed": ("vis_two", dict(mlp_dims=[256, 256, 256], activation="ReLU", action_dim=4, horizon_steps=4),
                         dict(fixed_std=0.08, learn_fixed_std=False, std_min=0.01, std_max=1.0, clip_ploss_coef=0.02,
                              clip_vloss_coef=0.2, randn_clip_value=3)),
}

# plain (non-residual) MLP trunks, model/common/mlp.py:27-81: name -> (spec, model kwargs)
PLAIN_CASES = {
    "plain_ddpm": ("plain_256", dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.01, clip_ploss_coef_base=0.001)),
    "plain_small_ddim": ("plain_mlp", dict(denoising_steps=100, ft_denoising_steps=5, use_ddim=True, ddim_steps=5, clip_ploss_coef=0.01,
                                           clip_vloss_coef=0.2)),
}

# mixture-of-Gaussians PPO: name -> (cond_dim, trunk kwargs, Ta, Da, GmmCfg kwargs, critic mlp_dims)
GMM_CASES = {
    # cfg/robomimic/finetune/can/ft_ppo_gmm_mlp.yaml:80-99: residual 512 trunks, 5 modes, Ta 4, Da 7 (140 mean outputs), learned std.
    # The reference clamps the SUMMED log-prob to [-5, 2] (gmm_ppo.py:62-63): at the shipped std 0.1 it sits near +25 and the policy
    # gradient is identically zero; the fixtures use stds that put most samples inside the window (0.27 here, 0.4 below) so that
    # the pass-through mask, the responsibilities and both trunks' gradients are exercised.
    "gmm_can": (23, dict(mlp_dims=[512, 512, 512], activation="Mish", residual=True), 4, 7,
                dict(num_modes=5, fixed_std=0.27, learn_fixed_std=True, std_min=0.01, std_max=0.5, clip_ploss_coef=0.01), [256, 256, 256]),
    # cfg/d3il/finetune/avoid_m1/ft_ppo_gmm_mlp.yaml:84-103: plain 256 x 2 trunks, fixed std, Ta 4, Da 2
    "gmm_d3il": (4, dict(mlp_dims=[256, 256], activation="ReLU", residual=False), 4, 2,
                 dict(num_modes=5, fixed_std=0.4, learn_fixed_std=False, std_min=0.01, std_max=1.0, clip_ploss_coef=0.1,
                      clip_vloss_coef=0.2), [256, 256, 256]),
}
