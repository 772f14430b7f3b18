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
