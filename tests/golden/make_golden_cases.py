"""Case tables shared by make_golden.py (generator, needs the reference) and the tests (no reference needed)."""
SCHED_CASES = {
    # name: (kwargs, n_steps, optimizer lr)
    "hopper_actor": (dict(first_cycle_steps=1000, cycle_mult=1.0, max_lr=1e-4, min_lr=1e-4, warmup_steps=10, gamma=1.0), 15, 1e-4),
    "warm5_gamma": (dict(first_cycle_steps=20, cycle_mult=1.0, max_lr=1e-3, min_lr=1e-5, warmup_steps=5, gamma=0.5), 50, 1e-3),
    "nowarm_mult2": (dict(first_cycle_steps=10, cycle_mult=2.0, max_lr=3e-4, min_lr=1e-6, warmup_steps=0, gamma=0.9), 45, 3e-4),
    "pretrain_like": (dict(first_cycle_steps=200, cycle_mult=1.0, max_lr=1e-3, min_lr=1e-4, warmup_steps=1, gamma=1.0), 30, 1e-3),
}
