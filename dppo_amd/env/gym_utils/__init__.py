"""Factory of the vectorised host environments (reference ``dppo/env/gym_utils/__init__.py:13-230`` ``make_async``).

Builds ``num_envs`` copies of ``wrappers(simulator)`` -- wrapper names and order as in the reference's cfgs
(``env.wrappers: {mujoco_locomotion_lowdim: {...}, multi_step: {...}}``) -- on the worker pool of
``dppo_amd.env.gym_utils.async_vector_env``.  The simulator itself comes from ``gym.make(id)`` when gym and the MuJoCo
stack are installed (they are not in this image), or from ``env_fn`` (any zero-argument callable returning an object with
the gym 0.22 ``reset`` / ``step`` API): that is how the tests and a user with a different simulator plug in.
"""
from typing import Callable, Optional

from dppo_amd.env.gym_utils.async_vector_env import AsyncVectorEnv, SyncVectorEnv
from dppo_amd.env.gym_utils.wrapper import wrapper_dict


class _EnvFactory:
    """Picklable ``env_fn``: simulator, then the cfg's wrappers in cfg order."""

    def __init__(self, id, env_fn, wrappers, max_episode_steps):
        self.id, self.env_fn, self.wrappers, self.max_episode_steps = id, env_fn, wrappers, max_episode_steps

    def __call__(self):
        if self.env_fn is not None:
            env = self.env_fn()
        else:
            try:
                import gym
            except ImportError as e:
                raise ImportError(f"environment {self.id!r}: gym is not installed; pass env_fn=... to make_async") from e
            env = gym.make(self.id)
        for name, args in (self.wrappers or {}).items():
            if name not in wrapper_dict:
                raise NotImplementedError(f"wrapper {name!r} is not built (have: {sorted(wrapper_dict)})")
            env = wrapper_dict[name](env, **dict(args))
        return env


def make_async(id, num_envs=1, asynchronous=True, wrappers=None, env_fn: Optional[Callable] = None, env_type=None,
               max_episode_steps=None, n_workers=None, **kwargs):
    if env_type in ("furniture", "robomimic") and env_fn is None:
        raise NotImplementedError(f"env_type={env_type!r} needs its simulator stack; pass env_fn=...")
    fns = [_EnvFactory(id, env_fn, wrappers, max_episode_steps) for _ in range(num_envs)]
    return AsyncVectorEnv(fns, n_workers=n_workers) if asynchronous else SyncVectorEnv(fns)
