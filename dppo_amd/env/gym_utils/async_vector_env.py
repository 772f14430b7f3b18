"""Vectorised host environments on a pool of worker processes.

Replaces the reference's ``AsyncVectorEnv`` (``dppo/env/gym_utils/async_vector_env.py:56-840``, gym 0.22's class with
terminated / truncated, no auto-reset, ``reset_arg`` / ``reset_one_arg`` / ``render``) for the calls the agents make
(agent/finetune/train_agent.py:45-66,107-120; train_ppo_diffusion_agent.py:101-145): ``reset_arg``, ``reset_one_arg``,
``step`` (also split ``step_async`` / ``step_wait``), ``seed``, ``call`` / ``get_attr`` / ``set_attr``, ``render``, ``close``.

What is different, and why.  The reference starts ONE process PER ENVIRONMENT (it never ran more than 50); BASELINE's
configs ask for 512 envs per GPU and 4096 per node, which is a host-side scaling problem of its own (SURVEY.md section 7
(vii)).  Here a pool of ``n_workers`` processes (default: the CPU share of this process, at most one per env) each owns
a contiguous slice of the envs and steps them back to back per command, so the number of processes, pipes and context
switches follows the core count instead of the env count.  Observations, rewards and the two done flags are written by
the workers straight into POSIX shared memory laid out ``(n_envs, ...)`` per observation key -- the parent hands out
views (or copies, ``copy=True``) and the rollout loop copies one contiguous block per key into its pinned staging
buffer; pipes carry only the command, the action slice and the per-env ``info`` dicts.  Per-env semantics are the
reference's: ``env_fn()`` objects (normally ``MultiStep(Wrapper(simulator))``) with ``reset(**kwargs)``,
``step(action) -> (obs, reward, terminated, truncated, info)``, ``seed``; no auto-reset (``MultiStep`` resets within
the step).  Worker exceptions are re-raised in the parent with the worker's traceback.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import pickle
import traceback
from multiprocessing import shared_memory
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

__all__ = ["AsyncVectorEnv", "SyncVectorEnv"]


def _dumps(fn):
    """env_fn closures must cross a spawn boundary: cloudpickle when it is there (lambdas), plain pickle otherwise."""
    try:
        import cloudpickle
        return cloudpickle.dumps(fn)
    except ImportError:
        return pickle.dumps(fn)


def _as_dict(obs):
    return obs if isinstance(obs, dict) else {"state": obs}


class _Shm:
    """A named shared-memory block viewed as one numpy array."""

    def __init__(self, shape, dtype, name=None):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        nbytes = max(int(np.prod(self.shape)) * self.dtype.itemsize, 1)
        self.shm = shared_memory.SharedMemory(create=name is None, size=nbytes, name=name)
        self.owner = name is None
        self.array = np.ndarray(self.shape, dtype=self.dtype, buffer=self.shm.buf)

    def spec(self):
        return (self.shape, self.dtype.str, self.shm.name)

    def close(self):
        self.array = None
        self.shm.close()
        if self.owner:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass


def _worker(widx, lo, hi, env_fns_blob, pipe, specs):
    envs, blocks = [], {}
    try:
        envs = [fn() for fn in pickle.loads(env_fns_blob)]
        # the observation layout comes from one reset of this worker's first env (the reference's wrappers reset in
        # their constructors for the same purpose); the agent seeds and resets every env before it steps
        pipe.send(("ready", _spec_of(_as_dict(envs[0].reset()))))
        specs = pipe.recv()
        blocks = {k: _Shm(shape, dt, name) for k, (shape, dt, name) in specs.items()}
        obs_keys = [k for k in blocks if k.startswith("obs:")]

        def put_obs(i, obs):
            obs = _as_dict(obs)
            for k in obs_keys:
                blocks[k].array[lo + i] = obs[k[4:]]

        while True:
            cmd, data = pipe.recv()
            if cmd == "step":
                infos = []
                for i, env in enumerate(envs):
                    obs, rew, term, trunc, info = env.step(data[i])
                    put_obs(i, obs)
                    blocks["reward"].array[lo + i] = rew
                    blocks["terminated"].array[lo + i] = term
                    blocks["truncated"].array[lo + i] = trunc
                    infos.append(info)
                pipe.send((True, infos))
            elif cmd == "reset":  # data: {local index: kwargs}
                for i, kw in data.items():
                    put_obs(i, envs[i].reset(**kw))
                pipe.send((True, None))
            elif cmd == "seed":
                for env, s in zip(envs, data):
                    env.seed(s)
                pipe.send((True, None))
            elif cmd == "call":
                name, args, kwargs, which = data
                out = []
                for i in (range(len(envs)) if which is None else which):
                    f = getattr(envs[i], name)
                    out.append(f(*args, **kwargs) if callable(f) else f)
                pipe.send((True, out))
            elif cmd == "setattr":
                name, values = data
                for env, v in zip(envs, values):
                    setattr(env, name, v)
                pipe.send((True, None))
            elif cmd == "close":
                pipe.send((True, None))
                break
            else:
                raise RuntimeError(f"unknown command {cmd!r}")
    except (KeyboardInterrupt, Exception):
        try:
            pipe.send((False, f"worker {widx} (envs {lo}..{hi - 1}):\n{traceback.format_exc()}"))
        except (BrokenPipeError, OSError):
            pass
    finally:
        for env in envs:
            if hasattr(env, "close"):
                try:
                    env.close()
                except Exception:
                    pass
        for b in blocks.values():
            b.close()


def _spec_of(obs: Dict[str, np.ndarray]):
    return {k: (np.asarray(v).shape, np.asarray(v).dtype.str) for k, v in obs.items()}


def usable_cores() -> int:
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


class AsyncVectorEnv:
    def __init__(self, env_fns: Sequence[Callable], n_workers: Optional[int] = None, copy: bool = True,
                 context: Optional[str] = None, daemon: bool = True, **kwargs):
        self.env_fns = list(env_fns)
        self.num_envs = self.n_envs = len(self.env_fns)
        assert self.num_envs > 0
        n_workers = min(self.num_envs, n_workers or usable_cores())
        self.copy, self.closed, self._pending = copy, False, None
        ctx = mp.get_context(context or "spawn")  # spawn: the parent holds a GPU context that must not be forked
        bounds = np.linspace(0, self.num_envs, n_workers + 1).astype(int)
        self.slices = [(int(bounds[w]), int(bounds[w + 1])) for w in range(n_workers)]
        self.pipes, self.procs = [], []
        for w, (lo, hi) in enumerate(self.slices):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker, name=f"AsyncVectorEnv-worker-{w}", daemon=daemon,
                            args=(w, lo, hi, pickle.dumps([_dumps_wrap(f) for f in self.env_fns[lo:hi]]), child, None))
            p.start()
            child.close()
            self.pipes.append(parent)
            self.procs.append(p)
        # every worker builds its envs and reports the observation layout of its first one; the blocks are sized from
        # worker 0's report and their names sent back
        reports = [self._recv(pipe, raw=True) for pipe in self.pipes]
        spec = reports[0]
        for r in reports[1:]:
            if r != spec:
                raise RuntimeError(f"sub-environments disagree on the observation layout: {spec} vs {r}")
        self.blocks = {f"obs:{k}": _Shm((self.num_envs,) + tuple(shape), dt) for k, (shape, dt) in spec.items()}
        self.blocks["reward"] = _Shm((self.num_envs,), np.float64)
        self.blocks["terminated"] = _Shm((self.num_envs,), np.bool_)
        self.blocks["truncated"] = _Shm((self.num_envs,), np.bool_)
        specs = {k: b.spec() for k, b in self.blocks.items()}
        for pipe in self.pipes:
            pipe.send(specs)
        self.obs_keys = list(spec.keys())

    # ------------------------------------------------------------------ plumbing
    def _recv(self, pipe, raw=False):
        try:
            msg = pipe.recv()
        except EOFError as e:
            raise RuntimeError("an environment worker died") from e
        if raw:
            if msg[0] is False:
                self.close(terminate=True)
                raise RuntimeError(msg[1])
            return msg[1]
        ok, payload = msg
        if not ok:
            self.close(terminate=True)
            raise RuntimeError(payload)
        return payload

    def _all(self, cmd, per_worker):
        assert not self.closed, "environment is closed"
        assert self._pending is None, f"a {self._pending} call is pending"
        for pipe, data in zip(self.pipes, per_worker):
            pipe.send((cmd, data))
        return [self._recv(pipe) for pipe in self.pipes]

    def _obs(self):
        out = {k: self.blocks[f"obs:{k}"].array for k in self.obs_keys}
        return {k: v.copy() for k, v in out.items()} if self.copy else out

    # ------------------------------------------------------------------ the reference's surface
    def seed(self, seeds=None):
        if seeds is None:
            seeds = [None] * self.num_envs
        if isinstance(seeds, int):
            seeds = [seeds + i for i in range(self.num_envs)]
        assert len(seeds) == self.num_envs
        self._all("seed", [list(seeds[lo:hi]) for lo, hi in self.slices])

    def reset_arg(self, options_list=None):
        """Reset every env, env i with ``options=options_list[i]`` (reference :300-357)."""
        options_list = options_list if options_list is not None else [{} for _ in range(self.num_envs)]
        assert len(options_list) == self.num_envs
        self._all("reset", [{i - lo: {"options": options_list[i]} for i in range(lo, hi)} for lo, hi in self.slices])
        return self._obs()

    def reset(self, **kwargs):
        self._all("reset", [{i - lo: dict(kwargs) for i in range(lo, hi)} for lo, hi in self.slices])
        return self._obs()

    def reset_one_arg(self, env_ind, options=None):
        """Reset env ``env_ind`` only; returns ITS observation (reference :700-712)."""
        self._all("reset", [({env_ind - lo: {"options": options or {}}} if lo <= env_ind < hi else {})
                            for lo, hi in self.slices])
        return {k: self.blocks[f"obs:{k}"].array[env_ind].copy() for k in self.obs_keys}

    def step_async(self, actions):
        assert not self.closed and self._pending is None
        actions = np.asarray(actions)
        assert len(actions) == self.num_envs
        for pipe, (lo, hi) in zip(self.pipes, self.slices):
            pipe.send(("step", actions[lo:hi]))
        self._pending = "step"

    def step_wait(self):
        assert self._pending == "step", "step_wait without step_async"
        self._pending = None
        infos: List[dict] = []
        for pipe in self.pipes:
            infos.extend(self._recv(pipe))
        b = self.blocks
        return (self._obs(), b["reward"].array.copy(), b["terminated"].array.copy(), b["truncated"].array.copy(), infos)

    def step(self, actions):
        """actions (n_envs, ...) -> (obs dict of (n_envs, ...), reward, terminated, truncated, infos)."""
        self.step_async(actions)
        return self.step_wait()

    def call(self, name, *args, **kwargs):
        out = self._all("call", [(name, args, kwargs, None)] * len(self.pipes))
        return tuple(x for part in out for x in part)

    def call_sync(self, name, indices=None, **kwargs):
        """The reference's ``call_sync(name, indices=[...], **kwargs)`` (used by ``reset_one_arg`` there)."""
        indices = range(self.num_envs) if indices is None else indices
        per = [[i - lo for i in indices if lo <= i < hi] for lo, hi in self.slices]
        out = self._all("call", [(name, (), kwargs, w) for w in per])
        return [x for part in out for x in part]

    def get_attr(self, name):
        return self.call(name)

    def set_attr(self, name, values):
        if not isinstance(values, (list, tuple)):
            values = [values] * self.num_envs
        assert len(values) == self.num_envs
        self._all("setattr", [(name, list(values[lo:hi])) for lo, hi in self.slices])

    def render(self, *args, **kwargs):
        return self.call("render", *args, **kwargs)

    def close(self, terminate=False):
        if self.closed:
            return
        self.closed = True
        for pipe, p in zip(self.pipes, self.procs):
            try:
                if not terminate and p.is_alive():
                    pipe.send(("close", None))
                    pipe.recv()
            except (BrokenPipeError, EOFError, OSError):
                pass
        for p in self.procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()  # the exact process this object started
        for pipe in self.pipes:
            pipe.close()
        for b in getattr(self, "blocks", {}).values():
            b.close()

    def __del__(self):
        try:
            self.close(terminate=True)
        except Exception:
            pass


class _Thunk:
    """Picklable holder of an env_fn serialised with cloudpickle (so lambdas / closures cross the spawn boundary)."""

    def __init__(self, fn):
        self.blob = _dumps(fn)

    def __call__(self):
        try:
            import cloudpickle  # noqa: F401  (registers nothing; loads is pickle-compatible)
        except ImportError:
            pass
        return pickle.loads(self.blob)()


def _dumps_wrap(fn):
    return _Thunk(fn)


class SyncVectorEnv:
    """The same surface in the calling process (``asynchronous=False`` of the reference's ``make_async``; tests)."""

    def __init__(self, env_fns: Sequence[Callable], **kwargs):
        self.envs = [fn() for fn in env_fns]
        self.num_envs = self.n_envs = len(self.envs)

    @staticmethod
    def _stack(obs_list):
        obs_list = [_as_dict(o) for o in obs_list]
        return {k: np.stack([np.asarray(o[k]) for o in obs_list]) for k in obs_list[0]}

    def seed(self, seeds=None):
        if seeds is None:
            seeds = [None] * self.num_envs
        if isinstance(seeds, int):
            seeds = [seeds + i for i in range(self.num_envs)]
        for e, s in zip(self.envs, seeds):
            e.seed(s)

    def reset_arg(self, options_list=None):
        options_list = options_list if options_list is not None else [{} for _ in range(self.num_envs)]
        self._last = [e.reset(options=o) for e, o in zip(self.envs, options_list)]
        return self._stack(self._last)

    def reset(self, **kwargs):
        self._last = [e.reset(**kwargs) for e in self.envs]
        return self._stack(self._last)

    def reset_one_arg(self, env_ind, options=None):
        self._last[env_ind] = self.envs[env_ind].reset(options=options or {})
        return {k: np.asarray(v) for k, v in _as_dict(self._last[env_ind]).items()}

    def step(self, actions):
        res = [e.step(a) for e, a in zip(self.envs, np.asarray(actions))]
        self._last = [r[0] for r in res]
        return (self._stack(self._last), np.array([r[1] for r in res], dtype=np.float64),
                np.array([r[2] for r in res], dtype=np.bool_), np.array([r[3] for r in res], dtype=np.bool_),
                [r[4] for r in res])

    def call(self, name, *args, **kwargs):
        out = []
        for e in self.envs:
            f = getattr(e, name)
            out.append(f(*args, **kwargs) if callable(f) else f)
        return tuple(out)

    def get_attr(self, name):
        return self.call(name)

    def set_attr(self, name, values):
        if not isinstance(values, (list, tuple)):
            values = [values] * self.num_envs
        for e, v in zip(self.envs, values):
            setattr(e, name, v)

    def render(self, *args, **kwargs):
        return self.call("render", *args, **kwargs)

    def close(self):
        for e in self.envs:
            if hasattr(e, "close"):
                e.close()
