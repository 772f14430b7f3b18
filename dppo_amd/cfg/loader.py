"""Minimal config loader for the YAML surface of the reference's fine-tuning cfgs.

hydra / omegaconf are not available in this environment, so this implements the subset those cfgs use
(reference script/run.py:22-24,45,85-87 and cfg/**/ft_ppo_diffusion_mlp.yaml): nested attribute access with
``.get``, ``${a.b}`` interpolation, the resolvers ``${eval:'...'}``, ``${oc.env:VAR[,default]}``, ``${now:fmt}``,
``${round_up:..}``/``${round_down:..}``, ``key=value`` command-line overrides, and ``_target_`` instantiation
(recursive, like ``hydra.utils.instantiate``).  ``_target_`` strings that name the reference package
(``dppo.…``) are mapped onto this package (``dppo_amd.…``) when a module of that name exists here.
"""
from __future__ import annotations

import datetime
import importlib
import math
import os
import re
from typing import Any

import yaml


class _Loader(yaml.SafeLoader):
    """SafeLoader that, like OmegaConf, reads ``1e-4`` / ``3E5`` (no dot) as floats."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:[0-9][0-9_]*)(?:\.[0-9_]*)?[eE][-+]?[0-9]+$"), list("-+0123456789"))


class Cfg(dict):
    """dict with attribute access; ``None`` for missing keys is NOT implied -- use .get like OmegaConf."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


_NOW = datetime.datetime.now()
_TOKEN = re.compile(r"\$\{([^${}]*)\}")


def _lookup(root, path):
    cur = root
    for part in path.split("."):
        m = re.fullmatch(r"([^\[\]]+)((?:\[\d+\])*)", part)  # OmegaConf's `key[i]` (shape_meta.obs.rgb.shape[1])
        key, idx = (m.group(1), re.findall(r"\[(\d+)\]", m.group(2))) if m else (part, [])
        cur = cur[int(key)] if isinstance(cur, list) else cur[key]
        for i in idx:
            cur = cur[int(i)]
    return cur


def _resolve_token(root, body: str):
    if body.startswith("eval:"):
        expr = body[5:].strip()
        if len(expr) >= 2 and expr[0] == expr[-1] and expr[0] in "'\"":
            expr = expr[1:-1]
        return eval(expr, {"__builtins__": {}}, {"math": math, "min": min, "max": max, "int": int, "float": float,
                                                 "round": round, "abs": abs})
    if body.startswith("oc.env:"):
        name, _, default = body[7:].partition(",")
        if name in os.environ:
            return os.environ[name]
        if default != "":
            return default
        raise KeyError(f"environment variable {name} is not set (needed by the config)")
    if body.startswith("now:"):
        return _NOW.strftime(body[4:])
    if body.startswith("round_up:"):
        return math.ceil(float(body[9:]))
    if body.startswith("round_down:"):
        return math.floor(float(body[11:]))
    return _lookup(root, body)


def _resolve_value(root, v, depth=0):
    if depth > 32:
        raise RecursionError("config interpolation does not terminate")
    if not isinstance(v, str) or "${" not in v:
        return v
    m = _TOKEN.fullmatch(v)
    if m:  # the whole value is one token: keep the type
        return _resolve_value(root, _resolve_token(root, _resolve_inner(root, m.group(1), depth)), depth + 1)
    out = _TOKEN.sub(lambda mm: str(_resolve_value(root, _resolve_token(root, mm.group(1)), depth + 1)), v)
    return _resolve_value(root, out, depth + 1)


def _resolve_inner(root, body, depth):
    return body if "${" not in body else _resolve_value(root, body, depth + 1)


def resolve(cfg: Cfg) -> Cfg:
    """Resolve every interpolation in place (OmegaConf.resolve, reference run.py:45)."""

    def walk(node):
        items = node.items() if isinstance(node, dict) else enumerate(node)
        for k, v in list(items):
            if isinstance(v, (dict, list)):
                walk(v)
            else:
                node[k] = _resolve_value(cfg, v)

    walk(cfg)
    return cfg


def load_config(path: str, overrides=()) -> Cfg:
    with open(path) as f:
        raw = yaml.load(f, Loader=_Loader)
    raw.pop("defaults", None)
    raw.pop("hydra", None)
    cfg = _wrap(raw)
    for ov in overrides:
        key, _, val = ov.partition("=")
        node = cfg
        parts = key.lstrip("+").split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, Cfg())
        node[parts[-1]] = yaml.load(val, Loader=_Loader)
    return resolve(cfg)


def get_class(target: str):
    """hydra.utils.get_class with the ``dppo.`` -> ``dppo_amd.`` package mapping."""
    mod, _, name = target.rpartition(".")
    candidates = [mod]
    if mod == "dppo" or mod.startswith("dppo."):
        candidates.insert(0, "dppo_amd" + mod[4:])
    err = None
    for m in candidates:
        try:
            return getattr(importlib.import_module(m), name)
        except (ImportError, AttributeError) as e:
            err = e
    raise ImportError(f"cannot locate {target}: {err}")


def instantiate(node: Any, **extra):
    """Recursive ``_target_`` instantiation (hydra.utils.instantiate, reference train_agent.py:84)."""
    if isinstance(node, dict) and "_target_" in node:
        kwargs = {k: instantiate(v) for k, v in node.items() if k != "_target_"}
        kwargs.update(extra)
        return get_class(node["_target_"])(**kwargs)
    if isinstance(node, dict):
        return Cfg({k: instantiate(v) for k, v in node.items()})
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    return node
