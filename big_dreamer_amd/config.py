"""Config loading with the reference CLI's surface (`python src/main.py key=value ...`, Hydra-style dotted
overrides such as ``ActorCritic.entropy_weight=1e-4``) without Hydra/OmegaConf, which are not installed."""
from __future__ import annotations

import os
from typing import Any, Dict, List

import yaml

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf", "config.yaml")


def _coerce(x: Any) -> Any:
    """PyYAML (YAML 1.1) reads '2e-4' as a string; OmegaConf reads a float.  Match OmegaConf."""
    if isinstance(x, dict):
        return {k: _coerce(v) for k, v in x.items()}
    if isinstance(x, str):
        try:
            return float(x)
        except ValueError:
            return x
    return x


def _parse_value(text: str) -> Any:
    return _coerce(yaml.safe_load(text))


def load_config(overrides: List[str] = (), path: str = DEFAULT_CONFIG) -> Dict[str, Any]:
    with open(path) as f:
        cfg = _coerce(yaml.safe_load(f))
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not of the form key=value")
        key, val = ov.split("=", 1)
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            if p not in node or not isinstance(node[p], dict):
                raise KeyError(f"unknown config group '{p}' in override '{ov}'")
            node = node[p]
        if parts[-1] not in node:
            raise KeyError(f"unknown config key '{key}' (the reference's Hydra config rejects unknown keys too)")
        node[parts[-1]] = _parse_value(val)
    return cfg
