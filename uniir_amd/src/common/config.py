"""YAML config loader with ${a.b.c} interpolation and attribute access: the subset of OmegaConf that UniIR's entry
points use (OmegaConf.load / attribute access / assignment / to_yaml; e.g. inbatch.yaml:3-7 uses "${experiment.exp_name}").
omegaconf itself is not installed in the MI355X image, so this small loader keeps the YAML surface drop-in."""
import re

import yaml

_PAT = re.compile(r"\$\{([^}]+)\}")
_SCI = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")   # "1e-5": a float for OmegaConf, a string for YAML 1.1


class Config(dict):
    """dict with attribute access; nested dicts are wrapped on the way in"""

    def __init__(self, data=None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, Config):
            return Config(v)
        if isinstance(v, list):
            return [Config._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        def un(v):
            if isinstance(v, Config):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, list):
                return [un(x) for x in v]
            return v
        return un(self)


def _lookup(root, dotted):
    cur = root
    for part in dotted.strip().split("."):
        cur = cur[part]
    return cur


def _resolve(root, node, depth=0):
    if depth > 20:
        raise ValueError("config interpolation too deep (cycle?)")
    if isinstance(node, dict):
        for k in list(node.keys()):
            node[k] = _resolve(root, node[k], depth)
        return node
    if isinstance(node, list):
        return [_resolve(root, x, depth) for x in node]
    if isinstance(node, str) and "${" in node:
        m = _PAT.fullmatch(node)
        if m:  # whole value is one reference: keep the referenced type
            return _resolve(root, _lookup(root, m.group(1)), depth + 1)
        return _resolve(root, _PAT.sub(lambda mm: str(_lookup(root, mm.group(1))), node), depth + 1)
    if isinstance(node, str) and _SCI.match(node.strip()):
        return float(node)
    return node


def load_config(path):
    with open(path, "r") as f:
        raw = yaml.safe_load(f) or {}
    return Config(_resolve(raw, raw))


class OmegaConf:
    """name-compatible facade for the three calls the reference scripts make"""

    @staticmethod
    def load(path):
        return load_config(path)

    @staticmethod
    def create(d=None):
        return Config(d or {})

    @staticmethod
    def to_yaml(cfg, sort_keys=False):
        return yaml.safe_dump(cfg.to_dict() if isinstance(cfg, Config) else cfg, sort_keys=sort_keys)
