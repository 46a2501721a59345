"""Config / checkpoint helpers: host-side mirror of ``flocoder/general.py`` plus a small YAML composer for the reference's
Hydra config tree (hydra / omegaconf are not required; ``configs/*.yaml`` with ``# @package _global_`` and a ``defaults:``
list are composed the way Hydra composes them for this repo's layout).  Boundary code only -- no arithmetic here.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path
from typing import Any, Iterable, List, Optional

import torch
import yaml


class Config(dict):
    """dict with attribute access and ``.get`` -- what the reference touches on an OmegaConf node
    (``config.codec.choice``, ``config.flow.get('lambda_lowres', 0.1)``, ``hasattr(config, 'vqgan_checkpoint')``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def to_container(self, resolve: bool = True) -> dict:
        return _unwrap(self)


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, Config):
        return Config({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def _unwrap(v):
    if isinstance(v, dict):
        return {k: _unwrap(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_unwrap(x) for x in v]
    return v


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _set_dotted(cfg: dict, key: str, value: Any) -> None:
    node = cfg
    parts = key.split(".")
    for p in parts[:-1]:
        if not isinstance(node.get(p), dict):
            node[p] = {}
        node = node[p]
    node[parts[-1]] = value


def load_config(path: str, overrides: Iterable[str] = ()) -> Config:
    """Compose ``path`` (a config of the reference's tree, e.g. ``configs/flowers_sd.yaml``): entries of ``defaults:`` are
    files relative to the config's directory merged in order, ``_self_`` is the file's own body, later wins; every file is
    ``# @package _global_`` so bodies merge at the root.  ``overrides`` are Hydra-style ``a.b=c`` / ``+a.b=c`` strings."""
    path = os.path.expanduser(path)
    base = os.path.dirname(os.path.abspath(path))

    def read(p):
        with open(p) as f:
            return yaml.safe_load(f) or {}

    body = read(path)
    defaults: List[Any] = body.pop("defaults", ["_self_"])
    if "_self_" not in defaults:
        defaults = list(defaults) + ["_self_"]
    out: dict = {}
    for d in defaults:
        if d == "_self_":
            _merge(out, body)
        else:
            sub = read(os.path.join(base, str(d) + ".yaml"))
            sub.pop("defaults", None)
            _merge(out, sub)
    for ov in overrides:
        k, _, v = ov.lstrip("+").partition("=")
        _set_dotted(out, k, yaml.safe_load(v))
    return _wrap(out)


def key_usable(d, key):
    """general.py:18-20."""
    return (d is not None) and isinstance(d, dict) and (d.get(key) is not None)


def handle_config_path():
    """general.py:23-47 -- accept ``--config-name path/to/x.yaml`` (with or without '=') by rewriting sys.argv into
    ``--config-path=dir --config-name=x``."""
    for i, arg in enumerate(sys.argv):
        if arg == '--config-name' and i + 1 < len(sys.argv):
            sys.argv[i] = f"--config-name={sys.argv[i + 1]}"
            sys.argv.pop(i + 1)
            break
    for i, arg in enumerate(sys.argv):
        path = arg.split('=', 1)[1] if arg.startswith('--config-name=') else None
        if path and '/' in path and path.endswith('.yaml') and os.path.exists(os.path.expanduser(path)):
            full = os.path.expanduser(path)
            sys.argv[i] = f"--config-name={os.path.basename(full).replace('.yaml', '')}"
            sys.argv.insert(i, f"--config-path={os.path.dirname(full)}")
            break


def config_from_argv(argv: Optional[List[str]] = None) -> Config:
    """What ``@hydra.main`` does for the reference's scripts, after handle_config_path(): resolve --config-path /
    --config-name and treat the remaining ``k=v`` arguments as overrides."""
    argv = list(sys.argv[1:] if argv is None else argv)
    cdir, cname, ov = "configs", None, []
    for a in argv:
        if a.startswith("--config-path="):
            cdir = a.split("=", 1)[1]
        elif a.startswith("--config-name="):
            cname = a.split("=", 1)[1]
        elif "=" in a and not a.startswith("--"):
            ov.append(a)
    if cname is None:
        raise ValueError("--config-name is mandatory (the reference's default 'flowers' does not exist, SURVEY Q11)")
    return load_config(os.path.join(cdir, cname if cname.endswith(".yaml") else cname + ".yaml"), ov)


def ldcfg(config, key, default=None, supply_defaults=False, debug=False, verbose=True):
    """general.py:50-74 -- look ``key`` up in flow, then preencoding, then codec, then the top level; a miss returns None
    unless ``supply_defaults`` (that precedence is part of the behaviour)."""
    assert config is not None, 'ldcfg: config is None, and needs to be not-None'
    cfg = config.to_container(resolve=True) if hasattr(config, 'to_container') else config
    if 'flow' in cfg and cfg['flow'] is not None and key in cfg['flow']:
        answer = cfg['flow'][key]
    elif 'preencoding' in cfg and key in cfg['preencoding']:
        answer = cfg['preencoding'][key]
    elif 'codec' in cfg and key in cfg['codec']:
        answer = cfg['codec'][key]
    elif key in cfg:
        answer = cfg[key]
    else:
        if verbose:
            print(f"ldcfg: Warning: couldn't find key '{key}' in config keys: {list(cfg.keys())}")
        answer = default if supply_defaults else None
    if verbose:
        print(f'lcfg: {key} := {answer}')
    return answer


def keep_recent_files(keep=5, directory='checkpoints', pattern='*.pt'):
    """Disk-fill guard of the checkpoint directory (general.py:77-81): of the files matching ``pattern`` only the ``keep`` most
    recently modified survive."""
    by_age = sorted(Path(directory).glob(pattern), key=os.path.getmtime)
    for stale in by_age[:max(0, len(by_age) - keep)]:
        stale.unlink()


def save_checkpoint(model, epoch=None, optimizer=None, keep=5, prefix="vqgan", ckpt_dir='checkpoints', config=None):
    """Write ``{ckpt_dir}/{prefix}[_{epoch}].pt`` after pruning older ``{prefix}*.pt`` files (general.py:120-137).  The file is the dict
    the reference's loaders expect (generate_samples.py:76-108): ``model_state_dict`` always, ``epoch`` / ``optimizer_state_dict`` /
    ``config`` when given.  Returns the path written."""
    keep_recent_files(keep=keep, directory=ckpt_dir, pattern=f'{prefix}*.pt')
    payload = {'model_state_dict': model.state_dict()}
    name = prefix
    if epoch is not None:
        payload['epoch'] = epoch
        name = f'{prefix}_{epoch}'
    if optimizer is not None:
        payload['optimizer_state_dict'] = optimizer.state_dict()
    if config is not None:
        payload['config'] = _unwrap(config) if isinstance(config, dict) else config
    os.makedirs(ckpt_dir, exist_ok=True)
    target = f'{ckpt_dir}/{name}.pt'
    torch.save(payload, target)
    print(f"Checkpoint saved to {target}")
    return target


def load_flow_model(vmodel_path: str, config, device, n_classes: Optional[int] = None):
    """Checkpoint -> Unet, the way generate_samples.load_models_once does (generate_samples.py:76-108): width and channels
    come from ``init_conv.weight``, ``dim_mults`` from the config, ``strict=False``."""
    from .unet import Unet
    ckpt = torch.load(vmodel_path, map_location="cpu", weights_only=False)
    sd = ckpt['model_state_dict']
    w = sd['init_conv.weight']
    if n_classes is None:
        n_classes = sd['class_cond_mlp.0.weight'].shape[0] if 'class_cond_mlp.0.weight' in sd else 0
    model = Unet(dim=w.shape[0], channels=w.shape[1], dim_mults=tuple(ldcfg(config, 'dim_mults', verbose=False) or (1, 2, 4, 8)),
                 n_classes=n_classes, mask_cond='mask_fusion_conv.0.weight' in sd)
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    return model.eval().to(device)
