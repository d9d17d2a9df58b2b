"""Minimal stand-in for `mmengine.Config.fromfile` (run.py:335): python config files with `_base_` inheritance and
attribute-dict access (`cfg.data.inverse_y`, `getattr(cfg.pnp, 'use_identical', False)`, `cfg_train.keys()`).
mmengine is not a dependency of the hot path and is absent offline."""
import os


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def _merge(base, over):
    """recursive dict merge; a dict carrying `_delete_=True` replaces instead of merging (mmengine semantics)."""
    out = dict(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = {kk: vv for kk, vv in v.items() if kk != '_delete_'} if isinstance(v, dict) else v
    return out


def _load(path):
    path = os.path.abspath(path)
    ns = {}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v) and not isinstance(v, type(os))}
    bases = cfg.pop('_base_', None)
    if bases:
        if isinstance(bases, str):
            bases = [bases]
        merged = {}
        for b in bases:
            merged = _merge(merged, _load(os.path.join(os.path.dirname(path), b)))
        cfg = _merge(merged, cfg)
    return cfg


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        return Config(_wrap(_load(path)))
