"""Type-name registries + build_from_cfg — the reference's plugin mechanism (python/jnerf/utils/registry.py:1-55), restated on torch."""


class Registry:
    def __init__(self):
        self._modules = {}

    def register_module(self, name=None, module=None):
        def _register(m):
            key = name if name is not None else m.__name__
            assert key not in self._modules, f"{key} is already registered."
            self._modules[key] = m
            return m
        return _register(module) if module is not None else _register

    def get(self, name):
        assert name in self._modules, f"{name} is not registered."
        return self._modules[name]


def build_from_cfg(cfg, registry, **kwargs):
    if isinstance(cfg, str):
        return registry.get(cfg)(**kwargs)
    if isinstance(cfg, dict):
        args = dict(cfg)
        args.update(kwargs)
        cls = registry.get(args.pop("type"))
        try:
            return cls(**args)
        except TypeError as e:
            raise TypeError(f"{cls}.{e}" if "<class" not in str(e) else str(e))
    if isinstance(cfg, list):
        import torch
        return torch.nn.Sequential(*[build_from_cfg(c, registry, **kwargs) for c in cfg])
    if cfg is None:
        return None
    raise TypeError(f"type {type(cfg)} not support")


DATASETS = Registry()
ENCODERS = Registry()
NETWORKS = Registry()
SAMPLERS = Registry()
LOSSES = Registry()
OPTIMS = Registry()
SCHEDULERS = Registry()
