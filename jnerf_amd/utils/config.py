"""Global Config singleton with the reference's semantics (python/jnerf/utils/config.py:16-151): python/yaml config files,
`_base_` inheritance, `_cover_` override, attribute access where a MISSING key reads as None, auto `name` / `work_dir`,
and live objects stuffed into the same object (cfg.dataset_obj / model_obj / sampler_obj / m_training_step)."""
import copy
import importlib.util
import inspect
import os
from collections import OrderedDict

BASE_KEY, COVER_KEY = "_base_", "_cover_"


class Config(OrderedDict):
    def __init__(self, *args):
        super().__init__()
        if len(args) == 1:
            self.load_from_file(args[0])
        else:
            assert len(args) == 0

    def __getattr__(self, name):
        return self[name] if name in self else None

    def __setattr__(self, name, value):
        self[name] = value

    @staticmethod
    def _load_no_base(filename):
        assert os.path.isfile(filename), f"{filename} does not exist"
        if filename.endswith(".yaml"):
            import yaml
            with open(filename) as f:
                return yaml.safe_load(f.read())
        assert filename.endswith(".py"), "unsupported config type."
        spec = importlib.util.spec_from_file_location("_jnerf_cfg_" + os.path.basename(filename)[:-3], filename)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return {k: v for k, v in mod.__dict__.items() if not k.startswith("__")}

    @staticmethod
    def _load(filename):
        cfg = Config._load_no_base(filename)
        if BASE_KEY in cfg:
            bases = cfg.pop(BASE_KEY)
            bases = bases if isinstance(bases, list) else [bases]
            merged = {}
            for b in bases:
                Config.merge_dict_b2a(merged, Config._load(os.path.join(os.path.dirname(filename), b)))
            Config.merge_dict_b2a(merged, cfg)
            cfg = merged
        return cfg

    @staticmethod
    def merge_dict_b2a(a, b):
        def clear(x):
            if not isinstance(x, dict):
                return x
            out = copy.deepcopy(x)
            out.pop(COVER_KEY, None)
            return {k: clear(v) for k, v in out.items()}
        if COVER_KEY in b:
            a.clear()
            a.update(clear(copy.deepcopy(b)))
            return
        for k, v in b.items():
            if k not in a or (isinstance(v, dict) and v.get(COVER_KEY, False)) or not isinstance(v, dict) or not isinstance(a[k], dict):
                a[k] = clear(copy.deepcopy(v))
            else:
                Config.merge_dict_b2a(a[k], v)

    def load_from_file(self, filename):
        cfg = Config._load(filename)
        self.clear()
        self.update(self.dfs(cfg))
        if self.name is None:
            self.name = os.path.splitext(os.path.basename(filename))[0]
        if self.work_dir is None:
            self.work_dir = f"work_dirs/{self.name}"

    def dfs(self, other):
        if isinstance(other, dict):
            now = Config()
            for k, d in other.items():
                if not inspect.ismodule(d):
                    now[k] = self.dfs(d)
            return now
        if isinstance(other, list):
            return [self.dfs(d) for d in other if not inspect.ismodule(d)]
        return copy.deepcopy(other)

    def dump(self):
        now = {}
        for k, d in self.items():
            if isinstance(d, Config):
                d = d.dump()
            if isinstance(d, list):
                d = [x.dump() if isinstance(x, Config) else x for x in d]
            now[k] = d
        return now


_cfg = Config()


def init_cfg(filename):
    import sys
    print("Loading config from: ", filename, file=sys.stderr)
    _cfg.load_from_file(filename)


def get_cfg():
    return _cfg


def update_cfg(**kwargs):
    _cfg.update(kwargs)


def reset_cfg(**kwargs):
    """(ours) start from an empty config — tests and bench.py build configs in code."""
    _cfg.clear()
    _cfg.update(_cfg.dfs(kwargs))
    return _cfg


def save_cfg(save_file):
    import yaml
    with open(save_file, "w") as f:
        f.write(yaml.dump({k: v for k, v in _cfg.dump().items() if isinstance(v, (int, float, str, list, dict, bool, type(None)))}))
