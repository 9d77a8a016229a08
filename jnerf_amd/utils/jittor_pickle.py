"""Reader / writer of the container `jt.save` / `jt.load` use for `.pkl` files - WITHOUT Jittor (the reference's Runner checkpoints,
python/jnerf/runner/runner.py:123-151).

Jittor is an un-vendored dependency of the reference (`jittor>=1.3.5.25`, setup.py:22), so the format is restated here from Jittor 1.3.x's published source
(`python/jittor/__init__.py`: `save`, `safepickle`, `safeunpickle`), not from anything under /root/reference - parity UNPINNED: no Jittor install exists in this image to
produce a real file.  What that source does:
  * `jt.save(obj, path)` first walks `obj` (dicts and lists, in place) and replaces every `jittor.Var` by `var.numpy()`; the reference's load code confirms the payload is
    numpy - it wraps the loaded values in `jt.array(...)` again (runner.py:139-146);
  * `safepickle`: `s = pickle.dumps(obj, 4)`, then the file is `s + sha1(s).digest() (20 bytes) + b"HCAJSLHD"`;
  * `safeunpickle`: a file that ends in the 8-byte magic has its checksum verified and stripped before `pickle.loads`; a file without the magic is unpickled as is.
So a checkpoint written by the reference is an ordinary pickle of nested dicts / lists of numpy arrays and Python scalars plus a 28-byte trailer, and can be read with the
standard library; `dump` below writes the same container (tensors become numpy arrays), which `jt.load` accepts."""
import hashlib
import io
import pickle

MAGIC = b"HCAJSLHD"


class _NumpyOnlyUnpickler(pickle.Unpickler):
    """the payload is numpy arrays and builtins: refuse to import anything else (a checkpoint is data, not code)"""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("collections", "OrderedDict"), ("builtins", "slice"), ("builtins", "set"),
                ("builtins", "frozenset"), ("builtins", "complex"), ("builtins", "bytearray")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to load {module}.{name} from a checkpoint (only numpy arrays and builtins are expected in a jt.save file)")


def loads(data, path="<bytes>"):
    if data.endswith(MAGIC):
        checksum, body = data[-28:-8], data[:-28]
        if hashlib.sha1(body).digest() != checksum:
            raise ValueError(f"{path}: pickle checksum does not match (the file is corrupted or truncated)")
        data = body
    return _NumpyOnlyUnpickler(io.BytesIO(data)).load()


def load(path):
    with open(path, "rb") as f:
        return loads(f.read(), path)


def to_numpy(obj):
    """what jt.save's walk does to jittor.Var, done to torch tensors (detached, on the host; everything else is kept)"""
    import numpy as np
    try:
        import torch
    except ImportError:      # pragma: no cover
        torch = None
    if isinstance(obj, dict):
        return {k: to_numpy(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_numpy(v) for v in obj]
    if torch is not None and torch.is_tensor(obj):
        return obj.detach().cpu().numpy()
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def dumps(obj):
    s = pickle.dumps(to_numpy(obj), 4)
    return s + hashlib.sha1(s).digest() + MAGIC


def dump(obj, path):
    with open(path, "wb") as f:
        f.write(dumps(obj))


def to_torch(obj, half_to_float=True):
    """numpy arrays -> torch tensors (fp16 payloads - the reference's fp16 parameters - become fp32 masters), recursively"""
    import numpy as np
    import torch
    if isinstance(obj, dict):
        return {k: to_torch(v, half_to_float) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_torch(v, half_to_float) for v in obj]
    if isinstance(obj, np.ndarray):
        a = obj.astype(np.float32) if (half_to_float and obj.dtype == np.float16) else obj
        return torch.from_numpy(np.ascontiguousarray(a))
    return obj
