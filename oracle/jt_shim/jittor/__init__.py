"""TEST INFRASTRUCTURE ONLY - a torch-backed stand-in for the handful of Jittor calls the reference's PURE-PYTHON modules make, so that those modules (NeuS networks and
renderer, FrequencyEncoder, EMA, ExpDecay, HuberLoss, camera_path ...) can be EXECUTED in the build container, where Jittor is not installed and cannot be, and their
outputs committed as golden fixtures (tests/golden/make_golden_pyref.py).  Nothing of the product imports this package; nothing under `-m gpu`, smoke() or bench.py does.

What this pins and what it does not: the fixtures come out of the reference's own Python source - its control flow, formulas, index arithmetic, argument order - run
line by line; the tensor primitives underneath are torch's, with Jittor's semantics restated HERE where the two libraries differ (each such place is marked "jittor:"
below and follows Jittor's published source, jittor >= 1.3.5 as pinned by the reference's setup.py:22; Jittor itself is an un-vendored dependency).  So a transcription
error in jnerf_amd's restatement of those modules shows up as a mismatch; a misreading of a Jittor primitive would be shared by both sides and does not."""
import math
import numpy as np
import torch

float32, float16, int32, int64 = torch.float32, torch.float16, torch.int32, torch.int64
flags = type("flags", (), {"use_cuda": 0, "cuda_archs": [80]})()


class _VarMeta(type):
    def __instancecheck__(cls, obj):
        return isinstance(obj, torch.Tensor)


class Var(metaclass=_VarMeta):
    """jt.Var(data) builds a float32 variable out of a Python number / list; isinstance(x, jt.Var) is true for every tensor"""
    def __new__(cls, data, dtype=None):
        if isinstance(data, torch.Tensor):
            return data
        arr = np.asarray(data)
        if dtype is None:
            dtype = torch.float32 if arr.dtype.kind == "f" else (torch.int32 if arr.dtype.kind in "iu" else torch.bool)
        return torch.tensor(arr, dtype=dtype)


def array(data, dtype=None):
    return Var(data, dtype)


# ---- methods Jittor's Var has and torch.Tensor lacks (patched onto torch.Tensor in THIS process only: the fixture generator)
def _safe_clip(self, lo, hi):
    """jittor: Var.safe_clip - the value is clamped, the gradient passes through unchanged (jittor/__init__.py: `return self.maximum(lo).minimum(hi)` under a
    stop-gradient correction: x + (clip(x) - x).stop_grad())"""
    return self + (self.clamp(lo, hi) - self).detach()


def _update(self, other):
    with torch.no_grad():
        self.copy_(other)
    return self


torch.Tensor.safe_clip = _safe_clip
torch.Tensor.float16 = lambda self: self.half()
torch.Tensor.float32 = lambda self: self.float()
torch.Tensor.int32 = lambda self: self.to(torch.int32)
torch.Tensor.copy = lambda self: self.detach().clone()
torch.Tensor.update = _update
torch.Tensor.stop_grad = lambda self: self.detach()
torch.Tensor.sync = lambda self: self
torch.Tensor.assign = lambda self, other: self.set_(other)            # jittor: Var.assign rebinds the variable's data (DensityGridSampler.enlarge)
# jittor: binary operators promote mixed dtypes (camera_path.py multiplies an integer matrix into a float one)
_mm = torch.Tensor.__matmul__
torch.Tensor.__matmul__ = lambda a, b: _mm(*(t.to(torch.promote_types(a.dtype, b.dtype)) for t in (a, b)))
# jittor: Var.transpose(*axes) / fuse_transpose(axes) permute
_tr = torch.Tensor.transpose
torch.Tensor.transpose = lambda self, *a: self.permute(*a) if len(a) > 2 else _tr(self, *a)
torch.Tensor.fuse_transpose = lambda self, axes: self.permute(*axes)
# jittor: a Python list operand becomes a Var (dataset.py adds the offset list to a pose column)
_add = torch.Tensor.__add__
torch.Tensor.__add__ = lambda a, b: _add(a, torch.tensor(b, dtype=a.dtype) if isinstance(b, (list, tuple)) else b)
_expand = torch.Tensor.expand


def _jt_expand(self, *shape):
    """jittor: Var.expand broadcasts BOTH ways - a target extent of 1 (or -1) keeps the variable's own extent (neus_dataset.py expands a [3,3] matrix to (bs,1,1))"""
    shape = list(shape[0]) if len(shape) == 1 and isinstance(shape[0], (list, tuple, torch.Size)) else list(shape)
    off = len(shape) - self.dim()
    for i in range(self.dim()):
        if shape[off + i] in (1, -1):
            shape[off + i] = self.shape[i]
    return _expand(self, *shape)


torch.Tensor.expand = _jt_expand


class _Shape(tuple):
    """jittor: Var.shape is a NanoVector - `[n] + x.shape[1:]` is a list (runner.py:218 pads the last ray chunk that way); a torch.Size refuses"""
    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return _Shape(r) if isinstance(i, slice) else r

    def __radd__(self, other):
        return list(other) + list(self)

    def __add__(self, other):
        return _Shape(tuple(self) + tuple(other))

    def numel(self):
        return int(np.prod(self)) if len(self) else 1


def enable_jittor_shapes():
    """(opt-in: replaces the Python-level Tensor.shape attribute for this process)"""
    get = torch.Tensor.shape.__get__
    torch.Tensor.shape = property(lambda self: _Shape(get(self)))
_orig_numpy = torch.Tensor.numpy
torch.Tensor.numpy = lambda self, *a, **k: _orig_numpy(self.detach(), *a, **k)


def _kd(kw):
    """Jittor spells it `keepdims`, accepts `keepdim` too"""
    return bool(kw.get("keepdims", kw.get("keepdim", False)))


def concat(arr, dim=0):
    return torch.cat(list(arr), dim)


def stack(arr, dim=0):
    return torch.stack(list(arr), dim)


def linspace(start, end, steps):
    return torch.linspace(float(start), float(end), int(steps))


def arange(*a):
    return torch.arange(*a)


_DTYPES = {"float32": torch.float32, "float": torch.float32, "float16": torch.float16, "int32": torch.int32, "int": torch.int32, "int64": torch.int64, "uint8": torch.uint8,
           "bool": torch.bool}


def _shape(shape):
    return [int(shape)] if isinstance(shape, (int, float)) else [int(v) for v in shape]


def _dtype(dtype):
    return _DTYPES[dtype] if isinstance(dtype, str) else dtype


def ones(shape, dtype=torch.float32):
    return torch.ones(_shape(shape), dtype=_dtype(dtype))


def zeros(shape, dtype=torch.float32):
    return torch.zeros(_shape(shape), dtype=_dtype(dtype))


ones_like, zeros_like = torch.ones_like, torch.zeros_like
sigmoid, exp, log, sin, cos, abs, sqrt, matmul = torch.sigmoid, torch.exp, torch.log, torch.sin, torch.cos, torch.abs, torch.sqrt, torch.matmul
maximum = lambda a, b: torch.maximum(a, b if isinstance(b, torch.Tensor) else torch.as_tensor(b, dtype=a.dtype))
minimum = lambda a, b: torch.minimum(a, b if isinstance(b, torch.Tensor) else torch.as_tensor(b, dtype=a.dtype))
where = torch.where


def ternary(cond, a, b):
    return torch.where(cond, a, b)


def rand(*shape):
    """uniform [0, 1) from torch's global generator (the fixture generator seeds it; jnerf_amd draws the same shapes in the same order)"""
    if len(shape) == 1 and isinstance(shape[0], (list, tuple)):
        shape = shape[0]
    return torch.rand(list(shape))


def sum(x, dim=None, **kw):
    return x.sum() if dim is None else x.sum(dim, keepdim=_kd(kw))


def mean(x, dim=None, **kw):
    return x.mean() if dim is None else x.mean(dim, keepdim=_kd(kw))


def max(x, dim=None, **kw):
    """jittor: a reduction over `dim` returns the values only"""
    return x.max() if dim is None else x.max(dim, keepdim=_kd(kw)).values


def min(x, dim=None, **kw):
    return x.min() if dim is None else x.min(dim, keepdim=_kd(kw)).values


def norm(x, p=2, dim=-1, keepdim=False, keepdims=False, eps=1e-30):
    """jittor: misc.py norm, p == 2: `(x.sqr()).sum(dim, keepdims).maximum(eps).sqrt()` - eps floors the SUM OF SQUARES"""
    assert p == 2
    return (x * x).sum(dim, keepdim=bool(keepdim or keepdims)).clamp_min(eps).sqrt()


def cumsum(x, dim=-1):
    return torch.cumsum(x, dim)


def cumprod(x, dim=-1):
    """jittor: misc.py cumprod = exp(cumsum(log(x)))"""
    return torch.exp(torch.cumsum(torch.log(x), dim))


def gather(x, dim, index):
    """jittor: gather with an index of the variable's rank is the usual element gather.  dataset.py:205-226 calls it with a 1-D index on 2-D / 3-D variables
    (`jt.gather(self.focal_lengths, 0, img_ids)`): read as a ROW gather x[index] - the only reading under which those lines make sense (ASSUMED: Jittor's reindex
    semantics for a lower-rank index are not restated from source)"""
    if index.dim() < x.dim() and dim == 0:
        return x[index.long()]
    return torch.gather(x, dim, index.long())


def argsort(x, dim=-1, descending=False):
    """jittor: returns (index, sorted values)"""
    values, index = torch.sort(x, dim=dim, descending=descending, stable=True)
    return index, values


def searchsorted(sorted_seq, values, right=False):
    return torch.searchsorted(sorted_seq.contiguous(), values.contiguous(), right=right)


def meshgrid(*tensors):
    return torch.meshgrid(*tensors, indexing="ij")


def flip(x, dim=0):
    return torch.flip(x, [dim] if isinstance(dim, int) else list(dim))


def grad(y, x, retain_graph=True):
    """jittor: d sum(y) / d x.  (x has to be part of the graph: the fixture generator marks the ray tensors as requiring a gradient)"""
    (g,) = torch.autograd.grad(y.sum(), x, create_graph=True, retain_graph=True)
    return g


no_grad = torch.no_grad


def empty(shape, dtype=torch.float32):
    """zeros, so that fixtures never depend on uninitialised memory (GridEncode's gigabyte of scratch stays untouched virtual memory either way)"""
    return torch.zeros(_shape(shape), dtype=_dtype(dtype))


class flag_scope:
    def __init__(self, **kw):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def gc():
    pass


def sync_all(*a):
    pass


def random(shape, dtype=torch.float32):
    """jittor: jt.random = uniform [0, 1)"""
    return torch.rand(_shape(shape), dtype=_dtype(dtype))


def randperm(n):
    return torch.randperm(int(n))


def randint(low, high, shape):
    return torch.randint(low, high, _shape(shape))


def normalize(x, p=2, dim=1, eps=1e-30):
    """jittor: misc.py normalize = x / x.norm(p, dim, keepdim, eps)"""
    return x / norm(x, p, dim, keepdim=True, eps=eps)


pow = torch.pow
import types as _types                                        # noqa: E402
linalg = _types.SimpleNamespace(inv=torch.linalg.inv)


class Function:
    """jittor: jt.Function - `execute` is the forward, `grad(*output gradients)` returns one gradient (or None) per argument of execute; calling the object records both"""
    def __init__(self, *a, **k):
        pass

    def __call__(self, *args):
        fn = self

        class _Bridge(torch.autograd.Function):
            @staticmethod
            def forward(ctx, *a):
                out = fn.execute(*a)
                ctx.n_args = len(a)
                return tuple(out) if isinstance(out, (list, tuple)) else out

            @staticmethod
            def backward(ctx, *grads):
                res = fn.grad(*grads)
                res = list(res) if isinstance(res, (list, tuple)) else [res]
                return tuple(res + [None] * (ctx.n_args - len(res)))
        return _Bridge.apply(*args)


from ._code import code                                      # noqa: E402


class _Log:
    def i(self, *a):
        pass
    v = w = e = i


LOG = _Log()

from . import nn, init                                       # noqa: E402
from .nn import Module                                       # noqa: E402,F401
