"""jittor.init for the shim: only what the reference's modules import at load time.  invariant_uniform is NOT restated (its bound lives inside Jittor and the fixtures
never draw from it: every weight they use is set explicitly)."""
from .nn import init as _nn_init

gauss_, constant_ = _nn_init.gauss_, _nn_init.constant_


def invariant_uniform(shape, dtype="float32", mode="fan_in"):
    raise NotImplementedError("Jittor's initialiser is not restated in the stand-in; pass explicit weights")


def uniform(shape, dtype="float32", low=0.0, high=1.0):
    import torch
    from . import _shape, _dtype
    return torch.rand(_shape(shape), dtype=_dtype(dtype)) * (high - low) + low


def zero_(x):
    return x.zero_()
