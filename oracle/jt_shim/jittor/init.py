"""jittor.init for the shim: only what the reference's modules import at load time.  invariant_uniform is NOT restated (its bound lives inside Jittor and the fixtures
never draw from it: every weight they use is set explicitly)."""
from .nn import init as _nn_init

gauss_, constant_ = _nn_init.gauss_, _nn_init.constant_


def invariant_uniform(shape, dtype="float32", mode="fan_in"):
    raise NotImplementedError("Jittor's initialiser is not restated in the stand-in; pass explicit weights")
