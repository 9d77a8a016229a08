"""Host-side PCG32 (state bookkeeping only) — same generator as the reference's ops/op_include/pcg32/pcg32.h, whose global instance
pcg32{1337} (ops/code_ops/global_vars.py:13-16) is passed by value into the sampling kernels and advanced on the host after each use."""
import numpy as np

_MULT = 0x5851F42D4C957F2D
_M64 = (1 << 64) - 1


def _next(state, inc):
    return (state * _MULT + inc) & _M64


def pcg32_seed(initstate, initseq=1):
    inc = ((initseq << 1) | 1) & _M64
    state = _next(0, inc)
    state = (state + initstate) & _M64
    state = _next(state, inc)
    return np.array([state, inc], dtype=np.uint64)


def pcg32_advance(st, delta):
    cur_mult, cur_plus, acc_mult, acc_plus = _MULT, int(st[1]), 1, 0
    delta &= _M64
    while delta > 0:
        if delta & 1:
            acc_mult = (acc_mult * cur_mult) & _M64
            acc_plus = (acc_plus * cur_mult + cur_plus) & _M64
        cur_plus = ((cur_mult + 1) * cur_plus) & _M64
        cur_mult = (cur_mult * cur_mult) & _M64
        delta >>= 1
    st[0] = np.uint64((acc_mult * int(st[0]) + acc_plus) & _M64)
    return st
