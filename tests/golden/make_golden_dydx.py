"""Mints tests/golden/golden_dydx_v1.npz from oracle/_ref: the reference's kernel_grid (HashEncode.h:117-251) run with its dy_dx output ENABLED - the branch
grid_encode.py:96 leaves off.  Run in the build container (needs /root/reference):   python tests/golden/make_golden_dydx.py
The fixture holds inputs and outputs; checking against it needs neither /root/reference nor oracle/_ref."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import synth  # noqa: E402
from oracle import ref as R, oracle as O  # noqa: E402


def main():
    assert R.build() and R.available()
    g = {}
    x = synth.uniform_positions(192, seed=77)
    x[:6] = [[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [0.5, 0.5, 0.5], [0.999999, 0.3, 0.7], [1e-7, 1, 0]]
    g["x"] = x
    for s in (1, 4):
        table, offsets, n_params = O.level_table(s)
        for dt, nm in ((np.float32, "f32"), (np.float16, "f16")):
            grid = synth.table(n_params, dt, amp=2.0)
            out, dydx = R.hash_fwd_dydx(x, grid, offsets, s)
            g[f"out_s{s}_{nm}"] = out
            g[f"dydx_s{s}_{nm}"] = dydx
    np.savez_compressed(os.path.join(HERE, "golden_dydx_v1.npz"), **g)
    print("wrote golden_dydx_v1.npz:", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
