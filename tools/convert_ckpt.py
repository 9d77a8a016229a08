#!/usr/bin/env python
"""Checkpoint interchange with the reference (runner/runner.py:123-151).

The reference's Runner pickles {global_step, model, sampler, optimizer, nested_optimizer, ema_optimizer} with jt.save: the payloads are jittor.Var objects,
so reading or writing that file needs Jittor itself (not installable in the MI355X image: no network).  This tool is the bridge and runs wherever BOTH
packages import:

    python tools/convert_ckpt.py jittor2hip  logs/lego/params.pkl          out/params.pkl
    python tools/convert_ckpt.py hip2jittor  out/params.pkl                logs/lego/params.pkl

The dictionaries have the same keys on both sides (jnerf_amd keeps the reference's module tree and parameter names: pos_encoder.m_grid,
density_mlp.con_weights / rgb_mlp.con_weights for the fused fp16 model, density_mlp.0.weight ... rgb_mlp.4.weight for the fp32 model, the sampler's
density_grid / density_grid_bitfield / density_grid_mean buffers), so the conversion is tensor-type only: jittor.Var <-> torch.Tensor via numpy, recursively.
jnerf_amd's own "extra" entry (marcher generator state, adaptive ray count) has no counterpart in the reference and is dropped / defaulted.
fp16 parameters of a reference checkpoint become fp32 masters here (the fp16 shadows are rebuilt on load)."""
import sys


def _map(x, leaf):
    if isinstance(x, dict):
        return {k: _map(v, leaf) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_map(v, leaf) for v in x)
    return leaf(x)


def main(direction, src, dst):
    import numpy as np
    import torch
    try:
        import jittor as jt
    except ImportError as e:
        raise SystemExit("convert_ckpt.py needs Jittor next to PyTorch (run it on the machine that wrote / will read the reference checkpoint): %r" % (e,))
    if direction == "jittor2hip":
        ck = jt.load(src)

        def leaf(v):
            if isinstance(v, jt.Var):
                a = v.numpy()
                return torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
            if isinstance(v, np.ndarray):
                return torch.from_numpy(v.astype(np.float32) if v.dtype == np.float16 else v)
            return v
        out = _map(ck, leaf)
        torch.save(out, dst)
    elif direction == "hip2jittor":
        ck = torch.load(src, map_location="cpu", weights_only=False)
        ck.pop("extra", None)

        def leaf(v):
            return jt.array(v.numpy()) if torch.is_tensor(v) else v
        jt.save(_map(ck, leaf), dst)
    else:
        raise SystemExit(__doc__)
    print("wrote", dst)


if __name__ == "__main__":
    if len(sys.argv) != 4:
        raise SystemExit(__doc__)
    main(*sys.argv[1:4])
