#!/usr/bin/env python
"""Checkpoint interchange with the reference (runner/runner.py:123-151), WITHOUT Jittor.

The reference's Runner writes {global_step, model, sampler, optimizer, nested_optimizer, ema_optimizer} with jt.save.  jt.save turns every jittor.Var into a numpy array
and writes an ordinary pickle (protocol 4) followed by sha1(pickle) and the magic b"HCAJSLHD" (restated from Jittor 1.3.x's published source in
jnerf_amd/utils/jittor_pickle.py - Jittor itself is not installable here, so this is unpinned against a real Jittor install).

    python tools/convert_ckpt.py jittor2hip  logs/lego/params.pkl   out/params.pkl       # -> torch.save container (what Runner.save_ckpt writes by default)
    python tools/convert_ckpt.py hip2jittor  out/params.pkl         logs/lego/params.pkl  # -> jt.save container (what the reference's Runner.load_ckpt reads)

Runner.load_ckpt recognises either container by itself; `ckpt_format = "jittor"` in the config makes Runner.save_ckpt write the reference's.  The dictionaries have the
same keys on both sides (jnerf_amd keeps the reference's module tree and parameter names: pos_encoder.m_grid, density_mlp.con_weights / rgb_mlp.con_weights for the fused
fp16 model, density_mlp.0.weight ... rgb_mlp.4.weight for the fp32 model, the sampler's density_grid / density_grid_bitfield / density_grid_mean buffers), so the
conversion is tensor-type only.  jnerf_amd's own "extra" entry (marcher generator state, adaptive ray count) has no counterpart in the reference and is dropped /
defaulted.  fp16 parameters of a reference checkpoint become fp32 masters here (the fp16 shadows are rebuilt on load)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(direction, src, dst):
    import torch
    from jnerf_amd.utils import jittor_pickle as JP
    if direction == "jittor2hip":
        torch.save(JP.to_torch(JP.load(src)), dst)
    elif direction == "hip2jittor":
        ck = torch.load(src, map_location="cpu", weights_only=False)
        ck.pop("extra", None)
        JP.dump(ck, dst)
    else:
        raise SystemExit(__doc__)
    print("wrote", dst)


if __name__ == "__main__":
    if len(sys.argv) != 4:
        raise SystemExit(__doc__)
    main(*sys.argv[1:4])
