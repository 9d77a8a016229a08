"""Hashes of the field kernels' outputs (fp16 and split-fp32; forward, density, backward; both layouts) on fixed inputs - run once per library file by
tools/ab_prebuilt.sh to show whether a new build returns the same BITS as an earlier commit's (profiles/r06s_lib_bits.txt)."""
import hashlib, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from jnerf_amd import ops
import synth

h = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
for n in (8192 + 17, 1 << 18):
    rng = np.random.default_rng(10)
    feat = rng.normal(size=(n, 32)) * 0.5
    d = synth.unit_dirs01(n, seed=11)
    wd, wc = synth.mlp_weights(12)
    dout = np.random.default_rng(20).normal(size=(n, 4)) * 1e-2
    T = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).cuda()
    for name, fwd, den, bwd, dt in (("fp16", ops.field_fwd, ops.density_fwd, ops.field_bwd, np.float16), ("fp32", ops.field32_fwd, ops.density32_fwd, ops.field32_bwd, np.float32)):
        for layout in (ops.LAYOUT_AOS, ops.LAYOUT_SOA):
            f = feat if layout == ops.LAYOUT_AOS else feat.reshape(n, 16, 2).transpose(1, 0, 2)
            out = fwd(T(f, dt), T(d, np.float32), T(wd, dt), T(wc, dt), layout=layout)
            dn = den(T(f, dt), T(wd, dt), n, layout=layout)
            dfeat, slabs = bwd(T(f, dt), T(d, np.float32), T(wd, dt), T(wc, dt), T(dout, dt), layout=layout)
            torch.cuda.synchronize()
            print("bits", n, name, layout, h(out), h(dn), h(dfeat), h(slabs), "%.6e" % float(slabs.double().abs().sum()), flush=True)
