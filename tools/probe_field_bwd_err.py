"""error distribution of the fp16 fused field backward vs the fp32 oracle chain (data for the counted-outlier bound of tests/test_hip_parity.py::test_field_bwd_vs_oracle)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
import synth, hip_impl as H
from jnerf_amd import ops
for n in (64, 1000, 8192 + 17, 1 << 16):
    rng = np.random.default_rng(10)
    feat = (rng.normal(size=(n, 32)) * 0.5).astype(np.float16)
    d = synth.unit_dirs01(n, seed=11)
    wd, wc = synth.mlp_weights(12); wd, wc = wd.astype(np.float16), wc.astype(np.float16)
    dout = (np.random.default_rng(20).normal(size=(n, 4)) * 1e-2).astype(np.float16)
    sh = O.sh_encode(d, np.float32)
    rdf, _, _ = O.field_bwd(feat.astype(np.float32), sh, wd.astype(np.float32), wc.astype(np.float32), dout.astype(np.float32))
    dfeat, _ = ops.field_bwd(H.T(feat), H.T(d), H.T(wd), H.T(wc), H.T(dout))
    e = np.abs(H.N(dfeat).astype(np.float32) - rdf) / np.abs(rdf).max()
    print(n, "max", float(e.max()), {f">{t}": int((e > t).sum()) for t in (0.01, 0.02, 0.05, 0.1)}, "of", e.size)
