"""Accuracy of the field kernels of the fp32 configuration against an fp64 evaluation of the same network (forward, dL/dfeatures, weight gradients) - for the variant the
environment selects (NGP_FIELD32_FWD = split | mfma32, NGP_FIELD32_BWD = 3 | 2).  tools/gpu_r3_ai.sh runs it once per variant.  Errors are maxima relative to the largest
magnitude of the quantity (the way the parity tests state their bounds)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from jnerf_amd import ops
from oracle import oracle as O
import synth

n = 1 << 16
rng = np.random.default_rng(5)
feat = (rng.normal(size=(n, 32)) * 0.5).astype(np.float32)
d = synth.unit_dirs01(n, seed=6)
wd, wc = synth.mlp_weights(7)
wd, wc = wd.astype(np.float32), wc.astype(np.float32)
dout = (rng.normal(size=(n, 4)) * 1e-4 * np.exp(rng.normal(size=(n, 1)) * 2)).astype(np.float32)      # wide dynamic range, like compositing gradients
sh = O.sh_encode(d, np.float32).astype(np.float64)
W0, W1 = wd[:2048].reshape(64, 32).astype(np.float64), wd[2048:].reshape(16, 64).astype(np.float64)
V0, V1, V2 = wc[:2048].reshape(64, 32).astype(np.float64), wc[2048:6144].reshape(64, 64).astype(np.float64), wc[6144:].reshape(16, 64).astype(np.float64)
f64 = feat.astype(np.float64)
z0 = f64 @ W0.T; h = np.maximum(z0, 0); den = h @ W1.T
in2 = np.concatenate([den, sh], 1); z2 = in2 @ V0.T; g0 = np.maximum(z2, 0); z3 = g0 @ V1.T; g1 = np.maximum(z3, 0); rgb = g1 @ V2.T
out_ref = np.concatenate([rgb[:, :3], den[:, :1]], 1)
go = dout.astype(np.float64)
dO = np.zeros((n, 16)); dO[:, :3] = go[:, :3]
dV2 = dO.T @ g1; dG1 = (dO @ V2) * (z3 > 0); dV1 = dG1.T @ g0; dG0 = (dG1 @ V1) * (z2 > 0); dV0 = dG0.T @ in2
dD = (dG0 @ V0)[:, :16]; dD[:, 0] += go[:, 3]
dW1 = dD.T @ h; dH = (dD @ W1) * (z0 > 0); dW0 = dH.T @ f64; dF = dH @ W0
dwd_ref = np.concatenate([dW0.reshape(-1), dW1.reshape(-1)]); dwc_ref = np.concatenate([dV0.reshape(-1), dV1.reshape(-1), dV2.reshape(-1)])
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
out = ops.field32_fwd(T(feat), T(d), T(wd), T(wc)).cpu().numpy().astype(np.float64)
dfeat, slabs = ops.field32_bwd(T(feat), T(d), T(wd), T(wc), T(dout))
dw = ops.reduce_slabs(slabs).cpu().numpy().astype(np.float64)
dfeat = dfeat.cpu().numpy().astype(np.float64)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
# a ReLU whose pre-activation is within rounding of zero may open on one side only: report the tail separately
e = np.abs(dfeat - dF) / np.abs(dF).max()
print(f"fwd={os.environ.get('NGP_FIELD32_FWD', 'split')} bwd={os.environ.get('NGP_FIELD32_BWD', '3')}: forward {rel(out, out_ref):.2e} | dL/dfeatures max {e.max():.2e}, 99.99th percentile {np.quantile(e, 0.9999):.2e} | "
      f"dW density {rel(dw[:3072], dwd_ref):.2e} | dW colour {rel(dw[3072:], dwc_ref):.2e}   (n = {n}, relative to the largest magnitude, vs fp64)")
