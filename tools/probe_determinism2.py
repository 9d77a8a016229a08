"""Repeats the density-only forward on identical inputs (refresh size) and counts elements that differ from the first result; run two copies concurrently on one GPU
(tools/gpu_r3_ac.sh).  kind = split | mfma32 (NGP_FIELD32_FWD) | fp16 (the fox kernel)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd import ops
kind = sys.argv[1]
n = 1 << 21
torch.manual_seed(0)
wd = (torch.rand(3072, device="cuda") - 0.5) * 0.6
if kind == "fp16":
    feat = (torch.randn(16, n, 2, device="cuda") * 0.3).half().contiguous()
    fn = lambda: ops.density_fwd(feat, wd.half(), n, layout=ops.LAYOUT_SOA, out_dtype=torch.float16)
else:
    feat = (torch.randn(16, n, 2, device="cuda") * 0.3).contiguous()
    fn = lambda: ops.density32_fwd(feat, wd, n, layout=ops.LAYOUT_SOA)
ref = fn().clone()
bad, worst = 0, 0
t0 = time.time()
reps = 0
while time.time() - t0 < float(sys.argv[2]):
    out = fn()
    k = int((out != ref).sum())
    reps += 1
    if k:
        bad += 1; worst = max(worst, k)
print(f"{kind}: {bad} of {reps} repetitions differ from the first result (worst: {worst} of {n} elements)")
