"""main-stream time between the last launch of one training step and the first launch of the next (HIP events recorded inside ngp_train_step):
idle time + cross-stream waits at the step boundary, without a profiler slowing the host down.  python tools/probe_boundary.py [lego|fox] [user-stream]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner
lego = (sys.argv[1] if len(sys.argv) > 1 else "lego") == "lego"
ngp_cfg(fp16=not lego, aabb_scale=1 if lego else 4, const_dt=lego, n_images=100 if lego else 50, W=800 if lego else 400, H=800 if lego else 400, device="cuda:0")
r = Runner()
import contextlib
ctx = torch.cuda.stream(torch.cuda.Stream()) if len(sys.argv) > 2 and sys.argv[2] == "user-stream" else contextlib.nullcontext()   # main work on a created stream instead of the default one
ctx.__enter__()
for i in range(600):
    r.train_step(i)
f = r._fast
f.timed_stage = "boundary"
f.stage_timings()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(600, 600 + 160):
    r.train_step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms = np.array(f.stage_timings())
print(f"boundary n={len(ms)} mean {ms.mean() * 1e3:7.1f} us  median {np.median(ms) * 1e3:7.1f}  p90 {np.percentile(ms, 90) * 1e3:7.1f}  max {ms.max() * 1e3:8.1f}   step {dt / 160 * 1e3:.4f} ms")
