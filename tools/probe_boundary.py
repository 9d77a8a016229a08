"""main-stream time between the last launch of one training step and the first launch of the next (HIP events recorded inside ngp_train_step):
idle time + cross-stream waits at the step boundary, without a profiler slowing the host down"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner
ngp_cfg(fp16=True, aabb_scale=4, const_dt=False, n_images=50, W=400, H=400, device="cuda:0")
r = Runner()
for i in range(300):
    r.train_step(i)
f = r._fast
for stage in ("boundary",):
    f.timed_stage = stage
    f.stage_timings()
    for i in range(300 + 0, 300 + 160):
        r.train_step(i)
    torch.cuda.synchronize()
    ms = np.array(f.stage_timings())
    print(f"{stage:10s} n={len(ms)} mean {ms.mean() * 1e3:7.1f} us  median {np.median(ms) * 1e3:7.1f}  p90 {np.percentile(ms, 90) * 1e3:7.1f}  max {ms.max() * 1e3:8.1f}")
