"""How fast can the host issue training steps?  Runs the bench configuration (lego | fox) with a tiny sample budget (the GPU work becomes negligible, the launch
sequence stays the same) and with the real one, and prints steps/s: the first is the CPU-side ceiling of the training loop.   python tools/probe_cpu_bound.py [lego|fox]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner

lego = (sys.argv[1] if len(sys.argv) > 1 else "lego") == "lego"
for tb in (1 << 13, 1 << 16, 1 << 18):
    ngp_cfg(fp16=not lego, aabb_scale=1 if lego else 4, const_dt=lego, n_images=100 if lego else 50, W=800 if lego else 400, H=800 if lego else 400, device="cuda:0",
            target_batch_size=tb, n_rays_per_batch=max(64, 4096 * tb >> 18))
    r = Runner()
    step = 0
    for _ in range(400):
        r.train_step(step); step += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(320):
        r.train_step(step); step += 1
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"target_batch_size {tb}: {320 / t_all:.0f} steps/s (host finished issuing after {t_issue / 320 * 1e3:.3f} ms/step, GPU done at {t_all / 320 * 1e3:.3f} ms/step)")
    r.drain(); del r
