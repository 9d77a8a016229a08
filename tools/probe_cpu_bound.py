"""How fast can the host issue training steps?  Runs the bench configuration with a tiny sample budget (2^12 samples per step: the GPU work becomes
negligible, the launch sequence stays the same) and prints steps/s = the CPU-side ceiling of the training loop."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner

for tb in (1 << 16, 1 << 17, 1 << 18):
    ngp_cfg(fp16=True, aabb_scale=4, const_dt=False, n_images=50, W=400, H=400, device="cuda:0", target_batch_size=tb, n_rays_per_batch=4096 * tb >> 18)
    r = Runner()
    step = 0
    for _ in range(200):
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
