"""convergence of the tiny test configuration at pipeline depths 0/1/2 (debug aid)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner
for depth, fp16 in ((2, False), (2, False), (2, True)):
    torch.manual_seed(0)
    kw = dict(pipeline_sampling=False) if depth == 0 else dict(pipeline_depth=depth)
    ngp_cfg(n_images=8, W=96, H=96, target_batch_size=1 << 16, n_rays_per_batch=1024, fp16=fp16, aabb_scale=1, const_dt=True, **kw)
    r = Runner()
    losses = []
    for i in range(400):
        l = r.train_step(i)
        if i % 50 == 0:
            losses.append(round(float(l.mean().item()), 4))
    print("depth", depth, "fp16", fp16, losses, flush=True)
    r.drain(); del r
