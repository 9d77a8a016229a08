"""bench configuration with a 2^12-sample budget: what does the GPU do per step when the sample-proportional work is negligible?  (run under rocprofv3)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner
tb = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 12
ngp_cfg(fp16=True, aabb_scale=4, const_dt=False, n_images=50, W=400, H=400, device="cuda:0", target_batch_size=tb, n_rays_per_batch=max(4096 * tb >> 18, 64))
r = Runner()
for i in range(200):
    r.train_step(i)
torch.cuda.synchronize()
