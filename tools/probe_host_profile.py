"""cProfile of the host side of the training loop (tiny sample budget so the GPU never back-pressures): where do the ~0.7 ms of Python per step go?"""
import cProfile
import os
import pstats
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner

ngp_cfg(fp16=True, aabb_scale=4, const_dt=False, n_images=50, W=400, H=400, device="cuda:0", target_batch_size=1 << 12, n_rays_per_batch=64)
r = Runner()
step = 0
for _ in range(100):
    r.train_step(step); step += 1
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(320):
    r.train_step(step); step += 1
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
