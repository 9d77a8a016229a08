#!/usr/bin/env python
"""Exclusive timing of the sampling kernels (nothing else on the GPU): trains `--train` steps so the occupancy grid is the steady-state one, then times
ngp_march_rays_compacted_pos alone on fresh ray batches of the steady-state size with every kernel bracketed (csrc/prof.hip).
    python tools/bench_march.py --config lego|fox"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from jnerf_amd import ops
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="lego")
ap.add_argument("--train", type=int, default=600)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
lego = a.config == "lego"
torch.manual_seed(0)
ngp_cfg(fp16=not lego, aabb_scale=1 if lego else 4, const_dt=lego, n_images=100 if lego else 50, W=800 if lego else 400, H=800 if lego else 400)
r = Runner()
for i in range(a.train):
    r.train_step(i)
r.drain(); torch.cuda.synchronize()
s, ds = r.sampler, r.dataset["train"]
res = {"config": a.config, "rays": s.n_rays_per_batch}
ops.prof_enable("*"); ops.prof_read()
tot = 0
scratch = None
for k in range(a.reps):
    img_ids, ro, rd, _ = next(ds)
    torch.cuda.synchronize()
    s.sample(img_ids, ro, rd, is_training=True) if False else None
    bs = s._sets[0]
    need = ops.march_scratch_elems(ro.shape[0])
    if scratch is None or scratch.numel() < need:
        scratch = torch.empty(need, dtype=torch.int32, device="cuda")
    ops.march_rays_compacted(ro.contiguous(), rd.contiguous(), s.density_grid_bitfield, s.aabb_range, s.rng_state, s.max_samples, s.target_batch_size, s.cone_angle_constant,
                             s.near_distance, s.const_dt, s.NERF_CASCADES, coords_out=bs["coords"], numsteps=bs["numsteps"][:ro.shape[0]], numsteps_c=bs["numsteps_c"][:ro.shape[0]],
                             counters=bs["counters"], scratch=scratch, pos_out=bs["pos"])
    torch.cuda.synchronize()
    tot += int(bs["counters"][2].item())
ops.prof_enable("")
ms = ops.prof_read()
res["samples_per_batch"] = tot / a.reps
res["kernels_us"] = {k: round(1e3 * sorted(v)[len(v) // 2], 1) for k, v in ms.items() if k.startswith("k_m")}
print(json.dumps(res))
