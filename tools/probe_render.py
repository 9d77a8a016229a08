"""Inference throughput of the lego bench configuration: trains `steps` iterations, then renders test views with several chunk sizes and prints wall time per view,
Msamples/s and the per-kernel GPU time (csrc/prof.hip brackets) - is the render loop bound by its kernels or by the host?   python tools/probe_render.py [steps]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd import ops
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner
from jnerf_amd.utils.registry import build_from_cfg, DATASETS

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1288
torch.manual_seed(0)
ngp_cfg(fp16=False, aabb_scale=1, const_dt=True, n_images=100, W=800, H=800, device="cuda:0", render_streams=int(os.environ.get("RENDER_STREAMS", "3")))
r = Runner()
with r.training_stream():
    for i in range(steps):
        r.train_step(i)
    r.drain()
r.dataset["test"] = build_from_cfg(r.cfg.dataset.test, DATASETS)
for chunk in [int(c) for c in os.environ.get("RENDER_CHUNKS", "16384,32768").split(",")]:
    r.render_chunk = chunk
    r.render_img("test", 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n_s = 0
    for v in range(4):
        r.render_img("test", v % r.dataset["test"].n_images); n_s += r.n_samples_rendered
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.prof_enable("*"); ops.prof_read()
    r.render_img("test", 0)
    torch.cuda.synchronize()
    ops.prof_enable("")
    ms = ops.prof_read()
    tot = {k: sum(v) for k, v in ms.items()}
    print(f"chunk {chunk}: {dt / 4 * 1e3:.2f} ms per view, {n_s / dt / 1e6:.0f} Msamples/s, {n_s / 4 / 1e6:.2f} M samples per view; kernel ms per view: total {sum(tot.values()):.2f} "
          + ", ".join(f"{k} {v:.2f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]), flush=True)

# where one render_img call spends its wall time (synchronised segments)
def seg(f):
    torch.cuda.synchronize(); t = time.perf_counter(); out = f(); torch.cuda.synchronize(); return out, (time.perf_counter() - t) * 1e3
r.render_chunk = 16384
ds = r.dataset["test"]; W, H = int(r.W), int(r.H)
for rep in range(2):
    ids, t_ids = seg(lambda: torch.full((H * W,), 0, dtype=torch.int32, device=ds.device))
    (ro, rd, _), t_rays = seg(lambda: ds.generate_rays_total_test(ids, W, H))
    (imgs, alphas), t_render = seg(lambda: r._render_rays(ids, ro, rd, r.render_chunk))
    _, t_post = seg(lambda: (imgs.view(H, W, 3) + torch.tensor(r.background_color, dtype=torch.float32, device=ds.device) * (1 - alphas.view(H, W, 1))))
    _, t_d2h = seg(lambda: (imgs.cpu().numpy(), ds.image_data[0].view(H, W, 4)[..., :3].contiguous().cpu().numpy()))
    print(f"segments ms: ids {t_ids:.2f}  rays {t_rays:.2f}  render {t_render:.2f}  post {t_post:.2f}  two D2H copies {t_d2h:.2f}", flush=True)
