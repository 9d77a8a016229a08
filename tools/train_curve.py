"""PSNR-vs-wall-time of the FULL reference schedule (40 000 iterations, ExpDecay at 20 k / 30 k) on the bench scene: BASELINE.json's second headline
("PSNR@5min") restated for a scene that exists on the GPU box.  Writes a markdown table.  usage: train_curve.py out.md [steps] [lego|fox|bricks]
bricks = the ngp_base.py hyper-parameters on the lego-DIFFICULTY stand-in (jnerf_amd/dataset.py: bricks_field - hard surfaces, thin parts, pixel-scale texture, specular shading)
lego = projects/ngp/configs/ngp_base.py hyper-parameters (fp32, aabb 1, constant step) on the 100 x 800 x 800 procedural scene; fox = ngp_fox.py's on 50 x 400 x 400."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner
from jnerf_amd.utils.registry import build_from_cfg, DATASETS

out = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
which = sys.argv[3] if len(sys.argv) > 3 else "lego"
extra = {"invariant_uniform_gain": float(os.environ["INVARIANT_UNIFORM_GAIN"])} if os.environ.get("INVARIANT_UNIFORM_GAIN") else {}     # (r4) A/B of the two readings of Jittor's init (network.py)
torch.manual_seed(1234)
if which in ("lego", "bricks"):
    ngp_cfg(fp16=False, aabb_scale=1, const_dt=True, n_images=100, W=800, H=800, device="cuda:0", tot_train_steps=steps, scene="bricks" if which == "bricks" else "spheres", **extra)
    title = ("lego-difficulty stand-in `bricks`" if which == "bricks" else "procedural") + " 100 x 800 x 800 RGBA, ngp_base.py (lego) hyper-parameters, fp32"
else:
    ngp_cfg(fp16=True, aabb_scale=4, const_dt=False, n_images=50, W=400, H=400, device="cuda:0", tot_train_steps=steps)
    title = "procedural 50 x 400 x 400 RGBA, fox hyper-parameters, fp16"
r = Runner()
r.dataset["test"] = build_from_cfg(r.cfg.dataset.test, DATASETS)
marks = [m for m in (100, 250, 500, 1000, 2000, 5000, 10000, 20000, 30000, 40000) if m <= steps]


def psnr_now():
    vals = []
    for v in range(r.dataset["test"].n_images):
        img, _, tar = r.render_img("test", v)
        vals.append(-10 * np.log10(np.mean((img - tar) ** 2)))
    return float(np.mean(vals))


rows, train_s = [], 0.0
with r.training_stream():                                   # as Runner.train does
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        r.train_step(i)
        if i + 1 in marks:
            r.drain(); torch.cuda.synchronize()
            train_s += time.perf_counter() - t0
            rows.append((i + 1, train_s, psnr_now(), r.optimizer._nested_optimizer.lr, r.sampler.n_rays_per_batch))
            if i + 1 == steps:
                ds = r.dataset["train"]
                bits = r.sampler.density_grid_bitfield[:128 ** 3 // 8].to(torch.int32)
                occ = float((((bits[:, None] >> torch.arange(8, device=bits.device)[None]) & 1).float().mean()).item())
                extra_note = (f"occupied cells of the 128^3 grid (cascade 0) at the end: {100 * occ:.2f} %; alpha coverage of the training images {float(ds.image_data[..., 3].mean().item()):.3f}; "
                              f"samples per ray at the end {(1 << 18) / max(r.sampler.n_rays_per_batch, 1):.1f}")
            print(rows[-1], flush=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
with open(out, "w") as f:
    f.write(f"# Full training schedule on one MI355X (bench scene: {title}" + (f"; invariant_uniform_gain = {extra['invariant_uniform_gain']}" if extra else "") + ")\n\n")
    f.write(f"`python tools/train_curve.py out.md {steps} {which}` - {steps} iterations of 2^18 samples, ExpDecay x0.33 at 20 k and 30 k (ngp_base.py:31-37); PSNR = mean over the held-out test views;\n")
    f.write("training time excludes the evaluation renders.\n\n| iteration | training seconds | it/s so far | test PSNR (dB) | lr | rays / batch |\n|---|---|---|---|---|---|\n")
    for it, s, p, lr, nr in rows:
        f.write(f"| {it} | {s:.2f} | {it / s:.0f} | {p:.2f} | {lr:.4g} | {nr} |\n")
    f.write("\n" + extra_note + "\n")
print(open(out).read())
