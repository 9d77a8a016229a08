"""NeuS on the procedural DTU-layout scene (tests/synth_dtu.py): training curve of the frequency-encoded configuration family (reduced widths) and of the hash-grid SDF
variant (projects/neus/configs/neus_hash.py's shape) - iterations/s, colour PSNR of a training view, silhouette IoU against the view's mask and volume IoU of the learnt
SDF's interior against the scene's exact interior.  Usage: python tools/neus_curve.py out.md [steps]   (run through gpurun)"""
import os
import sys
import tempfile
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import synth_dtu
import tests.test_neus_gpu as T
from tests.test_neus_cpu import tiny_cfg
from jnerf_amd.neus_runner import NeuSRunner

ENC = dict(nerf_pos_encoder=dict(type="FrequencyEncoder", multires=4, input_dims=4), nerf_dir_encoder=dict(type="FrequencyEncoder", multires=2, input_dims=3),
           rendering_encoder=dict(type="FrequencyEncoder", multires=4, input_dims=3))
VARIANTS = {
    "frequency-encoded SDF network 4 x 128 (the reference's family, reduced)": dict(
        model=dict(type="NeuS", nerf_network=dict(D=3, W=32, output_ch=4, skips=[1], use_viewdirs=True),
                   sdf_network=dict(d_out=129, d_hidden=128, n_layers=4, skip_in=[2], bias=0.5, scale=1.0, geometric_init=True, weight_norm=True), variance_network=dict(init_val=0.3),
                   rendering_network=dict(d_feature=128, mode="idr", d_out=3, d_hidden=128, n_layers=2, weight_norm=True, squeeze_out=True)),
        encoder=dict(ENC, sdf_encoder=dict(type="FrequencyEncoder", multires=6, input_dims=3)), optim=dict(type="Adam", lr=1e-3, eps=1e-15, betas=(0.9, 0.99))),
    "hash-grid SDF network 2 x 64 (neus_hash.py)": dict(
        model=dict(type="NeuS", nerf_network=dict(D=3, W=32, output_ch=4, skips=[1], use_viewdirs=True),
                   sdf_network=dict(d_out=65, d_hidden=64, n_layers=2, skip_in=[], bias=0.5, scale=1.0, geometric_init=True, weight_norm=True), variance_network=dict(init_val=0.3),
                   rendering_network=dict(d_feature=64, mode="idr", d_out=3, d_hidden=64, n_layers=2, weight_norm=True, squeeze_out=True)),
        encoder=dict(ENC, sdf_encoder=dict(type="HashEncoder")), optim=dict(type="Adam", lr=2e-3, eps=1e-15, betas=(0.9, 0.99))),
}


def main():
    out_path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    lines = ["# NeuS training curves on the procedural DTU-layout scene (round 3)", "",
             "`python tools/neus_curve.py` on one MI355X: 24 views 192 x 144 of tests/synth_dtu.py's two-sphere scene (exact SDF known), 512 rays x (64 + 64) sections per iteration,",
             "mask loss 0.1, eikonal 0.1, no background model, warm-up 200 then cosine learning rate.  PSNR (object pixels) / silhouette IoU on training view 3 at half resolution; volume IoU on a 48^3",
             "lattice.  The networks are torch modules (rocBLAS + autograd double backward); the compositing and - in the hash variant - the encoder incl. its second-order terms are HIP.", ""]
    for name, over in VARIANTS.items():
        root = tempfile.mkdtemp(prefix="neus_curve_")
        truth = synth_dtu.make_scene(root, n_images=24, W=192, H=144)
        tiny_cfg(root, device="cuda", batch_size=512, end_iter=steps, warm_up_end=200, anneal_end=0, mask_weight=0.1,
                 render=dict(type="NeuSRenderer", n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0), **over)
        torch.manual_seed(0)
        np.random.seed(0)
        r = NeuSRunner()
        init = T._volume_iou(r)
        perm = r.get_image_perm()
        r.update_learning_rate()
        lines += [f"## {name}", "", f"volume IoU of the initial sphere: {init:.3f}", "", "| step | it/s (since last row) | colour loss | eikonal | 1/s | PSNR dB | silhouette IoU | volume IoU |", "|---|---|---|---|---|---|---|---|"]
        torch.cuda.synchronize(); t0 = time.perf_counter(); last = 0
        for it in range(steps):
            out = r.train_step(perm[it % len(perm)])
            r.update_learning_rate()
            if (it + 1) % max(steps // 6, 1) == 0:
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                psnr, sil, vol, _ = T._quality(r, truth)
                lines.append(f"| {it + 1} | {(it + 1 - last) / dt:.0f} | {float(out['color_loss']):.4f} | {float(out['eikonal_loss']):.4f} | {1.0 / float(out['s_val']):.1f} | {psnr:.2f} | {sil:.3f} | {vol:.3f} |")
                print(lines[-1], flush=True)
                torch.cuda.synchronize(); t0 = time.perf_counter(); last = it + 1
        lines.append("")
        del r
    open(out_path, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
