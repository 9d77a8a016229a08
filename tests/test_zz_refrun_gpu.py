"""-m gpu: the HIP path replays the iterations that the REFERENCE'S OWN training loop ran (tests/golden/golden_refrun_v1.npz - the reference's unmodified Python package
over the Jittor stand-in with every CUDA launch bound to the reference's kernels compiled for the host; tests/golden/make_golden_refrun.py).  Same pixels, backgrounds,
initial parameters, pcg32 seed; the HIP side with its own occupancy grid, samples, gradients and optimiser state, through Runner's native step.  The CPU suite already
holds the C oracle to this fixture (tests/test_refrun_golden.py) and test_trajectory_gpu.py holds the HIP path to the oracle; this closes the triangle directly.

(r4) First run on an MI355X by the driver at the end of round 3 (XPASS, loss within 3 %, sample counts within 3 %: ~0.5 % of the occupancy cells sit on the threshold
after the first refresh and the __expf of the splat decides them differently).  Now the reference's own bitfields travel with the fixture (`refresh.bitfield`) and are
teacher-forced after each of the HIP side's refreshes - exactly what tests/test_trajectory_gpu.py does against the oracle - so the remaining edge of the triangle
reference -> HIP is tight: identical sample counts, losses to 1e-4; the HIP side's own bitfield may differ from the reference's only in cells that sit on the threshold."""
import os
import numpy as np
import pytest
import torch

from tests.golden import pyref_scene as S

pytestmark = [pytest.mark.gpu]


def test_hip_path_replays_the_references_training_run(tmp_path):
    from jnerf_amd import ops
    from jnerf_amd.presets import ngp_cfg
    from jnerf_amd.runner import Runner
    from jnerf_amd.fastpath import FusedTrainStep
    C = S.REFRUN_CASES["lego"]
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", C["file"]))
    R = S.REFRUN
    steps = G["log"].shape[0]
    S.write_rendered_nerf_dataset(str(tmp_path))
    cfg = ngp_cfg(fp16=False, aabb_scale=1, const_dt=True, n_rays_per_batch=R["n_rays_per_batch"], target_batch_size=R["target_batch_size"], pipeline_sampling=False,
                  log_dir=str(tmp_path / "logs"))
    one = dict(type="NerfDataset", root_dir=str(tmp_path), batch_size=R["n_rays_per_batch"])
    cfg.dataset = cfg.dfs(dict(train=dict(one, mode="train"), val=dict(one, mode="val"), test=dict(one, mode="test")))
    r = Runner()
    s, enc, ds = r.sampler, r.model.pos_encoder, r.dataset["train"]
    assert FusedTrainStep.applicable(r)
    r._fast = FusedTrainStep(r)
    assert r._fast.native and not r._fast.half and s.max_samples == R["n_rays_per_batch"] * 1024
    dev = ds.device
    # ---- the reference's frame order (it walks the directory in file-system order), then the data set's arrays in that order
    ours, ref = ds.transforms_gpu.cpu().numpy(), G["dataset.transforms_gpu"]
    order = [int(np.argmin(np.abs(ours - ref[i][None]).reshape(len(ours), -1).max(-1))) for i in range(len(ref))]
    assert sorted(order) == list(range(ds.n_images))
    idx_t = torch.as_tensor(order, device=dev)
    ds.transforms_gpu, ds.focal_lengths, ds.metadata = ds.transforms_gpu[idx_t].contiguous(), ds.focal_lengths[idx_t].contiguous(), ds.metadata[idx_t].contiguous()
    ds.image_data = ds.image_data.view(ds.n_images, -1, 4)[idx_t].contiguous()
    W, H = ds.resolution
    pixels = ds.image_data.view(-1, 4)
    # ---- initial parameters
    with torch.no_grad():
        enc.m_grid.data.copy_((torch.rand([enc.m_grid.numel()], generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["grid"])) * 2e-4 - 1e-4).to(dev))
        Ws = [G[f"init.W{i}"] for i in range(5)]
        pad = lambda w: np.concatenate([w, np.zeros((16 - w.shape[0], w.shape[1]), np.float32)], 0)
        pack = np.concatenate([Ws[0].ravel(), Ws[1].ravel(), Ws[2].ravel(), Ws[3].ravel(), pad(Ws[4]).ravel()]).astype(np.float32)
        r.model._pack32.copy_(torch.as_tensor(pack, device=dev))
        r.model._weights_version = getattr(r.model, "_weights_version", 0) + 1
    perms, log, refresh = {}, [], []
    # ---- teacher-forced occupancy: the HIP side runs its own refresh (grid, mean, bitfield), is compared with the reference's, and then marches through the reference's bits
    ref_bits = G["refresh.bitfield"]
    own_refresh = s.update_density_grid
    G3 = s.NERF_GRIDSIZE ** 3

    def forced_refresh():
        own_refresh()
        k = len(refresh)
        mine = s.density_grid_bitfield.cpu().numpy()
        mean = float(s.density_grid_mean.item())
        refresh.append([int(r.cfg.m_training_step), mean, int(np.unpackbits(mine).sum())])
        a, b = np.unpackbits(mine[:G3 // 8], bitorder="little"), np.unpackbits(ref_bits[k][:G3 // 8], bitorder="little")
        differ = np.nonzero(a != b)[0]
        grid0 = s.density_grid[:G3].cpu().numpy()
        thresh = min(s.NERF_MIN_OPTICAL_THICKNESS, mean)
        print(f"refresh {k}: cascade-0 cells set {int(a.sum())} (reference {int(b.sum())}), {differ.size} differ; threshold {thresh:.6e}")
        if differ.size:
            # refresh 0 (untrained network): measured 0 differing cells.  Refresh 1 follows 16 Adam steps with eps = 1e-15, which turn 1e-7 gradient differences into
            # full-size steps of rarely hit table entries: the two sides' densities agree to ~1 %, so cells within that of the threshold may fall either way
            # (measured on MI355X: 18 192 of 996 k cells, the farthest 0.73 % from the threshold)
            off = np.abs(grid0[differ] / thresh - 1)
            print(f"           the differing cells lie within {float(off.max()):.3%} of the threshold")
            assert off.max() < (2e-3 if k == 0 else 3e-2), ("a cell whose occupancy bit differs from the reference's is not on the threshold", float(off.max()))
        assert differ.size <= (0.002 if k == 0 else 0.04) * max(int(b.sum()), 1)
        s.density_grid_bitfield.copy_(torch.as_tensor(ref_bits[k], device=s.density_grid_bitfield.device))
        if s._occ_bounds is not None and r.cfg.march_occupancy_bounds is not False:
            ops.grid_occupied_bounds(s.density_grid_bitfield, s.NERF_CASCADES, out=s._occ_bounds)
    s.update_density_grid = forced_refresh
    for i in range(steps):
        r.cfg.m_training_step = i
        s.finish_batch_rays_update()
        pid, start, count = (int(v) for v in G["batches"][i])
        assert count == s.n_rays_per_batch, (i, count, s.n_rays_per_batch)          # the adaptive ray count we arrived at is the batch size the reference drew
        if pid not in perms:
            perms[pid] = torch.randperm(int(G["perm_sizes"][pid - 1]), generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["perm"] + pid))
        index = perms[pid][start:start + count].to(dev)
        img_ids, ro, rd, _ = ops.generate_rays(index, W, H, ds.focal_lengths, ds.metadata, ds.transforms_gpu)
        rgba = pixels[index]
        bg = torch.rand([count, 3], generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["bg"] + i + 1)).to(dev)
        target = (rgba[:, :3] * rgba[:, 3:] + bg * (1 - rgba[:, 3:])).contiguous()
        pos, dirs = s.sample(img_ids, ro, rd, is_training=True)
        b = {"step": i, "bg": bg, "target": target, "pos": pos, "dirs": dirs, "state": s.export_batch_state(), "keep": (img_ids, ro, rd)}
        loss = r._fast(b)
        log.append([float(loss.double().mean().item()), int(s._counters[3].item())])
    r.drain()
    log, want = np.asarray(log), G["log"]
    print("loss HIP      :", np.round(log[:, 0], 6))
    print("loss reference:", np.round(want[:, 0], 6))
    print("refresh HIP:", refresh, "reference:", G["refresh"][:, :3].tolist())
    assert len(refresh) == len(G["refresh"])
    for a, b in zip(refresh, G["refresh"]):
        assert a[0] == int(b[0]) and abs(a[1] / b[1] - 1) < 1e-3 and abs(a[2] / b[2] - 1) < 0.03   # step; grid mean; occupied bits of the HIP side's OWN refresh (before forcing)
    assert np.array_equal(log[:, 1].astype(np.int64), want[:, 3].astype(np.int64)), (log[:, 1], want[:, 3])        # samples trained on: identical, every iteration
    rel = np.abs(log[:, 0] / want[:, 0] - 1)
    print("relative loss difference:", np.round(rel, 7))
    # same samples, same pixels, same backgrounds: what is left is fp32 rounding of two implementations of the network / scatter / Adam over 18 iterations
    # (iterations 0-15 share the untrained network's occupancy: pure rounding, 1e-5 or better; after the 16 eps = 1e-15 Adam steps the parameters of rarely hit entries
    # have drifted apart - the same drift tests/test_trajectory_gpu.py documents against the oracle - and the last two iterations agree to 1e-3)
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):          # the MEASURED values next to the bounds: kept with the run's artefacts, quoted in DESIGN.md 7
        json.dump({"max_rel_loss_diff_iterations_0_15": float(rel[:16].max()), "bound_0_15": 1e-4, "max_rel_loss_diff_all_18": float(rel.max()), "bound_all": 1e-3,
                   "rel": [float(x) for x in rel]}, open(os.path.join(out, "refrun_measured.json"), "w"))
    assert rel[:16].max() < 1e-4 and rel.max() < 1e-3, rel
    assert np.array_equal(s.rng_state, G["final.rng_state"])                        # the global pcg32 stream was consumed identically
