"""The C oracle (oracle/ngp_oracle.c - the restatement every HIP kernel is held to) replays the iterations that the REFERENCE'S OWN training loop ran on the CPU of the
build container: the reference's unmodified Python package over the Jittor stand-in, with every CUDA launch bound to the reference's own kernels compiled for the host
(tests/golden/make_golden_refrun.py -> tests/golden/golden_refrun_v1.npz).  Same data set, same pixel batches, same background colours, same initial parameters, the
global pcg32 stream from the same seed; each side with its own occupancy grid, samples, gradients, Adam moments and EMA from then on.

Checked: the occupancy statistics of every refresh, the number of samples marched and trained on, the adaptive ray count, the loss of every iteration, the parameters at
the end.  NOT bit-exact by construction: on the first refresh every trained cell holds ~1.7e-3 and the threshold IS their mean, so 1e-7 differences between the two
field-network implementations (torch GEMMs there, plain C loops here) flip cells that sit on it; the tolerances below are what that costs (measured values in the asserts'
comments).  Nothing here touches /root/reference."""
import os
import numpy as np
import torch

from tests.golden import pyref_scene as S

import pytest


@pytest.mark.parametrize("case", [c for c, v in S.REFRUN_CASES.items() if v["steps"] > 0])
def test_oracle_replays_the_references_training_run(case, tmp_path):
    """lego: ngp_base.py as shipped (unit box, constant step), 18 iterations incl. the ray-count update and the second refresh.
    cone: the same with aabb_scale 2 and const_dt False - two cascades and cone stepping, what ngp_fox.py samples with - 6 iterations."""
    from oracle import oracle as O
    C = S.REFRUN_CASES[case]
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", C["file"]))
    from jnerf_amd.utils.config import reset_cfg
    from jnerf_amd.dataset import NerfDataset
    R = S.REFRUN
    steps = G["log"].shape[0]
    S.write_rendered_nerf_dataset(str(tmp_path))
    reset_cfg(device="cpu")
    ds = NerfDataset(str(tmp_path), batch_size=R["n_rays_per_batch"], mode="train", **({"aabb_scale": C["aabb_scale"]} if C["aabb_scale"] else {}))
    assert ds.n_images == int(G["dataset.n_images"])
    ours, ref = ds.transforms_gpu.numpy(), G["dataset.transforms_gpu"]
    order = [int(np.argmin(np.abs(ours - ref[i][None]).reshape(len(ours), -1).max(-1))) for i in range(len(ref))]
    assert sorted(order) == list(range(ds.n_images))
    xforms, focal = np.ascontiguousarray(ours[order]), np.ascontiguousarray(ds.focal_lengths.numpy()[order])
    pp = np.ascontiguousarray(ds.metadata.numpy()[order][:, 4:6])
    pixels = np.ascontiguousarray(ds.image_data.numpy().reshape(ds.n_images, -1, 4)[order]).reshape(-1, 4)
    W, H = ds.resolution
    scale = C["aabb_scale"] or 1
    aabb, cascades, G3, const_dt = (0.5 - scale / 2, 0.5 + scale / 2), 5, 128 ** 3, C["const_dt"]
    assert tuple(ds.aabb_range) == aabb
    max_cascade = 0
    while (1 << max_cascade) < scale:
        max_cascade += 1
    table, offsets, n_params = O.level_table(scale)
    # ---- initial parameters: the hash table redrawn from its seed, the MLP weights from the fixture (FMLP layout: (out, in) row-major, last layer padded to 16 rows)
    grid = (torch.rand([n_params], generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["grid"])) * 2e-4 - 1e-4).numpy()
    Ws = [G[f"init.W{i}"] for i in range(5)]
    pad = lambda w: np.concatenate([w, np.zeros((16 - w.shape[0], w.shape[1]), np.float32)], 0)
    pack = np.concatenate([Ws[0].ravel(), Ws[1].ravel(), Ws[2].ravel(), Ws[3].ravel(), pad(Ws[4]).ravel()]).astype(np.float32)
    state = {k: (np.zeros_like(a), np.zeros_like(a), a.copy()) for k, a in (("grid", grid), ("pack", pack))}
    rng = O.PCG32(1337)
    density_grid, ema_step, bits, mean = None, 0, None, 0.0
    n_rays, measured, CAP, max_samples = R["n_rays_per_batch"], 0, R["target_batch_size"], R["n_rays_per_batch"] * 1024
    perms = {}
    log, refresh, ray_updates = [], [], []
    for i in range(steps):
        if i % 16 == 0:                                          # density_grid_sampler.py:137-139, 204-264
            if i == 0:
                density_grid = O.grid_mark_untrained(cascades * G3, focal, xforms, W, H)
            n_u, n_nu = (G3 * (max_cascade + 1), 0) if i < 256 else (G3 * (max_cascade + 1) // 4,) * 2
            pos, idx = O.grid_generate_samples(n_u, rng, ema_step, aabb, density_grid, max_cascade + 1, -0.01)
            if n_nu:
                p2, i2 = O.grid_generate_samples(n_nu, rng, ema_step, aabb, density_grid, max_cascade + 1, 0.01)
                pos, idx = np.concatenate([pos, p2]), np.concatenate([idx, i2])
            else:
                rng.advance()
            dens = O.density_fwd(O.hash_encode_fwd(pos, grid, table), pack[:3072])
            tmp = O.grid_splat_max(idx, dens, np.zeros(cascades * G3, np.float32))
            density_grid = O.grid_ema(density_grid, tmp)
            ema_step += 1
            bits, m = O.grid_update_bitfield(density_grid, cascades)
            mean = float(m[0])
            refresh.append([i, mean, int(np.unpackbits(bits).sum()), float((density_grid > 0).sum()), float(np.maximum(density_grid.astype(np.float64), 0).sum())])
        pid, start, count = (int(v) for v in G["batches"][i])
        assert count == n_rays, (i, count, n_rays)              # the adaptive ray count we arrived at is the batch size the reference drew
        if pid not in perms:
            perms[pid] = torch.randperm(int(G["perm_sizes"][pid - 1]), generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["perm"] + pid)).numpy()
        index = perms[pid][start:start + count].astype(np.int64)
        _, ro, rd = O.generate_rays(index, W, H, focal, pp, xforms)
        rgba = pixels[index]
        bg = torch.rand([count, 3], generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["bg"] + i + 1)).numpy()
        target = (rgba[:, :3] * rgba[:, 3:] + bg * (1 - rgba[:, 3:])).astype(np.float32)              # runner.py:66-67
        co, ns, cnt, _ = O.march_rays(ro, rd, bits, aabb, rng, max_samples, const_dt=const_dt, cascades=cascades)
        M = int(min(cnt[1], max_samples))
        cc, nsc, counter = O.compact_coords(co[:M], ns, CAP)
        k = int(min(int(nsc[:, 0].sum()), CAP))
        measured += int(counter[0])
        x, dirs = np.ascontiguousarray(cc[:k, :3]), np.ascontiguousarray(cc[:k, 4:])
        feat, sh = O.hash_encode_fwd(x, grid, table), O.sh_encode(dirs, np.float32)
        out = np.zeros((CAP, 4), np.float32)
        out[:k] = O.field_fwd(feat, sh, pack[:3072], pack[3072:])
        rgb = O.composite_fwd(out, cc, ns, nsc, bg, cascades)
        loss, dloss = O.huber(rgb, target, 0.1)
        dout = O.composite_bwd(out, cc, nsc, dloss, rgb, mean, cascades)
        dfeat, dwd, dwc = O.field_bwd(feat, sh, pack[:3072], pack[3072:], dout[:k])
        gg = O.hash_encode_bwd(x, dfeat, table, n_params)
        for name, p, g in (("grid", grid, gg), ("pack", pack, np.concatenate([dwd, dwc]))):
            mm, vv, ee = state[name]
            O.adam_ema_step(p, g, mm, vv, ee, 0.1, i + 1)       # ngp_base.py:21-37: lr 0.1 (ExpDecay starts at 20 000), betas (0.9, 0.99), eps 1e-15, EMA 0.95
        log.append([float(loss.astype(np.float64).mean()), float(loss.astype(np.float64).sum()), count, k, int(ns[:, 0].sum())])
        mid_pack = pack.copy() if i == 14 else (mid_pack if i > 14 else None)                               # the weights after 15 updates: what the fixture holds as mid.W*
        if i % 16 == 15:                                         # density_grid_sampler.py:266-271
            per_batch = max(measured / 16, 1)
            n_rays = int(min((int(n_rays * CAP / per_batch) + 127) // 128 * 128, CAP))
            ray_updates.append([i, measured, n_rays])
            measured = 0
    log, want = np.asarray(log), G["log"]
    print("loss oracle   :", np.round(log[:, 0], 6))
    print("loss reference:", np.round(want[:, 0], 6))
    print("samples trained on (oracle / reference):", log[:, 3].astype(int), want[:, 3].astype(int))
    print("refresh (step, mean, bits set, cells > 0, sum) oracle:", refresh, "reference:", G["refresh"].tolist())
    print("ray updates oracle:", ray_updates, "reference:", G["ray_updates"].tolist())
    ref_refresh = G["refresh"]
    assert len(refresh) == len(ref_refresh)
    for a, b in zip(refresh, ref_refresh):
        assert a[0] == b[0] and abs(a[3] / b[3] - 1) < 1e-3                  # the same cells have been touched (exp() of the two sides may round a splat to 0 differently)
        assert abs(a[1] / b[1] - 1) < 1e-4 and abs(a[4] / b[4] - 1) < 1e-4   # mean / sum of the grid
        assert abs(a[2] / b[2] - 1) < 0.02                                   # occupied bits: cells sitting on the threshold may fall either way
    assert np.array_equal(log[:, 2], want[:, 2])                             # rays per iteration
    assert np.abs(log[:, 4] / want[:, 4] - 1).max() < 0.02 and np.abs(log[:, 3] / want[:, 3] - 1).max() < 0.02        # samples marched / trained on
    rel = np.abs(log[:, 0] / want[:, 0] - 1)
    # up to the second refresh both sides march the same cells: measured 1e-6.  From step 16 on the two bitfields differ in ~200 of 1.3 M cells and the batch is 128 rays
    assert rel[:16].max() < 1e-4 and rel.max() < 2e-2, rel
    if len(G["ray_updates"]):
        assert [u[0] for u in ray_updates] == G["ray_updates"][:, 0].tolist()
        assert np.abs(np.asarray(ray_updates)[:, 1:] / G["ray_updates"][:, 1:] - 1).max() < 0.02
    unpack = lambda p: [p[:2048].reshape(64, 32), p[2048:3072].reshape(16, 64), p[3072:5120].reshape(64, 32), p[5120:9216].reshape(64, 64), p[9216:10240].reshape(16, 64)[:3]]
    for j, w in enumerate(unpack(mid_pack) if steps > 15 else []):
        d = np.abs(w - G[f"mid.W{j}"]).max() / np.abs(G[f"mid.W{j}"]).max()
        print(f"W{j} after 15 updates: largest difference {d:.2e} of the largest weight")
        assert d < 5e-3, (j, d)                                  # measured 1.4e-3 at worst (W0, fed by 1e-4-sized features) (Adam with lr 0.1 and eps 1e-15 turns the sign of a 1e-9 gradient into a 0.1 step: this is the tight point)
    fw = [pack[:2048].reshape(64, 32), pack[2048:3072].reshape(16, 64), pack[3072:5120].reshape(64, 32), pack[5120:9216].reshape(64, 64), pack[9216:10240].reshape(16, 64)[:3]]
    for j in range(5):
        d = np.abs(fw[j] - G[f"final.W{j}"]).max() / np.abs(G[f"final.W{j}"]).max()
        print(f"final W{j}: largest difference {d:.2e} of the largest weight")
        assert d < (0.05 if steps > 16 else 5e-3), (j, d)     # (after the second refresh the runs march different cells; a run that ends before it stays together)
    assert np.array_equal(rng.st, G["final.rng_state"])                      # the global pcg32 stream was consumed identically (one advance per generator call)
    assert bytes(G["ckpt_keys"]).decode() == "ema_optimizer,global_step,model,nested_optimizer,optimizer,sampler"


def test_oracle_replays_the_references_inference_path(tmp_path):
    """the reference's render_img / render_img_with_pose (runner.py:197-264: full-image rays, 4096-ray chunks with the last one padded by dummy rays, sampler.sample ->
    model -> rays2rgb(inference), assembly, background) on the freshly initialised model after one occupancy refresh, against the C oracle on the same rays"""
    from oracle import oracle as O
    from jnerf_amd.utils.config import reset_cfg
    from jnerf_amd.dataset import NerfDataset
    R = S.REFRUN
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", S.REFRUN_CASES["render"]["file"]))
    S.write_rendered_nerf_dataset(str(tmp_path), S.REFRUN_CASES["render"]["res"])
    reset_cfg(device="cpu")
    train = NerfDataset(str(tmp_path), batch_size=R["n_rays_per_batch"], mode="train")
    test = NerfDataset(str(tmp_path), batch_size=R["n_rays_per_batch"], mode="test")
    W, H = train.resolution
    aabb, cascades, G3 = (0.0, 1.0), 5, 128 ** 3
    table, _, n_params = O.level_table(1)
    grid = (torch.rand([n_params], generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["grid"])) * 2e-4 - 1e-4).numpy()
    Ws = [G[f"init.W{i}"] for i in range(5)]
    pad = lambda w: np.concatenate([w, np.zeros((16 - w.shape[0], w.shape[1]), np.float32)], 0)
    pack = np.concatenate([Ws[0].ravel(), Ws[1].ravel(), Ws[2].ravel(), Ws[3].ravel(), pad(Ws[4]).ravel()]).astype(np.float32)
    rng = O.PCG32(1337)
    # ---- the one refresh (density_grid_sampler.py:204-264 at step 0); mark_untrained works on the TRAIN set's cameras (any frame order gives the same grid)
    density_grid = O.grid_mark_untrained(cascades * G3, train.focal_lengths.numpy(), train.transforms_gpu.numpy(), W, H)
    pos, idx = O.grid_generate_samples(G3, rng, 0, aabb, density_grid, 1, -0.01)
    rng.advance()
    dens = O.density_fwd(O.hash_encode_fwd(pos, grid, table), pack[:3072])
    density_grid = O.grid_ema(density_grid, O.grid_splat_max(idx, dens, np.zeros(cascades * G3, np.float32)))
    bits, mean = O.grid_update_bitfield(density_grid, cascades)
    assert int(np.unpackbits(bits).sum()) == int(G["refresh"][0, 2]) and abs(float(mean[0]) / G["refresh"][0, 1] - 1) < 1e-6

    def render(ro, rd):
        n, chunk = ro.shape[0], R["n_rays_per_batch"]
        img, alpha = np.zeros((n, 3)), np.zeros((n, 1))
        for p0 in range(0, n, chunk):
            o, d = ro[p0:p0 + chunk], rd[p0:p0 + chunk]
            if o.shape[0] < chunk:                               # runner.py:214-219: the last chunk is filled up with rays (1,1,1) -> (1,1,1)
                o, d = (np.concatenate([a, np.ones((chunk - a.shape[0], 3), np.float32)]) for a in (o, d))
            co, ns, cnt, _ = O.march_rays(o, d, bits, aabb, rng, chunk * 1024, const_dt=True, cascades=cascades)
            M = int(min(cnt[1], chunk * 1024))
            x, dirs = np.ascontiguousarray(co[:M, :3]), np.ascontiguousarray(co[:M, 4:])
            out = O.field_fwd(O.hash_encode_fwd(x, grid, table), O.sh_encode(dirs, np.float32), pack[:3072], pack[3072:])
            rgb, a = O.composite_inference(out, co[:M], ns, cascades)
            img[p0:p0 + chunk], alpha[p0:p0 + chunk] = rgb[:n - p0], a[:n - p0]
        return img.reshape(H, W, 3), alpha.reshape(H, W, 1)
    assert test.n_images == 1 and np.allclose(test.transforms_gpu.numpy(), G["test.transforms_gpu"])
    every = np.arange(H * W, dtype=np.int64)
    pp = np.ascontiguousarray(test.metadata.numpy()[:, 4:6])
    _, ro, rd = O.generate_rays(every, W, H, test.focal_lengths.numpy(), pp, test.transforms_gpu.numpy())
    bg = np.zeros(3)                                             # ngp_base.py: background_color = [0, 0, 0]
    img, alpha = render(ro, rd)
    np.testing.assert_allclose(img + bg * (1 - alpha), G["render.img"], rtol=1e-4, atol=2e-6)
    rgba = test.image_data.numpy().reshape(H, W, 4)
    np.testing.assert_allclose(rgba[..., :3] * rgba[..., 3:] + bg * (1 - rgba[..., 3:]), G["render.target"], atol=1e-7)
    img2, alpha2 = render(ro, rd)                                # alpha_image = True: the colour without the background, and the opacity
    np.testing.assert_allclose(img2, G["render.img_alpha"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(alpha2, G["render.alpha"], rtol=1e-4, atol=2e-6)
    assert alpha2.max() > 0.05 and alpha2.std() > 1e-3           # not an empty render: the untrained field is a thin fog, denser along longer chords
    m = train.matrix_nerf2ngp(G["render.pose"].copy(), train.scale, train.offset)
    tp = np.ascontiguousarray(train.metadata.numpy()[:1, 4:6])
    _, ro, rd = O.generate_rays(every, W, H, train.focal_lengths.numpy()[:1], tp, np.ascontiguousarray(m.T)[None])
    img3, alpha3 = render(ro, rd)
    np.testing.assert_allclose(img3 + bg * (1 - alpha3), G["render.img_pose"], rtol=1e-4, atol=2e-6)
    assert np.array_equal(rng.st, G["final.rng_state"])


@pytest.mark.parametrize("case", [c for c, v in S.REFRUN_CASES.items() if v["steps"] > 0])
def test_host_stack_replays_the_references_training_run(case, tmp_path):
    """THIS build's host side - NerfDataset, DensityGridSampler.sample (refresh schedule, marching + compaction state, adaptive ray count), NGPNetworks on HashEncoder /
    SHEncoder with their autograd bridges, the compositing bridge, HuberLoss, Adam + ExpDecay + EMA, in Runner's module-path order - executed on the CPU with
    jnerf_amd.ops re-bound to the C oracle (tests/cpu_ops.py; the product itself binds libngp_hip.so only), on the batches of the reference's own training run."""
    from tests.cpu_ops import oracle_backed_ops
    from jnerf_amd.presets import ngp_cfg
    from jnerf_amd.runner import Runner
    from jnerf_amd import ops
    C = S.REFRUN_CASES[case]
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", C["file"]))
    R = S.REFRUN
    steps = G["log"].shape[0]
    S.write_rendered_nerf_dataset(str(tmp_path))
    cfg = ngp_cfg(fp16=False, aabb_scale=C["aabb_scale"] or 1, const_dt=C["const_dt"], n_rays_per_batch=R["n_rays_per_batch"], target_batch_size=R["target_batch_size"],
                  pipeline_sampling=False, device="cpu", log_dir=str(tmp_path / "logs"))
    one = dict(type="NerfDataset", root_dir=str(tmp_path), batch_size=R["n_rays_per_batch"], **({"aabb_scale": C["aabb_scale"]} if C["aabb_scale"] else {}))
    cfg.dataset = cfg.dfs(dict(train=dict(one, mode="train"), val=dict(one, mode="val"), test=dict(one, mode="test")))
    with oracle_backed_ops():
        r = Runner()
        s, enc, ds = r.sampler, r.model.pos_encoder, r.dataset["train"]
        assert not getattr(r.model, "fused", False) and s.max_samples == R["n_rays_per_batch"] * 1024 and s.const_dt == C["const_dt"] and ds.aabb_scale == (C["aabb_scale"] or 1)
        ours, ref = ds.transforms_gpu.numpy(), G["dataset.transforms_gpu"]
        order = torch.as_tensor([int(np.argmin(np.abs(ours - ref[i][None]).reshape(len(ours), -1).max(-1))) for i in range(len(ref))])
        assert sorted(order.tolist()) == list(range(ds.n_images))
        ds.transforms_gpu, ds.focal_lengths, ds.metadata = ds.transforms_gpu[order].contiguous(), ds.focal_lengths[order].contiguous(), ds.metadata[order].contiguous()
        ds.image_data = ds.image_data.view(ds.n_images, -1, 4)[order].contiguous()
        W, H = ds.resolution
        pixels = ds.image_data.view(-1, 4)
        Ws = [G[f"init.W{i}"] for i in range(5)]
        with torch.no_grad():
            enc.m_grid.data.copy_(torch.rand([enc.m_grid.numel()], generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["grid"])) * 2e-4 - 1e-4)
            for lin, w in zip(r.model._linears(), Ws):          # (on the CPU the network is the plain nn.Linear chain: five separate weights)
                lin.weight.copy_(torch.as_tensor(w))
        r.ema_optimizer.param_groups[0]["values"] = [p.detach().clone() for p in r.ema_optimizer.param_groups[0]["params"]]        # the EMA starts from the parameters just loaded
        perms, log, refresh, rays = {}, [], [], []
        for i in range(steps):
            cfg.m_training_step = i
            s.finish_batch_rays_update()
            pid, start, count = (int(v) for v in G["batches"][i])
            assert count == s.n_rays_per_batch == ds.batch_size, (i, count, s.n_rays_per_batch, ds.batch_size)
            if pid not in perms:
                perms[pid] = torch.randperm(int(G["perm_sizes"][pid - 1]), generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["perm"] + pid))
            index = perms[pid][start:start + count]
            img_ids, ro, rd, _ = ops.generate_rays(index, W, H, ds.focal_lengths, ds.metadata, ds.transforms_gpu)
            rgba = pixels[index]
            bg = torch.rand([count, 3], generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["bg"] + i + 1))
            target = (rgba[:, :3] * rgba[:, 3:] + bg * (1 - rgba[:, 3:])).contiguous()
            pos, dirs = s.sample(img_ids, ro, rd, is_training=True)
            if s.grid_updated_in_last_sample:
                refresh.append([i, float(s.density_grid_mean.item()), int(np.unpackbits(s.density_grid_bitfield.numpy()).sum())])
            out = r.model(pos, dirs)                              # Runner.train_step's module path (runner.py:70-76 of the reference)
            rgb = s.rays2rgb(out, bg)
            loss = r.loss_func(rgb, target)
            r.optimizer.step(loss)
            r.ema_optimizer.ema_step()
            log.append([float(loss.detach().double().mean()), int(s._counters[3]), int(s._counters[2])])
            rays.append(int(s.n_rays_per_batch))
    log, want = np.asarray(log), G["log"]
    print("loss host stack:", np.round(log[:, 0], 6))
    print("loss reference :", np.round(want[:, 0], 6))
    print("refresh host stack:", refresh, "reference:", G["refresh"][:, :3].tolist())
    for a, b in zip(refresh, G["refresh"]):
        assert a[0] == b[0] and abs(a[1] / b[1] - 1) < 1e-4 and abs(a[2] / b[2] - 1) < 0.02
    assert len(refresh) == len(G["refresh"])
    assert np.abs(log[:, 1] / want[:, 3] - 1).max() < 0.02
    rel = np.abs(log[:, 0] / want[:, 0] - 1)
    assert rel[:16].max() < 1e-4 and rel.max() < 2e-2, rel
    assert np.array_equal(s.rng_state, G["final.rng_state"])
    assert int(r.ema_optimizer.steps) == int(G["final.ema_steps"]) == steps
    fin = [lin.weight.detach().numpy() for lin in r.model._linears()]
    for j in range(5):
        d = np.abs(fin[j] - G[f"final.W{j}"]).max() / np.abs(G[f"final.W{j}"]).max()
        print(f"final W{j}: largest difference {d:.2e} of the largest weight")
        assert d < 0.05, (j, d)


def test_host_stack_replays_the_references_inference_path(tmp_path):
    """Runner.render_img / render_img_with_pose of THIS build on the CPU (module path over oracle-backed ops: full-image ray generation, chunk loop with device-side
    sample counts, sampler.sample(inference) -> model -> rays2rgb(inference), assembly, background) against the images the reference's own methods produced"""
    from tests.cpu_ops import oracle_backed_ops
    from jnerf_amd.presets import ngp_cfg
    from jnerf_amd.runner import Runner
    from jnerf_amd.utils.registry import build_from_cfg, DATASETS
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", S.REFRUN_CASES["render"]["file"]))
    R = S.REFRUN
    S.write_rendered_nerf_dataset(str(tmp_path), S.REFRUN_CASES["render"]["res"])
    cfg = ngp_cfg(fp16=False, aabb_scale=1, const_dt=True, n_rays_per_batch=R["n_rays_per_batch"], target_batch_size=R["target_batch_size"], pipeline_sampling=False,
                  device="cpu", log_dir=str(tmp_path / "logs"))
    one = dict(type="NerfDataset", root_dir=str(tmp_path), batch_size=R["n_rays_per_batch"])
    cfg.dataset = cfg.dfs(dict(train=dict(one, mode="train"), val=dict(one, mode="val"), test=dict(one, mode="test")))
    with oracle_backed_ops():
        r = Runner()
        s, enc = r.sampler, r.model.pos_encoder
        with torch.no_grad():
            enc.m_grid.data.copy_(torch.rand([enc.m_grid.numel()], generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["grid"])) * 2e-4 - 1e-4)
            for lin, w in zip(r.model._linears(), [G[f"init.W{i}"] for i in range(5)]):
                lin.weight.copy_(torch.as_tensor(w))
        cfg.m_training_step = 0
        s.update_density_grid()
        assert int(np.unpackbits(s.density_grid_bitfield.numpy()).sum()) == int(G["refresh"][0, 2])
        r.dataset["test"] = build_from_cfg(cfg.dataset.test, DATASETS)
        img, _, tar = r.render_img("test", 0)
        np.testing.assert_allclose(img, G["render.img"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(tar, G["render.target"], atol=1e-7)
        r.alpha_image = True
        img_a, alpha, _ = r.render_img("test", 0)
        np.testing.assert_allclose(img_a, G["render.img_alpha"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(alpha, G["render.alpha"], rtol=1e-4, atol=2e-6)
        r.alpha_image = False
        np.testing.assert_allclose(r.render_img_with_pose(G["render.pose"]), G["render.img_pose"], rtol=1e-4, atol=2e-6)
        assert np.array_equal(s.rng_state, G["final.rng_state"])             # three marches and one refresh: the global pcg32 stream ends where the reference's did
