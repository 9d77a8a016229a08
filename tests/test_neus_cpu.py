"""NeuS (BASELINE.json configs[4]) on the CPU: configuration files against the reference's, the DTU-layout data set against the generator's ground
truth, networks (state-dict keys, geometric initialisation, double backward), the renderer against the numpy restatement (oracle/neus_oracle.py), iso-surface
extraction, and a short NeuSRunner training run with checkpoint round trip.  The GPU side (HIP compositing kernel, hash-grid SDF network) is tests/test_neus_gpu.py."""
import os
import numpy as np
import pytest
import torch

from jnerf_amd.utils.config import Config, get_cfg, reset_cfg
from tests import synth_dtu

REF = "/root/reference/projects/neus/configs"
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "projects", "neus", "configs")


def tiny_cfg(root, **over):
    cfg = dict(
        device="cpu",
        dataset=dict(type="NeuSDataset", dataset_dir=root, render_cameras_name="cameras_sphere.npz", object_cameras_name="cameras_sphere.npz"),
        encoder=dict(nerf_pos_encoder=dict(type="FrequencyEncoder", multires=4, input_dims=4), nerf_dir_encoder=dict(type="FrequencyEncoder", multires=2, input_dims=3),
                     sdf_encoder=dict(type="FrequencyEncoder", multires=4, input_dims=3), rendering_encoder=dict(type="FrequencyEncoder", multires=2, input_dims=3)),
        model=dict(type="NeuS", nerf_network=dict(D=3, W=32, output_ch=4, skips=[1], use_viewdirs=True),
                   sdf_network=dict(d_out=33, d_hidden=32, n_layers=4, skip_in=[2], bias=0.5, scale=1.0, geometric_init=True, weight_norm=True),
                   variance_network=dict(init_val=0.3),
                   rendering_network=dict(d_feature=32, mode="idr", d_out=3, d_hidden=32, n_layers=2, weight_norm=True, squeeze_out=True)),
        render=dict(type="NeuSRenderer", n_samples=16, n_importance=16, n_outside=8, up_sample_steps=2, perturb=1.0),
        optim=dict(type="Adam", lr=2e-3, eps=1e-15, betas=(0.9, 0.99)),
        base_exp_dir=os.path.join(root, "log"), learning_rate_alpha=0.05, end_iter=40, batch_size=128, validate_resolution_level=4, warm_up_end=5, anneal_end=20,
        use_white_bkgd=False, save_freq=1000, val_freq=1000, val_mesh_freq=1000, report_freq=1000, igr_weight=0.1, mask_weight=0.0)
    cfg.update(over)
    return reset_cfg(**cfg)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("name", ["neus_womask.py", "neus_wmask.py"])
def test_neus_configs_equal_the_references(name):
    ours, ref = Config(os.path.join(HERE, name)).dump(), Config(os.path.join(REF, name)).dump()
    for d in (ours, ref):
        for k in ("name", "work_dir"):
            d.pop(k, None)
    assert ours == ref, {k: (ours.get(k), ref.get(k)) for k in set(ours) | set(ref) if ours.get(k) != ref.get(k)}


def test_projection_split_and_dataset_against_the_generators_truth(tmp_path):
    from jnerf_amd.neus_dataset import decompose_projection, NeuSDataset
    rng = np.random.default_rng(0)
    for _ in range(20):                                    # K [R | -R C] with random K, R, C comes apart again
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        q *= np.sign(np.linalg.det(q))
        K = np.array([[800 + 400 * rng.random(), 3 * rng.normal(), 300 + 50 * rng.random()], [0, 700 + 300 * rng.random(), 250 + 40 * rng.random()], [0, 0, 1.0]])
        C = rng.normal(size=3) * 3
        P = (K @ np.concatenate([q, (-q @ C)[:, None]], 1)) * (0.2 + 3 * rng.random())       # a projection matrix is defined up to scale
        K2, R2, t2 = decompose_projection(P)
        assert np.allclose(K2 / K2[2, 2], K, atol=1e-6) and np.allclose(R2, q, atol=1e-8) and np.allclose(t2[:3, 0] / t2[3, 0], C, atol=1e-8)
    truth = synth_dtu.make_scene(str(tmp_path), n_images=4, W=48, H=36)
    tiny_cfg(str(tmp_path))
    ds = NeuSDataset(str(tmp_path), "cameras_sphere.npz", "cameras_sphere.npz")
    assert ds.n_images == 4 and (ds.H, ds.W) == (36, 48) and ds.images.shape == (4, 36, 48, 3)
    assert np.allclose(ds.intrinsics_all[0, :3, :3].numpy(), truth["K"], rtol=1e-4, atol=1e-3)
    for i in range(4):                                     # poses live in the NORMALISED frame (scale_mat folded in)
        assert np.allclose(ds.pose_all[i].numpy(), truth["poses"][i], atol=2e-4)
    assert np.array_equal((ds.images[1].numpy() * 256 + 0.5).astype(np.uint8)[..., ::-1], truth["images"][1])        # BGR in memory, like cv2.imread
    # full-resolution rays hit the analytic object exactly where the mask says
    rays_o, rays_d = ds.gen_rays_at(2, resolution_level=1)
    hit, _, _ = synth_dtu._trace(rays_o.numpy().astype(np.float64), rays_d.numpy().astype(np.float64))
    assert (hit != truth["masks"][2]).mean() < 2e-3
    near, far = ds.near_far_from_sphere(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3))
    closest = (rays_o.reshape(-1, 3) + rays_d.reshape(-1, 3) * (near + 1.0)).norm(dim=-1)
    assert torch.allclose((rays_o.reshape(-1, 3) + rays_d.reshape(-1, 3) * (near + 1.0) * 1.0001).norm(dim=-1).clamp_min(0), closest, atol=1e-2) and float((far - near).mean()) == 2.0
    torch.manual_seed(0)
    batch = ds.gen_random_rays_at(1, 64)
    assert batch.shape == (64, 10) and torch.allclose(batch[:, 3:6].norm(dim=-1), torch.ones(64), atol=1e-5)
    assert set(np.unique((batch[:, 9] > 0.5).numpy())) <= {False, True}
    o2, d2 = ds.gen_rays_between(0, 1, 0.5, resolution_level=4)
    assert o2.shape == (9, 12, 3) and torch.allclose(d2.norm(dim=-1), torch.ones(9, 12), atol=1e-5)
    assert ds.image_at(0, 4).shape == (9, 12, 3)
    assert np.allclose(ds.object_bbox_min, -1.01, atol=1e-5) and np.allclose(ds.object_bbox_max, 1.01, atol=1e-5)


def test_neus_networks_keys_init_and_double_backward(tmp_path):
    from jnerf_amd.neus_network import NeuS
    synth_dtu.make_scene(str(tmp_path), n_images=2, W=16, H=12)
    cfg = tiny_cfg(str(tmp_path))
    cfg.model.sdf_network.d_hidden = 128                  # (the sphere initialisation is a statement about wide layers)
    torch.manual_seed(0)
    net = NeuS(**{k: v for k, v in cfg.model.items() if k != "type"})
    keys = set(net.state_dict().keys())
    expect = {"deviation_network.variance"}
    expect |= {f"sdf_network.lin{l}.{p}" for l in range(5) for p in ("weight", "bias")}
    expect |= {f"color_network.lin{l}.{p}" for l in range(3) for p in ("weight", "bias")}
    expect |= {f"nerf_outside.pts_linears.{l}.{p}" for l in range(3) for p in ("weight", "bias")}
    expect |= {f"nerf_outside.{m}.{p}" for m in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear") for p in ("weight", "bias")}
    assert keys == expect, keys ^ expect
    # layers without a special initialisation carry Jittor's nn.Linear default: weights U(+-sqrt(3 / fan_in)), biases U(+-1 / sqrt(fan_in))
    for lin in (net.color_network.lin1, net.nerf_outside.pts_linears[2], net.nerf_outside.feature_linear):
        fan_in = lin.weight.shape[1]
        w, b = lin.weight.detach(), lin.bias.detach()
        assert float(w.abs().max()) <= (3.0 / fan_in) ** 0.5 + 1e-6 and abs(float(w.std()) - (1.0 / fan_in) ** 0.5) < 0.15 * (1.0 / fan_in) ** 0.5
        assert float(b.abs().max()) <= (1.0 / fan_in) ** 0.5 + 1e-6
    sdf_net = net.sdf_network
    # shapes follow neus_network.py:22-47: the layer before the skip layer shrinks by the width of the embedded input (3 + 3*2*4 = 27)
    assert sdf_net.lin1.weight.shape == (128 - 27, 128) and sdf_net.lin2.weight.shape == (128, 128) and sdf_net.lin0.weight.shape == (128, 27) and sdf_net.lin4.weight.shape == (33, 128)
    assert float(sdf_net.lin0.weight.detach()[:, 3:].abs().max()) == 0.0 and float(sdf_net.lin2.weight.detach()[:, -24:].abs().max()) == 0.0
    # geometric initialisation: the fresh network is (roughly) the signed distance of a sphere of radius `bias`
    x = torch.randn(4000, 3)
    x = x / x.norm(dim=-1, keepdim=True) * torch.linspace(0.05, 1.2, 4000)[:, None]
    with torch.no_grad():
        s = sdf_net.sdf(x)[:, 0]
    r = x.norm(dim=-1)
    assert float(np.corrcoef(s.numpy(), r.numpy())[0, 1]) > 0.85          # (0.92-0.98 for widths 128-256: IDR's initialisation is a sphere up to O(1/sqrt(width)) noise)
    assert float(((s > 0) == (r > 0.5))[((r - 0.5).abs() > 0.25)].float().mean()) > 0.9
    # gradient(): equals finite differences of sdf(), and back-propagates into the parameters (second order)
    net64 = sdf_net.double()
    xs = x[:16].double()
    g = net64.gradient(xs)
    eps = 1e-6
    for d in range(3):
        e = torch.zeros(3, dtype=torch.float64)
        e[d] = eps
        with torch.no_grad():
            fd = (net64.sdf(xs + e) - net64.sdf(xs - e))[:, 0] / (2 * eps)
        assert torch.allclose(g[:, d], fd, atol=1e-6)
    eik = ((g.norm(dim=-1) - 1.0) ** 2).mean()
    grads = torch.autograd.grad(eik, [net64.lin0.weight, net64.lin3.bias])
    assert all(torch.isfinite(t).all() and float(t.abs().max()) > 0 for t in grads)
    w = net64.lin0.weight
    with torch.no_grad():                                  # one entry of the second-order gradient against a finite difference of the eikonal term
        old = float(w[3, 1])
        vals = []
        for delta in (1e-5, -1e-5):
            w[3, 1] = old + delta
            with torch.enable_grad():
                gg = net64.gradient(xs)
            vals.append(float(((gg.norm(dim=-1) - 1.0) ** 2).mean()))
        w[3, 1] = old
    assert abs((vals[0] - vals[1]) / 2e-5 - float(grads[0][3, 1])) < 1e-5 * max(1.0, abs(float(grads[0][3, 1])))


def test_renderer_pieces_equal_the_numpy_restatement(tmp_path):
    from oracle import neus_oracle as O
    from jnerf_amd import neus_renderer as R
    rng = np.random.default_rng(3)
    B, n, n_out = 7, 12, 5
    # sample_pdf (deterministic branch) incl. empty-weight rows and a weight spike
    bins = np.sort(rng.random((B, n)) * 2 + 0.5, axis=1)
    w = rng.random((B, n - 1)) ** 3
    w[0] = 0.0
    w[1, 4] = 50.0
    got = R.sample_pdf(torch.tensor(bins), torch.tensor(w), 9, det=True).numpy()
    assert np.allclose(got, O.sample_pdf_det(bins, w, 9), atol=1e-9)
    # SDF -> opacity -> weights -> colour, with and without the background model, two annealing ratios
    sdf = rng.normal(size=(B, n)) * 0.2
    cos = rng.uniform(-1.2, 1.2, size=(B, n))
    dists = rng.uniform(0.005, 0.06, size=(B, n))
    color = rng.random((B, n, 3))
    inside = (rng.random((B, n)) < 0.8).astype(np.float64)
    bg_alpha, bg_color = rng.random((B, n + n_out)) * 0.3, rng.random((B, n + n_out, 3))
    for ratio in (0.0, 0.37, 1.0):
        for bg in (False, True):
            a, p, c = R.neus_alpha(torch.tensor(sdf), torch.tensor(cos), torch.tensor(dists), 37.5, ratio)
            alpha = a.clamp(0.0, 1.0)
            col = torch.tensor(color)
            if bg:
                ins = torch.tensor(inside)
                alpha = torch.cat([alpha * ins + torch.tensor(bg_alpha[:, :n]) * (1 - ins), torch.tensor(bg_alpha[:, n:])], -1)
                col = torch.cat([col * ins[..., None] + torch.tensor(bg_color[:, :n]) * (1 - ins)[..., None], torch.tensor(bg_color[:, n:])], 1)
            weights = R._transmittance_weights(alpha)
            rgb = (col * weights[..., None]).sum(1)
            oc, ow, oa = O.composite(sdf, cos, dists, 37.5, color, inside, bg_alpha if bg else None, bg_color if bg else None, ratio)
            assert np.allclose(alpha.numpy(), oa, atol=1e-12) and np.allclose(weights.numpy(), ow, atol=1e-12) and np.allclose(rgb.numpy(), oc, atol=1e-12)
    # up_sample: the weights it samples from, through the public method (rays through the unit sphere)
    synth_dtu.make_scene(str(tmp_path), n_images=2, W=16, H=12)
    cfg = tiny_cfg(str(tmp_path))
    ren = R.NeuSRenderer(**{k: v for k, v in cfg.render.items() if k != "type"})
    o = torch.tensor(rng.normal(size=(B, 3)) * 0.1 + np.array([0, 0, -2.5]))
    d = torch.nn.functional.normalize(torch.tensor(rng.normal(size=(B, 3)) * 0.15 + np.array([0, 0, 1.0])), dim=-1)
    z = torch.tensor(np.sort(rng.random((B, n)) * 2.4 + 1.3, axis=1))
    s = torch.tensor(sdf)
    new_z = ren.up_sample(o, d, z, s, 6, 64.0)
    radius = (o[:, None, :] + d[:, None, :] * z[..., None]).norm(dim=-1).numpy()
    expect = O.sample_pdf_det(z.numpy(), O.up_sample_weights(z.numpy(), sdf, radius, 64.0), 6)
    assert np.allclose(new_z.numpy(), expect, atol=1e-9)


def test_isosurface_of_an_analytic_field(tmp_path):
    from jnerf_amd.utils.isosurface import marching_tetrahedra, write_ply
    n = 40
    ax = np.linspace(-1, 1, n)
    pts = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1)
    v, f = marching_tetrahedra(-synth_dtu.scene_sdf(pts), 0.0)
    w = v / (n - 1) * 2 - 1
    err = np.abs(synth_dtu.scene_sdf(w))                                # vertices lie on the zero set up to the linear interpolation error (largest on the crease between the spheres)
    assert err.max() < 0.02 and err.mean() < 1e-3
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key, rkey = e[:, 0] * len(v) + e[:, 1], e[:, 1] * len(v) + e[:, 0]
    assert len(np.unique(key)) == len(key) and set(key) == set(rkey)   # closed, consistently oriented 2-manifold: every directed edge once, its reverse once
    tri = w[f]
    vol = (tri[:, 0] * np.cross(tri[:, 1], tri[:, 2])).sum() / 6       # positive = outward normals
    mc = np.random.default_rng(0).uniform(-1, 1, size=(400000, 3))
    assert abs(vol - 8.0 * (synth_dtu.scene_sdf(mc) < 0).mean()) < 0.02
    assert marching_tetrahedra(np.ones((4, 4, 4)), 0.0)[1].shape == (0, 3)
    write_ply(str(tmp_path / "m.ply"), w, f)
    raw = open(tmp_path / "m.ply", "rb").read()
    head = raw[:raw.index(b"end_header\n") + 11]
    assert b"element vertex %d" % len(w) in head and len(raw) == len(head) + 12 * len(w) + 13 * len(f)


def test_neus_runner_trains_saves_and_validates_on_cpu(tmp_path):
    from jnerf_amd.neus_runner import NeuSRunner
    from jnerf.runner import NeuSRunner as Alias
    assert Alias is NeuSRunner
    synth_dtu.make_scene(str(tmp_path), n_images=4, W=32, H=24)
    tiny_cfg(str(tmp_path))
    torch.manual_seed(0)
    np.random.seed(0)
    runner = NeuSRunner()
    assert runner.get_cos_anneal_ratio() == 0.0
    runner.update_learning_rate()
    assert runner.optimizer.param_groups[0]["lr"] == 0.0               # linear warm-up from zero (neus_runner.py:148-150)
    verts, tris = runner.validate_mesh(resolution=24)                  # the freshly initialised SDF: IDR's sphere of radius ~0.5
    assert len(tris) > 100 and os.path.isfile(os.path.join(runner.base_exp_dir, "meshes_24", "00000000.ply"))
    assert 0.25 < np.linalg.norm(verts, axis=1).mean() < 0.9
    first, last = [], []
    runner.end_iter = 40
    perm = runner.get_image_perm()
    for it in range(40):
        out = runner.train_step(perm[it % 4])
        runner.update_learning_rate()
        (first if it < 8 else last if it >= 32 else []).append(float(out["color_loss"]))
        assert np.isfinite(float(out["loss"]))
    assert np.mean(last) < 0.8 * np.mean(first), (first, last)
    assert runner.iter_step == 40 and runner.get_cos_anneal_ratio() == 1.0
    assert abs(runner.optimizer.param_groups[0]["lr"] - 2e-3 * 0.05) < 1e-9       # cosine schedule ends at learning_rate_alpha
    runner.save_checkpoint()
    img = runner.validate_image(idx=1, resolution_level=4)
    assert img.shape == (6, 8, 3)
    for sub in ("validations_fine", "normals", "depths"):
        assert os.path.isfile(os.path.join(runner.base_exp_dir, sub, "00000040_0_1.png"))
    novel = runner.render_novel_image(0, 1, 0.5, resolution_level=4)
    assert novel.shape == (6, 8, 3) and novel.dtype == np.uint8
    before = {k: v.clone() for k, v in runner.neus_network.state_dict().items()}
    tiny_cfg(str(tmp_path))
    again = NeuSRunner(is_continue=True)                               # picks up checkpoints/ckpt_000040.pkl (a jt.save container)
    assert again.iter_step == 40
    for k, v in again.neus_network.state_dict().items():
        assert torch.equal(v, before[k]), k


def test_torch_hash_reference_of_the_gpu_tests_equals_the_c_oracle():
    """tests/test_neus_gpu.py checks the second-order hash kernels against fp64 autograd of a pure-torch hash encoding; that encoding is pinned here to the C oracle
    (itself bit-exact against the reference's kernel_grid incl. its dy_dx branch, tests/test_oracle_vs_ref.py): values and the dL/dx contraction"""
    from oracle import oracle as O
    from tests.test_neus_gpu import _hash_encode_ref
    for aabb in (1, 4):
        lt, _, n_params = O.level_table(aabb)
        rng = np.random.default_rng(aabb)
        x = (rng.random((257, 3)) * 0.98 + 0.01).astype(np.float32)
        table = (rng.normal(size=n_params) * 0.1).astype(np.float32)
        y, dydx = O.hash_encode_fwd_dydx(x, table, lt)
        x64 = torch.tensor(x.astype(np.float64), requires_grad=True)
        yr = _hash_encode_ref(x64, torch.tensor(table.astype(np.float64)), lt)
        assert np.abs(yr.detach().numpy() - y).max() < 2e-7
        v = rng.normal(size=(257, 32)).astype(np.float32)
        (g,) = torch.autograd.grad(yr, x64, torch.tensor(v.astype(np.float64)))
        want = O.hash_encode_bwd_input(v, dydx)
        assert np.abs(g.numpy() - want).max() < 2e-6 * np.abs(want).max()


def test_neus_golden_fixture_matches_the_checkers_that_minted_it():
    """tests/golden/golden_neus_v1.npz (minted by tests/golden/make_golden_neus.py) against a fresh evaluation of the numpy restatement: guards the fixture the GPU tests
    compare the HIP kernels with against drifting away from oracle/neus_oracle.py"""
    from oracle import neus_oracle as NO
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_neus_v1.npz"))
    for tag in ("bg", "nobg", "long"):
        inp = {k[len(f"comp_{tag}_in_"):]: g[k].astype(np.float64) for k in g.files if k.startswith(f"comp_{tag}_in_")}
        oc, ow, oa = NO.composite(inp["sdf"], inp["cos"], inp["dists"], float(g[f"comp_{tag}_inv_s"]), inp["color"], inp["inside"], inp.get("bg_alpha"), inp.get("bg_color"),
                                  float(g[f"comp_{tag}_ratio"]))
        assert np.allclose(oc, g[f"comp_{tag}_out_color"], atol=1e-12) and np.allclose(ow, g[f"comp_{tag}_out_weights"], atol=1e-12) and np.allclose(oa, g[f"comp_{tag}_out_alpha"], atol=1e-12)
        assert np.all(np.isfinite(g[f"comp_{tag}_grad_sdf"])) and abs(float(g[f"comp_{tag}_grad_inv_s"])) > 0
    for aabb in (1, 4):
        assert g[f"hash2_s{aabb}_ddy"].shape == (128, 32) and len(g[f"hash2_s{aabb}_grid_idx"]) > 10000 and np.all(np.diff(g[f"hash2_s{aabb}_grid_idx"]) > 0)
