"""jnerf_amd's host-side modules against vectors that came out of the REFERENCE's own Python source (tests/golden/golden_pyref_v1.npz, written by
tests/golden/make_golden_pyref.py in the build container by executing the reference's files over the torch-backed Jittor stand-in oracle/jt_shim - see that package's
header for what such a fixture can and cannot prove).  Same inputs, same weights, same seeds (tests/golden/pyref_scene.py); nothing here touches /root/reference.

Covered: NeuS networks (parameter names / shapes, forward, input gradient), sample_pdf, NeuSRenderer.render in three configurations incl. parameter gradients through the
eikonal term, the numpy NeuS oracle (oracle/neus_oracle.py - pinned here), EMA, ExpDecay, HuberLoss, path_spherical, NerfDataset's host state, and the C oracle's ray
generation (which the HIP kernel is compared with on the GPU) against the reference's generate_random_data / generate_rays_total / generate_rays_with_pose."""
import os
import numpy as np
import pytest
import torch

from jnerf_amd.utils.config import reset_cfg
from tests.golden import pyref_scene as S

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_pyref_v1.npz"))
# fp32 chains of a few dozen operations in two different operator orders (torch's fused kernels vs the stand-in's compositions): a few ulp of the largest intermediate
TOL = dict(rtol=2e-5, atol=2e-6)


def _neus():
    reset_cfg(device="cpu", encoder=S.NEUS_ENCODERS)
    from jnerf_amd.neus_network import NeuS
    m = NeuS(**S.NEUS_MODEL)
    sd = {k[len("neus.param."):]: torch.tensor(G[k]) for k in G.files if k.startswith("neus.param.")}
    assert set(sd) == set(m.state_dict()), "parameter names differ from the reference's"
    assert all(tuple(sd[k].shape) == tuple(v.shape) for k, v in m.state_dict().items())
    m.load_state_dict(sd)
    return m


def test_neus_networks_forward_and_input_gradient():
    m = _neus()
    x = torch.tensor(G["neus.points"], requires_grad=True)
    out = m.sdf_network(x)
    np.testing.assert_allclose(out.detach().numpy(), G["neus.sdf_out"], **TOL)
    grad = m.sdf_network.gradient(x)
    np.testing.assert_allclose(grad.detach().numpy(), G["neus.sdf_gradient"], **TOL)
    col = m.color_network(x, grad, torch.tensor(G["neus.dirs"]), out[:, 1:])
    np.testing.assert_allclose(col.detach().numpy(), G["neus.color"], **TOL)
    a, c = m.nerf_outside(torch.tensor(G["neus.nerf_in"]), torch.tensor(G["neus.dirs"]))
    np.testing.assert_allclose(a.detach().numpy(), G["neus.nerf_alpha"], **TOL)
    np.testing.assert_allclose(c.detach().numpy(), G["neus.nerf_rgb"], **TOL)
    np.testing.assert_allclose(m.deviation_network(torch.zeros(1, 3)).detach().numpy(), G["neus.inv_s"], rtol=1e-6)


def test_geometric_initialisation_has_the_references_statistics():
    """our own fresh initialisation against the reference's (different generators, same distributions): IDR's sphere of radius `bias`"""
    reset_cfg(device="cpu", encoder=S.NEUS_ENCODERS)
    from jnerf_amd.neus_network import NeuS
    torch.manual_seed(3)
    m = NeuS(**S.NEUS_MODEL)
    ours = m.state_dict()
    for k in ours:
        ref = G["neus.param." + k]
        if k.startswith("sdf_network") and k.endswith("bias"):
            np.testing.assert_allclose(ours[k].numpy(), ref, atol=1e-7)                       # constants: 0 everywhere, -bias on the last layer
    w0, r0 = ours["sdf_network.lin0.weight"].numpy(), G["neus.param.sdf_network.lin0.weight"]
    assert (w0[:, 3:] == 0).all() and (r0[:, 3:] == 0).all() and abs(w0[:, :3].std() / r0[:, :3].std() - 1) < 0.35
    w2, r2 = ours["sdf_network.lin2.weight"].numpy(), G["neus.param.sdf_network.lin2.weight"]          # the skip layer: the columns fed by the encoded input start at 0
    assert (w2[:, -24:] == 0).all() and (r2[:, -24:] == 0).all() and (w2[:, :-24] != 0).all()
    w4, r4 = ours["sdf_network.lin4.weight"].numpy(), G["neus.param.sdf_network.lin4.weight"]
    assert abs(w4.mean() - r4.mean()) < 1e-3 and w4.std() < 2e-4 and r4.std() < 2e-4
    x = torch.tensor(G["neus.points"])
    ref_sdf = G["neus.sdf_out"][:, 0]
    radius = np.linalg.norm(G["neus.points"], axis=-1)
    ours_sdf = m.sdf_network.sdf(x).detach().numpy()[:, 0]
    # both are rough spheres of radius 0.5 (narrow network: rough), and equally rough
    assert np.corrcoef(ours_sdf, radius - 0.5)[0, 1] > 0.6 and np.corrcoef(ref_sdf, radius - 0.5)[0, 1] > 0.6


def test_sample_pdf():
    from jnerf_amd.neus_renderer import sample_pdf
    from oracle import neus_oracle as NO
    bins, w = torch.tensor(G["pdf.bins"]), torch.tensor(G["pdf.weights"])
    np.testing.assert_allclose(sample_pdf(bins, w, 6, det=True).numpy(), G["pdf.det"], **TOL)
    np.testing.assert_allclose(NO.sample_pdf_det(G["pdf.bins"], G["pdf.weights"], 6), G["pdf.det"], **TOL)             # the numpy oracle, pinned
    torch.manual_seed(77)
    np.testing.assert_allclose(sample_pdf(bins, w, 6, det=False).numpy(), G["pdf.rand"], **TOL)


@pytest.mark.parametrize("tag", list(S.NEUS_RENDER_CASES))
def test_renderer_outputs_and_parameter_gradients(tag):
    from jnerf_amd.neus_renderer import NeuSRenderer
    perturb, n_outside, anneal, white = S.NEUS_RENDER_CASES[tag]
    m = _neus()
    r = NeuSRenderer(**dict(S.NEUS_RENDERER, n_outside=n_outside, perturb=perturb), fused_composite=False)
    r.set_neus_network(m)
    o, d, near, far = (torch.tensor(a) for a in S.neus_rays())
    torch.manual_seed(4321)
    res = r.render(o, d, near, far, background_rgb=torch.ones(1, 3) if white else None, cos_anneal_ratio=anneal)
    want = {k[len(f"render.{tag}."):]: G[k] for k in G.files if k.startswith(f"render.{tag}.") and ".grad." not in k}
    assert set(want) == set(res), "the dict NeuSRenderer.render returns has other keys than the reference's"
    # sample positions first: everything else depends on them (up-sampling sorts and merges them - a single swapped pair would show here)
    np.testing.assert_allclose(res["z_vals"].detach().numpy(), want["z_vals"], rtol=1e-5, atol=1e-6)
    for k in want:
        got = res[k].detach().numpy().astype(np.float32)
        np.testing.assert_allclose(got.reshape(want[k].shape), want[k], rtol=1e-5, atol=1e-6, err_msg=k)         # measured: 1.2e-7 absolute at worst (one ulp)
    scalar = (res["color_fine"] * torch.tensor(S.NEUS_COLOR_PROBE)).sum() + 0.1 * res["gradient_error"] + 0.05 * res["weight_sum"].sum()
    names = [k for k, _ in m.named_parameters()]
    grads = torch.autograd.grad(scalar, [p for _, p in m.named_parameters()], allow_unused=True)
    for k, g in zip(names, grads):
        ref = G[f"render.{tag}.grad.{k}"]
        got = np.zeros_like(ref) if g is None else g.numpy()
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(got - ref).max() <= 1e-4 * scale + 1e-7, (k, np.abs(got - ref).max(), scale)          # measured: 1e-5 of the largest entry at worst
    if n_outside == 0:
        assert all(np.abs(G[f"render.{tag}.grad.{k}"]).max() == 0 for k in names if k.startswith("nerf_outside"))     # no background model: its parameters get nothing


def test_numpy_neus_oracle_against_the_reference_run():
    """oracle/neus_oracle.py (the checker of the HIP compositing kernel) reproduces the reference's alpha / weights / colour from the reference's own sdf, gradients
    and positions - this is what pins it"""
    from oracle import neus_oracle as NO
    m = _neus()
    tag = "plain"
    z = G[f"render.{tag}.z_vals"].astype(np.float64)
    o, d, _, _ = S.neus_rays()
    B, n = z.shape
    dists = np.concatenate([z[:, 1:] - z[:, :-1], np.full((B, 1), 2.0 / S.NEUS_RENDERER["n_samples"])], -1)
    mid = z + 0.5 * dists
    pts = o[:, None, :] + d[:, None, :] * mid[:, :, None]
    sdf = G[f"render.{tag}.sdf"].reshape(B, n).astype(np.float64)
    grads = G[f"render.{tag}.gradients"].astype(np.float64)
    cos = (d[:, None, :] * grads).sum(-1)
    inside = G[f"render.{tag}.inside_sphere"].reshape(B, n)
    np.testing.assert_array_equal(inside, (np.sqrt(np.maximum((pts.astype(np.float32) ** 2).sum(-1), 1e-6)) < 1.0).astype(np.float32))
    inv_s = float(np.exp(0.3 * 10.0))
    with torch.no_grad():
        x = torch.tensor(pts.reshape(-1, 3).astype(np.float32), requires_grad=True)
    with torch.enable_grad():
        out = m.sdf_network(x)
        g = m.sdf_network.gradient(x)
        color = m.color_network(x, g, torch.tensor(np.broadcast_to(d[:, None, :], (B, n, 3)).reshape(-1, 3).copy()), out[:, 1:]).detach().numpy().reshape(B, n, 3)
    oc, ow, oa = NO.composite(sdf, cos, dists, inv_s, color.astype(np.float64), inside.astype(np.float64), None, None, S.NEUS_RENDER_CASES[tag][2])
    np.testing.assert_allclose(oa, G[f"render.{tag}.alpha"], rtol=2e-4, atol=5e-6)
    np.testing.assert_allclose(ow, G[f"render.{tag}.weights"], rtol=2e-4, atol=5e-6)
    np.testing.assert_allclose(oc, G[f"render.{tag}.color_fine"], rtol=2e-4, atol=5e-6)


def test_ema_expdecay_huber_camera_path():
    from jnerf_amd.optim import EMA, ExpDecay
    from jnerf_amd.losses import HuberLoss
    from jnerf_amd.camera_path import path_spherical
    p0 = G["ema.p0"]
    ps = [torch.tensor(p0[:7].copy()), torch.tensor(p0[7:].reshape(3, 2).copy())]
    ema = EMA(ps, decay=0.95)
    for step in range(6):
        for i, p in enumerate(ps):
            p.data.add_(torch.tensor(S.ema_delta(step, i, p.shape)))
        ema.ema_step()
        np.testing.assert_allclose(np.concatenate([p.numpy().reshape(-1) for p in ps]), G["ema.trajectory"][step], rtol=1e-6, atol=1e-7)

    class Nested:
        lr = 0.1

        def step(self, loss=None):
            pass
    dec = ExpDecay(Nested(), decay_start=20, decay_interval=10, decay_base=0.33, decay_end=45)
    lrs = []
    for _ in range(70):
        dec.step()
        lrs.append(dec._nested_optimizer.lr)
    np.testing.assert_allclose(lrs, G["expdecay.lrs"], rtol=1e-12)
    loss = HuberLoss(delta=0.1)(torch.tensor(G["huber.x"]), torch.tensor(G["huber.target"]))
    np.testing.assert_allclose(loss.numpy(), G["huber.loss"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.stack([np.asarray(p) for p in path_spherical(7)]), G["camera_path.poses"], atol=1e-6)


@pytest.mark.parametrize("mode", ["train", "val", "test"])
def test_nerf_dataset_host_state_and_oracle_ray_generation(mode, tmp_path):
    from jnerf_amd.dataset import NerfDataset
    from oracle import oracle as O
    S.write_nerf_dataset(str(tmp_path))
    reset_cfg(device="cpu")
    ds = NerfDataset(str(tmp_path), batch_size=32, mode=mode, **S.NERF_DATASET_ARGS)
    pre = f"dataset.{mode}."
    assert ds.n_images == int(G[pre + "n_images"]) and list(ds.resolution) == list(G[pre + "resolution"])
    assert (ds.aabb_scale, ds.aabb_range[0], ds.aabb_range[1]) == tuple(G[pre + "aabb"])
    # the reference walks the directory in file-system order, we in sorted order: the SET of frames is the contract, so frames are matched by their transform
    ref_x, our_x = G[pre + "transforms_gpu"], ds.transforms_gpu.numpy()
    order = [int(np.argmin(np.abs(our_x - ref_x[i][None]).reshape(len(our_x), -1).max(-1))) for i in range(len(ref_x))]
    assert sorted(order) == list(range(ds.n_images))
    np.testing.assert_array_equal(our_x[order], ref_x)
    np.testing.assert_array_equal(ds.metadata.numpy()[order], G[pre + "metadata"])
    np.testing.assert_array_equal(ds.focal_lengths.numpy()[order], G[pre + "focal_lengths"])
    np.testing.assert_array_equal(ds.image_data.numpy().reshape(ds.n_images, -1, 4)[order], G[pre + "image_data"])
    if mode != "train":
        return
    # ray generation: the C oracle (what the HIP kernel k_generate_rays is held to, tests/test_hip_parity.py) on the REFERENCE's arrays against the reference's rays
    W, H = ds.resolution
    idx = G[pre + "index"]
    ids, ro, rd = O.generate_rays(idx, W, H, G[pre + "focal_lengths"], np.ascontiguousarray(G[pre + "metadata"][:, 4:6]), ref_x)
    np.testing.assert_array_equal(ids, G[pre + "img_id"])
    np.testing.assert_array_equal(ro, G[pre + "rays_o"])
    np.testing.assert_allclose(rd, G[pre + "rays_d"], rtol=2e-6, atol=2e-7)
    np.testing.assert_array_equal(ds.image_data.numpy().reshape(ds.n_images, -1, 4)[order].reshape(-1, 4)[idx], G[pre + "rgb"])
    every = np.arange(H * W, dtype=np.int64) + 1 * H * W                      # generate_rays_total(img 1): all its pixels, row-major
    _, ro, rd = O.generate_rays(every, W, H, G[pre + "focal_lengths"], np.ascontiguousarray(G[pre + "metadata"][:, 4:6]), ref_x)
    np.testing.assert_array_equal(ro, np.broadcast_to(G[pre + "total.rays_o"], ro.shape))
    np.testing.assert_allclose(rd, G[pre + "total.rays_d"], rtol=2e-6, atol=2e-7)
    # generate_rays_with_pose: the pose goes through matrix_nerf2ngp, then the same arithmetic with image 0's intrinsics
    m = ds.matrix_nerf2ngp(S.NOVEL_POSE.copy(), ds.scale, ds.offset)
    xf = np.ascontiguousarray(m.T)[None]
    _, ro, rd = O.generate_rays(np.arange(H * W, dtype=np.int64), W, H, G[pre + "focal_lengths"][:1], np.ascontiguousarray(G[pre + "metadata"][:1, 4:6]), xf)
    np.testing.assert_allclose(ro, G[pre + "pose.rays_o"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rd, G[pre + "pose.rays_d"][..., 0], rtol=2e-6, atol=2e-7)
