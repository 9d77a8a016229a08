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


def test_ngp_network_wiring_and_weight_packing():
    """models/networks/ngp_network.py executed with stub encoders: (i) FMLP's flat parameter is the (out, in) matrices row-major, the last one zero-padded to 16 rows;
    (ii) the C oracle's field network - what every HIP field kernel is held to - on that flat parameter reproduces the reference's nn.Linear chain (concat order, the
    density column that is passed on, no biases), forward and backward (torch autograd through the reference's forward source)."""
    from oracle import oracle as O
    W = [G[f"ngp.W{i}"] for i in range(5)]
    pad = lambda w: np.concatenate([w, np.zeros((16 - w.shape[0], w.shape[1]), np.float32)], 0)
    wd = np.concatenate([W[0].ravel(), W[1].ravel()])
    wc = np.concatenate([W[2].ravel(), W[3].ravel(), pad(W[4]).ravel()])
    np.testing.assert_array_equal(G["ngp.pack_density"], wd)
    np.testing.assert_array_equal(G["ngp.pack_rgb"], wc)
    assert wd.size == 3072 and wc.size == 7168 and list(G["ngp.pack_out_dims"]) == [16, 3]
    feat, sh, dout = G["ngp.feat"], G["ngp.sh"], G["ngp.dout"]
    np.testing.assert_allclose(O.field_fwd(feat, sh, wd, wc), G["ngp.out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.density_fwd(feat, wd), G["ngp.density"][:, 0], rtol=1e-5, atol=1e-6)
    dfeat, dwd, dwc = O.field_bwd(feat, sh, wd, wc, dout)
    np.testing.assert_allclose(dfeat, G["ngp.dfeat"], rtol=1e-4, atol=1e-5)
    want_dwd = np.concatenate([G["ngp.dW0"].ravel(), G["ngp.dW1"].ravel()])
    want_dwc = np.concatenate([G["ngp.dW2"].ravel(), G["ngp.dW3"].ravel(), pad(G["ngp.dW4"]).ravel()])
    np.testing.assert_allclose(dwd, want_dwd, rtol=1e-4, atol=1e-4 * np.abs(want_dwd).max())
    np.testing.assert_allclose(dwc, want_dwc, rtol=1e-4, atol=1e-4 * np.abs(want_dwc).max())
    # our FMLP module (the fp16 configuration's parameter container): same flat layout, same function on its generic path
    from jnerf_amd.network import FMLP
    f = FMLP([32, 64, 16], device="cpu")
    assert f.con_weights.shape == (3072,)
    with torch.no_grad():
        f.con_weights.copy_(torch.tensor(G["ngp.pack_density"]))
    np.testing.assert_array_equal(f.layers()[0].detach().numpy(), W[0])
    np.testing.assert_array_equal(f.layers()[1].detach().numpy(), W[1])
    np.testing.assert_allclose(f(torch.tensor(feat)).detach().numpy(), G["ngp.density16"], rtol=1e-5, atol=1e-6)
    c = FMLP([32, 64, 64, 3], device="cpu")
    assert c.con_weights.shape == (7168,) and c.output_shape1 == 3
    with torch.no_grad():
        c.con_weights.copy_(torch.tensor(G["ngp.pack_rgb"]))
    rgb_in = np.concatenate([G["ngp.density16"], sh], -1)
    np.testing.assert_allclose(c(torch.tensor(rgb_in)).detach().numpy(), G["ngp.out"][:, :3], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", list(S.SAMPLER_CASES))
def test_sampler_orchestration_follows_the_references_call_trace(case, monkeypatch):
    """density_grid_sampler.py executed with recording stubs for its CUDA ops, against jnerf_amd.sampler.DensityGridSampler with recording stubs for the HIP ops:
    cascade count, what the refresh at training steps 0 / 16 / 240 / 256 / 272 launches (sample counts, thresholds, cascades, model.density block sizes, ema step),
    and the adaptive ray count of update_batch_rays."""
    import json
    from jnerf_amd import sampler as smod
    ref = json.loads(bytes(G["sampler.traces"]).decode())[case]
    args = S.SAMPLER_CASES[case]
    trace = []

    class Model(torch.nn.Module):
        def density(self, pos):
            trace.append(("model.density", int(pos.shape[0])))
            return torch.zeros(pos.shape[0], 1)

    class Dataset:
        n_images, resolution, aabb_scale = 7, [12, 10], args["aabb_scale"]
        aabb_range = (0.5 - args["aabb_scale"] / 2, 0.5 + args["aabb_scale"] / 2)
        focal_lengths, transforms_gpu, metadata, batch_size = torch.zeros(7, 2), torch.zeros(7, 4, 3), torch.zeros(7, 11), 4096
    ds = Dataset()
    cfg = reset_cfg(device="cpu", model_obj=Model(), dataset_obj=ds, pipeline_buffer_sets=1, **dict(S.SAMPLER_CFG, const_dt=args["const_dt"]))
    monkeypatch.setattr(smod.ops, "grid_mark_untrained", lambda n, focal, xf, W, H, grid=None: trace.append(("mark", int(n), int(W), int(H))))
    monkeypatch.setattr(smod.ops, "grid_generate_samples",
                        lambda n, rng, step, aabb, grid, n_cascades, thresh, pos=None, idx=None, morton_order=False:
                        trace.append(("generate", int(n), int(step.item()), int(n_cascades), float(thresh), tuple(float(v) for v in aabb))))
    monkeypatch.setattr(smod.ops, "grid_splat_max", lambda idx, d, tmp: trace.append(("splat", int(idx.shape[0]))))
    monkeypatch.setattr(smod.ops, "grid_ema", lambda grid, tmp, decay=0.95: trace.append(("ema", int(grid.shape[0]), float(decay))))
    monkeypatch.setattr(smod.ops, "grid_update_bitfield", lambda grid, cascades, mean=None, bitfield=None: trace.append(("bitfield", int(grid.shape[0]), int(cascades), int(bitfield.shape[0]))))
    smp = smod.DensityGridSampler(update_den_freq=16, update_block_size=args["block"])
    assert [smp.NERF_CASCADES, smp.max_cascade] == ref["cascades"]
    ctor = {e[1]: e for e in ref["ctor"]}
    # what the reference hands its op wrappers at construction is what ours keeps as attributes and passes per call
    assert ctor["RaySampler"][2] == [smp.near_distance, smp.cone_angle_constant, list(smp.aabb_range), smp.n_rays_per_batch, smp.MAX_STEP]
    assert ctor["CompactedCoord"][2][-1] == smp.target_batch_size and ctor["ema_grid_samples_nerf"][3]["decay"] == smp.density_grid_decay
    assert smp.max_samples == ctor["RaySampler"][2][3] * ctor["RaySampler"][2][4]            # ray_sampler.py:15: 4096 * 1024 whatever the ray count becomes
    for step in S.SAMPLER_STEPS:
        del trace[:]
        cfg.m_training_step = step
        smp.update_density_grid()
        want = ref["refresh"][str(step)]
        w_mark = [e[2][2] for e in want if e[1] == "mark_untrained_density_grid"]
        assert [t[1] for t in trace if t[0] == "mark"] == w_mark and (len(w_mark) == 1) == (step == 0)
        assert all(t[2:] == (12, 10) for t in trace if t[0] == "mark")
        # generate: (n, ema step, cascades, threshold); the reference launches the second call even for n = 0 (ours only advances the generator then)
        w_gen = [(e[2][1], int(e[2][2][1]), e[2][3] + 1, e[2][4]) for e in want if e[1] == "generate_grid_samples_nerf_nonuniform" and e[2][1] > 0]
        assert [t[1:5] for t in trace if t[0] == "generate"] == w_gen
        assert all(t[5] == tuple(ctor["generate_grid_samples_nerf_nonuniform"][3]["aabb_range"]) for t in trace if t[0] == "generate")
        assert [t[1] for t in trace if t[0] == "model.density"] == [e[2][0] for e in want if e[1] == "model.density"]
        assert sum(t[1] for t in trace if t[0] == "splat") == sum(e[2][3] for e in want if e[1] == "splat_grid_samples_nerf_max_nearest_neighbor")
        assert [t[1:3] for t in trace if t[0] == "ema"] == [(e[2][2], 0.95) for e in want if e[1] == "ema_grid_samples_nerf"]
        w_bits = [(e[2][0][1][0], e[2][2][1][0]) for e in want if e[1] == "update_bitfield"]
        assert [(t[1], t[3]) for t in trace if t[0] == "bitfield"] == w_bits and all(t[2] == ref["cascades"][0] for t in trace if t[0] == "bitfield")
        assert int(smp.density_grid_ema_step.item()) == want[-1][2]
        order = [t[0] for t in trace if t[0] in ("generate", "model.density", "ema", "bitfield")]
        assert order == sorted(order, key=["generate", "model.density", "ema", "bitfield"].index)          # samples, then queries, then the grid update
    for measured, want in zip(S.SAMPLER_MEASURED, ref["rays"]):
        smp.measured_batch_size.fill_(measured)
        smp.update_batch_rays()
        smp.finish_batch_rays_update()
        assert [smp.n_rays_per_batch, smp.dataset.batch_size, int(smp.measured_batch_size.item())] == want


@pytest.mark.parametrize("aabb", S.LEVEL_TABLE_AABBS)
def test_level_table_equals_grid_encode_init(aabb):
    """grid_encode.py:17-40 executed: offsets and parameter count of the C oracle's and of jnerf_amd.ops' level table"""
    from oracle import oracle as O
    from jnerf_amd import ops
    want = G[f"levels.{aabb}.offsets"]
    for table, offsets, n_params in (O.level_table(aabb), ops.level_table(aabb)):
        np.testing.assert_array_equal(offsets.astype(np.int64), want)
        np.testing.assert_array_equal(table[:, 0].astype(np.int64), want[:16])
        np.testing.assert_array_equal(table[:, 1].astype(np.int64), np.diff(want))
        assert n_params == int(G[f"levels.{aabb}.n_params"])
        # the kernels' fp32 per-level scale (HashEncode.h:149-151) follows the same geometric progression: res = ceil(16 s^l - 1) + 1
        s = float(G[f"levels.{aabb}.per_level_scale"])
        res = table[:, 2].astype(np.int64)
        assert all(abs(int(res[l]) - (int(np.ceil(16.0 * s ** l - 1.0)) + 1)) <= 1 for l in range(16))


def test_neus_dataset_against_the_references(tmp_path):
    """dataset/neus_dataset.py executed on tests/synth_dtu.py's scene (cv2 replaced by Pillow and by our own projection split - circular for that one function, see
    make_golden_pyref.py): what NeuSDataset holds and every ray generator"""
    from tests import synth_dtu
    from jnerf_amd.neus_dataset import NeuSDataset
    synth_dtu.make_scene(str(tmp_path), **S.NEUS_SCENE)
    reset_cfg(device="cpu")
    ds = NeuSDataset(str(tmp_path), "cameras_sphere.npz", "cameras_sphere.npz")
    pre = "neusds."
    assert [ds.n_images, ds.H, ds.W] == list(G[pre + "shape"])
    np.testing.assert_array_equal(ds.images.numpy(), G[pre + "images"])
    np.testing.assert_array_equal(ds.masks.numpy(), G[pre + "masks"])
    np.testing.assert_allclose(ds.intrinsics_all.numpy(), G[pre + "intrinsics_all"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ds.intrinsics_all_inv.numpy(), G[pre + "intrinsics_all_inv"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ds.pose_all.numpy(), G[pre + "pose_all"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(float(ds.focal), float(G[pre + "focal"]), rtol=1e-6)
    np.testing.assert_allclose(np.stack([ds.object_bbox_min, ds.object_bbox_max]), G[pre + "bbox"], rtol=1e-6, atol=1e-6)
    for lvl in (1, 2):
        o, v = ds.gen_rays_at(1, resolution_level=lvl)
        np.testing.assert_allclose(o.numpy(), G[pre + f"rays_at.{lvl}.o"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(v.numpy(), G[pre + f"rays_at.{lvl}.v"], rtol=1e-5, atol=1e-6)
    torch.manual_seed(55)
    rays = ds.gen_random_rays_at(2, 20)
    np.testing.assert_allclose(rays.numpy(), G[pre + "random_rays"], rtol=1e-5, atol=1e-6)
    o, v = ds.gen_rays_between(0, 1, 0.3, resolution_level=2)
    np.testing.assert_allclose(o.numpy(), G[pre + "between.o"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v.numpy(), G[pre + "between.v"], rtol=1e-5, atol=2e-6)
    near, far = ds.near_far_from_sphere(rays[:, :3], rays[:, 3:6])
    np.testing.assert_allclose(near.numpy(), G[pre + "near"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(far.numpy(), G[pre + "far"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", list(S.NEUS_RUN_CASES))
def test_neus_runner_six_iterations_follow_the_references(tag, tmp_path):
    """runner/neus_runner.py's train() executed end to end for six iterations (plain SGD standing in for Adam on both sides) against jnerf_amd.neus_runner.NeuSRunner from
    the same initial parameters and generator seed: loss and learning rate of every iteration, final parameters.  Covers the ray batches (random pixels of a permuted
    view order), near / far, the colour / eikonal / mask losses incl. the reference's clipped-opacity-as-logits BCE, the warm-up + cosine schedule and the annealing ratio."""
    from tests import synth_dtu
    from jnerf_amd.utils.registry import OPTIMS
    from jnerf_amd.neus_runner import NeuSRunner

    class PlainSGD:
        def __init__(self, params, lr, **kw):
            self.params = list(params)
            self.param_groups = [{"lr": lr, "params": self.params}]
            self.log = []

        def zero_grad(self):
            self.grads = None

        def backward(self, loss):
            self.grads = torch.autograd.grad(loss, self.params, allow_unused=True)
            self.log.append([float(loss.detach()), float(self.param_groups[0]["lr"])])

        def step(self):
            with torch.no_grad():
                for p, g in zip(self.params, self.grads):
                    if g is not None:
                        p -= self.param_groups[0]["lr"] * g

    synth_dtu.make_scene(str(tmp_path), **S.NEUS_SCENE)
    reset_cfg(device="cpu", **S.neus_run_cfg(str(tmp_path), **S.NEUS_RUN_CASES[tag]))
    OPTIMS._modules["PlainSGD"] = PlainSGD
    try:
        run = NeuSRunner()
    finally:
        del OPTIMS._modules["PlainSGD"]
    run.renderer.fused_composite = False
    init = {k[len(f"neusrun.{tag}.init."):]: torch.tensor(G[k]) for k in G.files if k.startswith(f"neusrun.{tag}.init.")}
    run.neus_network.load_state_dict(init)
    torch.manual_seed(777)
    run.train()
    log, want = np.asarray(run.optimizer.log), G[f"neusrun.{tag}.log"]
    assert run.iter_step == int(G[f"neusrun.{tag}.iter_step"]) == 6 and log.shape == want.shape
    np.testing.assert_allclose(log[:, 1], want[:, 1], rtol=1e-12)                         # learning rates: 0, lr/2, then the cosine
    np.testing.assert_allclose(log[:, 0], want[:, 0], rtol=2e-5)                          # losses (iterations 2.. already depend on the updated parameters)
    assert want[0, 1] == 0.0 and want[2, 1] == 0.02 and want[5, 1] < want[3, 1]
    for k, v in run.neus_network.state_dict().items():
        ref = G[f"neusrun.{tag}.final.{k}"]
        np.testing.assert_allclose(v.numpy(), ref, rtol=1e-4, atol=2e-6 + 1e-5 * np.abs(ref).max(), err_msg=k)
    moved = max(float(np.abs(G[f"neusrun.{tag}.final.{k}"] - G[f"neusrun.{tag}.init.{k}"]).max()) for k in init)
    assert moved > 1e-3                                                                      # the six steps did change the parameters


def test_origin_nerf_network_equals_the_references():
    """models/networks/ori_nerf_network.py (nerf_base.py's model) executed on the reference's FrequencyEncoders: names, shapes, outputs, .density"""
    reset_cfg(device="cpu", encoder=S.ORI_ENCODERS)
    from jnerf_amd.networks_ori import OriginNeRFNetworks
    m = OriginNeRFNetworks(**S.ORI_MODEL)
    sd = {k[len("ori.param."):]: torch.tensor(G[k]) for k in G.files if k.startswith("ori.param.")}
    assert set(sd) == set(m.state_dict()) and all(tuple(sd[k].shape) == tuple(v.shape) for k, v in m.state_dict().items())
    m.load_state_dict(sd)
    pos, d = torch.tensor(G["ori.pos"]), torch.tensor(G["ori.dir"])
    np.testing.assert_allclose(m(pos, d).detach().numpy(), G["ori.out"], **TOL)
    np.testing.assert_allclose(m.density(pos).detach().numpy(), G["ori.density"], **TOL)
