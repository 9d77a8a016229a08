#!/usr/bin/env python
"""Golden vectors out of the reference's PURE-PYTHON modules, executed in the build container (where /root/reference exists) over the torch-backed Jittor stand-in
oracle/jt_shim (Jittor itself is not installed and cannot be): the reference's files are loaded from where they lie, nothing of them is copied.

    python tests/golden/make_golden_pyref.py          ->  tests/golden/golden_pyref_v1.npz

What runs (reference file -> what the fixture holds):
  models/networks/neus_network.py + models/position_encoders/freq_encoder/freq_encoder.py
        NeuS networks with their geometric initialisation: every parameter, SDF / feature / gradient / colour / background outputs on fixed points
  models/samplers/neus_render/renderer.py
        sample_pdf (det and random), NeuSRenderer.render on a ray batch - without and with perturbation, with and without the background model - every entry of the
        returned dict, plus the gradients of a scalar of those outputs w.r.t. every network parameter (torch autograd through the REFERENCE's forward code, eikonal
        term included: a double backward)
  optims/ema.py, optims/expdecay.py, models/losses/huber_loss.py, dataset/camera_path.py
        EMA.ema_step trajectories, ExpDecay's learning-rate sequence on the ngp_base.py schedule, HuberLoss values, path_spherical poses
  models/networks/ngp_network.py
        NGPNetworks in the configuration ngp_base.py selects (fp16 unset -> the nn.Linear / ReLU chain), with the CUDA encoders replaced by stubs that return supplied
        encodings: outputs [n, 4], .density, and autograd gradients w.r.t. the encodings and the five weight matrices; FMLP's flat `con_weights` for the same matrices
  models/networks/ori_nerf_network.py
        OriginNeRFNetworks (nerf_base.py's model) on the reference's FrequencyEncoders: parameters, outputs, .density
  dataset/neus_dataset.py
        NeuSDataset on tests/synth_dtu.py's scene (cv2.imread / decomposeProjectionMatrix replaced: see the comment at the stub): images, masks, intrinsics and their
        inverses, poses, focal, object bounding box, gen_rays_at (two resolution levels), gen_random_rays_at, gen_rays_between, near_far_from_sphere
  runner/neus_runner.py
        NeuSRunner.train for six iterations end to end (data set -> rays -> renderer -> colour / eikonal / mask losses -> backward -> step, learning-rate and cosine
        annealing schedules, image permutation) with a plain-SGD stand-in for Adam: loss and learning rate of every iteration, initial and final parameters
  models/position_encoders/hash_encoder/grid_encode.py
        GridEncode.__init__'s level table (offsets, parameter count, per-level scale) for aabb_scale 1 .. 128
  models/samplers/density_grid_sampler/density_grid_sampler.py
        DensityGridSampler with its jt.code wrappers replaced by recorders: constructor arguments handed to the ops, the occupancy-refresh call sequence at several
        training steps (sample counts, thresholds, model.density block sizes, ema step) and update_batch_rays' adaptive ray count - as a JSON trace
  dataset/dataset.py (+ dataset_util.py)
        NerfDataset on a small transforms_*.json data set written by tests/golden/pyref_scene.py: transforms, metadata, focal lengths, aabb, image data,
        generate_random_data on fixed pixel indices, generate_rays_total, generate_rays_with_pose - the latter two called the way runner.py:207-208,243 call this
        family: (W, H) handed to the parameters named (H, W), which is what makes their pixel grid row-major for non-square images
        (generate_rays_total_test is NOT executed: it calls jt.gather with an index of lower rank than its input, whose reindex semantics the stand-in does not restate)
The consumer is tests/test_pyref_golden.py (CPU): jnerf_amd's modules on the same inputs / weights / seeds against these vectors."""
import importlib.util
import os
import tempfile
import sys
import types
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/python/jnerf"
sys.path.insert(0, os.path.join(ROOT, "oracle", "jt_shim"))
sys.path.insert(0, ROOT)
import torch                                                  # noqa: E402
import jittor as jt                                           # noqa: E402  (the stand-in)
from tests.golden import pyref_scene                          # noqa: E402



# ------------------------------------------------------------------ the few jnerf.* names the reference modules import, as stand-ins (none of this is reference code)
class AttrDict(dict):
    """attribute access, missing keys are None (jnerf/utils/config.py's Config behaves like this)"""
    def __getattr__(self, k):
        v = self.get(k)
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v


class Registry:
    def __init__(self):
        self.m = {}

    def register_module(self, name=None):
        def deco(cls):
            self.m[name or cls.__name__] = cls
            return cls
        return deco


def build_from_cfg(cfg, registry, **kw):
    if cfg is None:
        return None
    args = dict(cfg)
    args.update(kw)
    return registry.m[args.pop("type")](**args)


CFG = AttrDict()
REG = {n: Registry() for n in ("SAMPLERS", "NETWORKS", "ENCODERS", "OPTIMS", "LOSSES", "DATASETS")}


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


stub("jnerf")
stub("jnerf.utils")
stub("jnerf.utils.config", get_cfg=lambda: CFG, init_cfg=lambda *a: None)
stub("jnerf.utils.registry", build_from_cfg=build_from_cfg, **REG)
stub("jnerf.ops")
stub("jnerf.ops.code_ops")
stub("jnerf.ops.code_ops.global_vars", global_headers="", proj_options={})
stub("mcubes")
stub("cv2")
stub("jittor_utils")
stub("jittor.dataset", Dataset=object)


def _imread(path):
    from PIL import Image
    return np.asarray(Image.open(path))


stub("imageio", imread=_imread, imwrite=None)


def load(rel, name, package=None):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel), submodule_search_locations=None)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def sampler_traces():
    """models/samplers/density_grid_sampler/density_grid_sampler.py with its eight jt.code wrappers replaced by recorders: which op is called when, with which scalar
    arguments (occupancy-grid refresh schedule, sample counts, thresholds, block splitting of model.density, the adaptive ray count)."""
    trace = []

    def op(name, ret):
        class Op:
            def __init__(self, header, *a, **kw):
                trace.append(["ctor", name, [_plain(v) for v in a], {k: _plain(v) for k, v in kw.items()}])

            def execute(self, *a, **kw):
                trace.append(["call", name, [_plain(v) for v in a]])
                return ret(*a)
            __call__ = execute
        Op.__name__ = name
        return Op

    def _plain(v):
        if isinstance(v, torch.Tensor):
            return ["tensor", list(v.shape)] if v.numel() != 1 else ["scalar", float(v.reshape(-1)[0])]
        if isinstance(v, (list, tuple)):
            return [_plain(x) for x in v]
        return v if isinstance(v, (int, float, str, bool, type(None))) else str(type(v).__name__)

    pkg = "ref_dgs_pkg"
    stub(pkg)
    stub(pkg + ".ema_grid_samples_nerf", ema_grid_samples_nerf=op("ema_grid_samples_nerf", lambda tmp, grid, n: grid))
    stub(pkg + ".generate_grid_samples_nerf_nonuniform",
         generate_grid_samples_nerf_nonuniform=op("generate_grid_samples_nerf_nonuniform", lambda grid, n, step, casc, thresh: (torch.zeros(n, 3), torch.zeros(n, dtype=torch.int32))))
    stub(pkg + ".splat_grid_samples_nerf_max_nearest_neighbor",
         splat_grid_samples_nerf_max_nearest_neighbor=op("splat_grid_samples_nerf_max_nearest_neighbor", lambda idx, mlp, tmp, n: tmp))
    stub(pkg + ".update_bitfield", update_bitfield=op("update_bitfield", lambda grid, mean, bits: (bits, mean)))
    stub(pkg + ".mark_untrained_density_grid", mark_untrained_density_grid=op("mark_untrained_density_grid", lambda focal, xf, n: torch.zeros(n)))
    stub(pkg + ".compacted_coord", CompactedCoord=op("CompactedCoord", lambda *a: None))
    stub(pkg + ".ray_sampler", RaySampler=op("RaySampler", lambda *a: None))
    stub(pkg + ".calc_rgb", CalcRgb=op("CalcRgb", lambda *a: None))
    mod = load("models/samplers/density_grid_sampler/density_grid_sampler.py", pkg + ".density_grid_sampler", package=pkg)

    class Model:
        def density(self, pos):
            trace.append(["call", "model.density", [int(pos.shape[0])]])
            return torch.zeros(pos.shape[0], 1)

    result = {}
    for case, args in pyref_scene.SAMPLER_CASES.items():
        del trace[:]

        class Dataset:
            n_images, resolution, aabb_scale = 7, [12, 10], args["aabb_scale"]
            aabb_range = (0.5 - args["aabb_scale"] / 2, 0.5 + args["aabb_scale"] / 2)
            focal_lengths, transforms_gpu, metadata, batch_size = torch.zeros(7, 2), torch.zeros(7, 4, 3), torch.zeros(7, 11), 4096
        CFG.clear()
        CFG.update(model_obj=Model(), dataset_obj=Dataset(), **pyref_scene.SAMPLER_CFG)
        CFG["const_dt"] = args["const_dt"]
        smp = mod.DensityGridSampler(update_den_freq=16, update_block_size=args["block"])
        res = {"ctor": list(trace), "cascades": [smp.NERF_CASCADES, smp.max_cascade], "refresh": {}, "rays": []}
        for step in pyref_scene.SAMPLER_STEPS:
            del trace[:]
            CFG["m_training_step"] = step
            smp.update_density_grid()
            res["refresh"][str(step)] = list(trace) + [["state", "density_grid_ema_step", int(smp.density_grid_ema_step.item())]]
        for measured in pyref_scene.SAMPLER_MEASURED:
            smp.measured_batch_size = torch.tensor([measured], dtype=torch.int32)
            smp.update_batch_rays()
            res["rays"].append([smp.n_rays_per_batch, smp.dataset.batch_size, int(smp.measured_batch_size.item())])
        result[case] = res
    return result


def main():
    assert os.path.isdir(REF), "run this in the build container: the reference tree is needed"
    out = {}
    load("models/position_encoders/freq_encoder/freq_encoder.py", "ref_freq_encoder")
    net = load("models/networks/neus_network.py", "ref_neus_network")
    ren = load("models/samplers/neus_render/renderer.py", "ref_neus_renderer")

    # ---------------------------------------------------------------- NeuS
    enc = pyref_scene.NEUS_ENCODERS
    CFG.clear()
    CFG.update(encoder=enc, fp16=False)
    torch.manual_seed(1234)
    neus = net.NeuS(**pyref_scene.NEUS_MODEL)
    params = dict(neus.named_parameters())
    for k, v in params.items():
        out["neus.param." + k] = npy(v)
    # the freshly initialised SDF network is IDR's sphere: keep its statistics too (the consumer checks its own initialisation against them)
    pts = pyref_scene.neus_points()
    x = torch.tensor(pts, requires_grad=True)
    sdf_all = neus.sdf_network(x)
    out["neus.points"] = pts
    out["neus.sdf_out"] = npy(sdf_all)
    out["neus.sdf_gradient"] = npy(neus.sdf_network.gradient(x))
    dirs = pyref_scene.neus_dirs(len(pts))
    out["neus.dirs"] = dirs
    out["neus.color"] = npy(neus.color_network(x, neus.sdf_network.gradient(x), torch.tensor(dirs), sdf_all[:, 1:]))
    p4 = np.concatenate([pts / 2.0, np.full((len(pts), 1), 0.5, np.float32)], -1).astype(np.float32)
    a, c = neus.nerf_outside(torch.tensor(p4), torch.tensor(dirs))
    out["neus.nerf_in"], out["neus.nerf_alpha"], out["neus.nerf_rgb"] = p4, npy(a), npy(c)
    out["neus.inv_s"] = npy(neus.deviation_network(torch.zeros(1, 3)))

    # sample_pdf
    rng = np.random.default_rng(5)
    bins = np.sort(rng.random((5, 9)).astype(np.float32) * 3.0, -1)
    w = rng.random((5, 8)).astype(np.float32) ** 3
    out["pdf.bins"], out["pdf.weights"] = bins, w
    out["pdf.det"] = npy(ren.sample_pdf(torch.tensor(bins), torch.tensor(w), 6, det=True))
    torch.manual_seed(77)
    out["pdf.rand"] = npy(ren.sample_pdf(torch.tensor(bins), torch.tensor(w), 6, det=False))

    # render: (perturb, n_outside, cos_anneal_ratio, background)
    for tag, (perturb, n_outside, anneal, white) in pyref_scene.NEUS_RENDER_CASES.items():
        # weights as leaves so that parameter gradients through the reference's forward exist
        for v in params.values():
            v.requires_grad_(True)
        r = ren.NeuSRenderer(**dict(pyref_scene.NEUS_RENDERER, n_outside=n_outside, perturb=perturb))
        r.set_neus_network(neus)
        rays_o, rays_d, near, far = pyref_scene.neus_rays()
        ro, rd = torch.tensor(rays_o, requires_grad=True), torch.tensor(rays_d, requires_grad=True)
        torch.manual_seed(4321)
        bg = torch.ones(1, 3) if white else None
        res = r.render(ro, rd, torch.tensor(near), torch.tensor(far), background_rgb=bg, cos_anneal_ratio=anneal)
        for k, v in res.items():
            out[f"render.{tag}.{k}"] = npy(v).astype(np.float32) if npy(v).dtype != np.bool_ else npy(v)
        scalar = (res["color_fine"] * torch.tensor(pyref_scene.NEUS_COLOR_PROBE)).sum() + 0.1 * res["gradient_error"] + 0.05 * res["weight_sum"].sum()
        names = [k for k in params]
        grads = torch.autograd.grad(scalar, [params[k] for k in names], allow_unused=True)
        for k, g in zip(names, grads):
            out[f"render.{tag}.grad.{k}"] = npy(g) if g is not None else np.zeros(params[k].shape, np.float32)
        for v in params.values():
            v.requires_grad_(False)

    # ---------------------------------------------------------------- optimiser wrappers, loss, camera path
    ema_mod = load("optims/ema.py", "ref_ema")
    dec_mod = load("optims/expdecay.py", "ref_expdecay")
    hub_mod = load("models/losses/huber_loss.py", "ref_huber")
    cam_mod = load("dataset/camera_path.py", "ref_camera_path")
    rng = np.random.default_rng(11)
    p0 = [rng.normal(size=(7,)).astype(np.float32), rng.normal(size=(3, 2)).astype(np.float32)]
    ps = [torch.tensor(p) for p in p0]
    ema = ema_mod.EMA(ps, decay=0.95)
    traj = []
    for step in range(6):
        for i, p in enumerate(ps):                           # the "optimiser" moves the parameters, then the EMA pulls them back
            p.update(p + torch.tensor(pyref_scene.ema_delta(step, i, p.shape)))
        ema.ema_step()
        traj.append(np.concatenate([npy(p).reshape(-1) for p in ps]))
    out["ema.p0"] = np.concatenate([p.reshape(-1) for p in p0])
    out["ema.trajectory"] = np.stack(traj)

    class Nested:
        lr = 0.1

        def step(self, loss=None):
            pass
    dec = dec_mod.ExpDecay(Nested(), decay_start=20, decay_interval=10, decay_base=0.33, decay_end=45)
    lrs = []
    for _ in range(70):
        dec.step()
        lrs.append(dec._nested_optimizer.lr)
    out["expdecay.lrs"] = np.asarray(lrs, np.float64)
    hub = hub_mod.HuberLoss(delta=0.1)
    a, b = rng.random((64, 3)).astype(np.float32), rng.random((64, 3)).astype(np.float32)
    b[:8] = a[:8] + np.float32(0.1) * np.sign(rng.normal(size=(8, 3))).astype(np.float32)      # |d| at the switch-over
    out["huber.x"], out["huber.target"], out["huber.loss"] = a, b, npy(hub(torch.tensor(a), torch.tensor(b)))
    out["camera_path.poses"] = np.stack([npy(p) for p in cam_mod.path_spherical(7)])

    # ---------------------------------------------------------------- NerfDataset
    stub("ref_dataset_pkg")
    load("dataset/dataset_util.py", "ref_dataset_pkg.dataset_util", package="ref_dataset_pkg")
    ds_mod = load("dataset/dataset.py", "ref_dataset_pkg.dataset", package="ref_dataset_pkg")
    with tempfile.TemporaryDirectory() as d:
        pyref_scene.write_nerf_dataset(d)
        for mode in ("train", "val", "test"):
            torch.manual_seed(9)
            ds = ds_mod.NerfDataset(d, batch_size=32, mode=mode, **pyref_scene.NERF_DATASET_ARGS)
            pre = f"dataset.{mode}."
            out[pre + "n_images"] = np.int64(ds.n_images)
            out[pre + "resolution"] = np.asarray(ds.resolution, np.int64)
            out[pre + "aabb"] = np.asarray([ds.aabb_scale, ds.aabb_range[0], ds.aabb_range[1]], np.float64)
            out[pre + "transforms_gpu"] = npy(ds.transforms_gpu)
            out[pre + "metadata"] = npy(ds.metadata)
            out[pre + "focal_lengths"] = npy(ds.focal_lengths)
            out[pre + "image_data"] = npy(ds.image_data).astype(np.float32)
            if mode == "train":
                idx = pyref_scene.pixel_indices(ds.n_images, ds.H, ds.W)
                ids, ro, rd, rgb = ds.generate_random_data(torch.tensor(idx), len(idx))
                out[pre + "index"], out[pre + "img_id"], out[pre + "rays_o"], out[pre + "rays_d"], out[pre + "rgb"] = idx, npy(ids), npy(ro), npy(rd), npy(rgb)
                ro, rd = ds.generate_rays_total(1, ds.W, ds.H)              # (W, H) into the parameters named (H, W): how runner.py:207-208,243 call the siblings
                out[pre + "total.rays_o"], out[pre + "total.rays_d"] = npy(ro), npy(rd)
                pose = torch.tensor(pyref_scene.NOVEL_POSE)
                ro, rd = ds.generate_rays_with_pose(pose, ds.W, ds.H)      # runner.py:243
                out[pre + "pose.rays_o"], out[pre + "pose.rays_d"] = npy(ro), npy(rd)
    # ---------------------------------------------------------------- NGPNetworks: the nn.Linear chain ngp_base.py runs (fp32), and FMLP's weight packing
    stub("jnerf.ops.code_ops.fully_fused_mlp", FullyFusedMlp_weight=lambda weights: None)          # the binary tiny-cuda-nn call: never executed here

    class StubEncoder(jt.nn.Module):
        """stands in for HashEncoder / SHEncoder (CUDA ops): returns the encodings the fixture supplies"""
        def __init__(self, out_dim):
            self.out_dim, self.values = out_dim, None

        def execute(self, x):
            return self.values

    REG["ENCODERS"].m["StubPos"] = lambda: StubEncoder(32)
    REG["ENCODERS"].m["StubDir"] = lambda: StubEncoder(16)
    ngp = load("models/networks/ngp_network.py", "ref_ngp_network")
    CFG.clear()
    CFG.update(encoder=dict(pos_encoder=dict(type="StubPos"), dir_encoder=dict(type="StubDir")), fp16=False)
    net = ngp.NGPNetworks(use_fully=True)                         # fp16 unset: falls through to the nn.Linear branch (ngp_network.py:59-67)
    mats = pyref_scene.ngp_weights()                              # (out, in) matrices, every value exactly representable in fp16
    lins = [net.density_mlp[0], net.density_mlp[2], net.rgb_mlp[0], net.rgb_mlp[2], net.rgb_mlp[4]]
    for lin, w in zip(lins, mats):
        assert tuple(lin.weight.shape) == w.shape and lin.bias is None
        lin.weight = torch.tensor(w, requires_grad=True)
    feat, sh, dout = pyref_scene.ngp_inputs()
    net.pos_encoder.values = torch.tensor(feat, requires_grad=True)
    net.dir_encoder.values = torch.tensor(sh)
    dummy = torch.zeros(len(feat), 3)
    res = net(dummy, dummy)
    out["ngp.feat"], out["ngp.sh"], out["ngp.dout"] = feat, sh, dout
    out["ngp.out"] = npy(res)
    out["ngp.density"] = npy(net.density(dummy))
    out["ngp.density16"] = npy(net.density_mlp(net.pos_encoder.values))
    grads = torch.autograd.grad((res * torch.tensor(dout)).sum(), [net.pos_encoder.values] + [l.weight for l in lins])
    out["ngp.dfeat"] = npy(grads[0])
    for i, g in enumerate(grads[1:]):
        out[f"ngp.dW{i}"] = npy(g)
    for i, w in enumerate(mats):
        out[f"ngp.W{i}"] = w
    # FMLP's flat parameter for the same matrices, through the reference's packing code (FMLP takes (in, out) matrices: ngp_network.py:16, 23-30)
    pack_d = ngp.FMLP(None, weights=[torch.tensor(mats[0].T.copy()), torch.tensor(mats[1].T.copy())])
    pack_c = ngp.FMLP(None, weights=[torch.tensor(m.T.copy()) for m in mats[2:]])
    out["ngp.pack_density"], out["ngp.pack_rgb"] = npy(pack_d.con_weights).astype(np.float32), npy(pack_c.con_weights).astype(np.float32)
    out["ngp.pack_out_dims"] = np.asarray([pack_d.output_shape1, pack_c.output_shape1], np.int64)

    # ---------------------------------------------------------------- OriginNeRFNetworks (models/networks/ori_nerf_network.py; BASELINE config 0) on real FrequencyEncoders
    ori = load("models/networks/ori_nerf_network.py", "ref_ori_nerf_network")
    CFG.clear()
    CFG.update(encoder=pyref_scene.ORI_ENCODERS, fp16=False)
    torch.manual_seed(8)
    onet = ori.OriginNeRFNetworks(**pyref_scene.ORI_MODEL)
    for k, v in onet.named_parameters():
        out["ori.param." + k] = npy(v)
    opos, odir = torch.tensor(pyref_scene.neus_points(30) * 0.5 + 0.5), torch.tensor(pyref_scene.neus_dirs(30))
    out["ori.pos"], out["ori.dir"] = npy(opos), npy(odir)
    out["ori.out"], out["ori.density"] = npy(onet(opos, odir)), npy(onet.density(opos))

    # ---------------------------------------------------------------- NeuSDataset (dataset/neus_dataset.py) on the procedural DTU-layout scene of tests/synth_dtu.py
    from tests import synth_dtu
    from jnerf_amd.neus_dataset import decompose_projection

    def cv_imread(path):
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1])          # cv2.imread: BGR, uint8

    # cv2 is not installed: imread through Pillow, decomposeProjectionMatrix through jnerf_amd's own RQ split (which tests/test_neus_cpu.py checks against closed forms) -
    # for THAT function the fixture is circular; everything the reference computes from K, R, t onwards is its own code
    stub("cv2", imread=cv_imread, decomposeProjectionMatrix=lambda P: decompose_projection(P))
    nds = load("dataset/neus_dataset.py", "ref_neus_dataset")
    with tempfile.TemporaryDirectory() as d:
        synth_dtu.make_scene(d, **pyref_scene.NEUS_SCENE)
        ds = nds.NeuSDataset(d, "cameras_sphere.npz", "cameras_sphere.npz")
        pre = "neusds."
        out[pre + "shape"] = np.asarray([ds.n_images, ds.H, ds.W], np.int64)
        out[pre + "images"], out[pre + "masks"] = npy(ds.images).astype(np.float32), npy(ds.masks).astype(np.float32)
        out[pre + "intrinsics_all"], out[pre + "intrinsics_all_inv"], out[pre + "pose_all"] = npy(ds.intrinsics_all), npy(ds.intrinsics_all_inv), npy(ds.pose_all)
        out[pre + "focal"] = npy(ds.focal)
        out[pre + "bbox"] = np.stack([ds.object_bbox_min, ds.object_bbox_max]).astype(np.float64)
        for lvl in (1, 2):
            ro, rv = ds.gen_rays_at(1, resolution_level=lvl)
            out[pre + f"rays_at.{lvl}.o"], out[pre + f"rays_at.{lvl}.v"] = npy(ro), npy(rv)
        torch.manual_seed(55)
        out[pre + "random_rays"] = npy(ds.gen_random_rays_at(2, 20))
        ro, rv = ds.gen_rays_between(0, 1, 0.3, resolution_level=2)
        out[pre + "between.o"], out[pre + "between.v"] = npy(ro), npy(rv)
        rays = torch.tensor(out[pre + "random_rays"])
        near, far = ds.near_far_from_sphere(rays[:, :3], rays[:, 3:6])
        out[pre + "near"], out[pre + "far"] = npy(near), npy(far)

    # ---------------------------------------------------------------- NeuSRunner.train (runner/neus_runner.py): six iterations end to end with a plain-SGD optimiser
    class PlainSGD:
        """stands in for Adam (which lives inside Jittor): p -= lr * g.  Records loss and learning rate of every iteration."""
        def __init__(self, params, lr, **kw):
            self.params = [p.requires_grad_(True) for p in params]
            self.param_groups = [{"lr": lr, "params": self.params}]
            self.log = []

        def zero_grad(self):
            self.grads = None

        def backward(self, loss):
            self.grads = torch.autograd.grad(loss, self.params, allow_unused=True)
            self.log.append([float(loss), float(self.param_groups[0]["lr"])])

        def step(self):
            with torch.no_grad():
                for p_, g_ in zip(self.params, self.grads):
                    if g_ is not None:
                        p_ -= self.param_groups[0]["lr"] * g_

    REG["OPTIMS"].m["PlainSGD"] = PlainSGD
    jt.nn.binary_cross_entropy_with_logits = lambda output, target: torch.nn.functional.binary_cross_entropy_with_logits(output, target)     # jittor: mean of the stable form
    stub("trimesh")
    stub("jnerf.dataset")
    stub("jnerf.dataset.neus_dataset", NeuSDataset=nds.NeuSDataset)
    stub("jnerf.models")
    stub("jnerf.models.networks")
    stub("jnerf.models.networks.neus_network", NeuS=sys.modules["ref_neus_network"].NeuS)
    stub("jnerf.models.samplers")
    stub("jnerf.models.samplers.neus_render")
    stub("jnerf.models.samplers.neus_render.renderer", NeuSRenderer=ren.NeuSRenderer)
    sys.modules["jnerf.utils.registry"].SCHEDULERS = Registry()
    rmod = load("runner/neus_runner.py", "ref_neus_runner")
    for tag, over in pyref_scene.NEUS_RUN_CASES.items():
        with tempfile.TemporaryDirectory() as d:
            synth_dtu.make_scene(d, **pyref_scene.NEUS_SCENE)
            CFG.clear()
            CFG.update(pyref_scene.neus_run_cfg(d, **over))
            torch.manual_seed(99)
            run = rmod.NeuSRunner()
            for k, v in run.neus_network.named_parameters():
                out[f"neusrun.{tag}.init.{k}"] = npy(v).copy()
            run.dataset.pose_all.requires_grad_(True)       # jt.grad(sdf, points) differentiates w.r.t. ANY variable; torch needs the points to hang off a leaf that requires a gradient
            torch.manual_seed(777)
            run.train()
            out[f"neusrun.{tag}.log"] = np.asarray(run.optimizer.log, np.float64)
            for k, v in run.neus_network.named_parameters():
                out[f"neusrun.{tag}.final.{k}"] = npy(v).copy()
            out[f"neusrun.{tag}.iter_step"] = np.int64(run.iter_step)

    # ---------------------------------------------------------------- GridEncode.__init__: the level table (grid_encode.py:17-40)
    stub("jnerf.utils.common", enlarge=None)
    ge = load("models/position_encoders/hash_encoder/grid_encode.py", "ref_grid_encode")
    for aabb in pyref_scene.LEVEL_TABLE_AABBS:
        g = ge.GridEncode("", aabb_scale=aabb, n_rays_per_batch=8, MAX_STEP=8)        # tiny scratch buffers; the table does not depend on them
        out[f"levels.{aabb}.offsets"] = npy(g.m_hashmap_offsets_table).astype(np.int64)
        out[f"levels.{aabb}.n_params"] = np.int64(g.m_n_params)
        out[f"levels.{aabb}.per_level_scale"] = np.float64(g.m_per_level_scale)

    # ---------------------------------------------------------------- DensityGridSampler: the Python orchestration around the CUDA ops (which are stubs that record)
    import json
    out["sampler.traces"] = np.frombuffer(json.dumps(sampler_traces()).encode(), dtype=np.uint8)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_pyref_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
