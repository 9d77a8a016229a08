#!/usr/bin/env python
"""The reference's OWN training loop, end to end, on the CPU of the build container:

    python tests/golden/make_golden_refrun.py [lego|cone]     ->  tests/golden/golden_refrun_v1.npz | golden_refrun_cone_v1.npz          (a few minutes each)

What runs is the reference's unmodified Python package (`import jnerf` from /root/reference/python: Runner, NerfDataset, NGPNetworks, HashEncoder, SHEncoder,
DensityGridSampler and its eight op wrappers, HuberLoss, Adam / ExpDecay / EMA, utils.config with projects/ngp/configs/ngp_base.py) over
  * oracle/jt_shim - torch primitives for the Jittor calls those modules make (Jittor is not installable here), and
  * oracle/_ref    - the reference's own kernel headers compiled for the host, to which the stand-in's jt.code binds every launch the wrappers' CUDA sources make.
Nothing of the reference is copied or edited.  What is NOT the reference: Jittor's primitives incl. nn.Adam (restated in the stand-in), the initialisation draws (supplied
by seeded generators so that the consumer can redraw them), and three instance-level switches below (no checkpoint file, no test-set render, no JIT).

The fixture holds, for every iteration: which pixels formed the batch, the loss, the sample counts, the adaptive ray count; the occupancy statistics of every refresh;
the initial MLP weights; digests of the final parameters.  tests/test_refrun_golden.py replays the same iterations through the C oracle (oracle/ngp_oracle.c) - the
restatement every HIP kernel is held to - and compares."""
import os
import sys
if __debug__:                                                  # the reference asserts `var.dtype == 'float32'` (a Jittor dtype equals its name; a torch dtype does not): run without asserts
    os.execv(sys.executable, [sys.executable, "-O"] + sys.argv)
import importlib.util
import tempfile
import time
import types
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]          # /root/repo holds an alias package called jnerf: keep it off the path
sys.path.insert(0, os.path.join(ROOT, "oracle", "jt_shim"))
sys.path.insert(0, "/root/reference/python")
for _name in ("cv2", "mcubes", "trimesh", "open3d", "jittor_utils"):           # imported at module level by files of the package, never called on this path
    sys.modules[_name] = types.ModuleType(_name)


def _imread(path):
    from PIL import Image
    return np.asarray(Image.open(path))


sys.modules["imageio"] = types.ModuleType("imageio")
sys.modules["imageio"].imread = _imread
import torch                                                  # noqa: E402
import jittor as jt                                           # noqa: E402  (the stand-in)
from jittor import _code                                      # noqa: E402

_spec = importlib.util.spec_from_file_location("pyref_scene", os.path.join(ROOT, "tests", "golden", "pyref_scene.py"))
S = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(S)


jt.enable_jittor_shapes()


def main(case):
    import jnerf                                               # the reference package itself
    if not jnerf.__file__.startswith("/root/reference/"):
        raise RuntimeError("not the reference package: " + jnerf.__file__)
    from jnerf.utils.config import init_cfg, get_cfg
    from jnerf.runner import Runner
    R = dict(S.REFRUN, **S.REFRUN_CASES[case])
    if os.environ.get("REFRUN_STEPS"):                         # (debugging aid; the committed fixtures are made without it)
        R["steps"] = int(os.environ["REFRUN_STEPS"])
    out = {}
    perms, batches, bgs = [], [], []
    # ---- supplied randomness, redrawable by the consumer
    def randperm(n):                                           # draw number k (from 1) of the run: seed perm + k
        perms.append(int(n))
        return torch.randperm(int(n), generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["perm"] + len(perms)))

    def random(shape, dtype="float32"):                        # the k-th background batch (runner.py:65, one per iteration): seed bg + k
        bgs.append([int(v) for v in shape])
        return torch.rand(bgs[-1], generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["bg"] + len(bgs)))
    jt.randperm, jt.random = randperm, random
    jt.init.uniform = lambda shape, dtype="float32", low=0.0, high=1.0: (
        torch.rand([int(v) for v in shape], generator=torch.Generator().manual_seed(S.REFRUN_SEEDS["grid"])) * (high - low) + low)
    jt.save = lambda obj, path: out.setdefault("ckpt_keys", np.frombuffer(",".join(sorted(obj)).encode(), np.uint8))       # runner.py:123-131's dictionary, not written

    d = tempfile.mkdtemp(prefix="refrun_")
    S.write_rendered_nerf_dataset(d, R.get("res"))
    init_cfg("/root/reference/projects/ngp/configs/ngp_base.py")
    cfg = get_cfg()
    for mode in ("train", "val", "test"):
        cfg.dataset[mode].root_dir = d
        cfg.dataset[mode].batch_size = R["n_rays_per_batch"]
        if R["aabb_scale"] is not None:
            cfg.dataset[mode].aabb_scale = R["aabb_scale"]
    cfg.const_dt = R["const_dt"]
    cfg.n_rays_per_batch, cfg.target_batch_size, cfg.tot_train_steps = R["n_rays_per_batch"], R["target_batch_size"], R["steps"]
    cfg.log_dir, cfg.exp_name = os.path.join(d, "logs"), "refrun"
    torch.manual_seed(S.REFRUN_SEEDS["mlp"])                   # the stand-in's nn.Linear draws its weights from the global generator
    _code.reset_rng(1337)
    r = Runner()
    r.test = lambda *a, **k: None                              # train() ends with a render of the test set (runner.py:84): not part of this fixture
    lins = [r.model.density_mlp[0], r.model.density_mlp[2], r.model.rgb_mlp[0], r.model.rgb_mlp[2], r.model.rgb_mlp[4]]
    for i, lin in enumerate(lins):
        out[f"init.W{i}"] = lin.weight.detach().numpy().copy()
    ds = r.dataset["train"]
    out["dataset.n_images"], out["dataset.perm_sizes_at_start"] = np.int64(ds.n_images), np.asarray(perms, np.int64)
    out["dataset.transforms_gpu"] = ds.transforms_gpu.numpy().copy()        # the reference walks the directory in file-system order: this is how the consumer finds its frame order

    class Recording(type(ds)):
        def __next__(self):
            if self.idx_now + self.batch_size >= self.shuffle_index.shape[0]:
                start, perm_id = 0, len(perms) + 1             # __next__ will draw a fresh permutation (dataset.py:58-62)
            else:
                start, perm_id = self.idx_now, self._perm_id
            self._perm_id = perm_id
            batches.append([perm_id, start, int(self.batch_size)])
            return super().__next__()
    ds.__class__ = Recording
    ds._perm_id = 1                                            # the train set's first permutation is draw 1 (the val set's is draw 2)

    log = []
    huber = r.loss_func.execute

    def recording_loss(x, target):
        loss = huber(x, target)
        s = r.sampler
        k = int(min(int(s._rays_numsteps_compacted[:, 0].sum()), R["target_batch_size"]))
        log.append([float(loss.detach().double().mean()), float(loss.detach().double().sum()), x.shape[0], k, int(s._rays_numsteps[:, 0].sum())])
        return loss
    r.loss_func.execute = recording_loss
    refresh, bitfields = [], []
    upd = r.sampler.update_density_grid

    def recording_refresh():
        upd()
        s = r.sampler
        bitfields.append(s.density_grid_bitfield.numpy().astype(np.uint8).copy())     # (r4) what the reference marches through until the next refresh: lets a replay teacher-force it
        refresh.append([int(cfg.m_training_step), float(s.density_grid_mean.reshape(-1)[0]), int(np.unpackbits(s.density_grid_bitfield.numpy()).sum()),
                        float((s.density_grid > 0).sum()), float(s.density_grid.double().clamp_min(0).sum())])
    r.sampler.update_density_grid = recording_refresh
    rays = []
    ubr = r.sampler.update_batch_rays

    def recording_rays():
        measured = int(r.sampler.measured_batch_size.item())
        for j, lin in enumerate(lins):                         # sample() calls this at step 15 BEFORE that step's forward: the weights after 15 updates
            out[f"mid.W{j}"] = lin.weight.detach().numpy().copy()
        ubr()
        rays.append([int(cfg.m_training_step), measured, int(r.sampler.n_rays_per_batch)])
    r.sampler.update_batch_rays = recording_rays

    if R["steps"] == 0:
        # ---- the inference path (runner.py:197-264) on the freshly initialised model: one occupancy refresh (an empty bitfield renders nothing), then a test view and a
        #      free pose, chunked / padded / assembled by the reference's own render_img and render_img_with_pose
        from jnerf.utils.registry import build_from_cfg, DATASETS
        cfg.m_training_step = 0
        r.sampler.update_density_grid()
        r.dataset["test"] = build_from_cfg(cfg.dataset.test, DATASETS)
        out["test.transforms_gpu"] = r.dataset["test"].transforms_gpu.numpy().copy()
        with jt.no_grad():
            img, _, tar = r.render_img("test", 0)
            out["render.img"], out["render.target"] = np.asarray(img, np.float64), np.asarray(tar, np.float64)
            r.alpha_image = True
            img_a, alpha, _ = r.render_img("test", 0)
            out["render.img_alpha"], out["render.alpha"] = np.asarray(img_a, np.float64), np.asarray(alpha, np.float64)
            r.alpha_image = False
            out["render.pose"] = np.asarray(S.NOVEL_POSE_NGP, np.float32)
            out["render.img_pose"] = np.asarray(r.render_img_with_pose(torch.tensor(S.NOVEL_POSE_NGP)), np.float64)
        out["refresh"] = np.asarray(refresh, np.float64)
        out["launches"] = np.frombuffer(";".join(f"{n}:{c}" for n, c in _code.CALLS).encode(), np.uint8)
        out["final.rng_state"] = _code.RNG.st.copy()
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), R["file"])
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
        return
    t0 = time.time()
    r.train()
    print("trained", R["steps"], "iterations in", round(time.time() - t0, 1), "s; losses", np.round([v[0] for v in log], 5))
    out["log"] = np.asarray(log, np.float64)                    # per iteration: loss mean, loss sum, rays, samples trained on, samples marched
    out["batches"] = np.asarray(batches, np.int64)              # per iteration: permutation draw, first slot, ray count
    out["perm_sizes"] = np.asarray(perms, np.int64)
    out["bg_shapes"] = np.asarray(bgs, np.int64)
    out["refresh"] = np.asarray(refresh, np.float64)            # per refresh: step, grid mean, bits set, cells > 0, sum of max(grid, 0)
    out["refresh.bitfield"] = np.stack(bitfields)               # per refresh: the occupancy bitfield (5 * 128^3 / 8 bytes, morton order)
    out["ray_updates"] = np.asarray(rays, np.int64)             # per update: step, measured samples over 16 iterations, new ray count
    out["launches"] = np.frombuffer(";".join(f"{n}:{c}" for n, c in _code.CALLS).encode(), np.uint8)
    for i, lin in enumerate(lins):
        out[f"final.W{i}"] = lin.weight.detach().numpy().copy()
    grid = r.model.pos_encoder.m_grid.detach().numpy()
    probe = np.random.default_rng(S.REFRUN_SEEDS["probe"]).integers(0, grid.size, size=8192)
    out["final.grid_probe"] = grid[probe].copy()
    offs = r.model.pos_encoder.encoder.m_hashmap_offsets_table.numpy().astype(np.int64) * 2
    out["final.grid_level_sums"] = np.asarray([[grid[offs[l]:offs[l + 1]].astype(np.float64).sum(), np.abs(grid[offs[l]:offs[l + 1]].astype(np.float64)).sum()] for l in range(16)])
    out["final.rng_state"] = _code.RNG.st.copy()
    out["final.ema_steps"] = np.int64(r.ema_optimizer.steps)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), R["file"])
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "lego")             # one run per process (the reference keeps global state: config, registries, the pcg32 stream)
