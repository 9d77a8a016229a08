"""Host-side logic that needs no GPU: Config semantics (python/jnerf/utils/config.py), registries, pose convention, camera path, presets."""
import os
import numpy as np
import pytest
from jnerf_amd.utils.config import Config, init_cfg, get_cfg, reset_cfg
from jnerf_amd.utils.registry import Registry, build_from_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_missing_key_is_none_and_base_cover(tmp_path):
    (tmp_path / "base.py").write_text("a = 1\nd = dict(x=1, y=dict(p=1, q=2))\nlst = [1, 2]\n")
    (tmp_path / "child.py").write_text("_base_ = 'base.py'\nb = 2\nd = dict(y=dict(q=3), z=5)\n")
    (tmp_path / "cover.py").write_text("_base_ = 'base.py'\nd = dict(_cover_=True, only=1)\n")
    c = Config(str(tmp_path / "child.py"))
    assert c.a == 1 and c.b == 2 and c.nothing is None                       # config.py:24-27
    assert c.d.x == 1 and c.d.y.p == 1 and c.d.y.q == 3 and c.d.z == 5       # recursive merge (config.py:75-92)
    assert c.name == "child" and c.work_dir == "work_dirs/child"             # config.py:107-110
    k = Config(str(tmp_path / "cover.py"))
    assert dict(k.d) == {"only": 1}                                          # _cover_ replaces (config.py:85-88)
    c.live_object = object()
    assert c["live_object"] is c.live_object                                 # live objects are stuffed into the same dict (runner.py:25)


def test_shipped_configs_load():
    init_cfg(os.path.join(ROOT, "projects/ngp/configs/ngp_fox.py"))
    c = get_cfg()
    assert c.fp16 is True and c.const_dt is False and c.dataset.val is None and c.dataset.train.root_dir == "data/fox"
    assert c.optim.lr == 0.1 and c.expdecay.decay_start == 20000 and c.target_batch_size == 1 << 18 and c.hash_func.startswith("p0 ^ p1")
    init_cfg(os.path.join(ROOT, "projects/ngp/configs/ngp_base.py"))
    assert get_cfg().fp16 is None and get_cfg().const_dt is True and get_cfg().dataset.val.mode == "val"
    reset_cfg(x=1)
    assert get_cfg().x == 1 and get_cfg().fp16 is None


def test_registry_and_build_from_cfg():
    R = Registry()

    @R.register_module()
    class Foo:
        def __init__(self, a, b=2):
            self.a, self.b = a, b
    f = build_from_cfg(dict(type="Foo", a=1), R, b=5)
    assert (f.a, f.b) == (1, 5) and build_from_cfg(None, R) is None
    with pytest.raises(AssertionError):
        R.register_module()(Foo)
    with pytest.raises(TypeError):
        build_from_cfg(dict(type="Foo"), R)
    import jnerf_amd.runner  # noqa: F401  registers everything
    from jnerf_amd.utils import registry as G
    for reg, names in ((G.ENCODERS, ["HashEncoder", "SHEncoder", "FrequencyEncoder"]), (G.NETWORKS, ["NGPNetworks"]), (G.SAMPLERS, ["DensityGridSampler"]),
                       (G.LOSSES, ["HuberLoss", "MSELoss"]), (G.OPTIMS, ["Adam", "ExpDecay", "EMA"]), (G.DATASETS, ["NerfDataset", "SyntheticNerfDataset"])):
        for n in names:
            reg.get(n)


def test_nerf2ngp_pose_convention():
    from jnerf_amd.dataset import _RayDatasetBase
    d = _RayDatasetBase.__new__(_RayDatasetBase)
    d.correct_pose = [1, -1, -1]
    m = np.arange(12, dtype=np.float32).reshape(3, 4)
    out = d.matrix_nerf2ngp(m.copy(), 0.33, [0.5, 0.5, 0.5])
    ref = m.copy(); ref[:, 1] *= -1; ref[:, 2] *= -1; ref[:, 3] = ref[:, 3] * 0.33 + 0.5; ref = ref[[1, 2, 0]]   # dataset.py:255-262
    assert np.array_equal(out, ref)


def test_expdecay_schedule():
    from jnerf_amd.optim import ExpDecay

    class Fake:
        lr = 0.1
        def step(self, loss=None):
            pass
    e = ExpDecay(Fake(), decay_start=3, decay_interval=2, decay_base=0.33)
    lrs = []
    for _ in range(8):
        e.step()
        lrs.append(round(e._nested_optimizer.lr, 6))
    assert lrs == [0.1, 0.1, 0.1, 0.033, 0.033, round(0.1 * 0.33 ** 2, 6), round(0.1 * 0.33 ** 2, 6), round(0.1 * 0.33 ** 3, 6)]   # expdecay.py:20-25


def test_camera_path():
    from jnerf_amd.camera_path import path_spherical
    p = path_spherical(8)
    assert len(p) == 8 and p[0].shape == (3, 4)
    for m in p:
        assert np.allclose(m[:, :3] @ m[:, :3].T, np.eye(3), atol=1e-5) and abs(np.linalg.norm(m[:, 3]) - 4.0) < 1e-4


@pytest.mark.skipif(not os.path.isdir("/root/reference/projects/ngp/configs"), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("name", ["ngp_base.py", "ngp_fox.py", "ngp_comp.py"])
def test_reference_config_files_load_unchanged_and_equal_ours(name):
    """the reference's own config files go through jnerf_amd.utils.config untouched, and projects/ngp/configs/<name> in this repo resolves to the same keys and
    values (ours are written with `_base_` inheritance instead of being copies)"""
    from jnerf_amd.utils.config import Config
    ref = Config(os.path.join("/root/reference/projects/ngp/configs", name)).dump()
    ours = Config(os.path.join(ROOT, "projects", "ngp", "configs", name)).dump()
    for d in (ref, ours):
        for k in ("name", "work_dir", "dataset_dir", "dataset_type"):
            d.pop(k, None)
    if name in ("ngp_fox.py", "ngp_comp.py"):        # keys our configs inherit from ngp_base.py that the reference's file simply leaves unset (all read as None / False there)
        for k in ("load_ckpt", "ckpt_path", "alpha_image"):
            assert not ours.pop(k, None)
    if name == "ngp_fox.py":
        ours["dataset"].pop("val", None)
    assert ours == ref, {k: (ours.get(k), ref.get(k)) for k in set(ours) | set(ref) if ours.get(k) != ref.get(k)}


def test_jnerf_alias_package_resolves_to_this_build():
    """scripts written against the reference's package layout import `jnerf.*` (tools/run_net.py:7-9 there): the alias package maps those names onto jnerf_amd"""
    import jnerf.models  # noqa: F401
    from jnerf.runner import Runner, NeuSRunner
    from jnerf.utils.config import init_cfg, get_cfg as gc2
    from jnerf.utils.registry import build_from_cfg as b2, NETWORKS, SCHEDULERS, DATASETS, OPTIMS, SAMPLERS, LOSSES
    import jnerf_amd.runner, jnerf_amd.utils.config
    assert Runner is jnerf_amd.runner.Runner and gc2 is get_cfg and b2 is build_from_cfg and init_cfg is jnerf_amd.utils.config.init_cfg
    for name in ("NGPNetworks", "OriginNeRFNetworks"):
        assert NETWORKS.get(name) is not None
    assert all(r is not None for r in (SCHEDULERS, DATASETS, OPTIMS, SAMPLERS, LOSSES))
    import jnerf_amd.neus_runner
    assert NeuSRunner is jnerf_amd.neus_runner.NeuSRunner            # (tests/test_neus_cpu.py drives it)
    # the methods the reference's tools/run_net.py and tools/extract_mesh.py call on a Runner (runner.py:62-264), with the reference's leading arguments
    import inspect
    for name, args in (("train", []), ("test", ["load_ckpt"]), ("render", ["load_ckpt", "save_path"]), ("save_ckpt", ["path"]), ("load_ckpt", ["path"]), ("val_img", None),
                       ("render_test", ["save_img", "save_path"]), ("save_img", ["path", "img", "alpha"]), ("render_img", ["dataset_mode", "img_id"]),
                       ("render_img_with_pose", ["pose"])):
        fn = getattr(Runner, name)
        if args is not None:
            assert list(inspect.signature(fn).parameters)[1:1 + len(args)] == args, name


@pytest.mark.skipif(not os.path.isdir("/root/reference/projects/nerf/configs"), reason="reference checkout not present (GPU box)")
def test_nerf_base_config_equals_reference():
    ref = Config("/root/reference/projects/nerf/configs/nerf_base.py").dump()
    ours = Config(os.path.join(ROOT, "projects", "nerf", "configs", "nerf_base.py")).dump()
    for d in (ref, ours):
        for k in ("name", "work_dir"):
            d.pop(k, None)
    assert ours == ref, {k: (ours.get(k), ref.get(k)) for k in set(ours) | set(ref) if ours.get(k) != ref.get(k)}


def test_origin_nerf_network_on_cpu():
    """BASELINE config [0] ("original NeRF ... CPU path - plumbing only"): OriginNeRFNetworks + FrequencyEncoder build from the registry and run forward / backward
    on CPU tensors (plain torch): output [n,4] = (rgb logits, density logit), the skip connection re-injects the encoded position at layer 5"""
    import torch
    from jnerf_amd import networks_ori, encoders  # noqa: F401
    from jnerf_amd.utils.registry import NETWORKS
    reset_cfg(device="cpu", fp16=False, encoder=dict(pos_encoder=dict(type="FrequencyEncoder", multires=10), dir_encoder=dict(type="FrequencyEncoder", multires=4)))
    net = build_from_cfg(dict(type="OriginNeRFNetworks"), NETWORKS)
    assert net.pos_encoder.out_dim == 63 and net.dir_encoder.out_dim == 27
    assert [l.in_features for l in net.pts_linears] == [63, 256, 256, 256, 256, 319, 256, 256] and net.views_linears[0].in_features == 283
    x, d = torch.rand(50, 3), torch.rand(50, 3)
    out = net(x, d)
    assert out.shape == (50, 4) and out.dtype == torch.float32 and net.density(x).shape == (50, 1)
    out.sum().backward()
    assert all(p.grad is not None for p in net.parameters())


def test_constant_step_chain_closed_form_is_bit_identical_to_the_serial_sum():
    """The claim k_march_wave's chain phase rests on (csrc/sampler.hip): inside a binade, t_{k+1} = fl(t_k + dt) with a constant dt adds the same integer
    q = rint(dt / ulp) at every step as long as the exact sum stays below the binade's end (no tie), so lane i can write (a0 + i*q) * ulp directly; the 1-3 steps
    around a binade crossing are real fp32 adds.  Restated in numpy (same decisions as the kernel) and compared bit for bit with the serial recurrence."""
    import math
    f32 = np.float32

    def closed(t0, nc, dt):
        tch = np.full(nc + 1, np.nan, dtype=np.float32)
        k, t = 0, f32(t0)
        while k <= nc:
            _, e = math.frexp(float(t))
            m = 0
            if -100 < e < 100:
                sc = f32(math.ldexp(1.0, 24 - e))
                a0f = f32(t * sc)
                if f32(8388608.0) <= a0f < f32(16777216.0):
                    delta = f32(dt * sc)
                    if delta < f32(16777216.0):
                        a0, q = int(a0f), int(np.rint(delta))
                        x = 16777215 - a0 - int(np.ceil(delta))
                        if f32(delta - np.floor(delta)) != f32(0.5) and x >= 0 and q > 0:
                            m = x // q + 1
            if m == 0:
                tch[k] = t; t = f32(t + dt); k += 1
                continue
            last = min(m, nc - k)
            u = f32(math.ldexp(1.0, e - 24))
            tch[k:k + last + 1] = (np.arange(last + 1, dtype=np.int64) * q + a0).astype(np.float32) * u
            t = tch[k + last]; k += last
            if k == nc:
                break
        return tch

    rng = np.random.default_rng(0)
    steps = [f32(f32(math.sqrt(3.0) / 1024.0) * f32(0.5)), f32(0.001953125), f32(0.0009765625 * 1.5), f32(1e-3), f32(3e-8)]     # the marcher's constant step, ties, tiny steps
    for dt in steps:
        for trial in range(200):
            t0 = f32(rng.uniform(*[(3.9, 4.0), (0.05, 0.3), (1.7, 2.0), (0.1, 8.0)][trial % 4]))      # around binade ends and anywhere
            serial = np.empty(257, dtype=np.float32)
            t = t0
            for k in range(257):
                serial[k] = t; t = f32(t + dt)
            assert np.array_equal(closed(t0, 256, dt), serial), (float(dt), float(t0))


def test_parallel_walk_guess_is_accepted_only_when_it_is_the_orbit():
    """The claim k_march_wave's walk rests on (csrc/sampler.hip): candidates c of a round continue at nx[c] > c; the visited ones are the orbit of the start s.
    G[c] = (c == s) or (c > s and max(nx[s..c-1]) == c) is computed with a prefix maximum; it is ACCEPTED only if every member of G before the first member
    outside the box continues at the next member of G.  Property checked on random link arrays (runs of equal targets with fuzz, like candidates of one empty
    cell): whenever the check accepts, G (up to that first outside member) equals the serial orbit; and a plain successor chain is always accepted."""
    rng = np.random.default_rng(7)
    NC = 256
    accepted = rejected = 0
    for trial in range(3000):
        # links: blocks of candidates that jump to (about) the block's end, or step to their successor
        nx = np.arange(1, NC + 1)
        c = 0
        while c < NC:
            ln = int(rng.integers(1, 14))
            if rng.random() < 0.6:                                     # an "empty cell": everyone lands just past it, with occasional one-off fuzz
                land = min(NC, c + ln)
                for p in range(c, min(NC, c + ln)):
                    nx[p] = min(NC, max(p + 1, land + (int(rng.integers(-1, 2)) if rng.random() < 0.004 else 0)))
            c += ln
        inside = np.ones(NC, bool)
        first_out = int(rng.integers(NC // 2, NC + 40))
        inside[first_out:] = False
        nx[~inside] = np.arange(1, NC + 1)[~inside]
        s = int(rng.integers(0, 20))
        orbit, v = [], s
        while v < NC:
            orbit.append(v)
            if not inside[v]:
                break
            v = int(nx[v])
        pm = np.maximum.accumulate(np.where(np.arange(NC) >= s, nx, 0))
        G = [c for c in range(s, NC) if c == s or pm[c - 1] == c]
        fo = next((c for c in G if not inside[c]), NC)
        ok = True
        for k, c in enumerate(G):
            if c >= fo or not inside[c]:
                break
            ng = G[k + 1] if k + 1 < len(G) else NC
            if ng != nx[c]:
                ok = False
                break
        if ok:
            accepted += 1
            assert [c for c in G if c <= fo] == orbit, (trial, s)
        else:
            rejected += 1
    assert accepted > 500 and rejected > 100, (accepted, rejected)       # both branches were exercised
    nx = np.arange(1, NC + 1)                                           # every candidate occupied: G is everything, accepted
    pm = np.maximum.accumulate(nx)
    assert all(pm[c - 1] == c for c in range(1, NC))


def test_jittor_pickle_container_round_trip(tmp_path):
    """the container jt.save / jt.load use for .pkl files (pickle protocol 4 of numpy arrays + sha1 + b'HCAJSLHD'), restated from Jittor's published
    source and read / written WITHOUT Jittor: round trip, trailer, corruption detection, a bare pickle without trailer, and refusal to import code."""
    import hashlib, pickle
    import torch
    from jnerf_amd.utils import jittor_pickle as JP
    ck = {"global_step": 123, "model": {"pos_encoder.m_grid": torch.arange(12, dtype=torch.float32), "density_mlp.con_weights": torch.ones(4).half()},
          "nested_optimizer": {"defaults": {"lr": 0.1, "param_groups": [{"values": [torch.zeros(3)], "m": [torch.ones(3)]}]}}}
    p = tmp_path / "params.pkl"
    JP.dump(ck, str(p))
    raw = p.read_bytes()
    assert raw.endswith(b"HCAJSLHD") and hashlib.sha1(raw[:-28]).digest() == raw[-28:-8]
    plain = pickle.loads(raw[:-28])                                    # what jt.load hands to the reference's Runner: numpy arrays and Python scalars only
    assert isinstance(plain["model"]["pos_encoder.m_grid"], np.ndarray) and plain["global_step"] == 123
    back = JP.to_torch(JP.load(str(p)))
    assert torch.equal(back["model"]["pos_encoder.m_grid"], ck["model"]["pos_encoder.m_grid"])
    assert back["model"]["density_mlp.con_weights"].dtype == torch.float32          # fp16 payloads become fp32 masters
    assert torch.equal(back["nested_optimizer"]["defaults"]["param_groups"][0]["m"][0], torch.ones(3)) and back["nested_optimizer"]["defaults"]["lr"] == 0.1
    bad = bytearray(raw); bad[10] ^= 0xFF
    with pytest.raises(ValueError, match="checksum"):
        JP.loads(bytes(bad))
    assert JP.loads(pickle.dumps({"a": np.arange(3)}, 4))["a"].tolist() == [0, 1, 2]     # no trailer: unpickled as is (safeunpickle's fallback)
    import os as _os
    evil = pickle.dumps(_os.getcwd)                                    # a pickle that imports something other than numpy must be refused
    with pytest.raises(pickle.UnpicklingError):
        JP.loads(evil)


def test_fixed_point_conversion_equals_round_half_even():
    """(r4) `fixed_rn` (csrc/hash_encode.hip): contribution * 2^k -> nearest 64-bit integer, ties to even, in seven instructions instead of __float2ll_rn's generic expansion.
    The accumulate kernel's exactness rests on it; the same function compiled for the host is held to Python's exact arithmetic here (c * 2^k is exact in a double)."""
    import ctypes as C
    from jnerf_amd import _lib
    f = _lib.lib().ngp_x_fixed_rn
    f.restype, f.argtypes = C.c_longlong, [C.c_float, C.c_float]
    rng = np.random.default_rng(0)
    for k in (0, 1, 20, 31, 32, 33, 38, 45, 60):
        s = float(2.0 ** k)
        c = np.concatenate([rng.standard_normal(4000).astype(np.float32) * np.float32(10.0 ** rng.integers(-12, 2)),
                            (rng.integers(-2 ** 22, 2 ** 22, 2000).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -k),     # exact ties
                            np.array([0.0, -0.0, 1.0, -1.0, 2.0 ** -40, -(2.0 ** -40)], np.float32)])
        for v in c:
            t = float(v) * s
            if abs(t) >= 2.0 ** 62:
                continue
            assert f(float(v), s) == round(t), (float(v), k, f(float(v), s), round(t))


@pytest.mark.parametrize("aabb_scale", [1, 2, 4])
def test_edge_records_x_neighbours_share_a_4096_entry_bin(aabb_scale):
    """(r4) the invariant the edge records rest on: index = (x ^ y*P1 ^ z*P2) & (2^19 - 1) and x + 1 <= res <= 2048 touches bits 0..11 only, so both x-neighbours of any
    cell edge of such a level fall into the same 4096-entry slice - for every level the kernel routes that way (res <= 2048), every x, random y / z"""
    from jnerf_amd import ops
    t, _, _ = ops.level_table(aabb_scale)
    rng = np.random.default_rng(1)
    for size, res in ((int(r[1]), int(r[2])) for r in t):
        if size != 1 << 19 or res > 2048:
            continue
        x = np.arange(0, res, dtype=np.uint32)                     # cell corner 0 .. res - 1, neighbour x + 1 <= res
        for _ in range(8):
            y, z = rng.integers(0, res + 1, 2, dtype=np.uint32)
            h = np.uint32((int(y) * 19349663) & 0xffffffff) ^ np.uint32((int(z) * 83492791) & 0xffffffff)
            i0, i1 = (x ^ h) & np.uint32(size - 1), ((x + np.uint32(1)) ^ h) & np.uint32(size - 1)
            assert np.array_equal(i0 >> 12, i1 >> 12), (res, int(y), int(z))
    fine = [int(r[2]) for r in t if int(r[1]) == 1 << 19 and int(r[2]) > 2048]
    assert (len(fine) > 0) == (aabb_scale > 1)                     # aabb_scale 1 (ngp_base.py): every hashed level pairs; beyond: the finest levels emit single records


def test_hash_backward_workspace_size_is_monotonic_and_bounded():
    """ngp_hash_bwd_workspace_bytes: what the caller allocates once for the whole run: grows with n, both level tables, stays far below the 288 GB of the device"""
    from jnerf_amd import ops
    for aabb_scale in (1, 4):
        t, _, _ = ops.level_table(aabb_scale)
        sizes = [ops.hash_bwd_workspace_bytes(t, n) for n in (1024, 1 << 14, 1 << 18, 1 << 20)]
        assert all(a < b for a, b in zip(sizes, sizes[1:])), sizes
        assert sizes[2] < 3 * 2 ** 30 and sizes[3] < 12 * 2 ** 30, sizes
        # (r5, VERDICT r4 #12 / ADVICE r4) sized for the path the dtypes take, not for the sum of both designs: the fp32 configuration's record regions need well under
        # a gigabyte at the training batch (was ~2.3 GB), and the dtype-agnostic figure is the larger of the two
        import torch
        f32, f16 = ops.hash_bwd_workspace_bytes(t, 1 << 18, torch.float32), ops.hash_bwd_workspace_bytes(t, 1 << 18, torch.float16)
        assert f32 < (0.65 if aabb_scale == 1 else 0.85) * 2 ** 30 and f16 < 1.9 * 2 ** 30 and sizes[2] == max(f32, f16), (f32, f16, sizes[2])


def test_edge_record_multiplication_order_carries_the_references_error_bound():
    """(r4) an edge record carries a = g * (wy * wz) and the accumulate kernel forms a * (1 - fx) | a * fx, where the reference forms g * ((wx * wy) * wz)
    (HashEncode.h:299-396): three fp32 roundings either way.  Both must lie within (1 + 2^-24)^3 - 1 = 1.8e-7 of the exact product, and they are never
    more than 4 ulp apart - 2 M random cases per corner, in numpy's fp32 against fp64."""
    rng = np.random.default_rng(7)
    n = 1 << 21
    f = rng.random((n, 3), dtype=np.float32)
    g = (rng.standard_normal(n) * 10.0 ** rng.integers(-8, 0, n)).astype(np.float32)
    bound = (1 + 2.0 ** -24) ** 3 - 1
    for xb in (0, 1):
        wx = f[:, 0] if xb else np.float32(1) - f[:, 0]
        for yb in (0, 1):
            wy = f[:, 1] if yb else np.float32(1) - f[:, 1]
            for zb in (0, 1):
                wz = f[:, 2] if zb else np.float32(1) - f[:, 2]
                ref = g * ((wx * wy) * wz)                      # the reference's order
                ours = (g * (wy * wz)) * wx                     # record, then the accumulate kernel's factor
                exact = g.astype(np.float64) * wx.astype(np.float64) * wy.astype(np.float64) * wz.astype(np.float64)
                m = np.abs(exact) > 1e-30                       # (denormal results aside)
                assert (np.abs(ours[m] - exact[m]) <= bound * np.abs(exact[m]) * (1 + 1e-9)).all() and (np.abs(ref[m] - exact[m]) <= bound * np.abs(exact[m]) * (1 + 1e-9)).all()
                ulp = np.spacing(np.abs(ref[m])).astype(np.float64)
                assert (np.abs(ours[m].astype(np.float64) - ref[m].astype(np.float64)) <= 4.0 * ulp).all()


@pytest.mark.parametrize("aabb_scale", [1, 4])
def test_region_path_bin_mapping_is_a_bijection_on_every_level(aabb_scale):
    """(r4) entry -> (bin, slot) -> entry of the region kernels / k_bin_accumulate2 (csrc/hash_encode.hip: bin2_of, local2_of, entry2_of), compiled for the host: every entry of every
    level maps to a slot inside the range its bin's accumulate workgroup zeroes and writes out, no two entries share a slot, and the way back is exact - small (dense) levels are dealt to
    the 128 bins in interleaved groups of eight, full 2^19-entry levels in 4096-entry slices"""
    import ctypes as C
    from jnerf_amd import _lib, ops
    f = _lib.lib().ngp_x_bin2_map
    f.restype, f.argtypes = None, [C.c_uint32, C.c_uint32, C.c_void_p]
    out = (C.c_uint32 * 4)()
    t, _, _ = ops.level_table(aabb_scale)
    for size in sorted({int(r[1]) for r in t}):
        step = 1 if size <= 80000 else 37                          # the two largest dense levels and the hashed ones: a stride of entries (plus both ends)
        seen = set()
        for e in list(range(0, size, step)) + [size - 1]:
            f(size, e, out)
            b, l, back, n_local = int(out[0]), int(out[1]), int(out[2]), int(out[3])
            assert back == e and b < 128 and l < n_local <= 4096, (size, e, b, l, back, n_local)
            assert (b, l) not in seen or e == size - 1
            seen.add((b, l))


@pytest.mark.parametrize("scene", ["Scar", "Car", "Scarf"])
def test_ngp_comp_config_builds_its_datasets(scene, tmp_path):
    """projects/ngp/configs/ngp_comp.py (the reference's competition config, projects/ngp/configs/ngp_comp.py:41-100 there) goes through init_cfg and the DATASETS
    registry: per-scene aabb_scale / scale / offset, correct_pose = [-1, -1, 1], a test split WITHOUT images at 800 x 800, white background, fp16"""
    import json
    import shutil
    from PIL import Image
    from jnerf_amd import dataset as _ds  # noqa: F401  (registers NerfDataset)
    from jnerf_amd.utils.config import init_cfg
    from jnerf_amd.utils.registry import DATASETS
    cdir = tmp_path / "configs"
    cdir.mkdir()
    shutil.copy(os.path.join(ROOT, "projects", "ngp", "configs", "ngp_base.py"), cdir / "ngp_base.py")
    src = open(os.path.join(ROOT, "projects", "ngp", "configs", "ngp_comp.py")).read()
    data = tmp_path / "data"
    data.mkdir()
    src = src.replace('exp_name = "Scar"', f'exp_name = "{scene}"').replace("dataset_dir = 'my/data/' + exp_name", f"dataset_dir = {str(data)!r}")
    assert f'exp_name = "{scene}"' in src and str(data) in src
    open(cdir / "ngp_comp.py", "w").write(src)
    rng = np.random.default_rng(1)
    W, H = 16, 10

    def frames(prefix, n, with_files=True):
        out = []
        for i in range(n):
            if with_files:
                Image.fromarray(rng.integers(0, 255, (H, W, 4), dtype=np.uint8)).save(data / f"{prefix}_{i}.png")
            m = np.eye(4); m[:3, :3] = rng.normal(size=(3, 3)); m[:3, 3] = rng.normal(size=3)
            out.append({"file_path": f"./{prefix}_{i}", "transform_matrix": m.tolist()})
        return out
    tr = frames("train", 3)
    json.dump({"camera_angle_x": 0.6, "frames": tr}, open(data / "transforms_train.json", "w"))
    json.dump({"camera_angle_x": 0.6, "frames": frames("val", 11)}, open(data / "transforms_val.json", "w"))
    te = frames("test", 2, with_files=False)
    json.dump({"camera_angle_x": 0.6, "frames": te}, open(data / "transforms_test.json", "w"))
    init_cfg(str(cdir / "ngp_comp.py"))
    cfg = get_cfg()
    cfg.device = "cpu"
    assert cfg.fp16 is True and cfg.const_dt is True and cfg.background_color == [1, 1, 1] and cfg.tot_train_steps == 40000 and cfg.exp_name == scene
    want_aabb = {"Scar": 5, "Car": 4, "Scarf": 8}[scene]
    want_scale = 0.05 if scene == "Scarf" else 0.33
    want_off = np.array([-2.0, -0.5, 0.0] if scene == "Car" else [0.5, 0.5, 0.5], np.float32)
    train = build_from_cfg(cfg.dataset.train, DATASETS)
    assert train.n_images == 3 + 11 and train.aabb_scale == want_aabb and train.aabb_range == (0.5 - want_aabb / 2, 0.5 + want_aabb / 2)       # train includes val (dataset.py:77)
    # dataset.py:255-262 with correct_pose = [-1, -1, 1]: columns 0 and 1 negated, translation * scale + offset, rows cycled [1, 2, 0]
    by_t = {}
    for fr in tr:
        m = np.array(fr["transform_matrix"], np.float32)[:3]
        m[:, 0] *= -1; m[:, 1] *= -1
        m[:, 3] = m[:, 3] * np.float32(want_scale) + want_off
        by_t[tuple(np.round(m[[1, 2, 0]][:, 3], 5))] = m[[1, 2, 0]]
    ours = train.transforms_gpu.numpy().transpose(0, 2, 1)                  # [n, 3, 4]
    hits = 0
    for x in ours:
        k = tuple(np.round(x[:, 3], 5))
        if k in by_t:
            np.testing.assert_allclose(x, by_t[k], atol=1e-6); hits += 1
    assert hits == 3
    val = build_from_cfg(cfg.dataset.val, DATASETS)
    assert val.n_images == 2                                                # every 10th validation frame (dataset.py:93)
    test = build_from_cfg(cfg.dataset.test, DATASETS)
    assert test.have_img is False and test.n_images == 2 and test.resolution == [800, 800] and test.image_data.shape == (2, 800 * 800, 4) and float(test.image_data.abs().max()) == 0.0
    get_cfg().clear()


def test_hash_backward_routing_by_level_table_and_dtype():
    """(r5) csrc/hash_encode.hip: hash_bwd_path - which of the three scatters a workspace call takes is decided by the level table and the dtypes alone (that is what lets the
    workspace be sized for the path): every table GridEncode builds -> record regions for fp32 dL/dy, per-corner lists for fp16; a level beyond 2^19 entries or a hashed
    table that is not a power of two -> the reference's global float atomics (the one fallback), whatever the dtypes."""
    import ctypes as C
    from jnerf_amd import _lib, ops
    lib = _lib.lib()
    lib.ngp_x_hash_bwd_path.restype = C.c_int
    F32, F16 = _lib.F32, _lib.F16
    path = lambda t, a, b: lib.ngp_x_hash_bwd_path(np.ascontiguousarray(t, dtype=np.uint32).ctypes.data_as(C.c_void_p), a, b)
    for aabb in (1, 2, 4, 5, 8, 16):
        t, _, _ = ops.level_table(aabb)
        assert path(t, F32, F32) == 2 and path(t, F16, F32) == 1 and path(t, F16, F16) == 1, aabb
        assert ops.hash_bwd_workspace_bytes(t, 1 << 16, torch_dtype(F32)) < ops.hash_bwd_workspace_bytes(t, 1 << 16, torch_dtype(F16))
    # aabb_scale >= 32: the finest level has resolution >= 2^16, the reference's uint32 stride loop (HashEncode.h:82-91) overflows on it and indexes it as "dense" - neither
    # a run level nor edge-capable, so the whole call takes the per-corner lists (exact, just not the region path)
    for aabb in (32, 128):
        t, _, _ = ops.level_table(aabb)
        assert path(t, F32, F32) == 1, aabb
    t, _, _ = ops.level_table(1)
    big = np.array(t, dtype=np.uint32).copy().reshape(16, 4)
    big[15, 1] = 1 << 20                                             # a level of 2^20 entries: the bins end at 2^19
    assert path(big, F32, F32) == 0 and path(big, F16, F32) == 0
    odd = np.array(t, dtype=np.uint32).copy().reshape(16, 4)
    odd[15, 1] = (1 << 19) - 8                                       # hashed (res^3 > size) but not a power of two: the record kernels' mask arithmetic does not apply
    assert path(odd, F32, F32) == 0
    assert ops.hash_bwd_workspace_bytes(big, 1 << 16, torch_dtype(F32)) == 0       # the fallback needs no workspace


def torch_dtype(code):
    import torch
    from jnerf_amd import _lib
    return torch.float16 if code == _lib.F16 else torch.float32
