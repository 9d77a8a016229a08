"""-m gpu: end-to-end training through the module API on a procedural scene — loss falls, PSNR rises, the fused (Adam+EMA) optimiser
path matches the un-fused reference formulation, checkpoint round trip."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _runner(**kw):
    from jnerf_amd.presets import ngp_cfg
    from jnerf_amd.runner import Runner
    torch.manual_seed(0)
    ngp_cfg(n_images=8, W=96, H=96, target_batch_size=1 << 16, n_rays_per_batch=1024, **kw)
    return Runner()


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.parametrize("fp16,aabb_scale,const_dt,min_psnr,max_loss_ratio", [(True, 1, True, 30.0, 0.5), (False, 1, True, 28.0, 0.5), (True, 4, False, 17.0, 0.7)])
def test_training_converges(fp16, aabb_scale, const_dt, min_psnr, max_loss_ratio, tmp_path):
    # (fused fp16-MFMA path | fp32 path the reference's ngp_base.py takes | fox-style aabb 4 + cone stepping, which carves 8 tiny views slowly in any precision)
    r = _runner(fp16=fp16, aabb_scale=aabb_scale, const_dt=const_dt, log_dir=str(tmp_path))
    from jnerf_amd.utils.registry import build_from_cfg, DATASETS
    losses = []
    for i in range(400):
        l = r.train_step(i)
        if i % 50 == 0:
            losses.append(float(l.mean().item()))
    assert np.isfinite(losses).all() and losses[-1] < max_loss_ratio * losses[0], losses
    r.dataset["test"] = build_from_cfg(r.cfg.dataset.test, DATASETS)
    img, _, tar = r.render_img("test", 0)
    psnr = -10 * np.log10(np.mean((img - tar) ** 2))
    img_tr, _, tar_tr = r.render_img("train", 0)
    psnr_tr = -10 * np.log10(np.mean((img_tr - tar_tr) ** 2))
    print(f"fp16={fp16} aabb={aabb_scale} const_dt={const_dt}: loss {losses[0]:.4f} -> {losses[-1]:.4f}, PSNR train view {psnr_tr:.1f} dB, held-out view {psnr:.1f} dB (8 images of 96x96, 400 steps)")
    assert psnr_tr > min_psnr and psnr > 14.0, (psnr_tr, psnr)
    r.drain()
    assert r.sampler.n_ray_count_updates == 400 // 16 and r.sampler.n_rays_per_batch % 128 == 0          # update_batch_rays adapted the ray count every 16 steps
    # checkpoint round trip (runner.py:123-151 keys)
    p = str(tmp_path / "params.pkl")
    r.save_ckpt(p)
    ck = torch.load(p, map_location="cpu", weights_only=False)
    assert set(["global_step", "model", "sampler", "optimizer", "nested_optimizer", "ema_optimizer"]) <= set(ck)
    before = r.model.pos_encoder.m_grid.detach().clone()
    r.model.pos_encoder.m_grid.data.zero_()
    r.load_ckpt(p)
    assert torch.equal(r.model.pos_encoder.m_grid.detach(), before)
    # the same through the reference's own container (jt.save: pickle of numpy arrays + sha1 + magic, read and written without Jittor - utils/jittor_pickle.py)
    from jnerf_amd.utils import jittor_pickle
    pj = str(tmp_path / "params_jittor.pkl")
    r.cfg.ckpt_format = "jittor"
    r.save_ckpt(pj)
    r.cfg.ckpt_format = None
    plain = jittor_pickle.load(pj)
    assert set(plain) == {"global_step", "model", "sampler", "optimizer", "nested_optimizer", "ema_optimizer"}          # exactly runner.py:124-131's keys
    assert isinstance(plain["model"]["pos_encoder.m_grid"], np.ndarray) and isinstance(plain["nested_optimizer"]["defaults"]["param_groups"][0]["m"][0], np.ndarray)
    m_before = [t.clone() for t in r.optimizer._nested_optimizer.param_groups[0]["m"]]
    r.model.pos_encoder.m_grid.data.zero_()
    for t in r.optimizer._nested_optimizer.param_groups[0]["m"]:
        t.zero_()
    r.load_ckpt(pj)                                   # recognised by its trailer
    assert torch.equal(r.model.pos_encoder.m_grid.detach(), before)
    assert all(torch.equal(a, b) for a, b in zip(r.optimizer._nested_optimizer.param_groups[0]["m"], m_before))
    assert np.isfinite(float(r.train_step(400).mean().item()))           # and training goes on from it
    r.drain()


def test_module_api_standalone():
    """HashEncoder / SHEncoder / NGPNetworks used on their own, like the reference's modules (autograd through the kernels)"""
    r = _runner()
    enc = r.model.pos_encoder
    x = torch.rand((1000, 3), device="cuda")
    y = enc(x)
    assert y.shape == (1000, 32) and y.dtype == torch.float16
    enc.m_grid.grad = None
    y.float().sum().backward()
    assert enc.m_grid.grad is not None and enc.m_grid.grad.abs().sum() > 0
    # sum of all trilinear weights is 1 per (sample, level, feature): the gradient mass equals the number of outputs
    assert abs(enc.m_grid.grad.sum().item() - 1000 * 32) < 1.0
    d = r.model.dir_encoder(torch.rand((10, 3), device="cuda"))
    assert d.shape == (10, 16)
    out = r.model(x, torch.rand((1000, 3), device="cuda"))
    assert out.shape == (1000, 4)
    assert r.model.density(x).shape == (1000, 1)


def test_render_matches_cpu_fp32_oracle_on_identical_rays(tmp_path):
    """north_star's end-to-end gate: the SAME rays and the SAME trained weights rendered by (a) the HIP path (fp16 gather + fp16-MFMA fused MLP +
    HIP compositing) and (b) the CPU oracle in fp32 (hash encode, SH, both MLPs, compositing restated from the reference) must give the same image.
    Tolerance: PSNR between the two renders >= 60 dB and max |diff| <= 5e-3 in colour and alpha (measured: 87 dB, 4e-4; the fp16 table shadow and the fp16
    activations inside the fused MLP are the only precision gap)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import oracle as O
    from jnerf_amd import ops
    from jnerf_amd.utils.registry import build_from_cfg, DATASETS
    r = _runner(fp16=True, aabb_scale=1, const_dt=True, log_dir=str(tmp_path))
    for i in range(300):
        r.train_step(i)
    r.drain()
    ds = r.dataset["test"] = build_from_cfg(r.cfg.dataset.test, DATASETS)
    W, H = int(r.W), int(r.H)
    img_ids = torch.zeros((H * W,), dtype=torch.int32, device=ds.device)
    rays_o, rays_d, _ = ds.generate_rays_total_test(img_ids, W, H)
    s = r.sampler
    # (a) HIP: march (bit-identical to the oracle's marcher, test_hip_parity) -> network -> compositing
    with torch.no_grad():
        coords, numsteps, counters, _ = ops.march_rays(rays_o.contiguous(), rays_d.contiguous(), s.density_grid_bitfield, s.aabb_range, s.rng_state.copy(), s.max_samples,
                                                       s.cone_angle_constant, s.near_distance, s.const_dt, s.NERF_CASCADES)
        n = int(counters[1].item())
        assert 10000 < n < s.max_samples
        coords = coords[:n].contiguous()
        s._coords, s._n_valid = coords, None
        out_hip = r.model(coords[:, :3], coords[:, 4:])
        rgb_hip, alpha_hip = ops.composite_inference(out_hip.contiguous(), coords, numsteps, s.NERF_CASCADES)
    # (b) oracle, fp32 end to end, same samples and weights (fp32 masters)
    c = coords.cpu().numpy()
    table, _, n_params = O.level_table(1)
    grid = r.model.pos_encoder.m_grid.detach().float().cpu().numpy()
    wd = r.model.density_mlp.con_weights.detach().float().cpu().numpy()
    wc = r.model.rgb_mlp.con_weights.detach().float().cpu().numpy()
    feat = O.hash_encode_fwd(np.ascontiguousarray(c[:, :3]), grid, table)
    sh = O.sh_encode(np.ascontiguousarray(c[:, 4:]), np.float32)
    out_ref = O.field_fwd(feat, sh, wd, wc)
    rgb_ref, alpha_ref = O.composite_inference(out_ref, c, numsteps.cpu().numpy().view(np.uint32), s.NERF_CASCADES)
    a, b = rgb_hip.cpu().numpy(), rgb_ref
    mse = float(np.mean((a - b) ** 2))
    psnr = -10 * np.log10(max(mse, 1e-12))
    print(f"HIP (fp16) vs CPU oracle (fp32) on {H * W} identical rays / {n} samples: PSNR {psnr:.1f} dB, max |diff| {np.abs(a - b).max():.4f}, alpha max |diff| {np.abs(alpha_hip.cpu().numpy() - alpha_ref).max():.4f}")
    assert psnr >= 60.0 and np.abs(a - b).max() <= 5e-3 and np.abs(alpha_hip.cpu().numpy() - alpha_ref).max() <= 5e-3
    assert float(np.mean(alpha_ref)) > 0.02           # the view actually shows the object


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_data_parallel_on_one_gpu(scaling):
    """the N>1 code path of bench.py (per-rank ray batches, gradient all-reduce on a side stream, deferred fused sweep, synchronised ray-count
    adaptation) with two ranks sharing cuda:0 over gloo — RCCL itself needs one GPU per rank and is exercised by the driver's scaling run"""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "20", "--warmup", "8", "--burn-in", "32", "--config", "fox", "--images", "4", "--res", "64", "--no-psnr", "--scaling", scaling]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and np.isfinite(d["loss"]) and d["config"]["parallelism"].startswith("ray-batch dp2") and d["scaling"] == scaling
    assert d["config"]["samples_per_iter_per_gpu"] == ((1 << 18) if scaling == "weak" else (1 << 17))       # strong: ONE 2^18-sample iteration split over the ranks
    assert d["extra"]["replicas_identical"] is True           # both ranks hold bit-identical parameters after 60 data-parallel steps
    assert d["extra"]["native_step"] is True                  # data parallel keeps the one-call native step (here: two phases around gloo's all-reduce)
    # (r6) one run prints both scaling modes: the contract line in the requested one, a second timed region in the other
    o = d["extra"]["dp"]["other_scaling"]
    assert "error" not in o, o
    assert o["scaling"] == ("strong" if scaling == "weak" else "weak") and o["value"] > 0 and np.isfinite(o["loss"])
    assert o["samples_per_iter_per_gpu"] == ((1 << 17) if scaling == "weak" else (1 << 18))


@pytest.mark.parametrize("world,n_buckets,half", [(2, 1, False), (2, 2, False), (3, 2, True), (8, 1, True), (8, 2, False)])
def test_sharded_sweep_tiles_the_replicated_sweep(world, n_buckets, half):
    """The in-library exchange (reduce-scatter -> sweep of the rank's shard -> all-gather) needs one GPU per rank, which this box does not have.  Its SHARD ARITHMETIC is
    exercised here without a process group: for every rank of a `world`-rank plan, ngp_train_step(NGP_PHASE_SWEEP, plan of that rank, no communicator) sweeps the rank's
    shards + the replicated tail of its own copy of (parameters, moments, fp16 shadow); stitching the ranks' shards together (the all-gather) must reproduce, bit for bit,
    ONE replicated sweep of the whole table - parameters, both Adam moments and the fp16 shadow - and leave everything outside a rank's shards and the tail untouched."""
    import ctypes as C
    from jnerf_amd import _lib as L, ops, dp
    lt, _, n_params = ops.level_table(1)
    n_params_t = n_params - 4                                  # a ragged tail (the sweep's vectors need a multiple of 4; real tables are multiples of 8): 12 elements for every world size here
    torch.manual_seed(world * 10 + n_buckets)
    dev = "cuda"
    p0 = torch.randn(n_params_t, device=dev) * 1e-2
    g0 = torch.randn(n_params_t, device=dev) * 1e-4
    m0, v0 = torch.randn(n_params_t, device=dev) * 1e-5, torch.rand(n_params_t, device=dev) * 1e-9
    lr, step, b0, b1, eps, decay = 1e-2, 7, 0.9, 0.99, 1e-15, 0.95
    # the replicated sweep (EMA aliasing the parameter, as the training path has it)
    pr, mr, vr = p0.clone(), m0.clone(), v0.clone()
    hr = torch.zeros(n_params_t, dtype=torch.float16, device=dev) if half else None
    ops.adam_ema_step(pr, g0.clone(), mr, vr, pr, hr, lr, step, b0, b1, eps, decay, zero_grad=False)
    stitched = [torch.full_like(p0, float("nan")) for _ in range(3)]
    stitched_h = torch.zeros(n_params_t, dtype=torch.float16, device=dev) if half else None
    covered = torch.zeros(n_params_t, dtype=torch.int32, device=dev)
    for rank in range(world):
        plan = L.NgpDpPlan()
        L.check(L.lib().ngp_dp_plan(ops._tbl(lt), n_params_t, world, rank, n_buckets, C.byref(plan)), "ngp_dp_plan")
        p, m, v, g = p0.clone(), m0.clone(), v0.clone(), g0.clone()
        h = torch.full((n_params_t,), -7.0, dtype=torch.float16, device=dev) if half else None
        a = L.NgpTrainStep()
        a.run_optimizer, a.phase, a.dtype, a.timed_stage, a.grad_overwrite = 1, L.PHASE_SWEEP, (L.F16 if half else L.F32), -1, 1
        a.n_opt, a.step, a.lr, a.beta0, a.beta1, a.eps, a.ema_decay = 1, step, lr, b0, b1, eps, decay
        a.p[0], a.g[0], a.m[0], a.v[0], a.ema[0], a.numel[0] = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.data_ptr(), n_params_t
        a.p_half[0] = h.data_ptr() if half else None
        a.table_grad, a.n_params, a.dp_table, a.dp = g.data_ptr(), n_params_t, 0, C.addressof(plan)
        L.check(L.lib().ngp_train_step(ops._stream(), C.byref(a)), "ngp_train_step(sweep, sharded)")
        own = torch.zeros(n_params_t, dtype=torch.bool, device=dev)
        for b in range(plan.n_buckets):
            own[plan.shard_begin[b]:plan.shard_begin[b] + plan.shard_count[b]] = True
        tail = torch.zeros_like(own)
        tail[plan.tail_begin:plan.tail_begin + plan.tail_count] = True
        assert int(plan.tail_begin + plan.tail_count) == n_params_t and 0 < int(plan.tail_count) < 8 * world
        untouched = ~(own | tail)
        assert torch.equal(p[untouched], p0[untouched]) and torch.equal(m[untouched], m0[untouched]) and torch.equal(v[untouched], v0[untouched])
        if half:
            assert bool((h[untouched] == -7.0).all())
        for dst, src, whole in zip(stitched, (p, m, v), (pr, mr, vr)):
            dst[own] = src[own]
            assert torch.equal(src[tail], whole[tail])                                        # the tail is swept by every rank, identically
        if half:
            stitched_h[own] = h[own]
            assert torch.equal(h[tail], hr[tail])
        covered += own.int()
    main = covered[:int(plan.tail_begin)]
    assert bool((main == 1).all()) and bool((covered[int(plan.tail_begin):] == 0).all())         # every element of the main part belongs to exactly one rank
    k = int(plan.tail_begin)
    for got, want in zip(stitched, (pr, mr, vr)):
        assert torch.equal(got[:k], want[:k])
    if half:
        assert torch.equal(stitched_h[:k], hr[:k])


@pytest.mark.parametrize("config,overlap", [("fox", False), ("lego", True)])
def test_sharded_sweep_with_two_ranks_on_one_gpu(config, overlap):
    """the same shard arithmetic end to end: two ranks on one GPU, the host sums the gradient (gloo) and gathers the shards (`--dp-host-sharded`: NGP_PHASE_SWEEP with a plan
    and no communicator, Adam moments - fp16 mode: the fp32 master too - living on their owner's shard until sync_sharded_state()).  Replicas must end bit-identical and
    train like the replicated sweep.  (Not compared bit for bit with a second run: with two PROCESSES time-sharing one GPU, runs of this configuration are not
    reproducible to the last bit on this platform - DESIGN.md section 6 - while single-process runs are; the bitwise statement is the test above.)"""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for sharded in (False, True):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "20", "--warmup", "8", "--burn-in", "32", "--config", config, "--images", "4", "--res", "64",
               "--no-psnr", "--no-kernel-events"] + (["--dp-host-sharded"] if sharded else []) + (["--dp-overlap"] if overlap else [])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
        d = json.loads(lines[0])
        assert d["extra"]["replicas_identical"] is True and d["extra"]["native_step"] is True and np.isfinite(d["loss"])
        assert ("sharded sweep" in d["extra"]["dp"]["exchange"]) == sharded
        res.append(d)
    assert abs(res[0]["loss"] - res[1]["loss"]) < 0.05 * res[0]["loss"]
    for x, y in zip(res[0]["extra"]["param_signature"], res[1]["extra"]["param_signature"]):
        assert abs(x - y) <= 0.25 * max(abs(x), abs(y), 1.0), (res[0]["extra"]["param_signature"], res[1]["extra"]["param_signature"])


def _run_bench(extra_args, timeout=900, env=None):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra_args, capture_output=True, text=True, timeout=timeout, cwd=root, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("config", ["fox", "lego"])
def test_rccl_world_size_one(config):
    """RCCL on the one GPU this box has: bench.py --force-dist creates the nccl (= RCCL) process group with world size 1 and runs the COMPLETE data-parallel
    sequence every step inside the ONE native call - fp32 -> scaled fp16 gradient conversion (fox), reduce-scatter + tail / MLP all-reduce as one RCCL group, sweep of the
    rank's shard, all-gather of the updated parameters - plus the all-reduced ray-count adaptation.  (N > 1 needs one GPU per rank: the driver's scaling run; gloo covers two ranks in test_two_ranks_...)"""
    import os, socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    d = _run_bench(["--gpus", "1", "--force-dist", "--config", config, "--steps", "48", "--warmup", "8", "--burn-in", "64", "--images", "8", "--res", "96", "--no-psnr", "--no-fox",
                    "--no-cpu-baseline"], env=env)
    assert d["n_gpus"] == 1 and d["value"] > 0 and np.isfinite(d["loss"]) and d["loss"] < 0.2
    assert d["extra"]["dist_backend"] == "nccl" and d["extra"]["replicas_identical"] is True and d["extra"]["native_step"] is True
    assert d["extra"]["dp"]["exchange"].startswith("rccl in-library")


@pytest.mark.parametrize("overlap", [False, True])
def test_rccl_exchange_step_adds_no_arithmetic(overlap):
    """fp32 mode (ngp_base.py): the world-size-1 data-parallel run - in-library RCCL reduce-scatter, sharded sweep (shard + tail + MLP pack), all-gather, with and
    without the overlapped two-bucket variant - must leave bit-identical parameters to the plain single-GPU run of the same seed: the exchange step only moves data"""
    import os, socket
    common = ["--gpus", "1", "--config", "lego", "--steps", "24", "--warmup", "4", "--burn-in", "36", "--images", "8", "--res", "96", "--no-psnr", "--no-fox", "--no-cpu-baseline",
              "--no-kernel-events"]
    a = _run_bench(common)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    b = _run_bench(common + ["--force-dist"] + (["--dp-overlap"] if overlap else []), env=env)
    assert b["extra"]["native_step"] is True and b["extra"]["dp"]["overlap"] is overlap
    assert a["extra"]["param_signature"] == b["extra"]["param_signature"], (a["extra"]["param_signature"], b["extra"]["param_signature"])
    assert a["loss"] == b["loss"]


def test_rccl_fp16_gradient_wire_in_the_fp32_configuration():
    """(r4) `dp_grad_dtype = "fp16"` in the fp32 configuration: the table gradient travels as fp16 x 2^14 (half the reduce-scatter bytes) - opt-in, NOT bit-identical to the
    plain run (every element rounded to 11 significant bits once); the world-size-1 run must train like the plain one and say which wire it used"""
    import os, socket
    common = ["--gpus", "1", "--config", "lego", "--steps", "24", "--warmup", "4", "--burn-in", "36", "--images", "8", "--res", "96", "--no-psnr", "--no-fox", "--no-neus", "--no-spheres",
              "--no-cpu-baseline", "--no-kernel-events"]
    a = _run_bench(common)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_EXTRA_CFG='{"dp_grad_dtype": "fp16"}')
    b = _run_bench(common + ["--force-dist"], env=env)
    assert b["extra"]["native_step"] is True and b["extra"]["dp"]["grad_wire"].startswith("fp16") and b["extra"]["dp"]["n_ranks_seen"] == 1
    assert b["extra"]["dp"]["exchange"].startswith("rccl in-library") and b["extra"]["replicas_identical"] is True
    assert np.isfinite(b["loss"]) and abs(b["loss"] - a["loss"]) < 0.05 * a["loss"], (a["loss"], b["loss"])
    for x, y in zip(a["extra"]["param_signature"], b["extra"]["param_signature"]):
        assert abs(x - y) <= 0.05 * max(abs(x), abs(y), 1.0), (a["extra"]["param_signature"], b["extra"]["param_signature"])


def test_bench_contract_small():
    """the bench line's contract on a tiny workload: metric / value / roofline (dominant kernel chosen over ALL kernels, live durations) / per-kernel table"""
    d = _run_bench(["--steps", "16", "--warmup", "4", "--burn-in", "64", "--images", "8", "--res", "96", "--no-fox", "--no-cpu-baseline"])
    assert d["metric"] == "training iters/s" and d["dtype"] == "f32" and d["scaling"] == "weak" and d["n_gpus"] == 1 and d["value"] > 0
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["kernel"] in rf["ms_per_step_by_kernel"] and rf["launches_timed"] >= 8
    assert rf["ms_per_step_by_kernel"][rf["kernel"]] == max(rf["ms_per_step_by_kernel"].values())
    assert {"k_hash_fwd_x2", "k_bin_accumulate2_adam", "k_composite_train"} <= set(d["extra"]["probe_kernels"])      # (r5: compositing forward + Huber + backward are one launch in the native step; r6: the fp32 gather runs two lanes per (sample, level))
    assert not ({"k_reduce_slabs", "k_mlp32_sweep_pack", "k_adam_ema", "k_bin_accumulate2"} & set(d["extra"]["probe_kernels"]))   # (r6: the MLP tail rides in the hash backward's record launches, the table's sweep in its accumulate kernel)
    assert rf["stage"]["name"] == "hash_backward+table_sweep" and "k_bin_accumulate2_adam" in rf["stage"]["kernels"]
    assert any(k.startswith("k_march") for k in d["extra"]["probe_kernels"])
    # (r5) no fraction above 1 is printable: split-operand kernels are scored on the fp16 pipe they issue on; the whole-step HBM fraction is in the line;
    # `traffic` is null unless this round's counter pass of this scene is committed (this tiny scene has none)
    assert 0.0 < rf["frac"] <= 1.0 and rf["achieved"] <= rf["peak"]
    st = rf["step"]
    assert st["alg_bytes"] > 0 and 0.0 < st["frac_hbm"] <= 1.0 and abs(st["GBps"] - st["alg_bytes"] / (d["ms_per_step"] * 1e-3) / 1e9) <= 0.01 * st["GBps"] + 0.1
    assert rf["traffic"] is None and rf["traffic_source"] is None
    for k, row in d["extra"]["probe_kernels"].items():
        assert (row["frac_hbm"] is None and row.get("served_from_cache") and row["avg_launch_ms"] < 0.02) or row["frac_hbm"] <= 1.0, (k, row)      # (only a few-microsecond kernel on cache-resident data may exceed HBM's rate; it then carries no fraction)
        assert row.get("frac_mfma", 0.0) <= 1.0, (k, row)


def test_render_survives_sample_capacity_overflow():
    """a 32768-ray inference chunk can ask for more than the sampler's fixed 4096*1024 samples (fully occupied grid: up to 1024 per ray);
    the trailing rays used to come back black.  Now the overflow is detected on the device and the image is re-rendered in 4096-ray chunks."""
    r = _runner(fp16=True, aabb_scale=1, const_dt=True)
    for i in range(64):
        r.train_step(i)
    r.drain()
    r.sampler.density_grid_bitfield.fill_(255)                    # everything occupied: every ray takes the full 1024 steps inside the box
    ds = r.dataset["train"]
    W, H = int(r.W), int(r.H)
    ids = torch.zeros((H * W,), dtype=torch.int32, device=ds.device)
    ro, rd, _ = ds.generate_rays_total_test(ids, W, H)
    with torch.no_grad():
        big, a_big = r._render_rays(ids, ro, rd, 32768)           # 9216 rays x ~1000 samples > 4 M: overflows, must fall back
        ref, a_ref = r._render_rays(ids, ro, rd, 4096)
    assert float(a_ref[-1024:].mean()) > 0.01                     # the trailing rays do see density
    mse = float(((big - ref) ** 2).mean())
    assert mse < 1e-3, mse                                        # (the two renders differ only by the marcher's per-call start jitter)
    assert abs(float(a_big[-1024:].mean()) - float(a_ref[-1024:].mean())) < 0.05


def test_render_task_writes_the_camera_path(tmp_path):
    """Runner.render (runner.py:101-121; what the reference's tools/run_net.py --task render calls): checkpoint -> frames along camera_path.path_spherical().
    Without cv2 the frames are PNGs next to the requested .mp4."""
    import os
    from PIL import Image
    r = _runner(fp16=True, aabb_scale=1, const_dt=True)
    for i in range(48):
        r.train_step(i)
    r.drain()
    r.ckpt_path = str(tmp_path / "params.pkl")
    r.save_ckpt(r.ckpt_path)
    out = r.render(True, str(tmp_path / "demo.mp4"), nframe=3)
    if out.endswith(".mp4"):
        assert os.path.getsize(out) > 0
    else:
        frames = sorted(os.listdir(out))
        assert frames == ["0000.png", "0001.png", "0002.png"]
        img = np.asarray(Image.open(os.path.join(out, frames[1])))
        assert img.shape == (int(r.H), int(r.W), 3) and img.std() > 0


def test_reference_configs_train_on_real_fox(tmp_path):
    """projects/ngp/configs/ngp_fox.py (same keys/values as the reference's file, checked key by key against /root/reference in tests/test_host_cpu.py) on the REAL
    fox photographs: 50 images 1080x1920 load through NerfDataset, 64 training steps run on the native fast path, the loss falls."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isfile(os.path.join(root, "data", "fox", "transforms_train.json")):
        pytest.skip("data/fox not in the tree (build() copies it where /root/reference exists)")
    from jnerf_amd.utils.config import init_cfg, get_cfg
    from jnerf_amd.runner import Runner
    cwd = os.getcwd()
    os.chdir(root)
    try:
        init_cfg(os.path.join(root, "projects", "ngp", "configs", "ngp_fox.py"))
        get_cfg().log_dir = str(tmp_path)
        torch.manual_seed(0)
        r = Runner()
        ds = r.dataset["train"]
        assert ds.n_images == 50 and ds.resolution == [1080, 1920] and ds.aabb_scale == 4
        assert abs(float(ds.focal_lengths[0, 0]) - 1375.52) < 1e-2 and abs(float(ds.focal_lengths[0, 1]) - 1374.49) < 1e-2
        losses = [float(r.train_step(i).mean().item()) for i in range(64)]
        assert r._fast and r._fast.native
        assert np.isfinite(losses).all() and np.mean(losses[-8:]) < 0.7 * np.mean(losses[:8]), (losses[:8], losses[-8:])
        r.drain()
    finally:
        os.chdir(cwd)


def test_fp32_fused_network_equals_linear_chain():
    """cfg.fp16 unset (ngp_base.py): `use_fully` runs the five bias-free nn.Linear layers as ONE fp32-MFMA kernel whose weights are views of a flat pack; with
    use_fully = False the same module runs torch's nn.Linear / autograd chain (what the reference does, ngp_network.py:59-67).  Same parameters => same
    outputs and same parameter gradients (fp32 rounding apart), and the state dict has the reference's keys and shapes."""
    from jnerf_amd.presets import ngp_cfg
    from jnerf_amd.utils.registry import build_from_cfg, NETWORKS, DATASETS
    from jnerf_amd.utils.config import get_cfg
    import jnerf_amd.dataset, jnerf_amd.network, jnerf_amd.encoders  # noqa: F401  (register the modules: this test may be the first of its process)
    torch.manual_seed(3)
    outs = []
    sd = None
    for fully in (True, False):
        cfg = ngp_cfg(fp16=False, aabb_scale=1, const_dt=True, n_images=2, W=16, H=16)
        cfg.model.use_fully = fully
        cfg.dataset_obj = build_from_cfg(cfg.dataset.train, DATASETS)
        net = build_from_cfg(cfg.model, NETWORKS)
        assert net.fused == fully
        if sd is None:
            sd = {k: v.clone() for k, v in net.state_dict().items()}
            assert {k: tuple(v.shape) for k, v in sd.items() if "mlp" in k} == {"density_mlp.0.weight": (64, 32), "density_mlp.2.weight": (16, 64), "rgb_mlp.0.weight": (64, 32),
                                                                               "rgb_mlp.2.weight": (64, 64), "rgb_mlp.4.weight": (3, 64)}
        net.load_state_dict(sd)
        with torch.no_grad():
            net.pos_encoder.m_grid.mul_(1e3)                       # features of order 0.1 instead of 1e-4
        x, d = torch.rand((3000, 3), device="cuda"), torch.rand((3000, 3), device="cuda")
        torch.manual_seed(5)
        x, d = torch.rand((3000, 3), device="cuda"), torch.rand((3000, 3), device="cuda")
        out = net(x, d)
        g = torch.randn((3000, 4), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        out.backward(g)
        outs.append((out.detach().clone(), [p.grad.detach().clone() for p in net.mlp_params()], net.pos_encoder.m_grid.grad.detach().clone(), net.density(x).detach().clone()))
    (o1, w1, t1, d1), (o0, w0, t0, d0) = outs
    sc = float(o0.abs().max())
    assert (o1 - o0).abs().max() <= 1e-5 * max(sc, 1.0) and (d1 - d0).abs().max() <= 1e-5 * max(sc, 1.0)
    # (a ReLU whose pre-activation is within fp32 rounding of zero may open on one side only: that moves one sample's gradient by a whole neuron's contribution,
    # so gradients are compared in norm, not element by element)
    for a, b in zip(w1, w0):
        assert a.shape == b.shape and float((a - b).norm()) <= 1e-4 * float(b.norm()), (a.shape, float((a - b).norm()), float(b.norm()))
    assert float((t1 - t0).norm()) <= 1e-4 * float(t0.norm())


def test_original_nerf_config_plumbing(tmp_path):
    """BASELINE config [0]: projects/nerf/configs/nerf_base.py's stack (FrequencyEncoder 10 / 4 bands, OriginNeRFNetworks 8 x 256, fp16 autocast, DensityGridSampler, Huber,
    Adam lr 1e-2 + EMA) on a small procedural scene through the module path: occupancy refresh, marching, HIP compositing and its backward, torch MLP, fused sweep."""
    from jnerf_amd.utils.config import init_cfg, get_cfg
    from jnerf_amd.runner import Runner
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    init_cfg(os.path.join(root, "projects", "nerf", "configs", "nerf_base.py"))
    cfg = get_cfg()
    ds = dict(type="SyntheticNerfDataset", batch_size=512, n_images=6, W=64, H=64, aabb_scale=1)
    cfg.dataset = cfg.dfs(dict(train=dict(ds, mode="train"), test=dict(ds, mode="test", n_images=1)))
    cfg.target_batch_size, cfg.log_dir = 1 << 14, str(tmp_path)
    torch.manual_seed(0)
    r = Runner()
    assert type(r.model).__name__ == "OriginNeRFNetworks" and not r._fast
    w0 = [p.detach().clone() for p in r.model.parameters()]
    losses = [float(r.train_step(i).mean().item()) for i in range(48)]
    assert np.isfinite(losses).all()                               # (plumbing only: at lr 1e-2 the 8 x 256 MLP needs thousands of iterations to show in the loss)
    assert r._fast is False                                        # generic module path, not the fused fast path
    moved = [float((p.detach() - q).abs().max()) for p, q in zip(r.model.parameters(), w0)]
    assert all(np.isfinite(moved)) and min(moved) > 0.0, moved     # every tensor - the 1- and 3-element head biases included - received gradients and was stepped
    r.drain()


def test_nerf_dataset_on_disk(tmp_path):
    """NerfDataset (dataset.py:68-170 semantics): transforms JSON + PNGs, train includes val, missing files skipped, fl_x / camera_angle_x, aabb_scale"""
    import json
    from PIL import Image
    from jnerf_amd.utils.config import reset_cfg
    from jnerf_amd.dataset import NerfDataset, fov_to_focal_length
    reset_cfg(device="cuda")
    rng = np.random.default_rng(0)
    W, H = 20, 12

    def frames(prefix, n):
        out = []
        for i in range(n):
            Image.fromarray(rng.integers(0, 255, (H, W, 4), dtype=np.uint8)).save(tmp_path / f"{prefix}_{i}.png")
            m = np.eye(4); m[:3, 3] = rng.normal(size=3)
            out.append({"file_path": f"./{prefix}_{i}", "transform_matrix": m.tolist()})
        return out
    tr = frames("train", 3)
    tr.append({"file_path": "./missing", "transform_matrix": np.eye(4).tolist()})
    json.dump({"camera_angle_x": 0.7, "aabb_scale": 2, "frames": tr}, open(tmp_path / "transforms_train.json", "w"))
    json.dump({"camera_angle_x": 0.7, "aabb_scale": 2, "frames": frames("val", 2)}, open(tmp_path / "transforms_val.json", "w"))
    ds = NerfDataset(str(tmp_path), batch_size=64, mode="train")
    assert ds.n_images == 5 and ds.resolution == [W, H] and ds.aabb_scale == 2 and ds.aabb_range == (-0.5, 1.5)
    assert ds.image_data.shape == (5, H * W, 4) and ds.transforms_gpu.shape == (5, 4, 3) and ds.metadata.shape == (5, 11)
    f = fov_to_focal_length(W, 0.7 * 180 / np.pi)
    assert np.allclose(ds.focal_lengths.cpu().numpy(), f)
    img_ids, o, d, rgba = next(ds)
    assert img_ids.shape == (64,) and o.shape == (64, 3) and rgba.shape == (64, 4)
    assert torch.allclose(d.norm(dim=-1), torch.ones(64, device="cuda"), atol=1e-5)
    ro, rd, pix = ds.generate_rays_total_test(torch.zeros(H * W, dtype=torch.int32, device="cuda"), W, H)
    assert ro.shape == (H * W, 3) and torch.equal(pix, torch.arange(H * W, device="cuda"))
    # origin = translation * 0.33 + 0.5 with rows cycled [1,2,0] (dataset.py:255-262)
    t = np.array(tr[0]["transform_matrix"])[:3, 3] * 0.33 + 0.5
    assert np.allclose(ro[0].cpu().numpy(), t[[1, 2, 0]], atol=1e-6)


@pytest.mark.parametrize("fp16", [True, False])
def test_fast_path_equals_module_path(fp16):
    """fastpath.FusedTrainStep (native ngp_train_step, gradients overwritten) launches the same kernels as the autograd/module path (gradients accumulated, zeroed by
    the sweep): parameters after a few steps agree.  fp16 = the fused fp16-MFMA stack, fp32 = the fp32-MFMA stack ngp_base.py runs"""
    res = []
    for fast in (True, False):
        r = _runner(fp16=fp16, aabb_scale=1, const_dt=True, fast_path=fast, pipeline_sampling=False)
        for i in range(3):
            l = r.train_step(i)
        assert bool(r._fast) == fast and r.model.fused
        w = torch.cat([p.detach().reshape(-1) for p in r.model.mlp_params()])
        res.append((r.model.pos_encoder.m_grid.detach().clone(), w.clone(), float(l.sum().item())))
        r.drain()
    (g0, w0, l0), (g1, w1, l1) = res
    assert torch.allclose(w0, w1, rtol=2e-3, atol=2e-4), (w0 - w1).abs().max()
    assert (g0 - g1).abs().max() < 2e-3 and abs(l0 - l1) < 2e-2 * max(abs(l1), 1e-3)


def test_lego_gate_runs_when_the_dataset_is_mounted(tmp_path, monkeypatch):
    """tools/lego_gate.py (VERDICT r4 'missing' #1): with NeRF-synthetic lego absent it says so; with a data set in lego's LAYOUT at $NGP_LEGO_DIR (here: views of the
    procedural scene written as PNG + transforms_{train,val,test}.json in the NeRF convention) it runs projects/ngp/configs/ngp_base.py on it and reports it/s, wall
    seconds and the mean test PSNR - as a plumbing run, labelled as such, when the step count is not the schedule's 40 000."""
    import json
    import os
    import sys
    from PIL import Image
    from jnerf_amd.utils.config import reset_cfg
    from jnerf_amd.dataset import SyntheticNerfDataset, NERF_SCALE
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import lego_gate as LG
    monkeypatch.delenv("NGP_LEGO_DIR", raising=False)
    if LG.find_lego()[0] is None:
        assert LG.lego_gate()["gate"] == "not runnable: dataset absent"
    W = H = 64
    reset_cfg(device="cuda")
    fov = 40.0

    def split(mode, n, seed):
        ds = SyntheticNerfDataset(batch_size=64, n_images=n, W=W, H=H, aabb_scale=1, mode=mode, seed=seed, fov_deg=fov)
        frames = []
        P = ds.transforms_gpu.transpose(1, 2).cpu().numpy()                     # [n, 3, 4] ngp poses
        for i in range(n):
            rgba = (ds.image_data[i].view(H, W, 4).clamp(0, 1) * 255 + 0.5).to(torch.uint8).cpu().numpy()
            Image.fromarray(rgba, "RGBA").save(tmp_path / f"{mode}_{i}.png")
            m = P[i][[2, 0, 1]].copy()                                           # undo the row cycle [1, 2, 0] of matrix_nerf2ngp (dataset.py:255-262) ...
            m[:, 1] *= -1; m[:, 2] *= -1                                         # ... the correct_pose flips [1, -1, -1] ...
            m[:, 3] = (m[:, 3] - 0.5) / NERF_SCALE                               # ... and translation * 0.33 + 0.5
            frames.append({"file_path": f"./{mode}_{i}", "transform_matrix": np.vstack([m, [0, 0, 0, 1]]).tolist()})
        json.dump({"camera_angle_x": float(np.deg2rad(fov)), "frames": frames}, open(tmp_path / f"transforms_{mode}.json", "w"))
    split("train", 10, 0); split("val", 10, 1); split("test", 2, 2)
    monkeypatch.setenv("NGP_LEGO_DIR", str(tmp_path))
    out = LG.lego_gate(steps=400)
    assert out["gate"].startswith("not the gate: 400 steps") and out["steps"] == 400 and out["dataset"] == str(tmp_path)
    assert out["train_images"] == 20 and out["resolution"] == [W, H] and out["test_views"] == 2                     # train includes val (dataset.py:77)
    assert out["iters_per_s"] > 0 and np.isfinite(out["psnr_lego_test"]) and out["psnr_lego_test"] > 18.0, out     # the poses were understood: the scene is being learnt


@pytest.mark.parametrize("fp16", [True, False])
def test_fused_launches_of_the_native_step_change_no_bit(fp16, monkeypatch):
    """(r5) Two fusions inside ngp_train_step - compositing forward + Huber + backward as one launch (ngp_composite_train), and, fp16 configuration, the slab reduction that also
    sweeps the two MLP weight packs (ngp_reduce_slabs_sweep) - against the launches they replace (NGP_SPLIT_COMPOSITE / NGP_NO_FUSED_MLP_TAIL select those): 24 iterations
    from the same seed, refresh included, must leave the SAME BITS in every parameter, every Adam moment of the MLP packs and the last loss."""
    def run():
        r = _runner(fp16=fp16, aabb_scale=1, const_dt=True, pipeline_sampling=False)
        for i in range(24):
            loss = r.train_step(i)
        r.drain()
        assert r._fast and r._fast.native
        adam = r.optimizer._nested_optimizer
        state = [p.detach().clone() for p in r.model.parameters()] + [t.detach().clone() for t in adam.param_groups[0]["m"]] + [t.detach().clone() for t in adam.param_groups[0]["values"]]
        return state, loss.detach().clone()
    monkeypatch.delenv("NGP_SPLIT_COMPOSITE", raising=False); monkeypatch.delenv("NGP_NO_FUSED_MLP_TAIL", raising=False)
    fused, l_fused = run()
    monkeypatch.setenv("NGP_SPLIT_COMPOSITE", "1"); monkeypatch.setenv("NGP_NO_FUSED_MLP_TAIL", "1")
    split, l_split = run()
    assert torch.equal(l_fused, l_split)
    for a, b in zip(fused, split):
        assert torch.equal(a, b)
    assert any(float(t.abs().max()) > 0 for t in fused)
    # (r6) by default the MLP tail RIDES in the hash backward's record launches (csrc/mlp_tail.h) - fp32 configuration: the slab reduction in k_bin_runs2's grid, the pack's
    # sweep + fragment packing in k_bin_pairs'; fp16 configuration: the reduction that also sweeps the two packs in k_bin_records_runs'.  NGP_NO_TAIL_RIDE leaves them as the
    # launches of rounds 3-5 (k_reduce_slabs + k_mlp32_sweep_pack | k_reduce_slabs_sweep) - same bits again
    monkeypatch.delenv("NGP_SPLIT_COMPOSITE", raising=False); monkeypatch.delenv("NGP_NO_FUSED_MLP_TAIL", raising=False)
    monkeypatch.setenv("NGP_NO_TAIL_RIDE", "1")
    own, l_own = run()
    assert torch.equal(l_fused, l_own)
    for a, b in zip(fused, own):
        assert torch.equal(a, b)
    # (r6) fp32 configuration: the hash table's Adam + EMA sweep rides in the accumulate kernel (k_bin_accumulate2_adam applies the update where it would store the gradient,
    # which then is never written).  NGP_NO_ADAM_RIDE leaves the gradient store + the k_adam_ema launch - same bits in the table, its moments and everything downstream
    monkeypatch.delenv("NGP_NO_TAIL_RIDE", raising=False)
    monkeypatch.setenv("NGP_NO_ADAM_RIDE", "1")
    swept, l_swept = run()
    assert torch.equal(l_fused, l_swept)
    for a, b in zip(fused, swept):
        assert torch.equal(a, b)
