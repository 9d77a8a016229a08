#!/usr/bin/env python
"""bench.py — Instant-NGP training throughput on MI355X (BASELINE.json metric "training iters/s", config[1]: fox-config
Instant-NGP, L=16 hash levels, T=2^19, fp16 fused MLP, 2^18-sample batches; synthetic procedural scene, random-init weights).

One "step" = one full training iteration of Runner.train (ray generation + random background, occupancy-grid marching/compaction,
hash encode, fused MLP, compositing, Huber, backward, fused Adam+EMA; occupancy-grid update every 16th step) over one 2^18-sample batch.
`value` = (n_gpus * steps) / seconds: iterations per second where every rank trains its own 2^18-sample ray batch and the hash-table /
MLP gradients are all-reduced over RCCL each step (weak scaling; at n_gpus=1 this is exactly the reference's it/s).
Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _cpu_inputs(n_samples, n_rays, m_rays, seed):
    """synthetic inputs of one (slice of a) training iteration for the CPU port"""
    import numpy as np
    import synth
    x = synth.uniform_positions(n_samples, seed=seed) if "seed" in synth.uniform_positions.__code__.co_varnames else synth.uniform_positions(n_samples)
    d = synth.unit_dirs01(n_samples)
    coords = np.zeros((n_samples, 7), np.float32); coords[:, :3] = x; coords[:, 4:] = d
    per = n_samples // n_rays
    ns = np.stack([np.full(n_rays, per, np.uint32), (np.arange(n_rays) * per).astype(np.uint32)], 1)
    bg = np.random.default_rng(seed).random((n_rays, 3), dtype=np.float32)
    xf, focal, meta = synth.camera_ring(8, radius=1.3)
    _, ro, rd, _ = synth.rays_from_cameras(xf, focal, meta, 400, 400, m_rays, seed=1 + seed)
    return dict(x=x, d=d, coords=coords, ns=ns, bg=bg, ro=ro, rd=rd)


def _cpu_iteration(O, inp, table, n_params, grid, wd, wc, bits, npar, stages=None):
    """every hot-path stage of one iteration through the plain-C oracle (ctypes releases the GIL inside each call)"""
    import numpy as np

    def timed(name, fn):
        ts = time.perf_counter()
        r = fn()
        if stages is not None:
            stages[name] = round((time.perf_counter() - ts) * 1e3, 1)
        return r

    x, d, coords, ns, bg = inp["x"], inp["d"], inp["coords"], inp["ns"], inp["bg"]
    timed("march", lambda: O.march_rays(inp["ro"], inp["rd"], bits, (-1.5, 2.5), O.PCG32(1337), 4096 * 1024, const_dt=False))   # the batch's ray count through the two-pass marcher
    feat = timed("hash_fwd", lambda: O.hash_encode_fwd(x, grid, table))
    sh = timed("sh", lambda: O.sh_encode(d, np.float32))
    out = timed("field_fwd", lambda: O.field_fwd(feat.astype(np.float32), sh, wd, wc))
    rgb = timed("composite_fwd", lambda: O.composite_fwd(out, coords, ns, ns, bg))
    _, G = O.huber(rgb, bg)
    dout = timed("composite_bwd", lambda: O.composite_bwd(out, coords, ns, G, rgb, 0.001))
    dfeat, dwd, dwc = timed("field_bwd", lambda: O.field_bwd(feat.astype(np.float32), sh, wd, wc, dout))
    g = timed("hash_bwd", lambda: O.hash_encode_bwd(x, dfeat.astype(np.float16), table, n_params))
    p = np.zeros(npar, np.float32); m = np.zeros_like(p); v = np.zeros_like(p); e = np.zeros_like(p)
    timed("adam_ema", lambda: O.adam_ema_step(p, g[:npar].astype(np.float32), m, v, e, 0.1, 1))


def cpu_baseline(n_samples=1 << 18, n_rays=4096, n_march_rays=39424):
    """The oracle (plain-C port of the reference's kernels) on ONE training iteration of the bench workload, on this host's cores: the iteration is split
    into `cores` independent ray/sample slices that run concurrently (one thread each; the hash-table gradient and the parameter sweep are sliced the same
    way), plus the same iteration on a single core for the per-stage times.  `value` is iterations/s on `cores` cores."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    import synth
    frac = n_samples / float(1 << 18)
    table, offsets, n_params = O.level_table(4)
    grid = synth.table(n_params, np.float16, amp=2e-4)
    wd, wc = synth.mlp_weights()
    bits = synth.shell_bitfield()
    m_rays = int(n_march_rays * frac)
    npar = int(n_params * frac) // 4 * 4
    cores = max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    # (a) `cores` slices in parallel
    parts = [_cpu_inputs(n_samples // cores, max(n_rays // cores, 1), max(m_rays // cores, 1), seed=k) for k in range(cores)]
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda inp: _cpu_iteration(O, inp, table, n_params, grid, wd, wc, bits, npar // cores // 4 * 4), parts))
        t_par = time.perf_counter() - t0
    # (b) the whole iteration on one core, stage by stage
    stages = {}
    whole = _cpu_inputs(n_samples, n_rays, m_rays, seed=0)
    t0 = time.perf_counter()
    _cpu_iteration(O, whole, table, n_params, grid, wd, wc, bits, npar, stages)
    t_one = time.perf_counter() - t0
    return {"value": round(frac / t_par, 4), "unit": "iters/s", "cores": cores, "kind": "port", "single_core_value": round(frac / t_one, 4), "single_core_stage_ms": stages,
            "sample": f"{'one' if frac == 1 else f'1/{int(1 / frac)} of one'} training iteration: {m_rays} rays marched, {n_samples} samples through hash fwd/bwd, SH, both MLPs fwd/bwd, "
                      f"compositing fwd/bwd, Huber; Adam+EMA on {npar} of {n_params} parameters (occupancy-grid refresh not included); "
                      f"{t_par:.1f} s on {cores} cores ({cores} concurrent slices), {t_one:.1f} s on 1 core"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--images", type=int, default=50)
    ap.add_argument("--res", type=int, default=400)
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-kernel HIP-event brackets (then no roofline object)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo lets two ranks share one GPU in tests)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run --nproc-per-node {args.gpus})"
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    from jnerf_amd import ops
    from jnerf_amd.presets import ngp_cfg
    from jnerf_amd.runner import Runner
    torch.manual_seed(1234 + rank)
    ngp_cfg(fp16=True, aabb_scale=4, const_dt=False, n_images=args.images, W=args.res, H=args.res, device=f"cuda:{local_rank}", rank=rank, world_size=world)
    runner = Runner()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = 0
    probe = min(32, args.warmup // 2)
    for _ in range(args.warmup - probe):
        runner.train_step(step); step += 1
    # last `probe` warm-up steps: HIP-event brackets around every hot-path launch to find the dominant kernel and the per-step breakdown ...
    ops.PROFILE = None if args.no_kernel_events else {}
    valid_sum = torch.zeros(1, dtype=torch.int64, device="cuda")
    for _ in range(probe):
        runner.train_step(step); step += 1
        valid_sum += runner.sampler._counters[3]            # samples in the batch just trained on (device-side count; read back once, after the run)
    torch.cuda.synchronize()
    probe_prof, ops.PROFILE = (ops.PROFILE or {}), None
    # what an empty event pair measures on this stream (the two timestamp packets themselves): subtracted from every bracket below
    pairs = []
    for _ in range(64):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record(); pairs.append((a, b))
    torch.cuda.synchronize()
    ev_overhead = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
    single = ("hash_fwd", "field_fwd", "field_bwd", "composite_fwd", "composite_bwd", "adam_ema")          # brackets that contain exactly one kernel
    breakdown = {k: sum(max(a.elapsed_time(b) - ev_overhead, 0.0) for a, b in v) / max(probe, 1) for k, v in probe_prof.items()}
    dom = max((k for k in breakdown if k in single), key=lambda k: breakdown[k]) if breakdown else None
    # ... in the timed region only that kernel keeps its bracket (an event pair per launch costs ~2-3 us; eight of them per step were ~5 %).
    # Single-GPU runs issue the step through ngp_train_step, which records the pair itself; data-parallel runs keep the Python-side bracket.
    fast = runner._fast if getattr(runner, "_fast", None) else None
    native = bool(fast and fast.native)
    ops.PROFILE_ONLY = dom
    if native:
        fast.timed_stage = None if (args.no_kernel_events or dom is None) else dom
        fast.stage_timings()                                  # drop anything recorded so far
        ops.PROFILE = None
    else:
        ops.PROFILE = None if (args.no_kernel_events or dom is None) else {}
    barrier()
    t0 = time.perf_counter()
    loss = None
    for _ in range(args.steps):
        loss = runner.train_step(step); step += 1
    barrier()
    last_loss = loss.mean().item() if loss is not None else float("nan")
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = (ops.PROFILE or {}), None
    dom_ms = [a.elapsed_time(b) for a, b in prof.get(dom, [])] if dom else []
    if native and fast.timed_stage is not None:
        dom_ms = fast.stage_timings()
        fast.timed_stage = None
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    mean_valid = float(valid_sum.item()) / max(probe, 1)          # mean samples per batch over the probe steps (the adaptive ray count keeps it within ~1 % of 2^18)

    # ---- roofline of the dominant single kernel: HIP-event durations over the timed region, algorithmic bytes per launch
    P = runner.model.pos_encoder.n_params
    alg_bytes = {"hash_fwd": mean_valid * (12 + 16 * 8 * 4 + 64), "field_fwd": mean_valid * (64 + 12 + 8), "field_bwd": mean_valid * (64 + 12 + 8 + 64),
                 "composite_fwd": mean_valid * 36, "composite_bwd": mean_valid * 44, "adam_ema": P * 34}
    roof = None
    if dom is not None and dom_ms:
        ms = dom_ms
        avg_raw = sum(ms) / len(ms)
        avg_ms = max(avg_raw - ev_overhead, 1e-6)
        nbytes = alg_bytes[dom]
        achieved = nbytes / (avg_ms * 1e-3) / 1e9
        flops = {"field_fwd": 20480.0, "field_bwd": 61440.0}.get(dom)      # per sample: 20 MFMA 16x16x32 per 16 samples forward; recompute + dgrad + wgrad backward
        traffic = None      # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
            traffic = pm.get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            pass
        kname = {"hash_fwd": "k_hash_fwd", "field_fwd": "k_field_fwd", "field_bwd": "k_field_bwd", "composite_fwd": "k_composite_fwd",
                 "composite_bwd": "k_composite_bwd", "adam_ema": "k_adam_ema"}[dom]
        hbm = {"achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4)}
        if flops is None:
            roof = dict(bound="hbm", **hbm)
        else:       # the fused MLP kernels are MFMA work (fp16 16x16x32, dense peak 2.5 PFLOP/s); their HBM side is reported next to it
            tf = flops * mean_valid / (avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4), "hbm": hbm}
        roof.update({"kernel": kname, "traffic": traffic, "avg_launch_ms": round(avg_ms, 4), "avg_launch_ms_raw": round(avg_raw, 4), "event_pair_overhead_ms": round(ev_overhead, 4), "launches_timed": len(ms), "alg_bytes_per_launch": int(nbytes),
                     "ms_per_step_by_launch_group": {k: round(v, 4) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1])}})

    extra = {"mean_samples_per_batch": round(mean_valid, 1), "rays_per_batch": runner.sampler.n_rays_per_batch}
    if breakdown:       # every single-kernel bracket against the HBM roofline, from the probe steps (algorithmic bytes / HIP-event duration)
        extra["probe_kernels"] = {k: {"avg_launch_ms": round(breakdown[k], 4), "alg_GBps": round(alg_bytes[k] / (breakdown[k] * 1e-3) / 1e9, 1),
                                      "frac_hbm": round(alg_bytes[k] / (breakdown[k] * 1e-3) / 8e12, 4)} for k in single if breakdown.get(k)}
    if not args.no_psnr and rank == 0:
        import numpy as np
        from jnerf_amd.utils.registry import build_from_cfg, DATASETS
        runner.dataset["test"] = build_from_cfg(runner.cfg.dataset.test, DATASETS)
        img, _, tar = runner.render_img("test", 0)                      # first call allocates the inference buffers
        extra["psnr_test_view_after_%d_steps" % step] = round(float(-10 * np.log10(np.mean((img - tar) ** 2))), 2)
        torch.cuda.synchronize(); tr0 = time.perf_counter()
        n_s = 0
        for v in range(4):                                              # 4 full views incl. ray generation and the device->host copy of each image
            runner.render_img("test", v % runner.dataset["test"].n_images)
            n_s += runner.n_samples_rendered
        torch.cuda.synchronize(); tr = time.perf_counter() - tr0
        extra["render_Msamples_per_s"] = round(n_s / tr / 1e6, 2)
        extra["render_ms_per_%dx%d_view" % (args.res, args.res)] = round(tr / 4 * 1e3, 2)
    if world > 1:
        # data-parallel invariant: every rank must hold bit-identical parameters (identical summed gradients + a deterministic sweep)
        from jnerf_amd import optim as _optim
        _optim.flush_all()
        enc = runner.model.pos_encoder
        sig = torch.stack([enc.m_grid.detach().double().sum(), enc.m_grid.detach().double().abs().sum(),
                           runner.model.density_mlp.con_weights.detach().double().sum(), runner.model.rgb_mlp.con_weights.detach().double().sum()])
        sigs = [torch.empty_like(sig) for _ in range(world)]
        dist.all_gather(sigs, sig)
        extra["replicas_identical"] = bool(all(torch.equal(sigs[0], x) for x in sigs))
        dist.barrier()
    if rank == 0:
        line = {"metric": "training iters/s", "value": round(world * args.steps / dt, 2), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic", "loss": round(float(last_loss), 6),
                "config": {"workload": "Instant-NGP fox config (ngp_fox.py hyper-parameters: aabb_scale 4, L=16, T=2^19, F=2, fp16 fused MLP, const_dt=False, 2^18-sample batches), "
                                       f"procedural scene {args.images}x{args.res}x{args.res} RGBA, random-init weights",
                           "samples_per_iter_per_gpu": 1 << 18, "parallelism": f"ray-batch dp{world}" if world > 1 else "single"},
                "roofline": roof, "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(), "extra": extra}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
