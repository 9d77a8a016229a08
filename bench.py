#!/usr/bin/env python
"""bench.py — Instant-NGP training throughput on MI355X.

Headline workload (BASELINE.json metric "training iters/s ... on lego"): the reference's projects/ngp/configs/ngp_base.py hyper-parameters AS COMMITTED —
aabb_scale 1 (NeRF-synthetic), const_dt = True, fp32 hash table + fp32 field network (ngp_base.py leaves `fp16` unset, models/networks/ngp_network.py:57-67),
L=16, F=2, T=2^19, Adam lr 1e-1 / eps 1e-15 / betas (0.9, 0.99), EMA 0.95, Huber 0.1, 4096 initial rays, 2^18-sample batches, occupancy refresh every 16 steps —
on a procedural 100 x 800 x 800 RGBA scene (lego itself is not on the GPU box and cannot be downloaded; SURVEY.md §8d names this stand-in), random-init weights.

One "step" = one full training iteration of Runner.train (ray generation + random background, occupancy-grid marching + compaction, hash encode, field
network, compositing, Huber, backward, fused Adam+EMA; occupancy-grid refresh on every 16th step) over one 2^18-sample batch.

Regime: a FIXED burn-in of --burn-in (1024) untimed steps brings the occupancy grid and the adaptive ray count to steady state regardless of --warmup; then
--warmup untimed steps, then exactly --steps timed steps between barrier + synchronize.  The last 32 burn-in steps run with an event pair around EVERY kernel
launch of the library (csrc/prof.hip, on each launch's own stream) to find the kernel with the largest share of GPU time; in the timed region only that kernel
keeps its bracket, and `roofline` is computed from those live durations.

`value` = (n_gpus * steps) / seconds (weak scaling: every rank trains its own 2^18-sample ray batch, hash-table / MLP gradients all-reduced over RCCL each step);
--scaling strong splits ONE 2^18-sample iteration over the ranks (2^18 / n_gpus samples per rank), value = steps / seconds.
--config fox runs the ngp_fox.py hyper-parameters (aabb 4, cone stepping, fp16 fused MLP) instead; the default run reports that configuration on the REAL fox
images (data/fox, copied from the reference checkout by __graft_entry__.build()) under `extra.fox`.  Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BURN_IN_DEFAULT = 1024
PROBE_STEPS = 32


# ---------------------------------------------------------------------------------------------------------------- CPU baseline (oracle port; rank 0, N=1 only)
def _cpu_inputs(n_samples, n_rays, m_rays, seed, aabb):
    """synthetic inputs of one (slice of a) training iteration for the CPU port"""
    import numpy as np
    import synth
    x = synth.uniform_positions(n_samples, seed=seed)
    d = synth.unit_dirs01(n_samples)
    coords = np.zeros((n_samples, 7), np.float32); coords[:, :3] = x; coords[:, 4:] = d
    per = n_samples // n_rays
    ns = np.stack([np.full(n_rays, per, np.uint32), (np.arange(n_rays) * per).astype(np.uint32)], 1)
    bg = np.random.default_rng(seed).random((n_rays, 3), dtype=np.float32)
    xf, focal, meta = synth.camera_ring(8, radius=1.3)
    _, ro, rd, _ = synth.rays_from_cameras(xf, focal, meta, 400, 400, m_rays, seed=1 + seed)
    return dict(x=x, d=d, coords=coords, ns=ns, bg=bg, ro=ro, rd=rd, aabb=aabb)


def _cpu_iteration(O, inp, table, n_params, grid, wd, wc, bits, npar, fp16, const_dt, stages=None, R=None, offsets=None, aabb_scale=1):
    """every hot-path stage of one iteration on the host (ctypes releases the GIL inside each call).  R = oracle/_ref (the reference's own kernel headers compiled for the
    host): used for the stages it has - marcher, hash encode forward / backward, compositing forward / backward; the port (oracle/ngp_oracle.c) does the rest - SH, the
    two MLPs (binary-only in the reference), Adam + EMA (Jittor's) - and everything when R is None."""
    import numpy as np
    if R is not None:
        return _cpu_iteration_ref(O, R, inp, table, offsets, aabb_scale, n_params, grid, wd, wc, bits, npar, fp16, const_dt, stages)

    def timed(name, fn):
        ts = time.perf_counter()
        r = fn()
        if stages is not None:
            stages[name] = round((time.perf_counter() - ts) * 1e3, 1)
        return r

    x, d, coords, ns, bg = inp["x"], inp["d"], inp["coords"], inp["ns"], inp["bg"]
    timed("march", lambda: O.march_rays(inp["ro"], inp["rd"], bits, inp["aabb"], O.PCG32(1337), 4096 * 1024, const_dt=const_dt))
    feat = timed("hash_fwd", lambda: O.hash_encode_fwd(x, grid, table))
    sh = timed("sh", lambda: O.sh_encode(d, np.float32))
    out = timed("field_fwd", lambda: O.field_fwd(feat.astype(np.float32), sh, wd, wc))
    rgb = timed("composite_fwd", lambda: O.composite_fwd(out, coords, ns, ns, bg))
    _, G = O.huber(rgb, bg)
    dout = timed("composite_bwd", lambda: O.composite_bwd(out, coords, ns, G, rgb, 0.001))
    dfeat, dwd, dwc = timed("field_bwd", lambda: O.field_bwd(feat.astype(np.float32), sh, wd, wc, dout))
    g = timed("hash_bwd", lambda: O.hash_encode_bwd(x, dfeat.astype(np.float16 if fp16 else np.float32), table, n_params))
    p = np.zeros(npar, np.float32); m = np.zeros_like(p); v = np.zeros_like(p); e = np.zeros_like(p)
    timed("adam_ema", lambda: O.adam_ema_step(p, g[:npar].astype(np.float32), m, v, e, 0.1, 1))


def _cpu_iteration_ref(O, R, inp, table, offsets, aabb_scale, n_params, grid, wd, wc, bits, npar, fp16, const_dt, stages):
    import numpy as np

    def timed(name, fn):
        ts = time.perf_counter()
        r = fn()
        if stages is not None:
            stages[name] = round((time.perf_counter() - ts) * 1e3, 1)
        return r

    x, d, coords, ns, bg, aabb = inp["x"], inp["d"], inp["coords"], inp["ns"], inp["bg"], inp["aabb"]
    n_rays = inp["ro"].shape[0]
    meta = np.zeros((1, 11), np.float32); ids = np.zeros(n_rays, np.uint32); xf = np.zeros((1, 4, 3), np.float32)
    timed("march", lambda: R.march(inp["ro"], inp["rd"], bits, aabb, R.PCG32(1337).st, n_rays * 1024, meta, ids, xf, const_dt=const_dt))
    feat = timed("hash_fwd", lambda: R.hash_fwd(x, grid, offsets, aabb_scale))
    sh = timed("sh", lambda: O.sh_encode(d, np.float32))
    out = timed("field_fwd", lambda: O.field_fwd(np.asarray(feat, np.float32), sh, wd, wc))
    T = np.float16 if fp16 else np.float32
    rgb = timed("composite_fwd", lambda: R.rgb_fwd(out.astype(T), coords, ns, ns, bg, aabb))
    _, G = O.huber(rgb, bg)
    dout = timed("composite_bwd", lambda: R.rgb_bwd(out.astype(T), coords, ns, G, rgb, 0.001, aabb))
    dfeat, dwd, dwc = timed("field_bwd", lambda: O.field_bwd(np.asarray(feat, np.float32), sh, wd, wc, np.asarray(dout, np.float32)))
    g = timed("hash_bwd", lambda: R.hash_bwd(x, dfeat.astype(T), offsets, aabb_scale, n_params))
    p = np.zeros(npar, np.float32); m = np.zeros_like(p); v = np.zeros_like(p); e = np.zeros_like(p)
    timed("adam_ema", lambda: O.adam_ema_step(p, np.asarray(g[:npar], np.float32), m, v, e, 0.1, 1))


def _ref_stage_times(O, table, offsets, n_params, grid, bits, aabb, aabb_scale, fp16, const_dt, n=1 << 15, n_rays=512):
    """The reference's OWN kernels (oracle/_ref: its op_header/*.h compiled for the host by oracle/ref_shim, serial launcher) timed on one core for the stages that exist
    as compilable source - marcher, hash encode forward / backward, compositing forward / backward - on a bounded sample, with the port (oracle/ngp_oracle.c) timed on
    the SAME inputs beside it.  The fused MLP (binary-only in the reference) and Adam (Jittor's) have no reference build: the whole-iteration `value` stays the port's."""
    import numpy as np
    try:
        from oracle import ref as R
        if not R.available():
            return {"skipped": "oracle/_ref is not built on this host (it is built where /root/reference exists and travels with the snapshot)"}
    except Exception as e:                     # pragma: no cover
        return {"skipped": repr(e)}
    import synth
    inp = _cpu_inputs(n, n_rays, n_rays, seed=11, aabb=aabb)
    x, coords, ns, bg = inp["x"], inp["coords"], inp["ns"], inp["bg"]
    T = np.float16 if fp16 else np.float32
    dy = (np.random.default_rng(5).standard_normal((n, 32)) * 1e-3).astype(T)
    out = np.random.default_rng(6).standard_normal((n, 4)).astype(T)
    meta = np.zeros((1, 11), np.float32); ids = np.zeros(n_rays, np.uint32); xf = np.zeros((1, 4, 3), np.float32)
    ref_ms, port_ms = {}, {}

    def t(d, name, fn):
        ts = time.perf_counter(); r = fn(); d[name] = round((time.perf_counter() - ts) * 1e3, 2); return r
    t(ref_ms, "march", lambda: R.march(inp["ro"], inp["rd"], bits, aabb, R.PCG32(1337).st, 4096 * 1024, meta, ids, xf, const_dt=const_dt))
    t(port_ms, "march", lambda: O.march_rays(inp["ro"], inp["rd"], bits, aabb, O.PCG32(1337), 4096 * 1024, const_dt=const_dt))
    t(ref_ms, "hash_fwd", lambda: R.hash_fwd(x, grid, offsets, aabb_scale))
    t(port_ms, "hash_fwd", lambda: O.hash_encode_fwd(x, grid, table))
    t(ref_ms, "hash_bwd", lambda: R.hash_bwd(x, dy, offsets, aabb_scale, n_params))
    t(port_ms, "hash_bwd", lambda: O.hash_encode_bwd(x, dy, table, n_params))
    rgb = t(ref_ms, "composite_fwd", lambda: R.rgb_fwd(out, coords, ns, ns, bg, aabb))
    t(port_ms, "composite_fwd", lambda: O.composite_fwd(out, coords, ns, ns, bg))
    t(ref_ms, "composite_bwd", lambda: R.rgb_bwd(out, coords, ns, (rgb - bg).astype(np.float32), rgb, 0.001, aabb))
    t(port_ms, "composite_bwd", lambda: O.composite_bwd(out, coords, ns, (rgb - bg).astype(np.float32), rgb, 0.001))
    return {"kind": "reference", "cores": 1, "sample": f"{n_rays} rays marched through the shell bitfield; {n} samples ({n_rays} rays x {n // n_rays}) through hash encode fwd/bwd and compositing fwd/bwd",
            "stage_ms": ref_ms, "port_stage_ms_same_inputs": port_ms}


def cpu_baseline(aabb_scale, fp16, const_dt, n_samples=1 << 18, n_rays=4096, n_march_rays=4096):
    """The oracle (plain-C port of the reference's kernels) on ONE training iteration of the bench workload's shape, on this host's cores: the iteration is split
    into `cores` independent ray/sample slices that run concurrently (one thread each; the hash-table gradient and the parameter sweep are sliced the same
    way), plus the same iteration on a single core for the per-stage times.  `value` is iterations/s on `cores` cores."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    import synth
    frac = n_samples / float(1 << 18)
    table, offsets, n_params = O.level_table(aabb_scale)
    grid = synth.table(n_params, np.float16 if fp16 else np.float32, amp=2e-4)
    wd, wc = synth.mlp_weights()
    bits = synth.shell_bitfield()
    aabb = (0.5 - aabb_scale / 2, 0.5 + aabb_scale / 2)
    m_rays = int(n_march_rays * frac)
    npar = int(n_params * frac) // 4 * 4
    cores = max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    # (r4) the reference's own kernels wherever they exist as source: the port was 2.2x slower than them on the hash forward
    R = None
    try:
        from oracle import ref as _R
        if _R.available():
            R = _R
    except Exception:
        R = None
    parts = [_cpu_inputs(n_samples // cores, max(n_rays // cores, 1), max(m_rays // cores, 1), seed=k, aabb=aabb) for k in range(cores)]
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda inp: _cpu_iteration(O, inp, table, n_params, grid, wd, wc, bits, npar // cores // 4 * 4, fp16, const_dt, None, R, offsets, aabb_scale), parts))
        t_par = time.perf_counter() - t0
    stages = {}
    whole = _cpu_inputs(n_samples, n_rays, m_rays, seed=0, aabb=aabb)
    t0 = time.perf_counter()
    _cpu_iteration(O, whole, table, n_params, grid, wd, wc, bits, npar, fp16, const_dt, stages, R, offsets, aabb_scale)
    t_one = time.perf_counter() - t0
    return {"value": round(frac / t_par, 4), "unit": "iters/s", "cores": cores, "kind": "reference+port" if R is not None else "port",
            "kind_detail": ("marcher, hash encode fwd/bwd, compositing fwd/bwd: the reference's own kernel headers compiled for the host (oracle/_ref); SH, both MLPs, Huber, Adam+EMA: "
                            "the plain-C port (oracle/ngp_oracle.c) - the reference has no source for them" if R is not None else "plain-C port (oracle/ngp_oracle.c): oracle/_ref is not built on this host"),
            "single_core_value": round(frac / t_one, 4), "single_core_stage_ms": stages,
            "reference_kernels": _ref_stage_times(O, table, offsets, n_params, grid, bits, aabb, aabb_scale, fp16, const_dt),
            "sample": f"one training iteration of the bench workload's shape ({'fp16' if fp16 else 'fp32'} table, aabb_scale {aabb_scale}, const_dt {const_dt}): {m_rays} rays marched through a "
                      f"shell bitfield, {n_samples} samples through hash fwd/bwd, SH, both MLPs fwd/bwd, compositing fwd/bwd, Huber; Adam+EMA on {npar} of {n_params} parameters "
                      f"(occupancy-grid refresh not included); {t_par:.1f} s on {cores} cores ({cores} concurrent slices), {t_one:.1f} s on 1 core"}


# ---------------------------------------------------------------------------------------------------------------- roofline bookkeeping
def alg_bytes_table(n, P, R, n_refresh, fp16, n_runs=10, run_param_frac=0.1):
    """ALGORITHMIC bytes (and flops) per launch of every hot-path kernel: per-unit figure x units one launch processes.
    n = samples in the batch, P = hash-table parameters, R = rays in the batch, n_refresh = points of one occupancy-grid refresh launch, n_runs = levels with
    res <= 300 (k_bin_records_runs; the other 16 - n_runs go through k_bin_records), T = bytes per table / feature element."""
    T = 2 if fp16 else 4
    hf = 12 + 16 * 8 * 2 * T + 32 * T                      # hash fwd per sample: pos + 128 corner values + 32 features out (588 | 1164)
    fio = 32 * T + 12 + 4 * T                              # field fwd per sample: features + direction + 4 outputs
    d = {
        "k_hash_fwd": n * hf, "k_field_fwd": n * fio, "k_field_bwd": n * (fio + 32 * T),
        "k_field32_fwd": n * fio, "k_field32_bwd": n * (fio + 32 * T),
        "k_composite_fwd": n * (4 * T + 28), "k_composite_bwd": n * (4 * T + 28 + 4 * T), "k_composite_train": n * (4 * T + 28) + n * (4 * T + 28 + 4 * T),       # (the fused launch does both passes' algorithmic work)
        "k_adam_ema": P * (30 if fp16 else 28),                # read p, g, m, v; write p, m, v (+ the fp16 shadow); the gradient is overwritten by the next backward, not zeroed here
        # hash backward: the stage's necessary traffic is pos 12 + dL/dy 32*T + 128 scattered fp32 updates (4 B each as one write); attributed to the kernels that do each part
        "k_level_absmax": n * 32 * T, "k_bin_records": n * (12 + 32 * T) * (16 - n_runs) / 16, "k_bin_records_runs": n * (12 + 32 * T) * n_runs / 16, "k_bin_accumulate": n * 16 * 8 * 2 * 4,
        "k_bin_pairs": n * (12 + 32 * T) * (16 - n_runs) / 16, "k_bin_runs2": n * (12 + 32 * T) * n_runs / 16, "k_bin_accumulate2": n * 16 * 8 * 2 * 4,
        "k_bin_accumulate2_adam": n * 16 * 8 * 2 * 4 + P * 24,        # (r6) the accumulate with the table's sweep riding: + read p, m, v, write p, m, v; the gradient never exists in HBM
        # (r6) per-corner path (fp16 configuration), two launches: run levels (fp32 records; run_param_frac of the parameters) | fine levels (fp16 records); + the 2-byte shadow
        "k_bin_accumulate_adam_f32rec": n * 8 * 2 * 4 * n_runs + P * run_param_frac * (24 + (2 if fp16 else 0)),
        "k_bin_accumulate_adam_f16rec": n * 8 * 2 * 4 * (16 - n_runs) + P * (1 - run_param_frac) * (24 + (2 if fp16 else 0)),
        "k_reduce_slabs": 10240 * 4, "k_reduce_slabs_sweep": 10240 * (4 + 30), "k_pack_frags": 21504 * 2 * 2,
        # sampling: ray in (24 B) + one 28-byte record and one 12-byte position out per sample
        "k_march_count": R * 24 + n * 4, "k_march_wave": R * 24 + n * 4, "k_mscan_totals": R * 4, "k_mscan_ok": R * 4, "k_mscan_final": R * 20, "k_march_write_cached": n * (4 + 40),
        "k_generate_rays": R * (8 + 16 + 12 + 40),
        # occupancy refresh (one launch over all points)
        "k_grid_generate": n_refresh * (4 + 16), "k_grid_splat": n_refresh * (4 + T + 4), "k_grid_ema": 5 * 128 ** 3 * 12, "k_grid_mean": 128 ** 3 * 4, "k_grid_to_bitfield": 5 * 128 ** 3 * 4.125,
        "k_bitfield_max_pool": 128 ** 3 / 8 * 1.125, "k_refresh_fused": n_refresh * (4 + 16 * 8 * 2 * T + 4),
    }
    # 20 480 FLOP / sample forward, 40 960 backward.  (The backward kernels also RECOMPUTE the forward - 61 440 executed - which is the kernel's own
    # choice, not algorithmic work: `roofline.frac` is on the §8(d) figure, the executed figure is reported next to it.)
    flops = {"k_field_fwd": 20480.0 * n, "k_field_bwd": 40960.0 * n, "k_field32_fwd": 20480.0 * n, "k_field32_bwd": 40960.0 * n}
    for alias, base in KERNEL_VARIANTS.items():         # the same work under another kernel name (csrc variants selected by environment / default since r3)
        if base in d:
            d[alias] = d[base]
        if base in flops:
            flops[alias] = flops[base]
    return d, flops


# kernel variants: same algorithmic bytes / flops as the kernel they replace
KERNEL_VARIANTS = {"k_field32_bwd_2g": "k_field32_bwd", "k_hash_fwd_dydx": "k_hash_fwd", "k_hash_fwd_x2": "k_hash_fwd",
                   "k_field32_fwd_split": "k_field32_fwd", "k_field32_bwd_split": "k_field32_bwd"}
# kernels that do fp32-accurate work on the fp16 matrix cores (split operands, three v_mfma_f32_16x16x32_f16 per product sum, csrc/field_split.hip): the algorithmic
# FLOPs are SURVEY.md §8(d)'s, the pipe they run on peaks at 2.5 PFLOP/s dense, and they execute 3x the products
SPLIT_FP16_KERNELS = ("k_field32_fwd_split", "k_field32_bwd_split")
# FLOP per sample the kernels EXECUTE: the backward kernels recompute the forward (their choice, not algorithmic work); the r3 fp32 variants skip the rgb layer the backward never reads
EXECUTED_FLOPS_PER_SAMPLE = {"k_field_fwd": 20480.0, "k_field_bwd": 61440.0, "k_field32_fwd": 20480.0, "k_field32_bwd": 61440.0, "k_field32_bwd_2g": 59392.0,
                             "k_field32_fwd_split": 3 * 20480.0, "k_field32_bwd_split": 3 * 59392.0}
# the hash-backward stage's kernels: round 3's per-corner records (the fp16 configuration still) | round 4's region records of the fp32 configuration
HASH_BWD_STAGE = ("k_level_absmax", "k_bin_records_runs", "k_bin_records", "k_bin_accumulate", "k_bin_runs2", "k_bin_pairs", "k_bin_accumulate2", "k_bin_accumulate2_adam",
                  "k_bin_accumulate_adam_f32rec", "k_bin_accumulate_adam_f16rec")
HASH_BWD_ACC = ("k_bin_accumulate", "k_bin_accumulate2", "k_bin_accumulate2_adam")        # kernels of the stage that are launched more than once per step (per-launch average x launches)
HASH_BWD_RIDE = ("k_bin_accumulate2_adam", "k_bin_accumulate_adam_f32rec", "k_bin_accumulate_adam_f16rec")



PMC_ROUND = "r06z"          # the round whose counter passes describe THIS tree's kernels (profiles/<round>_pmc.json): taken before the last session of round 6, which changed
                            # only the field kernels' register arithmetic (same loads, stores and matrix instructions; the hash / record / accumulate kernels untouched)


def pmc_lookup(config, scene):
    """per-kernel counter entries of this round's committed --pmc passes for (config, scene), and where they came from; ({}, None) if there is no such pass"""
    path = os.path.join(ROOT, "profiles", f"{PMC_ROUND}_pmc.json")
    try:
        pm = json.load(open(path))
    except Exception:
        return {}, None
    if pm.get("_scenes", {}).get(config) != scene or not pm.get(config):
        return {}, None
    return pm[config], f"profiles/{PMC_ROUND}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `{pm.get('_commands', {}).get(config, '?')}`: this round's hash / record / accumulate kernels as they are in this tree, this scene; not measured in this run)"


def mfma_roofline(k, alg_flops, n, ms, fp16, counters=None):
    """The MFMA-side roofline of a fused field kernel, against the pipe its instructions ISSUE on.
    fp16 configuration / exact-product fp32 kernels: algorithmic FLOPs over the dense peak of their own MFMA (2.5 PFLOP/s fp16 16x16x32 | 157.3 TFLOP/s fp32 16x16x4).
    Split-operand kernels (fp32-accurate products from three v_mfma_f32_16x16x32_f16 each, csrc/field_split.hip): they run on the fp16 pipe, so `frac` is what they
    issue there - 3 x the fp32-equivalent products incl. the backward's forward recompute - over 2.5 PFLOP/s (the fp32-MFMA peak is not a ceiling for
    them - the forward would print 1.06 of it).  The fp32-equivalent rate and its ratio to the fp32 MFMA peak stay as the secondary `fp32_equivalent`."""
    counters = counters or {}
    sec = ms * 1e-3
    split = k in SPLIT_FP16_KERNELS
    tf_alg = alg_flops / sec / 1e12
    exec_fps = EXECUTED_FLOPS_PER_SAMPLE[k]                     # FLOP per sample the kernel issues on its pipe (split: 3 fp16 products per fp32-accurate product)
    tf_issued = exec_fps * n / sec / 1e12
    if split:
        r = {"bound": "mfma", "achieved": round(tf_issued, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf_issued / 2500.0, 4),
             "pipe": "fp16 MFMA (v_mfma_f32_16x16x32_f16), split operands: `achieved` = FLOPs ISSUED on that pipe (3 products per fp32-accurate product, incl. the backward's forward recompute)",
             "alg_frac_of_pipe": round(tf_alg / 2500.0, 4),         # SURVEY.md 8(d)'s fp32 FLOPs over the pipe the kernel occupies
             "frac_of_split_ceiling": round(tf_alg / (2500.0 / 3.0), 4),   # the technique's own ceiling: a third of the fp16 pipe in fp32-equivalent products
             "fp32_equivalent": {"achieved": round(tf_alg, 1), "peak": 157.3, "unit": "TFLOP/s", "ratio": round(tf_alg / 157.3, 4),
                                 "note": "fp32-accurate algorithmic FLOPs (SURVEY.md 8d) vs the fp32 MFMA peak the configuration would otherwise be limited by - a comparison, NOT a ceiling of this kernel (it can exceed 1)"}}
    else:
        peak = 2500.0 if fp16 else 157.3
        r = {"bound": "mfma", "achieved": round(tf_alg, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf_alg / peak, 4),
             "executed_frac": round(tf_issued / peak, 4)}         # counts the in-kernel forward recompute of the backward as work
    r.update({"alg_flop_per_sample": alg_flops / n, "issued_flop_per_sample": exec_fps,
              "issued_frac_counters": counters.get("mfma_issued_frac"), "pipe_util_counters": counters.get("mfma_pipe_util")})   # this round's MFMA-counter pass, null if absent
    return r


def step_algorithmic_bytes(n, P, fp16):
    """SURVEY.md 8(d) / DESIGN.md 4: algorithmic bytes of ONE training iteration - every logically required byte once.
    per sample: hash fwd 12 + 128 T2 + 32 T | field fwd 32 T + 12 + 4 T | compositing fwd 4 T + 28, bwd 8 T + 28 | field bwd 64 T + 12 + 4 T | hash bwd 12 + 32 T + 128 T2
    | marcher record + compact position 40;   per parameter: 28 (fp32) | 30 (fp16: + the shadow)          (T = 2 | 4 bytes, T2 = one 2-feature entry)"""
    T = 2 if fp16 else 4
    per_sample = (12 + 128 * 2 * T + 32 * T) + (36 * T + 12) + (4 * T + 28) + (8 * T + 28) + (68 * T + 12) + (12 + 32 * T + 128 * 2 * T) + 40
    return n * per_sample + P * (30 if fp16 else 28), per_sample


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--burn-in", type=int, default=BURN_IN_DEFAULT, help="fixed number of untimed steps BEFORE --warmup (occupancy grid + adaptive ray count reach steady state)")
    ap.add_argument("--config", default="lego", choices=["lego", "fox"], help="lego = ngp_base.py hyper-parameters (headline); fox = ngp_fox.py hyper-parameters")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--no-fox", action="store_true", help="skip the real-fox leg (extra.fox)")
    ap.add_argument("--no-neus", action="store_true", help="skip the NeuS leg (extra.neus: BASELINE configs[4] on the procedural DTU-layout scene)")
    ap.add_argument("--no-lego-gate", action="store_true", help="skip extra.lego_gate (the 40 000-step NeRF-synthetic lego run + test PSNR; only runs when the data set is mounted)")
    ap.add_argument("--scene", default=None, choices=["bricks", "spheres"], help="procedural scene.  lego config default: bricks = the lego-difficulty stand-in (2.5 %% occupied cells, hard edges, "
                    "thin parts, 35.5 dB after the full schedule; VERDICT r3) - spheres (four soft spheres, > 47 dB: rounds 1-3's headline scene) is reported as extra.spheres; fox config default: spheres")
    ap.add_argument("--no-spheres", action="store_true", help="skip the extra.spheres leg of the lego line")
    ap.add_argument("--images", type=int, default=0)
    ap.add_argument("--res", type=int, default=0)
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-kernel HIP-event brackets (then no roofline object)")
    ap.add_argument("--force-dist", action="store_true", help="with --gpus 1: still create the process group and run the data-parallel sequence (RCCL all-reduce at world size 1)")
    ap.add_argument("--dp-overlap", action="store_true", help="data parallel: two gradient buckets, the coarse levels' reduce-scatter on the library's communication stream under the fine levels' accumulate")
    ap.add_argument("--dp-host-sharded", action="store_true", help="(tests, --backend gloo) the sharded sweep without RCCL: the host sums the gradient and gathers the updated shards")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo lets two ranks share one GPU in tests)")
    ap.add_argument("--no-second-scaling", action="store_true", help="--gpus > 1: skip the second timed region in the OTHER scaling mode (extra.dp.other_scaling)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched the way the driver launches N = 1 (`python bench.py --gpus N ...`, no torchrun): spawn the N ranks ourselves, one process per GPU, and hand their
        # output through (rank 0 prints the JSON line)
        import socket
        import subprocess
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run --nproc-per-node {args.gpus}, or without WORLD_SIZE in the environment)"
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    from jnerf_amd import ops
    from jnerf_amd.presets import ngp_cfg
    from jnerf_amd.runner import Runner
    torch.manual_seed(1234 + rank)
    lego = args.config == "lego"
    fp16, aabb_scale, const_dt = (False, 1, True) if lego else (True, 4, False)
    n_images = args.images or (100 if lego else 50)
    res = args.res or (800 if lego else 400)
    share = world if args.scaling == "strong" else 1
    scene = args.scene or ("bricks" if lego else "spheres")
    ngp_cfg(scene=scene, fp16=fp16, aabb_scale=aabb_scale, const_dt=const_dt, n_images=n_images, W=res, H=res, device=f"cuda:{local_rank}", rank=rank, world_size=world,
            target_batch_size=(1 << 18) // share, n_rays_per_batch=4096 // share, dp_force_collectives=bool(args.force_dist), dp_overlap=bool(args.dp_overlap), dp_host_sharded=bool(args.dp_host_sharded),
            **json.loads(os.environ.get("BENCH_EXTRA_CFG", "{}")))       # probe hook: extra config keys as JSON, e.g. {"pipeline_sampling": false}
    runner = Runner()
    import contextlib
    on_stream = contextlib.ExitStack()
    on_stream.enter_context(runner.training_stream())       # the loop of Runner.train runs on a stream of its own (runner.py); left (synchronised) after the timed region

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    step = 0
    probe = 0 if args.no_kernel_events else min(PROBE_STEPS, args.burn_in // 2)
    for _ in range(args.burn_in - probe):
        runner.train_step(step); step += 1
    # ---- last `probe` burn-in steps: an event pair around EVERY kernel launch of the library (each on its own stream)
    if probe:
        ops.prof_enable("*")
    valid_sum = torch.zeros(1, dtype=torch.int64, device="cuda")
    rays_sum = 0
    for _ in range(probe):
        runner.train_step(step); step += 1
        valid_sum += runner.sampler._counters[3]            # samples in the batch just trained on (device-side count; read back once, after the run)
        rays_sum += int(runner.sampler._rays_numsteps.shape[0])
    ops.prof_enable("")
    probe_ms = ops.prof_read() if probe else {}
    # what an empty event pair measures on a stream (the two timestamp packets themselves): subtracted from every bracket
    pairs = []
    for _ in range(64):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record(); pairs.append((a, b))
    torch.cuda.synchronize()
    ev_overhead = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
    mean_valid = float(valid_sum.item()) / max(probe, 1) if probe else float(1 << 18) / share
    mean_rays = rays_sum / max(probe, 1) if probe else float(runner.sampler.n_rays_per_batch)
    per_step = {k: sum(max(x - ev_overhead, 0.0) for x in v) / probe for k, v in probe_ms.items()} if probe else {}
    # one kernel can serve launches of very different sizes (k_hash_fwd: the training batch vs the occupancy-grid refresh; k_adam_ema: the table vs the weight
    # packs): durations are clustered (a gap of more than 2x starts a new class) and the roofline uses the class that carries most of the kernel's time
    def batch_class(v):
        s = sorted(v)
        classes, cur = [], [s[0]]
        for x in s[1:]:
            if x > 2.0 * cur[0]:
                classes.append(cur); cur = [x]
            else:
                cur.append(x)
        classes.append(cur)
        return max(classes, key=lambda c: (len(c), sum(c)))     # the size class with the most launches (ties: the one that carries more time) - a few slow outliers must not win
    dom = max(per_step, key=lambda k: per_step[k]) if per_step else None
    # ---- warm-up + timed region: only the dominant kernel keeps its bracket
    for _ in range(args.warmup):
        runner.train_step(step); step += 1
    if dom is not None:
        ops.prof_enable(dom)
    barrier()
    t0 = time.perf_counter()
    loss = None
    for _ in range(args.steps):
        loss = runner.train_step(step); step += 1
    barrier()
    dt = time.perf_counter() - t0
    last_loss = loss.mean().item() if loss is not None else float("nan")
    runner.finish()                         # drain + the final (collective, synchronous) poll of the split kernels' range flag
    on_stream.close()
    torch.cuda.synchronize()
    ops.prof_enable("")
    dom_ms = ops.prof_read().get(dom, []) if dom is not None else []
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- roofline of the dominant kernel: live HIP-event durations over the timed region, algorithmic bytes per launch
    P = runner.model.pos_encoder.n_params
    n_refresh = 128 ** 3 * (runner.sampler.max_cascade + 1) // 2
    _lt = runner.model.pos_encoder.level_table.reshape(16, 4)
    n_runs = int((_lt[:, 2] <= 300).sum())
    run_param_frac = float(_lt[_lt[:, 2] <= 300, 1].sum()) / max(float(_lt[:, 1].sum()), 1.0)
    alg, flops = alg_bytes_table(mean_valid, P, mean_rays, n_refresh, fp16, n_runs, run_param_frac)
    roof = None
    # (the committed counter passes describe the BENCH workload: default image count and resolution - a run with --images / --res is another workload and gets no traffic figure)
    pmc, traffic_source = pmc_lookup(args.config, scene) if not (args.images or args.res) else ({}, None)
    if dom is not None and dom_ms:
        cls = batch_class(dom_ms)
        avg_raw = sum(cls) / len(cls)
        avg_ms = max(avg_raw - ev_overhead, 1e-6)
        nbytes = float(alg.get(dom, 0.0))
        achieved = nbytes / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch: rocprofv3 cannot run inside this process, so this is a LOOK-UP in the committed --pmc FETCH_SIZE / WRITE_SIZE passes of this round's tree
        # (tools/collect_profiles.sh -> profiles/<PMC_ROUND>_pmc.json).  Only a pass of THIS round, THIS configuration and THIS scene counts: anything else is
        # another kernel generation or another workload, and `traffic` is null.
        counters = pmc.get(dom, {})
        traffic = counters.get("hbm_bytes_per_launch")
        if traffic is None:
            traffic_source = None if not pmc else traffic_source + " - no entry for this kernel"
        hbm = {"achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4)}
        if dom in flops:    # the fused field kernels are MFMA work (fp16 16x16x32: dense peak 2.5 PFLOP/s; fp32 16x16x4: 157.3 TFLOP/s); their HBM side is reported next to it
            roof = mfma_roofline(dom, flops[dom], mean_valid, avg_ms, fp16, counters)
            roof["hbm"] = hbm
        else:
            roof = dict(bound="hbm", **hbm)
        # the hash-backward STAGE is four launches under four kernel names (abs-max, two record passes, accumulate): one roofline for the stage, from the probe steps
        stage = None
        st_kernels = [k for k in HASH_BWD_STAGE if k in probe_ms]      # (k_level_absmax is absent when the field backward kernel's epilogue computes the maxima)
        if any(k in st_kernels for k in HASH_BWD_ACC + HASH_BWD_RIDE):
            st_ms = sum(max(sum(batch_class(probe_ms[k])) / len(batch_class(probe_ms[k])) - ev_overhead, 0.0) * (len(probe_ms[k]) / probe if k in HASH_BWD_ACC else 1.0) for k in st_kernels)
            T_ = 2 if fp16 else 4
            st_bytes = mean_valid * (12 + 32 * T_ + 128 * 2 * T_)                  # pos + dL/dy + 128 scattered table-element updates per sample (588 | 1164 B)
            rides = any(k in st_kernels for k in HASH_BWD_RIDE)                     # (r6) the table's Adam + EMA sweep is done by the stage's accumulate kernel(s): its bytes belong to the stage
            if rides:
                st_bytes += P * (26 if fp16 else 24)
            st_traffic = [pmc.get(k, {}).get("hbm_bytes_per_launch") for k in st_kernels]
            st_traffic = None if (not st_traffic or any(x is None for x in st_traffic)) else int(sum(st_traffic))
            stage = {"name": "hash_backward+table_sweep" if rides else "hash_backward", "kernels": st_kernels, "ms": round(st_ms, 4), "alg_bytes": int(st_bytes), "achieved": round(st_bytes / (st_ms * 1e-3) / 1e9, 1),
                     "peak": 8000.0, "unit": "GB/s", "frac": round(st_bytes / (st_ms * 1e-3) / 8e12, 4),
                     # counter bytes of the stage's launches (this round's --pmc passes) over its algorithmic bytes: > 1 = records written and read back, re-reads
                     "traffic": st_traffic, "traffic_ratio": None if st_traffic is None else round(st_traffic / st_bytes, 3)}
        # the WHOLE iteration against HBM - the figure SURVEY.md 8(d) asks for: algorithmic bytes of one iteration over the measured step time (timed region, wall clock)
        step_bytes, per_sample_bytes = step_algorithmic_bytes(mean_valid, P, fp16)
        step_ms = dt / args.steps * 1e3
        step_traffic = None
        if pmc:     # counter bytes of every library launch of one step: per-launch figure x launches per step (probe steps); kernels the pass does not list make it null
            parts = [(pmc.get(k, {}).get("hbm_bytes_per_launch"), len(v) / probe) for k, v in probe_ms.items() if per_step.get(k, 0.0) > 0.002]
            step_traffic = None if any(x is None for x, _ in parts) else int(sum(x * c for x, c in parts))
        roof["step"] = {"alg_bytes": int(step_bytes), "alg_bytes_per_sample": per_sample_bytes, "bytes_per_parameter": 30 if fp16 else 28, "ms": round(step_ms, 4),
                        "GBps": round(step_bytes / (step_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "frac_hbm": round(step_bytes / (step_ms * 1e-3) / 8e12, 4),
                        "ceiling_iters_per_s": round(8e12 / step_bytes, 1), "traffic": step_traffic,
                        "traffic_ratio": None if step_traffic is None else round(step_traffic / step_bytes, 3)}
        roof.update({"kernel": dom, "traffic": traffic, "traffic_source": traffic_source, "stage": stage, "avg_launch_ms": round(avg_ms, 4), "avg_launch_ms_raw": round(avg_raw, 4), "event_pair_overhead_ms": round(ev_overhead, 4),
                     "launches_timed": len(cls), "launches_other_size_class": len(dom_ms) - len(cls), "alg_bytes_per_launch": int(nbytes),
                     "share_of_kernel_time": round(per_step[dom] / max(sum(per_step.values()), 1e-9), 4),
                     "ms_per_step_by_kernel": {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])}})

    extra = {"mean_samples_per_batch": round(mean_valid, 1), "rays_per_batch": runner.sampler.n_rays_per_batch, "burn_in_steps": args.burn_in, "probe_steps": probe,
             "native_step": bool(getattr(runner, "_fast", None) and runner._fast.native), "fast_path": bool(getattr(runner, "_fast", None))}
    # fingerprint of the trained parameters (double-precision sums): equal between a non-distributed run and the world-size-1 data-parallel run of the same seed in
    # fp32 mode - the exchange step adds no arithmetic (tests/test_train_gpu.py)
    from jnerf_amd import optim as _optim0
    _optim0.flush_all(); _optim0.sync_all_sharded()
    extra["param_signature"] = [float(p.detach().double().sum().item()) for p in runner.model.parameters()] + [float(p.detach().double().abs().sum().item()) for p in runner.model.parameters()]
    if probe_ms:        # every kernel against its roofline, from the probe steps (training-batch size class)
        pk = {}
        for k, v in probe_ms.items():
            cls = batch_class(v)
            ms = max(sum(cls) / len(cls) - ev_overhead, 1e-6)
            row = {"avg_launch_ms": round(ms, 4), "launches_per_step": round(len(v) / probe, 2), "alg_GBps": round(alg.get(k, 0.0) / (ms * 1e-3) / 1e9, 1),
                   "frac_hbm": round(alg.get(k, 0.0) / (ms * 1e-3) / 8e12, 4)}
            if row["frac_hbm"] > 1.0:
                # a small kernel whose whole working set (the 42 MB occupancy grid, say) is still in the 256 MB Infinity Cache from the launch before it moves its algorithmic
                # bytes faster than HBM could: HBM is not its roofline, and no fraction of it is printed (the rate stays in alg_GBps)
                row["frac_hbm"] = None
                row["served_from_cache"] = True
            if k in flops:
                mr = mfma_roofline(k, flops[k], mean_valid, ms, fp16)
                row.update({"TFLOPs": mr["achieved"], "mfma_peak": mr["peak"], "frac_mfma": mr["frac"]})
                if k in SPLIT_FP16_KERNELS:
                    row["pipe"] = "fp16 MFMA, split operands: TFLOPs = issued on the fp16 pipe (3 products per fp32-accurate product)"
                    row["fp32_equivalent_TFLOPs"] = mr["fp32_equivalent"]["achieved"]
            if pmc.get(k, {}).get("hbm_bytes_per_launch") is not None:
                row["traffic"] = pmc[k]["hbm_bytes_per_launch"]
            pk[k] = row
        extra["probe_kernels"] = pk
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
        extra["render_ms_per_%dx%d_view" % (res, res)] = round(tr / 4 * 1e3, 2)
    dp_exchange_used = getattr(getattr(runner, "_fast", None), "dp_exchange", None)
    if use_dist:
        extra["dist_backend"] = dist.get_backend()
        # data-parallel invariant: every rank must hold bit-identical parameters (identical summed gradients + a deterministic sweep)
        from jnerf_amd import optim as _optim
        _optim.flush_all()
        _optim.sync_all_sharded()            # the sharded sweep keeps Adam moments (fp16 mode: the fp32 masters too) on their owner's shard: collect them before comparing
        sig = torch.stack([p.detach().double().sum() for p in runner.model.parameters()] + [p.detach().double().abs().sum() for p in runner.model.parameters()])
        sigs = [torch.empty_like(sig) for _ in range(world)]
        dist.all_gather(sigs, sig)
        extra["replicas_identical"] = bool(all(torch.equal(sigs[0], x) for x in sigs))
        dist.barrier()
        from jnerf_amd import dp as _dpm
        in_lib = _dpm.n_ranks_seen() is not None                        # what actually ran (the ranks may have agreed to do without the library's communicator: dp.library_comm_or_fallback)
        extra["dp"] = {"exchange": "rccl in-library: reduce-scatter -> sharded sweep -> all-gather" if in_lib else ("host all-reduce -> sharded sweep -> host all-gather around the two phases of the native step" if args.dp_host_sharded else "host all-reduce between the two phases of the native step"),
                       "overlap": bool(args.dp_overlap), "n_ranks_seen": _dpm.n_ranks_seen() if in_lib else dist.get_world_size(),
                       "dp_exchange": getattr(getattr(runner, "_fast", None), "dp_exchange", None), "grad_wire": "fp16 x 2^14" if getattr(getattr(runner, "_fast", None), "_grad_wire", None) is not None else "fp32",
                       "scaling_curve": "this line is ONE point; no multi-GPU scaling curve has been measured by the authors (one-GPU boxes only)"}
    if use_dist and world > 1 and not args.no_second_scaling:
        # (r6) SURVEY.md 8(e) makes STRONG scaling (one 2^18-sample iteration split over the ranks: the same optimisation trajectory at every N) the primary
        # multi-GPU figure, the driver's contract line is weak scaling: one run prints both.  A second, shorter timed region in the other mode, on every rank (collective);
        # `value` above is untouched by it.
        other = "strong" if args.scaling == "weak" else "weak"
        try:
            runner = None
            torch.cuda.empty_cache()
            share2 = world if other == "strong" else 1
            ngp_cfg(scene=scene, fp16=fp16, aabb_scale=aabb_scale, const_dt=const_dt, n_images=n_images, W=res, H=res, device=f"cuda:{local_rank}", rank=rank, world_size=world,
                    target_batch_size=(1 << 18) // share2, n_rays_per_batch=4096 // share2, dp_force_collectives=bool(args.force_dist), dp_overlap=bool(args.dp_overlap), dp_host_sharded=bool(args.dp_host_sharded),
                    **json.loads(os.environ.get("BENCH_EXTRA_CFG", "{}")))
            torch.manual_seed(1234 + rank)
            r2 = Runner()
            with r2.training_stream():
                st2 = 0
                for _ in range(min(args.burn_in, 256) + min(args.warmup, 16)):
                    r2.train_step(st2); st2 += 1
                barrier()
                t2 = time.perf_counter()
                for _ in range(args.steps):
                    l2 = r2.train_step(st2); st2 += 1
                barrier()
                dt2 = time.perf_counter() - t2
                r2.finish()
            tm2 = torch.tensor([dt2], dtype=torch.float64, device="cuda")
            dist.all_reduce(tm2, op=dist.ReduceOp.MAX)
            dt2 = float(tm2.item())
            extra["dp"]["other_scaling"] = {"scaling": other, "value": round((world if other == "weak" else 1) * args.steps / dt2, 2), "unit": "iters/s", "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                                            "samples_per_iter_per_gpu": (1 << 18) // share2, "burn_in_steps": min(args.burn_in, 256), "steps": args.steps, "loss": round(float(l2.mean().item()), 6)}
            r2 = None
            torch.cuda.empty_cache()
        except Exception as e:        # (deterministic failures hit every rank alike; the contract line must survive them)
            extra["dp"]["other_scaling"] = {"scaling": other, "error": repr(e)[:300]}
        runner = None
    if rank == 0 and not use_dist and lego and scene != "spheres" and not args.no_spheres:
        del runner
        runner = None
        torch.cuda.empty_cache()
        extra["spheres"] = spheres_leg(n_images, res)
    if rank == 0 and not use_dist and not args.no_fox:
        runner = None
        torch.cuda.empty_cache()
        extra["fox"] = fox_leg()
    if rank == 0 and not use_dist and not args.no_neus:
        extra["neus"] = neus_leg()
    if rank == 0 and not use_dist and not args.no_lego_gate:
        # the headline gate itself (README.md:114-120): runs when NeRF-synthetic lego is mounted ($NGP_LEGO_DIR | data/lego), says "not runnable" otherwise (tools/lego_gate.py)
        runner = None
        torch.cuda.empty_cache()
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from lego_gate import lego_gate
            extra["lego_gate"] = lego_gate()
        except Exception as e:            # an extra leg must never take the headline line down with it
            extra["lego_gate"] = {"gate": "failed to run", "error": repr(e)[:300]}
    if rank == 0:
        exact = os.environ.get("NGP_FIELD32_FWD", "split")[:1] in ("m", "0") and os.environ.get("NGP_FIELD32_BWD", "3") != "3"
        field = ("fp32 field network on fp32 MFMAs" if exact else
                 "fp32 field network (fp32-accurate products from split fp16 operands on the fp16 matrix cores: 2e-7 of the largest magnitude vs fp64 in forward and gradients, profiles/r03_split_accuracy.md)")
        wl = (f"Instant-NGP lego config (projects/ngp/configs/ngp_base.py hyper-parameters: aabb_scale 1, L=16, T=2^19, F=2, fp32 table + {field}, const_dt=True, 2^18-sample batches), "
              if lego else "Instant-NGP fox config (ngp_fox.py hyper-parameters: aabb_scale 4, L=16, T=2^19, F=2, fp16 fused MLP, const_dt=False, 2^18-sample batches), ")
        line = {"metric": "training iters/s", "value": round((world if args.scaling == "weak" else 1) * args.steps / dt, 2), "unit": "iters/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f16" if fp16 else "f32",
                "data": "synthetic", "loss": round(float(last_loss), 6),
                "config": {"workload": wl + f"procedural scene '{scene}' {n_images}x{res}x{res} RGBA" + (" (lego-difficulty stand-in; the NeRF-synthetic lego gate - 36.3 dB at 5 min - is NOT runnable: "
                                                                                                          "the data set is not on the box and cannot be fetched)" if scene == "bricks" else "") +
                                       f", random-init weights, {args.burn_in}-step burn-in before warm-up", "scene": scene,
                           "samples_per_iter_per_gpu": (1 << 18) // share, "parallelism": f"ray-batch dp{world} ({args.scaling} scaling)" if world > 1 else "single",
                           **({"dp_exchange": dp_exchange_used} if use_dist else {})},
                "roofline": roof, "cpu_baseline": None if (args.no_cpu_baseline or use_dist) else cpu_baseline(aabb_scale, fp16, const_dt), "extra": extra}
        print(json.dumps(line), flush=True)
    if use_dist:
        from jnerf_amd import dp as _dp
        _dp.destroy()
        dist.destroy_process_group()


def neus_leg(warm=100, timed=200):
    """BASELINE config [4] as this build has it (NOT part of `value`): NeuSRunner on the procedural DTU-layout scene of tests/synth_dtu.py, the HIP hash grid under a small
    SDF network (projects/neus/configs/neus_hash.py's shape, 512 rays x 128 sections per iteration), eikonal term through the second-order hash kernels, HIP
    compositing.  The networks themselves are plain torch (rocBLAS + autograd double backward), as the reference's are plain Jittor ops: iterations/s here measure that."""
    import shutil
    import tempfile
    import numpy as np
    import torch
    root = tempfile.mkdtemp(prefix="neus_bench_")
    try:
        sys.path.insert(0, ROOT)
        from tests import synth_dtu
        from jnerf_amd.utils.config import init_cfg, get_cfg
        from jnerf_amd.neus_runner import NeuSRunner
        synth_dtu.make_scene(root, n_images=16, W=128, H=96)
        init_cfg(os.path.join(ROOT, "projects", "neus", "configs", "neus_hash.py"))
        cfg = get_cfg()
        cfg.device = "cuda"
        cfg.dataset.dataset_dir = root
        cfg.base_exp_dir = os.path.join(root, "log")
        cfg.end_iter, cfg.warm_up_end = 2000, 50
        torch.manual_seed(3)
        np.random.seed(3)
        r = NeuSRunner()
        perm = r.get_image_perm()
        r.update_learning_rate()
        first = None
        for i in range(warm):
            out = r.train_step(perm[i % len(perm)])
            r.update_learning_rate()
            first = float(out["color_loss"]) if first is None else first
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(warm, warm + timed):
            out = r.train_step(perm[i % len(perm)])
            r.update_learning_rate()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res = {"config": "projects/neus/configs/neus_hash.py (hash-grid SDF network, mask loss, no background model) on tests/synth_dtu.py 16x128x96", "iters_per_s": round(timed / dt, 1),
               "ms_per_step": round(dt / timed * 1e3, 3), "rays_per_iter": r.batch_size, "sections_per_ray": r.renderer.n_samples + r.renderer.n_importance,
               "color_loss_first": round(first, 4), "color_loss_last": round(float(out["color_loss"]), 4), "eikonal_last": round(float(out["eikonal_loss"]), 4),
               "fused_composite": bool(r.renderer._use_fused(torch.zeros(1, device="cuda"))), "dtype": "f32"}
        del r
        get_cfg().clear()
        return res
    except Exception as e:            # an extra leg must never take the headline line down with it
        return {"failed": repr(e)[:300]}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def spheres_leg(n_images, res, burn_in=768, timed=200):
    """the lego configuration on rounds 1-3's headline scene (four soft spheres: fewer occupied cells, ~5 % more it/s than `bricks`) - kept for continuity, NOT `value`"""
    import torch
    from jnerf_amd.presets import ngp_cfg
    from jnerf_amd.runner import Runner
    torch.manual_seed(1234)
    ngp_cfg(scene="spheres", fp16=False, aabb_scale=1, const_dt=True, n_images=n_images, W=res, H=res, device="cuda:0")
    r = Runner()
    with r.training_stream():
        for i in range(burn_in):
            r.train_step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(burn_in, burn_in + timed):
            r.train_step(i)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        r.drain()
    out = {"scene": "spheres", "iters_per_s": round(timed / dt, 1), "ms_per_step": round(dt / timed * 1e3, 4), "steps_timed": timed, "burn_in_steps": burn_in, "rays_per_batch": r.sampler.n_rays_per_batch}
    del r
    torch.cuda.empty_cache()
    return out


def fox_leg(burn_in=1024, timed=200, total=3000, psnr=True, marker=None):
    """BASELINE config [1] on the REAL scene: this repository's projects/ngp/configs/ngp_fox.py - the reference's file restated with `_base_` inheritance, equal key by key
    (tests/test_host_cpu.py) - i.e. fp16 fused MLP, aabb_scale 4, cone stepping, on data/fox
    (50 photographs 1080x1920, copied from the reference checkout by build()): iters/s in steady state and PSNR on the scene's own test split."""
    import numpy as np
    import torch
    root = os.path.join(ROOT, "data", "fox")
    if not os.path.isfile(os.path.join(root, "transforms_train.json")):
        return {"skipped": "data/fox is not in the tree (run __graft_entry__.build() where /root/reference exists)"}
    from jnerf_amd.utils.config import init_cfg, get_cfg
    from jnerf_amd.runner import Runner
    cwd = os.getcwd()
    os.chdir(ROOT)                          # the config names the dataset relative to the project root, like the reference's
    try:
        t0 = time.perf_counter()
        init_cfg(os.path.join(ROOT, "projects", "ngp", "configs", "ngp_fox.py"))
        cfg = get_cfg()
        cfg.log_dir = os.path.join(ROOT, "gpurun_out", "logs")
        torch.manual_seed(7)
        r = Runner()
        t_load = time.perf_counter() - t0
        with r.training_stream():
            for i in range(burn_in):
                r.train_step(i)
            if marker is not None:
                marker()                        # (tools/profile_part.py: the kernel trace is cut here)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(burn_in, burn_in + timed):
                r.train_step(i)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            for i in range(burn_in + timed, total):
                r.train_step(i)
            r.drain()
        out = {"config": "projects/ngp/configs/ngp_fox.py (the reference's values, restated), data/fox: %d images %dx%d" % (r.dataset["train"].n_images, r.W, r.H), "iters_per_s": round(timed / dt, 1),
               "ms_per_step": round(dt / timed * 1e3, 4), "steps_timed": timed, "burn_in_steps": burn_in,
               "rays_per_batch": r.sampler.n_rays_per_batch, "load_s": round(t_load, 1), "dtype": "f16"}
        # whole-step HBM fraction of this leg: algorithmic bytes of one iteration (2^18-sample budget x the fill the sampler reaches; counted at the full
        # budget here: the leg keeps no device-side sample statistics) over the measured step time
        sb, _ = step_algorithmic_bytes(float(r.sampler.target_batch_size), r.model.pos_encoder.n_params, True)
        out["step_roofline"] = {"alg_bytes_at_full_batch": int(sb), "GBps": round(sb / (dt / timed) / 1e9, 1), "frac_hbm": round(sb / (dt / timed) / 8e12, 4)}
        if psnr:
            from jnerf_amd.utils.registry import build_from_cfg, DATASETS
            r.dataset["test"] = build_from_cfg(cfg.dataset.test, DATASETS)
            ps = []
            for v in range(r.dataset["test"].n_images):
                img, _, tar = r.render_img("test", v)
                ps.append(float(-10 * np.log10(np.mean((img - tar) ** 2))))
            out["psnr_test_split_after_%d_steps" % total] = round(float(np.mean(ps)), 2)
        del r
        return out
    finally:
        os.chdir(cwd)


if __name__ == "__main__":
    main()
