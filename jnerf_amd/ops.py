"""Typed torch-tensor front-end of the C ABI (include/ngp_hip.h).  PyTorch is used for device memory and streams only;
every function below is one asynchronous launch sequence on the current stream of the tensors' device."""
import ctypes as C
import numpy as np
import torch
from . import _lib as L
from ._lib import F32, F16, LAYOUT_AOS, LAYOUT_SOA, check

CAP_RAYS = 1 << 18

# ------------------------------------------------------------------ per-kernel timing (bench.py): HIP event pairs recorded inside the library, on each launch's own stream
def prof_enable(names):
    """"" = off, "*" = every kernel, else an iterable / comma-separated string of kernel base names (csrc/prof.hip)"""
    if not isinstance(names, str):
        names = ",".join(names)
    check(L.lib().ngp_prof_enable(names.encode()), "ngp_prof_enable")


def prof_read(max_n=1 << 16):
    """-> {kernel name: [ms per launch since the last read]} (synchronises on the recorded events)"""
    out, idx = {}, 0
    name = C.create_string_buffer(128)
    buf = (C.c_float * max_n)()
    while True:
        n = L.lib().ngp_prof_read(idx, name, 128, buf, max_n)
        if n < 0:
            return out
        if n:
            out[name.value.decode()] = [buf[i] for i in range(n)]
        idx += 1


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    if _raw_stream is not None:                                   # ~0.3 us instead of ~10 us for torch.cuda.current_stream().cuda_stream
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None or t.numel() == 0:
        return None
    assert t.is_cuda, "libngp_hip needs device tensors (there is no CPU path)"
    return C.c_void_p(t.data_ptr())


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _rows(t, width):
    """(pointer-able tensor, row stride in floats) for a [n,width] fp32 view whose rows may be strided (e.g. coords[:, 4:])."""
    assert t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == width, (t.dtype, t.shape)
    if t.shape[0] == 0:
        return t, width
    assert t.stride(1) == 1, t.stride()
    return t, int(t.stride(0))


# ------------------------------------------------------------------ level table (host, fp32 semantics of HashEncode.h:149-151)
def level_table(aabb_scale, n_levels=16, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048.0):
    """Restates grid_encode.py:17-40 (offsets, fp64) and the per-level fp32 scale/resolution of HashEncode.h:149-151.
    -> (table np.uint32[16,4] = offset,size,res,scale_bits ; offsets np.uint32[17] ; n_params)"""
    from math import exp, log, log2, ceil
    assert n_levels == 16 and base_resolution == 16
    s = exp(log(desired_resolution * aabb_scale / base_resolution) / (n_levels - 1))
    log2s = np.float32(log2(s))
    table = np.zeros((16, 4), np.uint32)
    offsets = np.zeros(17, np.uint32)
    off = 0
    for l in range(16):
        scale_h = pow(2, l * log2(s)) * base_resolution - 1.0
        res_h = ceil(scale_h) + 1
        p = (int(res_h) ** 3 + 7) // 8 * 8
        p = min(p, 1 << log2_hashmap_size)
        arg = np.float32(l) * log2s
        scale_d = np.float32(np.float32(2.0 ** float(arg)) * np.float32(16.0)) - np.float32(1.0)
        res_d = int(np.ceil(scale_d)) + 1
        offsets[l] = off
        table[l] = (off, p, res_d, np.float32(scale_d).view(np.uint32))
        off += p
    offsets[16] = off
    return table, offsets, off * 2


def _tbl(table_np):
    assert table_np.dtype == np.uint32 and table_np.size == 64
    return np.ascontiguousarray(table_np).ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ hash grid
def hash_encode_fwd(pos, table, level_tbl, out=None, layout=LAYOUT_AOS, n_valid=None):
    pos, stride = _rows(pos, 3)
    n = pos.shape[0]
    if out is None:
        out = torch.empty((n, 32) if layout == LAYOUT_AOS else (16, n, 2), dtype=table.dtype, device=pos.device)
    check(L.lib().ngp_hash_encode_fwd(_stream(), n, _p(pos), stride, _p(table), _tbl(level_tbl), _p(out), _dt(table), layout, _p(n_valid)), "ngp_hash_encode_fwd")
    return out


def hash_encode_fwd_dydx(pos, table, level_tbl, out=None, layout=LAYOUT_AOS, n_valid=None, dy_dx=None):
    """forward + d(encoding)/d(position) (kernel_grid's dy_dx branch, HashEncode.h:205-251) -> (out, dy_dx f32[n,3,32])"""
    pos, stride = _rows(pos, 3)
    n = pos.shape[0]
    if out is None:
        out = torch.empty((n, 32) if layout == LAYOUT_AOS else (16, n, 2), dtype=table.dtype, device=pos.device)
    if dy_dx is None:
        dy_dx = torch.empty((n, 3, 32), dtype=torch.float32, device=pos.device)
    check(L.lib().ngp_hash_encode_fwd_dydx(_stream(), n, _p(pos), stride, _p(table), _tbl(level_tbl), _p(out), _dt(table), layout, _p(n_valid), _p(dy_dx)), "ngp_hash_encode_fwd_dydx")
    return out, dy_dx


def hash_encode_bwd_input(dLdy, dy_dx, layout=LAYOUT_AOS, n_valid=None):
    """dL/dpos [n,3] f32 = sum_k dLdy[:, k] * dy_dx[:, :, k]"""
    n = dy_dx.shape[0]
    assert dLdy.is_contiguous() and dy_dx.is_contiguous() and dy_dx.dtype == torch.float32
    out = torch.empty((n, 3), dtype=torch.float32, device=dy_dx.device)
    check(L.lib().ngp_hash_encode_bwd_input(_stream(), n, _p(dLdy), _dt(dLdy), layout, _p(dy_dx), _p(out), _p(n_valid)), "ngp_hash_encode_bwd_input")
    return out


def hash_encode_bwd_input_bwd_dy(u, dy_dx, dtype=torch.float32):
    """gradient of hash_encode_bwd_input w.r.t. its dLdy for an upstream gradient u [n,3]: [n,32] = sum_d u[:, d] * dy_dx[:, d, :]"""
    n = dy_dx.shape[0]
    assert dtype in (torch.float32, torch.float16)
    assert u.is_contiguous() and u.dtype == torch.float32 and u.shape == (n, 3) and dy_dx.is_contiguous() and dy_dx.dtype == torch.float32
    out = torch.empty((n, 32), dtype=dtype, device=dy_dx.device)
    check(L.lib().ngp_hash_encode_bwd_input_bwd_dy(_stream(), n, _p(u), _p(dy_dx), _p(out), F32 if dtype == torch.float32 else F16), "ngp_hash_encode_bwd_input_bwd_dy")
    return out


def hash_encode_bwd_input_bwd_grid(pos, dLdy, u, level_tbl, grad):
    """gradient of hash_encode_bwd_input w.r.t. the table (through dy_dx), ADDED into grad (fp32, n_params elements)"""
    pos, stride = _rows(pos, 3)
    n = pos.shape[0]
    assert dLdy.is_contiguous() and dLdy.shape == (n, 32) and u.is_contiguous() and u.dtype == torch.float32 and u.shape == (n, 3) and grad.dtype == torch.float32 and grad.is_contiguous()
    check(L.lib().ngp_hash_encode_bwd_input_bwd_grid(_stream(), n, _p(pos), stride, _p(dLdy), _dt(dLdy), _p(u), _tbl(level_tbl), _p(grad), grad.numel()), "ngp_hash_encode_bwd_input_bwd_grid")
    return grad


def hash_encode_bwd(pos, dLdy, level_tbl, n_params, grad=None, grad_dtype=None, layout=LAYOUT_AOS, zero_first=True, n_valid=None, workspace=None):
    """workspace: uint8 tensor of >= hash_bwd_workspace_bytes(level_tbl, n, dLdy.dtype) -> the binned scatter, no float atomics, bit-reproducible (ngp_hash_encode_bwd_ws);
    without one: the reference's scheme, one global float atomic per corner (ngp_hash_encode_bwd)"""
    pos, stride = _rows(pos, 3)
    n = pos.shape[0]
    assert dLdy.is_contiguous()
    if grad is None:
        grad = torch.empty(n_params, dtype=grad_dtype or dLdy.dtype, device=pos.device)
    if workspace is not None:
        check(L.lib().ngp_hash_encode_bwd_ws(_stream(), n, _p(pos), stride, _p(dLdy), _tbl(level_tbl), _p(grad), n_params, _dt(dLdy), _dt(grad), layout,
                                             int(zero_first), _p(n_valid), _p(workspace), workspace.numel() * workspace.element_size()), "ngp_hash_encode_bwd_ws")
    else:
        check(L.lib().ngp_hash_encode_bwd(_stream(), n, _p(pos), stride, _p(dLdy), _tbl(level_tbl), _p(grad), n_params, _dt(dLdy), _dt(grad), layout,
                                          int(zero_first), _p(n_valid)), "ngp_hash_encode_bwd")
    return grad


def hash_bwd_workspace_bytes(level_tbl, n, dtype=None, grad_dtype=torch.float32):
    """bytes of workspace the binned scatter needs for n samples: for the path dL/dy of `dtype` takes (fp32: record regions, fp16: per-corner lists), or - dtype None -
    enough for either"""
    if dtype is None:
        return int(L.lib().ngp_hash_bwd_workspace_bytes(_tbl(level_tbl), n))
    code = lambda t: L.F16 if t == torch.float16 else L.F32
    return int(L.lib().ngp_hash_bwd_workspace_bytes_for(_tbl(level_tbl), n, code(dtype), code(grad_dtype)))


def sh_encode(d, dtype=torch.float32):
    d, stride = _rows(d, 3)
    out = torch.empty((d.shape[0], 16), dtype=dtype, device=d.device)
    check(L.lib().ngp_sh_encode(_stream(), d.shape[0], _p(d), stride, _p(out), _dt(out)), "ngp_sh_encode")
    return out


# ------------------------------------------------------------------ field network
WEIGHTS_PACKED = 0x100
PACKED_WEIGHT_HALVES = 21504


def field_pack_weights(wd, wc, out=None):
    """MFMA-ordered fragments of both weight packs (f16[21504]); pass as `packed=` to field_fwd / field_bwd / density_fwd to build them once per step"""
    assert wd.dtype == torch.float16 and wc.dtype == torch.float16 and wd.numel() == 3072 and wc.numel() == 7168
    if out is None:
        out = torch.empty(PACKED_WEIGHT_HALVES, dtype=torch.float16, device=wd.device)
    check(L.lib().ngp_field_pack_weights(_stream(), _p(wd), _p(wc), _p(out)), "ngp_field_pack_weights")
    return out


def field_fwd(feat, d, wd, wc, layout=LAYOUT_AOS, out_dtype=torch.float16, out=None, n_valid=None, packed=None):
    assert feat.dtype == torch.float16 and feat.is_contiguous()
    if packed is not None:
        wd, wc, layout = packed, None, layout | WEIGHTS_PACKED
    else:
        assert wd.dtype == torch.float16 and wc.dtype == torch.float16 and wd.numel() == 3072 and wc.numel() == 7168
    d, stride = _rows(d, 3)
    n = d.shape[0]
    if out is None:
        out = torch.empty((n, 4), dtype=out_dtype, device=d.device)
    check(L.lib().ngp_field_fwd(_stream(), n, _p(feat), layout, _p(d), stride, _p(wd), _p(wc), _p(out), _dt(out), _p(n_valid)), "ngp_field_fwd")
    return out


def density_fwd(feat, wd, n, layout=LAYOUT_AOS, out_dtype=torch.float16, packed=None):
    assert feat.dtype == torch.float16 and feat.is_contiguous()
    if packed is not None:
        wd, layout = packed, layout | WEIGHTS_PACKED
    assert wd.dtype == torch.float16
    out = torch.empty((n,), dtype=out_dtype, device=feat.device)
    check(L.lib().ngp_density_fwd(_stream(), n, _p(feat), layout, _p(wd), _p(out), _dt(out)), "ngp_density_fwd")
    return out


def field_bwd_slabs(n):
    return int(L.lib().ngp_field_bwd_slabs(n))


def field_bwd(feat, d, wd, wc, dLdout, layout=LAYOUT_AOS, dfeat=None, slabs=None, n_valid=None, packed=None):
    """-> (dLdfeat f16 in `layout`, slabs f32[n_slabs,10240]); sum the slabs with reduce_slabs."""
    assert feat.dtype == torch.float16 and feat.is_contiguous() and dLdout.is_contiguous()
    if packed is not None:
        wd, wc, layout = packed, None, layout | WEIGHTS_PACKED
    d, stride = _rows(d, 3)
    n = d.shape[0]
    ns = field_bwd_slabs(n)
    if dfeat is None:
        dfeat = torch.zeros_like(feat)
    if slabs is None:
        slabs = torch.empty((ns, 10240), dtype=torch.float32, device=d.device)
    check(L.lib().ngp_field_bwd(_stream(), n, _p(feat), layout, _p(d), stride, _p(wd), _p(wc), _p(dLdout), _dt(dLdout), _p(dfeat), _p(slabs), ns, _p(n_valid)), "ngp_field_bwd")
    return dfeat, slabs


# ---- fp32 field network (cfg.fp16 unset: ngp_base.py / lego) on v_mfma_f32_16x16x4_f32
PACKED32_WEIGHT_FLOATS = 40960


def field32_pack_weights(wd, wc, out=None):
    assert wd.dtype == torch.float32 and wc.dtype == torch.float32 and wd.numel() == 3072 and wc.numel() == 7168 and wd.is_contiguous() and wc.is_contiguous()
    if out is None:
        out = torch.empty(PACKED32_WEIGHT_FLOATS, dtype=torch.float32, device=wd.device)
    check(L.lib().ngp_field32_pack_weights(_stream(), _p(wd), _p(wc), _p(out)), "ngp_field32_pack_weights")
    return out


def field32_fwd(feat, d, wd, wc, layout=LAYOUT_AOS, out=None, n_valid=None, packed=None):
    assert feat.dtype == torch.float32 and feat.is_contiguous()
    if packed is not None:
        wd, wc, layout = packed, None, layout | WEIGHTS_PACKED
    else:
        assert wd.dtype == torch.float32 and wc.dtype == torch.float32 and wd.numel() == 3072 and wc.numel() == 7168
    d, stride = _rows(d, 3)
    n = d.shape[0]
    if out is None:
        out = torch.empty((n, 4), dtype=torch.float32, device=d.device)
    check(L.lib().ngp_field32_fwd(_stream(), n, _p(feat), layout, _p(d), stride, _p(wd), _p(wc), _p(out), _p(n_valid)), "ngp_field32_fwd")
    return out


def density32_fwd(feat, wd, n, layout=LAYOUT_AOS, packed=None):
    assert feat.dtype == torch.float32 and feat.is_contiguous()
    if packed is not None:
        wd, layout = packed, layout | WEIGHTS_PACKED
    out = torch.empty((n,), dtype=torch.float32, device=feat.device)
    check(L.lib().ngp_density32_fwd(_stream(), n, _p(feat), layout, _p(wd), _p(out)), "ngp_density32_fwd")
    return out


def field32_range_check(reset=True, synchronize=True):
    """flag bits of the split-operand fp32 field kernels (include/ngp_hip.h): 1 = an operand came within 4x of fp16's range, 2 = one left it.
    synchronize: wait for the device first, so that every launch issued so far has reported (the read-back itself only orders against the null stream)."""
    if synchronize and torch.cuda.is_available():
        torch.cuda.synchronize()
    v = int(L.lib().ngp_field32_range_check(int(reset)))
    if v < 0:
        check(v, "ngp_field32_range_check")
    return v


def field32_select(exact):
    """exact=True: the exact-product fp32-MFMA kernels from now on in this process; False: the split-operand kernels (default).  Returns the previous choice."""
    return bool(L.lib().ngp_field32_select(int(bool(exact))))


def field32_bwd_slabs(n):
    return int(L.lib().ngp_field32_bwd_slabs(n))


def field32_bwd(feat, d, wd, wc, dLdout, layout=LAYOUT_AOS, dfeat=None, slabs=None, n_valid=None, packed=None):
    """-> (dLdfeat f32 in `layout`, slabs f32[n_slabs,10240]); sum the slabs with reduce_slabs."""
    assert feat.dtype == torch.float32 and feat.is_contiguous() and dLdout.dtype == torch.float32 and dLdout.is_contiguous()
    if packed is not None:
        wd, wc, layout = packed, None, layout | WEIGHTS_PACKED
    d, stride = _rows(d, 3)
    n = d.shape[0]
    ns = field32_bwd_slabs(n)
    if dfeat is None:
        dfeat = torch.zeros_like(feat)
    if slabs is None:
        slabs = torch.empty((ns, 10240), dtype=torch.float32, device=d.device)
    check(L.lib().ngp_field32_bwd(_stream(), n, _p(feat), layout, _p(d), stride, _p(wd), _p(wc), _p(dLdout), _p(dfeat), _p(slabs), ns, _p(n_valid)), "ngp_field32_bwd")
    return dfeat, slabs


def reduce_slabs(slabs, out=None, accumulate=False):
    ns, width = slabs.shape
    if out is None:
        out = torch.empty(width, dtype=torch.float32, device=slabs.device)
        accumulate = False
    check(L.lib().ngp_reduce_slabs(_stream(), _p(slabs), ns, width, _p(out), int(accumulate)), "ngp_reduce_slabs")
    return out


# ------------------------------------------------------------------ sampler
def march_rays(rays_o, rays_d, bitfield, aabb, rng_state, max_samples, cone_angle=1.0 / 256, near=0.2, const_dt=True, cascades=5, coords=None, zero_coords=True):
    """rng_state: np.uint64[2] (advanced by 2^32 in place).  -> coords[max_samples,7], numsteps[n,2] (i32 view of u32), counters[2], ray_indices[n]"""
    assert rays_o.is_contiguous() and rays_d.is_contiguous() and rays_o.dtype == torch.float32 and bitfield.dtype == torch.uint8
    n = rays_o.shape[0]
    dev = rays_o.device
    if coords is None:
        coords = torch.empty((max_samples, 7), dtype=torch.float32, device=dev)
    numsteps = torch.empty((n, 2), dtype=torch.int32, device=dev)
    counters = torch.empty(2, dtype=torch.int32, device=dev)
    ray_idx = torch.zeros(n, dtype=torch.int32, device=dev)
    scratch = torch.empty(n + 1024, dtype=torch.int32, device=dev)
    check(L.lib().ngp_march_rays(_stream(), n, _p(rays_o), _p(rays_d), _p(bitfield), aabb[0], aabb[1], near, cone_angle, int(const_dt), cascades,
                                 rng_state.ctypes.data_as(C.c_void_p), max_samples, _p(coords), _p(numsteps), _p(counters), _p(ray_idx), _p(scratch), int(zero_coords)), "ngp_march_rays")
    return coords, numsteps, counters, ray_idx


def compact_coords(coords_in, numsteps_in, cap, coords_out=None):
    n = numsteps_in.shape[0]
    dev = coords_in.device
    if coords_out is None:
        coords_out = torch.empty((cap, 7), dtype=torch.float32, device=dev)
    numsteps_out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    counter = torch.empty(1, dtype=torch.int32, device=dev)
    check(L.lib().ngp_compact_coords(_stream(), n, cap, _p(coords_in), _p(numsteps_in), _p(coords_out), _p(numsteps_out), _p(counter), None), "ngp_compact_coords")
    return coords_out, numsteps_out, counter


def march_scratch_elems(n_rays):
    return int(L.lib().ngp_march_scratch_elems(n_rays))


def march_rays_compacted(rays_o, rays_d, bitfield, aabb, rng_state, max_samples, cap, cone_angle=1.0 / 256, near=0.2, const_dt=True, cascades=5,
                         coords_out=None, numsteps=None, numsteps_c=None, counters=None, scratch=None, pos_out=None, occ_bounds=None):
    n = rays_o.shape[0]
    dev = rays_o.device
    if coords_out is None:
        coords_out = torch.zeros((cap, 7), dtype=torch.float32, device=dev)
    if numsteps is None:
        numsteps = torch.empty((n, 2), dtype=torch.int32, device=dev)
    if numsteps_c is None:
        numsteps_c = torch.empty((n, 2), dtype=torch.int32, device=dev)
    if counters is None:
        counters = torch.empty(4, dtype=torch.int32, device=dev)
    need = march_scratch_elems(n)
    if scratch is None:
        scratch = torch.empty(need, dtype=torch.int32, device=dev)
    assert scratch.numel() >= need, "march scratch too small: use ops.march_scratch_elems(n_rays)"
    check(L.lib().ngp_march_rays_compacted_bounds(_stream(), n, _p(rays_o), _p(rays_d), _p(bitfield), aabb[0], aabb[1], near, cone_angle, int(const_dt), cascades,
                                                  rng_state.ctypes.data_as(C.c_void_p), max_samples, cap, _p(coords_out), _p(numsteps), _p(numsteps_c), _p(counters), _p(scratch),
                                                  _p(pos_out), _p(occ_bounds)), "ngp_march_rays_compacted_bounds")
    return coords_out, numsteps, numsteps_c, counters


def composite_fwd(net, coords, numsteps, numsteps_c, bg, cascades=5, out=None):
    n = numsteps.shape[0]
    assert net.is_contiguous() and coords.is_contiguous() and bg.is_contiguous()
    if out is None:
        out = torch.empty((n, 3), dtype=torch.float32, device=net.device)
    check(L.lib().ngp_composite_fwd(_stream(), n, _p(net), _dt(net), _p(coords), _p(numsteps), _p(numsteps_c), _p(bg), cascades, _p(out)), "ngp_composite_fwd")
    return out


def composite_fwd_huber(net, coords, numsteps, numsteps_c, bg, target, delta, cascades=5, out=None, loss=None, grad=None):
    """composite_fwd + huber in one launch -> (rgb, loss, loss_grad), all [n_rays,3] f32"""
    n = numsteps.shape[0]
    assert net.is_contiguous() and coords.is_contiguous() and bg.is_contiguous() and target.is_contiguous()
    dev = net.device
    out = torch.empty((n, 3), dtype=torch.float32, device=dev) if out is None else out
    loss = torch.empty((n, 3), dtype=torch.float32, device=dev) if loss is None else loss
    grad = torch.empty((n, 3), dtype=torch.float32, device=dev) if grad is None else grad
    check(L.lib().ngp_composite_fwd_huber(_stream(), n, _p(net), _dt(net), _p(coords), _p(numsteps), _p(numsteps_c), _p(bg), cascades, _p(out), _p(target), delta, _p(loss), _p(grad)),
          "ngp_composite_fwd_huber")
    return out, loss, grad


def composite_train(net, coords, numsteps, numsteps_c, bg, target, delta, density_grid_mean, cascades=5, out=None, loss=None, grad=None, dout=None, n_elems=None):
    """composite_fwd_huber + composite_bwd in one launch (ngp_composite_train: the native training step's form) -> (rgb, loss, loss_grad, dLdout)"""
    n = numsteps.shape[0]
    out = out if out is not None else torch.empty((n, 3), dtype=torch.float32, device=net.device)
    loss = loss if loss is not None else torch.empty((n, 3), dtype=torch.float32, device=net.device)
    grad = grad if grad is not None else torch.empty((n, 3), dtype=torch.float32, device=net.device)
    dout = dout if dout is not None else torch.zeros_like(net)
    check(L.lib().ngp_composite_train(_stream(), n, int(n_elems or net.shape[0]), _p(net), _dt(net), _p(coords), _p(numsteps), _p(numsteps_c), _p(bg), cascades, _p(out), _p(target), delta,
                                      _p(loss), _p(grad), _p(density_grid_mean), _p(dout)), "ngp_composite_train")
    return out, loss, grad, dout


def composite_bwd(net, coords, numsteps_c, loss_grad, rgb_ray, density_grid_mean, cascades=5, dout=None, zero_first=True):
    n = numsteps_c.shape[0]
    assert net.is_contiguous() and loss_grad.is_contiguous() and rgb_ray.is_contiguous()
    if dout is None:
        dout = torch.empty_like(net)
    check(L.lib().ngp_composite_bwd(_stream(), n, net.shape[0], _p(net), _dt(net), _p(coords), _p(numsteps_c), _p(loss_grad), _p(rgb_ray), _p(density_grid_mean), cascades, _p(dout), int(zero_first)),
          "ngp_composite_bwd")
    return dout


def composite_inference(net, coords, numsteps, cascades=5):
    n = numsteps.shape[0]
    rgb = torch.empty((n, 3), dtype=torch.float32, device=net.device)
    alpha = torch.empty((n, 1), dtype=torch.float32, device=net.device)
    check(L.lib().ngp_composite_inference(_stream(), n, _p(net), _dt(net), _p(coords), _p(numsteps), cascades, _p(rgb), _p(alpha)), "ngp_composite_inference")
    return rgb, alpha


def huber(x, target, delta=0.1, want_loss=True, want_grad=True, loss=None, grad=None):
    x, target = x.contiguous(), target.contiguous()
    if loss is None and want_loss:
        loss = torch.empty_like(x)
    if grad is None and want_grad:
        grad = torch.empty_like(x)
    check(L.lib().ngp_huber(_stream(), x.numel(), _p(x), _p(target), delta, _p(loss), _p(grad)), "ngp_huber")
    return loss, grad


# ------------------------------------------------------------------ density grid
def grid_mark_untrained(n_elements, focal, xforms, W, H, grid=None):
    if grid is None:
        grid = torch.empty(n_elements, dtype=torch.float32, device=focal.device)
    check(L.lib().ngp_grid_mark_untrained(_stream(), n_elements, _p(grid), focal.shape[0], _p(focal.contiguous()), _p(xforms.contiguous()), int(W), int(H)), "ngp_grid_mark_untrained")
    return grid


def grid_generate_samples(n, rng_state, ema_step, aabb, grid, n_cascades, thresh, pos=None, idx=None, morton_order=False):
    """morton_order: same samples, stored so that consecutive slots are consecutive Morton cells (coherent gathers in the density query that follows)"""
    if pos is None:
        pos = torch.empty((n, 3), dtype=torch.float32, device=grid.device)
    if idx is None:
        idx = torch.empty(n, dtype=torch.int32, device=grid.device)
    check(L.lib().ngp_grid_generate_samples_ordered(_stream(), n, rng_state.ctypes.data_as(C.c_void_p), _p(ema_step), aabb[0], aabb[1], _p(grid), _p(pos), _p(idx), n_cascades, thresh,
                                                    int(morton_order)), "ngp_grid_generate_samples")
    return pos, idx


def grid_splat_max(indices, density, grid_tmp):
    check(L.lib().ngp_grid_splat_max(_stream(), indices.shape[0], _p(indices), _p(density), _dt(density), _p(grid_tmp)), "ngp_grid_splat_max")
    return grid_tmp


def grid_ema(grid, grid_tmp, decay=0.95):
    check(L.lib().ngp_grid_ema(_stream(), grid.shape[0], decay, _p(grid), _p(grid_tmp)), "ngp_grid_ema")
    return grid


OCC_BOUNDS_INTS = 64 + 2 * 32 ** 3 // 4      # NGP_OCC_BOUNDS_INTS


def grid_occupied_bounds(bitfield, cascades=5, out=None):
    """i32[OCC_BOUNDS_INTS]: [6c .. 6c+5] = (min x, y, z, max x, y, z) of cascade c's occupied cells (min > max: empty), then the dilated 32^3 coarse map; see ngp_grid_occupied_bounds"""
    if out is None:
        out = torch.empty(OCC_BOUNDS_INTS, dtype=torch.int32, device=bitfield.device)
    assert out.numel() >= OCC_BOUNDS_INTS
    check(L.lib().ngp_grid_occupied_bounds(_stream(), _p(bitfield), cascades, _p(out)), "ngp_grid_occupied_bounds")
    return out


def grid_update_bitfield(grid, cascades=5, mean=None, bitfield=None):
    if mean is None:
        mean = torch.empty(1, dtype=torch.float32, device=grid.device)
    if bitfield is None:
        bitfield = torch.zeros(128 ** 3 * cascades // 8, dtype=torch.uint8, device=grid.device)
    check(L.lib().ngp_grid_update_bitfield(_stream(), _p(grid), cascades, _p(mean), _p(bitfield)), "ngp_grid_update_bitfield")
    return bitfield, mean


# ------------------------------------------------------------------ optimiser / rays
def grad_to_half(g32, g16, zero_src=True, scale=1.0):
    assert g32.dtype == torch.float32 and g16.dtype == torch.float16 and g32.numel() == g16.numel()
    check(L.lib().ngp_grad_to_half_scaled(_stream(), g32.numel(), _p(g32), _p(g16), int(zero_src), float(scale)), "ngp_grad_to_half")
    return g16


def adam_ema_step(p, g, m, v, ema, p_half, lr, step, b0=0.9, b1=0.99, eps=1e-15, ema_decay=0.95, zero_grad=True, grad_mul=1.0):
    check(L.lib().ngp_adam_ema_step_scaled(_stream(), p.numel(), _p(p), _p(g), _dt(g), _p(m), _p(v), _p(ema), _p(p_half), lr, b0, b1, eps, step, ema_decay, int(zero_grad),
                                           float(grad_mul)), "ngp_adam_ema_step")


def generate_rays(pixel_index, W, H, focal, metadata, xforms, images=None, bg=None, out=None):
    """`out` = (img int32[cap], o [cap,3], d [cap,3], target [cap,3]) persistent buffers with cap >= n: the results are views of their first n rows (nothing is allocated)"""
    n = pixel_index.shape[0]
    dev = pixel_index.device
    if out is not None:
        img, o, d = out[0][:n], out[1][:n], out[2][:n]
        target = out[3][:n] if images is not None else None
    else:
        img = torch.empty(n, dtype=torch.int32, device=dev)
        o = torch.empty((n, 3), dtype=torch.float32, device=dev)
        d = torch.empty((n, 3), dtype=torch.float32, device=dev)
        target = torch.empty((n, 3), dtype=torch.float32, device=dev) if images is not None else None
    check(L.lib().ngp_generate_rays(_stream(), n, _p(pixel_index), int(W), int(H), _p(focal), _p(metadata), _p(xforms), _p(images), _p(bg), _p(img), _p(o), _p(d), _p(target)), "ngp_generate_rays")
    return img, o, d, target


def flag_signal(flag, value):
    """one-thread launch on the current stream: *flag = value (device int32 / uint32 scalar view), see ngp_flag_signal"""
    check(L.lib().ngp_flag_signal(_stream(), _p(flag), int(value) & 0xFFFFFFFF), "ngp_flag_signal")


def flag_wait(flag, value, status=None):
    check(L.lib().ngp_flag_wait(_stream(), _p(flag), int(value) & 0xFFFFFFFF, _p(status)), "ngp_flag_wait")


def selftest_mfma(device="cuda"):
    res = torch.zeros(4, dtype=torch.int32, device=device)
    check(L.lib().ngp_selftest_mfma(_stream(), _p(res)), "ngp_selftest_mfma")
    r = res.cpu().numpy()
    return int(r[0]), int(r[1]) & 0xFFFFFFFF
