"""ctypes binding of jnerf_amd/csrc/libngp_hip.so (C ABI: include/ngp_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this raises.  The product path never
touches oracle/ (CPU) code."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libngp_hip.so")
_lib = None

F32, F16 = 0, 1
LAYOUT_AOS, LAYOUT_SOA = 0, 1
STAGES = {"field_pack": 0, "hash_fwd": 1, "field_fwd": 2, "composite_fwd": 3, "composite_bwd": 4, "field_bwd": 5, "reduce_slabs": 6, "hash_bwd": 7, "adam_ema": 8, "boundary": 9}   # NGP_STAGE_*

_vp, _u32, _u64, _i32, _f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float


class NgpTrainStep(C.Structure):
    """mirror of `struct NgpTrainStep` in include/ngp_hip.h (argument block of ngp_train_step)"""
    _fields_ = [("n", _u32), ("n_rays", _u32), ("cascades", _i32), ("run_optimizer", _i32),
                ("coords", _vp), ("pos", _vp), ("numsteps", _vp), ("numsteps_compacted", _vp), ("n_valid", _vp), ("bg", _vp), ("target", _vp), ("density_grid_mean", _vp),
                ("table", _vp), ("level_table_host", _vp), ("table_grad", _vp), ("n_params", _u64), ("hash_workspace", _vp), ("hash_workspace_bytes", _u64),
                ("wd", _vp), ("wc", _vp), ("packed_weights", _vp), ("feat", _vp), ("dfeat", _vp), ("out", _vp), ("dout", _vp),
                ("wgrad_slabs", _vp), ("n_slabs", _u32), ("dtype", _i32), ("wgrad_flat", _vp),
                ("huber_delta", _f32), ("pad1", _f32), ("rgb", _vp), ("loss", _vp), ("loss_grad", _vp),
                ("n_opt", _i32), ("step", _u32), ("lr", _f32), ("beta0", _f32), ("beta1", _f32), ("eps", _f32), ("ema_decay", _f32), ("pad2", _f32),
                ("p", _vp * 4), ("g", _vp * 4), ("m", _vp * 4), ("v", _vp * 4), ("ema", _vp * 4), ("p_half", _vp * 4), ("numel", _u64 * 4),
                ("timed_stage", _i32), ("grad_overwrite", _i32),
                ("phase", _i32), ("dp_overlap", _i32), ("dp_table", _i32), ("dp_gather_master", _i32), ("comm", _vp), ("dp", _vp), ("grad_wire", _vp), ("wire_scale", _f32), ("frags_fresh", _i32),
                ("wait_flag", _vp), ("wait_status", _vp), ("wait_value", _u32), ("pad4", _u32)]


PHASE_ALL, PHASE_BACKWARD, PHASE_SWEEP = 0, 1, 2      # NGP_PHASE_*
COMM_ID_BYTES = 128


class NgpDpPlan(C.Structure):
    """mirror of `struct NgpDpPlan` (ngp_dp_plan): how the hash table is dealt to the ranks of a data-parallel run"""
    _fields_ = [("cut", _u64 * 3), ("shard_begin", _u64 * 2), ("shard_count", _u64 * 2), ("tail_begin", _u64), ("tail_count", _u64),
                ("n_buckets", _u32), ("cut_level", _i32), ("world", _i32), ("rank", _i32)]


class NgpRenderChunk(C.Structure):
    """mirror of `struct NgpRenderChunk` in include/ngp_hip.h (argument block of ngp_render_chunk)"""
    _fields_ = [("n_rays", _u32), ("cap", _u32), ("max_samples", _u32), ("const_dt", _i32), ("cascades", _i32), ("dtype", _i32),
                ("aabb0", _f32), ("aabb1", _f32), ("near_distance", _f32), ("cone_angle", _f32),
                ("rays_o", _vp), ("rays_d", _vp), ("bitfield", _vp), ("rng_state_host", _vp), ("coords", _vp), ("pos", _vp), ("numsteps", _vp), ("numsteps_compacted", _vp),
                ("counters", _vp), ("scratch", _vp), ("table", _vp), ("level_table_host", _vp), ("packed_weights", _vp), ("feat", _vp), ("out", _vp),
                ("rgb_out", _vp), ("alpha_out", _vp), ("totals", _vp), ("occ_bounds", _vp)]


SIGNATURES = {
    "ngp_abi_version": (C.c_int, []),
    "ngp_last_error": (C.c_char_p, []),
    "ngp_device_info": (C.c_int, [_i32, _vp]),
    "ngp_selftest_mfma": (C.c_int, [_vp, _vp]),
    "ngp_level_table": (C.c_uint32, [C.c_double, _vp]),
    "ngp_hash_encode_fwd": (C.c_int, [_vp, _u32, _vp, _u32, _vp, _vp, _vp, _i32, _i32, _vp]),
    "ngp_hash_encode_fwd_dydx": (C.c_int, [_vp, _u32, _vp, _u32, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "ngp_hash_encode_bwd_input": (C.c_int, [_vp, _u32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "ngp_hash_encode_bwd_input_bwd_dy": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _i32]),
    "ngp_hash_encode_bwd_input_bwd_grid": (C.c_int, [_vp, _u32, _vp, _u32, _vp, _i32, _vp, _vp, _vp, _u64]),
    "ngp_neus_composite_fwd": (C.c_int, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "ngp_neus_composite_bwd": (C.c_int, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngp_hash_encode_bwd": (C.c_int, [_vp, _u32, _vp, _u32, _vp, _vp, _vp, _u64, _i32, _i32, _i32, _i32, _vp]),
    "ngp_hash_bwd_workspace_bytes": (C.c_uint64, [_vp, _u32]),
    "ngp_hash_bwd_workspace_bytes_for": (C.c_uint64, [_vp, _u32, _i32, _i32]),
    "ngp_hash_encode_bwd_ws": (C.c_int, [_vp, _u32, _vp, _u32, _vp, _vp, _vp, _u64, _i32, _i32, _i32, _i32, _vp, _vp, _u64]),
    "ngp_sh_encode": (C.c_int, [_vp, _u32, _vp, _u32, _vp, _i32]),
    "ngp_field_fwd": (C.c_int, [_vp, _u32, _vp, _i32, _vp, _u32, _vp, _vp, _vp, _i32, _vp]),
    "ngp_density_fwd": (C.c_int, [_vp, _u32, _vp, _i32, _vp, _vp, _i32]),
    "ngp_field_bwd": (C.c_int, [_vp, _u32, _vp, _i32, _vp, _u32, _vp, _vp, _vp, _i32, _vp, _vp, _u32, _vp]),
    "ngp_field_bwd_slabs": (C.c_int, [_u32]),
    "ngp_field_pack_weights": (C.c_int, [_vp, _vp, _vp, _vp]),
    "ngp_field32_pack_weights": (C.c_int, [_vp, _vp, _vp, _vp]),
    "ngp_field32_fwd": (C.c_int, [_vp, _u32, _vp, _i32, _vp, _u32, _vp, _vp, _vp, _vp]),
    "ngp_density32_fwd": (C.c_int, [_vp, _u32, _vp, _i32, _vp, _vp]),
    "ngp_field32_bwd": (C.c_int, [_vp, _u32, _vp, _i32, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "ngp_field32_bwd_slabs": (C.c_int, [_u32]),
    "ngp_field32_range_check": (C.c_int, [_i32]),
    "ngp_field32_select": (C.c_int, [_i32]),
    "ngp_reduce_slabs": (C.c_int, [_vp, _vp, _u32, _u32, _vp, _i32]),
    "ngp_march_rays": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _i32, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _i32]),
    "ngp_compact_coords": (C.c_int, [_vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngp_march_scratch_elems": (C.c_uint64, [_u32]),
    "ngp_march_rays_compacted": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _i32, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp]),
    "ngp_march_rays_compacted_pos": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _i32, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngp_march_rays_compacted_bounds": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _i32, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngp_grid_occupied_bounds": (C.c_int, [_vp, _vp, _i32, _vp]),
    "ngp_composite_fwd": (C.c_int, [_vp, _u32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ngp_composite_fwd_huber": (C.c_int, [_vp, _u32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _f32, _vp, _vp]),
    "ngp_composite_bwd": (C.c_int, [_vp, _u32, _u32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32]),
    "ngp_composite_train": (C.c_int, [_vp, _u32, _u32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _f32, _vp, _vp, _vp, _vp]),
    "ngp_composite_inference": (C.c_int, [_vp, _u32, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "ngp_huber": (C.c_int, [_vp, _u32, _vp, _vp, _f32, _vp, _vp]),
    "ngp_grid_mark_untrained": (C.c_int, [_vp, _u32, _vp, _u32, _vp, _vp, _i32, _i32]),
    "ngp_grid_generate_samples": (C.c_int, [_vp, _u32, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _u32, _f32]),
    "ngp_grid_generate_samples_ordered": (C.c_int, [_vp, _u32, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _u32, _f32, _i32]),
    "ngp_grid_splat_max": (C.c_int, [_vp, _u32, _vp, _vp, _i32, _vp]),
    "ngp_grid_ema": (C.c_int, [_vp, _u32, _f32, _vp, _vp]),
    "ngp_grid_update_bitfield": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "ngp_train_step": (C.c_int, [_vp, C.POINTER(NgpTrainStep)]),
    "ngp_render_chunk": (C.c_int, [_vp, C.POINTER(NgpRenderChunk)]),
    "ngp_train_step_timings": (C.c_int, [_vp, _i32]),
    "ngp_grad_to_half": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _i32]),
    "ngp_grad_to_half_scaled": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _i32, _f32]),
    "ngp_adam_ema_step_scaled": (C.c_int, [_vp, _u64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _u32, _f32, _i32, _f32]),
    "ngp_adam_ema_step": (C.c_int, [_vp, _u64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _u32, _f32, _i32]),
    "ngp_prof_enable": (C.c_int, [C.c_char_p]),
    "ngp_prof_read": (C.c_int, [_i32, _vp, _i32, _vp, _i32]),
    "ngp_flag_signal": (C.c_int, [_vp, _vp, _u32]),
    "ngp_flag_wait": (C.c_int, [_vp, _vp, _u32, _vp]),
    "ngp_comm_unique_id": (C.c_int, [_vp]),
    "ngp_comm_init": (C.c_int, [C.POINTER(_vp), _i32, _i32, _vp]),
    "ngp_comm_destroy": (C.c_int, [_vp]),
    "ngp_comm_abort": (C.c_int, [_vp]),
    "ngp_comm_rank_world": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "ngp_allreduce_grads": (C.c_int, [_vp, _vp, _i32, C.POINTER(_vp), C.POINTER(_u64), C.POINTER(_i32)]),
    "ngp_dp_plan": (C.c_int, [_vp, _u64, _i32, _i32, _i32, C.POINTER(NgpDpPlan)]),
    "ngp_dp_allgather": (C.c_int, [_vp, _vp, C.POINTER(NgpDpPlan), _i32, C.POINTER(_vp), C.POINTER(_i32)]),
    "ngp_generate_rays": (C.c_int, [_vp, _u32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}


def build():
    """Compile every HIP source for gfx950 into csrc/libngp_hip.so (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["bash", os.path.join(_HERE, "csrc", "build.sh")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)       # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        if _lib.ngp_abi_version() != 3:
            raise RuntimeError("libngp_hip.so ABI version mismatch")
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"libngp_hip {what} failed (rc={rc}): {lib().ngp_last_error().decode()}")
