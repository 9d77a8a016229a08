"""Ray-batch data parallelism: the host side of csrc/dp_comm.hip (the reference itself has no collective call sites).

One process per GPU.  `torch.distributed` is the control plane (rendezvous, the initial parameter broadcast, the scalar of update_batch_rays, barriers); the
per-iteration exchange step runs INSIDE libngp_hip.so through RCCL on the training stream (reduce-scatter of the hash-table gradient -> sweep of this rank's shard
-> all-gather of the updated shard), so that a data-parallel iteration is still ONE call into the library (fastpath.py).  The library's communicator is created
here from the process group: rank 0 draws the RCCL unique id, torch.distributed broadcasts it.  Backends without RCCL (gloo: the two-ranks-on-one-GPU tests, CPU
unit tests) return None and the callers fall back to the phase-split step with torch.distributed collectives in between."""
import ctypes as C
import torch
import torch.distributed as dist
from . import _lib as L

_comm = None          # (handle, rank, world)
_comm_unavailable = False          # set by library_comm_or_fallback when the ranks agreed to do without it: every later caller gets None (the torch.distributed route)


def active():
    """gradients go through collectives: more than one rank, or `dp_force_collectives = True` in the config (a single-rank process group then runs the complete
    data-parallel sequence - that is how the RCCL path is exercised on a one-GPU box)"""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    from .utils.config import get_cfg
    return dist.get_world_size() > 1 or get_cfg().dp_force_collectives is True


def _create_comm(rank, world, lib=None, device="cuda"):
    """One attempt at the library's communicator with a SYMMETRIC collective sequence: rank 0 always broadcasts - the RCCL unique id, or None when drawing it
    failed - so no rank is ever left waiting in a broadcast that its peer skipped; every rank then runs the same ngp_comm_init or none.  -> (handle | None, error | None)
    (lib / device: the CPU test drives this sequence over gloo with a stand-in library whose rank 0 fails - tests/test_dist_cpu.py)"""
    lib = lib or L.lib()
    uid = (C.c_char * L.COMM_ID_BYTES)()
    box, err = [None], None
    if rank == 0:
        try:
            L.check(lib.ngp_comm_unique_id(uid), "ngp_comm_unique_id")
            box = [bytes(uid.raw)]
        except RuntimeError as e:
            err = e
    if world > 1:
        dist.broadcast_object_list(box, src=0, device=torch.device("cuda", torch.cuda.current_device()) if device == "cuda" else None)
    if box[0] is None:
        return None, err or RuntimeError("rank 0 could not draw an RCCL unique id")
    handle = C.c_void_p()
    try:
        L.check(lib.ngp_comm_init(C.byref(handle), rank, world, C.create_string_buffer(box[0], L.COMM_ID_BYTES)), "ngp_comm_init")
    except RuntimeError as e:
        return None, e
    return handle, None


def _selftest(handle, rank, world, dev, timeout_s):
    """SUM all-reduce of (rank + 1) over 1024 floats through the library's communicator on a private stream; -> None | the error (TimeoutError: never completed)"""
    import time
    from . import ops
    st = torch.cuda.Stream(device=dev)
    try:
        with torch.cuda.stream(st):
            buf = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=dev)
            bufs, counts, dts = (C.c_void_p * 1)(buf.data_ptr()), (C.c_uint64 * 1)(buf.numel()), (C.c_int * 1)(ops._dt(buf))
            L.check(L.lib().ngp_allreduce_grads(handle, st.cuda_stream, 1, bufs, counts, dts), "ngp_allreduce_grads(self-test)")
            done = torch.cuda.Event()
            done.record(st)
        t0 = time.time()
        while not done.query():
            if time.time() - t0 > timeout_s:
                return TimeoutError(f"the self-test collective of the in-library communicator did not complete within {timeout_s:.0f} s")
            time.sleep(0.002)
        want = world * (world + 1) / 2.0
        if not bool((buf == want).all().item()):
            return RuntimeError(f"the self-test collective of the in-library communicator returned {buf[0].item()} instead of {want}")
    except RuntimeError as e:
        return e
    return None


def n_ranks_seen():
    """world size as the library's communicator reports it (ngp_comm_rank_world), None without one - bench.py prints it so that a multi-GPU run diagnoses itself"""
    if _comm is None:
        return None
    r, w = C.c_int(-1), C.c_int(-1)
    L.check(L.lib().ngp_comm_rank_world(_comm[0], C.byref(r), C.byref(w)), "ngp_comm_rank_world")
    return int(w.value)


def library_comm():
    """handle of the library's RCCL communicator over the default process group, created on first use; None when the group's backend is not nccl (= RCCL)"""
    global _comm
    if _comm_unavailable or not active() or dist.get_backend() != "nccl" or not torch.cuda.is_available():
        return None
    rank, world = dist.get_rank(), dist.get_world_size()
    if _comm is not None and _comm[1:] == (rank, world):
        return _comm[0]
    handle, err = _create_comm(rank, world)
    if handle is None:
        raise err
    _comm = (handle, rank, world)
    return handle


def library_comm_or_fallback():
    """library_comm(), agreed on by ALL ranks: if creating the library's own communicator fails on any rank (a second RCCL instance beside torch's in one process is the
    one thing a single-GPU box cannot exercise), every rank drops it and the step runs in two phases with torch.distributed's all-reduce of the gradients (same RCCL
    backend) in between and a replicated sweep - said loudly on stderr, and visible as `dp_exchange` in bench.py's line.  `dp_require_library_comm = True` in the config turns
    the fallback into the error it replaces.  Every rank walks through the same collectives whatever fails where: broadcast of the id (or of None), then the MIN agreement."""
    import sys
    from .utils.config import get_cfg
    global _comm, _comm_unavailable
    if _comm_unavailable or not active() or dist.get_backend() != "nccl" or not torch.cuda.is_available():
        return None
    rank, world = dist.get_rank(), dist.get_world_size()
    if _comm is not None and _comm[1:] == (rank, world):
        return _comm[0]
    handle, err = _create_comm(rank, world)
    dev = torch.device("cuda", torch.cuda.current_device())
    ok = torch.tensor([0 if handle is None else 1], dtype=torch.int32, device=dev)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    hung = False
    if int(ok.item()) == 1:
        # every rank holds a communicator: one small collective through it BEFORE the training step depends on it (a second RCCL instance beside torch's in one process
        # has never run at world > 1 on the authoring side - no multi-GPU box).  On a stream of its own and polled with a deadline, so a collective that never
        # completes costs `dp_selftest_timeout` seconds (default 30) and that stream - not the run.
        err = _selftest(handle, rank, world, dev, float(get_cfg().dp_selftest_timeout or 30.0))
        hung = isinstance(err, TimeoutError)
        ok.fill_(0 if err is not None else 1)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if handle is not None and hung:
            # (r5) a collective of the self-test is still spinning on the device: ncclCommAbort terminates it - abandoning the communicator instead would leave
            # that kernel in flight for ever and every later device-wide synchronize (bench.py, _sync_params_from_rank0, the range-flag poll) would block on it
            L.lib().ngp_comm_abort(handle)
        elif handle is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            L.lib().ngp_comm_destroy(handle)
        if get_cfg().dp_require_library_comm:
            raise err or RuntimeError("the in-library RCCL communicator failed on another rank (dp_require_library_comm)")
        _comm_unavailable = True
        print(f"[jnerf_amd.dp] WARNING rank {rank}: the in-library RCCL communicator is unavailable ({err or 'failed on another rank'}); "
              "the exchange step runs through torch.distributed around a phase-split step", file=sys.stderr, flush=True)
        return None
    _comm = (handle, rank, world)
    return handle


def destroy():
    global _comm
    if _comm is not None:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        L.lib().ngp_comm_destroy(_comm[0])
        _comm = None


def plan(level_table, n_params, n_buckets=1, rank=None, world=None):
    """NgpDpPlan of a table (pure host arithmetic; works without a GPU or a process group when rank / world are given)"""
    if rank is None:
        rank, world = dist.get_rank(), dist.get_world_size()
    import numpy as np
    tbl = np.ascontiguousarray(level_table, dtype=np.uint32)
    p = L.NgpDpPlan()
    L.check(L.lib().ngp_dp_plan(tbl.ctypes.data_as(C.c_void_p), int(n_params), int(world), int(rank), int(n_buckets), C.byref(p)), "ngp_dp_plan")
    return p


def allreduce_grads(tensors, stream=None):
    """SUM all-reduce, in place, of a list of CUDA gradient tensors as ONE RCCL group on the current stream (ngp_allreduce_grads)"""
    comm = library_comm()
    assert comm is not None
    from . import ops
    n = len(tensors)
    bufs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    counts = (C.c_uint64 * n)(*[t.numel() for t in tensors])
    dts = (C.c_int * n)(*[ops._dt(t) for t in tensors])
    L.check(L.lib().ngp_allreduce_grads(comm, stream if stream is not None else ops._stream(), n, bufs, counts, dts), "ngp_allreduce_grads")


def allgather_shards_host(plan_, tensors):
    """the same gather through torch.distributed for process groups without RCCL (gloo: two ranks sharing one GPU in the tests): every rank zeroes what it does not
    own and the buffers are summed - x + 0 is exact, so the result is bit for bit the owners' values"""
    for t in tensors:
        flat = t.view(-1)
        for b in range(plan_.n_buckets):
            lo, hi = int(plan_.cut[b]), int(plan_.cut[b + 1])
            own_lo, cnt = int(plan_.shard_begin[b]), int(plan_.shard_count[b])
            flat[lo:own_lo].zero_()
            flat[own_lo + cnt:hi].zero_()
            seg = flat[lo:hi]
            if seg.dtype == torch.float16:                      # (gloo has no fp16 sum on every build: widen, sum, narrow - exact for x + 0)
                wide = seg.float()
                dist.all_reduce(wide, op=dist.ReduceOp.SUM)
                seg.copy_(wide)
            else:
                dist.all_reduce(seg, op=dist.ReduceOp.SUM)


def allgather_shards(plan_, tensors):
    """in-place all-gather of every rank's shards of tensors laid out like the table (ngp_dp_allgather) on the current stream"""
    comm = library_comm()
    if comm is None:
        return allgather_shards_host(plan_, tensors)
    from . import ops
    n = len(tensors)
    bufs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    dts = (C.c_int * n)(*[ops._dt(t) for t in tensors])
    L.check(L.lib().ngp_dp_allgather(comm, ops._stream(), C.byref(plan_), n, bufs, dts), "ngp_dp_allgather")
