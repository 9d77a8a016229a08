"""Host logic of the data-parallel exchange step, on CPU: how the hash table is dealt to the ranks (ngp_dp_plan - pure host arithmetic in the C-ABI library),
which gradient buffers Adam.allreduce_grads merges into one collective, and the torch fallback of the fused Adam+EMA sweep with an EMA that aliases the parameter."""
import ctypes as C
import os
import socket
import numpy as np
import pytest
import torch
from jnerf_amd import _lib, ops, dp


@pytest.mark.parametrize("aabb_scale", [1, 4])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("n_buckets", [1, 2])
def test_dp_plan_deals_every_element_exactly_once(aabb_scale, world, n_buckets):
    table, offsets, n_params = ops.level_table(aabb_scale)
    owner = np.full(n_params, -1, np.int64)
    for rank in range(world):
        p = dp.plan(table, n_params, n_buckets=n_buckets, rank=rank, world=world)
        assert p.world == world and p.rank == rank and 1 <= p.n_buckets <= n_buckets
        assert p.cut[0] == 0 and p.cut[p.n_buckets] == p.tail_begin and p.tail_begin + p.tail_count == n_params
        assert 0 <= p.tail_count < 8 * world and p.tail_count % 4 == 0
        for b in range(p.n_buckets):
            cnt = p.cut[b + 1] - p.cut[b]
            assert p.cut[b] % (8 * world) == 0 and cnt % (8 * world) == 0 and p.shard_count[b] * world == cnt          # 16-byte vectors of fp16 and fp32 alike
            assert p.shard_begin[b] == p.cut[b] + rank * p.shard_count[b]
            sl = slice(p.shard_begin[b], p.shard_begin[b] + p.shard_count[b])
            assert (owner[sl] == -1).all()
            owner[sl] = rank
        if p.n_buckets == 2:
            # the boundary is the first element of the first level finer than the scatter's run-combining limit, rounded DOWN: bucket 0 only holds finished levels
            lvl = p.cut_level
            assert table[lvl, 2] > 300 and (lvl == 0 or table[lvl - 1, 2] <= 300)
            assert 2 * int(table[lvl, 0]) - 8 * world < p.cut[1] <= 2 * int(table[lvl, 0])
    assert (owner[:n_params - int(dp.plan(table, n_params, 1, 0, world).tail_count)] >= 0).all() and (owner[n_params - int(dp.plan(table, n_params, 1, 0, world).tail_count):] == -1).all()


def test_dp_plan_argument_errors():
    lib = _lib.lib()
    table, _, n_params = ops.level_table(1)
    tp = np.ascontiguousarray(table).ctypes.data_as(C.c_void_p)
    p = _lib.NgpDpPlan()
    assert lib.ngp_dp_plan(tp, n_params, 2, 2, 1, C.byref(p)) == -1 and b"rank" in lib.ngp_last_error()
    assert lib.ngp_dp_plan(tp, n_params, 2, 0, 3, C.byref(p)) == -1 and b"buckets" in lib.ngp_last_error()
    assert lib.ngp_dp_plan(None, n_params, 2, 0, 1, C.byref(p)) == -1
    assert lib.ngp_allreduce_grads(None, None, 0, None, None, None) == -1            # no communicator
    assert lib.ngp_comm_destroy(None) == 0


def test_ngp_dp_plan_struct_layout(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [n for n, _ in _lib.NgpDpPlan._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "ngp_hip.h"', 'int main(void) {', '  printf("%zu\\n", sizeof(NgpDpPlan));']
    prog += [f'  printf("%zu\\n", offsetof(NgpDpPlan, {f}));' for f in fields] + ['  return 0;', '}']
    src, exe = tmp_path / "l.c", tmp_path / "l"
    src.write_text("\n".join(prog))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).decode().split()]
    assert out[0] == C.sizeof(_lib.NgpDpPlan) and out[1:] == [getattr(_lib.NgpDpPlan, f).offset for f in fields]


def test_torch_sweep_fallback_with_ema_aliasing_the_parameter():
    """tensors the 16-byte-vector kernel cannot take (OriginNeRFNetworks' 1- and 3-element head biases) go through plain torch ops; with the fused
    EMA the stored average IS the parameter (EMA.attach), so the blend has to read the value from BEFORE the Adam update - like the kernel's E = P.  Compared with the
    oracle's Adam+EMA on separate buffers."""
    from jnerf_amd.optim import Adam, EMA
    from oracle import oracle as O
    p0 = np.array([0.3, -0.2, 0.05], np.float32)
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    adam = Adam([p], lr=0.1, eps=1e-15, betas=(0.9, 0.99))
    ema = EMA([p], decay=0.95)
    ema.attach(adam)
    ema.param_groups[0]["values"] = [p.data]                   # what attach() does on CUDA: the EMA state aliases the parameter
    rp, rm, rv, re = p0.copy(), np.zeros(3, np.float32), np.zeros(3, np.float32), p0.copy()
    for step in range(1, 6):
        g = (np.random.default_rng(step).standard_normal(3) * 1e-2).astype(np.float32)
        p.grad = torch.from_numpy(g.copy())
        adam.step(); ema.ema_step()
        O.adam_ema_step(rp, g, rm, rv, re, np.float32(0.1), step)
        assert np.allclose(p.detach().numpy(), rp, rtol=2e-5, atol=1e-7), (step, p.detach().numpy(), rp)
    # and the EMA really smooths: the parameter is NOT the plain Adam iterate
    q = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    plain = Adam([q], lr=0.1, eps=1e-15, betas=(0.9, 0.99))
    for step in range(1, 6):
        q.grad = torch.from_numpy((np.random.default_rng(step).standard_normal(3) * 1e-2).astype(np.float32))
        plain.step()
    assert np.abs(q.detach().numpy() - p.detach().numpy()).max() > 1e-3


def _merge_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jnerf_amd.optim import Adam
    pack = torch.zeros(64)
    a, b = torch.nn.Parameter(torch.zeros(16)), torch.nn.Parameter(torch.zeros(16))
    other = torch.full((64,), 7.0)                      # a base that is NOT a registered gradient pack: its uncovered half must not be reduced
    a.grad, b.grad = pack[:16], pack[32:48]
    c, d = torch.nn.Parameter(torch.zeros(16)), torch.nn.Parameter(torch.zeros(16))
    c.grad, d.grad = other[:16], other[16:32]
    adam = Adam([a, b, c, d])
    adam.register_grad_pack(pack)
    pack[:16] = 1.0 + rank; pack[32:48] = 2.0
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda t, op=None: (calls.append(t.numel()), real(t, op=op))[1]
    adam.allreduce_grads()
    dist.all_reduce = real
    torch.save({"calls": calls, "pack": pack, "other": other}, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_allreduce_merges_registered_packs_only(tmp_path):
    """views of one flat buffer travel as ONE collective only when that buffer was registered as a gradient pack; any other shared base is reduced view
    by view, so memory between the views (not ours) is never summed"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_merge_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = torch.load(tmp_path / "r0.pt")
    assert sorted(r["calls"]) == [16, 16, 64]            # the registered pack as one collective, the two views of the other base separately
    assert torch.equal(r["pack"][:16], torch.full((16,), 3.0)) and torch.equal(r["pack"][32:48], torch.full((16,), 4.0)) and (r["pack"][16:32] == 0).all()
    assert torch.equal(r["other"][:32], torch.full((32,), 14.0)) and torch.equal(r["other"][32:], torch.full((32,), 7.0))       # the uncovered half is untouched
