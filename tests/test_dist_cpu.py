"""Data-parallel logic on CPU: world_size 2 over gloo.  Every rank holds different gradients; after Adam.step + EMA.ema_step all ranks hold
identical parameters, equal to a single process stepping on the SUMMED gradient (checked against the oracle's Adam+EMA)."""
import os
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jnerf_amd.optim import Adam, ExpDecay, EMA
    g = torch.Generator().manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(1024, generator=g)), torch.nn.Parameter(torch.randn(64, generator=g))]
    adam = Adam(params, lr=0.1, eps=1e-15, betas=(0.9, 0.99))
    opt = ExpDecay(adam, decay_start=2, decay_interval=1, decay_base=0.33)
    ema = EMA(params, decay=0.95)
    ema.attach(opt)
    for step in range(4):
        gr = torch.Generator().manual_seed(100 * step + rank)
        for p in params:
            p.grad = torch.randn(p.shape, generator=gr) * 1e-2      # rank-specific gradient (its own ray batch)
        opt.step()
        ema.ema_step()
    np.save(os.path.join(out_dir, f"p{rank}.npy"), np.concatenate([p.detach().numpy().ravel() for p in params]))
    dist.destroy_process_group()


def test_two_rank_adam_ema_matches_summed_gradient(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    assert np.array_equal(a, b), "ranks diverged"
    # single-process reference with the oracle on the summed gradients
    from oracle import oracle as O
    g = torch.Generator().manual_seed(0)
    ps = [torch.randn(1024, generator=g).numpy().copy(), torch.randn(64, generator=g).numpy().copy()]
    ms = [np.zeros_like(p) for p in ps]; vs = [np.zeros_like(p) for p in ps]; es = [p.copy() for p in ps]
    factor = 1.0
    for step in range(4):
        if step >= 2:
            factor *= 0.33                                           # ExpDecay (optims/expdecay.py:20-25)
        grads = []
        for r in range(2):
            gr = torch.Generator().manual_seed(100 * step + r)
            grads.append([(torch.randn(p.shape, generator=gr) * 1e-2).numpy() for p in ps])
        for i in range(2):
            O.adam_ema_step(ps[i], grads[0][i] + grads[1][i], ms[i], vs[i], es[i], np.float32(0.1 * factor), step + 1)
    ref = np.concatenate([p.ravel() for p in ps])
    assert np.allclose(a, ref, rtol=2e-5, atol=1e-6), np.abs(a - ref).max()


def test_rank_rng_streams_are_disjoint():
    from jnerf_amd.rng import pcg32_seed, pcg32_advance
    from oracle import oracle as O
    st = pcg32_seed(1337)
    r = O.PCG32(1337)
    assert (st == r.st).all()
    pcg32_advance(st, 5 << 40); r.advance(5 << 40)
    assert (st == r.st).all()
