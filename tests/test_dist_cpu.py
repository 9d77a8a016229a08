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


class _FakeLib:
    """stand-in for libngp_hip.so's communicator entry points: what fails is chosen per rank"""
    def __init__(self, rank, fail_id_on_rank0, fail_init_on):
        self.rank, self.fail_id, self.fail_init = rank, fail_id_on_rank0, fail_init_on
        self.inits = 0

    def ngp_comm_unique_id(self, uid):
        if self.fail_id:
            return -1
        uid.raw = bytes(range(128))[:len(uid.raw)]
        return 0

    def ngp_comm_init(self, handle_ref, rank, world, idbuf):
        self.inits += 1
        return -1 if self.rank in self.fail_init else 0

    def ngp_last_error(self):
        return b"stand-in failure"


def _comm_worker(rank, world, port, out_dir, case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jnerf_amd import dp
    fail_id, fail_init = {"id": (True, ()), "init1": (False, (1,)), "ok": (False, ())}[case]
    lib = _FakeLib(rank, fail_id, fail_init)
    handle, err = dp._create_comm(rank, world, lib=lib, device="cpu")
    # the agreement that follows in library_comm_or_fallback: every rank reaches it whatever failed where
    ok = torch.tensor([0 if handle is None else 1], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    np.save(os.path.join(out_dir, f"c{rank}.npy"), np.array([0 if handle is None else 1, int(ok.item()), lib.inits]))
    dist.destroy_process_group()


def test_communicator_agreement_is_symmetric_whatever_fails(tmp_path):
    """(r4) rank 0 failing to draw the RCCL unique id used to skip the broadcast its peers were waiting in.  Now rank 0 always broadcasts (the id or None) and every
    rank walks through the same collectives: with rank 0's id failing, with rank 1's init failing, and with nothing failing, both ranks finish and agree."""
    for case, want in (("id", [(0, 0, 0), (0, 0, 0)]), ("init1", [(1, 0, 1), (0, 0, 1)]), ("ok", [(1, 1, 1), (1, 1, 1)])):
        port = _free_port()
        d = tmp_path / case
        d.mkdir()
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, str(d), case)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0, (case, p.exitcode)          # (a deadlock shows up as a timeout here)
        got = [tuple(int(v) for v in np.load(d / f"c{r}.npy")) for r in range(2)]
        assert got == want, (case, got)



def _flag_worker(rank, world, port, out_dir):
    """Runner._agree_on_range_flag on two ranks over gloo: rank 1 sees bit 0 at the second poll and bit 1 (the error) at the fourth; rank 0 never sees anything itself."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jnerf_amd.runner import Runner

    class _S:
        device = torch.device("cpu")
    r = Runner.__new__(Runner)                      # only the agreement logic: no model, no data set
    r.sampler = _S()
    local = {1: [0, 1, 0, 2, 0], 0: [0, 0, 0, 0, 0]}[rank]
    seen = [Runner._agree_on_range_flag(r, f, final=False) for f in local]
    seen.append(Runner._agree_on_range_flag(r, 0, final=True))
    np.save(os.path.join(out_dir, f"f{rank}.npy"), np.asarray(seen))
    dist.destroy_process_group()


def test_range_flag_is_agreed_on_by_all_ranks(tmp_path):
    """(r5) a rank that acted on its own device's range flag - raised, or switched kernels - would leave its peers blocked in the next collective.  Every poll
    MAX-reduces the flags; in the loop the reduction issued at poll k is consumed at poll k + 1 (no host wait inside the pipeline), the final poll reduces synchronously.
    Both ranks must see the SAME sequence, every raised bit must arrive, one poll late at most."""
    port = _free_port()
    mp.spawn(_flag_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "f0.npy"), np.load(tmp_path / "f1.npy")
    assert np.array_equal(a, b), (a, b)
    assert a.tolist() == [0, 0, 1, 0, 2, 0], a      # raised at polls 1 and 3 on rank 1 -> agreed on at polls 2 and 4; nothing is pending at the final poll


def _local_poll_worker(rank, world, port, out_dir):
    """Runner._poll_field32_range on two ranks over gloo with the device flag stubbed: rank 1's flag shows bit 0 at a poll that follows a rendered image INSIDE the training loop"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jnerf_amd import ops
    from jnerf_amd.runner import Runner
    torch.cuda.is_available = lambda: True          # the poll is a no-op without a GPU; nothing below touches one (the two ops it calls are stubbed)

    class _S:
        device = torch.device("cpu")

    class _M:
        fused, fused_dtype = True, torch.float32
    r = Runner.__new__(Runner)
    r.sampler, r.model = _S(), _M()
    device_flag = [0]
    selected = []

    def check(reset=True, synchronize=True):
        v = device_flag[0]
        if reset:
            device_flag[0] = 0
        return v
    ops.field32_range_check, ops.field32_select = check, lambda exact: selected.append(bool(exact))
    trace = []

    def poll(**kw):
        Runner._poll_field32_range(r, **kw)
        trace.append(int(bool(getattr(r, "_field32_exact", False))))
    r._collective_polls = True                      # what train_step sets
    poll()                                          # step 16
    if rank == 1:
        device_flag[0] = 1                          # a training launch came within 4x of the range ...
    poll(local=True)                                # ... and val_img's poll is the one that reads it: must NOT switch this rank alone
    poll()                                          # step 32: the bit travels
    poll()                                          # step 48: both ranks act on it
    poll(local=True)                                # a later rendered image
    poll()                                          # step 64: a rank on the exact kernels still joins the collective
    Runner._poll_field32_range(r, final=True); trace.append(int(bool(getattr(r, "_field32_exact", False))))
    np.save(os.path.join(out_dir, f"l{rank}.npy"), np.asarray(trace + [len(selected)]))
    dist.destroy_process_group()


def test_poll_after_a_rendered_image_inside_training_never_acts_alone(tmp_path):
    """(r6) val_img runs on every rank inside the loop; its poll read the flag with reset and acted locally - one rank on the exact kernels, returning early
    from every later poll while its peer waits in the all-reduce.  Now the bits join the carry and the next collective poll decides: both ranks switch at the same poll, both
    keep polling, nobody hangs (a hang = the join timeout below)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_local_poll_worker, args=(rk, 2, port, str(tmp_path))) for rk in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, p.exitcode
    a, b = np.load(tmp_path / "l0.npy"), np.load(tmp_path / "l1.npy")
    assert np.array_equal(a, b), (a, b)
    assert a.tolist() == [0, 0, 0, 1, 1, 1, 1, 1], a      # switched once, at the poll after the bit was reduced, on BOTH ranks
