"""world_size-2 gloo test (CPU) of the flat-bucket gradient exchange used for the N>1 path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mlsp_amd.ddp import FlatGradSync
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    model[3].bias.requires_grad = False                       # a frozen parameter stays out of the bucket
    model.add_module("unused", torch.nn.Linear(4, 4))         # a head that never runs: its grads stay None (DGCNN.Rec_scan)
    sync = FlatGradSync(model)
    opt = sync.wrap(torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.1))
    assert isinstance(opt, torch.optim.Optimizer)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 10)      # PointDA/trainer.py:260 on the wrapped optimizer
    assert sync.numel == sum(p.numel() for p in model.parameters() if p.requires_grad)
    model = model[:4]                                         # forward without the unused head
    g = torch.Generator().manual_seed(100 + rank)             # different shard per rank
    x, y = torch.randn(12, 8, generator=g), torch.randn(12, 4, generator=g)
    opt.zero_grad()
    ((model(x) - y) ** 2).mean().backward()
    ((model(x) - y) ** 2).mean().backward()                   # two backwards per step accumulate, ONE all-reduce
    local = sync.pack().clone()
    assert torch.equal(sync.pack(), local) and local.abs().sum() > 0     # a second pack() must not wipe the aliased gradients
    opt.step()
    sched.step()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    ok_avg = torch.allclose(sync.flat, want, atol=1e-6)
    ok_views = all(p.grad is None or p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync.views))
    ok_views = ok_views and sum(p.grad is None for p in sync.params) == 2
    w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    ws = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    ok_same = all(torch.equal(ws[0], t) for t in ws)          # replicas stay identical after the step
    opt.zero_grad()
    ok_zero = all(p.grad is None for p in sync.params)
    q.put((rank, ok_avg, ok_views, ok_same, ok_zero))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def _worker_divergent(rank, world, port, q):
    """Rank 1 does not run one head this step (a data-dependent branch: an empty target batch): rank 0 has gradients for it,
    rank 1 has None.  Every rank must still hand its optimizer the same parameters, or Adam moves the head on rank 0 only."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mlsp_amd.ddp import FlatGradSync
    torch.manual_seed(0)
    model = torch.nn.ModuleDict({"enc": torch.nn.Linear(8, 8), "head_a": torch.nn.Linear(8, 4), "head_b": torch.nn.Linear(8, 4),
                                 "never": torch.nn.Linear(8, 4)})
    sync = FlatGradSync(model, presence="exchange")           # data-dependent branches: the host-side bitmap exchange
    opt = sync.wrap(torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=5e-5))
    g = torch.Generator().manual_seed(200 + rank)
    ok = True
    for step in range(3):
        x = torch.randn(6, 8, generator=g)
        opt.zero_grad()
        h = torch.relu(model["enc"](x))
        loss = model["head_a"](h).pow(2).mean()
        if rank == 0 or step == 2:                           # steps 0, 1: only rank 0 runs head_b; step 2: both do
            loss = loss + model["head_b"](h).pow(2).mean()
        loss.backward()
        had_b = model["head_b"].weight.grad is not None
        opt.step()
        # head_b steps on BOTH ranks (rank 1 contributes zeros to the average); `never` is skipped on both
        ok = ok and model["head_b"].weight.grad is not None and model["never"].weight.grad is None
        ok = ok and (had_b or rank == 1)
        w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        ok = ok and all(torch.equal(ws[0], t) for t in ws)   # replicas identical after every step
    q.put((rank, ok, sync.collectives == 3))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_divergent_absent_gradients_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_divergent, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def test_force_runs_the_exchange_on_one_rank():
    """force=True: pack -> (collective when a group exists) -> divide also at world size 1, bit-identical to the plain step."""
    from mlsp_amd.ddp import FlatGradSync
    torch.manual_seed(0)
    a, b = torch.nn.Linear(5, 3), torch.nn.Linear(5, 3)
    b.load_state_dict(a.state_dict())
    sync = FlatGradSync(b, force=True)
    oa = torch.optim.Adam(a.parameters(), lr=1e-2)
    ob = sync.wrap(torch.optim.Adam(b.parameters(), lr=1e-2))
    x = torch.randn(7, 5)
    for _ in range(3):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            m(x).pow(2).mean().backward()
            m(2 * x).pow(2).mean().backward()
            o.step()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync.views))
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))


def test_single_process_is_identity():
    from mlsp_amd.ddp import FlatGradSync
    m = torch.nn.Linear(3, 2)
    s = FlatGradSync(m)
    m(torch.ones(4, 3)).sum().backward()
    g0 = [p.grad.clone() for p in m.parameters()]
    s.allreduce()                                             # world size 1: nothing is copied, grads untouched
    assert s.world_size == 1 and all(torch.equal(a, p.grad) for a, p in zip(g0, m.parameters()))
    flat = s.pack().clone()                                   # explicit packing still works and aliases .grad
    assert torch.equal(flat, torch.cat([g.reshape(-1) for g in g0]))
    m.zero_grad(set_to_none=True)                             # a trainer doing this must not break the bucket
    m(torch.ones(4, 3)).sum().backward()
    assert torch.equal(s.pack(), flat)


def test_pack_with_absent_gradients_is_idempotent():
    """a parameter without a gradient must not make a second pack() zero the aliased gradients."""
    from mlsp_amd.ddp import FlatGradSync
    m = torch.nn.ModuleDict({"a": torch.nn.Linear(3, 2), "unused": torch.nn.Linear(2, 2)})
    s = FlatGradSync(m)
    opt = s.wrap(torch.optim.Adam(m.parameters(), lr=1e-2))
    assert isinstance(opt, torch.optim.Optimizer) and s.wrap(opt) is opt
    torch.optim.lr_scheduler.CosineAnnealingLR(opt, 5)
    m["a"](torch.ones(4, 3)).sum().backward()
    first = s.pack().clone()
    assert first.abs().sum() > 0
    assert torch.equal(s.pack(), first) and torch.equal(s.pack(), first)
    assert all(p.grad is None for p in m["unused"].parameters())
    s.flat[-1] = 7.0                                          # stale bytes in an absent view are cleared, present ones kept
    assert torch.equal(s.pack(), first)
    opt.step()
    opt.step()
    opt.zero_grad()
    assert all(p.grad is None for p in m.parameters())


def _worker_dgcnn(rank, world, port, q):
    """The REAL DGCNN's parameter list through FlatGradSync (the slice of the N > 1 path that needs no kernels): 4,548,899 trainable
    elements in ONE bucket, Density_cls.fc2 (frozen, PointDA/Models.py:267-270) outside it, Rec_scan (a head the default modes never run)
    absent on every rank, a head that ran on rank 0 only present everywhere after the exchange, replicas identical after Adam."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import golden_common as gc
    from mlsp_amd import Models
    from mlsp_amd.ddp import FlatGradSync
    torch.manual_seed(0)
    model = Models.DGCNN(gc.make_args(dropout=0.5))            # CPU module: construction and state need no kernel
    sync = FlatGradSync(model, presence="exchange")            # (step 0 is rank-divergent on purpose)
    opt = sync.wrap(torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5))
    ok_size = sync.numel == 4548899 and len(sync.params) == sum(1 for p in model.parameters() if p.requires_grad)
    ok_frozen = all(p is not model.Density_cls.fc2.weight for p in sync.params)
    named = dict(model.named_parameters())
    g = torch.Generator().manual_seed(300 + rank)
    for step in range(2):
        opt.zero_grad()
        # synthetic gradients stand in for two backward passes of a step: everything but Rec_scan; the normal head on rank 0 only
        for n, p_ in named.items():
            if not p_.requires_grad or n.startswith("Rec_scan") or (n.startswith("Norm_pred") and rank == 1 and step == 0):
                continue
            p_.grad = torch.randn(p_.shape, generator=g)
        local = torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).reshape(-1) for p_ in sync.params])
        opt.step()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        ok_avg = torch.allclose(sync.flat, sum(gathered) / world, atol=1e-6)
        ok_present = model.Norm_pred.conv1.weight.grad is not None and model.Rec_scan.conv1.weight.grad is None
        w = torch.cat([p_.detach().reshape(-1) for p_ in model.parameters()])
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        ok_same = all(torch.equal(ws[0], t) for t in ws)
        if not (ok_avg and ok_present and ok_same):
            break
    q.put((rank, ok_size, ok_frozen, ok_avg, ok_present, ok_same, sync.collectives == 2))
    dist.barrier()
    dist.destroy_process_group()


def test_real_dgcnn_parameter_list_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dgcnn, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def _worker_uniform_check(rank, world, port, q):
    """Default policy "uniform": no host exchange; a step in which the ranks' gradient patterns differ is DETECTED at the next step
    (the pattern hashes ride in the gradient bucket) and raises."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mlsp_amd.ddp import FlatGradSync
    torch.manual_seed(0)
    model = torch.nn.ModuleDict({"enc": torch.nn.Linear(8, 8), "head_a": torch.nn.Linear(8, 4), "head_b": torch.nn.Linear(8, 4)})
    sync = FlatGradSync(model)
    assert sync.presence_mode == "uniform" and sync.host_group is None
    opt = sync.wrap(torch.optim.SGD(model.parameters(), lr=1e-2))
    g = torch.Generator().manual_seed(400 + rank)
    raised_at = None
    for step in range(4):
        x = torch.randn(6, 8, generator=g)
        opt.zero_grad()
        h = torch.relu(model["enc"](x))
        loss = model["head_a"](h).pow(2).mean()
        if not (step == 1 and rank == 1):                     # step 1: rank 1 skips head_b -- the patterns differ
            loss = loss + model["head_b"](h).pow(2).mean()
        loss.backward()
        try:
            opt.step()
        except RuntimeError as e:
            raised_at = (step, "presence='exchange'" in str(e))
            break
    q.put((rank, raised_at == (2, True)))
    dist.barrier()
    dist.destroy_process_group()


def test_uniform_presence_is_verified_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_uniform_check, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def _worker_verify_last_step(rank, world, port, q):
    """FlatGradSync.verify(): the in-band presence check runs one step late, so a divergence in the LAST step of a run is only seen by
    an explicit verify() (before a checkpoint / at the end of training).  Three 8-bit hashes: sums exact in fp32 for any reduction order."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mlsp_amd.ddp import FlatGradSync
    torch.manual_seed(0)
    model = torch.nn.ModuleDict({"enc": torch.nn.Linear(8, 8), "head_a": torch.nn.Linear(8, 4), "head_b": torch.nn.Linear(8, 4)})
    sync = FlatGradSync(model)
    assert sync._NCHK == 6 and all(0.0 <= h <= 255.0 for h in sync._local_hashes())
    opt = sync.wrap(torch.optim.SGD(model.parameters(), lr=1e-2))
    g = torch.Generator().manual_seed(500 + rank)
    ok_clean, raised = False, False
    for step in range(3):
        x = torch.randn(6, 8, generator=g)
        opt.zero_grad()
        h = torch.relu(model["enc"](x))
        loss = model["head_a"](h).pow(2).mean()
        if not (step == 2 and rank == 1):                     # the LAST step: rank 1 skips head_b
            loss = loss + model["head_b"](h).pow(2).mean()
        loss.backward()
        opt.step()                                            # (checks the step before: nothing to report yet)
        if step == 1:
            sync.verify()                                     # uniform so far: passes, and consumes the pending check
            ok_clean = True
    try:
        sync.verify()
    except RuntimeError as e:
        raised = "presence='exchange'" in str(e)
    q.put((rank, ok_clean, raised))
    dist.barrier()
    dist.destroy_process_group()


def test_verify_reports_a_divergence_in_the_last_step_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_verify_last_step, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r
