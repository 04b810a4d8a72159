"""The N>1 gradient exchange on the real device path: a world-size-1 RCCL group on the GPU box drives the real DGCNN
through FlatGradSync(force=True), so pack -> RCCL all_reduce -> divide -> fused Adam runs exactly as it does on N ranks.
(The multi-rank semantics -- averaging, identical replicas, rank-divergent absent gradients -- are covered on CPU by
tests/test_ddp_gloo.py at world size 2.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import golden_common as gc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_world1():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield torch.device("cuda:0")
    dist.destroy_process_group()


def _losses(mlsp, args, logits, b):
    loss = mlsp.calc_loss(args, logits, b["gold"], b["mask"])
    mask_cord = b["mask"].permute(0, 2, 1)[:, :, 0] * 26 + 1
    loss = loss + mlsp.calc_masked_normal_loss(args, logits["Normal"], b["normal_gt"], mask_cord)
    kl, mae = mlsp.densityloss(args, logits, b["dens_val"], b["dens_vec"], mask=mask_cord.reshape(-1))
    return loss + kl + mae


def test_one_rccl_allreduce_per_step_bit_identical(rccl_world1):
    """Two backwards per optimizer step (source + target branch, PointDA/trainer.py:401,566) -> exactly ONE RCCL all-reduce of the
    18.2 MB bucket per step; parameters after three steps are bit-identical to the same steps without the exchange."""
    from mlsp_amd import Models, mlsp
    from mlsp_amd.ddp import FlatGradSync
    dev = rccl_world1
    args = gc.make_args(dropout=0.0, cuda=True)
    B, N = 4, 256
    batches = [{k: v.to(dev) for k, v in gc.make_inputs(s, B, N).items()} for s in (11, 12)]

    calls = []
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls.append(t.numel())
        return real_all_reduce(t, *a, **k)

    models, syncs = [], []
    for hooked in (False, True):
        torch.manual_seed(3)
        model = Models.DGCNN(args).to(dev).train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5, fused=True)
        sync = None
        if hooked:
            sync = FlatGradSync(model, force=True)
            opt = sync.wrap(opt)
            dist.all_reduce = counting_all_reduce
        try:
            for step in range(3):
                opt.zero_grad()
                src = model(batches[0]["x"], activate_DefRec=False)                    # backward #1: classifier only
                torch.nn.functional.cross_entropy(src["cls"], batches[0]["cls_label"].to(dev)).backward()
                logits = model(batches[1]["x"], activate_density_normal_ondef=True)     # backward #2: the three heads
                _losses(mlsp, args, logits, batches[1]).backward()
                opt.step()
        finally:
            dist.all_reduce = real_all_reduce
        torch.cuda.synchronize()
        models.append(model)
        syncs.append(sync)

    sync = syncs[1]
    # (the bucket carries the check words behind the gradients: the pattern hashes of the uniform-presence verification, ddp.py)
    assert sync.world_size == 1 and sync.collectives == 3 and calls == [sync.numel + sync._NCHK] * 3, (sync.collectives, calls)
    assert sync.numel == sum(p.numel() for p in models[1].parameters() if p.requires_grad) == 4548899   # 18.2 MB
    absent = [n for n, p in models[1].named_parameters() if p.requires_grad and p.grad is None]
    assert absent and all(n.startswith("Rec_scan.") for n in absent), absent       # the head that never runs is skipped, not zero-stepped
    for (n, a), (_, b) in zip(models[0].named_parameters(), models[1].named_parameters()):
        assert torch.equal(a, b), n
    for (n, a), (_, b) in zip(models[0].named_buffers(), models[1].named_buffers()):
        assert torch.equal(a, b), n
