"""Data-parallel gradient exchange: ONE flat fp32 all-reduce per optimizer step over RCCL/xGMI.

The reference's only multi-GPU mechanism is single-process nn.DataParallel (PointDA/trainer.py:251-252:
per-forward parameter broadcast + per-backward reduce onto GPU 0, per-replica BatchNorm statistics).
MI355X-native replacement: one process per GPU, parameters replicated, per-rank BN statistics (same
semantics as DataParallel's replicas), and a single all-reduce(sum) of one contiguous bucket holding all
4,548,899 trainable gradients (18.2 MB) per step -- issued once, after the LAST backward of the step
(the trainer calls backward() 2-6 times per step, trainer.py:392-566).  xGMI is point-to-point, so one
large collective beats many small ones; 18.2 MB rides a ring in ~0.2 ms.
"""
import torch
import torch.distributed as dist


class FlatGradSync:
    """Re-homes every trainable parameter's .grad into one flat buffer and all-reduces it once.

        sync = FlatGradSync(model)            # after model.to(device)
        opt = sync.wrap(torch.optim.Adam(model.parameters(), ...))
        ...  loss.backward() (any number of times) ...
        opt.step()                            # all-reduce(sum)/world_size, then the optimizer step
        opt.zero_grad()                       # zeroes the bucket in place (grads stay views)
    """

    def __init__(self, model, process_group=None):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.group = process_group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.numel = n

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def check_views(self):
        """Gradients must still alias the bucket (zero_grad(set_to_none=True) would break that)."""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                return False
            off += p.numel()
        return True

    def rehome(self):
        off = 0
        for p in self.params:
            view = self.flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            p.grad = view
            off += p.numel()

    def allreduce(self):
        if not self.check_views():
            self.rehome()
        ws = self.world_size
        if ws > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(ws)
        return self.flat

    def zero_grad(self):
        self.flat.zero_()

    def wrap(self, optimizer):
        return _SyncedOptimizer(optimizer, self)


class _SyncedOptimizer:
    """Optimizer facade: step() = one flat all-reduce + inner step; zero_grad() keeps the bucket views."""

    def __init__(self, inner, sync):
        self.inner, self.sync = inner, sync

    def step(self, *a, **kw):
        self.sync.allreduce()
        return self.inner.step(*a, **kw)

    def zero_grad(self, set_to_none=False):
        self.sync.zero_grad()

    def __getattr__(self, name):
        return getattr(self.inner, name)
