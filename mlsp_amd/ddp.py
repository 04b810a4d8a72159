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
    """One flat fp32 bucket for every trainable gradient, all-reduced once per optimizer step.

        sync = FlatGradSync(model)            # after model.to(device); on every rank (it may create a host-side group)
        opt = sync.wrap(torch.optim.Adam(model.parameters(), ...))
        ...  loss.backward() (any number of times) ...
        opt.step()                            # step pre-hook: flatten -> ONE all-reduce(sum) -> /world_size; then the step
        opt.zero_grad()                       # grads -> None (autograd then assigns instead of accumulating)

    Gradients are NOT pre-homed in the bucket during backward: autograd would then issue one `grad += new`
    kernel per parameter (77 launches, ~0.4 ms per step on MI355X).  Instead the gradients autograd produced
    are packed into the bucket by one multi-tensor copy right before the collective, and `.grad` is re-pointed
    at the bucket views for the optimizer.  With world_size == 1 nothing is copied at all (unless `force=True`,
    which runs the whole pack -> collective -> divide path on one rank: tests/test_gpu_ddp.py).

    Which parameters step.  A parameter whose gradient is None is skipped by the optimizer (a head that did not run:
    DGCNN.Rec_scan in the default modes) -- exactly what the reference's single process does.  Across ranks that set must be
    the SAME or the replicas drift apart (one rank applies the averaged gradient, the other nothing).  Two policies (`presence=`):

    "uniform" (default): every rank runs the same heads in a step -- true for PointDA/trainer.py, whose branches depend on flags, not
        on the data -- so nothing is exchanged on the host.  It is VERIFIED, not assumed: two 10-bit hashes of the local pattern (and
        their squares) ride at the end of the gradient bucket through the same RCCL all-reduce; the sums come back through a pinned
        buffer without a stream synchronisation and are checked at the NEXT step (by then the copy has long landed): if the ranks'
        patterns differed the step raises and names the remedy (`verify()` checks the LAST step: call it before a checkpoint / at the
        end of training).  Cost per step: 24 bytes in the bucket, one 24-byte async copy.  The sums are exact in fp32 for any
        summation order up to 256 ranks (ws * 255^2 < 2^24).
    "exchange": the ranks exchange a presence bitmap (one byte per parameter, MAX-reduced) on a HOST-side gloo group before packing: a
        parameter that has a gradient on ANY rank gets a (zero-filled) gradient on every rank -- for loops with data-dependent
        branches.  It is a BLOCKING host collective per step: 0.57 ms (world 2) / 3.2 ms (world 8, eight processes on eight cores)
        median over loopback on the build container (tools/r5/gloo_latency.py), against the ~1 ms the host runs ahead of the device
        at opt.step() (tools/r5/host_profile.py) -- which is why it is no longer the default.
    """

    _NCHK = 6                                 # trailing bucket elements of the uniform-presence check: (h, h^2) for three 8-bit hashes
    _MAX_UNIFORM_WORLD = 256                  # ws * 255^2 < 2^24: every partial sum of the ring is an exact fp32 integer

    def __init__(self, model, process_group=None, force=False, presence="uniform", align=1):
        # align: every gradient starts on a multiple of `align` elements of the bucket (zero padding between them; 4 = 16-byte aligned
        # gradient views: mlsp_amd.optim.FlatAdam, which reads the packed gradients in place, then moves 16 bytes per lane on every stream)
        if presence not in ("uniform", "exchange"):
            raise ValueError("presence must be 'uniform' or 'exchange'")
        from .optim import flat_offsets
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.group = process_group
        self.force = bool(force)
        self.presence_mode = presence
        self.numel = sum(p.numel() for p in self.params)      # gradients exchanged (the bucket may hold padding on top)
        self.offsets, padded = flat_offsets(self.params, int(align))
        dev = self.params[0].device
        self._bucket = torch.zeros(padded + self._NCHK, dtype=torch.float32, device=dev)
        self.flat = self._bucket[:padded]                     # the gradients; the check words sit behind them in the same allocation
        self._chk = self._bucket[padded:]
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]
        self.collectives = 0                  # device all-reduces issued so far (tests count them)
        self.steps = 0                        # optimizer steps this exchange ran in front of (bench.py: collectives / steps == 1)
        self.host_group = None
        self._pending = None                  # (host copy of the summed check words, event | None, world size) of the previous step
        if presence == "uniform" and self.world_size > self._MAX_UNIFORM_WORLD:
            raise ValueError("FlatGradSync(presence='uniform') checks its hashes in fp32: at most %d ranks" % self._MAX_UNIFORM_WORLD)
        if self.world_size > 1 and presence == "exchange":
            backend = dist.get_backend(self.group)
            # the presence bitmap travels host-side: the default group itself when it already is a CPU one
            self.host_group = self.group if backend == "gloo" else dist.new_group(backend="gloo")

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def presence(self):
        """[n_params] bool list: does ANY rank hold a gradient for the parameter this step (host-side MAX-reduce; "exchange" mode)."""
        local = torch.tensor([p.grad is not None for p in self.params], dtype=torch.uint8)
        if self.world_size > 1 and self.presence_mode == "exchange":
            dist.all_reduce(local, op=dist.ReduceOp.MAX, group=self.host_group)    # (host_group None = the default group, when that is a gloo one)
        return local.bool().tolist()

    def _local_hashes(self):
        import zlib
        bits = bytes(1 if p.grad is not None else 0 for p in self.params)
        c = zlib.crc32(bits)
        return float(c & 255), float((c >> 8) & 255), float((c >> 16) & 255)

    def _verify_previous(self):
        """Check the summed pattern hashes of the PREVIOUS step's all-reduce (uniform mode): equal patterns <=> ws * sum(h^2) == sum(h)^2."""
        if self._pending is None:
            return
        host, ev, ws = self._pending
        self._pending = None
        if ev is not None and not ev.query():
            ev.synchronize()
        v = [int(round(x)) for x in host.tolist()]
        if any(ws * q != s_ * s_ for s_, q in zip(v[0::2], v[1::2])):
            raise RuntimeError("FlatGradSync(presence='uniform'): in the previous step the ranks did not hold gradients for the same set of "
                               "parameters (a data-dependent branch skipped a head on some rank).  The replicas have diverged; construct "
                               "FlatGradSync(..., presence='exchange') for such loops.")

    def verify(self):
        """Check the LAST step's presence hashes now (uniform mode; the in-band check otherwise runs one step late, so the final step of a
        run would go unverified): call before saving a checkpoint and at the end of training.  Raises like the in-band check."""
        self._verify_previous()

    def pack(self, present=None):
        """Copy the present gradients into the bucket (absent ones count as zero) and alias .grad to the bucket.
        Idempotent: a gradient that already lives in its bucket view is left alone, and only the views of ABSENT
        parameters are zeroed (a head that did not run this step, e.g. DGCNN.Rec_scan in the default modes).
        `present` (from presence()): parameters some OTHER rank has a gradient for get their zeroed view as .grad here,
        so that every rank hands its optimizer the same set of parameters."""
        src, dst, absent = [], [], []
        for i, (v, p) in enumerate(zip(self.views, self.params)):
            if p.grad is None:
                absent.append(v)
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if absent:
            torch._foreach_zero_(absent)
        if src:
            torch._foreach_copy_(dst, src)
        for i, (v, p) in enumerate(zip(self.views, self.params)):
            if p.grad is not None or (present is not None and present[i]):
                p.grad = v
        return self.flat

    def allreduce(self):
        ws = self.world_size
        self.steps += 1
        if ws > 1 or self.force:
            uniform = self.presence_mode == "uniform"
            if uniform:
                self._verify_previous()
                self.pack(None)
                hs = self._local_hashes()
                self._chk.copy_(torch.tensor([w for h in hs for w in (h, h * h)], dtype=torch.float32), non_blocking=True)
            else:
                self.pack(self.presence())
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(self._bucket if uniform else self.flat, op=dist.ReduceOp.SUM, group=self.group)
                self.collectives += 1
                if uniform:                   # the summed check words travel back without a synchronisation; read at the next step
                    if self._bucket.is_cuda:
                        host = torch.empty(self._NCHK, dtype=torch.float32, pin_memory=True)
                        host.copy_(self._chk, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record()
                        self._pending = (host, ev, ws)
                    else:
                        self._pending = (self._chk.clone(), None, ws)
            self.flat.div_(ws)
        return self.flat

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def wrap(self, optimizer):
        """Install the exchange on `optimizer` itself (a step pre-hook) and return it: the object stays a real
        torch.optim.Optimizer, so `CosineAnnealingLR(opt, ...)` (PointDA/trainer.py:260), `opt.state_dict()` and
        `opt.param_groups` keep working.  `opt.zero_grad()` (set_to_none=True, torch's default) drops the bucket views
        again so that autograd assigns fresh gradients instead of accumulating into the bucket; with set_to_none=False the
        gradients stay in their (zeroed) views and later backwards accumulate there -- correct, one add per parameter slower."""
        if not getattr(optimizer, "_mlsp_flat_sync", None):
            optimizer.register_step_pre_hook(lambda *_a, **_k: (self.allreduce(), None)[1])
            optimizer._mlsp_flat_sync = self
        return optimizer
