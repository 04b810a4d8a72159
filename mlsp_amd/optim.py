"""Adam over ONE flat parameter buffer: the optimizer step of the hot path as a single multi-tensor launch.

PointDA/trainer.py:258-259 builds `optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.wd)`.  torch's fused Adam walks the 77
parameter tensors of DGCNN + heads in 64 Ki-element chunks, one chunk per workgroup: most tensors end in a mostly empty chunk, five to
six launches per step, 110 us at 1.2 TB/s for 127 MB of parameter / gradient / moment traffic (profiles/r5_*).  Here the parameters, their
gradients and both moments live in four flat fp32 buffers (the parameters become views, as in FlatGradSync's bucket), the gradients autograd
produced are read where they lie through a pointer table, and ONE launch of `mlsp_adam_flat_f32` (csrc/optim.hip: 2048-element tiles,
~2300 workgroups, the element-wise update of torch's fused kernel restated type by type) steps every parameter that holds a gradient.
tests/test_gpu_optim.py compares the parameters with `torch.optim.Adam(..., fused=True)` on the unflattened model step by step.
(MLSP_FLAT_ADAM_TORCH=1: the flat buffers stepped by `torch._fused_adam_` itself after one packing copy -- bit-identical to torch by
construction, chunk-limited like torch; kept for A/B.)

Semantics kept from torch: a parameter whose gradient is None is not stepped (DGCNN.Rec_scan in the default modes, Models.py:150) and gets
no state; `param_groups[0]["lr"]` is read every step (CosineAnnealingLR, trainer.py:260); `state_dict()` / `load_state_dict()` see per-
parameter `step` / `exp_avg` / `exp_avg_sq` entries (views of the flat moments).  The flat step needs one parameter group of fp32
parameters on one device and the SAME set of parameters holding gradients at every step (they share one step counter); anything else --
several groups, amsgrad / maximize, a parameter set that changes between steps -- falls back to torch's own per-tensor path for good, on
the same storage, with the state carried over.
"""
import ctypes
import os

import torch

from . import _lib

_TORCH_KERNEL = bool(os.environ.get("MLSP_FLAT_ADAM_TORCH"))


def flat_offsets(params, align=1):
    """(first element of every parameter, total elements) of a flat buffer that holds `params` back to back, each starting on a multiple
    of `align` elements.  The library's fast paths want 16-byte aligned weight rows (a torch allocation is; an arbitrary offset into a flat
    buffer is not: unaligned weights send every GEMM to its slow edge-tile instantiation), so parameter buffers use align = 64 (256 bytes);
    the padding elements are zeros that stay zeros under Adam (zero gradient, zero value)."""
    offs, n = [], 0
    for p in params:
        n = (n + align - 1) // align * align
        offs.append(n)
        n += p.numel()
    return offs, (n + align - 1) // align * align


class FlatAdam(torch.optim.Adam):
    ALIGN = 64

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_buffer=None):
        params = list(params)
        on_gpu = any(isinstance(p, torch.Tensor) and p.is_cuda for p in params) or any(
            isinstance(g, dict) and any(p.is_cuda for p in g["params"]) for g in params)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=on_gpu)
        self._flat = None              # (params, offsets, flat_p, flat_g, flat_m, flat_v, step) once built
        self._active = None            # indices (into the group's parameter list) of the parameters that step, fixed at the first step
        self._disabled = False         # True: torch's per-tensor path from now on
        self._grad_buffer = grad_buffer
        self.flat_steps = 0            # steps taken on the flat path (tests)

    # ---- layout ---------------------------------------------------------------------------------------------------------------
    def adopt_grad_buffer(self, flat, params, offsets):
        """Use `flat` (FlatGradSync's bucket: every trainable parameter's gradient in parameter order, FlatGradSync(align=FlatAdam.ALIGN))
        as the flat gradient buffer, so the exchange's pack is the only copy of the step.  Ignored when the layouts differ."""
        mine = [p for p in self.param_groups[0]["params"] if p.requires_grad] if len(self.param_groups) == 1 else None
        if mine is not None and len(mine) == len(params) and all(a is b for a, b in zip(mine, params)) and self._flat is None:
            offs, n = flat_offsets(mine, self.ALIGN)
            if offs == list(offsets) and flat.numel() == n:
                self._grad_buffer = flat

    def _eligible(self):
        if self._disabled or len(self.param_groups) != 1:
            return False
        g = self.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") or not g.get("fused"):
            return False
        ps = g["params"]
        dev = ps[0].device
        return all(p.dtype == torch.float32 and p.device == dev and p.is_cuda and not p.is_sparse for p in ps)

    def _build(self):
        ps = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        dev = ps[0].device
        offs, n = flat_offsets(ps, self.ALIGN)
        flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        views = [flat_p[o:o + p.numel()].view_as(p) for o, p in zip(offs, ps)]
        torch._foreach_copy_(views, [p.data for p in ps])
        for p, v in zip(ps, views):
            p.data = v                                    # the model now lives in the flat buffer (module.to() / load_state_dict copy into it)
        gb = self._grad_buffer
        flat_g = gb if (gb is not None and gb.numel() == n and gb.device == dev) else (
            torch.zeros(n, dtype=torch.float32, device=dev) if _TORCH_KERNEL else flat_p[:0])      # (the own kernel needs no packed gradients)
        flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        active = [i for i, p in enumerate(ps) if p.grad is not None]
        # state that already exists (load_state_dict before the first flat step): carried over when every active parameter is at the same step
        steps = {float(self.state[ps[i]]["step"]) for i in active if ps[i] in self.state and "step" in self.state[ps[i]]}
        have = [i for i in active if ps[i] in self.state and "exp_avg" in self.state[ps[i]]]
        if len(steps) > 1 or (have and len(have) != len(active)) or any(p in self.state and self.state[p] for i, p in enumerate(ps) if i not in active):
            return None
        step = torch.full((), steps.pop() if steps else 0.0, dtype=torch.float32, device=dev)
        for i in active:
            p, o = ps[i], offs[i]
            m, v = flat_m[o:o + p.numel()].view_as(p), flat_v[o:o + p.numel()].view_as(p)
            if i in have:
                m.copy_(self.state[p]["exp_avg"])
                v.copy_(self.state[p]["exp_avg_sq"])
            self.state[p] = {"step": step, "exp_avg": m, "exp_avg_sq": v}
        # maximal runs of adjacent active parameters: the tensors the one launch walks
        runs, i = [], 0
        while i < len(active):
            j = i
            while j + 1 < len(active) and active[j + 1] == active[j] + 1:
                j += 1
            runs.append((offs[active[i]], offs[active[j]] + ps[active[j]].numel()))
            i = j + 1
        gviews = [flat_g[o:o + p.numel()].view_as(p) for o, p in zip(offs, ps)] if flat_g.numel() == n else None
        self._active = active
        return {"params": ps, "offs": offs, "gviews": gviews, "p": flat_p, "g": flat_g, "m": flat_m, "v": flat_v, "step": step, "runs": runs}

    def _leave_flat(self):
        """torch's per-tensor path from now on (same storage): every stepped parameter gets its own step counter."""
        if self._flat is not None:
            for i in self._active:
                st = self.state[self._flat["params"][i]]
                st["step"] = st["step"].clone()
        self._flat, self._disabled = None, True

    # ---- the step -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._eligible():
            if self._flat is not None:
                self._leave_flat()
            super().step()
            return loss
        if self._flat is None:
            self._flat = self._build()
            if self._flat is None:                        # loaded state that does not fit one shared step counter
                self._disabled = True
                super().step()
                return loss
        f = self._flat
        ps, gv = f["params"], f["gviews"]
        act = self._active
        base, offs = f["p"].data_ptr(), f["offs"]
        if any(ps[i].data_ptr() != base + 4 * offs[i] for i in act):
            # the model left the flat buffer (module.to(), a parameter re-assigned): re-home it, moments and step counter stay
            views = [f["p"][offs[i]:offs[i] + p.numel()].view_as(p) for i, p in enumerate(ps)]
            torch._foreach_copy_(views, [p.data for p in ps])
            for p, v in zip(ps, views):
                p.data = v
        n_with = sum(1 for p in ps if p.grad is not None)
        if n_with != len(act) or any(ps[i].grad is None for i in act):
            self._leave_flat()                            # another set of parameters holds gradients this step: per-tensor semantics
            super().step()
            return loss
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        if not _TORCH_KERNEL:
            # one launch; every gradient is read where autograd (or the exchange's bucket) left it
            grads = [ps[i].grad if ps[i].grad.is_contiguous() else ps[i].grad.contiguous() for i in act]
            n = len(act)
            if f.get("seg") is None:
                f["seg"] = ((ctypes.c_uint32 * n)(*[f["offs"][i] for i in act]), (ctypes.c_uint32 * n)(*[ps[i].numel() for i in act]))
                f["host_step"] = int(round(float(f["step"])))
            f["host_step"] += 1
            gp = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
            lr = grp["lr"]
            _lib.check(_lib.load().mlsp_adam_flat_f32(f["p"].data_ptr(), f["m"].data_ptr(), f["v"].data_ptr(), f["seg"][0], f["seg"][1], gp, n,
                                                      float(lr), float(b1), float(b2), float(grp["weight_decay"]), float(grp["eps"]),
                                                      f["host_step"], f["step"].data_ptr(), _lib.stream()), "mlsp_adam_flat_f32")
            self.flat_steps += 1
            return loss
        src, dst = [], []
        for i in act:
            g = ps[i].grad
            if g.data_ptr() != gv[i].data_ptr():
                src.append(g)
                dst.append(gv[i])
        if src:
            torch._foreach_copy_(dst, src)
            for i in act:
                ps[i].grad = gv[i]
        f["step"] += 1
        runs = f["runs"]
        torch._fused_adam_([f["p"][a:b] for a, b in runs], [f["g"][a:b] for a, b in runs], [f["m"][a:b] for a, b in runs],
                           [f["v"][a:b] for a, b in runs], [], [f["step"]] * len(runs), lr=grp["lr"], beta1=b1, beta2=b2,
                           weight_decay=grp["weight_decay"], eps=grp["eps"], amsgrad=False, maximize=False, grad_scale=None, found_inf=None)
        self.flat_steps += 1
        return loss

    def load_state_dict(self, state_dict):
        # loaded moments are fresh tensors: rebuild the flat buffers from them at the next step (parameters stay where they are)
        if self._flat is not None:
            self._leave_flat()
            self._disabled = False
        super().load_state_dict(state_dict)
        self._flat = None
