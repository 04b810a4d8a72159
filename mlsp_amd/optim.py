"""Adam over ONE flat parameter buffer: the optimizer step of the hot path as a single launch.

PointDA/trainer.py:258-259 builds `optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.wd)`.  torch's fused Adam walks the 77
parameter tensors of DGCNN + heads in 64 Ki-element chunks, one chunk per workgroup: most tensors end in a mostly empty chunk, five to
six launches per step, 110 us at 1.2 TB/s for 127 MB of parameter / gradient / moment traffic (profiles/r5_*).  Here the parameters and
both moments live in three flat fp32 buffers (the parameters become views), the gradients autograd produced are read where they lie through a
pointer table, and ONE launch of `mlsp_adam_flat_f32` (csrc/optim.hip: 2048-element tiles, ~2300 workgroups, the element-wise update of
torch's fused kernel restated type by type and lowering by lowering) steps every parameter that holds a gradient: 19 us.
tests/test_gpu_optim.py asserts bit-identity with `torch.optim.Adam(..., fused=True)` on the unflattened model step by step.
(torch's own fused kernel on the same flat buffers, as one tensor, measured 88 us: its chunking is per 64 Ki elements whatever the
tensor list looks like.)

Layout.  Parameters that already SHARE a storage keep their relative places: the merged head layers make the parameters they read as one
operand adjacent in one buffer (functional.rehome_adjacent), and moving them apart again would cost a concatenation per forward.  Every such
storage -- and every stand-alone parameter -- becomes one unit, placed on a 256-byte boundary (an arbitrary offset into a flat buffer would
send every GEMM that reads the weight to its unaligned edge-tile instantiation: 4.48 -> 6.10 ms per step measured).  If parameters leave the
buffer later (module.to(), another rehome_adjacent) the layout is rebuilt at the next step, moments and step counter carried over.

Semantics kept from torch: a parameter whose gradient is None is not stepped (DGCNN.Rec_scan in the default modes, Models.py:150) and gets
no state; `param_groups[0]["lr"]` is read every step (CosineAnnealingLR, trainer.py:260); `state_dict()` / `load_state_dict()` see per-
parameter `step` / `exp_avg` / `exp_avg_sq` entries (views of the flat moments).  The flat step needs one parameter group of fp32
parameters on one device and the SAME set of parameters holding gradients at every step (they share one step counter); anything else --
several groups, amsgrad / maximize, a parameter set that changes between steps -- falls back to torch's own per-tensor path for good, on
the same storage, with the state carried over.
"""
import ctypes

import torch

from . import _lib


def flat_offsets(params, align=1):
    """(first element of every parameter, total elements) of a flat buffer that holds `params` back to back, each starting on a multiple
    of `align` elements (FlatGradSync's bucket layout)."""
    offs, n = [], 0
    for p in params:
        n = (n + align - 1) // align * align
        offs.append(n)
        n += p.numel()
    return offs, (n + align - 1) // align * align


def storage_unit_offsets(params, align):
    """Offsets for `params` in a flat buffer in which parameters that share a storage keep their relative byte offsets (one unit per
    shared storage, spanning its members; stand-alone parameters are units of their own), every unit starting on a multiple of `align`
    elements.  -> (offsets in elements, total elements), or None when some parameter is not a dense contiguous fp32 tensor."""
    units, order = {}, []
    for i, p in enumerate(params):
        if not p.is_contiguous() or p.dtype != torch.float32:
            return None
        key = p.untyped_storage().data_ptr()
        if key not in units:
            units[key] = []
            order.append(key)
        units[key].append(i)
    offs, n = [0] * len(params), 0
    for key in order:
        members = units[key]
        lo = min(params[i].data_ptr() for i in members)
        hi = max(params[i].data_ptr() + 4 * params[i].numel() for i in members)
        if len(members) == 1 or (hi - lo) // 4 > 8 * sum(params[i].numel() for i in members):
            # a lone parameter -- or members scattered over a big foreign storage (views of something else; span > 8x the content,
            # both in elements): place them one by one
            for i in members:
                n = (n + align - 1) // align * align
                offs[i] = n
                n += params[i].numel()
            continue
        n = (n + align - 1) // align * align
        for i in members:
            d = params[i].data_ptr() - lo
            if d % 4:
                return None
            offs[i] = n + d // 4
        n += (hi - lo) // 4
    # the members of a unit are distinct parameters of one concatenation: they must not overlap
    spans = sorted((offs[i], offs[i] + p.numel()) for i, p in enumerate(params))
    if any(a[1] > b[0] for a, b in zip(spans, spans[1:])):
        return None
    return offs, (n + align - 1) // align * align


class FlatAdam(torch.optim.Adam):
    ALIGN = 64

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(params)
        on_gpu = any(isinstance(p, torch.Tensor) and p.is_cuda for p in params) or any(
            isinstance(g, dict) and any(p.is_cuda for p in g["params"]) for g in params)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=on_gpu)
        self._flat = None              # the flat buffers and their layout once built (dict)
        self._active = None            # indices (into the trainable parameter list) of the parameters that step, fixed at the first step
        self._disabled = False         # True: torch's per-tensor path from now on
        self.flat_steps = 0            # steps taken on the flat path (tests)
        self.layouts_built = 0         # (tests: a rebuild happens only when parameters left the buffer)

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
        """Lay the trainable parameters out (storage units, see the module docstring), move them and whatever Adam state exists into
        fresh flat buffers.  None: the existing state does not fit one shared step counter."""
        ps = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        dev = ps[0].device
        lay = storage_unit_offsets(ps, self.ALIGN)
        if lay is None:
            lay = flat_offsets(ps, self.ALIGN)
        offs, n = lay
        active = self._active if self._active is not None else [i for i, p in enumerate(ps) if p.grad is not None]
        # (segments -- and with them the step kernel's tiles -- in ADDRESS order: parameters that one GEMM reads as a single operand, adjacent
        # in the buffer but not in model.parameters(), then own one contiguous run of tile maxima: weight_bounds)
        active = sorted(active, key=lambda i: offs[i])
        steps = {float(self.state[ps[i]]["step"]) for i in active if ps[i] in self.state and "step" in self.state[ps[i]]}
        have = [i for i in active if ps[i] in self.state and "exp_avg" in self.state[ps[i]]]
        if len(steps) > 1 or (have and len(have) != len(active)) or any(p in self.state and self.state[p] for i, p in enumerate(ps) if i not in active):
            return None
        flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        views = [flat_p[o:o + p.numel()].view_as(p) for o, p in zip(offs, ps)]
        torch._foreach_copy_(views, [p.data for p in ps])
        for p, v in zip(ps, views):
            p.data = v                                    # the model now lives in the flat buffer (load_state_dict copies into it)
        step = torch.full((), steps.pop() if steps else 0.0, dtype=torch.float32, device=dev)
        host_step = int(round(float(step)))
        for i in active:
            p, o = ps[i], offs[i]
            m, v = flat_m[o:o + p.numel()].view_as(p), flat_v[o:o + p.numel()].view_as(p)
            if i in have:
                m.copy_(self.state[p]["exp_avg"])
                v.copy_(self.state[p]["exp_avg_sq"])
            self.state[p] = {"step": step, "exp_avg": m, "exp_avg_sq": v}
        self._active = active
        self.layouts_built += 1
        n_act = len(active)
        seg = ((ctypes.c_uint32 * n_act)(*[offs[i] for i in active]), (ctypes.c_uint32 * n_act)(*[ps[i].numel() for i in active]))
        # per-tile maxima of the updated parameters (csrc/optim.hip tile_amax): tiles run segment by segment in the order of `active`
        tile_begin, t = [], 0
        for i in active:
            tile_begin.append(t)
            t += (ps[i].numel() + 2047) // 2048
        tile_begin.append(t)
        return {"params": ps, "offs": offs, "p": flat_p, "m": flat_m, "v": flat_v, "step": step, "host_step": host_step, "seg": seg,
                "tile_amax": torch.zeros(t, dtype=torch.float32, device=dev), "tile_begin": tile_begin, "amax_versions": None, "amax_map": {}}

    def _leave_flat(self, why=None):
        """torch's per-tensor path from now on (same storage): every stepped parameter gets its own step counter.  `why`: warn once --
        the results are the same, the one-launch step is lost."""
        if why and not self._disabled:
            import warnings
            warnings.warn("FlatAdam: leaving the one-launch flat step for torch's per-tensor fused Adam (%s); same results, but the "
                          "optimizer step takes ~5 launches / ~0.1 ms instead of 1 / ~0.02 ms from now on" % why, RuntimeWarning, stacklevel=3)
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
                self._leave_flat("the parameter groups / options are no longer the ones the flat step covers")
            super().step()
            return loss
        if self._flat is None:
            self._flat = self._build()
            if self._flat is None:                        # loaded state that does not fit one shared step counter
                self._disabled = True
                super().step()
                return loss
        f = self._flat
        ps, act = f["params"], self._active
        n_with = sum(1 for p in ps if p.grad is not None)
        if n_with != len(act) or any(ps[i].grad is None for i in act):
            # another set of parameters holds gradients this step (a head that runs only in some steps): per-tensor semantics
            self._leave_flat("the set of parameters holding gradients changed between steps")
            super().step()
            return loss
        base, offs = f["p"].data_ptr(), f["offs"]
        if any(p.data_ptr() != base + 4 * offs[i] for i, p in enumerate(ps)):
            # parameters left the buffer (module.to(), a rehome_adjacent of a head that ran for the first time): lay them out again
            # around their new storages, moments and step counter carried over
            f = self._flat = self._build()
            if f is None:
                self._disabled = True
                super().step()
                return loss
            ps, act = f["params"], self._active            # (the rebuild orders the segments by their NEW addresses)
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        # one launch; every gradient is read where autograd (or the exchange's bucket) left it
        grads = [ps[i].grad if ps[i].grad.is_contiguous() else ps[i].grad.contiguous() for i in act]
        n = len(act)
        f["host_step"] += 1
        gp = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
        _lib.check(_lib.load().mlsp_adam_flat_f32(f["p"].data_ptr(), f["m"].data_ptr(), f["v"].data_ptr(), f["seg"][0], f["seg"][1], gp, n,
                                                  float(grp["lr"]), float(b1), float(b2), float(grp["weight_decay"]), float(grp["eps"]),
                                                  f["host_step"], f["step"].data_ptr(), f["tile_amax"].data_ptr(), _lib.stream()), "mlsp_adam_flat_f32")
        # the kernel left the magnitude of every updated parameter tile: valid for as long as nobody else writes the parameters (an
        # in-place torch operation bumps the tensor's version counter -- the raw update above does not)
        f["amax_versions"] = [ps[i]._version for i in act]
        _lib.weight_bound_providers.add(self)
        self.flat_steps += 1
        return loss

    def invalidate_bounds(self):
        """Forget the parameter magnitudes the last step left (call after writing parameters in a way torch's version counters do not
        see: in-place operations on `.data`); the GEMMs measure their weights again until the next step."""
        if self._flat is not None:
            self._flat["amax_versions"] = None

    def weight_bounds(self, W):
        """(device pointer, n) of ready-made partial maxima bounding |W| for a GEMM operand W that lies inside the flat parameter buffer --
        the tiles of the parameters it overlaps, as the last step's kernel left them -- or None (not in the buffer, a parameter without
        a gradient in between, or written by someone else since the step: the caller measures)."""
        f = self._flat
        if f is None or f.get("amax_versions") is None or W.dim() != 2 or W.dtype != torch.float32 or W.stride(1) != 1:
            return None
        base, nbytes = f["p"].data_ptr(), f["p"].numel() * 4
        a = W.data_ptr() - base
        if a < 0 or a >= nbytes or a % 4:
            return None
        key = (a, W.shape[0], W.shape[1], W.stride(0))
        hit = f["amax_map"].get(key)
        if hit is None:
            a //= 4
            b = a + (W.shape[0] - 1) * W.stride(0) + W.shape[1]            # one past the last element W touches
            ps, offs, act = f["params"], f["offs"], self._active
            cover = [j for j, i in enumerate(act) if offs[i] < b and offs[i] + ps[i].numel() > a]
            ok = bool(cover) and cover == list(range(cover[0], cover[-1] + 1)) and a >= offs[act[cover[0]]] and b <= offs[act[cover[-1]]] + ps[act[cover[-1]]].numel()
            # (every element of W must lie inside some ACTIVE parameter: alignment gaps between units hold zeros, which is fine, but a
            # parameter that is not stepped has no tile)
            if ok:
                spans = sorted((offs[act[j]], offs[act[j]] + ps[act[j]].numel()) for j in cover)
                inactive = [(offs[i], offs[i] + p.numel()) for i, p in enumerate(ps) if i not in set(act)]
                ok = not any(lo < b and hi > a for lo, hi in inactive)
            hit = f["amax_map"][key] = (cover[0], cover[-1]) if ok else False
        if hit is False:
            return None
        j0, j1 = hit
        ps, act, ver = f["params"], self._active, f["amax_versions"]
        if any(ps[act[j]]._version != ver[j] for j in range(j0, j1 + 1)):
            return None
        t0, t1 = f["tile_begin"][j0], f["tile_begin"][j1 + 1]
        if t1 - t0 > 4096:
            return None
        return f["tile_amax"].data_ptr() + 4 * t0, t1 - t0

    def load_state_dict(self, state_dict):
        # loaded moments are fresh tensors: rebuild the flat buffers from them at the next step (parameters stay where they are)
        if self._flat is not None:
            self._leave_flat()
            self._disabled = False
        super().load_state_dict(state_dict)
        self._flat = None
        self._active = None
