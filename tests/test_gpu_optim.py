"""mlsp_amd.optim.FlatAdam against torch.optim.Adam(fused=True): the optimizer of PointDA/trainer.py:258-260 (Adam + weight decay +
CosineAnnealingLR) as one launch over flat buffers -- the same element-wise arithmetic restated type by type: bit-identical."""
import copy

import pytest
import torch
from torch import nn

import golden_common as gc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


class _Net(nn.Module):
    """a used trunk, a head that never runs (DGCNN.Rec_scan in the default modes: no gradient, never stepped) and a late head"""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(37, 129)
        self.bn = nn.BatchNorm1d(129)
        self.unused = nn.Linear(129, 5)
        self.b = nn.Linear(129, 70001 // 129)       # a tensor that does not end on a chunk boundary
        self.late = nn.Linear(129, 3)

    def forward(self, x, late=False):
        h = torch.relu(self.bn(self.a(x)))
        out = self.b(h).sum()
        return out + self.late(h).sum() if late else out


def _pair(dev):
    torch.manual_seed(3)
    m1 = _Net().to(dev)
    m2 = copy.deepcopy(m1)
    return m1, m2


def _same(m1, m2):
    """bit-identical: csrc/optim.hip restates torch's element-wise update with its types AND its lowering (tools/r5/adam_probe)"""
    for (n, p), q in zip(m1.named_parameters(), m2.parameters()):
        assert torch.equal(p, q), (n, (p != q).sum().item(), p.numel())


def test_flat_adam_is_bit_identical_to_fused_adam(dev):
    from mlsp_amd.optim import FlatAdam
    m1, m2 = _pair(dev)
    o1 = FlatAdam(m1.parameters(), lr=1e-3, weight_decay=5e-5)
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=5e-5, fused=True)
    s1 = torch.optim.lr_scheduler.CosineAnnealingLR(o1, 10)
    s2 = torch.optim.lr_scheduler.CosineAnnealingLR(o2, 10)
    for it in range(6):
        x = torch.randn(64, 37, device=dev, generator=torch.Generator(device=dev).manual_seed(it))
        for m, o, s in ((m1, o1, s1), (m2, o2, s2)):
            o.zero_grad()
            m(x).backward()
            o.step()
            s.step()
        _same(m1, m2)
    assert o1.flat_steps == 6
    # a parameter that leaves the flat buffer (module.to(), an assignment to .data) is re-homed at the next step
    m1.a.weight.data = m1.a.weight.data.clone()
    x = torch.randn(64, 37, device=dev, generator=torch.Generator(device=dev).manual_seed(99))
    for m, o in ((m1, o1), (m2, o2)):
        o.zero_grad()
        m(x).backward()
        o.step()
    _same(m1, m2)
    # the model lives in ONE buffer, the head that never ran has no state and did not move
    ptrs = sorted((p.data_ptr(), p.numel()) for p in m1.parameters())
    assert all(a + 4 * n <= b < a + 4 * n + 4 * FlatAdam.ALIGN and b % 256 == 0 for (a, n), (b, _) in zip(ptrs, ptrs[1:]))
    assert m1.unused.weight not in o1.state or not o1.state[m1.unused.weight]
    assert m1.late.weight not in o1.state or not o1.state[m1.late.weight]
    # state_dict round trip through a fresh optimizer, then more steps on both sides
    sd = copy.deepcopy(o1.state_dict())
    o3 = FlatAdam(m1.parameters(), lr=1e-3, weight_decay=5e-5)
    o3.load_state_dict(sd)
    for it in range(6, 9):
        x = torch.randn(64, 37, device=dev, generator=torch.Generator(device=dev).manual_seed(it))
        for m, o in ((m1, o3), (m2, o2)):
            o.zero_grad()
            m(x).backward()
            o.step()
        _same(m1, m2)
    assert o3.flat_steps == 3
    for p, q in zip(m1.parameters(), m2.parameters()):
        if q in o2.state and o2.state[q]:
            assert float(o3.state[p]["step"]) == float(o2.state[q]["step"])
            assert torch.equal(o3.state[p]["exp_avg_sq"], o2.state[q]["exp_avg_sq"])


def test_flat_adam_falls_back_when_the_stepped_set_changes(dev):
    """a head that starts running later gets its own step counter in torch: the flat step (one shared counter) hands over to torch's
    per-tensor path on the same storage, and stays bit-identical"""
    from mlsp_amd.optim import FlatAdam
    m1, m2 = _pair(dev)
    o1 = FlatAdam(m1.parameters(), lr=2e-3, weight_decay=1e-4)
    o2 = torch.optim.Adam(m2.parameters(), lr=2e-3, weight_decay=1e-4, fused=True)
    for it in range(5):
        x = torch.randn(32, 37, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + it))
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            m(x, late=it >= 2).backward()
            o.step()
        _same(m1, m2)
    assert o1.flat_steps == 2
    assert float(o1.state[m1.late.weight]["step"]) == 3.0 and float(o1.state[m1.a.weight]["step"]) == 5.0


def test_flat_adam_reads_the_exchange_bucket_in_place(dev):
    """with FlatGradSync the optimizer reads the packed (all-reduced) gradients where the bucket holds them: no second copy"""
    from mlsp_amd.ddp import FlatGradSync
    from mlsp_amd.optim import FlatAdam
    m1, m2 = _pair(dev)
    sync = FlatGradSync(m1, force=True, align=4)
    o1 = sync.wrap(FlatAdam(m1.parameters(), lr=1e-3, weight_decay=5e-5))
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=5e-5, fused=True)
    for it in range(3):
        x = torch.randn(64, 37, device=dev, generator=torch.Generator(device=dev).manual_seed(it))
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            m(x).backward()
            o.step()
        _same(m1, m2)
        lo, hi = sync.flat.data_ptr(), sync.flat.data_ptr() + 4 * sync.flat.numel()
        assert all(p.grad is None or lo <= p.grad.data_ptr() < hi for p in m1.parameters())
    assert o1.flat_steps == 3


def test_flat_adam_keeps_shared_storages_together(dev):
    """The merged head layers make the parameters they read as ONE operand adjacent in one storage (functional.rehome_adjacent) and check
    that adjacency every forward; the flat layout moves such a storage as a unit, so the check keeps passing (no concatenation per step),
    odd-sized members included (element-wise path of the kernel).  A group that is re-homed AFTER the first step -- a head that runs for
    the first time -- triggers ONE rebuild of the layout, with the moments carried over."""
    from mlsp_amd import functional as Fh
    from mlsp_amd.optim import FlatAdam

    class Heads(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.c = nn.Linear(19, 33), nn.Linear(19, 33), nn.Linear(19, 7)      # 627- and 133-element weights: odd offsets
            self.d, self.e = nn.Linear(33, 5), nn.Linear(33, 5)

        def forward(self, x):
            h = torch.relu(self.a(x)) + torch.relu(self.b(x))
            return self.d(h).sum() + self.e(h).sum() + self.c(x).sum()

    torch.manual_seed(5)
    m1 = Heads().to(dev)
    m2 = copy.deepcopy(m1)
    assert Fh.rehome_adjacent([m1.a.weight, m1.b.weight, m1.c.weight]) and Fh.rehome_adjacent([m1.a.bias, m1.b.bias, m1.c.bias])
    o1 = FlatAdam(m1.parameters(), lr=1e-3, weight_decay=5e-5)
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=5e-5, fused=True)
    for it in range(6):
        if it == 3:
            assert Fh.rehome_adjacent([m1.d.weight, m1.e.weight])          # leaves the flat buffer: rebuilt at the next step
        x = torch.randn(16, 19, device=dev, generator=torch.Generator(device=dev).manual_seed(it))
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            m(x).backward()
            o.step()
        _same(m1, m2)
        assert Fh._adjacent([m1.a.weight, m1.b.weight, m1.c.weight]) and Fh._adjacent([m1.a.bias, m1.b.bias, m1.c.bias])
        if it >= 3:
            assert Fh._adjacent([m1.d.weight, m1.e.weight])
    assert o1.flat_steps == 6 and o1.layouts_built == 2
    for p, q in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(o1.state[p]["exp_avg"], o2.state[q]["exp_avg"]) and torch.equal(o1.state[p]["exp_avg_sq"], o2.state[q]["exp_avg_sq"])


def test_flat_adam_publishes_weight_bounds(dev):
    """Round 6: the step kernel leaves the largest magnitude of every updated 2048-element parameter tile (mlsp_adam_flat_f32 tile_amax);
    FlatAdam.weight_bounds hands a GEMM the run of tiles of the parameters its weight operand overlaps -- a bound of |W| that costs no
    measuring launch (functional._weight_bounds, the two-piece f16 products).  The maxima are exact for a whole parameter, an upper bound
    for a strided view; they are withdrawn when torch sees somebody else write the parameter, and by invalidate_bounds()."""
    from mlsp_amd import Models, functional as Fh, _lib
    from mlsp_amd.optim import FlatAdam
    torch.manual_seed(3)
    m = Models.DGCNN(gc.make_args(cuda=True)).to(dev).train()
    opt = FlatAdam(m.parameters(), lr=1e-3, weight_decay=5e-5)
    x = (torch.rand(4, 3, 256, device=dev) * 2 - 1)
    W5 = m.conv5.weight.view(1024, -1)
    assert opt.weight_bounds(W5) is None                                   # nothing published before the first step
    for _ in range(2):
        opt.zero_grad()
        out = m(x, activate_density_normal_ondef=True)
        sum(v.float().sum() for v in out.values()).backward()
        opt.step()
    assert opt in _lib.weight_bound_providers
    W5 = m.conv5.weight.view(1024, -1)
    ptr, n = opt.weight_bounds(W5)
    f = opt._flat
    t0 = (ptr - f["tile_amax"].data_ptr()) // 4
    assert n == (W5.numel() + 2047) // 2048 and 0 <= t0 and t0 + n <= f["tile_amax"].numel()
    assert f["tile_amax"][t0:t0 + n].max().item() == W5.abs().max().item()      # exact for a whole parameter
    # the three heads' first-layer weights as ONE strided operand (Models.merged_first_layers): adjacent in the buffer -> one run of tiles
    Wm = Fh.row_blocks([h.conv1.weight.view(h.conv1.out_channels, -1) for h in (m.DefRec, m.Norm_pred, m.Density_cls)], rehome=False)
    r = opt.weight_bounds(Wm[:, :512])
    assert r is not None
    t0 = (r[0] - f["tile_amax"].data_ptr()) // 4
    assert f["tile_amax"][t0:t0 + r[1]].max().item() >= Wm[:, :512].abs().max().item()
    assert opt.weight_bounds(torch.zeros(64, 64, device=dev)) is None      # not a view of the flat buffer
    with torch.no_grad():
        m.conv5.weight.mul_(1.5)                                            # torch sees this write: the bounds of conv5 are withdrawn ...
    assert opt.weight_bounds(m.conv5.weight.view(1024, -1)) is None
    assert opt.weight_bounds(Wm[:, :512]) is not None                       # ... the others stand
    opt.invalidate_bounds()
    assert opt.weight_bounds(Wm[:, :512]) is None
    # and the step after republishes, the forward in between having measured its weights itself
    opt.zero_grad()
    out = m(x, activate_density_normal_ondef=True)
    sum(v.float().sum() for v in out.values()).backward()
    opt.step()
    assert opt.weight_bounds(m.conv5.weight.view(1024, -1)) is not None
