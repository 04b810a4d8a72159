"""GPU parity tests, module level: mlsp_amd.Models.DGCNN + mlsp_amd.mlsp losses against the golden
vectors captured from the reference and against the CPU oracle.  Run with `-m gpu` on an MI355X."""
import copy
import os

import numpy as np
import pytest
import torch

import golden_common as gc
from oracle import knn_canon, ref_cpu

pytestmark = pytest.mark.gpu

HEAD_KEYS = ("cls", "DefRec", "Normal", "density", "density_mse")
FIXTURES = [("dgcnn_s0_B6_N256.npz", 0, 6, 256), ("dgcnn_s1_B4_N1024.npz", 1, 4, 1024), ("dgcnn_s2_B4_N128.npz", 2, 4, 128)]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mlsp_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _model(seed, dev, dropout=0.0):
    from mlsp_amd import Models
    torch.manual_seed(seed)
    m = Models.DGCNN(gc.make_args(dropout=dropout, cuda=True))
    gc.perturb_params(m, seed)
    return m.to(dev)


def _gpu_total_loss(args, logits, inp):
    from mlsp_amd import mlsp
    loss_def = mlsp.calc_loss(args, logits, inp["gold"], inp["mask"])
    mask_cord = inp["mask"].permute(0, 2, 1)[:, :, 0] * 26 + 1
    nl = mlsp.calc_masked_normal_loss(args, logits["Normal"], inp["normal_gt"], mask_cord)
    kl, mae = mlsp.densityloss(args, logits, inp["dens_val"], inp["dens_vec"], mask=mask_cord.reshape(-1))
    ce = torch.nn.functional.cross_entropy(logits["cls"], inp["cls_label"])
    return loss_def + nl + kl + mae + ce, dict(DefRec=loss_def, normal=nl, kl=kl, mae=mae, ce=ce)


def _oracle_selections(m, inp, graphs, dtype=torch.float32):
    """arg-max of every max-pool of the reference's arithmetic (oracle/ref_cpu.py, pinned to the goldens) on the given graphs, in the
    HIP path's call order and layouts (Fh.forced_selections)."""
    params = {k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu()) for k, v in m.state_dict().items()}
    it = iter(graphs)
    with torch.no_grad():
        _, _, ctx = ref_cpu.dgcnn_forward(params, inp["x"].cpu().to(dtype), training=True, dropout_p=0.0, knn_fn=lambda x_, k_: next(it),
                                          activate_density_normal_ondef=True, return_ctx=True)
    return [a.clone() for a in ctx.sel]


# Gradient pins against the reference's fp32 goldens.  With the reference's neighbour graphs AND its max-pool selections forced
# (Fh.forced_graphs / Fh.forced_selections) nothing discrete is left to differ, so the pin measures the kernels' arithmetic; free
# selections add the re-routing of near-tied maxima, which at N >= 256 is the larger term.  What is left with
# everything forced is the conditioning of the step itself: the reference's own fp32 run sits up to ~1e-2 from the float64 value of
# the same gradients at these sizes (measured per fixture below: `fp32 golden vs float64`), so two correct fp32 evaluations that round
# differently (the f32-MFMA kernels, the split-product kernels) can only be held to that distance -- the bar is
# max(floor, 3 x the golden's own distance from float64), per gradient, against the FLOAT64 oracle.
GRAD_FLOOR_FORCED = {"dgcnn_s2": 2e-3, "dgcnn_s0": 5e-3, "dgcnn_s1": 5e-3}
GRAD_TOL_FREE = {"dgcnn_s2": 2e-2, "dgcnn_s0": 5e-2, "dgcnn_s1": 5e-2}


def _oracle_grads_f64(m, inp, graphs, keys):
    """gradients of the trainer's summed loss in float64 (oracle/ref_cpu.py on the model's parameters, given graphs)"""
    ref = copy.deepcopy(m).cpu().double()
    params = dict(ref.state_dict(keep_vars=True))
    inp64 = {k: (v.cpu().double() if v.is_floating_point() else v.cpu()) for k, v in inp.items()}
    it = iter(graphs)
    out, _ = ref_cpu.dgcnn_forward(params, inp64["x"], training=True, dropout_p=0.0, knn_fn=lambda x_, k_: next(it),
                                   activate_density_normal_ondef=True)
    loss, _ = gc.total_loss(gc.make_args(), ref_cpu, out, inp64)
    loss.backward()
    named = dict(ref.named_parameters())
    return {k: named[k].grad.numpy() for k in keys}


@pytest.mark.parametrize("fname,seed,B,N", FIXTURES)
@pytest.mark.parametrize("tnet_fwd", ["default", "f32-products"])
def test_dgcnn_vs_reference_golden(dev, golden_dir, fname, seed, B, N, tnet_fwd):
    """With the reference's neighbour indices forced at the five graph stages (the dynamic graph is
    discontinuous, see tests/test_oracle_golden.py), logits / losses / grads / running stats of the HIP
    model match the reference within 1e-3.  Runs in the default product mode (bf16x6: the split-products GEMM and T-Net forward kernels
    the bench times -- no size gate any more) and with every product on the f32 MFMA; gradients once with the reference's max-pool
    selections forced as well (tight pin) and once free-running."""
    from mlsp_amd import functional as Fh
    g = dict(np.load(os.path.join(golden_dir, fname)))
    tag = fname[:8]
    args = gc.make_args(cuda=True)
    inp = {k: v.to(dev) for k, v in gc.make_inputs(seed, B, N).items()}
    forced = [torch.from_numpy(g["knn%d" % i].astype(np.int64)) for i in range(5)]
    mode = Fh.gemm_precision.current if tnet_fwd == "default" else "fp32"
    sel = None
    for pass_ in ("free", "forced-selections"):
        m = _model(seed, dev)
        m.train()
        if pass_ == "forced-selections":
            sel = _oracle_selections(m, inp, forced, dtype=torch.float64)     # the float64 oracle's choices: the same ones its gradients below route through
        with Fh.gemm_precision(mode), Fh.forced_graphs(forced), Fh.recorded_selections() as rec:
            if sel is None:
                logits = m(inp["x"], activate_density_normal_ondef=True)
                own = rec.sel
            else:
                with Fh.forced_selections(sel):
                    logits = m(inp["x"], activate_density_normal_ondef=True)
        assert logits["DefRec"].shape == (B, N, 3) and logits["density"].shape == (B * N, 16)
        assert logits["density_mse"].shape == (B * N,) and logits["cls"].shape == (B, 10)
        for key in HEAD_KEYS:
            np.testing.assert_allclose(logits[key].detach().cpu().numpy(), g["train/" + key], rtol=1e-3, atol=1e-3, err_msg=key)
        loss, parts = _gpu_total_loss(args, logits, inp)
        np.testing.assert_allclose(loss.item(), g["train/loss"], rtol=1e-3)
        for k, v in parts.items():
            np.testing.assert_allclose(v.item(), g["train/loss_" + k], rtol=1e-3, atol=1e-5, err_msg=k)
        loss.backward()
        named = dict(m.named_parameters())
        assert named["Density_cls.fc2.weight"].grad is None                    # frozen (Models.py:270)
        rels = {}
        gkeys = [k[5:] for k in g if k.startswith("grad/") and k != "grad/bn5.bias"]
        for key in gkeys:
            ref = g["grad/" + key]
            got = named[key].grad.cpu().numpy()[:ref.shape[0]]
            rels[key] = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)
        print("%s %s %s: grad rel-L2 vs the reference's fp32 golden: max %.2e (%s)" % (tag, mode, pass_, max(rels.values()), max(rels, key=rels.get)))
        if pass_ == "free":
            tol = GRAD_TOL_FREE[tag]
            assert all(r < tol for r in rels.values()), (pass_, {k: "%.1e" % v for k, v in rels.items() if v >= tol})
        else:
            g64 = _oracle_grads_f64(m, inp, forced, gkeys)
            bad = {}
            for key in gkeys:
                w = g64[key][:g["grad/" + key].shape[0]]
                nrm = np.linalg.norm(w) + 1e-300
                yard = np.linalg.norm(g["grad/" + key].astype(np.float64) - w) / nrm          # the reference's fp32 run vs float64
                mine = np.linalg.norm(named[key].grad.double().cpu().numpy()[:w.shape[0]] - w) / nrm
                if mine >= max(GRAD_FLOOR_FORCED[tag], 3.0 * yard):
                    bad[key] = "%.1e (fp32 golden vs float64: %.1e)" % (mine, yard)
            print("%s %s forced: worst distance from float64 %.2e" % (tag, mode, max(
                np.linalg.norm(named[k_].grad.double().cpu().numpy()[:g64[k_].shape[0]] - g64[k_]) / (np.linalg.norm(g64[k_]) + 1e-300) for k_ in gkeys)))
            assert not bad, bad
        if pass_ == "forced-selections":
            flips = [int((a.cpu().long() != b.long()).sum()) for a, b in zip(own, sel)]
            print("%s %s: selections that differ from the reference's, per max-pool (tnet k, tnet N, conv1-4 k, conv5 N): %s of %s"
                  % (tag, mode, flips, [a.numel() for a in sel]))
            assert sum(flips) <= 2e-3 * sum(a.numel() for a in sel), flips
    st = m.state_dict()
    for key in [k[4:] for k in g if k.startswith("run/")]:
        np.testing.assert_allclose(st[key].cpu().numpy(), g["run/" + key], rtol=1e-3, atol=1e-5, err_msg=key)
    assert int(st["bn5.num_batches_tracked"]) == 1 and int(st["conv1.conv.1.num_batches_tracked"]) == 1

    # eval mode uses the running statistics just updated; embedding path too
    m.eval()
    if fname.startswith("dgcnn_s2"):
        with torch.no_grad(), Fh.gemm_precision(mode):
            le = m(inp["x"], activate_density_normal_ondef=True)
            emb = m(inp["x"], visualization=True)
        for key in HEAD_KEYS:
            np.testing.assert_allclose(le[key].cpu().numpy(), g["eval/" + key], rtol=1e-3, atol=1e-3, err_msg="eval " + key)
        np.testing.assert_allclose(emb.cpu().numpy(), g["eval/embedding"], rtol=1e-3, atol=1e-3)


GRAD_KEYS = ["conv1.conv.0.weight", "conv1.conv.1.weight", "conv4.conv.0.weight", "conv5.weight", "DefRec.conv1.weight",
             "input_transform_net.fc3.weight", "input_transform_net.conv2d2.conv.0.weight", "Density_cls.mlp3.weight",
             "C.mlp1.fc.0.weight", "bn5.bias"]        # tools/make_golden.py GRAD_KEYS


def _check_grads_vs_f64(named, want, yardstick, floor=2e-3):
    """Relative L2 distance of the GPU's fp32 gradients from a DOUBLE-precision evaluation of the reference arithmetic.
    `yardstick[key]` is the same distance for a CPU fp32 run of the reference's operators on the same inputs and graphs, i.e.
    the rounding noise of this step in fp32 (it grows from 1e-4 at the heads to ~5e-3 at conv1 / the T-Net: sparse Chamfer
    gradients through six train-mode BatchNorms).  Bar: `floor`, or 3x that noise where the noise itself is larger.
    Measured on MI355X: heads 2e-5..3e-3, conv5 1e-3..4e-3, conv1 / T-Net 6e-3..9e-3 -- the closed-form
    (Gram-matrix, folded-EdgeConv) backward passes re-associate the sums, so their rounding differs from the CPU's; it grows
    with the batch (longer sums) as rounding does, not as a formula error would."""
    worst, bad = {}, {}
    for key, ref in want.items():
        got = named[key].grad.double().cpu().numpy()[:ref.shape[0]]
        if key == "bn5.bias":     # zero except for the few channels whose per-cloud max is negative: compare on the gradient's own scale
            assert np.abs(got - ref).max() < 5e-3 * max(np.abs(ref).max(), 1e-3), key
            continue
        worst[key] = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-300)
        if worst[key] >= max(floor, 3.0 * float(yardstick[key])):
            bad[key] = (worst[key], float(yardstick[key]))
    print("grad rel-L2 vs float64 (HIP fp32 | CPU fp32):", {k: "%.1e | %.1e" % (v, float(yardstick[k])) for k, v in worst.items()})
    assert not bad, bad


def test_dgcnn_vs_reference_float64_golden(dev, golden_dir):
    """N = 1024 gradient pin.  tests/golden/dgcnn_f64_s3_B4_N1024.npz is the REFERENCE evaluated in float64 (tools/make_golden.py
    round2): the rounding-free value of its arithmetic.  Two fp32 evaluations of this step differ by 1-3e-2 in their gradients
    at this size (tools/grad_conditioning.py), so the fp32 goldens can only bound gradients at 5e-2; against the double-precision
    value the HIP path is held to 2e-3 -- or 3x the reference's own fp32 distance from it (recorded in the fixture as
    ref32_rel/*), whichever is larger -- and logits and losses to the 1e-3 contract."""
    from mlsp_amd import functional as Fh
    seed, B, N = 3, 4, 1024
    g = dict(np.load(os.path.join(golden_dir, "dgcnn_f64_s3_B4_N1024.npz")))
    m = _model(seed, dev)
    for k, v in gc.state_checksums(m).items():
        np.testing.assert_allclose(v, g["chk/" + k], rtol=1e-6, atol=1e-9, err_msg=k)
    args = gc.make_args(cuda=True)
    inp = {k: v.to(dev) for k, v in gc.make_inputs(seed, B, N).items()}
    m.train()
    graphs = [torch.from_numpy(g["knn%d" % i].astype(np.int64)) for i in range(5)]
    # the reference's float64 max-pool selections (the oracle in float64 reproduces that run to 1e-9, tests/test_oracle_golden.py): with the
    # graphs AND the selections forced the comparison is about arithmetic, whichever product mode / kernel computes the forward
    sel = _oracle_selections(m, inp, graphs, dtype=torch.float64)
    want = {k[5:]: g[k] for k in g if k.startswith("grad/")}
    yard = {k[10:]: g[k] for k in g if k.startswith("ref32_rel/")}
    T = {}
    # "fp32": every product on the f32 MFMA (the arithmetic of the reference's own fp32 run, up to summation order) -- the tight pin.
    # default ("bf16x6"): split products, closer to float64 per kernel (tests/test_gpu_kernels.py) but rounding differently: at B = 4 the
    # T-Net's FC tail normalises over FOUR rows (BatchNorm1d, PointDA/model_utils.py:118-121: 1 / sqrt(var + eps) reaches 10^2), so a
    # last-bit difference of the per-edge stage moves the 3 x 3 transform in the 4th digit (printed below) and every gradient behind it in
    # the 3rd; the bench-size test below (B = 32) holds the same mode to 5e-3.
    for mode, floor in (("fp32", 2e-3), (Fh._lib.DEFAULT_GEMM_PRECISION, 2e-2)):
        mm = copy.deepcopy(m)
        mm.input_transform_net.register_forward_hook(lambda mod, args_, out: T.__setitem__(mode, out.detach().double().cpu()))
        with Fh.gemm_precision(mode), Fh.forced_graphs(graphs), Fh.forced_selections(sel):
            logits = mm(inp["x"], activate_density_normal_ondef=True)
        for key in HEAD_KEYS:
            np.testing.assert_allclose(logits[key].detach().cpu().numpy(), g["train/" + key], rtol=1e-3, atol=1e-3, err_msg=key)
        loss, parts = _gpu_total_loss(args, logits, inp)
        np.testing.assert_allclose(loss.item(), float(g["train/loss"]), rtol=1e-3)
        for k, v in parts.items():
            np.testing.assert_allclose(v.item(), float(g["train/loss_" + k]), rtol=1e-3, atol=1e-5, err_msg=k)
        loss.backward()
        print("gemm_precision(%r):" % mode)
        _check_grads_vs_f64(dict(mm.named_parameters()), want, yard, floor=floor)
    if "fp32" in T and Fh._lib.DEFAULT_GEMM_PRECISION in T and Fh._lib.DEFAULT_GEMM_PRECISION != "fp32":
        a, b_ = T["fp32"], T[Fh._lib.DEFAULT_GEMM_PRECISION]
        print("input transform (B x 3 x 3): rel-L2 between the two product modes %.2e" % ((a - b_).norm() / a.norm()).item())


def test_bench_config_vs_oracle_float64(dev):
    """BASELINE.json configs[1] itself (B = 32, N = 1024, k = 20): the whole step on the HIP path against the oracle evaluated in
    float64 with the oracle's (canonical-kNN) graphs forced on both sides -- logits <= 1e-3, losses <= 1e-3 relative, the ten
    fixture gradients <= 5e-3 (or 3x the fp32 noise of the stock-torch CPU run of the same step, oracle/ref_torch_modules.py) --
    and the free-running first graph stage bit-exact at this size."""
    from oracle import ref_torch_modules as rtm
    from mlsp_amd import functional as Fh
    seed, B, N = 5, 32, 1024
    m = _model(seed, dev)
    ref = copy.deepcopy(m).cpu().double()
    params = dict(ref.state_dict(keep_vars=True))
    inp = gc.make_inputs(seed, B, N)
    inp64 = {k: (v.double() if v.is_floating_point() else v) for k, v in inp.items()}
    want, _, ctx = ref_cpu.dgcnn_forward(params, inp64["x"], training=True, dropout_p=0.0,
                                         knn_fn=lambda x_, k_: knn_canon.knn(x_.float(), k_),
                                         activate_density_normal_ondef=True, return_ctx=True)
    wloss, wparts = gc.total_loss(gc.make_args(), ref_cpu, want, inp64)
    wloss.backward()
    wgrads = {k: dict(ref.named_parameters())[k].grad.numpy() for k in GRAD_KEYS}
    want = {k: v.detach().numpy() for k, v in want.items()}
    wparts = {k: v.item() for k, v in wparts.items()}
    graphs = [i.clone() for i in ctx.knn_idx]
    sel64 = [a.clone() for a in ctx.sel]                  # the float64 run's max-pool selections (forced on the HIP run below)
    del ctx, params, ref, inp64
    it = iter(graphs)
    stock = rtm.StockDGCNN(gc.make_args(dropout=0.0), knn_fn=lambda x_, k_: next(it)).train()
    stock.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()}, strict=True)
    sl, _ = gc.total_loss(gc.make_args(), ref_cpu, stock(inp["x"], activate_density_normal_ondef=True), inp)
    sl.backward()
    noise = {k: np.linalg.norm(dict(stock.named_parameters())[k].grad.double().numpy() - wgrads[k]) / np.linalg.norm(wgrads[k])
             for k in GRAD_KEYS}
    del stock, sl
    ginp = {k: v.to(dev) for k, v in inp.items()}
    xp = ginp["x"].transpose(2, 1).contiguous().view(B * N, 3)
    assert torch.equal(Fh.knn_graph(xp, B, N, 20).idx.view(B, N, 20).cpu().long(), graphs[0]), "raw-cloud kNN must be bit-exact"
    m.train()
    with Fh.forced_graphs(graphs), Fh.forced_selections(sel64):
        logits = m(ginp["x"], activate_density_normal_ondef=True)
    for key in HEAD_KEYS:
        np.testing.assert_allclose(logits[key].detach().cpu().numpy(), want[key], rtol=1e-3, atol=1e-3, err_msg=key)
    loss, parts = _gpu_total_loss(gc.make_args(cuda=True), logits, ginp)
    np.testing.assert_allclose(loss.item(), wloss.item(), rtol=1e-3)
    for k, v in parts.items():
        np.testing.assert_allclose(v.item(), wparts[k], rtol=1e-3, atol=1e-5, err_msg=k)
    loss.backward()
    _check_grads_vs_f64(dict(m.named_parameters()), wgrads, noise, floor=5e-3)      # B = 32: 8x longer sums than the B = 4 fixture


def _run_recording_graphs(monkeypatch, Fh, fn):
    """fn() with every neighbour graph it builds recorded: ([B, N, k] int64 index tensors on the host, in call order) for Fh.forced_graphs."""
    rec, real = [], Fh.knn_graph

    def recording(xp, B_, N_, k_, need_reverse=True):
        g = real(xp, B_, N_, k_, need_reverse=need_reverse)
        rec.append(g.idx.view(B_, N_, k_).long().cpu())
        return g
    monkeypatch.setattr(Fh, "knn_graph", recording)
    try:
        out = fn()
    finally:
        monkeypatch.setattr(Fh, "knn_graph", real)
    return out, rec


def test_gemm_product_modes_agree_on_bench_step(dev, monkeypatch):
    """BASELINE.json configs[1] size, the same step under gemm_precision "bf16x6" (fp32-accurate products on the bf16 cores), "f16x3"
    (the same on the f16 cores with two pieces) and "fp32" (f32 MFMA) with the first mode's neighbour graphs forced on the second: both are fp32 computations of the same network,
    so the logits agree to 1e-4 relative L2 (measured 2.5e-5 on the deepest output, cls: two different fp32 rounding sequences through
    ten BatchNorm'd layers; the parity contract against the oracle is 1e-3), the losses to 1e-4."""
    from mlsp_amd import functional as Fh
    seed, B, N = 6, 32, 1024
    m = _model(seed, dev)
    ginp = {k: v.to(dev) for k, v in gc.make_inputs(seed, B, N).items()}
    m.train()
    out = {}
    graphs = None
    for mode in ("bf16x6", "fp32", "f16x3"):
        mm = copy.deepcopy(m)
        with Fh.gemm_precision(mode):
            if graphs is None:
                logits, graphs = _run_recording_graphs(monkeypatch, Fh, lambda: mm(ginp["x"], activate_density_normal_ondef=True))
            else:
                with Fh.forced_graphs(graphs):
                    logits = mm(ginp["x"], activate_density_normal_ondef=True)
            loss, parts = _gpu_total_loss(gc.make_args(cuda=True), logits, ginp)
        out[mode] = ({k: logits[k].detach().double() for k in HEAD_KEYS}, loss.item(), {k: v.item() for k, v in parts.items()})
    for other in ("bf16x6", "f16x3"):
        for k in HEAD_KEYS:
            a, b = out[other][0][k], out["fp32"][0][k]
            rel = ((a - b).norm() / b.norm()).item()
            print("%s: rel-L2 between the product modes %s and fp32 %.2e" % (k, other, rel))
            assert rel < 1e-4, (other, k, rel)
            assert (a - b).abs().max().item() < 1e-3 * max(1.0, b.abs().max().item()), (other, k)
        np.testing.assert_allclose(out[other][1], out["fp32"][1], rtol=1e-4)
        for k, v in out[other][2].items():
            np.testing.assert_allclose(v, out["fp32"][2][k], rtol=1e-4, atol=1e-6, err_msg=other + " " + k)


# Free-running parity (nothing forced): the dynamic graph is discontinuous -- a last-bit difference in a feature flips a near-tied
# neighbour, moves that row by O(1), and the next stage's graph inherits it (one flipped row at stage 2 became 6 and 19 at stages 3 and 4
# of the B = 4, N = 256 case) -- so "how close is free-running" needs a YARDSTICK: the reference's own operators in fp32 on the CPU
# (oracle/ref_torch_modules.py::StockDGCNN with the canonical fp32 kNN) measured against the SAME network in float64 (matmul + topk in
# double: the truth).  Measured on MI355X (round 5; rows whose neighbour SET differs from the truth per stage, HIP | stock fp32 CPU):
#     B = 4,  N = 128 (512 rows):      0 0 0 0 0            | 0 0 0 0 0
#     B = 4,  N = 256 (1,024 rows):    0 0 1 6 19           | 0 0 0 0 0
#     B = 32, N = 1024 (32,768 rows):  1 363 1133 5062 12514 | 1 363 1128 5063 12504
# At the bench size ANY fp32 evaluation parts from the float64 graphs (the fp32 distance formula -xx - inner - xx^T cancels badly on
# 3-D coordinates: 363 rows at the transformed-cloud stage) and the HIP path does so exactly as often as the reference's operators do.
# Bars: at the FIRST stage where the HIP graph differs, at most max(FREE_RUN_MIN_ROWS, FREE_RUN_FLIP_FACTOR x the yardstick's count)
# rows (independent events); downstream (inherited flips) at most max(FACTOR x yardstick, FREE_RUN_CASCADE_CAP of the rows); outputs:
# 1e-3 element-wise when nothing flipped, else rel-L2 from the truth <= max(1e-3, FREE_RUN_DIST_FACTOR x the yardstick's + sqrt(the
# fraction of rows flipped BEYOND the yardstick's count)): where the flip counts equal the yardstick's, so must the distance (the
# measured B = 32 pairs are in DESIGN.md section 12).
FREE_RUN_FLIP_FACTOR = 3
FREE_RUN_DIST_FACTOR = 1.5
FREE_RUN_MIN_ROWS = 2
FREE_RUN_CASCADE_CAP = 0.03


def _stock_free_run(m_hip, x, dtype, knn_fn):
    """the reference's operator sequence on the CPU in `dtype`, free-running: -> (logits, the five [B, N, k] index tensors it built)"""
    from oracle import ref_torch_modules as rtm
    rec = []

    def spy(x_, k_):
        idx = knn_fn(x_, k_)
        rec.append(idx.clone().long())
        return idx
    ref = rtm.StockDGCNN(gc.make_args(dropout=0.0), knn_fn=spy)
    ref.load_state_dict({k: v.detach().cpu() for k, v in m_hip.state_dict().items()}, strict=True)
    ref = ref.to(dtype).train()
    with torch.no_grad():
        out = ref(x.cpu().to(dtype), activate_density_normal_ondef=True)
    return out, rec


@pytest.mark.parametrize("seed,B,N", [(2, 4, 128), (3, 4, 256), (5, 32, 1024)])
def test_dgcnn_free_running_vs_oracle(dev, seed, B, N):
    """No forcing, up to BASELINE.json configs[1]'s own size (B = 32, N = 1024): the raw-cloud graph bit-exact with the canonical kNN;
    at the four feature-space stages the neighbour sets that differ from the float64 truth are counted and held to the yardstick above;
    the five outputs are held to 1e-3 element-wise when no set flipped and to the yardstick's relative-L2 distance from the truth
    otherwise (a flipped neighbour changes its row by O(1): DESIGN.md section 2)."""
    from mlsp_amd import functional as Fh
    from oracle import ref_torch_modules as rtm
    m = _model(seed, dev)
    inp = gc.make_inputs(seed, B, N)
    truth, idx64 = _stock_free_run(m, inp["x"], torch.float64, rtm.knn_matmul_topk)
    yard, idx32 = _stock_free_run(m, inp["x"], torch.float32, knn_canon.knn)
    seen = []
    orig = Fh.knn_graph

    def spy(xp, B_, N_, k_, need_reverse=True):
        gr = orig(xp, B_, N_, k_, need_reverse)
        seen.append(gr.idx.view(B_, N_, k_).cpu().long())
        return gr
    Fh.knn_graph = spy
    try:
        m.train()
        with torch.no_grad():
            got = m(inp["x"].to(dev), activate_density_normal_ondef=True)
    finally:
        Fh.knn_graph = orig
    assert torch.equal(seen[0], idx32[0]), "raw-cloud kNN must be bit-exact with the canonical arithmetic"

    def set_flips(a, b):
        return [int((x.sort(-1)[0] != y.sort(-1)[0]).any(-1).sum()) for x, y in zip(a, b)]
    hip_f, yard_f = set_flips(seen, idx64), set_flips(idx32, idx64)
    print("B=%d N=%d free-running, rows (of %d) whose neighbour SET differs from the float64 truth per stage: HIP %s | stock fp32 CPU %s"
          % (B, N, B * N, hip_f, yard_f))
    first = next((s for s, h in enumerate(hip_f) if h), None)
    for s, (h, y) in enumerate(zip(hip_f, yard_f)):
        bar = max(FREE_RUN_MIN_ROWS, FREE_RUN_FLIP_FACTOR * y) if s == first else max(FREE_RUN_FLIP_FACTOR * y, FREE_RUN_CASCADE_CAP * B * N)
        assert h <= bar, (s, hip_f, yard_f)
    excess = max(max(0, h - y) for h, y in zip(hip_f, yard_f)) / float(B * N)
    for key in HEAD_KEYS:
        t = truth[key].double()
        rel_h = ((got[key].cpu().double() - t).norm() / t.norm()).item()
        rel_y = ((yard[key].double() - t).norm() / t.norm()).item()
        print("   %s: rel-L2 from the float64 truth: HIP %.2e | stock fp32 CPU %.2e" % (key, rel_h, rel_y))
        if sum(hip_f) == 0:
            np.testing.assert_allclose(got[key].cpu().numpy(), t.float().numpy(), rtol=1e-3, atol=1e-3, err_msg=key)
        else:
            # rows the HIP path flipped BEYOND the yardstick's count (independent events inside the allowance above, and their cascade)
            # each move their output row by O(1): sqrt(excess fraction) on top of the yardstick's own distance
            assert rel_h <= max(1e-3, FREE_RUN_DIST_FACTOR * rel_y + excess ** 0.5), (key, rel_h, rel_y, excess)


def test_module_api_surface(dev):
    """state_dict keys/shapes, deepcopy, flags -> dict keys, eval determinism, DataParallel-free .to()."""
    from mlsp_amd import Models
    m = _model(0, dev, dropout=0.5)
    sd = m.state_dict()
    assert len(sd) == 159
    assert sd["input_transform_net.conv2d1.conv.0.weight"].shape == (64, 6, 1, 1)
    assert sd["Density_cls.fc2.weight"].shape == (1, 16)
    assert torch.equal(sd["Density_cls.fc2.weight"].cpu()[0], torch.arange(16, dtype=torch.float32) * 2)
    m2 = copy.deepcopy(m)                                   # utils/log.py:33, trainer.py:253
    m2.load_state_dict(sd, strict=True)                     # train_spst.py:141
    x = (torch.rand(4, 3, 128) * 2 - 1).to(dev)
    m.eval()
    with torch.no_grad():
        a = m(x)
        assert set(a.keys()) == {"cls"}
        assert set(m(x, activate_DefRec=True).keys()) == {"cls", "DefRec"}
        assert set(m(x, activate_normal=True).keys()) == {"cls", "Normal"}
        assert set(m(x, activate_scan=True).keys()) == {"cls", "Rec_scan"}
        assert set(m(x, activate_density=True).keys()) == {"cls", "density", "density_mse"}
        full = m(x, activate_density_normal_ondef=True)
        assert set(full.keys()) == {"cls", "DefRec", "density", "density_mse", "Normal"}
        emb = m(x, visualization=True)
        assert emb.shape == (4, 1024)
        b = m2.eval()(x)
    assert torch.equal(a["cls"], b["cls"])
    # train mode with dropout on: runs, differs between calls, gradients accumulate over two backwards
    m.train()
    l1 = m(x, activate_density_normal_ondef=True)
    l2 = m(x, activate_density_normal_ondef=True)
    assert not torch.equal(l1["DefRec"], l2["DefRec"])
    l1["DefRec"].sum().backward()
    g1 = m.conv1.conv[0].weight.grad.clone()
    l2["DefRec"].sum().backward()
    assert not torch.equal(g1, m.conv1.conv[0].weight.grad)
    # standalone heads keep the reference's [B,C,N] signature
    head_in = torch.rand(2, 1536, 64, device=dev)
    assert m.DefRec(head_in).shape == (2, 64, 3)
    pv, dn = m.Density_cls(head_in)
    assert pv.shape == (128, 16) and dn.shape == (128,)


@pytest.mark.parametrize("heads", [("DefRec",), ("DefRec", "density"), ("DefRec", "Normal"), ("Normal", "density")])
def test_loss_on_a_subset_of_the_heads_fused_vs_unfused(dev, heads, monkeypatch):
    """The reference trainer's default step: all three heads run (activate_density_normal_ondef=True), the Normal / density losses are
    gated by args.Normal_ondef / args.Density_ondef (PointDA/trainer.py:551-565) -- the backward reaches a SUBSET of the heads.  With
    deferred activations + fused BatchNorm-backward statistics (the defaults) the consumers that run store masked gradients; the
    producers must not mask them again (functional.BwdStats.take, the pre_parts = -1 path of include/mlsp_hip.h).  Dropout 0.5, same
    dropout streams: every gradient against the materialised, unfused path."""
    import itertools as it
    from mlsp_amd import functional as Fh, mlsp
    args = gc.make_args(cuda=True)
    inp = {k: v.to(dev) for k, v in gc.make_inputs(7, 8, 512).items()}    # (4,096 rows: the GEMM dgrads of the merged layers fuse the sums too)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(Fh, "_FUSE_BWD_STATS", fused)
        monkeypatch.setattr(Fh, "_DEFER_CHAINS", fused)
        monkeypatch.setattr(Fh, "_seed_counter", it.count(777), raising=False)
        torch.manual_seed(11)
        m = _model(7, dev, dropout=0.5).train()
        torch.manual_seed(11)               # (the dropout streams are keyed by torch.initial_seed())
        logits = m(inp["x"], activate_density_normal_ondef=True)     # (the encoder runs the same kernels in both passes: same graphs)
        loss = 0.0
        mask_cord = inp["mask"].permute(0, 2, 1)[:, :, 0] * 26 + 1
        if "DefRec" in heads:
            loss = loss + mlsp.calc_loss(args, logits, inp["gold"], inp["mask"])
        if "Normal" in heads:
            loss = loss + mlsp.calc_masked_normal_loss(args, logits["Normal"], inp["normal_gt"], mask_cord)
        if "density" in heads:
            kl, mae = mlsp.densityloss(args, logits, inp["dens_val"], inp["dens_vec"], mask=mask_cord.reshape(-1))
            loss = loss + kl + mae
        loss.backward()
        res[fused] = ({k: v.detach().clone() for k, v in logits.items()}, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    (la, ga), (lb, gb_) = res[True], res[False]
    for k in la:
        np.testing.assert_allclose(la[k].cpu().numpy(), lb[k].cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    assert set(ga) == set(gb_)
    worst = {}
    for k in ga:
        rel = ((ga[k] - gb_[k]).double().norm() / (gb_[k].double().norm() + 1e-30)).item()
        worst[k.split(".")[0]] = max(worst.get(k.split(".")[0], 0.0), rel)
        # heads: the same masked gradients, column sums in another order; upstream of x_cat the gradients are ill-conditioned (DESIGN.md
        # section 2) -- a doubled mask (the defect this test pins) shows as O(1) on the head layers themselves
        direct = k.split(".")[0] in ("DefRec", "Norm_pred", "Density_cls")
        assert rel < (2e-4 if direct else 5e-2), (k, rel, heads)
    print("subset %s: worst rel-L2 per module %s" % (heads, {k: "%.1e" % v for k, v in worst.items()}))


def test_merged_head_first_layers_equal_separate(dev):
    """Models.merged_first_layers (the three heads' conv1 + bn1 as ONE wide layer, parameters re-homed back to back) against the
    one-layer-per-head path: same logits, losses, gradients and running statistics; parameters keep identity / state_dict /
    deepcopy / optimizer behaviour; a subset of heads and the eval mode take the merged path as well."""
    from mlsp_amd import Models, functional as Fh
    args = gc.make_args(cuda=True)
    inp = {k: v.to(dev) for k, v in gc.make_inputs(4, 4, 256).items()}
    res = {}
    for merged in (False, True, "tails"):
        Models.MERGE_HEAD_FIRST_LAYERS = bool(merged)
        Models.MERGE_HEAD_TAILS = merged == "tails"
        try:
            m = _model(4, dev).train()
            ids = {n: id(p) for n, p in m.named_parameters()}
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5, fused=True)
            logits = m(inp["x"], activate_density_normal_ondef=True)
            loss, _ = _gpu_total_loss(args, logits, inp)
            loss.backward()
            grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
            state = {k: v.clone() for k, v in m.state_dict().items()}                     # running statistics of this forward
            with torch.no_grad():
                two = m(inp["x"], activate_DefRec=True, activate_normal=True)             # a two-head subset
                m.eval()
                ev = m(inp["x"], activate_density_normal_ondef=True)
            opt.step()                                                                    # updates in place, through the re-homed views
            assert all(torch.isfinite(p).all().item() for p in m.parameters())
            assert ids == {n: id(p) for n, p in m.named_parameters()}
            adjacent = Fh._adjacent([m.DefRec.conv1.weight, m.Norm_pred.conv1.weight, m.Density_cls.conv1.weight])   # region heads first
            assert adjacent == bool(merged)
            m2 = copy.deepcopy(m)
            m2.load_state_dict(m.state_dict(), strict=True)
            with torch.no_grad():
                ev1, ev2 = m(inp["x"], activate_density_normal_ondef=True), m2(inp["x"], activate_density_normal_ondef=True)
            assert all(torch.equal(ev1[k], ev2[k]) for k in ev1)
            res[merged] = (logits, loss.item(), grads, state, two, ev)
        finally:
            Models.MERGE_HEAD_FIRST_LAYERS = Models.MERGE_HEAD_TAILS = True
    for other in (True, "tails"):
        _compare_head_paths(res[False], res[other])


def _compare_head_paths(ra, rb):
    (la, lossa, ga, sa, twoa, eva), (lb, lossb, gb_, sb, twob, evb) = ra, rb
    for k in HEAD_KEYS:
        np.testing.assert_allclose(lb[k].detach().cpu().numpy(), la[k].detach().cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
        np.testing.assert_allclose(evb[k].cpu().numpy(), eva[k].cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg="eval " + k)
    for k in twoa:
        np.testing.assert_allclose(twob[k].cpu().numpy(), twoa[k].cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg="two " + k)
    assert abs(lossa - lossb) <= 1e-5 * abs(lossa)
    assert set(ga) == set(gb_)
    for k in ga:
        rel = ((ga[k] - gb_[k]).norm() / (ga[k].norm() + 1e-30)).item()
        # the heads / conv5 / classifier gradients do not depend on the summation order of the x_cat gradient; everything upstream of
        # x_cat sees it re-associated (one K = 1024 dgrad instead of three accumulating ones), and those gradients are ill-conditioned
        # (two fp32 evaluations of this step differ by 0.3-2 % there; at this size -- B = 4, N = 256 -- the T-Net's
        # BatchNorm biases move by up to 0.10 when only the summation order of one per-cloud dgrad changes: round 6, the x5 halves'
        # input gradients summed by skinny_bwd_pair_kernel instead of the tiled kernel)
        direct = k.split(".")[0] in ("DefRec", "Norm_pred", "Density_cls", "C", "conv5", "bn5")
        assert rel < (1e-4 if direct else 0.15), (k, rel)
    for k in sa:                                                          # running statistics (and the untouched parameters)
        np.testing.assert_allclose(sb[k].float().cpu().numpy(), sa[k].float().cpu().numpy(), rtol=1e-3, atol=1e-5, err_msg=k)


def test_edge_grad_into_slice_equals_autograd(dev):
    """Models.EDGE_GRAD_INTO_SLICE (off by default: measured slower): layer l+1 adds its input gradient into layer l's column slice of
    the x_cat gradient through the dgrad GEMM's beta = 1 epilogue and dx row pitch (mlsp_edgeconv_bwd_f32 lddx / dx_accumulate)
    instead of autograd's accumulation: same gradients."""
    from mlsp_amd import Models
    args = gc.make_args(cuda=True)
    inp = {k: v.to(dev) for k, v in gc.make_inputs(6, 3, 256).items()}
    grads = {}
    for flag in (False, True):
        Models.EDGE_GRAD_INTO_SLICE = flag
        try:
            m = _model(6, dev).train()
            loss, _ = _gpu_total_loss(args, m(inp["x"], activate_density_normal_ondef=True), inp)
            loss.backward()
            grads[flag] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            Models.EDGE_GRAD_INTO_SLICE = False
    assert set(grads[False]) == set(grads[True])
    for k, a in grads[False].items():
        rel = ((a - grads[True][k]).norm() / (a.norm() + 1e-30)).item()
        direct = k.split(".")[0] in ("DefRec", "Norm_pred", "Density_cls", "C", "conv5", "bn5", "conv4")
        assert rel < (1e-5 if direct else 0.1), (k, rel)          # upstream of the re-associated sums: ill-conditioned


def test_dataparallel_and_replica_threads(dev):
    """PointDA/trainer.py:251-253 wraps the model in nn.DataParallel: replicas run forward in one Python thread each.  (1) the
    wrapper works end to end on this box's GPU; (2) two model copies driven concurrently by torch's own parallel_apply (two
    threads, as DataParallel does) give the results of the serial runs and lose no num_batches_tracked increment."""
    from torch.nn.parallel import parallel_apply
    m = _model(7, dev).train()
    x = (torch.rand(4, 3, 256, generator=torch.Generator().manual_seed(7)) * 2 - 1).to(dev)
    dp = torch.nn.DataParallel(m, [0])
    out = dp(x, activate_density_normal_ondef=True)
    sum(v.float().sum() for v in out.values()).backward()
    assert int(m.bn5.num_batches_tracked) == 1 and m.conv1.conv[0].weight.grad is not None
    a, b = copy.deepcopy(m), copy.deepcopy(m)
    xa, xb = x[:2].contiguous(), x[2:].contiguous()
    with torch.no_grad():
        want_a = copy.deepcopy(m)(xa, activate_density_normal_ondef=True)
        want_b = copy.deepcopy(m)(xb, activate_density_normal_ondef=True)
    for rep in range(5):
        with torch.no_grad():
            got = parallel_apply([a, b], [(xa,), (xb,)], [dict(activate_density_normal_ondef=True)] * 2, [0, 0])
        if rep == 0:
            for k in HEAD_KEYS:
                assert torch.equal(got[0][k], want_a[k]) and torch.equal(got[1][k], want_b[k]), k
    for r in (a, b):
        st = r.state_dict()
        assert all(int(v) == 1 + 5 for k, v in st.items() if k.endswith("num_batches_tracked") and not k.startswith("Rec_scan")), \
            {k: int(v) for k, v in st.items() if k.endswith("num_batches_tracked")}


def test_full_size_step_properties(dev):
    """BASELINE config (B=32, N=1024, k=20): one fwd+bwd; size-independent properties."""
    m = _model(5, dev, dropout=0.5)
    args = gc.make_args(cuda=True)
    inp = {k: v.to(dev) for k, v in gc.make_inputs(5, 32, 1024).items()}
    m.train()
    logits = m(inp["x"], activate_density_normal_ondef=True)
    loss, parts = _gpu_total_loss(args, logits, inp)
    loss.backward()
    assert torch.isfinite(loss).item()
    for k, v in logits.items():
        assert torch.isfinite(v).all().item(), k
    p = logits["density"]
    np.testing.assert_allclose(p.sum(1).detach().cpu().numpy(), 1.0, rtol=1e-4)          # softmax rows
    np.testing.assert_allclose(logits["density_mse"].detach().cpu().numpy(),
                               (p.detach().cpu() * torch.arange(16) * 2.0).sum(1).numpy(), rtol=1e-4, atol=1e-4)
    for name, q in m.named_parameters():
        if q.requires_grad and not name.startswith("Rec_scan"):
            assert q.grad is not None and torch.isfinite(q.grad).all().item(), name
    # permutation equivariance of the encoder+heads within each cloud (kNN graph is permutation-covariant):
    m.eval()
    with torch.no_grad():
        perm = torch.randperm(1024, device=dev)
        a = m(inp["x"][:2], activate_DefRec=True)["DefRec"]
        b = m(inp["x"][:2][:, :, perm], activate_DefRec=True)["DefRec"]
    assert (a[:, perm] - b).abs().max().item() < 5e-3


# ----------------------------------------------------------------------------- PointSegDA variant
SEG_KEYS = ("seg", "DefRec", "Normal", "density", "density_mse")


def _seg_model(seed, dev, dropout=0.0, K=None):
    from mlsp_amd import seg_models
    torch.manual_seed(seed)
    m = seg_models.DGCNN_DefRec(gc.make_seg_args(dropout=dropout, gpu=True), in_size=3, num_classes=8)
    gc.perturb_params(m, seed)
    if K is not None:                     # PointSegDA/Models.py:6 is a module constant read by the constructors (:150,:202)
        m.k = m.shared_layers.k = K
    return m.to(dev)


@pytest.mark.parametrize("fname,seed,B,N,K", [("segda_s0_B4_N256.npz", 0, 4, 256, None), ("segda_s1_B3_N512.npz", 1, 3, 512, None),
                                              ("segda_k40_s2_B2_N512.npz", 2, 2, 512, 40)])
def test_segda_vs_reference_golden(dev, golden_dir, fname, seed, B, N, K):
    """PointSegDA DGCNN_DefRec on the HIP path vs the reference's outputs / grads / running stats, with the
    reference's neighbour indices forced at the four graph stages."""
    from mlsp_amd import functional as Fh
    g = dict(np.load(os.path.join(golden_dir, fname)))
    m = _seg_model(seed, dev, K=K)
    assert len(m.state_dict()) == 109 and g["knn0"].shape[-1] == (K or 20)
    x = torch.from_numpy(g["x"]).to(dev)
    forced = [torch.from_numpy(g["knn%d" % i].astype(np.int64)) for i in range(4)]
    m.train()
    with Fh.forced_graphs(forced):
        logits = m(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
    assert logits["seg"].shape == (B, N, 8)
    loss = 0.0
    for key in SEG_KEYS:
        np.testing.assert_allclose(logits[key].detach().cpu().numpy(), g["train/" + key], rtol=1e-3, atol=1e-3, err_msg=key)
        loss = loss + (logits[key] * torch.from_numpy(g["w/" + key]).to(dev)).sum() / logits[key].numel()
    np.testing.assert_allclose(loss.item(), g["train/loss"], rtol=1e-3, atol=1e-6)
    loss.backward()
    named = dict(m.named_parameters())
    for key in [k[5:] for k in g if k.startswith("grad/")]:
        if key.startswith("shared_layers") and key.endswith(".bias"):
            continue    # analytically ~0 (see tests/test_oracle_golden.py)
        ref = g["grad/" + key]
        got = named[key].grad.cpu().numpy()[:ref.shape[0]]
        rel = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)
        assert rel < 5e-2, (key, rel)
    st = m.state_dict()
    for key in [k[4:] for k in g if k.startswith("run/")]:
        np.testing.assert_allclose(st[key].cpu().numpy(), g["run/" + key], rtol=1e-3, atol=1e-5, err_msg=key)
    # Tight gradient pin (the method of the DGCNN fixtures): the reference's graphs AND the float64 oracle's max-pool selections forced,
    # every gradient against the FLOAT64 oracle, bar = max(5e-3, 3 x the distance of the oracle's own fp32 run from float64)
    from oracle import ref_seg_cpu
    wts = {key: torch.from_numpy(g["w/" + key]) for key in SEG_KEYS}

    def oracle_run(dtype):
        ref = copy.deepcopy(_seg_model(seed, torch.device("cpu"), K=K)).to(dtype)
        params = dict(ref.state_dict(keep_vars=True))
        it = iter(forced)
        out, _, ctx = ref_seg_cpu.dgcnn_defrec_forward(params, x.cpu().to(dtype), training=True, k=(K or 20), knn_fn=lambda x_, k_: next(it),
                                                       make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True, return_ctx=True)
        sum((out[key] * wts[key].to(dtype)).sum() / out[key].numel() for key in SEG_KEYS).backward()
        return [a.clone() for a in ctx.sel], {n: q.grad.double().numpy() for n, q in ref.named_parameters() if q.grad is not None}
    sel64, g64 = oracle_run(torch.float64)
    _, g32 = oracle_run(torch.float32)
    m2 = _seg_model(seed, dev, K=K)
    m2.train()
    with Fh.forced_graphs(forced), Fh.forced_selections(sel64):
        lg = m2(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
    sum((lg[key] * wts[key].to(dev)).sum() / lg[key].numel() for key in SEG_KEYS).backward()
    worst = (None, 0.0, 0.0)
    for n, q in m2.named_parameters():
        if q.grad is None or n not in g64 or (n.startswith("shared_layers") and n.endswith(".bias")):
            continue
        den = np.linalg.norm(g64[n]) + 1e-30
        if den < 1e-9 * (1.0 + np.linalg.norm(q.detach().cpu().numpy())):      # a bias in front of a batch-statistics BatchNorm: analytically zero
            assert np.linalg.norm(q.grad.cpu().numpy()) < 1e-5, n
            continue
        rel = np.linalg.norm(q.grad.cpu().double().numpy() - g64[n]) / den
        yard = np.linalg.norm(g32[n] - g64[n]) / den
        if rel > worst[1]:
            worst = (n, rel, yard)
        assert rel <= max(5e-3, 3 * yard), (n, rel, yard)
    print("%s: worst gradient vs float64 with forced selections: %s rel %.2e (fp32 oracle itself %.2e)" % ((fname,) + worst))
    m.eval()
    with torch.no_grad(), Fh.forced_graphs(forced):
        le = m(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
    # eval uses the running stats after one update; the reference evaluated with its own (free) graphs, so compare shapes only
    for key in SEG_KEYS:
        assert le[key].shape == g["eval/" + key].shape and torch.isfinite(le[key]).all().item()


def test_segda_config4_shape_vs_oracle(dev):
    """BASELINE.json configs[4] geometry: N = 2048, k = 40.  fp32 arithmetic, free-running graphs (the k > 24 two-pass kNN) against
    the oracle with the canonical kNN: first-stage indices bit-exact, outputs inside the 1e-3 contract except for the rows a
    near-tied neighbour flip moves; then the same step with the bf16 activation storage of configs[4] stays close to fp32."""
    from oracle import ref_seg_cpu
    from mlsp_amd import functional as Fh
    B, N, K = 16, 2048, 40           # 16 clouds: the reference trainer's batch (PointSegDA/trainer.py:103); M = 32768 rows
    m = _seg_model(8, dev, K=K)
    x = torch.rand(B, 3, N, generator=torch.Generator().manual_seed(8)) * 2 - 1
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want, _, ctx = ref_seg_cpu.dgcnn_defrec_forward(params, x, training=True, k=K, knn_fn=knn_canon.knn,
                                                        activate_density_normal_ondef=True, return_ctx=True)
        xp = x.to(dev).transpose(2, 1).contiguous().view(B * N, 3)
        assert torch.equal(Fh.knn_graph(xp, B, N, K).idx.view(B, N, K).cpu().long(), ctx.knn_idx[0])
        m.train()
        got = m(x.to(dev), activate_density_normal_ondef=True)
        bad = 0
        for key in SEG_KEYS:
            d = (got[key].cpu() - want[key]).abs()
            bad += int((d > 1e-3 + 1e-3 * want[key].abs()).sum())
        assert bad <= 0.02 * B * N * 8, bad
        # with the oracle's graphs forced everything matches
        with Fh.forced_graphs([i.clone() for i in ctx.knn_idx]):
            got = m(x.to(dev), activate_density_normal_ondef=True)
        for key in SEG_KEYS:
            np.testing.assert_allclose(got[key].cpu().numpy(), want[key].numpy(), rtol=1e-3, atol=1e-3, err_msg=key)
    # configs[4] arithmetic: bf16 activation storage in the head stacks (+ bf16 GEMM operands elsewhere), same graphs -> close to fp32
    rel = lambda a, b: ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()

    def step(storage, precision):
        m.zero_grad(set_to_none=True)
        with Fh.forced_graphs([i.clone() for i in ctx.knn_idx]), Fh.activation_storage(storage), Fh.gemm_precision(precision):
            out = m(x.to(dev), activate_density_normal_ondef=True)
            sum((out[k].float() * (1.0 + 0.1 * i)).mean() for i, k in enumerate(SEG_KEYS)).backward()
        return {k: v.detach() for k, v in out.items()}, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    o32, g32 = step("fp32", "fp32")
    o16, g16 = step("bf16", "bf16")
    for key in SEG_KEYS:
        assert torch.isfinite(o16[key]).all().item() and rel(o16[key], o32[key]) < 5e-2, (key, rel(o16[key], o32[key]))
    # gradients: finite here; their distance from the truth is pinned against the ORACLE with a bf16-rounding yardstick in
    # test_segda_config4_bf16_step_vs_oracle below (round 6: replaces the self-comparison with the HIP fp32 step that stood here)
    for n in g32:
        assert torch.isfinite(g16[n]).all().item(), n
    assert Fh.activation_storage.current == "fp32" and Fh.gemm_precision.current == Fh._lib.DEFAULT_GEMM_PRECISION


def test_segda_config4_bf16_step_vs_oracle(dev):
    """BASELINE.json configs[4]'s arithmetic (bf16 GEMM operands + bf16 activation storage, N = 2048, k = 40) against the ORACLE, not
    against the HIP fp32 step: truth = oracle/ref_seg_cpu.py in float64; yardstick = the same float64 oracle with every contraction
    reading bf16-ROUNDED operands (round_bf16_ste: the rounding the bf16 matrix cores / the bf16-stored activations see, straight-through
    gradient) -- how far this network moves under bf16 operands when nothing else differs.  Graphs (canonical kNN of the truth) and
    every max-pool selection (the truth's) are forced on all three, so they differ in arithmetic only.  Bar per output and per parameter
    gradient: distance of the HIP bf16 step from the truth <= max(5e-3, 3 x the yardstick's).  (B = 4 clouds: the float64 autograd of
    the per-edge convolutions at N = 2048, k = 40 is what bounds the size.)"""
    from oracle import ref_seg_cpu
    from mlsp_amd import functional as Fh
    B, N, K = 4, 2048, 40
    seed = 9
    x = torch.rand(B, 3, N, generator=torch.Generator().manual_seed(seed)) * 2 - 1
    w = {k: 1.0 + 0.1 * i for i, k in enumerate(SEG_KEYS)}

    def oracle_run(graphs, sel, quant):
        ref = copy.deepcopy(_seg_model(seed, torch.device("cpu"), K=K)).to(torch.float64)
        params = dict(ref.state_dict(keep_vars=True))
        it = iter(graphs) if graphs is not None else None
        knn_fn = (lambda x_, k_: next(it)) if it is not None else (lambda x_, k_: knn_canon.knn(x_.float(), k_))
        out, _, ctx = ref_seg_cpu.dgcnn_defrec_forward(params, x.double(), training=True, k=K, knn_fn=knn_fn, make_seg=True, activate_DefRec=True,
                                                       activate_density_normal_ondef=True, return_ctx=True, force_sel=sel, quant=quant)
        sum((out[k] * w[k]).mean() for k in SEG_KEYS).backward()
        return ({k: v.detach() for k, v in out.items()}, {n: q.grad.clone() for n, q in ref.named_parameters() if q.grad is not None},
                [i.clone() for i in ctx.knn_idx], [a.clone() for a in ctx.sel])
    o64, g64, graphs, sel = oracle_run(None, None, None)
    oy, gy, _, _ = oracle_run(graphs, sel, ref_seg_cpu.round_bf16_ste)
    m = _seg_model(seed, dev, K=K).train()
    with Fh.forced_graphs([i.clone() for i in graphs]), Fh.forced_selections(sel), Fh.activation_storage("bf16"), Fh.gemm_precision("bf16"):
        out = m(x.to(dev), make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
        sum((out[k].float() * w[k]).mean() for k in SEG_KEYS).backward()
    rel = lambda a, b: ((a.double() - b).norm() / (b.norm() + 1e-30)).item()
    for k in SEG_KEYS:
        rh, ry = rel(out[k].detach().cpu(), o64[k]), rel(oy[k], o64[k])
        print("configs[4] bf16 vs float64 oracle, %s: HIP %.2e | float64 with bf16-rounded operands %.2e" % (k, rh, ry))
        assert rh <= max(5e-3, 3 * ry), (k, rh, ry)
    worst = {}
    for n, q in m.named_parameters():
        if q.grad is None or n not in g64 or (n.startswith("shared_layers") and n.endswith(".bias")):
            continue
        if g64[n].norm() < 1e-9 * (1.0 + q.detach().norm().item()):            # analytically zero (a bias in front of a batch-statistics BatchNorm)
            continue
        rh, ry = rel(q.grad.cpu(), g64[n]), rel(gy[n], g64[n])
        grp = n.split(".")[0]
        if rh > worst.get(grp, (0.0, 0.0))[0]:
            worst[grp] = (rh, ry)
        assert rh <= max(5e-3, 3 * ry), (n, rh, ry)
    print("configs[4] bf16 step, worst gradient rel-L2 from the float64 oracle per module (HIP | yardstick):",
          {k: "%.3f | %.3f" % v for k, v in worst.items()})
    assert Fh.activation_storage.current == "fp32" and Fh.gemm_precision.current == Fh._lib.DEFAULT_GEMM_PRECISION


def test_segda_free_running_and_full_size(dev):
    from oracle import ref_seg_cpu
    m = _seg_model(3, dev)
    B, N = 4, 256
    x = torch.rand(B, 3, N, generator=torch.Generator().manual_seed(3)) * 2 - 1
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want, _ = ref_seg_cpu.dgcnn_defrec_forward(params, x, training=True, knn_fn=knn_canon.knn, activate_density_normal_ondef=True)
        m.train()
        got = m(x.to(dev), activate_density_normal_ondef=True)
    bad = 0
    for key in SEG_KEYS:
        d = (got[key].cpu() - want[key]).abs()
        bad += int((d > 1e-3 + 1e-3 * want[key].abs()).sum())
    assert bad <= 0.02 * B * N * 8, bad                   # a few near-tied neighbour flips may move a few rows
    # BASELINE.json configs[4] shape (N = 2048): one fwd+bwd, finite
    m2 = _seg_model(4, dev, dropout=0.5).train()
    xb = (torch.rand(8, 3, 2048, device=dev) * 2 - 1)
    out = m2(xb, activate_density_normal_ondef=True)
    sum(v.float().mean() for v in out.values()).backward()
    assert all(torch.isfinite(v).all().item() for v in out.values())
    assert all(p.grad is None or torch.isfinite(p.grad).all().item() for p in m2.parameters())


# ----------------------------------------------------------------------------- PointNet (the trainer's other --model)
def test_pointnet_vs_reference_golden(dev, golden_dir):
    """mlsp_amd.Models.PointNet on the GPU against the reference's PointNet (tests/golden/pointnet_*.npz): no dynamic graph,
    so the whole network is compared directly -- outputs within the 1e-3 contract, gradients, BN buffers."""
    from test_oracle_golden import _pointnet_from_golden
    g = dict(np.load(os.path.join(golden_dir, "pointnet_s0_B4_N256.npz")))
    m = _pointnet_from_golden(g).to(dev)
    m.train()
    logits = m(torch.from_numpy(g["x"]).to(dev), activate_DefRec=True)
    np.testing.assert_allclose(logits["cls"].detach().cpu().numpy(), g["cls"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(logits["DefRec"].detach().cpu().numpy(), g["DefRec"], rtol=1e-3, atol=1e-3)
    loss = (logits["cls"] * torch.from_numpy(g["w_cls"]).to(dev)).sum() + (logits["DefRec"] * torch.from_numpy(g["w_rec"]).to(dev)).sum()
    loss.backward()
    named = dict(m.named_parameters())
    for key in [k[5:] for k in g if k.startswith("grad/")]:
        want = g["grad/" + key]
        got = named[key].grad.cpu().numpy()
        err = np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-12)
        assert err < 5e-2, (key, err)
    for k, v in gc.state_checksums(m).items():
        if "running" in k or "num_batches" in k:
            np.testing.assert_allclose(v, g["chk_after/" + k], rtol=5e-4, atol=1e-5, err_msg=k)
    # full-size smoke: B=32, N=1024 through the drop-in import path
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mlsp_amd", "shims"))
    from PointDA.Models import PointNet
    args = gc.make_args(dropout=0.5, cuda=True)
    args.model, args.encoder_type = "pointnet", "none"
    big = PointNet(args).to(dev).train()
    out = big(torch.rand(32, 3, 1024, device=dev) * 2 - 1, activate_DefRec=True)
    assert out["cls"].shape == (32, 10) and out["DefRec"].shape == (32, 1024, 3)
    (out["cls"].sum() + out["DefRec"].sum()).backward()
    assert all(torch.isfinite(p.grad).all().item() for p in big.parameters() if p.grad is not None)


def test_bf16_operand_mode_on_segda_and_dgcnn(dev, monkeypatch):
    """BASELINE.json configs[4] arithmetic (bf16 operands, fp32 accumulation in the GEMM family): the PointSegDA model at its
    N = 2048 shape runs end to end.  Free-running, the dynamic graphs differ (near-tied neighbours flip), so distributions are compared;
    with the fp32 run's neighbour graphs forced on the bf16 run the heads are held row-wise: relative L2 <= 3e-2 (operands rounded to 8
    significand bits through ten layers; measured values are printed)."""
    from mlsp_amd import functional as Fh
    m = _seg_model(6, dev).train()
    x = (torch.rand(4, 3, 2048, generator=torch.Generator().manual_seed(6)) * 2 - 1).to(dev)
    with torch.no_grad():
        xp = x.transpose(2, 1).reshape(-1, 3)
        ref, graphs = _run_recording_graphs(monkeypatch, Fh, lambda: m(x, activate_density_normal_ondef=True))
        with Fh.gemm_precision("bf16"):
            got = m(x, activate_density_normal_ondef=True)
            with Fh.forced_graphs(graphs):
                pinned = m(x, activate_density_normal_ondef=True)
    for k in ("DefRec", "Normal", "density_mse"):
        rel = ((pinned[k].double() - ref[k].double()).norm() / ref[k].double().norm()).item()
        print("bf16 operands vs default products, graphs pinned, %s: rel-L2 %.2e" % (k, rel))
        assert rel < 3e-2, (k, rel)
    for k in ("DefRec", "Normal", "density_mse"):
        assert torch.isfinite(got[k]).all().item()
    # the dynamic graph makes a row-wise comparison meaningless (near-tied neighbours flip); compare distribution-level statistics
    for k in ("DefRec", "Normal"):
        a, b = ref[k].float(), got[k].float()
        assert abs(a.mean().item() - b.mean().item()) < 5e-2 * (a.abs().mean().item() + 1e-3) + 5e-3
        assert abs(a.std().item() - b.std().item()) < 1e-1 * a.std().item() + 5e-3
    # backward in bf16 mode: finite gradients
    with Fh.gemm_precision("bf16"):
        out = m(x, activate_density_normal_ondef=True)
        sum(v.float().mean() for v in out.values()).backward()
    assert all(p.grad is None or torch.isfinite(p.grad).all().item() for p in m.parameters())
    assert Fh.gemm_precision.current == Fh._lib.DEFAULT_GEMM_PRECISION
