"""Generate tests/golden/*.npz by running the UNMODIFIED reference (imported from
/root/reference, CPU) on seeded synthetic inputs.  Build-container only; the fixtures are
data (inputs + expected outputs) and travel to the GPU box, the reference does not.

    python tools/make_golden.py
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_import  # noqa: E402
import golden_common as gc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
GRAD_KEYS = ["conv1.conv.0.weight", "conv1.conv.1.weight", "conv4.conv.0.weight", "conv5.weight",
             "DefRec.conv1.weight", "input_transform_net.fc3.weight", "input_transform_net.conv2d2.conv.0.weight",
             "Density_cls.mlp3.weight", "C.mlp1.fc.0.weight", "bn5.bias"]


def npy(t):
    return t.detach().cpu().numpy()


def knn_case(ref_mu, seed, B, C, N, k=20):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, C, N, generator=g) * 2 - 1
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pd = -xx - inner - xx.transpose(2, 1)
    idx = ref_mu.knn(x, k)
    srt = pd.sort(dim=-1, descending=True)[0]
    if N > k:
        gap_k = (srt[..., k - 1] - srt[..., k])          # gap between kth and (k+1)th
    else:
        gap_k = torch.full(srt.shape[:-1], float('inf'))
    gap_in = (srt[..., :k - 1] - srt[..., 1:k]).min(-1)[0]   # smallest gap inside the top-k
    return dict(x=npy(x), idx=npy(idx).astype(np.int32), gap_k=npy(gap_k), gap_in=npy(gap_in))


def model_case(RefModels, ref_mu, ref_mlsp, seed, B, N, keep_feats, dtype=torch.float32, force_knn=None):
    """`dtype=torch.float64` evaluates the SAME reference modules in double precision (parameters and inputs are the fp32
    values, cast): the rounding-free limit of the reference's arithmetic.  fp32 gradients of this network carry ~1e-2 of
    summation noise at N = 1024 (both the reference's and any restatement's: tools/grad_conditioning.py), so the double
    evaluation is the well-conditioned fixture that pins gradients tightly at full N."""
    torch.manual_seed(seed)
    args = gc.make_args(dropout=0.0)
    model = RefModels.DGCNN(args)
    gc.perturb_params(model, seed)
    chk = gc.state_checksums(model)
    inp = gc.make_inputs(seed, B, N)
    out = {"x": npy(inp["x"])}
    if dtype != torch.float32:
        model = model.to(dtype)
        inp = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in inp.items()}
    for k, v in chk.items():
        out["chk/" + k] = v

    # capture intermediates with forward hooks on the max-over-k outputs (conv1..4 feed x1..x4)
    feats = {}
    model.train()
    # knn indices of the five graph stages, captured by wrapping the reference's knn
    knn_log = []
    orig_knn = ref_mu.knn
    import model_utils as bare_mu   # Models.py:10 imported get_graph_feature from the bare module

    def logging_knn(x, k):
        i = orig_knn(x, k) if force_knn is None else force_knn[len(knn_log)].clone()
        knn_log.append(i)
        return i
    bare_mu.knn = logging_knn
    ref_mu.knn = logging_knn
    try:
        logits = model(inp["x"], activate_density_normal_ondef=True)
    finally:
        bare_mu.knn = orig_knn
        ref_mu.knn = orig_knn
    assert len(knn_log) == 5, len(knn_log)
    for i, t in enumerate(knn_log):
        out["knn%d" % i] = npy(t).astype(np.int16)
    loss, parts = gc.total_loss(args, ref_mlsp, logits, inp)
    model.zero_grad()
    x_req = inp["x"]
    loss.backward()
    for k, v in logits.items():
        out["train/" + k] = npy(v)
    out["train/loss"] = npy(loss)
    for k, v in parts.items():
        out["train/loss_" + k] = npy(v)
    sd = dict(model.named_parameters())
    for k in GRAD_KEYS:
        gk = npy(sd[k].grad)
        if gk.size > 70000:          # keep fixtures small: first 64 output rows
            gk = gk[:64]
        out["grad/" + k] = gk
    if dtype != torch.float32:       # the double-precision fixture is a gradient/logit pin: no eval pass, compact storage
        for k in list(out):
            if k.startswith("train/") and out[k].ndim > 0:
                out[k] = out[k].astype(np.float32)
        return out
    st = model.state_dict()
    for k in ["conv1.conv.1.running_mean", "conv1.conv.1.running_var", "conv4.conv.1.running_var",
              "bn5.running_mean", "bn5.running_var", "DefRec.bn1.running_mean", "DefRec.bn1.running_var",
              "input_transform_net.conv2d2.conv.1.running_var", "input_transform_net.fc1.fc.1.running_mean",
              "C.mlp2.fc.1.running_var", "Density_cls.mlp1.fc.1.running_mean"]:
        out["run/" + k] = npy(st[k])

    # eval mode (running stats just updated by the step above), all heads
    model.eval()
    with torch.no_grad():
        le = model(inp["x"], activate_density_normal_ondef=True)
        emb = model(inp["x"], visualization=True)
    for k, v in le.items():
        out["eval/" + k] = npy(v)
    out["eval/embedding"] = npy(emb)
    if keep_feats:
        # x_cat is not returned by the reference; recompute it the reference's way in eval mode
        with torch.no_grad():
            x0 = ref_mu.get_graph_feature(inp["x"], args, k=20)
            T = model.input_transform_net(x0)
            out["eval/tnet"] = npy(T)
            x = torch.matmul(T, inp["x"])
            x = ref_mu.get_graph_feature(x, args, k=20)
            x1 = model.conv1(x).max(dim=-1)[0]
            out["eval/x1"] = npy(x1)
    return out


def loss_case(ref_mlsp, seed, B, N):
    args = gc.make_args()
    inp = gc.make_inputs(seed, B, N)
    g = torch.Generator().manual_seed(77 + seed)
    logits = {
        "DefRec": (inp["gold"].permute(0, 2, 1) + 0.1 * torch.randn(B, N, 3, generator=g)).requires_grad_(True),
        "Normal": torch.randn(B, N, 3, generator=g).requires_grad_(True),
    }
    lg = torch.randn(B * N, 16, generator=g).requires_grad_(True)
    p = torch.softmax(lg, dim=1)
    w = torch.arange(16, dtype=torch.float32) * 2.0
    logits["density"] = p
    logits["density_mse"] = (p * w).sum(1)
    loss, parts = gc.total_loss(args, ref_mlsp, logits, inp)
    loss.backward()
    out = dict(pred=npy(logits["DefRec"]), normal=npy(logits["Normal"]), dlogits=npy(lg),
               loss=npy(loss), g_pred=npy(logits["DefRec"].grad), g_normal=npy(logits["Normal"].grad),
               g_dlogits=npy(lg.grad))
    for k, v in parts.items():
        out["loss_" + k] = npy(v)
    # unmasked variants (MLSP/mlsp.py:275-287, :446-453 else-branches)
    with torch.no_grad():
        out["normal_unmasked"] = npy(ref_mlsp.calc_normal_loss(args, logits["Normal"], inp["normal_gt"]))
        kl, mae = ref_mlsp.densityloss(args, logits, inp["dens_val"], inp["dens_vec"])
        out["kl_unmasked"] = npy(kl)
        out["mae_unmasked"] = npy(mae)
    return out


def chamfer_dir_case(ref_mlsp, seed, B, N):
    """MLSP/mlsp.py:115-153 chamfer_distance(p1, p2, mask) on its own (one direction), gradients w.r.t. both clouds."""
    inp = gc.make_inputs(seed, B, N)
    g = torch.Generator().manual_seed(91 + seed)
    p1 = inp["gold"].permute(0, 2, 1).contiguous().requires_grad_(True)
    p2 = (inp["gold"].permute(0, 2, 1) + 0.1 * torch.randn(B, N, 3, generator=g)).requires_grad_(True)
    mask = inp["mask"].permute(0, 2, 1).contiguous()
    d = ref_mlsp.chamfer_distance(p1, p2, mask)
    d.backward()
    return dict(p1=npy(p1), p2=npy(p2), mask=npy(mask), dist=npy(d), g_p1=npy(p1.grad), g_p2=npy(p2.grad))


def main_round3():
    """Round-3 fixtures: the one-directional chamfer_distance entry."""
    _, _, ref_mlsp = ref_import.import_reference()
    for seed, B, N in [(4, 3, 256), (5, 2, 1024)]:
        f = os.path.join(OUT, "chamfer_dir_s%d_B%d_N%d.npz" % (seed, B, N))
        np.savez_compressed(f, **chamfer_dir_case(ref_mlsp, seed, B, N))
        print(f, os.path.getsize(f))


def graph_feature_case(ref_mu, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(2, 5, 64, generator=g) * 2 - 1
    f = ref_mu.get_graph_feature(x, gc.make_args(), k=20)
    return dict(x=npy(x), feat=npy(f.contiguous()))


def import_seg_reference():
    """PointSegDA/Models.py imports with torch + numpy only -- load it under its own module name."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_seg_models", ref_import.REF_ROOT + "/PointSegDA/Models.py")
    mod = importlib.util.module_from_spec(spec)
    import warnings
    warnings.filterwarnings("ignore")
    spec.loader.exec_module(mod)
    return mod


def seg_case(seg, seed, B, N, K=None):
    """DGCNN_DefRec (PointSegDA/Models.py:197-242): all heads, train mode (dropout 0) + eval mode, grads of a fixed
    linear functional of the outputs, kNN indices of the four graph stages.  `K` overrides the module constant
    PointSegDA/Models.py:6 (read by the constructors at :150,:202) -- BASELINE.json configs[4] asks for k = 40."""
    if K is not None:
        seg.K = K
    torch.manual_seed(seed)
    args = gc.make_seg_args(dropout=0.0)
    model = seg.DGCNN_DefRec(args, in_size=3, num_classes=8)
    gc.perturb_params(model, seed)
    chk = gc.state_checksums(model)
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, N, generator=g) * 2 - 1
    out = {"x": npy(x)}
    for k, v in chk.items():
        out["chk/" + k] = v
    knn_log = []
    orig = seg.knn

    def logging_knn(xx, k):
        i = orig(xx, k)
        knn_log.append(i)
        return i
    seg.knn = logging_knn
    model.train()
    try:
        logits = model(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
    finally:
        seg.knn = orig
    assert len(knn_log) == 4, len(knn_log)
    for i, t in enumerate(knn_log):
        out["knn%d" % i] = npy(t).astype(np.int16)
    loss = 0.0
    wg = torch.Generator().manual_seed(500 + seed)
    for key in ("seg", "DefRec", "Normal", "density", "density_mse"):
        w = torch.randn(logits[key].shape, generator=wg)
        out["w/" + key] = npy(w)
        loss = loss + (logits[key] * w).sum() / logits[key].numel()
        out["train/" + key] = npy(logits[key])
    out["train/loss"] = npy(loss)
    loss.backward()
    sd = dict(model.named_parameters())
    for k in ["shared_layers.conv1.weight", "shared_layers.conv2.bias", "shared_layers.conv4.weight", "shared_layers.conv5.weight",
              "shared_layers.conv6.weight", "shared_layers.conv6.bias", "input_transform_net.conv2d2.conv.0.weight",
              "input_transform_net.fc1.fc.0.weight", "input_transform_net.fc3.bias", "seg.conv1.weight", "seg.conv4.bias",
              "DefRec.conv1.weight", "Density_cls.mlp3.weight", "Norm_pred.conv2.weight"]:
        gk = npy(sd[k].grad)
        if gk.size > 70000:
            gk = gk[:64]
        out["grad/" + k] = gk
    st = model.state_dict()
    for k in ["seg.bn1.running_mean", "seg.bn1.running_var", "DefRec.bn3.running_var", "Density_cls.mlp2.fc.1.running_mean"]:
        out["run/" + k] = npy(st[k])
    model.eval()
    with torch.no_grad():
        le = model(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
    for k, v in le.items():
        out["eval/" + k] = npy(v)
    return out


def pointnet_case(RefModels, seed, B, N):
    """The reference's PointNet (PointDA/Models.py:24-79): forward with the DefRec head, a scalar loss, backward, BN buffers."""
    args = gc.make_args()
    args.model, args.encoder_type = "pointnet", "none"
    torch.manual_seed(seed)
    m = RefModels.PointNet(args)
    m.train()
    with torch.no_grad():
        gc.perturb_params(m, seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, N, generator=g) * 2 - 1
    before = gc.state_checksums(m)
    logits = m(x, activate_DefRec=True)
    w_cls = torch.randn(logits["cls"].shape, generator=g)
    w_rec = torch.randn(logits["DefRec"].shape, generator=g)
    loss = (logits["cls"] * w_cls).sum() + (logits["DefRec"] * w_rec).sum()
    loss.backward()
    out = {"x": npy(x), "cls": npy(logits["cls"]), "DefRec": npy(logits["DefRec"]), "w_cls": npy(w_cls), "w_rec": npy(w_rec),
           "loss": npy(loss)}
    for k, v in before.items():
        out["chk_before/" + k] = v
    for k, v in gc.state_checksums(m).items():
        out["chk_after/" + k] = v
    for k in POINTNET_GRAD_KEYS:                               # a spread of layers; the full gradient set would be 15 MB
        out["grad/" + k] = npy(dict(m.named_parameters())[k].grad)
    return out


POINTNET_GRAD_KEYS = ["conv1.conv.0.weight", "conv2.conv.0.weight", "conv3.conv.1.weight", "conv4.conv.0.weight", "conv5.conv.1.bias",
                      "trans_net1.fc3.weight", "trans_net1.conv2d1.conv.0.weight", "trans_net2.conv2d1.conv.0.weight", "trans_net2.fc3.bias",
                      "trans_net2.fc2.fc.1.weight", "C.mlp1.fc.1.weight", "C.mlp3.weight", "DefRec.bn1.weight", "DefRec.conv4.weight"]


def main_pointnet():
    ref_import.install_stubs()
    RefModels, _, _ = ref_import.import_reference()
    np.savez_compressed(os.path.join(OUT, "pointnet_s0_B4_N256.npz"), **pointnet_case(RefModels, 0, 4, 256))


def import_sa_reference():
    """PointDA/hengshuang_transformer/pointnet_util.py is pure torch + numpy: it imports with no stubs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_pointnet_util", os.path.join(ref_import.REF_ROOT, "PointDA", "hengshuang_transformer",
                                                                                    "pointnet_util.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sa_case(pu, seed, B, N, D, npoint, radius, nsample, mlp, group_all):
    """One PointNetSetAbstraction layer of the reference: forward, a scalar loss, backward, BN buffers after the step."""
    torch.manual_seed(seed)
    layer = pu.PointNetSetAbstraction(npoint, radius, nsample, 3 + D, mlp, group_all)
    layer.train()
    with torch.no_grad():
        gc.perturb_params(layer, seed)
        g = torch.Generator().manual_seed(77 + seed)
        for name, p in layer.named_parameters():               # BN affine terms off their init (names differ from DGCNN's)
            if "mlp_bns" in name:
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 if name.endswith("bias") else 0.5) + (0.0 if name.endswith("bias") else 1.0))
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    points = (torch.randn(B, N, D, generator=g)).requires_grad_(True) if D else None
    before = {k: npy(v).copy() for k, v in layer.state_dict().items()}
    out = {"xyz": npy(xyz), "cfg": np.array([npoint or 0, nsample or 0, int(group_all), D] + list(mlp)), "radius": np.array(radius or 0.0)}
    if D:
        out["points"] = npy(points)
    if not group_all:
        torch.manual_seed(1234 + seed)                          # the first RNG draw of forward() is the FPS start (:65)
        fps_idx = pu.farthest_point_sample(xyz, npoint)
        new_xyz_ref = pu.index_points(xyz, fps_idx)
        out["fps_idx"] = npy(fps_idx)
        out["start"] = npy(fps_idx[:, 0])
        out["group_idx"] = npy(pu.query_ball_point(radius, nsample, xyz, new_xyz_ref))
    torch.manual_seed(1234 + seed)
    new_xyz, new_points = layer(xyz, points)
    wgt = torch.randn(new_points.shape, generator=g)
    loss = (new_points * wgt).sum()
    loss.backward()
    out.update({"new_xyz": npy(new_xyz), "new_points": npy(new_points), "wgt": npy(wgt), "loss": npy(loss)})
    if D:
        out["d_points"] = npy(points.grad)
    for k, v in before.items():
        out["state_before/" + k] = v
    for k, v in layer.state_dict().items():
        out["state_after/" + k] = npy(v)
    for k, p in layer.named_parameters():
        out["grad/" + k] = npy(p.grad)
    return out


def sa_index_case(pu, seed, B, N, npoint, radius, nsample):
    """Indices only, at BASELINE.json configs[3] scale (N = 2048): FPS + ball query of the reference."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    torch.manual_seed(99 + seed)
    fps_idx = pu.farthest_point_sample(xyz, npoint)
    gidx = pu.query_ball_point(radius, nsample, xyz, pu.index_points(xyz, fps_idx))
    return {"xyz": npy(xyz), "start": npy(fps_idx[:, 0]), "fps_idx": fps_idx.numpy().astype(np.int16),
            "group_idx": gidx.numpy().astype(np.int16), "radius": np.array(radius), "nsample": np.array(nsample)}


def main():
    os.makedirs(OUT, exist_ok=True)
    RefModels, ref_mu, ref_mlsp = ref_import.import_reference()
    torch.set_num_threads(8)
    for seed, B, C, N in [(0, 2, 3, 256), (1, 2, 3, 1024), (2, 2, 64, 256), (3, 1, 128, 1024), (4, 3, 3, 20), (5, 2, 64, 33)]:
        np.savez_compressed(os.path.join(OUT, "knn_s%d_C%d_N%d.npz" % (seed, C, N)), **knn_case(ref_mu, seed, B, C, N))
    np.savez_compressed(os.path.join(OUT, "graph_feature.npz"), **graph_feature_case(ref_mu, 0))
    for seed, B, N in [(0, 2, 256), (1, 2, 1024)]:
        np.savez_compressed(os.path.join(OUT, "loss_s%d_N%d.npz" % (seed, N)), **loss_case(ref_mlsp, seed, B, N))
    for seed, B, N, keep in [(0, 6, 256, True), (1, 4, 1024, False), (2, 4, 128, False)]:
        np.savez_compressed(os.path.join(OUT, "dgcnn_s%d_B%d_N%d.npz" % (seed, B, N)),
                            **model_case(RefModels, ref_mu, ref_mlsp, seed, B, N, keep))
    seg = import_seg_reference()
    for seed, B, N in [(0, 4, 256), (1, 3, 512)]:
        np.savez_compressed(os.path.join(OUT, "segda_s%d_B%d_N%d.npz" % (seed, B, N)), **seg_case(seg, seed, B, N))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


def corrupt_cases():
    """deform_input / assign_region_to_point / PCM.mix_shapes of the reference with seeded (and recorded) random draws."""
    import types
    ref_import.install_stubs()
    sys.path.insert(0, ref_import.REF_ROOT)
    import utils.pc_utils as rpc                                   # noqa: E402  (reference, CPU)
    from MLSP import mlsp as rmlsp, PCM as rpcm                      # noqa: E402
    g = torch.Generator().manual_seed(5)
    B, N = 6, 1024
    X = torch.rand(B, 3, N, generator=g) * 2.2 - 1.1              # some points outside [-1,1] (clamped), all voxels populated
    X[0, :, :8] = torch.tensor([[-1.0, -1 / 3, 1 / 3, 1.0, 0.0, -0.33333334, 0.33333334, 0.9999999]]).expand(3, 8)   # voxel faces
    X[1, :, :600] = X[1, :, :600] * 0.3 + 0.6                      # a crowded voxel
    X[2] = X[2] * 0.05                                             # everything in the centre voxel
    regions = rpc.assign_region_to_point(X.clone(), "cpu")
    lookup = torch.Tensor(rpc.region_mean(3))
    out = {"X": npy(X), "regions": npy(regions), "lookup": npy(lookup)}
    for groups in (1, 3):
        np.random.seed(11 + groups)
        perm = np.random.permutation(27)                            # first draw inside deform_input (:27)
        np.random.seed(11 + groups)
        Xd, mask = rmlsp.deform_input(X.clone(), lookup, 'volume_based_voxels', 'cpu', groups)
        out["perm_g%d" % groups] = perm
        out["X_out_g%d" % groups] = npy(Xd)
        out["mask_g%d" % groups] = npy(mask)
    np.savez_compressed(os.path.join(OUT, "deform_s5_B6_N1024.npz"), **out)

    # scan_input: python `random` draws the pixel size, numpy the three rotation angles of every cloud (replayed from the seeds)
    import random
    np.int = int                                                   # the reference still spells np.int (removed in numpy 1.24)
    Xs = (torch.rand(5, 512, 3, generator=g) * 2 - 1) * 0.55        # inside the unit ball, as the datasets are normalised
    random.seed(21)
    np.random.seed(21)
    px = random.uniform(0.045, 0.075)
    angs = np.stack([np.random.rand(3) * 2 * np.pi for _ in range(5)])
    random.seed(21)
    np.random.seed(21)
    Xo, mo = rmlsp.scan_input(Xs.clone(), "cpu")
    np.savez_compressed(os.path.join(OUT, "scan_s21_B5_N512.npz"), X=npy(Xs), pixel_size=np.array(px), angles=angs, X_out=npy(Xo), mask=npy(mo))

    # PCM.mix_shapes: record the reference's random draws by wrapping the RNG entry points it calls
    rec = {"randperm": [], "randint": []}
    orig_randperm, orig_randint, orig_beta = torch.randperm, torch.randint, np.random.beta

    def randperm(n, *a, **k):
        r = orig_randperm(n, *a, **k)
        rec["randperm"].append(r.clone())
        return r

    def randint(*a, **k):
        r = orig_randint(*a, **k)
        rec["randint"].append(r.clone())
        return r

    args = types.SimpleNamespace(cuda=False, mixup_params=1.0)
    Xp = torch.rand(5, 3, 256, generator=g) * 2 - 1
    Yp = torch.arange(5)
    torch.manual_seed(3)
    np.random.seed(3)
    lam = np.random.beta(1.0, 1.0)
    np.random.seed(3)
    torch.randperm, torch.randint = randperm, randint
    try:
        mixed, (Ya, Yb, lam2) = rpcm.mix_shapes(args, Xp.clone(), Yp)
    finally:
        torch.randperm, torch.randint = orig_randperm, orig_randint
    assert lam == lam2
    np.savez_compressed(os.path.join(OUT, "pcm_s3_B5_N256.npz"), X=npy(Xp), Y=npy(Yp), lam=np.array(lam), index=npy(rec["randperm"][0]),
                        points_perm=npy(rec["randperm"][1]), start_a=npy(rec["randint"][0]), start_b=npy(rec["randint"][1]),
                        mixed=npy(mixed), Ya=npy(Ya), Yb=npy(Yb))


def radius_case():
    """deform_input(..., 'volume_based_radius') of the reference (pc_utils.collapse_to_point): the picked points (np.random.choice)
    and the Gaussian draws (draw_from_gaussian) are recorded by wrapping the two entry points."""
    ref_import.install_stubs()
    sys.path.insert(0, ref_import.REF_ROOT)
    import utils.pc_utils as rpc                                   # noqa: E402
    from MLSP import mlsp as rmlsp                                   # noqa: E402
    g = torch.Generator().manual_seed(9)
    B, N = 5, 512
    X = (torch.rand(B, 3, N, generator=g) * 2 - 1) * torch.tensor([1.0, 0.8, 0.6, 0.9, 0.3]).view(B, 1, 1)
    rec = {"choice": [], "gauss": []}
    orig_choice, orig_gauss = np.random.choice, rpc.draw_from_gaussian

    def choice(a, *aa, **kk):
        r = orig_choice(a, *aa, **kk)
        rec["choice"].append(int(r))
        return r

    def gauss(mean, n):
        r = orig_gauss(mean, n)
        rec["gauss"].append((np.array(mean, dtype=np.float64), r.copy()))
        return r
    np.random.seed(17)
    np.random.choice, rpc.draw_from_gaussian = choice, gauss
    try:
        Xd, mask = rmlsp.deform_input(X.clone(), torch.Tensor(rpc.region_mean(3)), 'volume_based_radius', 'cpu')
    finally:
        np.random.choice, rpc.draw_from_gaussian = orig_choice, orig_gauss
    noise = np.zeros((B, 3, N), np.float64)
    for b in range(B):
        mean, draw = rec["gauss"][b]
        hit = npy(mask)[b, 0] > 0
        noise[b][:, hit] = (draw - mean[:, None]) / np.sqrt(0.001)       # the standard-normal draws behind the reference's samples
    np.savez_compressed(os.path.join(OUT, "collapse_s9_B5_N512.npz"), X=npy(X), X_out=npy(Xd), mask=npy(mask),
                        choice=np.array(rec["choice"], np.int32), noise=noise.astype(np.float32))
    print("collapse_s9_B5_N512.npz", os.path.getsize(os.path.join(OUT, "collapse_s9_B5_N512.npz")), "picked", rec["choice"],
          "moved per cloud", npy(mask)[:, 0].sum(1))


def main_round2():
    """Round-2 fixtures: the reference in float64 at N = 1024 (gradient pin at full N) and PointSegDA with K = 40."""
    RefModels, ref_mu, ref_mlsp = ref_import.import_reference()
    torch.set_num_threads(8)
    f64 = model_case(RefModels, ref_mu, ref_mlsp, 3, 4, 1024, False, dtype=torch.float64)
    # the reference's OWN fp32 run on the same inputs and graphs: its distance from the double-precision gradients is the
    # rounding noise of this step in fp32 -- the yardstick the HIP path's fp32 gradients are held to (no worse than that)
    f32 = model_case(RefModels, ref_mu, ref_mlsp, 3, 4, 1024, False,
                     force_knn=[torch.from_numpy(f64["knn%d" % i].astype(np.int64)) for i in range(5)])
    for k in GRAD_KEYS:
        a, b = f32["grad/" + k].astype(np.float64), f64["grad/" + k]
        f64["ref32_rel/" + k] = np.array(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
        print("reference fp32 vs float64 %-45s %.2e" % (k, f64["ref32_rel/" + k]))
    np.savez_compressed(os.path.join(OUT, "dgcnn_f64_s3_B4_N1024.npz"), **f64)
    seg = import_seg_reference()
    np.savez_compressed(os.path.join(OUT, "segda_k40_s2_B2_N512.npz"), **seg_case(seg, 2, 2, 512, K=40))
    for f in ("dgcnn_f64_s3_B4_N1024.npz", "segda_k40_s2_B2_N512.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


def _perturb_sa(layer, seed):
    with torch.no_grad():
        gc.perturb_params(layer, seed)
        g = torch.Generator().manual_seed(77 + seed)
        for name, p in layer.named_parameters():               # BN affine terms off their init
            if "bn" in name:
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 if name.endswith("bias") else 0.5) + (0.0 if name.endswith("bias") else 1.0))


def sa_round2_cases(pu):
    """Round-2 set-abstraction fixtures from the reference file itself: knn=True grouping, PointNetSetAbstractionMsg (ball query and
    knn branches), PointNetFeaturePropagation (3-NN interpolation and the S == 1 broadcast)."""
    out = {}
    # (1) sample_and_group(knn=True) inside PointNetSetAbstraction
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(11)
    B, N, D = 3, 300, 5
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    pts = torch.randn(B, N, D, generator=g).requires_grad_(True)
    layer = pu.PointNetSetAbstraction(48, 0.3, 12, 3 + D, [16, 32], False, knn=True).train()
    _perturb_sa(layer, 11)
    torch.manual_seed(2024)
    fps_idx = pu.farthest_point_sample(xyz, 48)
    torch.manual_seed(2024)
    new_xyz, new_points = layer(xyz, pts)
    w = torch.randn(new_points.shape, generator=g)
    (new_points * w).sum().backward()
    c = {"xyz": npy(xyz), "points": npy(pts), "start": npy(fps_idx[:, 0]), "fps_idx": npy(fps_idx), "new_xyz": npy(new_xyz),
         "new_points": npy(new_points), "wgt": npy(w), "d_points": npy(pts.grad),
         "group_idx": npy(pu.square_distance(pu.index_points(xyz, fps_idx), xyz).argsort()[:, :, :12])}
    for k, v in layer.state_dict().items():
        c["state_after/" + k] = npy(v)
    for k, p in layer.named_parameters():
        c["grad/" + k] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, "sa_knn_s11_B3_N300.npz"), **c)
    # (2) PointNetSetAbstractionMsg, ball-query branches with features, and a knn variant without features
    for tag, knn, D in (("msg", False, 6), ("msgknn", True, 0)):
        g = torch.Generator().manual_seed(12)
        B, N = 2, 256
        xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
        pts = torch.randn(B, N, D, generator=g).requires_grad_(True) if D else None
        torch.manual_seed(12)
        layer = pu.PointNetSetAbstractionMsg(32, [0.2, 0.45], [8, 16], D, [[16, 24], [16, 32]], knn=knn).train()
        _perturb_sa(layer, 12)
        torch.manual_seed(7)
        fps_idx = pu.farthest_point_sample(xyz, 32)
        new_xyz, new_points = layer(xyz, pts, seed_idx=fps_idx)
        w = torch.randn(new_points.shape, generator=g)
        (new_points * w).sum().backward()
        c = {"xyz": npy(xyz), "fps_idx": npy(fps_idx), "new_xyz": npy(new_xyz), "new_points": npy(new_points), "wgt": npy(w)}
        if D:
            c["points"], c["d_points"] = npy(pts), npy(pts.grad)
        for k, v in layer.state_dict().items():
            c["state_after/" + k] = npy(v)
        for k, p in layer.named_parameters():
            c["grad/" + k] = npy(p.grad)
        np.savez_compressed(os.path.join(OUT, "sa_%s_s12_B2_N256.npz" % tag), **c)
    # (3) PointNetFeaturePropagation: S = 40 (3-NN interpolation) and S = 1 (broadcast)
    for tag, S in (("fp", 40), ("fp1", 1)):
        g = torch.Generator().manual_seed(13)
        B, N, D1, D2 = 2, 200, 4, 7
        xyz1 = (torch.rand(B, 3, N, generator=g) * 2 - 1)
        xyz2 = xyz1[:, :, :S].clone() + 0.01 * torch.randn(B, 3, S, generator=g)
        p1 = torch.randn(B, D1, N, generator=g).requires_grad_(True)
        p2 = torch.randn(B, D2, S, generator=g).requires_grad_(True)
        torch.manual_seed(13)
        layer = pu.PointNetFeaturePropagation(D1 + D2, [16, 12]).train()
        _perturb_sa(layer, 13)
        outp = layer(xyz1, xyz2, p1, p2)
        w = torch.randn(outp.shape, generator=g)
        (outp * w).sum().backward()
        c = {"xyz1": npy(xyz1), "xyz2": npy(xyz2), "points1": npy(p1), "points2": npy(p2), "out": npy(outp), "wgt": npy(w),
             "d_points1": npy(p1.grad), "d_points2": npy(p2.grad)}
        for k, v in layer.state_dict().items():
            c["state_after/" + k] = npy(v)
        for k, p in layer.named_parameters():
            c["grad/" + k] = npy(p.grad)
        np.savez_compressed(os.path.join(OUT, "sa_%s_s13_B2_N200.npz" % tag), **c)
    for f in sorted(os.listdir(OUT)):
        if f.startswith(("sa_knn", "sa_msg", "sa_fp")):
            print(f, os.path.getsize(os.path.join(OUT, f)))


def main_sa():
    pu = import_sa_reference()
    np.savez_compressed(os.path.join(OUT, "sa_s0_B4_N256.npz"), **sa_case(pu, 0, 4, 256, 6, 64, 0.4, 16, [32, 32, 64], False))
    np.savez_compressed(os.path.join(OUT, "sa_s1_B3_N200_nofeat.npz"), **sa_case(pu, 1, 3, 200, 0, 50, 0.3, 8, [16, 24], False))
    np.savez_compressed(os.path.join(OUT, "sa_s2_B4_N128_all.npz"), **sa_case(pu, 2, 4, 128, 16, None, None, None, [32, 64], True))
    np.savez_compressed(os.path.join(OUT, "sa_idx_s3_B2_N2048.npz"), **sa_index_case(pu, 3, 2, 2048, 512, 0.2, 32))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sa":
        main_sa()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pointnet":
        main_pointnet()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "corrupt":
        corrupt_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sa2":
        sa_round2_cases(import_sa_reference())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "radius":
        radius_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round3":
        main_round3()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        main_round2()
        sys.exit(0)
    main()
