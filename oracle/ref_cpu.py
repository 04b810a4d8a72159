"""ORACLE -- test infrastructure, NOT the product.

A CPU restatement (plain torch fp32 ops, our own words) of the reference's data-parallel hot
path: DGCNN encoder + MLSP heads (PointDA/Models.py, PointDA/model_utils.py) and the three
MLSP losses (MLSP/mlsp.py).  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file; the product path (mlsp_amd/) never does and fails
loudly without its HIP library.

Parity pin: tests/test_oracle_golden.py checks every function here against fixtures captured
from the unmodified reference (tools/make_golden.py -> tests/golden/*.npz).

Everything is functional: parameters and buffers come in as a flat dict with the reference's
state_dict key names (strict-load compatible, PointDA/train_spst.py:141), so the oracle has
no nn.Module of its own and can be driven with the product module's tensors directly.
"""
import torch
import torch.nn.functional as F

K_DEFAULT = 20          # PointDA/Models.py:13
BN_EPS = 1e-5
BN_MOM = 0.1
DEFREC_SCALER = 20.0    # MLSP/mlsp.py:7


# --------------------------------------------------------------------------- graph ops
def knn_reference_formula(x, k):
    """PointDA/model_utils.py:9-16.  x [B,C,N] -> idx [B,N,k] (int64), nearest first."""
    inner = -2.0 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x * x, dim=1, keepdim=True)
    pd = -xx - inner - xx.transpose(2, 1)
    return pd.topk(k=k, dim=-1)[1]


def graph_feature(x, idx):
    """PointDA/model_utils.py:18-42.  x [B,C,N], idx [B,N,k] -> [B,2C,N,k] = cat(nbr-ctr, ctr)."""
    B, C, N = x.shape
    k = idx.shape[-1]
    xt = x.transpose(2, 1).reshape(B * N, C)
    flat = (idx.long() + torch.arange(B, device=x.device).view(B, 1, 1) * N).reshape(-1)
    nbr = xt[flat].view(B, N, k, C)
    ctr = xt.view(B, N, 1, C).expand(B, N, k, C)
    return torch.cat((nbr - ctr, ctr), dim=3).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------- layer blocks
class _Ctx:
    """Carries mode flags and collects BatchNorm running-stat updates."""

    def __init__(self, params, training, dropout_p, knn_fn):
        self.p = params
        self.training = training
        self.dropout_p = dropout_p
        self.knn_fn = knn_fn
        self.new_buffers = {}
        self.knn_idx = []
        self.sel = []          # arg-max of every max-pool, in call order (tests: mlsp_amd.functional.forced_selections)
        self.feats = {}
        self.force_sel = None  # tests: an iterator over selections recorded by ANOTHER evaluation -- the pools take those entries instead of
                               # their own arg-max (float64 truth, a rounding emulation and the HIP path then differ in arithmetic only)
        self.q = None          # tests: operand rounding emulation (oracle/ref_seg_cpu.py bf16 yardstick); None = exact

    def bn(self, y, prefix, reduce_dims):
        """torch BatchNorm semantics (model_utils.py:56-58): biased var to normalise, unbiased
        for the running estimate, momentum 0.1, eps 1e-5.  y is [B,C,...]."""
        g, b = self.p[prefix + ".weight"], self.p[prefix + ".bias"]
        shape = [1, -1] + [1] * (y.dim() - 2)
        if self.training:
            n = y.numel() // y.shape[1]
            mean = y.mean(dim=reduce_dims)
            var = y.var(dim=reduce_dims, unbiased=False)
            with torch.no_grad():
                # chained: a module invoked twice in one forward (DefRec under activate_DefRec + ..._ondef,
                # PointSegDA/Models.py:227-236) updates its running statistics twice, as nn.BatchNorm does
                nb = self.new_buffers
                rm = nb.get(prefix + ".running_mean", self.p[prefix + ".running_mean"])
                rv = nb.get(prefix + ".running_var", self.p[prefix + ".running_var"])
                nt = nb.get(prefix + ".num_batches_tracked", self.p[prefix + ".num_batches_tracked"])
                nb[prefix + ".running_mean"] = (1 - BN_MOM) * rm + BN_MOM * mean
                nb[prefix + ".running_var"] = (1 - BN_MOM) * rv + BN_MOM * var * (n / max(n - 1, 1))
                nb[prefix + ".num_batches_tracked"] = nt + 1
        else:
            mean, var = self.p[prefix + ".running_mean"], self.p[prefix + ".running_var"]
        yhat = (y - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS)
        return yhat * g.view(shape) + b.view(shape)

    def max_k(self, x):
        """max over the k neighbours of [B, C, N, k]; records the winning slot as [B*N, C] (the HIP path's point-major layout)"""
        if self.force_sel is not None:
            a = next(self.force_sel).view(x.shape[0], x.shape[2], x.shape[1]).permute(0, 2, 1)
            v = x.gather(-1, a.unsqueeze(-1)).squeeze(-1)
        else:
            v, a = x.max(dim=-1)
        self.sel.append(a.permute(0, 2, 1).reshape(-1, a.shape[1]))
        return v

    def max_n(self, x):
        """max over the N points of [B, C, N]; records the winning point (local to its cloud) as [B, C]"""
        if self.force_sel is not None:
            a = next(self.force_sel)
            v = x.gather(2, a.unsqueeze(2)).squeeze(2)
        else:
            v, a = x.max(dim=2)
        self.sel.append(a)
        return v

    def drop(self, x):
        if self.training and self.dropout_p > 0:
            return F.dropout(x, self.dropout_p, True)
        return x


def _conv2d_bn_lrelu(c, x, prefix):
    """conv_2d (model_utils.py:45-63): 1x1 Conv2d(bias=False for dgcnn) + BN2d + LeakyReLU(0.2)."""
    w = c.p[prefix + ".conv.0.weight"]                      # [Cout,Cin,1,1]
    y = torch.einsum("oc,bcnk->bonk", w[:, :, 0, 0], x)
    bkey = prefix + ".conv.0.bias"
    if bkey in c.p:
        y = y + c.p[bkey].view(1, -1, 1, 1)
    y = c.bn(y, prefix + ".conv.1", (0, 2, 3))
    return F.leaky_relu(y, 0.2)


def _fc_bn_act(c, x, prefix, act="leakyrelu"):
    """fc_layer (model_utils.py:66-87): Linear + BN1d + activation."""
    y = x @ c.p[prefix + ".fc.0.weight"].t()
    bkey = prefix + ".fc.0.bias"
    if bkey in c.p:
        y = y + c.p[bkey]
    y = c.bn(y, prefix + ".fc.1", (0,))
    return F.leaky_relu(y, 0.2) if act == "leakyrelu" else F.relu(y)


def _transform_net(c, x0, prefix="input_transform_net"):
    """transform_net.forward (model_utils.py:108-127), dgcnn branch."""
    x = _conv2d_bn_lrelu(c, x0, prefix + ".conv2d1")
    x = _conv2d_bn_lrelu(c, x, prefix + ".conv2d2")
    x = c.max_k(x).unsqueeze(-1)
    x = _conv2d_bn_lrelu(c, x, prefix + ".conv2d3")
    x = c.max_n(x[..., 0]).reshape(x.shape[0], -1)
    x = _fc_bn_act(c, x, prefix + ".fc1")
    x = _fc_bn_act(c, x, prefix + ".fc2")
    x = x @ c.p[prefix + ".fc3.weight"].t() + c.p[prefix + ".fc3.bias"]
    x = x + torch.eye(3, device=x.device).reshape(1, 9)
    return x.view(-1, 3, 3)


def _classifier(c, x, prefix="C"):
    """classifier.forward (model_utils.py:142-146)."""
    x = c.drop(_fc_bn_act(c, x, prefix + ".mlp1"))
    x = c.drop(_fc_bn_act(c, x, prefix + ".mlp2"))
    return x @ c.p[prefix + ".mlp3.weight"].t() + c.p[prefix + ".mlp3.bias"]


def _conv1d(c, x, key):
    return torch.einsum("oc,bcn->bon", c.p[key][:, :, 0], x)


def _region_head(c, x, prefix):
    """RegionReconstruction / Normal_prediction .forward (Models.py:192-197, 226-231)."""
    x = c.drop(F.relu(c.bn(_conv1d(c, x, prefix + ".conv1.weight"), prefix + ".bn1", (0, 2))))
    x = c.drop(F.relu(c.bn(_conv1d(c, x, prefix + ".conv2.weight"), prefix + ".bn2", (0, 2))))
    x = F.relu(c.bn(_conv1d(c, x, prefix + ".conv3.weight"), prefix + ".bn3", (0, 2)))
    x = _conv1d(c, x, prefix + ".conv4.weight")
    return x.permute(0, 2, 1)


def _density_head(c, x, prefix="Density_cls"):
    """Density_prediction.forward (Models.py:272-285).  Note dp1 is used twice (:273,:278)."""
    x = c.drop(F.relu(c.bn(_conv1d(c, x, prefix + ".conv1.weight"), prefix + ".bn1", (0, 2))))
    x = x.permute(0, 2, 1).reshape(-1, x.shape[1])
    x = c.drop(_fc_bn_act(c, x, prefix + ".mlp1"))
    x = c.drop(_fc_bn_act(c, x, prefix + ".mlp2"))
    logits = x @ c.p[prefix + ".mlp3.weight"].t() + c.p[prefix + ".mlp3.bias"]
    p_vec = F.softmax(logits, dim=1)
    density = p_vec @ c.p[prefix + ".fc2.weight"].t()
    return p_vec, density[:, 0]


def dgcnn_forward(params, x, training=True, dropout_p=0.0, k=K_DEFAULT, knn_fn=knn_reference_formula,
                  visualization=False, activate_DefRec=False, activate_normal=False, activate_scan=False,
                  activate_density=False, activate_density_normal_ondef=False, return_ctx=False):
    """DGCNN.forward (PointDA/Models.py:106-162).  `params` = reference-keyed dict of tensors.
    Returns (logits dict | embedding, new_buffers dict)."""
    c = _Ctx(params, training, dropout_p, knn_fn)
    B, _, N = x.shape

    def graph(xin):
        idx = c.knn_fn(xin.detach(), k)
        c.knn_idx.append(idx)
        return graph_feature(xin, idx)

    x0 = graph(x)
    T = _transform_net(c, x0)
    c.feats["tnet"] = T
    xt = torch.matmul(T, x)
    feats = []
    h = xt
    for name in ("conv1", "conv2", "conv3", "conv4"):
        h = c.max_k(_conv2d_bn_lrelu(c, graph(h), name))
        feats.append(h)
    x_cat = torch.cat(feats, dim=1)
    c.feats["x_cat"] = x_cat
    x5 = F.leaky_relu(c.bn(_conv1d(c, x_cat, "conv5.weight"), "bn5", (0, 2)), 0.2)
    x5 = c.max_n(x5)
    logits = {"cls": _classifier(c, x5)}
    if visualization:
        out = x5
    else:
        head_in = None
        if activate_DefRec or activate_normal or activate_scan or activate_density or activate_density_normal_ondef:
            head_in = torch.cat((x_cat, x5.unsqueeze(2).expand(B, x5.shape[1], N)), dim=1)
        if activate_DefRec:
            logits["DefRec"] = _region_head(c, head_in, "DefRec")
        if activate_normal:
            logits["Normal"] = _region_head(c, head_in, "Norm_pred")
        if activate_scan:
            logits["Rec_scan"] = _region_head(c, head_in, "Rec_scan")
        if activate_density:
            logits["density"], logits["density_mse"] = _density_head(c, head_in)
        if activate_density_normal_ondef:
            logits["DefRec"] = _region_head(c, head_in, "DefRec")
            logits["density"], logits["density_mse"] = _density_head(c, head_in)
            logits["Normal"] = _region_head(c, head_in, "Norm_pred")
        out = logits
    if return_ctx:
        return out, c.new_buffers, c
    return out, c.new_buffers


# --------------------------------------------------------------------------- PointNet (the trainer's other --model)
def _conv2d_bn_relu(c, x, prefix):
    """conv_2d with the pointnet defaults (model_utils.py:45-63): 1x1 Conv2d + bias + BN2d + ReLU.  x [B,C,N,1]."""
    y = torch.einsum("oc,bcnk->bonk", c.p[prefix + ".conv.0.weight"][:, :, 0, 0], x) + c.p[prefix + ".conv.0.bias"].view(1, -1, 1, 1)
    return F.relu(c.bn(y, prefix + ".conv.1", (0, 2, 3)))


def _transform_net_pn(c, x, prefix, K):
    """transform_net.forward (model_utils.py:108-127), non-dgcnn branch: ReLU, biases, no max over the last axis."""
    for name in ("conv2d1", "conv2d2", "conv2d3"):
        x = _conv2d_bn_relu(c, x, prefix + "." + name)
    x = x.max(dim=2)[0].reshape(x.shape[0], -1)
    x = _fc_bn_act(c, x, prefix + ".fc1", act="relu")
    x = _fc_bn_act(c, x, prefix + ".fc2", act="relu")
    x = x @ c.p[prefix + ".fc3.weight"].t() + c.p[prefix + ".fc3.bias"]
    return (x + torch.eye(K).reshape(1, K * K)).view(-1, K, K)


def pointnet_forward(params, x, training=True, dropout_p=0.0, activate_DefRec=False):
    """PointNet.forward (PointDA/Models.py:43-79).  Returns (logits dict, new_buffers dict)."""
    c = _Ctx(params, training, dropout_p, None)
    B, _, N = x.shape
    h = x.unsqueeze(3)
    T1 = _transform_net_pn(c, h, "trans_net1", 3)
    h = torch.bmm(x.transpose(2, 1), T1).transpose(2, 1).unsqueeze(3)
    x1 = _conv2d_bn_relu(c, h, "conv1")
    x2 = _conv2d_bn_relu(c, x1, "conv2")
    T2 = _transform_net_pn(c, x2, "trans_net2", 64)
    h = torch.bmm(x2.squeeze(3).transpose(2, 1), T2).transpose(2, 1).unsqueeze(3)
    x3 = _conv2d_bn_relu(c, h, "conv3")
    x4 = _conv2d_bn_relu(c, x3, "conv4")
    x_cat = torch.cat((x1, x2, x3, x4), dim=1).squeeze(3)
    x5 = _conv2d_bn_relu(c, x4, "conv5").max(dim=2)[0]            # [B,1024,1]
    cls_in = x5.squeeze(2)
    h1 = c.drop(_fc_bn_act(c, cls_in, "C.mlp1", act="relu"))
    h2 = c.drop(_fc_bn_act(c, h1, "C.mlp2", act="relu"))
    logits = {"cls": h2 @ c.p["C.mlp3.weight"].t() + c.p["C.mlp3.bias"]}
    if activate_DefRec:
        logits["DefRec"] = _region_head(c, torch.cat((x_cat, x5.expand(B, 1024, N)), dim=1), "DefRec")
    return logits, c.new_buffers


# --------------------------------------------------------------------------- losses
def chamfer_distance(p1, p2, mask):
    """MLSP/mlsp.py:115-153.  p1,p2 [B,N,3]; mask [B,N,3] (0/1).  For each masked point of p1 the
    squared distance to the nearest MASKED point of p2 (+100 penalty on unmasked columns),
    averaged over the masked points of each cloud, summed over the batch."""
    d = p1.unsqueeze(2) - p2.unsqueeze(1)                    # [B,N1,N2,3]
    dist = torch.norm(d, 2, dim=3) ** 2                      # sqrt-then-square, as the reference
    mc = mask[:, :, 0]
    pen = torch.where(mc == 0, torch.full_like(mc, 100.0), torch.zeros_like(mc))
    dist = dist + pen.unsqueeze(1)
    dmin = dist.min(dim=2)[0]
    per_cloud = (dmin * mc).sum(1) / mc.sum(1)
    return per_cloud.sum()


def reconstruction_loss(pred, gold, mask):
    """MLSP/mlsp.py:156-182.  pred [B,N,3]; gold, mask [B,3,N]."""
    gold = gold.permute(0, 2, 1)
    mask = mask.permute(0, 2, 1)
    return (chamfer_distance(gold, pred, mask) + chamfer_distance(pred, gold, mask)) / pred.shape[0]


def calc_loss(args, logits, labels, mask):
    """MLSP/mlsp.py:222-229."""
    return args.DefRec_weight * reconstruction_loss(logits["DefRec"], labels, mask) * DEFREC_SCALER


def calc_scan_loss(args, logits, labels, mask):
    """MLSP/mlsp.py:231-238."""
    return args.Scan_Rec_weight * reconstruction_loss(logits["Rec_scan"], labels, mask) * DEFREC_SCALER


def normal_prediction_loss(pred, gt, weight=None):
    """MLSP/mlsp.py:275-283 (weight=None) and the masked variant inlined at
    PointDA/trainer.py:551-556 (weight = mask*26+1, [B,N])."""
    cos = (F.normalize(pred, p=2, dim=-1) * F.normalize(gt, p=2, dim=-1)).sum(-1).abs()
    if weight is None:
        return -cos.sum() / (pred.shape[0] * pred.shape[1])
    return -(cos * weight).sum() / weight.sum()


def calc_normal_loss(args, prediction, labels):
    """MLSP/mlsp.py:285-287."""
    return args.normal_pred_weight * normal_prediction_loss(prediction, labels)


def densityloss(args, logits, target, target_vec, mask=None):
    """MLSP/mlsp.py:430-454."""
    p_vec, p_val = logits["density"], logits["density_mse"]
    ce = (target_vec * torch.log(p_vec + 1e-10)).sum(1)
    l1 = (p_val - target).abs()
    if mask is not None:
        kl = -args.Density_weight * (ce * mask).sum() / mask.sum()
        mae = args.Density_weight * (l1 * mask).sum() / mask.sum() * 0.05
    else:
        kl = -args.Density_weight * ce.mean()
        mae = args.Density_weight * l1.mean() * 0.05
    return kl, mae
