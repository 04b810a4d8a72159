"""ORACLE / CPU BASELINE -- test infrastructure, NOT the product.

The DGCNN + MLSP-heads step written with STOCK torch modules and the reference's exact operator sequence
(PointDA/Models.py:106-162, PointDA/model_utils.py:9-146): `matmul` + `topk` kNN, advanced-index gather, `repeat`, `cat`,
`nn.Conv2d` / `nn.Conv1d` 1x1 convolutions, `nn.BatchNorm2d/1d`, `LeakyReLU`, `max` over k, `adaptive_max_pool1d`, the
1536-channel head input built with `repeat` + `cat` for every head.  This is what the reference costs on a CPU (the functional
restatement oracle/ref_cpu.py uses einsum and a hand-written BatchNorm and is ~2x slower), so bench.py's `cpu_baseline` times
THIS, and tests/test_oracle_golden.py pins it to the golden vectors captured from the reference (same state_dict keys: a
fixture's parameters load with strict=True).

Only tests/ and bench.py's cpu_baseline leg import this file.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def knn_matmul_topk(x, k):
    """model_utils.py:9-16: ranking by -|xi|^2 + 2 xi.xj - |xj|^2 from one batched matmul, `topk` picks the k largest."""
    gram = torch.matmul(x.transpose(2, 1), x)
    sq = (x ** 2).sum(dim=1, keepdim=True)
    return (-sq - (-2 * gram) - sq.transpose(2, 1)).topk(k=k, dim=-1)[1]


def edge_features(x, k, knn_fn):
    """model_utils.py:18-42: [B,C,N] -> [B,2C,N,k] = cat(neighbour - centre, centre), gathered by flat row index."""
    B, C, N = x.shape
    idx = knn_fn(x, k)
    flat = (idx + torch.arange(B, device=x.device).view(B, 1, 1) * N).view(-1)
    rows = x.transpose(2, 1).contiguous()
    nbr = rows.view(B * N, C)[flat, :].view(B, N, k, C)
    ctr = rows.view(B, N, 1, C).repeat(1, 1, k, 1)
    return torch.cat((nbr - ctr, ctr), dim=3).permute(0, 3, 1, 2)


class _Block(nn.Module):
    """A parameter group registered under the attribute name the reference uses (`conv` or `fc`)."""

    def __init__(self, attr, *layers):
        super().__init__()
        setattr(self, attr, nn.Sequential(*layers))
        self._attr = attr

    def forward(self, x):
        return getattr(self, self._attr)(x)


def _conv2d_block(cin, cout, bias=False):
    return _Block("conv", nn.Conv2d(cin, cout, kernel_size=1, bias=bias), nn.BatchNorm2d(cout), nn.LeakyReLU(0.2, inplace=True))


def _fc_block(cin, cout, bias=True):
    return _Block("fc", nn.Linear(cin, cout, bias=bias), nn.BatchNorm1d(cout), nn.LeakyReLU(0.2, inplace=True))


class _TNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv2d1, self.conv2d2, self.conv2d3 = _conv2d_block(6, 64), _conv2d_block(64, 128), _conv2d_block(128, 1024)
        self.fc1, self.fc2 = _fc_block(1024, 512, bias=False), _fc_block(512, 256)
        self.fc3 = nn.Linear(256, 9)

    def forward(self, f):
        h = self.conv2d2(self.conv2d1(f)).max(dim=-1, keepdim=True)[0]
        h = self.conv2d3(h).max(dim=2, keepdim=False)[0].view(f.size(0), -1)
        h = self.fc3(self.fc2(self.fc1(h)))
        return (h + torch.eye(3, device=h.device, dtype=h.dtype).view(1, 9)).view(-1, 3, 3)


class _Classifier(nn.Module):
    def __init__(self, p, ncls):
        super().__init__()
        self.mlp1, self.mlp2, self.mlp3 = _fc_block(1024, 512), _fc_block(512, 256), nn.Linear(256, ncls)
        self.dp1, self.dp2 = nn.Dropout(p), nn.Dropout(p)

    def forward(self, x):
        return self.mlp3(self.dp2(self.mlp2(self.dp1(self.mlp1(x)))))


class _PointHead(nn.Module):
    """Models.py:165-231 (position / normal / scan heads): three Conv1d+BN1d+ReLU stages, dropout after the first two, Conv1d to 3."""

    def __init__(self, p, cin):
        super().__init__()
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm1d(256), nn.BatchNorm1d(256), nn.BatchNorm1d(128)
        self.dp1, self.dp2 = nn.Dropout(p), nn.Dropout(p)
        self.conv1, self.conv2 = nn.Conv1d(cin, 256, 1, bias=False), nn.Conv1d(256, 256, 1, bias=False)
        self.conv3, self.conv4 = nn.Conv1d(256, 128, 1, bias=False), nn.Conv1d(128, 3, 1, bias=False)

    def forward(self, x):
        x = self.dp1(F.relu(self.bn1(self.conv1(x))))
        x = self.dp2(F.relu(self.bn2(self.conv2(x))))
        x = F.relu(self.bn3(self.conv3(x)))
        return self.conv4(x).permute(0, 2, 1)


class _DensityHead(nn.Module):
    """Models.py:233-285 (cardinality head); dp1 is applied twice (:273,:278) and fc2 is the frozen expectation layer."""

    def __init__(self, p, cin, ncls, pergroup):
        super().__init__()
        self.bn1 = nn.BatchNorm1d(512)
        self.conv1 = nn.Conv1d(cin, 512, 1, bias=False)
        self.mlp1, self.mlp2, self.mlp3 = _fc_block(512, 256), _fc_block(256, 256), nn.Linear(256, ncls)
        self.dp1, self.dp2 = nn.Dropout(p), nn.Dropout(p)
        self.fc2 = nn.Linear(ncls, 1, bias=False)
        with torch.no_grad():
            self.fc2.weight.copy_(torch.arange(ncls, dtype=torch.float32).view(1, -1) * pergroup)
        self.fc2.weight.requires_grad = False

    def forward(self, x):
        x = self.dp1(F.relu(self.bn1(self.conv1(x))))
        x = x.permute(0, 2, 1).reshape(-1, 512)
        x = self.dp2(self.mlp2(self.dp1(self.mlp1(x))))
        p_vec = F.softmax(self.mlp3(x), dim=1)
        return p_vec, self.fc2(p_vec)[:, 0]


class StockDGCNN(nn.Module):
    """Same state_dict keys and forward flags as PointDA/Models.py::DGCNN.  `knn_fn(x [B,C,N], k) -> idx [B,N,k]` defaults to
    the reference's matmul + topk; tests pass the fixture's recorded indices to pin everything downstream."""

    def __init__(self, args, knn_fn=knn_matmul_topk):
        super().__init__()
        self.k, self.knn_fn = 20, knn_fn
        p = args.dropout
        self.input_transform_net = _TNet()
        self.conv1, self.conv2 = _conv2d_block(6, 64), _conv2d_block(128, 64)
        self.conv3, self.conv4 = _conv2d_block(128, 128), _conv2d_block(256, 256)
        self.bn5 = nn.BatchNorm1d(1024)
        self.conv5 = nn.Conv1d(512, 1024, kernel_size=1, bias=False)
        self.C = _Classifier(p, int(args.num_class))
        self.DefRec, self.Norm_pred, self.Rec_scan = _PointHead(p, 1536), _PointHead(p, 1536), _PointHead(p, 1536)
        self.Density_cls = _DensityHead(p, 1536, args.density_num_class, args.pergroup)

    def forward(self, x, visualization=False, activate_DefRec=False, activate_normal=False, activate_scan=False,
                activate_density=False, activate_density_normal_ondef=False):
        B, _, N = x.shape
        T = self.input_transform_net(edge_features(x, self.k, self.knn_fn))
        h = torch.matmul(T, x)
        feats = []
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4):
            h = conv(edge_features(h, self.k, self.knn_fn)).max(dim=-1, keepdim=False)[0]
            feats.append(h)
        x_cat = torch.cat(feats, dim=1)
        x5 = F.adaptive_max_pool1d(F.leaky_relu(self.bn5(self.conv5(x_cat)), negative_slope=0.2), 1).view(B, -1)
        out = {"cls": self.C(x5)}
        if visualization:
            return x5

        def head_input():       # the reference rebuilds the [B,1536,N] tensor for every head
            return torch.cat((x_cat, x5.unsqueeze(2).repeat(1, 1, N)), dim=1)
        if activate_DefRec:
            out["DefRec"] = self.DefRec(head_input())
        if activate_normal:
            out["Normal"] = self.Norm_pred(head_input())
        if activate_scan:
            out["Rec_scan"] = self.Rec_scan(head_input())
        if activate_density:
            out["density"], out["density_mse"] = self.Density_cls(head_input())
        if activate_density_normal_ondef:
            hin = head_input()
            out["DefRec"] = self.DefRec(hin)
            out["density"], out["density_mse"] = self.Density_cls(hin)
            out["Normal"] = self.Norm_pred(hin)
        return out


# --------------------------------------------------------------------------- losses, stock ops (MLSP/mlsp.py:115-182, 275-287, 430-454)
def chamfer_one_way(p1, p2, mask):
    """mlsp.py:115-153: p1, p2, mask [B,N,3].  Both point sets are expanded to dense [B,N,N,3] tensors with `repeat` before the
    subtraction, as the reference does (on a CPU this is ~6x faster than a broadcasting subtract with a 3-wide inner axis)."""
    n = p1.size(1)
    a = p1.unsqueeze(1).repeat(1, n, 1, 1).transpose(1, 2)
    b = p2.unsqueeze(1).repeat(1, n, 1, 1)
    d = torch.norm(torch.add(a, torch.neg(b)), 2, dim=3) ** 2
    m = mask[:, :, 0]
    d = d + ((m == 0).to(d.dtype) * 100.0).unsqueeze(1)
    return ((d.min(dim=2)[0] * m).sum(1) / m.sum(1)).sum()


def step_loss(args, logits, batch):
    """The trainer's target-branch loss (PointDA/trainer.py:544-565): position (x DefRec_weight x 20), masked normal, cardinality."""
    gold, mask = batch["gold"].permute(0, 2, 1), batch["mask"].permute(0, 2, 1)
    pred = logits["DefRec"]
    loss = args.DefRec_weight * 20.0 * (chamfer_one_way(gold, pred, mask) + chamfer_one_way(pred, gold, mask)) / pred.shape[0]
    w = mask[:, :, 0] * 26 + 1
    cos = (F.normalize(logits["Normal"], p=2, dim=-1) * F.normalize(batch["normal_gt"], p=2, dim=-1)).sum(-1).abs()
    loss = loss + args.normal_pred_weight * (-(cos * w).sum() / w.sum())
    wf = w.reshape(-1)
    ce = (batch["dens_vec"] * torch.log(logits["density"] + 1e-10)).sum(1)
    l1 = (logits["density_mse"] - batch["dens_val"]).abs()
    return loss - args.Density_weight * (ce * wf).sum() / wf.sum() + args.Density_weight * 0.05 * (l1 * wf).sum() / wf.sum()
