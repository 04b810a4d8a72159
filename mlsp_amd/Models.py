"""Mirror of the reference's PointDA/Models.py:24-285 (PointNet, DGCNN + the three MLSP heads) on the HIP path.

Same constructors, forward flags, output dict keys/shapes and state_dict keys (159 entries) as the
reference, so PointDA/trainer.py and train_spst.py (strict load_state_dict) work unchanged.
Parameters live in ordinary nn.Conv*/nn.Linear/nn.BatchNorm* children created in the reference's
order (identical init under the same seed); compute runs in libmlsp_hip.so.
"""
import torch
import torch.nn as nn

from . import functional as Fh
from .model_utils import transform_net, conv_2d, fc_layer, classifier, _bn_buffers, flushing_forward

K = 20   # Models.py:13
# EdgeConv layer l+1 adding its input gradient straight into layer l's slice of the x_cat gradient (fan_out + join_columns(accs=))
# instead of autograd's three accumulation passes: measured 20-40 us SLOWER per step on MI355X (the beta = 1 output pass of the
# narrow dgrad GEMMs costs more than the element-wise adds it removes): off.
EDGE_GRAD_INTO_SLICE = False
MERGE_HEAD_TAILS = True          # merged_tails(): the heads' later layers depth by depth in shared wide matrices
MERGE_HEAD_FIRST_LAYERS = True   # merged_first_layers(): the heads' first layers as one wide layer (tests switch it off to compare)


def _bn_layer(X, conv, bn, training, act, p_drop=0.0, gbias=None, rows_per_group=0, W=None, grad_accum=None, chain=False,
              grad_cols=None, defer=False):
    """Conv1d(k=1, bias=False) + BatchNorm1d + act (+dropout) on a [rows, Cin] matrix."""
    rm, rv = _bn_buffers(bn, training)
    if W is None:
        W = conv.weight.view(conv.out_channels, conv.in_channels)
    return Fh.pointmlp(X, W, bias=conv.bias, gbias=gbias, gamma=bn.weight, beta=bn.bias, run_mean=rm, run_var=rv,
                       rows_per_group=rows_per_group, training=training, act=act, slope=0.2, p_drop=p_drop,
                       momentum=bn.momentum, eps=bn.eps, grad_accum=grad_accum, chain=chain, grad_cols=grad_cols, defer=defer)


def _first_head_layer(head, x_cat, x5, N, act, p_drop, grad_accum=None):
    """conv1 of a head on cat(x_cat, x5 repeated over N) (Models.py:156-160) without building the
    [B,1536,N] tensor: the x5 half of the weight acts on one row per cloud and enters as a per-cloud bias."""
    W = head.conv1.weight.view(head.conv1.out_channels, head.conv1.in_channels)
    Cc = x_cat.shape[1]
    Wc, W5 = Fh.split_columns(W, Cc)
    gb = Fh.pointmlp(x5, W5, training=head.training)                   # [B, Cout]
    return _bn_layer(x_cat, head.conv1, head.bn1, head.training, act, p_drop=p_drop, gbias=gb, rows_per_group=N,
                     W=Wc, grad_accum=grad_accum, chain=True)


def merged_first_layers(heads, x_cat, x5, N, grad_accum=None, x_bounds=None):
    """conv1 + bn1 + ReLU + dropout of SEVERAL heads on the same input (Models.py:156-160 feeds every head the same concatenation;
    :192, :226, :273 are the three first layers) as ONE layer of sum(of1) output channels: one wide GEMM over x_cat instead of one
    per head, one per-cloud-bias product with x5, one BatchNorm pass over all the channels (BatchNorm is per channel, so running the
    three BatchNorm1d side by side is exact), and in backward ONE K = sum(of1) dgrad and one wgrad -- x_cat and its gradient are read
    and written once per step instead of once per head.  The heads' parameters keep their identity (state_dict keys, optimizer
    entries): their storage is re-homed back to back (functional.rehome_adjacent) so that the merged operands are views.
    -> (h1 [P, sum of1], [of1 per head]): head i continues on columns sum(of1[:i]) .. of h1."""
    mods = list(heads)
    training, p_drop = mods[0].training, mods[0].dp1.p
    Cc = x_cat.shape[1]
    # (the replicas nn.DataParallel builds per forward share their buffers with the wrapped module: never re-point those)
    own = not any(getattr(m, "_is_replica", False) for m in mods)
    W = Fh.row_blocks([m.conv1.weight for m in mods], rehome=own)               # [sum of1, Cc + 1024, 1]
    W = W.view(W.shape[0], W.shape[1])
    Wc, W5 = Fh.split_columns(W, Cc)
    has_bias = mods[0].conv1.bias is not None
    bias = Fh.row_blocks([m.conv1.bias for m in mods], rehome=own) if has_bias else None
    gamma = Fh.row_blocks([m.bn1.weight for m in mods], rehome=own)
    beta = Fh.row_blocks([m.bn1.bias for m in mods], rehome=own)
    bufs = [_bn_buffers(m.bn1, training) for m in mods]
    rm, rv = Fh.merged_buffers([b[0] for b in bufs], rehome=own), Fh.merged_buffers([b[1] for b in bufs], rehome=own)
    gb = Fh.pointmlp(x5, W5, training=training)                                  # [B, sum of1]: the x5 half as a per-cloud bias
    bn = mods[0].bn1
    h1 = Fh.pointmlp(x_cat, Wc, bias=bias, gbias=gb, gamma=gamma, beta=beta, run_mean=rm.tensor, run_var=rv.tensor, rows_per_group=N,
                     training=training, act=Fh.ACT_RELU, slope=0.2, p_drop=p_drop, momentum=bn.momentum, eps=bn.eps,
                     grad_accum=grad_accum, chain=True, x_bounds=x_bounds)
    if training:
        rm.writeback()
        rv.writeback()
    return h1, [m.conv1.out_channels for m in mods]


def merged_tails(heads, h1, widths, B, N):
    """The layers after conv1 of several heads, depth by depth (csrc/multi.hip): every head has two Linear + BatchNorm + activation
    layers and a final Linear (Models.py:193-197, :227-231, :274-285), so layer d of ALL heads is one [P, sum Cout] matrix -- each
    head's Linear writes its column slice, ONE BatchNorm + activation + dropout pass (per-channel activation parameters) serves all
    of them, forward and backward.  Returns the heads' outputs in order, or None when a shape / mode is outside that kernel (the
    caller then runs the heads one by one on their slices of h1)."""
    mods = list(heads)
    training = mods[0].training
    own = not any(getattr(m, "_is_replica", False) for m in mods)
    specs = [m.tail_spec() for m in mods]            # per head: [(W, bias, bn, negative-side factor, dropout rate)] x 2
    if any(len(sp) != 2 for sp in specs):
        return None
    X, in_w = h1, list(widths)
    for d in range(2):
        layer = [sp[d] for sp in specs]
        rates = {float(l[4]) for l in layer if l[4] > 0}
        bn0 = layer[0][2]
        if len(rates) > 1 or any(l[2].momentum != bn0.momentum or l[2].eps != bn0.eps for l in layer):
            return None if d == 0 else _tails_one_by_one(mods, X, in_w, B, N, first=1)
        x_cols, o = [], 0
        for w in in_w:
            x_cols.append(o)
            o += w
        Ws = [l[0] for l in layer]
        if any(W.shape[1] != w for W, w in zip(Ws, in_w)) or not Fh.multimlp_supported(X.shape[0], X, Ws, x_cols):
            return None if d == 0 else _tails_one_by_one(mods, X, in_w, B, N, first=1)
        gamma = Fh.row_blocks([l[2].weight for l in layer], rehome=own)
        beta = Fh.row_blocks([l[2].bias for l in layer], rehome=own)
        bufs = [_bn_buffers(l[2], training) for l in layer]
        rm, rv = Fh.merged_buffers([b[0] for b in bufs], rehome=own), Fh.merged_buffers([b[1] for b in bufs], rehome=own)
        spec = tuple((W.shape[0], float(l[3]), l[4] > 0) for W, l in zip(Ws, layer))
        chan = Fh.channel_params(X.device, spec)
        # (chain: the consumers of both merged depths are GEMM layers -- the next depth, then the heads' final Linear layers)
        X = Fh.multimlp(X, [(xc, W, l[1]) for xc, W, l in zip(x_cols, Ws, layer)], gamma, beta, rm.tensor, rv.tensor, chan,
                        training=training, p_drop=(rates.pop() if rates else 0.0), momentum=bn0.momentum, eps=bn0.eps, chain=True, spec=spec)
        if training:
            rm.writeback()
            rv.writeback()
        in_w = [W.shape[0] for W in Ws]
    slices, cols = Fh.split_columns_shared(X, in_w)
    outs, col = [], 0
    for m, sl in zip(mods, slices):
        outs.append(m.tail_final(sl, B, N, grad_cols=(cols, col)))
        col += sl.shape[1]
    return outs


def _tails_one_by_one(mods, X, in_w, B, N, first):
    """fallback of merged_tails after its first merged depth: the remaining layers head by head on their column slices of X"""
    slices, cols = Fh.split_columns_shared(X, in_w)
    outs, col = [], 0
    for m, sl in zip(mods, slices):
        outs.append(m.tail_from(sl, B, N, first, grad_cols=(cols, col)))
        col += sl.shape[1]
    return outs


def can_merge_first_layers(heads):
    """distinct heads, same input width / bias setting / dropout rate / BatchNorm hyper-parameters / mode"""
    mods = list(heads)
    if not MERGE_HEAD_FIRST_LAYERS or len(mods) < 2 or len({id(m) for m in mods}) != len(mods):
        return False
    m0 = mods[0]
    return all(m.conv1.in_channels == m0.conv1.in_channels and (m.conv1.bias is None) == (m0.conv1.bias is None) and
               m.dp1.p == m0.dp1.p and m.training == m0.training and m.bn1.momentum == m0.bn1.momentum and m.bn1.eps == m0.bn1.eps and
               m.conv1.out_channels % 128 == 0 for m in mods)


class _RegionHead(nn.Module):
    """Shared body of RegionReconstruction / Normal_prediction (Models.py:165-231)."""

    def __init__(self, args, input_size):
        super().__init__()
        dropout = args if isinstance(args, float) else args.dropout
        self.of1, self.of2, self.of3 = 256, 256, 128
        self.bn1 = nn.BatchNorm1d(self.of1)
        self.bn2 = nn.BatchNorm1d(self.of2)
        self.bn3 = nn.BatchNorm1d(self.of3)
        self.dp1 = nn.Dropout(p=dropout)
        self.dp2 = nn.Dropout(p=dropout)
        self.conv1 = nn.Conv1d(input_size, self.of1, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(self.of1, self.of2, kernel_size=1, bias=False)
        self.conv3 = nn.Conv1d(self.of2, self.of3, kernel_size=1, bias=False)
        self.conv4 = nn.Conv1d(self.of3, 3, kernel_size=1, bias=False)

    def _tail(self, h, B, N, grad_cols=None):
        h = _bn_layer(h, self.conv2, self.bn2, self.training, Fh.ACT_RELU, p_drop=self.dp2.p, chain=True, grad_cols=grad_cols)
        h = _bn_layer(h, self.conv3, self.bn3, self.training, Fh.ACT_RELU, defer=True)      # (consumer: the final Linear)
        h = Fh.pointmlp(h, self.conv4.weight.view(3, self.of3), training=self.training)
        return h.view(B, N, 3)                                       # == x.permute(0,2,1) of the reference

    def rows(self, x_cat, x5, B, N, grad_accum=None):
        h = _first_head_layer(self, x_cat, x5, N, Fh.ACT_RELU, self.dp1.p, grad_accum)
        return self._tail(h, B, N)

    def rows_from_first(self, h1, B, N, grad_cols=None):
        """the layers after conv1/bn1/dp1, on this head's column slice of merged_first_layers()"""
        return self._tail(h1, B, N, grad_cols=grad_cols)

    def tail_spec(self):
        """the two Linear + BatchNorm + ReLU (+dropout) layers after conv1: (W, bias, bn, negative-side factor, dropout rate)"""
        return [(self.conv2.weight.view(self.of2, self.of1), self.conv2.bias, self.bn2, 0.0, self.dp2.p),
                (self.conv3.weight.view(self.of3, self.of2), self.conv3.bias, self.bn3, 0.0, 0.0)]

    def tail_final(self, h, B, N, grad_cols=None):
        h = Fh.pointmlp(h, self.conv4.weight.view(3, self.of3), training=self.training, grad_cols=grad_cols)
        return h.view(B, N, 3)

    def tail_from(self, h, B, N, first, grad_cols=None):
        assert first == 1
        h = _bn_layer(h, self.conv3, self.bn3, self.training, Fh.ACT_RELU, grad_cols=grad_cols, defer=True)
        return self.tail_final(h, B, N)

    @flushing_forward
    def forward(self, x):
        """Reference signature: x [B,input_size,N] -> [B,N,3]."""
        B, C, N = x.shape
        X = x.transpose(2, 1).reshape(B * N, C)
        h = _bn_layer(X, self.conv1, self.bn1, self.training, Fh.ACT_RELU, p_drop=self.dp1.p, chain=True)
        return self._tail(h, B, N)


class RegionReconstruction(_RegionHead):
    """Models.py:165-197 -- position head."""


class Normal_prediction(_RegionHead):
    """Models.py:199-231 -- normal head."""


class Density_prediction(nn.Module):
    """Models.py:233-285 -- cardinality head."""

    def __init__(self, args, input_size):
        super(Density_prediction, self).__init__()
        dropout = args if isinstance(args, float) else args.dropout
        self.of1 = 512
        self.bn1 = nn.BatchNorm1d(self.of1)
        self.dp1 = nn.Dropout(p=dropout)
        self.conv1 = nn.Conv1d(input_size, self.of1, kernel_size=1, bias=False)
        self.num_class = args.density_num_class
        activate = 'leakyrelu' if args.model == 'dgcnn' else 'relu'
        bias = True if args.model == 'dgcnn' else False
        self.mlp1 = fc_layer(512, 256, bias=bias, activation=activate, bn=True)
        self.dp1 = nn.Dropout(p=args.dropout)       # assigned twice in the reference too (Models.py:248,262)
        self.mlp2 = fc_layer(256, 256, bias=True, activation=activate, bn=True)
        self.dp2 = nn.Dropout(p=args.dropout)
        self.mlp3 = nn.Linear(256, self.num_class)
        self.fc2 = torch.nn.Linear(self.num_class, 1, bias=False)
        with torch.no_grad():
            for i in range(self.num_class):
                self.fc2.weight[0, i] = args.pergroup * i               # Models.py:267-270
        self.fc2.weight.requires_grad = False

    def _tail(self, h, grad_cols=None):
        h = self.mlp1(h, p_drop=self.dp1.p, chain=True, grad_cols=grad_cols)   # dp1 applied twice (:273,:278)
        h = self.mlp2(h, p_drop=self.dp2.p, defer=True)                        # (consumer: the final Linear)
        logits = Fh.pointmlp(h, self.mlp3.weight, bias=self.mlp3.bias, training=self.training)
        return Fh.density_tail(logits, self.fc2.weight)

    def rows(self, x_cat, x5, B, N, grad_accum=None):
        h = _first_head_layer(self, x_cat, x5, N, Fh.ACT_RELU, self.dp1.p, grad_accum)
        return self._tail(h)

    def rows_from_first(self, h1, B, N, grad_cols=None):
        return self._tail(h1, grad_cols=grad_cols)

    def tail_spec(self):
        out = []
        for mlp, dp in ((self.mlp1, self.dp1), (self.mlp2, self.dp2)):        # dp1 follows conv1 AND mlp1 (Models.py:273,278)
            lin, bn = mlp.fc[0], mlp.fc[1]
            out.append((lin.weight, lin.bias, bn, 0.2 if mlp.act == Fh.ACT_LRELU else 0.0, dp.p))
        return out

    def tail_final(self, h, B, N, grad_cols=None):
        logits = Fh.pointmlp(h, self.mlp3.weight, bias=self.mlp3.bias, training=self.training, grad_cols=grad_cols)
        return Fh.density_tail(logits, self.fc2.weight)

    def tail_from(self, h, B, N, first, grad_cols=None):
        assert first == 1
        h = self.mlp2(h, p_drop=self.dp2.p, grad_cols=grad_cols, defer=True)
        return self.tail_final(h, B, N)

    @flushing_forward
    def forward(self, x):
        """Reference signature: x [B,input_size,N] -> (p_vec [B*N,nc], density [B*N])."""
        B, C, N = x.shape
        X = x.transpose(2, 1).reshape(B * N, C)
        h = _bn_layer(X, self.conv1, self.bn1, self.training, Fh.ACT_RELU, p_drop=self.dp1.p, chain=True)
        return self._tail(h)


class PointNet(nn.Module):
    """Models.py:24-79 -- the trainer's `--model pointnet` (PointDA/trainer.py:241-244): two T-Nets, five per-point
    conv+BN+ReLU layers, global max, classifier and the DefRec head.  No neighbourhood graph: every layer is a row GEMM with
    fused BN statistics; conv5 + max never writes its [B*N,1024] activation."""

    def __init__(self, args):
        super(PointNet, self).__init__()
        num_class = int(args.num_class)
        self.args = args
        self.trans_net1 = transform_net(args, 3, 3)
        self.trans_net2 = transform_net(args, 64, 64)
        self.conv1 = conv_2d(3, 64, 1)
        self.conv2 = conv_2d(64, 64, 1)
        self.conv3 = conv_2d(64, 64, 1)
        self.conv4 = conv_2d(64, 128, 1)
        self.conv5 = conv_2d(128, 1024, 1)
        num_f_prev = 64 + 64 + 64 + 128
        self.C = classifier(args, num_class)
        self.DefRec = RegionReconstruction(args, num_f_prev + 1024)

    @flushing_forward
    def forward(self, x, activate_DefRec=False):
        B, _, N = x.shape
        logits = {}
        xp = x.float().transpose(2, 1).contiguous().view(B * N, 3)                  # point-major rows
        T1 = self.trans_net1.rows(xp, B, N, 1)                                      # [B,3,3]   (Models.py:50-54)
        h = torch.bmm(xp.view(B, N, 3), T1).view(B * N, 3)
        x1 = self.conv1.rows(h)
        x2 = self.conv2.rows(x1)
        T2 = self.trans_net2.rows(x2, B, N, 1)                                      # [B,64,64] (:59-63)
        h = torch.bmm(x2.view(B, N, 64), T2).view(B * N, 64)
        x3 = self.conv3.rows(h)
        x4 = self.conv4.rows(x3)
        x_cat = torch.cat((x1, x2, x3, x4), dim=1)                                  # [P,320]
        x5 = self.conv5.rows_colmax(x4, B, N)                                       # [B,1024]  (:68-70)
        logits["cls"] = self.C(x5)
        if activate_DefRec:
            logits["DefRec"] = self.DefRec.rows(x_cat, x5, B, N)
        return logits


class DGCNN(nn.Module):
    """Models.py:82-162."""

    def __init__(self, args):
        super(DGCNN, self).__init__()
        num_class = int(args.num_class)
        self.args = args
        self.k = K
        self.input_transform_net = transform_net(args, 6, 3)
        self.conv1 = conv_2d(6, 64, kernel=1, bias=False, activation='leakyrelu')
        self.conv2 = conv_2d(64 * 2, 64, kernel=1, bias=False, activation='leakyrelu')
        self.conv3 = conv_2d(64 * 2, 128, kernel=1, bias=False, activation='leakyrelu')
        self.conv4 = conv_2d(128 * 2, 256, kernel=1, bias=False, activation='leakyrelu')
        num_f_prev = 64 + 64 + 128 + 256
        self.bn5 = nn.BatchNorm1d(1024)
        self.conv5 = nn.Conv1d(num_f_prev, 1024, kernel_size=1, bias=False)
        self.C = classifier(args, num_class)
        self.DefRec = RegionReconstruction(args, num_f_prev + 1024)
        self.Norm_pred = Normal_prediction(args, num_f_prev + 1024)
        self.Rec_scan = RegionReconstruction(args, num_f_prev + 1024)
        self.Density_cls = Density_prediction(args, num_f_prev + 1024)

    @flushing_forward
    def forward(self, x, visualization=False, activate_DefRec=False, activate_normal=False, activate_scan=False,
                activate_density=False, activate_density_normal_ondef=False):
        B = x.size(0)
        N = x.size(2)
        k = self.k
        logits = {}
        x = x.float()
        xp0 = x.transpose(2, 1).contiguous().view(B * N, 3)                 # point-major [P,3]

        # T-Net on the graph feature of the raw cloud (Models.py:111-113)
        g0 = Fh.knn_graph(xp0, B, N, k, need_reverse=xp0.requires_grad)    # (the T-Net of a cloud without a gradient needs no reverse index)
        T = self.input_transform_net.points(xp0, g0, B, N, k)                    # [B,3,3]
        xp = Fh.apply_transform(xp0, T)                                          # (T @ x)^T, point-major

        # EdgeConv 1-4 (Models.py:115-129) write their outputs straight into the column slices of the [P,512] concatenation
        # (:131): no torch.cat pass forward, no split copies backward
        convs = (self.conv1, self.conv2, self.conv3, self.conv4)
        x_cat_buf = torch.empty((B * N, sum(c.conv[0].out_channels for c in convs)), dtype=torch.float32, device=x.device)
        # A layer's output has two consumers: its slice of x_cat and the next layer.  The next layer ADDS its input gradient into
        # that slice of x_cat's gradient (fan_out + join_columns(accs=)): no separate accumulation pass per layer in backward.
        feats, accs, h, col, acc_in = [], [], xp, 0, None
        # mode "f16x3": every layer leaves the analytic bound of its output slice (|gamma| sqrt(P k) + |beta| per channel, from its BatchNorm
        # finalizer: training mode only) in one [512] vector -- the bound of x_cat for conv5 and the heads, and of each slice for the
        # next layer, without a pass over the data (functional.SliceBounds)
        f16 = Fh.gemm_precision.current == "f16x3"
        bvec = torch.empty(x_cat_buf.shape[1], dtype=torch.float32, device=x.device) if f16 else None
        slices, hb = [], None
        for li, conv in enumerate(convs):
            g = Fh.knn_graph(h, B, N, k)
            w = conv.conv[0].out_channels
            ob = Fh.SliceBounds(bvec, col, w) if f16 else None
            out = conv.edge(h, g, out=x_cat_buf[:, col:col + w], grad_accum=acc_in, out_bounds=ob, x_bounds=hb)
            slices.append(ob)
            hb = ob
            if li + 1 < len(convs) and EDGE_GRAD_INTO_SLICE:
                (to_cat, h), acc_in = Fh.fan_out(out, 2)
            else:
                to_cat, h, acc_in = out, out, None
            feats.append(to_cat)
            accs.append(acc_in)
            col += w
        x_cat = Fh.join_columns(x_cat_buf, feats, accs)                            # [P,512]

        # conv5 and every active head read x_cat: their input gradients are summed in ONE buffer (functional.SharedInputGrad)
        heads = []
        if not visualization:
            if activate_DefRec:
                heads.append(("DefRec", self.DefRec))
            if activate_normal:
                heads.append(("Normal", self.Norm_pred))
            if activate_scan:
                heads.append(("Rec_scan", self.Rec_scan))
            if activate_density:
                heads.append(("density", self.Density_cls))
            if activate_density_normal_ondef:
                heads += [("DefRec", self.DefRec), ("density", self.Density_cls), ("Normal", self.Norm_pred)]
        # (merged layers: the position / normal / scan heads -- identical stacks -- side by side, so that their later layers can run as
        # block-diagonal GEMM launches; the order inside the merged matrices is internal, the returned dict does not depend on it)
        heads.sort(key=lambda kh: 0 if isinstance(kh[1], _RegionHead) else 1)
        merge = can_merge_first_layers([h for _, h in heads])
        aliases, acc = Fh.fan_out(x_cat, 2 if merge else 1 + len(heads))
        # conv5 and the heads' first layer read x_cat forward and backward (four GEMM operands per step): ONE measurement of its magnitude
        # for the two-piece f16 products (functional.OperandBounds; a no-op in the other product modes)
        xb = None
        if f16:      # (the layers' analytic bounds when all four were left; else measured once by the first layer that reads x_cat)
            xb = Fh.SliceBounds(bvec, 0, bvec.numel(), valid=True) if all(b.valid for b in slices) else Fh.OperandBounds(x_cat.device)
        rm5, rv5 = _bn_buffers(self.bn5, self.training)                             # conv5+bn5+LReLU+max (Models.py:132-136)
        x5 = Fh.pointmlp_colmax(aliases[0], self.conv5.weight.view(1024, -1), self.bn5.weight, self.bn5.bias, rm5, rv5, B, N,
                                training=self.training, act=Fh.ACT_LRELU, slope=0.2, momentum=self.bn5.momentum,
                                eps=self.bn5.eps, grad_accum=acc, x_bounds=xb)     # [B,1024]

        logits["cls"] = self.C(x5)
        if visualization:
            return x5
        if merge:
            # every head's first layer reads the same [x_cat | x5]: ONE wide layer; then the heads continue depth by depth in shared
            # wide matrices (merged_tails), or one by one on their column slices of h1 where that does not apply
            mods = [h for _, h in heads]
            h1, widths = merged_first_layers(mods, aliases[1], x5, N, grad_accum=acc, x_bounds=xb)
            outs = merged_tails(mods, h1, widths, B, N) if MERGE_HEAD_TAILS else None
            if outs is None:
                slices, cols = Fh.split_columns_shared(h1, widths)
                outs, col = [], 0
                for head, sl in zip(mods, slices):
                    outs.append(head.rows_from_first(sl, B, N, grad_cols=(cols, col)))
                    col += sl.shape[1]
            for (key, _), out in zip(heads, outs):
                if key == "density":
                    logits['density'], logits['density_mse'] = out
                else:
                    logits[key] = out
            return logits
        for (key, head), xa in zip(heads, aliases[1:]):
            if key == "density":
                logits['density'], logits['density_mse'] = head.rows(xa, x5, B, N, grad_accum=acc)
            else:
                logits[key] = head.rows(xa, x5, B, N, grad_accum=acc)
        return logits
