"""Mirror of the reference's PointSegDA/Models.py (dense per-point segmentation variant) on the HIP path.

`DGCNN_DefRec(args, in_size=3, num_classes=8)` with the reference's constructor, forward flags, output dict and the
same 109 state_dict keys.  No new kernels: the variant's BatchNorm-free layers are run through the fused MI355X ops
with an IDENTITY BatchNorm in eval mode (scale 1, shift = the layer's bias), which is exact:
  * `shared_layers` stacks bias-carrying 1x1 convs with NO activation in between (PointSegDA/Models.py:173-184), so
    conv2(conv1(f)) = (W2 W1) f + (W2 b1 + b2) is one linear map and the folded EdgeConv kernel applies unchanged
    (max over k of a per-channel monotone map, here the identity);
  * the T-Net's Conv2d+LeakyReLU stages (:111-113) use the fused per-edge kernel, conv6 / conv2d3 + max over N use the
    Gram-matrix op, fc layers the generic MLP op.
"""
import torch
import torch.nn as nn

from . import functional as Fh
from .model_utils import _bn_buffers, flushing_forward

K = 20   # PointSegDA/Models.py:6

_ident_cache = {}


def _identity_bn(C, device):
    """(gamma=1, beta=0, running_mean=0, running_var=1) constants: with eps = 0 and eval mode the BN stage of a fused op is
    exactly the identity.  Never written (eval mode does not touch running statistics)."""
    key = (C, device.type, device.index)
    t = _ident_cache.get(key)
    if t is None:
        t = (torch.ones(C, device=device), torch.zeros(C, device=device), torch.zeros(C, device=device), torch.ones(C, device=device))
        _ident_cache[key] = t
    return t


def _linear_act(X, W, bias, act):
    """Linear (+bias) + activation without BatchNorm = identity-BN eval-mode MLP layer with beta = bias."""
    ones, zeros, rm, rv = _identity_bn(W.shape[0], X.device)
    return Fh.pointmlp(X, W, gamma=ones, beta=bias if bias is not None else zeros, run_mean=rm, run_var=rv, training=False,
                       act=act, slope=0.2, eps=0.0)


def _edge_linear_max(xp, graph, W, bias, out=None, grad_accum=None):
    """max over k of (W [x_j - x_i ; x_i] + bias): folded EdgeConv with identity BN, no activation.  `out` / `grad_accum`: as
    functional.edgeconv (the result written into a column slice of the concatenation, the input gradient added into a shared buffer)."""
    ones, zeros, rm, rv = _identity_bn(W.shape[0], xp.device)
    return Fh.edgeconv(xp, graph, W, ones, bias if bias is not None else zeros, rm, rv, False, act=Fh.ACT_NONE, eps=0.0, out=out,
                       grad_accum=grad_accum)


class conv_2d(nn.Module):
    """PointSegDA/Models.py:48-64: 1x1 Conv2d + ReLU | LeakyReLU (no BatchNorm)."""

    def __init__(self, in_ch, out_ch, kernel, activation='relu', bias=True):
        super(conv_2d, self).__init__()
        act = nn.ReLU(inplace=True) if activation == 'relu' else nn.LeakyReLU(negative_slope=0.2, inplace=True)
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, kernel_size=kernel, bias=bias), act)
        self.act = Fh.ACT_RELU if activation == 'relu' else Fh.ACT_LRELU

    def rows(self, X):
        c = self.conv[0]
        return _linear_act(X, c.weight.view(c.out_channels, c.in_channels), c.bias, self.act)

    @flushing_forward
    def forward(self, x):
        B, C, N, k = x.shape
        return self.rows(x.permute(0, 2, 3, 1).reshape(B * N * k, C)).view(B, N, k, -1).permute(0, 3, 1, 2)


class fc_layer(nn.Module):
    """PointSegDA/Models.py:66-80: Linear + LeakyReLU (the 'relu' branch of the reference never builds self.fc)."""

    def __init__(self, in_ch, out_ch, activation='relu'):
        super(fc_layer, self).__init__()
        if activation == 'relu':
            self.ac = nn.ReLU(inplace=True)
        elif activation == 'leakyrelu':
            self.ac = nn.LeakyReLU(negative_slope=0.2, inplace=True)
            self.fc = nn.Sequential(nn.Linear(in_ch, out_ch), self.ac)

    @flushing_forward
    def forward(self, x):
        lin = self.fc[0]
        return _linear_act(x, lin.weight, lin.bias, Fh.ACT_LRELU)


class fc_layer_v2(nn.Module):
    """PointSegDA/Models.py:82-103: Linear + BatchNorm1d + activation."""

    def __init__(self, in_ch, out_ch, bn=True, activation='relu', bias=True):
        super(fc_layer_v2, self).__init__()
        self.ac = nn.ReLU(inplace=True) if activation == 'relu' else nn.LeakyReLU(negative_slope=0.2, inplace=True)
        if bn:
            self.fc = nn.Sequential(nn.Linear(in_ch, out_ch, bias=bias), nn.BatchNorm1d(out_ch), self.ac)
        else:
            self.fc = nn.Sequential(nn.Linear(in_ch, out_ch), self.ac)
        self.has_bn = bn
        self.act = Fh.ACT_RELU if activation == 'relu' else Fh.ACT_LRELU

    @flushing_forward
    def forward(self, x, p_drop=0.0, chain=False):
        lin = self.fc[0]
        if not self.has_bn:
            return _linear_act(x, lin.weight, lin.bias, self.act)
        bn = self.fc[1]
        rm, rv = _bn_buffers(bn, self.training)
        return Fh.pointmlp(x, lin.weight, bias=lin.bias, gamma=bn.weight, beta=bn.bias, run_mean=rm, run_var=rv,
                           training=self.training, act=self.act, slope=0.2, p_drop=p_drop, momentum=bn.momentum, eps=bn.eps,
                           chain=chain)


class transform_net(nn.Module):
    """PointSegDA/Models.py:105-143."""

    def __init__(self, args, in_ch, out=3):
        super(transform_net, self).__init__()
        self.K = out
        self.args = args
        self.conv2d1 = conv_2d(in_ch, 64, kernel=1, activation='leakyrelu', bias=False)
        self.conv2d2 = conv_2d(64, 128, kernel=1, activation='leakyrelu', bias=False)
        self.conv2d3 = conv_2d(128, 1024, kernel=1, activation='leakyrelu', bias=False)
        self.fc1 = fc_layer(1024, 512, activation='leakyrelu')
        self.fc2 = fc_layer(512, 256, activation='leakyrelu')
        self.fc3 = nn.Linear(256, out * out)

    def points(self, xp, graph, B, N, k):
        c1, c2, c3 = self.conv2d1.conv[0], self.conv2d2.conv[0], self.conv2d3.conv[0]
        W1 = c1.weight.view(c1.out_channels, c1.in_channels)
        W2 = c2.weight.view(c2.out_channels, c2.in_channels)
        dev = xp.device
        if Fh.tnet_edge_supported(W1, W2, k) and W1.shape[1] == 2 * xp.shape[1]:
            o1, z1, m1, v1 = _identity_bn(64, dev)
            o2, z2, m2, v2 = _identity_bn(128, dev)
            h = Fh.tnet_edge(xp, graph, W1, o1, z1, m1, v1, W2, o2, z2, m2, v2, False, slope=0.2, eps=0.0)
        else:
            h = Fh.segmax(self.conv2d2.rows(self.conv2d1.rows(Fh.graph_feature(xp, graph))), k)
        o3, z3, m3, v3 = _identity_bn(c3.out_channels, dev)
        h = Fh.pointmlp_colmax(h, c3.weight.view(c3.out_channels, c3.in_channels), o3, z3, m3, v3, B, N, training=False,
                               act=Fh.ACT_LRELU, slope=0.2, eps=0.0)
        h = self.fc2(self.fc1(h))
        h = Fh.pointmlp(h, self.fc3.weight, bias=self.fc3.bias, training=self.training)
        iden = Fh.identity_row(self.K, h.device, h.dtype)
        return (h + iden).view(B, self.K, self.K)

    @flushing_forward
    def forward(self, x):
        """Reference signature: x [B,in_ch,N,k] (graph feature) -> [B,K,K]."""
        B, C, N, k = x.shape
        h = self.conv2d2.rows(self.conv2d1.rows(x.permute(0, 2, 3, 1).reshape(B * N * k, C)))
        h = Fh.colmax(self.conv2d3.rows(Fh.segmax(h, k)), B, N)
        h = self.fc2(self.fc1(h))
        h = Fh.pointmlp(h, self.fc3.weight, bias=self.fc3.bias, training=self.training)
        iden = Fh.identity_row(self.K, h.device, h.dtype)
        return (h + iden).view(B, self.K, self.K)


class shared_layers(nn.Module):
    """PointSegDA/Models.py:146-194."""

    def __init__(self, args, in_size=3):
        super(shared_layers, self).__init__()
        self.args = args
        self.k = K
        self.of1 = self.of2 = self.of3 = self.of4 = self.of5 = 64
        self.of6 = 1024
        self.conv1 = nn.Conv2d(in_size * 2, self.of1, kernel_size=1, bias=True)
        self.conv2 = nn.Conv2d(self.of1, self.of2, kernel_size=1, bias=True)
        self.conv3 = nn.Conv2d(self.of2 * 2, self.of3, kernel_size=1, bias=True)
        self.conv4 = nn.Conv2d(self.of3, self.of4, kernel_size=1, bias=True)
        self.conv5 = nn.Conv2d(self.of4 * 2, self.of5, kernel_size=1, bias=True)
        num_f_prev = self.of1 + self.of3 + self.of5
        self.conv6 = nn.Conv1d(num_f_prev, self.of6, kernel_size=1, bias=True)

    @staticmethod
    def _compose(ca, cb):
        """conv_b(conv_a(f)) with no activation in between == one linear map (functional.compose_linear: one launch forward, one
        backward; the gradient of the composite goes back to both layers)."""
        Wa = ca.weight.view(ca.out_channels, ca.in_channels)
        Wb = cb.weight.view(cb.out_channels, cb.in_channels)
        return Fh.compose_linear(Wa, ca.bias, Wb, cb.bias)

    def points(self, xp, B, N, consumers=0):
        """-> (x123 [P,192], x5 [B,1024]); consumers = n > 0: -> ([n aliases of x123], x5, SharedInputGrad) for n heads that take
        `grad_accum=` -- conv6 and the heads then sum their input gradients in ONE buffer through the beta = 1 epilogue of their dgrad
        instead of n element-wise adds of the autograd engine (as mlsp_amd/Models.py does for the PointDA model)."""
        # the three EdgeConv stages write their outputs straight into the column slices of the [P,192] concatenation (Models.py:186): no
        # torch.cat pass forward, no split copies backward; stage l+1 adds its input gradient into stage l's slice of that gradient
        stages = ((self.conv1, self.conv2, self.of1), (self.conv3, self.conv4, self.of3), (self.conv5, None, self.of5))
        buf = torch.empty((xp.shape[0], self.of1 + self.of3 + self.of5), dtype=torch.float32, device=xp.device)
        feats, accs, h, col, acc_in = [], [], xp, 0, None
        for li, (ca, cb, w) in enumerate(stages):
            g = Fh.knn_graph(h, B, N, self.k)
            W, b = self._compose(ca, cb) if cb is not None else (ca.weight.view(w, -1), ca.bias)
            out = _edge_linear_max(h, g, W, b, out=buf[:, col:col + w], grad_accum=acc_in)
            if li + 1 < len(stages):
                (to_cat, h), acc_in = Fh.fan_out(out, 2)
            else:
                to_cat, acc_in = out, None
            feats.append(to_cat)
            accs.append(acc_in)
            col += w
        x123 = Fh.join_columns(buf, feats, accs)                                   # [P,192]
        ones, zeros, rm, rv = _identity_bn(self.of6, xp.device)
        acc = None
        if consumers > 0:
            aliases, acc = Fh.fan_out(x123, 1 + consumers)
            x123 = aliases[0]
        x5 = Fh.pointmlp_colmax(x123, self.conv6.weight.view(self.of6, -1), ones, self.conv6.bias, rm, rv, B, N, training=False,
                                act=Fh.ACT_NONE, eps=0.0, grad_accum=acc)          # conv6 + adaptive_max_pool1d
        if consumers > 0:
            return list(aliases[1:]), x5, acc
        return x123, x5

    def layers_sum(self):
        return self.of1 + self.of3 + self.of5

    @flushing_forward
    def forward(self, x):
        """Reference signature: x [B,in,N] -> (x123 [B,192,N], x5 [B,1024,1])."""
        B, C, N = x.shape
        x123, x5 = self.points(x.transpose(2, 1).contiguous().view(B * N, C), B, N)
        return x123.view(B, N, -1).permute(0, 2, 1), x5.unsqueeze(2)


def _head_bn_layer(X, conv, bn, training, p_drop=0.0, gbias=None, rows_per_group=0, W=None, chain=False, grad_accum=None):
    rm, rv = _bn_buffers(bn, training)
    if W is None:
        W = conv.weight.view(conv.out_channels, conv.in_channels)
    return Fh.pointmlp(X, W, bias=conv.bias, gbias=gbias, gamma=bn.weight, beta=bn.bias, run_mean=rm, run_var=rv,
                       rows_per_group=rows_per_group, training=training, act=Fh.ACT_RELU, slope=0.2, p_drop=p_drop,
                       momentum=bn.momentum, eps=bn.eps, chain=chain, grad_accum=grad_accum)


def _first_layer(head, x_cat, x5, N, p_drop, grad_accum=None):
    """conv1 on cat(x123, x5 repeated): the x5 half enters as a per-cloud bias (see mlsp_amd/Models.py).  grad_accum = (SharedInputGrad
    of x_cat's fan_out, SharedInputGrad of x5's) or None."""
    W = head.conv1.weight.view(head.conv1.out_channels, head.conv1.in_channels)
    Cc = x_cat.shape[1]
    Wc, W5 = Fh.split_columns(W, Cc)
    acc_cat, acc5 = grad_accum if grad_accum is not None else (None, None)
    gb = Fh.pointmlp(x5, W5, training=head.training, grad_accum=acc5)
    return _head_bn_layer(x_cat, head.conv1, head.bn1, head.training, p_drop=p_drop, gbias=gb, rows_per_group=N, W=Wc, chain=True,
                          grad_accum=acc_cat)


class _BnHead(nn.Module):
    """Body shared by segmentation / DeformationReconstruction / Normal_prediction (PointSegDA/Models.py:245-330)."""

    def __init__(self, dropout, input_size, out_size, bias):
        super().__init__()
        self.of1, self.of2, self.of3 = 256, 256, 128
        self.bn1 = nn.BatchNorm1d(self.of1)
        self.bn2 = nn.BatchNorm1d(self.of2)
        self.bn3 = nn.BatchNorm1d(self.of3)
        self.dp1 = nn.Dropout(p=dropout)
        self.dp2 = nn.Dropout(p=dropout)
        self.conv1 = nn.Conv1d(input_size, self.of1, kernel_size=1, bias=bias)
        self.conv2 = nn.Conv1d(self.of1, self.of2, kernel_size=1, bias=bias)
        self.conv3 = nn.Conv1d(self.of2, self.of3, kernel_size=1, bias=bias)
        self.conv4 = nn.Conv1d(self.of3, out_size, kernel_size=1, bias=bias)

    def _tail(self, h, B, N):
        h = _head_bn_layer(h, self.conv2, self.bn2, self.training, p_drop=self.dp2.p, chain=True)
        h = _head_bn_layer(h, self.conv3, self.bn3, self.training)
        h = Fh.pointmlp(h, self.conv4.weight.view(self.conv4.out_channels, self.of3), bias=self.conv4.bias, training=self.training)
        return h.view(B, N, -1)

    def rows(self, x_cat, x5, B, N, grad_accum=None):
        return self._tail(_first_layer(self, x_cat, x5, N, self.dp1.p, grad_accum), B, N)

    @flushing_forward
    def forward(self, x):
        B, C, N = x.shape
        h = _head_bn_layer(x.transpose(2, 1).reshape(B * N, C), self.conv1, self.bn1, self.training, p_drop=self.dp1.p, chain=True)
        return self._tail(h, B, N)


class segmentation(_BnHead):
    """PointSegDA/Models.py:245-269."""

    def __init__(self, args, input_size, num_classes=8):
        super().__init__(args.dropout, input_size, num_classes, True)
        self.args = args


class DeformationReconstruction(_BnHead):
    """PointSegDA/Models.py:272-296."""

    def __init__(self, args, input_size, out_size=3):
        super().__init__(args.dropout, input_size, out_size, True)
        self.args = args


class Normal_prediction(_BnHead):
    """PointSegDA/Models.py:298-330."""

    def __init__(self, args, input_size):
        super().__init__(args if isinstance(args, float) else args.dropout, input_size, 3, False)


class Density_prediction(nn.Module):
    """PointSegDA/Models.py:333-385."""

    def __init__(self, args, input_size):
        super(Density_prediction, self).__init__()
        dropout = args if isinstance(args, float) else args.dropout
        self.of1 = 512
        self.bn1 = nn.BatchNorm1d(self.of1)
        self.dp1 = nn.Dropout(p=dropout)
        self.conv1 = nn.Conv1d(input_size, self.of1, kernel_size=1, bias=False)
        self.num_class = args.density_num_class
        self.mlp1 = fc_layer_v2(512, 256, bias=True, activation='leakyrelu', bn=True)
        self.dp1 = nn.Dropout(p=args.dropout)
        self.mlp2 = fc_layer_v2(256, 256, bias=True, activation='leakyrelu', bn=True)
        self.dp2 = nn.Dropout(p=args.dropout)
        self.mlp3 = nn.Linear(256, self.num_class)
        self.fc2 = torch.nn.Linear(self.num_class, 1, bias=False)
        with torch.no_grad():
            for i in range(self.num_class):
                self.fc2.weight[0, i] = args.pergroup * i
        self.fc2.weight.requires_grad = False

    def _tail(self, h):
        h = self.mlp1(h, p_drop=self.dp1.p, chain=True)
        h = self.mlp2(h, p_drop=self.dp2.p)
        logits = Fh.pointmlp(h, self.mlp3.weight, bias=self.mlp3.bias, training=self.training)
        return Fh.density_tail(logits, self.fc2.weight)

    def rows(self, x_cat, x5, B, N, grad_accum=None):
        return self._tail(_first_layer(self, x_cat, x5, N, self.dp1.p, grad_accum))

    @flushing_forward
    def forward(self, x):
        B, C, N = x.shape
        h = _head_bn_layer(x.transpose(2, 1).reshape(B * N, C), self.conv1, self.bn1, self.training, p_drop=self.dp1.p, chain=True)
        return self._tail(h)


class DGCNN_DefRec(nn.Module):
    """PointSegDA/Models.py:197-242."""

    def __init__(self, args, in_size=3, num_classes=8):
        super(DGCNN_DefRec, self).__init__()
        self.args = args
        self.k = K
        self.input_transform_net = transform_net(args, in_size * 2, in_size)
        self.shared_layers = shared_layers(args, in_size=in_size)
        self.num_f_prev = self.shared_layers.layers_sum()
        self.seg = segmentation(args, input_size=1024 + self.num_f_prev, num_classes=num_classes)
        self.DefRec = DeformationReconstruction(args, 1024 + self.num_f_prev, out_size=in_size)
        self.Norm_pred = Normal_prediction(args, self.num_f_prev + 1024)
        self.Density_cls = Density_prediction(args, self.num_f_prev + 1024)

    @flushing_forward
    def forward(self, x, make_seg=True, activate_DefRec=True, activate_normal=False, activate_density=False,
                activate_density_normal_ondef=False):
        B, C, N = x.shape
        logits = {}
        x = x.float()
        xp0 = x.transpose(2, 1).contiguous().view(B * N, C)
        g0 = Fh.knn_graph(xp0, B, N, self.k, need_reverse=xp0.requires_grad or C > 4)   # (the T-Net of a 3-channel cloud without a gradient needs no reverse index; wider inputs take the fold path, which does)
        T = self.input_transform_net.points(xp0, g0, B, N, self.k)                 # [B,3,3]
        xp = Fh.apply_transform(xp0, T) if C == 3 else torch.bmm(xp0.view(B, N, C), T.transpose(1, 2)).view(B * N, C)   # (T @ x)^T
        # every head reads x123 and x5: the calls below are the reference's (Models.py:226-241, DefRec computed twice when both of its flags
        # are set, the first result overwritten); each gets its own alias of the two inputs and all of them sum their input gradients in one
        # buffer per input (functional.fan_out / SharedInputGrad) instead of one element-wise add per head in backward
        calls = []
        if make_seg:
            calls.append(("seg", self.seg))
        if activate_DefRec:
            calls.append(("DefRec", self.DefRec))
        if activate_normal:
            calls.append(("Normal", self.Norm_pred))
        if activate_density:
            calls.append(("density", self.Density_cls))
        if activate_density_normal_ondef:
            calls += [("DefRec", self.DefRec), ("density", self.Density_cls), ("Normal", self.Norm_pred)]
        if not calls:
            self.shared_layers.points(xp, B, N)
            return logits
        xs, x5, acc = self.shared_layers.points(xp, B, N, consumers=len(calls))
        x5s, acc5 = Fh.fan_out(x5, len(calls))
        for (key, head), xa, x5a in zip(calls, xs, x5s):
            out = head.rows(xa, x5a, B, N, grad_accum=(acc, acc5))
            if key == "density":
                logits['density'], logits['density_mse'] = out
            else:
                logits[key] = out
        return logits
