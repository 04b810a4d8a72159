"""Mirror of the reference's PointDA/model_utils.py op layer (lines 9-166) on the HIP hot path.

Same names, constructor arguments, forward signatures and state_dict keys as the reference, so
`from model_utils import transform_net, conv_2d, get_graph_feature, fc_layer, classifier` keeps
working.  Module boundaries speak the reference's channel-major layouts ([B,C,N], [B,C,N,k]);
inside, everything is point-major and runs in libmlsp_hip.so (mlsp_amd/functional.py).
The nn.Conv2d / nn.Linear / nn.BatchNorm children are PARAMETER HOLDERS only (identical
initialisation order and RNG consumption as the reference); they are never called.
"""
import torch
import torch.nn as nn

from . import functional as Fh

_ACT = {"relu": Fh.ACT_RELU, "leakyrelu": Fh.ACT_LRELU}


def knn(x, k):
    """model_utils.py:9-16.  x [B,C,N] -> idx [B,N,k] int64, nearest first (canonical arithmetic)."""
    B, C, N = x.shape
    xp = x.detach().transpose(2, 1).contiguous().view(B * N, C).float()
    g = Fh.knn_graph(xp, B, N, k, need_reverse=False)
    return g.idx.view(B, N, k).long()


def get_graph_feature(x, args, k=20, idx=None):
    """model_utils.py:18-42.  x [B,C,N] -> [B,2C,N,k] = cat(x_j - x_i, x_i) (a permuted view of the
    edge-major matrix, exactly as the reference returns a permuted view)."""
    B = x.size(0)
    N = x.size(2)
    x = x.view(B, -1, N)
    C = x.size(1)
    xp = x.transpose(2, 1).contiguous().view(B * N, C)
    if idx is None:
        graph = Fh.knn_graph(xp, B, N, k)
    else:
        graph = Fh.graph_from_indices(idx, B, N, k)
    F = Fh.graph_feature(xp, graph)                       # [B*N*k, 2C]
    return F.view(B, N, k, 2 * C).permute(0, 3, 1, 2)


import functools
import threading


class _FwdState(threading.local):
    """Per-THREAD bookkeeping of the forward in flight: nn.DataParallel (PointDA/trainer.py:251-252) runs one replica per
    Python thread, so the queue of num_batches_tracked counters and the nesting depth must not be shared between threads."""

    def __init__(self):
        self.pending_nbt = []
        self.depth = 0


_state = _FwdState()


def _bn_buffers(bn, training):
    """Running-stat buffers of a BatchNorm holder; in training mode its num_batches_tracked is queued for ONE
    multi-tensor increment per forward (flush_bn_counters) instead of one tiny kernel per layer."""
    if training:
        _state.pending_nbt.append(bn.num_batches_tracked)
    return bn.running_mean, bn.running_var


def flush_bn_counters():
    pend = _state.pending_nbt
    if pend:
        torch._foreach_add_(pend, 1)
        pend.clear()


def flushing_forward(fn):
    """Decorator for the public forward() of every module here: the outermost call (of this thread) flushes the queued
    num_batches_tracked increments on exit, so state_dict() is exact after any forward."""

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        _state.depth += 1
        try:
            return fn(self, *a, **k)
        finally:
            _state.depth -= 1
            if _state.depth == 0:
                flush_bn_counters()
    return wrapper


class _ZeroGradTie(torch.autograd.Function):
    """y, with parameter `b` attached to the graph at an exactly-zero gradient: no elementwise pass, and y stays the tensor the
    caller handed in (a column slice of its concatenation buffer keeps being that slice)."""

    @staticmethod
    def forward(ctx, y, b):
        ctx.bshape, ctx.bdev = b.shape, b.device
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        return g, torch.zeros(ctx.bshape, dtype=g.dtype, device=ctx.bdev)


class conv_2d(nn.Module):
    """model_utils.py:45-63: 1x1 Conv2d + BatchNorm2d + ReLU | LeakyReLU(0.2)."""

    def __init__(self, in_ch, out_ch, kernel, activation='relu', bias=True):
        super(conv_2d, self).__init__()
        assert kernel == 1 or kernel == (1, 1), "the hot path only has 1x1 convolutions"
        if activation == 'relu':
            act = nn.ReLU(inplace=True)
        elif activation == 'leakyrelu':
            act = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        else:
            raise ValueError(activation)
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, kernel_size=kernel, bias=bias), nn.BatchNorm2d(out_ch), act)
        self.act = _ACT[activation]

    def rows(self, X):
        """[rows, Cin] -> [rows, Cout]  (BN statistics over all rows)."""
        conv, bn = self.conv[0], self.conv[1]
        rm, rv = _bn_buffers(bn, self.training)
        return Fh.pointmlp(X, conv.weight.view(conv.out_channels, conv.in_channels), bias=conv.bias, gamma=bn.weight,
                           beta=bn.bias, run_mean=rm, run_var=rv, training=self.training, act=self.act, slope=0.2,
                           momentum=bn.momentum, eps=bn.eps)

    def rows_colmax(self, X, B, N, x_bounds=None):
        """[B*N, Cin] -> [B, Cout]: this conv + BN + act followed by the max over the N rows of each cloud."""
        conv, bn = self.conv[0], self.conv[1]
        if conv.bias is not None:
            return Fh.colmax(self.rows(X), B, N)
        rm, rv = _bn_buffers(bn, self.training)
        return Fh.pointmlp_colmax(X, conv.weight.view(conv.out_channels, conv.in_channels), bn.weight, bn.bias, rm, rv, B, N,
                                  training=self.training, act=self.act, slope=0.2, momentum=bn.momentum, eps=bn.eps, x_bounds=x_bounds)

    def edge(self, xp, graph, out=None, grad_accum=None, out_bounds=None, x_bounds=None):
        """Fused get_graph_feature + this conv + max over k.  xp [P,C] -> [P,Cout] (written into `out` when given: a column
        slice of the caller's concatenation buffer)."""
        conv, bn = self.conv[0], self.conv[1]
        rm, rv = _bn_buffers(bn, self.training)
        W = conv.weight.view(conv.out_channels, conv.in_channels)
        if conv.bias is None:
            return Fh.edgeconv(xp, graph, W, bn.weight, bn.bias, rm, rv, self.training, act=self.act, slope=0.2,
                               momentum=bn.momentum, eps=bn.eps, out=out, grad_accum=grad_accum, out_bounds=out_bounds, x_bounds=x_bounds)
        # conv_2d(bias=True) (model_utils.py:45-63 default): a per-channel constant in front of the BatchNorm.  Training: the batch
        # mean absorbs it (output unchanged, its gradient is exactly zero), only the running mean sees it.  Eval: it shifts the
        # normalised value by scale * bias, i.e. it is the same as evaluating with running_mean - bias.
        b = conv.bias
        if self.training:
            y = Fh.edgeconv(xp, graph, W, bn.weight, bn.bias, rm, rv, True, act=self.act, slope=0.2, momentum=bn.momentum,
                            eps=bn.eps, out=out, grad_accum=grad_accum)
            with torch.no_grad():
                rm.add_(bn.momentum * b)
            return _ZeroGradTie.apply(y, b)        # keeps bias in the graph with the zero gradient the reference gives it
        return Fh.edgeconv(xp, graph, W, bn.weight, bn.bias, rm - b.detach(), rv, False, act=self.act, slope=0.2,
                           momentum=bn.momentum, eps=bn.eps, out=out, grad_accum=grad_accum)

    @flushing_forward
    def forward(self, x):
        """Reference signature: x [B,Cin,N,k] -> [B,Cout,N,k]."""
        B, C, N, k = x.shape
        X = x.permute(0, 2, 3, 1).reshape(B * N * k, C)
        Z = self.rows(X)
        return Z.view(B, N, k, -1).permute(0, 3, 1, 2)


class fc_layer(nn.Module):
    """model_utils.py:66-87: Linear + BatchNorm1d + activation."""

    def __init__(self, in_ch, out_ch, bn=True, activation='relu', bias=True):
        super(fc_layer, self).__init__()
        if activation == 'relu':
            self.ac = nn.ReLU(inplace=True)
        elif activation == 'leakyrelu':
            self.ac = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        else:
            raise ValueError(activation)
        if bn:
            self.fc = nn.Sequential(nn.Linear(in_ch, out_ch, bias=bias), nn.BatchNorm1d(out_ch), self.ac)
        else:
            self.fc = nn.Sequential(nn.Linear(in_ch, out_ch), self.ac)
        self.has_bn = bn
        self.act = _ACT[activation]

    @flushing_forward
    def forward(self, x, p_drop=0.0, chain=False, grad_cols=None, defer=False):
        """x [M,in] -> [M,out]; `p_drop` fuses the Dropout that follows this layer in the reference; `chain`: the output only
        feeds another Linear+BN layer (it may then be stored as bf16 under functional.activation_storage("bf16")); `defer`: it only
        feeds GEMM layers (functional.pointmlp)."""
        lin = self.fc[0]
        if not self.has_bn:
            # fc_layer(bn=False) (model_utils.py:80-84): Linear + activation = the fused layer with an identity BatchNorm in eval
            # mode (scale 1, shift = the bias, eps 0), which is exact
            n = lin.out_features
            one, zero = torch.ones(n, device=x.device), torch.zeros(n, device=x.device)
            return Fh.pointmlp(x, lin.weight, gamma=one, beta=lin.bias if lin.bias is not None else zero, run_mean=zero, run_var=one,
                               training=False, act=self.act, slope=0.2, eps=0.0)
        bn = self.fc[1]
        rm, rv = _bn_buffers(bn, self.training)
        return Fh.pointmlp(x, lin.weight, bias=lin.bias, gamma=bn.weight, beta=bn.bias, run_mean=rm, run_var=rv,
                           training=self.training, act=self.act, slope=0.2, p_drop=p_drop, momentum=bn.momentum, eps=bn.eps,
                           chain=chain, grad_cols=grad_cols, defer=defer)


class transform_net(nn.Module):
    """model_utils.py:90-127: input transform net (T-Net), dgcnn branch."""

    def __init__(self, args, in_ch, out=3):
        super(transform_net, self).__init__()
        self.K = out
        self.args = args
        dg = (args.model == 'dgcnn' or getattr(args, "encoder_type", None) == 'Dgcnn_Encoder')
        activation = 'leakyrelu' if dg else 'relu'
        bias = False if dg else True
        self.conv2d1 = conv_2d(in_ch, 64, kernel=1, activation=activation, bias=bias)
        self.conv2d2 = conv_2d(64, 128, kernel=1, activation=activation, bias=bias)
        self.conv2d3 = conv_2d(128, 1024, kernel=1, activation=activation, bias=bias)
        self.fc1 = fc_layer(1024, 512, activation=activation, bias=bias, bn=True)
        self.fc2 = fc_layer(512, 256, activation=activation, bn=True)
        self.fc3 = nn.Linear(256, out * out)
        self.dg = dg

    def rows(self, F, B, N, k):
        """F: edge-major graph feature [B*N*k, in_ch] -> T [B,K,K]."""
        h = self.conv2d1.rows(F)
        h = self.conv2d2.rows(h)
        if self.dg:
            h = Fh.segmax(h, k)                                  # [B*N, 128]   (model_utils.py:114)
            rows_per_cloud = N
        else:
            rows_per_cloud = N * k
        return self._tail(h, B, rows_per_cloud)

    def points(self, xp, graph, B, N, k):
        """xp [B*N, in_ch/2] raw points + their kNN graph -> T [B,K,K].  Uses the fused LDS-resident per-edge
        kernel (tnet.hip) when the layer sizes are the DGCNN ones, else the materialised graph feature."""
        c1, b1 = self.conv2d1.conv[0], self.conv2d1.conv[1]
        c2, b2 = self.conv2d2.conv[0], self.conv2d2.conv[1]
        W1 = c1.weight.view(c1.out_channels, c1.in_channels)
        W2 = c2.weight.view(c2.out_channels, c2.in_channels)
        fused = (self.dg and c1.bias is None and c2.bias is None and W1.shape[1] == 2 * xp.shape[1] and
                 self.conv2d1.act == Fh.ACT_LRELU and Fh.tnet_edge_supported(W1, W2, k))
        if not fused:
            return self.rows(Fh.graph_feature(xp, graph), B, N, k)
        rm1, rv1 = _bn_buffers(b1, self.training)
        rm2, rv2 = _bn_buffers(b2, self.training)
        # (mode "f16x3": the stage leaves the analytic bound of its output for conv2d3's GEMMs, as the EdgeConv layers do: Models.py)
        hb = Fh.SliceBounds(torch.empty(W2.shape[0], dtype=torch.float32, device=xp.device), 0, W2.shape[0]) if Fh.gemm_precision.current == "f16x3" else None
        h = Fh.tnet_edge(xp, graph, W1, b1.weight, b1.bias, rm1, rv1, W2, b2.weight, b2.bias, rm2, rv2, self.training,
                         slope=0.2, momentum=b1.momentum, eps=b1.eps, out_bounds=hb)
        return self._tail(h, B, N, x_bounds=hb if hb is not None and hb.valid else None)

    def _tail(self, h, B, rows_per_cloud, x_bounds=None):
        h = self.conv2d3.rows_colmax(h, B, rows_per_cloud, x_bounds=x_bounds)       # conv2d3 + max over points (model_utils.py:116-117)
        h = self.fc1(h)
        h = self.fc2(h)
        h = Fh.pointmlp(h, self.fc3.weight, bias=self.fc3.bias, training=self.training)
        iden = Fh.identity_row(self.K, h.device, h.dtype)
        return (h + iden).view(B, self.K, self.K)

    @flushing_forward
    def forward(self, x):
        """Reference signature: x [B,in_ch,N,k] (graph feature) -> [B,K,K]."""
        B, C, N, k = x.shape
        F = x.permute(0, 2, 3, 1).reshape(B * N * k, C)
        return self.rows(F, B, N, k)


class classifier(nn.Module):
    """model_utils.py:129-146."""

    def __init__(self, args, num_class=10):
        super(classifier, self).__init__()
        activate = 'leakyrelu' if args.model == 'dgcnn' else 'relu'
        bias = True if args.model == 'dgcnn' else False
        self.mlp1 = fc_layer(1024, 512, bias=bias, activation=activate, bn=True)
        self.dp1 = nn.Dropout(p=args.dropout)
        self.mlp2 = fc_layer(512, 256, bias=True, activation=activate, bn=True)
        self.dp2 = nn.Dropout(p=args.dropout)
        self.mlp3 = nn.Linear(256, int(num_class))

    @flushing_forward
    def forward(self, x):
        x = self.mlp1(x, p_drop=self.dp1.p)
        x2 = self.mlp2(x, p_drop=self.dp2.p)
        return Fh.pointmlp(x2, self.mlp3.weight, bias=self.mlp3.bias, training=self.training)


class density_classifier(nn.Module):
    """model_utils.py:148-165 (unused by DGCNN: commented out at Models.py:256; kept for API parity)."""

    def __init__(self, args, num_class=10):
        super(density_classifier, self).__init__()
        activate = 'leakyrelu' if args.model == 'dgcnn' else 'relu'
        bias = True if args.model == 'dgcnn' else False
        self.mlp1 = fc_layer(512, 256, bias=bias, activation=activate, bn=True)
        self.dp1 = nn.Dropout(p=args.dropout)
        self.mlp2 = fc_layer(256, 256, bias=True, activation=activate, bn=True)
        self.dp2 = nn.Dropout(p=args.dropout)
        self.mlp3 = nn.Linear(256, int(num_class))

    @flushing_forward
    def forward(self, x):
        x = self.mlp1(x, p_drop=self.dp1.p)
        x2 = self.mlp2(x, p_drop=self.dp2.p)
        return Fh.pointmlp(x2, self.mlp3.weight, bias=self.mlp3.bias, training=self.training)
