"""ORACLE -- test infrastructure, NOT the product.

CPU restatement (plain torch fp32 ops) of the PointSegDA variant of the hot path, PointSegDA/Models.py:
`DGCNN_DefRec` (:197-242) = T-Net without BatchNorm (:105-143), `shared_layers` (:146-194: three graph stages of
bias-carrying, activation-free stacked 1x1 convs + max over k; conv6 + global max), `segmentation` (:245-269),
`DeformationReconstruction` (:272-296), `Normal_prediction` (:298-330), `Density_prediction` (:333-385).
Functional, keyed by the reference's state_dict names.  Pinned by tests/test_oracle_golden.py against
tests/golden/segda_*.npz (captured from the unmodified reference by tools/make_golden.py).
"""
import torch
import torch.nn.functional as F

from .ref_cpu import _Ctx, graph_feature, knn_reference_formula, K_DEFAULT


def round_bf16_ste(t):
    """t rounded to bf16 (RNE) in t's own dtype, gradient passed straight through: the rounding a bf16 GEMM operand / a bf16-stored
    activation sees, for the YARDSTICK of tests/test_gpu_model.py::test_segda_config4_bf16_step_vs_oracle (how far does this network move
    when its contractions read bf16 operands -- evaluated in float64, so nothing else differs from the truth)"""
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


def _q(c, t):
    return t if c.q is None else c.q(t)


def _conv2d(c, x, key):
    y = torch.einsum("oc,bcnk->bonk", _q(c, c.p[key + ".weight"][:, :, 0, 0]), _q(c, x))
    if key + ".bias" in c.p:
        y = y + c.p[key + ".bias"].view(1, -1, 1, 1)
    return y


def _conv1d(c, x, key):
    y = torch.einsum("oc,bcn->bon", _q(c, c.p[key + ".weight"][:, :, 0]), _q(c, x))
    if key + ".bias" in c.p:
        y = y + c.p[key + ".bias"].view(1, -1, 1)
    return y


def _transform_net(c, x0, prefix="input_transform_net"):
    """PointSegDA/Models.py:119-143: Conv2d(bias=False)+LeakyReLU x2, max over k, conv, max over N, fc+LeakyReLU x2, fc3 + I."""
    x = F.leaky_relu(_conv2d(c, x0, prefix + ".conv2d1.conv.0"), 0.2)
    x = F.leaky_relu(_conv2d(c, x, prefix + ".conv2d2.conv.0"), 0.2)
    x = c.max_k(x).unsqueeze(-1)                     # (max pools through the context: their arg-max is recorded, ref_cpu._Ctx.sel)
    x = F.leaky_relu(_conv2d(c, x, prefix + ".conv2d3.conv.0"), 0.2)
    x = c.max_n(x[..., 0]).reshape(x.shape[0], -1)
    x = F.leaky_relu(_q(c, x) @ _q(c, c.p[prefix + ".fc1.fc.0.weight"]).t() + c.p[prefix + ".fc1.fc.0.bias"], 0.2)
    x = F.leaky_relu(_q(c, x) @ _q(c, c.p[prefix + ".fc2.fc.0.weight"]).t() + c.p[prefix + ".fc2.fc.0.bias"], 0.2)
    x = _q(c, x) @ _q(c, c.p[prefix + ".fc3.weight"]).t() + c.p[prefix + ".fc3.bias"]
    return (x + torch.eye(3).reshape(1, 9)).view(-1, 3, 3)


def _bn_head(c, x, prefix, nout_key="conv4"):
    """segmentation / DeformationReconstruction / Normal_prediction .forward (:264-269, :291-296, :325-330)."""
    x = c.drop(F.relu(c.bn(_conv1d(c, x, prefix + ".conv1"), prefix + ".bn1", (0, 2))))
    x = c.drop(F.relu(c.bn(_conv1d(c, x, prefix + ".conv2"), prefix + ".bn2", (0, 2))))
    x = F.relu(c.bn(_conv1d(c, x, prefix + ".conv3"), prefix + ".bn3", (0, 2)))
    return _conv1d(c, x, prefix + ".conv4").permute(0, 2, 1)


def _fc_bn_lrelu(c, x, prefix):
    y = _q(c, x) @ _q(c, c.p[prefix + ".fc.0.weight"]).t()
    if prefix + ".fc.0.bias" in c.p:
        y = y + c.p[prefix + ".fc.0.bias"]
    return F.leaky_relu(c.bn(y, prefix + ".fc.1", (0,)), 0.2)


def _density_head(c, x, prefix="Density_cls"):
    """Density_prediction.forward (:372-385)."""
    x = c.drop(F.relu(c.bn(_conv1d(c, x, prefix + ".conv1"), prefix + ".bn1", (0, 2))))
    x = x.permute(0, 2, 1).reshape(-1, x.shape[1])
    x = c.drop(_fc_bn_lrelu(c, x, prefix + ".mlp1"))
    x = c.drop(_fc_bn_lrelu(c, x, prefix + ".mlp2"))
    logits = _q(c, x) @ _q(c, c.p[prefix + ".mlp3.weight"]).t() + c.p[prefix + ".mlp3.bias"]
    p_vec = F.softmax(logits, dim=1)
    return p_vec, (p_vec @ c.p[prefix + ".fc2.weight"].t())[:, 0]


def dgcnn_defrec_forward(params, x, training=True, dropout_p=0.0, k=K_DEFAULT, knn_fn=knn_reference_formula, make_seg=True,
                         activate_DefRec=True, activate_normal=False, activate_density=False,
                         activate_density_normal_ondef=False, return_ctx=False, force_sel=None, quant=None):
    """DGCNN_DefRec.forward (PointSegDA/Models.py:212-242).  Returns (logits dict, new BN buffers).
    Tests only: `force_sel` (selections recorded by another evaluation, in call order) pins every max-pool; `quant` (e.g. round_bf16_ste)
    is applied to both operands of every contraction -- the rounding emulation behind the bf16 yardstick."""
    c = _Ctx(params, training, dropout_p, knn_fn)
    c.force_sel = iter(force_sel) if force_sel is not None else None
    c.q = quant
    B, _, N = x.shape

    def graph(xin):
        idx = c.knn_fn(xin.detach(), k)
        c.knn_idx.append(idx)
        return graph_feature(xin, idx)

    T = _transform_net(c, graph(x))
    h = torch.matmul(T, x)
    sl = "shared_layers"
    x1 = c.max_k(_conv2d(c, _conv2d(c, graph(h), sl + ".conv1"), sl + ".conv2"))
    x2 = c.max_k(_conv2d(c, _conv2d(c, graph(x1), sl + ".conv3"), sl + ".conv4"))
    x3 = c.max_k(_conv2d(c, graph(x2), sl + ".conv5"))
    x123 = torch.cat((x1, x2, x3), dim=1)
    x5 = c.max_n(_conv1d(c, x123, sl + ".conv6"))
    head_in = torch.cat((x123, x5.unsqueeze(2).expand(B, x5.shape[1], N)), dim=1)
    logits = {}
    if make_seg:
        logits["seg"] = _bn_head(c, head_in, "seg")
    if activate_DefRec:
        logits["DefRec"] = _bn_head(c, head_in, "DefRec")
    if activate_normal:
        logits["Normal"] = _bn_head(c, head_in, "Norm_pred")
    if activate_density:
        logits["density"], logits["density_mse"] = _density_head(c, head_in)
    if activate_density_normal_ondef:
        logits["DefRec"] = _bn_head(c, head_in, "DefRec")
        logits["density"], logits["density_mse"] = _density_head(c, head_in)
        logits["Normal"] = _bn_head(c, head_in, "Norm_pred")
    if return_ctx:
        return logits, c.new_buffers, c
    return logits, c.new_buffers
