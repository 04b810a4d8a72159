"""Shared, reference-free helpers for golden-vector generation and the parity tests.

Both tools/make_golden.py (which imports the reference, in the build container only) and the
tests (which never see the reference) build models with `torch.manual_seed(seed)` followed by
`perturb_params`, so the two sides hold bit-identical parameters without committing an 18 MB
state_dict.  The fixture stores a checksum of every state_dict entry to prove it.
"""
import argparse
import numpy as np
import torch


def make_args(dropout=0.0, cuda=False):
    # fields read by the model: PointDA/Models.py:85,175,255,269; model_utils.py:25,98-99,133
    return argparse.Namespace(num_class=10, dropout=dropout, model="dgcnn", encoder_type=None,
                              cuda=cuda, density_num_class=16, pergroup=2.0,
                              DefRec_weight=0.5, normal_pred_weight=0.5, Density_weight=0.05,
                              Scan_Rec_weight=0.5)


def make_seg_args(dropout=0.0, gpu=False):
    # fields read by PointSegDA/Models.py (:25-28 gpus, :257-258 dropout, :353 density_num_class, :370 pergroup)
    return argparse.Namespace(gpus=[0] if gpu else [-1], dropout=dropout, density_num_class=16, pergroup=2.0,
                              DefRec_weight=0.5, normal_pred_weight=0.5, Density_weight=0.05)


@torch.no_grad()
def perturb_params(model, seed):
    """Deterministically move parameters off their init so that BN affine terms, biases and
    NEGATIVE BN scales (the min-instead-of-max branch of the fused EdgeConv) are exercised."""
    g = torch.Generator().manual_seed(1000 + seed)
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.dim() == 1 and ("bn" in name or ".1.weight" in name or ".1.bias" in name):
            # BatchNorm gamma / beta
            if name.endswith("weight"):
                p.mul_(1.0 + 0.3 * torch.randn(p.shape, generator=g))
                flip = torch.rand(p.shape, generator=g) < 0.2
                p[flip] = -p[flip]
            else:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
        else:
            p.add_(0.05 * p.abs().mean() * torch.randn(p.shape, generator=g))


def state_checksums(model):
    out = {}
    for k, v in model.state_dict().items():
        v = v.double()
        out[k] = np.array([float(v.sum()), float(v.abs().sum())])
    return out


def make_inputs(seed, B, N, num_cls=16, pergroup=2):
    """Synthetic batch shaped like the trainer's target branch (PointDA/trainer.py:522-566)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, N, generator=g) * 2 - 1                      # [B,3,N]
    gold = x + 0.05 * torch.randn(B, 3, N, generator=g)               # "original" cloud
    mask1 = torch.zeros(B, N)
    for b in range(B):
        n_m = int(torch.randint(40, 81, (1,), generator=g))
        perm = torch.randperm(N, generator=g)[:n_m]
        mask1[b, perm] = 1.0
    mask = mask1[:, None, :].repeat(1, 3, 1).contiguous()            # [B,3,N] as deform_input returns
    normal_gt = torch.randn(B, N, 3, generator=g)
    count = torch.randint(0, 31, (B * N,), generator=g).double().numpy()
    # soft label of MLSP/mlsp.py:259-263
    c1 = np.floor(count / pergroup).astype(np.int64)
    c2 = np.ceil(count / pergroup).astype(np.int64)
    eye = np.identity(num_cls)
    dens_vec = torch.tensor((eye[c1] + eye[c2]) / 2.0, dtype=torch.float32)   # [B*N,16]
    dens_val = torch.tensor(count, dtype=torch.float32)                       # [B*N]
    cls_label = torch.randint(0, 10, (B,), generator=g)
    return dict(x=x, gold=gold, mask=mask, normal_gt=normal_gt, dens_vec=dens_vec, dens_val=dens_val,
                cls_label=cls_label)


def total_loss(args, mlsp_mod, logits, inp):
    """The trainer's target-branch loss (PointDA/trainer.py:544-565, flags Normal_ondef,
    Density_ondef, Density_normal_defpart=False) written against an `mlsp`-shaped module."""
    import torch.nn.functional as F
    loss_def = mlsp_mod.calc_loss(args, logits, inp["gold"], inp["mask"])
    m = inp["mask"].permute(0, 2, 1)
    mask_cord = m[:, :, 0] * 26 + 1
    npred = F.normalize(logits["Normal"], p=2, dim=-1)
    ngt = F.normalize(inp["normal_gt"], p=2, dim=-1)
    norm_loss = -torch.sum(torch.abs(torch.sum(npred * ngt, dim=-1)) * mask_cord) / torch.sum(mask_cord)
    norm_loss = args.normal_pred_weight * norm_loss
    kl, mae = mlsp_mod.densityloss(args, logits, inp["dens_val"], inp["dens_vec"], mask=mask_cord.reshape(-1))
    parts = dict(DefRec=loss_def, normal=norm_loss, kl=kl, mae=mae)
    loss = loss_def + norm_loss + kl + mae
    if "cls" in logits:
        # source-branch criterion (PointDA/trainer.py:262 nn.CrossEntropyLoss) so that the
        # classifier's backward is covered by the same fixture
        ce = F.cross_entropy(logits["cls"], inp["cls_label"])
        parts["ce"] = ce
        loss = loss + ce
    return loss, parts
