"""A PointDA/trainer.py-shaped loop (lines 341-611) on synthetic loaders, importing the model and losses through the
drop-in shim paths exactly as the trainer does (`from PointDA.Models import DGCNN`, `from MLSP import mlsp`, `import pcl`)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import golden_common as gc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trainer_shaped_loop():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(ROOT, "mlsp_amd", "shims"))
    try:
        for mod in ("PointDA", "PointDA.Models", "MLSP", "MLSP.mlsp", "MLSP.PCM", "pcl"):
            sys.modules.pop(mod, None)
        from PointDA.Models import DGCNN                     # trainer.py:14
        from MLSP import PCM, mlsp                           # trainer.py:15
        import pcl                                           # trainer.py:18
        from mlsp_amd import pc_utils
        device = torch.device("cuda:0")
        args = gc.make_args(dropout=0.5, cuda=True)
        args.radius, args.near = 0.135, 20
        args.DefRec_dist, args.mixup_params = 'volume_based_voxels', 1.0             # trainer.py:112,119
        torch.manual_seed(1)
        np.random.seed(1)
        model = DGCNN(args).to(device)                       # trainer.py:244-249
        model = nn.DataParallel(model, [0])                  # :251-252 (the trainer's only multi-GPU mechanism)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5)      # :258
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 3)              # :260
        criterion = nn.CrossEntropyLoss()
        lookup = torch.Tensor(pc_utils.region_mean(3)).to(device)                    # :270
        B, N = 8, 512       # clouds in [-0.66, 0.66]^3: the centre voxel of the 3x3x3 grid holds ~N/8 >= 40 points (deform_input's min_pts)
        g = torch.Generator().manual_seed(0)
        losses = []
        for it in range(3):
            model.train()
            opt.zero_grad()
            # ---- source branch (trainer.py:376-401): DefRec on the deformed source, then PCM mixup on the original
            src = (torch.rand(B, N, 3, generator=g) * 2 - 1) * 0.66
            src_label = torch.randint(0, 10, (B,), generator=g).to(device)
            src_data = src.to(device).permute(0, 2, 1)       # a NON-contiguous [B,3,N] view, exactly as the trainer passes it
            assert not src_data.is_contiguous()
            src_data_orig = src_data.clone()
            src_data, src_mask = mlsp.deform_input(src_data, lookup, args.DefRec_dist, device)       # :386
            assert (src_mask[:, 0].sum(1) >= 40).all() and not torch.equal(src_data, src_data_orig)
            src_logits = model(src_data, activate_DefRec=True)
            loss = mlsp.calc_loss(args, src_logits, src_data_orig, src_mask)
            loss.backward()
            src_data = src_data_orig.clone()
            src_data, mixup_vals = PCM.mix_shapes(args, src_data, src_label)                          # :396
            assert src_data.shape == (B, 3, N)
            loss = PCM.calc_loss(args, model(src_data, activate_DefRec=False), mixup_vals, criterion)
            loss.backward()
            # ---- target branch (trainer.py:522-566, Density_normal_viainput)
            trgt = ((torch.rand(B, N, 3, generator=g) * 2 - 1) * 0.66).to(device)
            normal_gt = []
            for i in range(trgt.size(0)):                    # the trainer's per-cloud pcl loop, served by the shim
                cloud = pcl.PointCloud()
                cloud.from_array(np.array(trgt[i].cpu().numpy(), dtype=np.float32))
                ne = cloud.make_NormalEstimation()
                ne.set_SearchMethod(cloud.make_kdtree())
                ne.set_KSearch(args.near)
                normal_gt.append(ne.compute().to_array()[:, :3])
            normal_gt = torch.tensor(np.array(normal_gt), dtype=torch.float32).to(device)
            dl, dml = mlsp.cal_density(trgt, radius=args.radius, num_cls=args.density_num_class, pergroup=args.pergroup)
            density_label = torch.tensor(dl, dtype=torch.float).to(device).reshape(-1, args.density_num_class)
            density_mse_label = torch.tensor(dml, dtype=torch.float).to(device).reshape(-1)
            trgt = trgt.permute(0, 2, 1)
            orig = trgt.clone()
            trgt, mask = mlsp.deform_input(trgt, lookup, args.DefRec_dist, device)                   # :541
            moved = (trgt != orig).any(1)
            assert torch.equal(moved, mask[:, 0] > 0) and (mask[:, 0].sum(1) >= 40).all()
            lp = model(trgt, activate_density_normal_ondef=True)
            loss = mlsp.calc_loss(args, lp, orig, mask)
            m2 = mask.permute(0, 2, 1)
            mask_cord = m2[:, :, 0] * 26 + 1
            npred = F.normalize(lp["Normal"], p=2, dim=-1)   # the trainer's inline masked normal loss (:551-556)
            ngt = F.normalize(normal_gt, p=2, dim=-1)
            norm_loss = -torch.sum(torch.abs(torch.sum(npred * ngt, dim=-1)) * mask_cord) / torch.sum(mask_cord)
            loss = loss + args.normal_pred_weight * norm_loss
            kl, mae = mlsp.densityloss(args, lp, density_mse_label, density_label, mask=mask_cord.reshape(-1))
            loss = loss + kl + mae
            losses.append(loss.item())
            loss.backward()
            opt.step()                                       # :571
            scheduler.step()
        assert all(np.isfinite(losses))
        # ---- test() (trainer.py:298-331)
        model.eval()
        with torch.no_grad():
            data = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(device).permute(0, 2, 1)
            lg = model(data, activate_DefRec=False)
            assert lg["cls"].max(dim=1)[1].shape == (B,)
        sd = model.module.state_dict()                       # io.save_model (utils/log.py:33-34) / strict reload (train_spst.py:141)
        m2 = DGCNN(args).to(device)
        m2.load_state_dict(sd, strict=True)
        assert int(sd["bn5.num_batches_tracked"]) == 9       # 3 training forwards per step x 3 steps
    finally:
        sys.path.remove(os.path.join(ROOT, "mlsp_amd", "shims"))
        for mod in ("PointDA", "PointDA.Models", "MLSP", "MLSP.mlsp", "MLSP.PCM", "pcl"):
            sys.modules.pop(mod, None)


def test_segda_trainer_mixup_through_shim():
    """PointSegDA/trainer.py:306: PCM.mix_shapes_segmentation through the shim -- labels follow their points."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(ROOT, "mlsp_amd", "shims"))
    try:
        for mod in ("MLSP", "MLSP.PCM"):
            sys.modules.pop(mod, None)
        from MLSP import PCM
        import types
        dev = torch.device("cuda:0")
        g = torch.Generator().manual_seed(4)
        B, N = 4, 256
        X = (torch.rand(B, 3, N, generator=g) * 2 - 1).to(dev)
        Y = (X[:, 0] * 1000).round().long()                  # a label that identifies its point
        args = types.SimpleNamespace(gpus=[0], mixup_params=1.0)
        mX, mY = PCM.mix_shapes_segmentation(args, X, Y)
        assert mX.shape == (B, 3, N) and mY.shape == (B, N)
        assert torch.equal((mX[:, 0] * 1000).round().long(), mY)
        crit = nn.CrossEntropyLoss()
        lg = torch.randn(B, 10, device=dev)
        ya, yb = torch.randint(0, 10, (B,), device=dev), torch.randint(0, 10, (B,), device=dev)
        want = 0.3 * crit(lg, ya) + 0.7 * crit(lg, yb)
        assert torch.allclose(PCM.calc_loss_ptrans(args, lg, (ya, yb, 0.3), crit), want)
    finally:
        sys.path.remove(os.path.join(ROOT, "mlsp_amd", "shims"))
        for mod in ("MLSP", "MLSP.PCM"):
            sys.modules.pop(mod, None)
