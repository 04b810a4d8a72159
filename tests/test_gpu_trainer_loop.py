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
        for mod in ("PointDA", "PointDA.Models", "MLSP", "MLSP.mlsp", "pcl"):
            sys.modules.pop(mod, None)
        from PointDA.Models import DGCNN                     # trainer.py:14
        from MLSP import mlsp                                # trainer.py:15
        import pcl                                           # trainer.py:18
        device = torch.device("cuda:0")
        args = gc.make_args(dropout=0.5, cuda=True)
        args.radius, args.near = 0.135, 20
        torch.manual_seed(1)
        model = DGCNN(args).to(device)                       # trainer.py:244-249
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5)      # :258
        criterion = nn.CrossEntropyLoss()
        B, N = 8, 256
        g = torch.Generator().manual_seed(0)
        losses = []
        for it in range(3):
            model.train()
            opt.zero_grad()
            # ---- source branch (trainer.py:379-401, without PCM): CE on cls
            src = (torch.rand(B, N, 3, generator=g) * 2 - 1)
            src_label = torch.randint(0, 10, (B,), generator=g).to(device)
            logits = model(src.to(device).permute(0, 2, 1), activate_DefRec=False)
            loss = criterion(logits["cls"], src_label)
            loss.backward()
            # ---- target branch (trainer.py:522-566, Density_normal_viainput)
            trgt = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(device)
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
            mask = torch.zeros_like(trgt)
            mask[:, :, :45] = 1                              # stands in for mlsp.deform_input's region mask
            trgt = trgt + mask * 0.1 * torch.randn(trgt.shape, generator=g).to(device)
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
        assert all(np.isfinite(losses))
        # ---- test() (trainer.py:298-331)
        model.eval()
        with torch.no_grad():
            data = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(device).permute(0, 2, 1)
            lg = model(data, activate_DefRec=False)
            assert lg["cls"].max(dim=1)[1].shape == (B,)
        sd = model.state_dict()                              # io.save_model (utils/log.py:34) / strict reload (train_spst.py:141)
        m2 = DGCNN(args).to(device)
        m2.load_state_dict(sd, strict=True)
        assert int(sd["bn5.num_batches_tracked"]) == 6       # 2 training forwards per step x 3 steps
    finally:
        sys.path.remove(os.path.join(ROOT, "mlsp_amd", "shims"))
        for mod in ("PointDA", "PointDA.Models", "MLSP", "MLSP.mlsp", "pcl"):
            sys.modules.pop(mod, None)
