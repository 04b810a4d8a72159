"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the input-corruption row: numpy restatement of
utils/pc_utils.py:33-73 (assign_region_to_point), MLSP/mlsp.py:10-51 (deform_input, 'volume_based_voxels') and
MLSP/PCM.py:6-38 (mix_shapes) / utils/pc_utils.py:137-161 (farthest_point_sample) with every random draw passed in.
Pinned to the reference: tests/golden/deform_*.npz and pcm_*.npz were captured by running the reference functions with
seeded / recorded random draws (tools/make_golden.py corrupt); tests/test_corrupt_oracle_cpu.py checks this file against them.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np


def assign_region(X, n=3):
    """X [B,C,N] float32 -> [B,N] int64.  Open voxel boxes with fp32 edges; points on a face keep label 0 (:60-71)."""
    d = 2 / n
    thr = np.array([-1 + i * d for i in range(n + 1)], dtype=np.float64).astype(np.float32)
    clip = np.float32(0.99999999)
    Xc = np.clip(X[:, :3].astype(np.float32), -clip, clip)
    Y = np.zeros((X.shape[0], X.shape[2]), dtype=np.int64)
    rid = 0
    for x in range(n):
        for y in range(n):
            for z in range(n):
                inside = ((thr[x] < Xc[:, 0]) & (Xc[:, 0] < thr[x + 1]) & (thr[y] < Xc[:, 1]) & (Xc[:, 1] < thr[y + 1]) &
                          (thr[z] < Xc[:, 2]) & (Xc[:, 2] < thr[z + 1]))
                Y[inside] = rid
                rid += 1
    return Y


def deform(X, lookup, region_ids, noise, groups=1, min_pts=40, n=3):
    """X [B,C,N] -> (deformed copy, mask).  noise [B,3,N] standard normal; replaced value = centre + sqrt(0.001) * noise."""
    X = X.copy()
    regions = assign_region(X, n)
    mask = np.zeros_like(X)
    sig = np.float32(np.sqrt(0.001))
    for b in range(X.shape[0]):
        it = 0
        for i in region_ids:
            ind = regions[b] == i
            if ind.sum() >= min_pts:
                it += 1
                mask[b, :3, ind] = 1
                X[b, :3, ind] = (lookup[i].astype(np.float32)[None, :] + noise[b][:, ind].T.astype(np.float32) * sig)
                if it >= groups:
                    break
    return X, mask


def fps(xyz, npoint, start):
    """utils/pc_utils.py:137-161 on [B,C,N]: indices [B,npoint] and the sampled columns [B,C,npoint]."""
    B, C, N = xyz.shape
    idx = np.zeros((B, npoint), dtype=np.int64)
    dist = np.full((B, N), 1e10, dtype=np.float32)
    far = np.asarray(start, dtype=np.int64).copy()
    rows = np.arange(B)
    for i in range(npoint):
        idx[:, i] = far
        c = xyz[rows, :, far][:, :, None]
        diff = (xyz - c).astype(np.float32)
        d = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
        dist = np.minimum(dist, d)
        far = dist.argmax(-1)
    vals = np.take_along_axis(xyz, idx[:, None, :].repeat(C, 1), 2)
    return idx, vals


def mix_shapes(X, index, lam, start_a, start_b, points_perm):
    """MLSP/PCM.py:6-38 with the draws passed in."""
    N = X.shape[2]
    na = round(lam * N)
    nb = N - round(lam * N)
    _, va = fps(X, na, start_a) if na else (None, np.zeros((X.shape[0], X.shape[1], 0), np.float32))
    _, vb = fps(X[index], nb, start_b) if nb else (None, np.zeros((X.shape[0], X.shape[1], 0), np.float32))
    return np.concatenate([va, vb], 2)[:, :, points_perm]


def rotation_matrix(angles):
    """MLSP/mlsp.py:91-112 rotate_point_cloud_3d: R = R1(y) R2(x) R3(z), float64."""
    c, s_ = np.cos(angles), np.sin(angles)
    r1 = np.array([[c[0], 0, s_[0]], [0, 1, 0], [-s_[0], 0, c[0]]])
    r2 = np.array([[1, 0, 0], [0, c[1], -s_[1]], [0, s_[1], c[1]]])
    r3 = np.array([[c[2], -s_[2], 0], [s_[2], c[2], 0], [0, 0, 1]])
    return r1 @ r2 @ r3


def scan(X, pixel_size, angles):
    """MLSP/mlsp.py:54-89 scan_input / p_scan with the random draws passed in.  X [B,N,C] -> (scanned copy, mask).
    The rotation is evaluated as ((p0*R0c + p1*R1c) + p2*R2c) in float64 (the reference calls np.dot, whose BLAS may fuse)."""
    X = X.astype(np.float32)
    out, mask = np.zeros_like(X), np.ones_like(X)
    pixel = int(2 / pixel_size)
    for b in range(X.shape[0]):
        pc = X[b].astype(np.float64)
        R = rotation_matrix(np.asarray(angles[b], dtype=np.float64))
        rot = (pc[:, 0:1] * R[0:1, :] + pc[:, 1:2] * R[1:2, :]) + pc[:, 2:3] * R[2:3, :]
        cell = ((rot[:, 2] + 1) / 2 * pixel * pixel + (rot[:, 1] + 1) / 2 * pixel).astype(np.int64)
        best = {}
        for i in range(X.shape[1]):
            c = int(cell[i])
            if c not in best or rot[i, 0] > rot[best[c], 0]:
                best[c] = i
        keep = np.array(sorted(best.values()))
        mask[b][keep, :3] = 0.0
        out[b][keep] = X[b][keep]
    return out, mask


def collapse_to_point(X, choice, noise, radius=0.5, min_pts=20):
    """deform_input(..., 'volume_based_radius') = utils/pc_utils.py:76-111 collapse_to_point per cloud, with the random draws passed in:
    `choice` [B] the picked candidate of every cloud, `noise` [B,3,N] standard normal (scaled by sqrt(0.001) like draw_from_gaussian).
    X [B,3,N] -> (deformed copy, mask [B,3,N], candidate flags [B,N]).  fp32 arithmetic in the reference's association:
    pd = (xx_j + (-2 x_i.x_j)) + xx_i, in-radius iff pd <= radius^2, candidate iff >= min_pts points in radius."""
    X = X.astype(np.float32).copy()
    B, _, N = X.shape
    mask = np.zeros_like(X)
    cand = np.zeros((B, N), bool)
    r2 = np.float32(radius ** 2)
    std = np.float32(np.sqrt(0.001))
    for b in range(B):
        x = X[b]
        xx = (x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]
        dot = (x[0][:, None] * x[0][None, :] + x[1][:, None] * x[1][None, :]) + x[2][:, None] * x[2][None, :]
        pd = (xx[None, :] + np.float32(-2.0) * dot) + xx[:, None]
        inr = pd <= r2
        cand[b] = inr.sum(1) >= min_pts
        p = int(choice[b])
        if p < 0:
            continue
        hit = inr[p]
        centre = x[:, p].copy()
        X[b][:, hit] = centre[:, None] + noise[b][:, hit].astype(np.float32) * std
        mask[b][:, hit] = 1.0
    return X, mask, cand
