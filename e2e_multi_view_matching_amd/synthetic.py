"""Seeded synthetic image tuples in the shape the matcher consumes.

Stands in for the reference's SuperPoint front-end + dataset (``helpers.py:83-96``,
``datasets/``), which are out of scope: there are no datasets or checkpoints offline.  A
tuple = T views of one random 3-D point cloud; a fraction ``rho`` of each image's N
keypoints are noisy projections of shared points (random order per image), the rest are
uniform random pixels; descriptors are unit 256-vectors (SuperPoint descriptors are
unit-norm), shared points get a common vector plus noise.  Layout follows the reference's
data dict (App. A.3): ``keypoints{m}`` [B,N,2] pixel xy, ``scores{m}`` [B,N],
``descriptors{m}`` [B,D,N] (N contiguous), ``intr{m}``/``pose{m}`` [B,4,4].
"""
import numpy as np
import torch

IMG_W, IMG_H, FOCAL = 640, 480, 600.0


def _rand_rotation(rng, max_angle):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    a = rng.uniform(0.0, max_angle)
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * (Kx @ Kx)


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def make_tuples(batch, tuple_size=2, n_kpts=1024, desc_dim=256, seed=0, rho=0.7, noise_px=0.5,
                desc_noise=0.05, max_angle=0.3, transl_sigma=0.5, desc_dtype=torch.float32):
    """Returns the matcher input dict (torch CPU tensors) plus ground truth.

    Extra keys: ``image_size{m}`` = (H, W); ``T_{i}to{j}`` [B,4,4] = pose_j @ inv(pose_i);
    ``gt_matches{i}_{i}_{j}`` [B,N] int64 (index in image j of the same 3-D point or -1);
    ``ids`` (length T, the way the reference discovers T - ``helpers.py:84``).
    """
    rng = np.random.default_rng(seed)
    B, T, N, D = batch, tuple_size, n_kpts, desc_dim
    n_shared = int(round(rho * N))
    Kmat = np.eye(4)
    Kmat[0, 0] = Kmat[1, 1] = FOCAL
    Kmat[0, 2], Kmat[1, 2] = IMG_W / 2, IMG_H / 2
    kp = np.zeros((T, B, N, 2), np.float32)
    sc = rng.uniform(0, 1, size=(T, B, N)).astype(np.float32)
    de = np.zeros((T, B, D, N), np.float32)
    poses = np.zeros((T, B, 4, 4), np.float64)
    slot = np.full((T, B, n_shared), -1, np.int64)  # keypoint index of shared point s in image t
    for b in range(B):
        X = np.stack([rng.uniform(-2, 2, n_shared), rng.uniform(-2, 2, n_shared), rng.uniform(3, 7, n_shared)], -1)
        base = _unit(rng.normal(size=(n_shared, D)))
        for t in range(T):
            P = np.eye(4)
            if t > 0:
                P[:3, :3] = _rand_rotation(rng, max_angle)
                P[:3, 3] = rng.normal(0, transl_sigma, 3)
            poses[t, b] = P
            Xc = X @ P[:3, :3].T + P[:3, 3]
            uv = Xc[:, :2] / Xc[:, 2:3] * FOCAL + np.array([IMG_W / 2, IMG_H / 2])
            uv += rng.normal(0, noise_px, uv.shape)
            perm = rng.permutation(N)
            pts = np.concatenate([uv, np.stack([rng.uniform(0, IMG_W, N - n_shared),
                                                rng.uniform(0, IMG_H, N - n_shared)], -1)], 0)
            dsc = np.concatenate([_unit(base + rng.normal(0, desc_noise, base.shape)),
                                  _unit(rng.normal(size=(N - n_shared, D)))], 0)
            kp[t, b, perm] = pts
            de[t, b][:, perm] = dsc.T
            slot[t, b] = perm[:n_shared]
    data = {"ids": list(range(T))}
    for t in range(T):
        data[f"keypoints{t}"] = torch.from_numpy(kp[t])
        data[f"scores{t}"] = torch.from_numpy(sc[t])
        data[f"descriptors{t}"] = torch.from_numpy(de[t]).to(desc_dtype)
        data[f"image_size{t}"] = (IMG_H, IMG_W)
        data[f"intr{t}"] = torch.from_numpy(np.broadcast_to(Kmat, (B, 4, 4)).astype(np.float32).copy())
        data[f"pose{t}"] = torch.from_numpy(poses[t].astype(np.float32))
    for j in range(T):
        for i in range(j):
            Tij = poses[j] @ np.linalg.inv(poses[i])
            data[f"T_{i}to{j}"] = torch.from_numpy(Tij.astype(np.float32))
            gt = np.full((B, N), -1, np.int64)
            for b in range(B):
                gt[b, slot[i, b]] = slot[j, b]
            data[f"gt_matches{i}_{i}_{j}"] = torch.from_numpy(gt)
    return data


def identity_like_state(module):
    """Weight set "W-id" (SURVEY.md 8(d)): zero every residual branch so the matcher scores
    raw descriptor similarity - gives non-degenerate matches/poses with random-init nets.

    kenc last conv, every GNN MLP last conv -> 0; final_proj -> scale*I; bin_score kept.
    """
    sd = module.state_dict()
    with torch.no_grad():
        last = max(int(k.split(".")[2]) for k in sd if k.startswith("kenc.encoder.") and k.endswith(".weight")
                   and sd[k].dim() == 3)
        sd[f"kenc.encoder.{last}.weight"].zero_()
        sd[f"kenc.encoder.{last}.bias"].zero_()
        for k in sd:
            if k.startswith("gnn.layers.") and (k.endswith("mlp.3.weight") or k.endswith("mlp.3.bias")):
                sd[k].zero_()
        D = sd["final_proj.weight"].shape[0]
        # scores = <d0,d1> * s^2 / sqrt(D); s chosen so matched pairs (cos ~ 1) reach ~ 20
        s = (20.0 * D ** 0.5) ** 0.5
        sd["final_proj.weight"].copy_((torch.eye(D) * s).unsqueeze(-1))
        sd["final_proj.bias"].zero_()
    module.load_state_dict(sd)
    return module


def make_depth_pairs(batch, n_kpts=256, seed=0, height=IMG_H, width=IMG_W, noise_px=0.6, invalid_frac=0.05):
    """Image pairs WITH depth maps for the ground-truth-match builder (``helpers.py:121-203``): a slanted plane seen by
    two cameras (analytic depth in both views), keypoints of image 0 = random integer pixels, a fraction of them
    re-observed in image 1 (+ noise), the rest of image 1 random; some depth pixels are invalidated (0) like real sensors.
    Returns dict(keypoints0/1 [B,N,2], intr0/1, pose0/1 [B,4,4], depth0/1 [B,H,W], T_0to1)."""
    rng = np.random.default_rng(seed)
    B, N = batch, n_kpts
    K = np.eye(4)
    K[0, 0] = K[1, 1] = FOCAL
    K[0, 2], K[1, 2] = width / 2, height / 2
    Ki = np.linalg.inv(K)
    uu, vv = np.meshgrid(np.arange(width), np.arange(height))
    rays = np.stack([uu, vv, np.ones_like(uu)], -1).reshape(-1, 3) @ Ki[:3, :3].T  # z = 1 rays
    out = {k: [] for k in ("keypoints0", "keypoints1", "depth0", "depth1", "pose1")}
    for b in range(B):
        nrm = _unit(np.array([rng.normal(0, 0.15), rng.normal(0, 0.15), 1.0]))
        dist = rng.uniform(3.5, 5.5)  # plane n.X = dist in camera-0 coordinates
        P = np.eye(4)
        P[:3, :3] = _rand_rotation(rng, 0.25)
        P[:3, 3] = rng.normal(0, 0.3, 3)
        R, t = P[:3, :3], P[:3, 3]
        depth0 = (dist / (rays @ nrm)).reshape(height, width)
        # camera 1 ray: X0 = R^T (lam r - t);  n.X0 = dist  ->  lam = (dist + n.R^T t) / (n.R^T r)
        Rn = R @ nrm
        depth1 = ((dist + nrm @ (R.T @ t)) / (rays @ Rn)).reshape(height, width)
        k0 = np.stack([rng.integers(8, width - 8, N), rng.integers(8, height - 8, N)], -1).astype(np.float64)
        X0 = (np.concatenate([k0, np.ones((N, 1))], 1) @ Ki[:3, :3].T) * depth0[k0[:, 1].astype(int), k0[:, 0].astype(int)][:, None]
        X1 = X0 @ R.T + t
        proj = X1[:, :2] / X1[:, 2:3] * FOCAL + np.array([width / 2, height / 2]) + rng.normal(0, noise_px, (N, 2))
        inside = (proj[:, 0] > 8) & (proj[:, 0] < width - 8) & (proj[:, 1] > 8) & (proj[:, 1] < height - 8)
        keep = inside & (rng.uniform(size=N) < 0.7)
        k1 = np.where(keep[:, None], proj, np.stack([rng.uniform(8, width - 8, N), rng.uniform(8, height - 8, N)], -1))
        k1 = k1[rng.permutation(N)]
        for dm in (depth0, depth1):
            bad = rng.uniform(size=dm.shape) < invalid_frac
            dm[bad] = 0.0
        out["keypoints0"].append(k0 + rng.uniform(0, 0.9, (N, 2)))  # sub-pixel positions: the builder truncates them
        out["keypoints1"].append(k1)
        out["depth0"].append(depth0)
        out["depth1"].append(depth1)
        out["pose1"].append(P)
    res = {k: torch.from_numpy(np.stack(v).astype(np.float32)) for k, v in out.items()}
    res["pose0"] = torch.eye(4).unsqueeze(0).repeat(B, 1, 1)
    res["intr0"] = torch.from_numpy(np.broadcast_to(K, (B, 4, 4)).astype(np.float32).copy())
    res["intr1"] = res["intr0"].clone()
    res["T_0to1"] = res["pose1"] @ torch.linalg.inv(res["pose0"])
    return res
