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
