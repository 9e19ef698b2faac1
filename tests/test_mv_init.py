"""Known-answer tests of the multi-view global initialisation, restated from the reference's own
gtests: pose_optimization/multi_view/bundle_adjustment/ba_init/test/test_ba_init.cpp (RotationAveraging.* :95-180,
TranslationAveraging.* :183-266, TransformationAveraging.* :268-282, BaInit.* :310-327).  Same four cameras, same
perturbation magnitudes, same tolerances; the noise comes from glibc rand() (default seed, tests in file order) exactly
like the gtest binary draws it.  Host code: runs without a GPU.
"""
import ctypes
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from e2e_multi_view_matching_amd import _lib

libc = ctypes.CDLL("libc.so.6")
libc.rand.restype = ctypes.c_int
RAND_MAX = 2147483647


def err(max_err):  # test_ba_init.cpp:10-13 (consumes one rand() even when max_err == 0)
    return libc.rand() / RAND_MAX * 2.0 * max_err - max_err


def cameras():  # CreateCameraExtrinsics :83-91: world -> camera = inverse(translation * rotation about z)
    out = []
    for c, ang in (((0, 0, 0), 0.0), ((1, 0, 0), np.pi / 4), ((1, 1, 0), np.pi / 2), ((0, 1, 0), -3 * np.pi / 4)):
        T = np.eye(4)
        T[:3, :3] = Rotation.from_rotvec([0, 0, ang]).as_matrix()
        T[:3, 3] = c
        out.append(np.linalg.inv(T))
    return out


def view_pairs(extr, max_err=0.0):  # CreateViewPairs :15-34
    ids, rots, poss = [], [], []
    for id1 in range(len(extr)):
        for id0 in range(id1):
            T = extr[id1] @ np.linalg.inv(extr[id0])
            r = Rotation.from_matrix(T[:3, :3]).as_rotvec() + np.array([err(max_err), err(max_err), err(max_err)])
            p = np.linalg.inv(T)[:3, 3] + np.array([err(max_err), err(max_err), err(max_err)])
            ids.append((id0, id1))
            rots.append(r)
            poss.append(p)
    return np.array(ids, np.int32), np.array(rots), np.array(poss)


def global_rotations(extr, max_err=0.0):  # GetGlobalRotations :36-47
    return np.array([Rotation.from_matrix(T[:3, :3]).as_rotvec() + np.array([err(max_err), err(max_err), err(max_err)])
                     for T in extr])


def p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def est_rot(ids, rots, init):
    lib = _lib.load_library()
    rot = np.ascontiguousarray(init, np.float64).copy()
    ids, rots = np.ascontiguousarray(ids), np.ascontiguousarray(rots)
    assert lib.e2emv_mv_estimate_rotations(len(rot), len(ids), p(ids), p(rots), p(rot)) == 0
    return rot


def est_pos(ids, poss, rot):
    lib = _lib.load_library()
    out = np.zeros((len(rot), 3))
    ids, poss, rot = np.ascontiguousarray(ids), np.ascontiguousarray(poss), np.ascontiguousarray(rot)
    assert lib.e2emv_mv_estimate_positions(len(rot), len(ids), p(ids), p(poss), p(rot), p(out)) == 0
    return out


def expect_rot(extr, rot, tol):  # ExpectRotationsEqual :49-65
    exp = np.array([Rotation.from_matrix(T[:3, :3]).as_rotvec() for T in extr])
    assert np.abs(exp - rot).max() < tol, np.abs(exp - rot).max()


def expect_pos(extr, pos, tol):  # ExpectTranslationsEqual :67-81
    exp = np.array([np.linalg.inv(T)[:3, 3] for T in extr])
    assert np.abs(exp - pos).max() < tol, np.abs(exp - pos).max()


def test_reference_gtest_known_answers(tmp_path):
    libc.srand(1)  # the gtest binary never seeds: glibc's default
    extr = cameras()
    # ---- RotationAveraging ----
    ids, rots, _ = view_pairs(extr)
    expect_rot(extr, est_rot(ids, rots, global_rotations(extr)), 1e-6)  # PerfectInitPerfectRel
    ids, rots, _ = view_pairs(extr)
    init = global_rotations(extr)
    k = [tuple(i) for i in ids].index((1, 2))
    rots[k] = -0.5 * rots[k]
    expect_rot(extr, est_rot(ids, rots, init), 1e-4)  # PerfectInitOutlierRel
    ids, rots, _ = view_pairs(extr, 0.05)
    expect_rot(extr, est_rot(ids, rots, global_rotations(extr)), 4e-2)  # PerfectInitNoisyRel
    ids, rots, _ = view_pairs(extr)
    init = global_rotations(extr)
    init[2] = -0.5 * init[2]
    expect_rot(extr, est_rot(ids, rots, init), 1e-6)  # OutlierInitPerfectRel
    ids, rots, _ = view_pairs(extr)
    expect_rot(extr, est_rot(ids, rots, global_rotations(extr, 0.03)), 3e-2)  # NoisyInitPerfectRel
    ids, rots, _ = view_pairs(extr, 0.02)
    expect_rot(extr, est_rot(ids, rots, global_rotations(extr, 0.03)), 3e-2)  # NoisyInitNoisyRel
    # ---- TranslationAveraging ----
    ids, _, poss = view_pairs(extr)
    expect_pos(extr, est_pos(ids, poss, global_rotations(extr)), 1e-6)  # PerfectInitPerfectRel
    ids, _, poss = view_pairs(extr)
    poss[k] = -0.5 * poss[k]
    expect_pos(extr, est_pos(ids, poss, global_rotations(extr)), 1e-4)  # PerfectInitOutlierRel
    ids, _, poss = view_pairs(extr, 0.05)
    expect_pos(extr, est_pos(ids, poss, global_rotations(extr)), 5e-2)  # PerfectInitNoisyRel
    ids, _, poss = view_pairs(extr)
    rot = global_rotations(extr)
    rot[1] = 0.9 * rot[1]
    expect_pos(extr, est_pos(ids, poss, rot), 1e-1)  # OutlierInitPerfectRel
    ids, _, poss = view_pairs(extr)
    expect_pos(extr, est_pos(ids, poss, global_rotations(extr, 0.03)), 4e-2)  # NoisyInitPerfectRel
    ids, _, poss = view_pairs(extr, 0.03)
    expect_pos(extr, est_pos(ids, poss, global_rotations(extr, 0.03)), 3e-2)  # NoisyInitNoisyRel
    # ---- TransformationAveraging.NoisyInitNoisyRel ----
    ids, rots, poss = view_pairs(extr, 0.02)
    rot = est_rot(ids, rots, global_rotations(extr, 0.03))
    expect_rot(extr, rot, 3e-2)
    expect_pos(extr, est_pos(ids, poss, rot), 3e-2)
    # ---- BaInit.PerfectInitPerfectRel: through the CSV wire format (WriteFile :284-308) ----
    ids, rots, poss = view_pairs(extr)
    init = global_rotations(extr)
    fin, fout = str(tmp_path / "ba_init_in.csv"), str(tmp_path / "ba_init_out.csv")
    with open(fin, "w") as f:
        for v, r in enumerate(init):
            R = Rotation.from_rotvec(r).as_matrix()
            f.write(",".join([str(v)] + ["%.12g" % x for x in R.T.reshape(-1)]) + "\n")
        for (i, j), r, t in zip(ids, rots, poss):
            R = Rotation.from_rotvec(r).as_matrix()
            f.write(",".join([str(i), str(j)] + ["%.12g" % x for x in R.T.reshape(-1)] + ["%.12g" % x for x in t]) + "\n")
    assert _lib.load_library().e2emv_mv_init_files(fin.encode(), fout.encode()) == 0
    rows = [[float(x) for x in line.split(",")] for line in open(fout)]
    assert len(rows) == 4 and all(len(r) == 12 for r in rows)
    for T, row in zip(extr, rows):
        R = np.array(row[:9]).reshape(3, 3).T  # column-major
        assert np.abs(R - T[:3, :3]).max() < 1e-6 and np.abs(np.array(row[9:]) - T[:3, 3]).max() < 1e-6


def test_random_scenes_recover_ground_truth():
    """Seeded scenes with 5 views (the reference's tuple size): exact relative poses + a bad chained initialisation and
    one gross outlier pair -> ground truth recovered; unit-norm baselines (what w8pt delivers) -> positions up to scale."""
    rng = np.random.default_rng(0)
    lib = _lib.load_library()
    for trial in range(5):
        n = 5
        Rw = [np.eye(3)] + [Rotation.from_rotvec(rng.normal(0, 0.4, 3)).as_matrix() for _ in range(n - 1)]
        c = [np.zeros(3)] + [rng.normal(0, 1.0, 3) for _ in range(n - 1)]
        ids, pR, pp = [], [], []
        for j in range(n):
            for i in range(j):
                Rij = Rw[j] @ Rw[i].T
                pos = Rw[i] @ (c[j] - c[i])
                ids.append((i, j))
                pR.append(Rij.T.reshape(-1))  # column-major
                pp.append(pos / np.linalg.norm(pos))
        pR, pp = np.array(pR), np.array(pp)
        if trial % 2 == 1:  # one gross outlier rotation
            pR[3] = Rotation.from_rotvec([0.9, -0.7, 0.4]).as_matrix().T.reshape(-1)
        init = np.array([(Rw[v] @ Rotation.from_rotvec(rng.normal(0, 0.05, 3)).as_matrix()).T.reshape(-1) for v in range(n)])
        init[0] = np.eye(3).reshape(-1)
        ids = np.array(ids, np.int32)
        oR, ot, st = np.zeros((n, 9)), np.zeros((n, 3)), ctypes.c_int32(0)
        assert lib.e2emv_mv_init(n, p(init), len(ids), p(ids), p(pR), p(pp), p(oR), p(ot), ctypes.byref(st)) == 0
        assert st.value == 0
        for v in range(n):
            R = oR[v].reshape(3, 3).T
            assert np.abs(R - Rw[v]).max() < (2e-3 if trial % 2 else 1e-5), (trial, v, np.abs(R - Rw[v]).max())
        if trial % 2 == 0:
            pos = np.array([-oR[v].reshape(3, 3).T.T @ ot[v] for v in range(n)])
            gt = np.array(c)
            s = np.linalg.norm(pos[1:]) / np.linalg.norm(gt[1:])
            assert np.abs(pos - s * gt).max() < 1e-3 * max(1.0, s), (trial, np.abs(pos - s * gt).max())


def test_tokenizer_and_bad_files(tmp_path):
    lib = _lib.load_library()
    assert lib.e2emv_mv_init_files(str(tmp_path / "missing.csv").encode(), str(tmp_path / "o.csv").encode()) == _lib.EINVAL
    fin = tmp_path / "in.csv"
    # a view row with an empty field (dropped by SplitByChar -> 9 fields -> ignored) leaves a gap in the ids -> EINVAL
    fin.write_text("0,1,0,0,0,1,0,0,0,1\n1,,0,0,0,1,0,0,0,1\n2,1,0,0,0,1,0,0,0,1\n")
    assert lib.e2emv_mv_init_files(str(fin).encode(), str(tmp_path / "o.csv").encode()) == _lib.EINVAL
    fin.write_text("0,1,0,0,0,1,0,0,0,1\n1,1,0,0,0,1,0,0,0,1\n")  # two views, no pairs: rotations kept, positions 0
    assert lib.e2emv_mv_init_files(str(fin).encode(), str(tmp_path / "o.csv").encode()) == 0
    rows = [line.strip().split(",") for line in open(tmp_path / "o.csv")]
    assert len(rows) == 2 and [float(x) for x in rows[1]] == [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]


def test_view_without_any_pair_keeps_its_initialisation():
    """An image that matched nothing has no row in the pair list: it must not poison the solve of the others (singular normal
    equations) - it keeps its initial rotation and the origin, the connected views are recovered as usual.  Also with the
    isolated view in the middle of the id range and with two separate components."""
    rng = np.random.default_rng(3)
    lib = _lib.load_library()
    for isolated in ([4], [2], [0]):
        n = 5
        Rw = [np.eye(3)] + [Rotation.from_rotvec(rng.normal(0, 0.4, 3)).as_matrix() for _ in range(n - 1)]
        c = [np.zeros(3)] + [rng.normal(0, 1.0, 3) for _ in range(n - 1)]
        ids, pR, pp = [], [], []
        for j in range(n):
            for i in range(j):
                if i in isolated or j in isolated:
                    continue
                ids.append((i, j))
                pR.append((Rw[j] @ Rw[i].T).T.reshape(-1))
                pos = Rw[i] @ (c[j] - c[i])
                pp.append(pos / np.linalg.norm(pos))
        init = np.array([(Rw[v] @ Rotation.from_rotvec(rng.normal(0, 0.03, 3)).as_matrix()).T.reshape(-1) for v in range(n)])
        ids, pR, pp = np.array(ids, np.int32), np.array(pR), np.array(pp)
        oR, ot, st = np.zeros((n, 9)), np.zeros((n, 3)), ctypes.c_int32(0)
        assert lib.e2emv_mv_init(n, p(init), len(ids), p(ids), p(pR), p(pp), p(oR), p(ot), ctypes.byref(st)) == 0
        assert st.value == 0 and np.isfinite(oR).all() and np.isfinite(ot).all()
        R0 = oR[0].reshape(3, 3).T
        assert np.abs(R0 - np.eye(3)).max() < 1e-9  # output frame = camera 0
        conn = [v for v in range(n) if v not in isolated]
        ref = conn[0]
        for a in conn:   # relative rotations among the connected views are exact
            Ra, Rr = oR[a].reshape(3, 3).T, oR[ref].reshape(3, 3).T
            assert np.abs(Ra @ Rr.T - Rw[a] @ Rw[ref].T).max() < 1e-5, (isolated, a)
