"""End-to-end multi-view evaluation flow (eval_multi_view.py:154-162 + eval_bundle_adjust :21-68) on a synthetic 5-tuple:
matcher (multi-frame) -> pairwise w8pt+BA -> spanning tree -> ba_init CSV -> averaging -> BA CSV -> device LM -> AUC."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_five_tuple_back_end(gpu, tmp_path):
    from e2e_multi_view_matching_amd import MultiViewMatcher, multi_view, pose_auc
    from e2e_multi_view_matching_amd.synthetic import identity_like_state, make_tuples
    T = 5
    cfg = {"GNN_layers": ["self", "cross"] * 2, "sinkhorn_iterations": 50, "multi_frame_matching": True, "tuple_size": T}
    model = identity_like_state(MultiViewMatcher(cfg).eval()).to(gpu)  # no conf_mlp, like eval_multi_view.py:130-132 (E13)
    errs_init, errs = [[], [], []], [[], [], []]
    for seed in range(3):
        data = make_tuples(batch=1, tuple_size=T, n_kpts=512, seed=20 + seed, noise_px=0.5, max_angle=0.25, transl_sigma=0.4)
        dev = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in data.items()}
        for m in range(T):  # the reference's pose{m} are camera -> world (eval_multi_view.py:56-58)
            dev[f"pose{m}"] = torch.linalg.inv(data[f"pose{m}"])
            dev[f"intr{m}"] = data[f"intr{m}"]
        with torch.no_grad():
            result = model(dev)
        d = str(tmp_path / f"t{seed}")
        multi_view.eval_bundle_adjust(T, dev, result, d, errs)
        for name in ("ba_init_in.csv", "ba_init_out.csv", "ba_in.csv", "ba_out.csv"):
            assert os.path.getsize(os.path.join(d, name)) > 0
        rows = [line.split(",") for line in open(os.path.join(d, "ba_init_in.csv"))]
        assert sum(len(r) == 10 for r in rows) == T and sum(len(r) == 14 for r in rows) == T * (T - 1) // 2
        hdr = open(os.path.join(d, "ba_in.csv")).readline().strip().split(",")
        assert int(hdr[0]) == T and int(hdr[1]) == 0 and int(hdr[3]) == 2 * int(hdr[2]) and int(hdr[2]) > 10 * 100
        # errors of the averaging stage alone, for comparison
        init = multi_view.read_bundle_adjust_result(os.path.join(d, "ba_init_out.csv"))
        assert np.abs(init[0] - np.eye(4)).max() < 1e-9  # output expressed in the frame of camera 0
        out = multi_view.read_bundle_adjust_result(os.path.join(d, "ba_out.csv"))
        assert np.abs(out[0][:3, :3] - np.eye(3)).max() < 1e-9
    e = np.array(errs[0])
    assert len(e) == 3 * 10
    auc = pose_auc(e, [5, 10, 20])
    assert e.max() < 2.0 and auc[0] > 0.8, (e, auc)  # degrees; 0.5 px noise at f = 600


def test_host_logic_matches_reference_bundle_adjust_io(gpu, tmp_path):
    """tests/golden/multi_view_io_reference.npz: ``ba_init_in.csv`` / ``ba_in.csv`` written by the reference's own
    bundle_adjust_io.py for a 4-image tuple (one weak pair that must be dropped, wrong matches with low confidence).
    Same row structure; integers exact; observations / confidences to float precision; poses within the fp32 noise of the
    two-view LM refinement; triangulated points relative to their depth."""
    from e2e_multi_view_matching_amd import multi_view
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "multi_view_io_reference.npz"))
    T = 4
    data = {k: torch.from_numpy(g[k]) for k in g.files if k.startswith(("keypoints", "intr"))}
    result = {k: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith(("matches", "conf_scores"))}
    pw = multi_view.initialize_bundle_adjust(T, data, result, str(tmp_path / "ba_init_in.csv"))
    counts = np.array([pw[f"inlier_count{i}_{j}"] for j in range(T) for i in range(j)])
    assert np.abs(counts - g["inlier_counts"]).max() <= 1, (counts, g["inlier_counts"])
    multi_view.write_bundle_adjust_problem(T, pw, g["extrinsics"], str(tmp_path / "ba_in.csv"))

    def rows(name):
        return [[float(x) for x in line.split(",")] for line in open(tmp_path / (name + ".csv"))]

    def ref_rows(name):
        out, k = [], 0
        for n in g[name + "_len"]:
            out.append(g[name + "_flat"][k:k + n])
            k += n
        return out

    mine, ref = rows("ba_init_in"), ref_rows("ba_init_in")
    assert [len(r) for r in mine] == [len(r) for r in ref]
    for a, b in zip(mine, ref):
        nid = 1 if len(a) == 10 else 2
        assert list(a[:nid]) == list(b[:nid])
        assert np.abs(np.array(a[nid:]) - b[nid:]).max() < 3e-3, (a, b)
    mine, ref = rows("ba_in"), ref_rows("ba_in")
    assert [len(r) for r in mine] == [len(r) for r in ref]
    for a, b in zip(mine, ref):
        a, b = np.array(a), np.array(b)
        if len(a) == 8:
            assert np.array_equal(a, b)
        elif len(a) == 5:  # cam, point, x, y, normalised confidence
            assert np.array_equal(a[:2], b[:2]) and np.abs(a[2:4] - b[2:4]).max() < 1e-6 and abs(a[4] - b[4]) < 1e-6 * max(1, abs(b[4]))
        elif len(a) == 12:
            assert np.abs(a - b).max() < 1e-12
        else:  # triangulated point
            assert np.abs(a - b).max() < 1e-5 * max(1.0, np.abs(b).max()), (a, b)


def test_five_tuple_with_an_unmatched_image(gpu, tmp_path):
    """One image of the tuple shares nothing with the others (all its matches are -1): its pairs yield no pose, it drops out of
    the spanning tree and of the averaging, and the rest of the tuple is still solved; every file stays well formed."""
    from e2e_multi_view_matching_amd import MultiViewMatcher, multi_view
    from e2e_multi_view_matching_amd.synthetic import identity_like_state, make_tuples
    T = 5
    cfg = {"GNN_layers": ["self", "cross"] * 2, "sinkhorn_iterations": 50, "multi_frame_matching": True, "tuple_size": T}
    model = identity_like_state(MultiViewMatcher(cfg).eval()).to(gpu)
    data = make_tuples(batch=1, tuple_size=T, n_kpts=256, seed=77, noise_px=0.5, max_angle=0.25, transl_sigma=0.4)
    g = torch.Generator().manual_seed(5)
    data["descriptors4"] = torch.nn.functional.normalize(torch.randn(1, 256, 256, generator=g), dim=1)  # unrelated content
    dev = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in data.items()}
    for m in range(T):
        dev[f"pose{m}"] = torch.linalg.inv(data[f"pose{m}"])
        dev[f"intr{m}"] = data[f"intr{m}"]
    with torch.no_grad():
        result = model(dev)
    assert all(int((result[f"matches{i}_{i}_4"] >= 0).sum()) < 8 for i in range(4))
    errs = [[], [], []]
    multi_view.eval_bundle_adjust(T, dev, result, str(tmp_path), errs)
    e = np.array(errs[0]).reshape(-1)
    assert len(e) == 10
    pairs = [(i, j) for j in range(T) for i in range(j)]
    good = np.array([e[k] for k, (i, j) in enumerate(pairs) if j != 4])
    # pairs with the unmatched image carry no information (the isolated camera stays at the origin: upstream's
    # angle_error_vec of a zero translation is 0/0); the four connected images are solved as before
    assert np.isfinite(good).all() and good.max() < 3.0, good
    init = multi_view.read_bundle_adjust_result(str(tmp_path / "ba_init_out.csv"))
    assert len(init) == T and all(np.isfinite(M).all() for M in init)
