"""Drop-in for ``pose_optimization/multi_view/bundle_adjust_io.py`` and the ``eval_bundle_adjust`` routine of
``eval_multi_view.py:21-68``: pairwise poses -> spanning-tree initialisation -> global rotation /
position averaging -> weighted bundle adjustment, keeping the reference's CSV wire format (``ba_init_in/out.csv``,
``ba_in/out.csv``).

Same function names, arguments, dictionary keys and file layouts as the reference.  What differs is where the work runs:
* relative poses: the HIP w8pt + two-view BA kernels (``pose.py``) instead of kornia/pytorch3d ops;
* ``ba_initializer`` / ``bundle_adjuster``: not separate executables built on Theia/Ceres but entry points of
  libe2emv.so called in-process (``run_ba_initializer`` = host C++ averaging, ``run_bundle_adjuster`` = one HIP workgroup
  doing the whole LM/Schur optimisation); ``python -m e2e_multi_view_matching_amd.multi_view ba_initializer <dir>`` and
  ``... bundle_adjuster <dir>`` give the reference's command-line shape;
* triangulation: ``cv2.triangulatePoints`` (OpenCV is absent) -> ``e2emv_mv_triangulate`` (same homogeneous DLT).
Host glue (dict plumbing, spanning tree via scipy like the reference, CSV text) stays in Python like the reference's.
No CPU fallback for the device parts.
"""
import ctypes
import logging
import os
import sys

import numpy as np
import torch
from scipy.sparse.csgraph import minimum_spanning_tree

from . import _lib
from .metrics import compute_pose_error
from .pose import estimate_relative_pose_w8pt, run_bundle_adjust_2_view


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("the multi-view back-end needs an MI355X (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def estimate_relative_pose_w8pt_ba(intr0, intr1, mkpts0, mkpts1, conf):
    """``estimate_relative_pose_w8pt_ba`` (bundle_adjust_io.py:12-23): numpy in, ``(success, R, t, inliers)`` out."""
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).unsqueeze(0)  # noqa: E731
    pred_T021, info = estimate_relative_pose_w8pt(t(mkpts0), t(mkpts1), t(intr0), t(intr1), t(conf), determine_inliers=True)
    if pred_T021 is None:
        return False, None, None, None
    confidence = info["confidence"]
    confidence[torch.logical_not(info["pos_depth_mask"]).reshape(confidence.shape)] = 0.
    pred_T021_refine, valid_refine = run_bundle_adjust_2_view(info["kpts0_norm"], info["kpts1_norm"], confidence, pred_T021,
                                                              n_iterations=10)
    pred_T021[valid_refine] = pred_T021_refine
    return True, pred_T021[0, :3, :3].cpu().numpy(), pred_T021[0, :3, 3].cpu().numpy(), info["inliers"].squeeze(0).cpu().numpy()


def normalize_confidences(obs_xyc):
    """bundle_adjust_io.py:56-60."""
    conf = obs_xyc[:, 2:]
    sum_conf = conf.sum(axis=0, keepdims=True) + 1e-3
    obs_xyc[:, 2:] = conf / (0.5 * sum_conf)  # 0.5 because each match leads to 2 observations
    return obs_xyc


def initialize_bundle_adjust(n_images, data, result, file_path, conf_thresh=0., rel_pose_method="w8pt_ba"):
    """``initialize_bundle_adjust`` (bundle_adjust_io.py:62-191): collects the matches of batch element 0, estimates all
    pairwise poses, chains them along the maximum spanning tree of the inlier-count graph and writes ``ba_init_in.csv``."""
    if rel_pose_method != "w8pt_ba":
        # the "ransac" / "ransac_ba" variants call OpenCV's findEssentialMat (absent submodule + absent cv2): out of scope
        raise NotImplementedError("relative pose method {} needs OpenCV RANSAC, which is outside this back-end".format(rel_pose_method))
    min_inliers = 20
    pair_wise_data = dict()
    match_graph = np.zeros((n_images, n_images), dtype=int)
    for id1 in range(n_images):
        for id0 in range(id1):
            matches_key = "matches{}_{}_{}".format(id0, id0, id1)
            if matches_key not in result:
                continue
            if "keypoints" + str(id0) in data:
                kpts0, kpts1 = data["keypoints" + str(id0)][0].cpu().numpy(), data["keypoints" + str(id1)][0].cpu().numpy()
            else:
                kpts0 = data["keypoints{}_{}_{}".format(id0, id0, id1)][0].cpu().numpy()
                kpts1 = data["keypoints{}_{}_{}".format(id1, id0, id1)][0].cpu().numpy()
            matches = result[matches_key][0].cpu().numpy()
            intr0, intr1 = data["intr" + str(id0)][0].cpu().numpy(), data["intr" + str(id1)][0].cpu().numpy()
            confidence = result["conf_scores_{}_{}".format(id0, id1)][0].cpu().numpy()
            valid = (matches >= 0) & np.all(confidence > conf_thresh, -1)
            pair_wise_data["mkpts{}_{}_{}".format(id0, id0, id1)] = kpts0[valid]
            pair_wise_data["mkpts{}_{}_{}".format(id1, id0, id1)] = kpts1[matches[valid]]
            confidence = confidence[valid]
            pair_wise_data["conf{}_{}_{}".format(id0, id0, id1)] = confidence
            pair_wise_data["conf{}_{}_{}".format(id1, id0, id1)] = confidence
            pair_wise_data["intr{}".format(id0)] = intr0
            pair_wise_data["intr{}".format(id1)] = intr1

    for id1 in range(n_images):
        for id0 in range(id1):
            k0, k1 = "mkpts{}_{}_{}".format(id0, id0, id1), "mkpts{}_{}_{}".format(id1, id0, id1)
            c0, c1 = "conf{}_{}_{}".format(id0, id0, id1), "conf{}_{}_{}".format(id1, id0, id1)
            if k0 not in pair_wise_data:
                continue
            mkpts0, mkpts1 = pair_wise_data[k0], pair_wise_data[k1]
            success, R, t, inliers = estimate_relative_pose_w8pt_ba(pair_wise_data["intr{}".format(id0)], pair_wise_data["intr{}".format(id1)],
                                                                    mkpts0, mkpts1, pair_wise_data[c0])
            if success:
                inlier_count = inliers.sum()
                inliers = np.full_like(inliers, True)  # every match is kept; the count only weights the graph (:111-113)
            else:
                inlier_count = 0
            pair_wise_data["inlier_count{}_{}".format(id0, id1)] = inlier_count
            if success:
                pair_wise_data[k0], pair_wise_data[k1] = mkpts0[inliers], mkpts1[inliers]
                pair_wise_data[c0], pair_wise_data[c1] = pair_wise_data[c0][inliers], pair_wise_data[c1][inliers]
                rel_pose = np.eye(4)
                rel_pose[:3, :3] = R
                rel_pose[:3, 3] = t
                pair_wise_data["rel_pose{}_{}".format(id0, id1)] = rel_pose
                match_graph[id0, id1] = inliers.sum()

    # absolute poses along the maximum spanning tree (inlier counts as edge weights), bundle_adjust_io.py:134-170
    max_inliers = np.amax(match_graph)
    non_zero_mask = match_graph != 0
    match_graph[non_zero_mask] = max_inliers - match_graph[non_zero_mask] + 1
    min_spanning_tree = minimum_spanning_tree(match_graph).toarray().astype(int)
    pair_wise_data["abs_init_pose0"] = np.eye(4)
    n_abs_poses = 1
    row, col = np.nonzero(min_spanning_tree)
    pairs_on_spanning_tree = []
    for _ in range(n_images):
        for r, c in zip(row, col):
            id0, id1 = (r, c) if r < c else (c, r)
            pairs_on_spanning_tree.append((id0, id1))
            a0, a1 = "abs_init_pose{}".format(id0), "abs_init_pose{}".format(id1)
            if a1 not in pair_wise_data and a0 in pair_wise_data:
                pair_wise_data[a1] = pair_wise_data[a0] @ np.linalg.inv(pair_wise_data["rel_pose{}_{}".format(id0, id1)])
                n_abs_poses += 1
            elif a0 not in pair_wise_data and a1 in pair_wise_data:
                pair_wise_data[a0] = pair_wise_data[a1] @ pair_wise_data["rel_pose{}_{}".format(id0, id1)]
                n_abs_poses += 1
        if n_abs_poses == n_images:
            break
    extr = [np.eye(4)]
    for id in range(1, n_images):
        key = "abs_init_pose{}".format(id)
        extr.append(np.linalg.inv(pair_wise_data[key]) if key in pair_wise_data else np.eye(4))
    extr = np.array(extr)

    with open(file_path, 'w') as f:  # wire format read by ba_init.cpp:13-51
        for id in range(n_images):
            R = extr[id, :3, :3]
            f.write("{},{},{},{},{},{},{},{},{},{}\n".format(id, R[0, 0], R[1, 0], R[2, 0], R[0, 1], R[1, 1], R[2, 1], R[0, 2], R[1, 2],
                                                             R[2, 2]))
        for id1 in range(n_images):
            for id0 in range(id1):
                rel_pose_key = "rel_pose{}_{}".format(id0, id1)
                if rel_pose_key in pair_wise_data:
                    n_inliers = pair_wise_data["inlier_count{}_{}".format(id0, id1)]
                    if n_inliers >= min_inliers or (id0, id1) in pairs_on_spanning_tree:
                        T_021 = pair_wise_data[rel_pose_key]
                        R_021 = T_021[:3, :3]
                        t_021 = -R_021.transpose() @ T_021[:3, 3]
                        f.write("{},{},{},{},{},{},{},{},{},{},{},{},{},{}\n".format(
                            id0, id1, R_021[0, 0], R_021[1, 0], R_021[2, 0], R_021[0, 1], R_021[1, 1], R_021[2, 1], R_021[0, 2],
                            R_021[1, 2], R_021[2, 2], t_021[0], t_021[1], t_021[2]))
    return pair_wise_data


def triangulate_points(P0, P1, x0, x1):
    """``cv2.triangulatePoints`` + dehomogenisation as used at bundle_adjust_io.py:226-227; P [3,4], x [n,2] -> [n,3]."""
    ctx = _lib.context(_dev())
    n = x0.shape[0]
    P0, P1 = np.ascontiguousarray(P0, np.float64), np.ascontiguousarray(P1, np.float64)
    x0, x1 = np.ascontiguousarray(x0, np.float64), np.ascontiguousarray(x1, np.float64)
    out = np.zeros((n, 3))
    ctx.call("e2emv_mv_triangulate", n, _p(P0), _p(P1), _p(x0), _p(x1), _p(out), _lib.stream_ptr(_dev()))
    return out


def write_bundle_adjust_problem(n_images, pair_wise_data, extrinsics, file_path):
    """``write_bundle_adjust_problem`` (bundle_adjust_io.py:193-259): one 3-D point per match (no track merging), two
    observations each, confidences normalised to sum 2, intrinsics folded into the observations (header says f=1, c=0)."""
    if extrinsics.ndim != 3:
        extrinsics = np.array([np.eye(4) for _ in range(n_images)])
    min_inliers = 0
    n_3d_pts = 0
    observations_img_id, observations_pt_id, observations_xyc, points_in_3d = [], [], [], []
    for id1 in range(n_images):
        for id0 in range(id1):
            mkpts0_key = "mkpts{}_{}_{}".format(id0, id0, id1)
            if mkpts0_key in pair_wise_data:
                n_inliers = pair_wise_data["inlier_count{}_{}".format(id0, id1)]
                if n_inliers >= min_inliers:
                    mkpts0 = pair_wise_data[mkpts0_key]
                    mkpts1 = pair_wise_data["mkpts{}_{}_{}".format(id1, id0, id1)]
                    conf0 = pair_wise_data["conf{}_{}_{}".format(id0, id0, id1)]
                    conf1 = pair_wise_data["conf{}_{}_{}".format(id1, id0, id1)]
                    intr0 = pair_wise_data["intr{}".format(id0)]
                    intr1 = pair_wise_data["intr{}".format(id1)]
                    mkpts0 = (mkpts0 - intr0[[0, 1], [2, 2]][None]) / intr0[[0, 1], [0, 1]][None]
                    mkpts1 = (mkpts1 - intr1[[0, 1], [2, 2]][None]) / intr1[[0, 1], [0, 1]][None]
                    if mkpts0.shape[0] != 0:
                        pts_3d = triangulate_points(extrinsics[id0, :3, :], extrinsics[id1, :3, :], mkpts0, mkpts1)
                    else:
                        pts_3d = np.zeros((0, 3))
                    for id, mkpts, conf in zip((id0, id1), (mkpts0, mkpts1), (conf0, conf1)):
                        observations_img_id.append(np.full(mkpts.shape[0], id, dtype=int))
                        observations_pt_id.append(np.arange(n_3d_pts, n_3d_pts + pts_3d.shape[0], dtype=int))
                        observations_xyc.append(np.concatenate((mkpts, conf), -1))
                    n_3d_pts += pts_3d.shape[0]
                    points_in_3d.append(pts_3d)
    observations_img_id = np.concatenate(observations_img_id, 0)
    observations_pt_id = np.concatenate(observations_pt_id, 0)
    observations_xyc = normalize_confidences(np.concatenate(observations_xyc, 0))
    points_in_3d = np.concatenate(points_in_3d, 0)
    with open(file_path, 'w') as f:  # wire format read by ba_problem.cpp:15-87
        ref_cam = 0
        f.write("{},{},{},{},{},{},{},{}\n".format(n_images, ref_cam, n_3d_pts, 2 * n_3d_pts, 1., 1., 0., 0.))
        for id, pt_id, kpt in zip(observations_img_id, observations_pt_id, observations_xyc):
            if kpt.shape[0] == 3:
                f.write("{},{},{},{},{}\n".format(id, pt_id, kpt[0], kpt[1], kpt[2]))
            elif kpt.shape[0] == 4:
                f.write("{},{},{},{},{},{}\n".format(id, pt_id, kpt[0], kpt[1], kpt[2], kpt[3]))
            else:
                logging.error("Unexpected number of confidence values")
        for id in range(n_images):
            R, t = extrinsics[id, :3, :3], extrinsics[id, :3, 3]
            f.write("{},{},{},{},{},{},{},{},{},{},{},{}\n".format(R[0, 0], R[1, 0], R[2, 0], R[0, 1], R[1, 1], R[2, 1], R[0, 2], R[1, 2],
                                                                   R[2, 2], t[0], t[1], t[2]))
        for pt_3d in points_in_3d:
            f.write("{},{},{}\n".format(pt_3d[0], pt_3d[1], pt_3d[2]))


def read_bundle_adjust_result(file_path):
    """``read_bundle_adjust_result`` (bundle_adjust_io.py:261-273): rows of column-major R + t -> list of 4x4."""
    extrinsics = []
    with open(file_path, "r") as f:
        for line in f:
            w = line.split(',')
            R = np.array([[float(w[0]), float(w[3]), float(w[6])], [float(w[1]), float(w[4]), float(w[7])],
                          [float(w[2]), float(w[5]), float(w[8])]])
            T = np.eye(4)
            T[:3, :3] = R
            T[:3, 3] = [float(w[9]), float(w[10]), float(w[11])]
            extrinsics.append(T)
    return extrinsics


def run_ba_initializer(directory):
    """The reference's ``ba_initializer <dir>`` (ba_initializer.cpp:7-23): ``<dir>/ba_init_in.csv`` -> ``ba_init_out.csv``."""
    lib = _lib.load_library()
    rc = lib.e2emv_mv_init_files(os.path.join(directory, "ba_init_in.csv").encode(), os.path.join(directory, "ba_init_out.csv").encode())
    if rc != 0:
        raise _lib.E2EMVError(rc, "ba_initializer failed on " + directory)


def run_bundle_adjuster(directory):
    """The reference's ``bundle_adjuster <dir>`` (bundle_adjuster.cpp:7-23): ``<dir>/ba_in.csv`` -> ``ba_out.csv``."""
    dev = _dev()
    ctx = _lib.context(dev)
    ctx.call("e2emv_mv_bundle_adjust_files", os.path.join(directory, "ba_in.csv").encode(), os.path.join(directory, "ba_out.csv").encode(),
             _lib.stream_ptr(dev))


def bundle_adjust(n_cams, fixed_cam, intr, cam_idx, pt_idx, obs_xy, obs_w, cams, pts, max_iterations=50):
    """In-memory form of the device solver (``e2emv_mv_bundle_adjust``); returns ``(cams, pts, summary)``."""
    dev = _dev()
    ctx = _lib.context(dev)
    cam_idx, pt_idx = np.ascontiguousarray(cam_idx, np.int32), np.ascontiguousarray(pt_idx, np.int32)
    obs_xy, obs_w = np.ascontiguousarray(obs_xy, np.float64), np.ascontiguousarray(obs_w, np.float64)
    cams, pts = np.array(cams, np.float64).reshape(-1, 6).copy(), np.array(pts, np.float64).reshape(-1, 3).copy()
    intr = np.ascontiguousarray(intr, np.float64)
    summary = np.zeros(4)
    ctx.call("e2emv_mv_bundle_adjust", int(n_cams), int(fixed_cam), len(pts), len(cam_idx), _p(intr), _p(cam_idx), _p(pt_idx), _p(obs_xy),
             _p(obs_w), _p(cams), _p(pts), int(max_iterations), _p(summary), _lib.stream_ptr(dev))
    names = ["max_iterations", "gradient_tolerance", "parameter_tolerance", "function_tolerance", "invalid_steps", "radius"]
    return cams, pts, dict(initial_cost=summary[0], final_cost=summary[1], iterations=int(summary[2]), termination=names[int(summary[3])])


def eval_bundle_adjust(tuple_size, data, result, tmp_dir, pose_errors, verbose=False):
    """``eval_bundle_adjust`` (eval_multi_view.py:21-68): the full multi-view back-end for one tuple; appends
    max(err_t, err_R), err_t, err_R (degrees) of every image pair to ``pose_errors``."""
    os.makedirs(tmp_dir, exist_ok=True)
    pair_wise_data = initialize_bundle_adjust(tuple_size, data, result, os.path.join(tmp_dir, "ba_init_in.csv"))
    run_ba_initializer(tmp_dir)
    extrinsics = np.array(read_bundle_adjust_result(os.path.join(tmp_dir, "ba_init_out.csv")))
    write_bundle_adjust_problem(tuple_size, pair_wise_data, extrinsics, os.path.join(tmp_dir, "ba_in.csv"))
    run_bundle_adjuster(tmp_dir)
    extrinsics = read_bundle_adjust_result(os.path.join(tmp_dir, "ba_out.csv"))
    for id1 in range(tuple_size):
        for id0 in range(id1):
            pose0, pose1 = data["pose{}".format(id0)][0].cpu().numpy(), data["pose{}".format(id1)][0].cpu().numpy()
            T_021 = np.linalg.inv(pose1) @ pose0
            T_021_pred = extrinsics[id1] @ np.linalg.inv(extrinsics[id0])
            err_t, err_R = compute_pose_error(T_021, T_021_pred[:3, :3], T_021_pred[:3, 3])
            pose_errors[0].append(np.maximum(err_t, err_R))
            pose_errors[1].append(err_t)
            pose_errors[2].append(err_R)
            if verbose:
                logging.info("{} -> {}: rot {:>5.1f}deg\tt {:>5.1f}deg".format(id0, id1, err_R, err_t))
    return pose_errors


if __name__ == "__main__":
    if len(sys.argv) != 3 or sys.argv[1] not in ("ba_initializer", "bundle_adjuster"):
        sys.stderr.write("Usage: python -m e2e_multi_view_matching_amd.multi_view {ba_initializer|bundle_adjuster} <path to read and write>\n")
        sys.exit(1)
    (run_ba_initializer if sys.argv[1] == "ba_initializer" else run_bundle_adjuster)(sys.argv[2])
