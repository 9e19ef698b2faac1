"""`models.models.utils` names the reference imports (`eval_pairs.py:16`, `eval_multi_view.py:15`,
`bundle_adjust_io.py:8`): the metric helpers of the hot path's evaluation, the two image-plane rotation helpers of the
ScanNet loader, and `estimate_pose` - OpenCV RANSAC, outside this implementation: importable, raises when called."""
from e2e_multi_view_matching_amd.metrics import (compute_pose_error, pose_auc, rotate_intrinsics,  # noqa: F401
                                                 rotate_pose_inplane)


def estimate_pose(kpts0, kpts1, K0, K1, thresh, conf=0.99999):
    raise NotImplementedError("estimate_pose is OpenCV's RANSAC essential-matrix solver (cv2.findEssentialMat / recoverPose); "
                              "the MI355X path implements the reference's w8pt / w8pt_ba modes only")
