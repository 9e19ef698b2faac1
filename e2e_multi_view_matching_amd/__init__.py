"""e2e_multi_view_matching_amd - MI355X-native matcher -> Sinkhorn -> weighted-8-point path.

Python surface (mirrors the reference's callables):
  MultiViewMatcher / SuperGlue, estimate_relative_pose_w8pt, run_weighted_8_point, get_kpts,
  normalize, compute_rotation_error, compute_translation_error_as_angle, pose_auc.
Everything computes in libe2emv.so (hand-written HIP for gfx950) through ctypes.
"""
from .matcher import MultiViewMatcher, SuperGlue, last_descriptors  # noqa: F401
from .metrics import compute_pose_error, pose_auc  # noqa: F401
from .ops import (attention, attention_bf16x3, attention_p2, extract_matches, gemm_bf16x3, gemm_nt, gemm_p2,  # noqa: F401
                  log_optimal_transport, qkv_p2)
from .pose import (compute_rotation_error, compute_translation_error_as_angle, estimate_relative_pose_w8pt,  # noqa: F401
                   get_kpts, mask_confidence, normalize, pose_errors, run_bundle_adjust_2_view,
                   run_weighted_8_point, run_weighted_8_point_tuple)

from .superpoint import SuperPoint  # noqa: F401,E402
from .targets import (compute_gt_matches_of_image_pair, compute_match_loss, gt_matches_for_tuple,  # noqa: F401,E402
                      relative_pose)

__all__ = ["SuperPoint", "compute_gt_matches_of_image_pair", "compute_match_loss", "gt_matches_for_tuple", "relative_pose",
           "run_weighted_8_point_tuple", "MultiViewMatcher", "SuperGlue", "estimate_relative_pose_w8pt", "run_weighted_8_point", "get_kpts",
           "run_bundle_adjust_2_view", "normalize", "compute_rotation_error", "compute_translation_error_as_angle", "pose_errors", "pose_auc",
           "compute_pose_error", "log_optimal_transport", "extract_matches", "gemm_nt", "attention"]
