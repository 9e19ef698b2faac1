"""`pose_optimization.two_view.estimate_relative_pose` of the reference (estimate_relative_pose.py:9-144), MI355X
implementation: same names, arguments and return conventions; imported unchanged by `helpers.py:12`, `eval_pairs.py:17`
and `pose_optimization/multi_view/bundle_adjust_io.py:9`."""
from e2e_multi_view_matching_amd.pose import (compute_rotation_error, compute_translation_error_as_angle,  # noqa: F401
                                              estimate_relative_pose_w8pt, get_kpts, normalize,
                                              run_bundle_adjust_2_view, run_weighted_8_point)
