"""`pose_optimization.two_view.compute_pose_error` of the reference (compute_pose_error.py:3-22), MI355X implementation;
imported unchanged by `helpers.py:13`."""
from e2e_multi_view_matching_amd.pose import compute_rotation_error, compute_translation_error_as_angle  # noqa: F401
