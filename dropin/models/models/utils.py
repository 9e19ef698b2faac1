"""Import-path shim for the metric helpers the reference imports from its absent submodule
(`eval_pairs.py:16`): only the two the hot path's evaluation needs."""
from e2e_multi_view_matching_amd.metrics import compute_pose_error, pose_auc  # noqa: F401
