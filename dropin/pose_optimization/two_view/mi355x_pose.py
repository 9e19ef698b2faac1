"""Import-path shim.  In a reference checkout, replace the body of
`pose_optimization/two_view/estimate_relative_pose.py` below its BA import with
`from pose_optimization.two_view.mi355x_pose import *` - names and signatures are the reference's."""
from e2e_multi_view_matching_amd.pose import (compute_rotation_error, compute_translation_error_as_angle,  # noqa: F401
                                              estimate_relative_pose_w8pt, get_kpts, normalize,
                                              run_bundle_adjust_2_view, run_weighted_8_point)
