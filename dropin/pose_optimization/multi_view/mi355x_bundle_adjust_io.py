"""Import-path shim.  In a reference checkout, replace the body of `pose_optimization/multi_view/bundle_adjust_io.py` by
`from pose_optimization.multi_view.mi355x_bundle_adjust_io import *` - names and signatures are the
reference's; `run_ba_initializer(dir)` / `run_bundle_adjuster(dir)` replace the two Theia/Ceres executables."""
from e2e_multi_view_matching_amd.multi_view import (estimate_relative_pose_w8pt_ba, eval_bundle_adjust,  # noqa: F401
                                                    initialize_bundle_adjust, normalize_confidences,
                                                    read_bundle_adjust_result, run_ba_initializer, run_bundle_adjuster,
                                                    write_bundle_adjust_problem)
