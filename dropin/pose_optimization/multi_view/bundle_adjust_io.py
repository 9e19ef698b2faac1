"""`pose_optimization.multi_view.bundle_adjust_io` of the reference (bundle_adjust_io.py:12-273), MI355X implementation;
imported unchanged by `eval_multi_view.py:19`.  The two executables `eval_multi_view.py:33,47` starts
(`bundle_adjustment/build/ba_initializer <dir>`, `.../bundle_adjuster <dir>`) are the launchers in `dropin/bin/`."""
from e2e_multi_view_matching_amd.multi_view import (estimate_relative_pose_w8pt_ba, eval_bundle_adjust,  # noqa: F401
                                                    initialize_bundle_adjust, normalize_confidences,
                                                    read_bundle_adjust_result, run_ba_initializer, run_bundle_adjuster,
                                                    write_bundle_adjust_problem)
