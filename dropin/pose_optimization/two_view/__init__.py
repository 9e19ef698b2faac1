"""Shadows `pose_optimization.two_view.{estimate_relative_pose, compute_pose_error}`.

The reference's `pose_optimization/two_view/` has no `__init__.py` (a namespace package), so this REGULAR package of the same name wins the
import wherever `dropin/` sits on `sys.path` (a regular package found anywhere on the path beats namespace portions),
also when the reference checkout is the script directory.  `extend_path` then appends the reference's own directory, so
its modules that are not replaced here (`bundle_adjust_gauss_newton_2_view.py`) stay importable."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
