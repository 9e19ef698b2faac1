"""No-edit drop-in root: put `dropin/` on PYTHONPATH of a reference checkout.

The reference's `pose_optimization/` has no `__init__.py` (a namespace package), so this REGULAR package of the same name wins the
import wherever `dropin/` sits on `sys.path` (a regular package found anywhere on the path beats namespace portions),
also when the reference checkout is the script directory.  `extend_path` then appends the reference's own directory, so
its modules that are not replaced here (`e.g. two_view/bundle_adjust_gauss_newton_2_view.py`) stay importable."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
