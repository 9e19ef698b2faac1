"""The no-edit drop-in (dropin/): with `dropin/` anywhere on sys.path the reference's import statements resolve to the
MI355X package, while reference modules that are NOT replaced stay importable.  Runs in subprocesses (clean sys.modules)
against a synthetic checkout with the reference's layout (namespace packages, no __init__.py) and, when it is present in
this container, against /root/reference itself."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "dropin")

CHECK = textwrap.dedent("""
    import sys
    ref, dropin, root = sys.argv[1:4]
    # `python eval_pairs.py` puts the checkout (script directory) FIRST; PYTHONPATH entries come after it
    sys.path[:0] = [ref, dropin, root]
    import e2e_multi_view_matching_amd as E
    from pose_optimization.two_view.estimate_relative_pose import (run_weighted_8_point, normalize, run_bundle_adjust_2_view,
                                                                  estimate_relative_pose_w8pt)          # helpers.py:12, eval_pairs.py:17
    from pose_optimization.two_view.compute_pose_error import compute_rotation_error, compute_translation_error_as_angle  # helpers.py:13
    from pose_optimization.multi_view.bundle_adjust_io import (initialize_bundle_adjust, write_bundle_adjust_problem,
                                                               read_bundle_adjust_result)            # eval_multi_view.py:19
    from models.models.multi_view_matcher import MultiViewMatcher                                     # train.py:18
    from models.models.superpoint import SuperPoint                                                   # train.py:17
    from models.models.utils import estimate_pose, pose_auc, compute_pose_error, rotate_pose_inplane, rotate_intrinsics  # eval_pairs.py:16
    assert run_weighted_8_point is E.run_weighted_8_point and normalize is E.normalize
    assert estimate_relative_pose_w8pt is E.estimate_relative_pose_w8pt and run_bundle_adjust_2_view is E.run_bundle_adjust_2_view
    assert compute_rotation_error is E.compute_rotation_error
    assert compute_translation_error_as_angle is E.compute_translation_error_as_angle
    assert MultiViewMatcher is E.MultiViewMatcher and SuperPoint is E.SuperPoint and pose_auc is E.pose_auc
    from e2e_multi_view_matching_amd import multi_view
    assert initialize_bundle_adjust is multi_view.initialize_bundle_adjust
    try:
        estimate_pose(None, None, None, None, 1.0)
    except NotImplementedError:
        pass
    else:
        raise SystemExit("estimate_pose must refuse (OpenCV RANSAC is out of scope)")
    import pose_optimization.two_view as tv
    assert any(p.startswith(ref) for p in tv.__path__), tv.__path__   # the checkout's own directory is still a portion
    print("RESOLVED")
""")


def _run(ref_dir, extra=""):
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-c", CHECK + extra, ref_dir, DROPIN, ROOT], capture_output=True, text=True, timeout=300,
                       env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RESOLVED" in r.stdout


def test_dropin_shadows_a_checkout_with_the_reference_layout(tmp_path):
    ref = tmp_path / "checkout"
    (ref / "pose_optimization" / "two_view").mkdir(parents=True)
    (ref / "pose_optimization" / "multi_view").mkdir(parents=True)
    (ref / "models").mkdir()  # un-initialised submodule: an empty directory
    poison = "raise ImportError('the reference module was imported instead of the drop-in')\n"
    (ref / "pose_optimization" / "two_view" / "estimate_relative_pose.py").write_text(poison)
    (ref / "pose_optimization" / "two_view" / "compute_pose_error.py").write_text(poison)
    (ref / "pose_optimization" / "multi_view" / "bundle_adjust_io.py").write_text(poison)
    (ref / "pose_optimization" / "two_view" / "bundle_adjust_gauss_newton_2_view.py").write_text("MARKER = 'reference file'\n")
    extra = textwrap.dedent("""
        from pose_optimization.two_view.bundle_adjust_gauss_newton_2_view import MARKER   # not replaced: the checkout's own
        assert MARKER == 'reference file'
    """)
    _run(str(ref), extra)


@pytest.mark.skipif(not os.path.isdir("/root/reference/pose_optimization"), reason="reference checkout not in this container")
def test_dropin_shadows_the_real_reference_tree():
    _run("/root/reference")


def test_executable_launchers_have_the_reference_command_line():
    for name in ("ba_initializer", "bundle_adjuster"):
        path = os.path.join(DROPIN, "bin", name)
        assert os.access(path, os.X_OK)
        assert "e2e_multi_view_matching_amd.multi_view " + name in open(path).read()
    r = subprocess.run([sys.executable, "-m", "e2e_multi_view_matching_amd.multi_view"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 1 and "Usage" in r.stderr


def test_rotation_helpers_roundtrip():
    import numpy as np
    from e2e_multi_view_matching_amd.metrics import rotate_intrinsics, rotate_pose_inplane
    K = np.array([[600.0, 0, 310.0], [0, 590.0, 250.0], [0, 0, 1.0]])
    # four quarter turns of the pose are the identity; two half turns of the intrinsics restore them
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [0.1, -0.2, 0.3]
    P = T
    for _ in range(4):
        P = rotate_pose_inplane(P, 1)
    assert np.allclose(P, T, atol=1e-6)
    K2 = rotate_intrinsics(rotate_intrinsics(K, (480, 640), 2), (480, 640), 2)
    assert np.allclose(K2, K)
    K1 = rotate_intrinsics(K, (640, 480), 1)  # shape AFTER the rotation: portrait
    assert K1[0, 0] == 590.0 and K1[1, 1] == 600.0 and K1[0, 2] == 250.0 and K1[1, 2] == 640 - 1 - 310.0
