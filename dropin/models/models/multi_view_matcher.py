"""Import-path shim: `from models.models.multi_view_matcher import MultiViewMatcher`
(reference `train.py:18`, `eval_pairs.py:14`, `eval_multi_view.py:14`) resolves to the MI355X implementation."""
from e2e_multi_view_matching_amd.matcher import MultiViewMatcher, SuperGlue  # noqa: F401
