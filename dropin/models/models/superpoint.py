"""Import-path shim: `from models.models.superpoint import SuperPoint` (reference `train.py:17`, `eval_pairs.py:15`,
`eval_multi_view.py:17`) resolves to the MI355X implementation."""
from e2e_multi_view_matching_amd.superpoint import SuperPoint  # noqa: F401
