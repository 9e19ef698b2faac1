"""C-ABI error behaviour (-m gpu): status codes + messages instead of crashes; reference conventions in the shim."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_status_codes_and_messages(gpu):
    from e2e_multi_view_matching_amd import _lib
    ctx = _lib.context(gpu)
    lib, h = ctx.lib, ctx.h
    z = torch.zeros(4, 4, device=gpu)
    sp = _lib.stream_ptr(gpu)
    assert lib.e2emv_sinkhorn(h, 0, 4, 4, _lib.ptr(z), 1.0, 3, _lib.ptr(z), sp) == _lib.ESHAPE
    assert b"sinkhorn" in lib.e2emv_last_error(h)
    assert lib.e2emv_sinkhorn(h, 1, 4, 4, None, 1.0, 3, _lib.ptr(z), sp) == _lib.EINVAL
    big = torch.zeros(1, 8, 4096, device=gpu)
    out = torch.zeros(1, 9, 4097, device=gpu)
    assert lib.e2emv_sinkhorn(h, 1, 8, 4096, _lib.ptr(big), 1.0, 1, _lib.ptr(out), sp) == _lib.ESHAPE  # N > 2048
    k = torch.zeros(1, 7, 2, device=gpu)
    eye = torch.eye(3, device=gpu)[None].contiguous()
    T = torch.zeros(1, 4, 4, device=gpu)
    u8 = torch.zeros(1, 7, dtype=torch.uint8, device=gpu)
    rc = lib.e2emv_w8pt(h, 1, 7, _lib.ptr(k), _lib.ptr(k), _lib.ptr(eye), _lib.ptr(eye), 3, 1, _lib.ptr(k), 0, None, 0,
                        _lib.ptr(T), _lib.ptr(k), _lib.ptr(k), _lib.ptr(k), None, _lib.ptr(u8), None, None, sp)
    assert rc == _lib.ESHAPE and b"fewer than 8" in lib.e2emv_last_error(h)
    rc = lib.e2emv_w8pt(h, 1, 9, _lib.ptr(k), _lib.ptr(k), _lib.ptr(eye), _lib.ptr(eye), 5, 1, _lib.ptr(k), 0, None, 0,
                        _lib.ptr(T), _lib.ptr(k), _lib.ptr(k), _lib.ptr(k), None, _lib.ptr(u8), None, None, sp)
    assert rc == _lib.ESHAPE  # intrinsics must be 3x3 or 4x4
    rc = lib.e2emv_w8pt(h, 1, 9, _lib.ptr(k), _lib.ptr(k), _lib.ptr(eye), _lib.ptr(eye), 3, 1, _lib.ptr(k), 1, None, 0,
                        _lib.ptr(T), _lib.ptr(k), _lib.ptr(k), _lib.ptr(k), None, _lib.ptr(u8), None, None, sp)
    assert rc == _lib.EINVAL  # choose_closest without T_021
    assert lib.e2emv_gemm_nt(h, 1, 8, 8, 24, 24, _lib.ptr(z), 24, 0, None, 0, 0, _lib.ptr(z), 24, 0, None, None, 0, 0,
                             _lib.ptr(z), 8, 0, 1.0, 0, sp) == _lib.ESHAPE  # K % 32
    assert lib.e2emv_set_precision(h, 7) == _lib.EINVAL
    torch.cuda.synchronize()


def test_forward_without_weights_and_bad_shapes(gpu):
    from e2e_multi_view_matching_amd import MultiViewMatcher, _lib
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    lib = _lib.load_library()
    h = ctypes.c_void_p()
    assert lib.e2emv_create(ctypes.byref(h), 0) == 0
    fd = _lib.ForwardDesc()
    fd.batch, fd.tuple_size, fd.n_kpts = 1, 2, 16
    a, _ = _lib.ptr_array([None, None])
    assert lib.e2emv_matcher_forward(h, ctypes.byref(fd), a, a, a, a, a, a, a, a, a, None) == _lib.ESTATE  # not committed
    md = _lib.ModelDesc()
    md.desc_dim, md.num_heads, md.n_kenc, md.n_layers = 256, 4, 4, 0
    for i, c in enumerate([32, 64, 128, 256]):
        md.kenc[i] = c
    assert lib.e2emv_commit_weights(h, ctypes.byref(md)) == _lib.ESTATE and b"missing weight" in lib.e2emv_last_error(h)
    md.num_heads = 8  # head dim 32 is not supported by the attention kernels
    assert lib.e2emv_commit_weights(h, ctypes.byref(md)) == _lib.ESHAPE
    lib.e2emv_destroy(h)
    model = MultiViewMatcher({"GNN_layers": ["self"], "sinkhorn_iterations": 2}).to(gpu).eval()
    d = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in make_tuples(batch=1, n_kpts=64).items()}
    d["scores1"] = d["scores1"][:, :32]
    with pytest.raises(AssertionError):
        model(d)
    with pytest.raises(KeyError):
        model({"keypoints0": d["keypoints0"]})
    too_many = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in make_tuples(batch=1, n_kpts=2100).items()}
    with pytest.raises(_lib.E2EMVError, match="2048"):
        model(too_many)


def test_empty_keypoints_follow_upstream(gpu):
    from e2e_multi_view_matching_amd import MultiViewMatcher, run_weighted_8_point
    model = MultiViewMatcher({"GNN_layers": ["self"]}).to(gpu).eval()
    d = {"keypoints0": torch.zeros(2, 0, 2, device=gpu), "keypoints1": torch.zeros(2, 0, 2, device=gpu),
         "scores0": torch.zeros(2, 0, device=gpu), "scores1": torch.zeros(2, 0, device=gpu),
         "descriptors0": torch.zeros(2, 256, 0, device=gpu), "descriptors1": torch.zeros(2, 256, 0, device=gpu),
         "image_size0": (480, 640), "image_size1": (480, 640)}
    out = model(d)
    assert out["matches0_0_1"].shape == (2, 0) and out["conf_scores_0_1"].shape == (2, 0, 1)
    assert run_weighted_8_point(d, out, 0, 1) == (None, None)  # estimate_relative_pose.py:132-136
