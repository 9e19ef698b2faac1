"""Drop-in for ``models.models.superpoint.SuperPoint`` (absent submodule of the reference; constructed at ``train.py:335-341``,
``eval_pairs.py:197-202``, ``eval_multi_view.py:135-140``; called by ``helpers.run_super_point``, ``helpers.py:83-96``).

An ``nn.Module`` shell holding upstream's parameters (``conv1a`` ... ``convDb``, so upstream ``superpoint_v1.pth`` loads with
``load_state_dict``); ``forward({"image": [batch, ...]})`` hands the images to ``e2emv_superpoint_forward`` (csrc/superpoint.hip:
NHWC implicit-GEMM convolutions on the fp32 matrix cores, fused NMS / top-k / descriptor sampling kernels) and returns the
reference's dict of per-image lists: ``keypoints`` [n,2] (x, y), ``scores`` [n], ``descriptors`` [256,n].  No torch op
computes anything; no CPU fallback.
"""
import ctypes
import warnings

import torch
from torch import nn

from . import _lib

LAYERS = [("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3), ("conv3a", 64, 128, 3),
          ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3), ("convPa", 128, 256, 3), ("convPb", 256, 65, 1),
          ("convDa", 128, 256, 3), ("convDb", 256, 256, 1)]
MAX_KEYPOINTS_CAPACITY = 4096


class SuperPoint(nn.Module):
    default_config = {"descriptor_dim": 256, "nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": -1, "remove_borders": 4,
                      "fill_with_random_keypoints": False}

    def __init__(self, config=None):
        super().__init__()
        self.config = {**self.default_config, **(config or {})}
        for name, cin, cout, k in LAYERS:
            setattr(self, name, nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=k // 2))
        mk = self.config["max_keypoints"]
        if mk == 0 or mk > MAX_KEYPOINTS_CAPACITY:
            raise ValueError('"max_keypoints" must be positive (<= {}) or -1'.format(MAX_KEYPOINTS_CAPACITY))
        self._token = _lib.new_owner_token()
        self.requires_grad_(False)  # the reference runs SuperPoint frozen, under no_grad (helpers.py:86)

    def _upload(self, ctx):
        # one SuperPoint weight set per context: re-upload whenever another instance (or other values) own it
        owner = (self._token, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        if ctx.sp_weights_owner == owner:
            return
        ctx.sp_weights_owner = None
        for k, v in self.state_dict().items():
            t = v.detach().to("cpu", torch.float32).contiguous()
            shape = (ctypes.c_int64 * t.dim())(*t.shape)
            ctx.call("e2emv_set_weight", ("superpoint." + k).encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim())
        ctx.call("e2emv_superpoint_commit")
        ctx.sp_weights_owner = owner

    def forward(self, data):
        images = data["image"]
        if torch.is_tensor(images):
            images = [images]
        out = {"keypoints": [], "scores": [], "descriptors": []}
        for batch in images:  # helpers.py:73-81: one merged batch, or one entry per image of the tuple
            if not batch.is_cuda:
                raise RuntimeError("SuperPoint needs its images on an MI355X (no CPU fallback)")
            dev = batch.device
            ctx = _lib.context(dev)
            self._upload(ctx)
            B, C, H, W = batch.shape
            if C != 1:
                raise AssertionError("SuperPoint takes one-channel images, got {}".format(tuple(batch.shape)))
            # the encoder pools three times: only the top-left (H//8*8) x (W//8*8) region yields keypoints, but upstream's
            # convolutions SEE the rows / columns beyond it before the pools floor them away.  Same here: the image goes in
            # zero-padded to the next multiple of 8 and the library masks every encoder level to its valid size.
            Hv, Wv = H, W
            if H % 8 or W % 8:
                Hp, Wp = (H + 7) // 8 * 8, (W + 7) // 8 * 8
                batch = torch.nn.functional.pad(batch, (0, Wp - W, 0, Hp - H))
                H, W = Hp, Wp
            img = batch.to(torch.float32).contiguous()
            cfg = self.config
            K = cfg["max_keypoints"] if cfg["max_keypoints"] > 0 else MAX_KEYPOINTS_CAPACITY
            d = _lib.SuperPointDesc(valid_height=Hv, valid_width=Wv, batch=B, height=H, width=W, nms_radius=int(cfg["nms_radius"]), max_keypoints=K,
                                    remove_borders=int(cfg["remove_borders"]), fill_random=1 if cfg["fill_with_random_keypoints"] else 0,
                                    keypoint_threshold=float(cfg["keypoint_threshold"]), seed=int(cfg.get("seed", 0)))
            kpts = torch.empty((B, K, 2), dtype=torch.float32, device=dev)
            scores = torch.empty((B, K), dtype=torch.float32, device=dev)
            desc = torch.empty((B, 256, K), dtype=torch.float32, device=dev)
            count = torch.empty((B,), dtype=torch.int32, device=dev)
            smap = torch.empty((B, Hv // 8 * 8, Wv // 8 * 8), dtype=torch.float32, device=dev) if cfg.get("return_score_map") else None
            with torch.cuda.device(dev), ctx.py_lock:
                self._upload(ctx)  # (again, now under the lock: another thread's model may have taken the context's weight set)
                ctx.call("e2emv_superpoint_forward", ctypes.byref(d), _lib.ptr(img), _lib.ptr(kpts), _lib.ptr(scores), _lib.ptr(desc),
                         _lib.ptr(count), _lib.ptr(smap), _lib.stream_ptr(dev))
            n = count.tolist()
            if cfg["max_keypoints"] <= 0 and max(n) >= MAX_KEYPOINTS_CAPACITY:
                warnings.warn("SuperPoint(max_keypoints=-1): an image reached the device capacity of {} keypoints; the lowest-"
                              "scoring candidates beyond it were dropped (upstream keeps all)".format(MAX_KEYPOINTS_CAPACITY))
            for b in range(B):
                out["keypoints"].append(kpts[b, :n[b]])
                out["scores"].append(scores[b, :n[b]])
                out["descriptors"].append(desc[b, :, :n[b]])
            if smap is not None:
                out.setdefault("score_map", []).append(smap)
        return out
