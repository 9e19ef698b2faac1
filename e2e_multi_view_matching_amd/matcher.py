"""``MultiViewMatcher`` - drop-in for ``models.models.multi_view_matcher.MultiViewMatcher``.

The reference imports it from an absent submodule (``train.py:18``, ``eval_pairs.py:14``,
``eval_multi_view.py:14``); the contract re-created here is the one its call sites use:

* ``MultiViewMatcher(config: dict)``, ``.config`` a mutable dict (``helpers.py:245`` sets
  ``config["full_output"]`` after construction), keys ``multi_frame_matching``,
  ``GNN_layers``, ``conf_mlp``, ``tuple_size`` (``train.py:343-348``) plus the upstream
  SuperGlue ones (``descriptor_dim``, ``keypoint_encoder``, ``sinkhorn_iterations``,
  ``match_threshold``).
* an ``nn.Module`` whose parameters carry the upstream names (``kenc.encoder.*``,
  ``gnn.layers.{i}.attn.{proj.{0,1,2},merge}``, ``gnn.layers.{i}.mlp.*``, ``final_proj``,
  ``bin_score``, ``conf_mlp.*``) so ``load_ckpt``'s ``load_state_dict(strict=False)``
  (``helpers.py:47-52``), ``get_parameters(..., "conf_mlp")`` (``helpers.py:63-71``) and
  DataParallel/DDP wrapping (``train.py:349-357``) work unchanged.
* ``forward(data) -> dict`` with ``scores_{i}_{j}`` [B,N+1,N+1], ``matches{i}_{i}_{j}``
  [B,N] int64, ``conf_scores_{i}_{j}`` [B,N,1] (App. A.4 of SURVEY.md).

The sub-modules below are parameter CONTAINERS only - their torch ``forward`` is never
called.  ``forward`` hands the weights to libe2emv.so (BN folding / head re-ordering happen
there) and runs the hand-written HIP path; without the library or an MI355X it raises.

Training, first slice: in ``.train()`` mode with gradients enabled the ``scores_{i}_{j}`` carry an autograd
graph (``_MatchScores``: the library's fp32 forward with a tape and its hand-written backward, csrc/train.hip), so the
reference's stage-1 step ``match_loss(...).backward(); optimizer.step()`` (``train.py:406-425``) works unchanged.
BatchNorm layers normalise with their running statistics there (they are not updated).  With ``full_output`` (the reference
sets it for the pose loss, ``helpers.py:245``) the matches come from the same fp32 forward and, for a model with ``conf_mlp``,
``conf_scores_{i}_{j}`` carry the graph too: the pose loss of stage 2 reaches ``conf_mlp`` and the GNN through
``pose.run_weighted_8_point`` (its backward: ``e2emv_w8pt_backward``).
"""
import ctypes

import torch
from torch import nn

from . import _lib

DEFAULT_CONFIG = {
    "descriptor_dim": 256,
    "keypoint_encoder": [32, 64, 128, 256],
    "GNN_layers": ["self", "cross"] * 9,
    "num_heads": 4,
    "sinkhorn_iterations": 100,
    "match_threshold": 0.2,
    "multi_frame_matching": False,
    "tuple_size": 2,
    "conf_mlp": False,
    "full_output": False,
    # arithmetic of the dense GNN contractions: "f32" (exact fp32 MFMA), "bf16x3" (fp32 operands split into three bf16
    # planes, 6 bf16-MFMA products) or "f16x2" (two fp16 planes, 3 fp16-MFMA products), all with fp32 accumulation and the
    # same parity bar.  None = the library default (environment variable E2EMV_PRECISION, else "f16x2").
    "mfma_precision": None,
    # Training (.train() + gradients): BatchNorm layers normalise with their RUNNING statistics and do not update them
    # (csrc/train.hip) - gradients for their affine parameters are returned.  Upstream's matcher.train() (train.py:131,348)
    # would use batch statistics on a stock torch BatchNorm; the fork's own behaviour is not in the checkout.  The first
    # differentiable forward of a model warns about this unless the key is set to True (= "I know").
    "frozen_batchnorm": None,
    # True: forward() synchronises and raises if the device reported non-finite scores (off by default: the reference's
    # forward is asynchronous too, and the NaN / inf is in the outputs either way)
    "check_finite": False,
    # None: scores are differentiable whenever the module is in .train() mode, autograd is enabled and a parameter requires
    # grad (fp32 training path).  False: always the inference path (no graph).
    "autograd": None,
}


class _MatchScores(torch.autograd.Function):
    """(scores_{i}_{j} ..., conf_scores_{i}_{j} ...) = f(parameters): ``e2emv_matcher_forward_train`` (+ ``e2emv_conf_forward_train``
    per pair when the model has a ``conf_mlp`` and ``full_output`` is on) and ``e2emv_matcher_backward`` behind torch.autograd.

    The context keeps the tape of its LAST training forward only: backward() of an older graph raises."""

    @staticmethod
    def forward(fctx, cfg, *params):
        ctx, fd, kpts, scores, descs, shapes, names, full, thr, holder = cfg
        dev = params[0].device
        P = len(shapes)
        logZ = [torch.empty(s, dtype=torch.float32, device=dev) for s in shapes]
        keep, args = [], []
        for lst in (kpts, scores, descs, logZ):
            p, arr = _lib.ptr_array(lst)
            keep.append(arr)
            args.append(p)
        with torch.cuda.device(dev):
            ctx.call("e2emv_matcher_forward_train", ctypes.byref(fd), *args, _lib.stream_ptr(dev))
        ctx.train_generation += 1
        conf = []
        use_conf = full and any(n.startswith("conf_mlp.") for n in names)
        if full:
            from . import ops
            extra = {"m0": [], "m1": [], "s0": [], "s1": [], "conf": []}
            for p in range(P):
                m0, m1, s0, s1 = ops.extract_matches(logZ[p], thr)   # the mutual-match block on the scores of THIS forward
                for k, v in zip(("m0", "m1", "s0", "s1"), (m0, m1, s0, s1)):
                    extra[k].append(v)
                if use_conf:
                    c = torch.empty(m0.shape, dtype=torch.float32, device=dev)
                    with torch.cuda.device(dev):
                        ctx.call("e2emv_conf_forward_train", p, _lib.ptr(m0), _lib.ptr(c), _lib.stream_ptr(dev))
                    conf.append(c)
                else:  # no conf_mlp: the confidence is the match score of the valid matches (quirk E13) - not differentiated
                    from .pose import mask_confidence
                    extra["conf"].append(mask_confidence(s0, m0 >= 0))
            holder.append(extra)
            fctx.keep = extra  # (the library reads the matches again in the backward)
        fctx.lib_ctx, fctx.generation, fctx.names, fctx.dev, fctx.P = ctx, ctx.train_generation, names, dev, P
        fctx.n_conf = len(conf)
        fctx.param_shapes = [p.shape for p in params]
        fctx.set_materialize_grads(False)
        return tuple(logZ) + tuple(conf)

    @staticmethod
    def backward(fctx, *grads):
        ctx, dev, P = fctx.lib_ctx, fctx.dev, fctx.P
        if ctx.train_generation != fctx.generation:
            raise RuntimeError("MultiViewMatcher: backward() of a forward that is not the last training forward on this device "
                               "(the library keeps one tape per context)")
        g = [None if x is None else x.to(torch.float32).contiguous() for x in grads]
        pz, arr_z = _lib.ptr_array(g[:P])
        pc, arr_c = _lib.ptr_array(g[P:]) if fctx.n_conf else (None, None)
        out = [None]
        # one critical section from the generation check to the last e2emv_get_grad: another thread's training forward on this
        # device would re-commit (gradient arena zeroed, tape dropped) in between and the remaining reads would be silently wrong
        with ctx.py_lock, torch.cuda.device(dev):
            if ctx.train_generation != fctx.generation:
                raise RuntimeError("MultiViewMatcher: backward() of a forward that is not the last training forward on this device "
                                   "(the library keeps one tape per context)")
            ctx.call("e2emv_matcher_backward", pz, pc, _lib.stream_ptr(dev))
            for n, (name, shape) in enumerate(zip(fctx.names, fctx.param_shapes)):
                if not fctx.needs_input_grad[1 + n]:
                    out.append(None)
                    continue
                if name.startswith("conf_mlp.") and not fctx.n_conf:
                    out.append(torch.zeros(shape, dtype=torch.float32, device=dev))  # conf head not on this graph
                    continue
                t = torch.empty(shape, dtype=torch.float32, device=dev)
                ctx.call("e2emv_get_grad", name.encode(), ctypes.c_void_p(t.data_ptr()), t.numel(), _lib.stream_ptr(dev))
                out.append(t)
            if ctx.train_generation != fctx.generation:
                raise RuntimeError("MultiViewMatcher: another training forward ran on this device during backward()")
        return tuple(out)


def _mlp(channels, do_bn=True):
    """Upstream ``MLP``: Sequential(Conv1d, BN, ReLU, ..., Conv1d) - container only."""
    layers = []
    n = len(channels)
    for i in range(1, n):
        layers.append(nn.Conv1d(channels[i - 1], channels[i], kernel_size=1, bias=True))
        if i < n - 1:
            if do_bn:
                layers.append(nn.BatchNorm1d(channels[i]))
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class _KeypointEncoder(nn.Module):
    def __init__(self, feature_dim, layers):
        super().__init__()
        self.encoder = _mlp([3] + list(layers) + [feature_dim])
        nn.init.constant_(self.encoder[-1].bias, 0.0)


class _Attention(nn.Module):
    def __init__(self, d_model):
        super().__init__()
        self.merge = nn.Conv1d(d_model, d_model, kernel_size=1)
        self.proj = nn.ModuleList([nn.Conv1d(d_model, d_model, kernel_size=1) for _ in range(3)])


class _Propagation(nn.Module):
    def __init__(self, feature_dim):
        super().__init__()
        self.attn = _Attention(feature_dim)
        self.mlp = _mlp([feature_dim * 2, feature_dim * 2, feature_dim])
        nn.init.constant_(self.mlp[-1].bias, 0.0)


class _GNN(nn.Module):
    def __init__(self, feature_dim, n_layers):
        super().__init__()
        self.layers = nn.ModuleList([_Propagation(feature_dim) for _ in range(n_layers)])


class MultiViewMatcher(nn.Module):
    default_config = DEFAULT_CONFIG

    def __init__(self, config=None):
        super().__init__()
        self.config = {**self.default_config, **(config or {})}
        D = self.config["descriptor_dim"]
        self.kenc = _KeypointEncoder(D, self.config["keypoint_encoder"])
        self.gnn = _GNN(D, len(self.config["GNN_layers"]))
        self.final_proj = nn.Conv1d(D, D, kernel_size=1, bias=True)
        self.bin_score = nn.Parameter(torch.tensor(1.0))
        if self.config["conf_mlp"]:
            self.conf_mlp = _mlp([2 * D, D, 1])
        self._token = _lib.new_owner_token()

    # ------------------------------------------------------------------ weights -> library
    def _fingerprint(self):
        # (object, storage, version) of every floating-point tensor: what the library's committed copy is compared with on
        # every call.  state_dict() alone costs ~0.2 ms per call (a tenth of a batch-1 forward), so the WALK is cached - as
        # (owning module, name, is-buffer) slots, never as tensor objects: each call re-reads module._parameters[name] /
        # module._buffers[name], so a replaced object anywhere in the tree (submodule.load_state_dict(assign=True),
        # submodule.float() / .cuda() re-assigning BN buffers, submodule.weight = nn.Parameter(...), pruning) is seen, and
        # the module edges are re-checked so that a replaced SUBMODULE drops the walk.
        c = self.__dict__.get("_fp_slots")
        if c is not None:
            slots, edges = c
            for parent, name, child in edges:
                # name None: the module's own slot names (parameters | buffers, None-valued ones marked) and its child count -
                # a None slot filled later (bias=None -> nn.Parameter) or a submodule added to a nested module changes them
                if (self._slot_sig(parent) != child) if name is None else (parent._modules.get(name) is not child):
                    c = None
                    break
        if c is None:
            slots, edges = [], []
            for prefix, mod in self.named_modules():
                edges.append((mod, None, self._slot_sig(mod)))  # (a slot registered or filled later, a child added later)
                for name, child in mod._modules.items():
                    edges.append((mod, name, child))
                for name, v in mod._parameters.items():
                    if v is not None and v.dtype.is_floating_point:
                        slots.append((prefix + "." + name if prefix else name, mod, name, False))
                for name, v in mod._buffers.items():
                    if v is not None and v.dtype.is_floating_point and name not in mod._non_persistent_buffers_set:
                        slots.append((prefix + "." + name if prefix else name, mod, name, True))
            self.__dict__["_fp_slots"] = (slots, edges)
        fp = []
        for key, mod, name, is_buf in slots:
            v = (mod._buffers if is_buf else mod._parameters).get(name)
            if v is None or not v.dtype.is_floating_point:  # the slot went away or changed kind: rebuild the walk
                self.__dict__["_fp_slots"] = None
                return self._fingerprint()
            fp.append((key, id(v), v.data_ptr(), v._version))
        return tuple(fp)

    @staticmethod
    def _slot_sig(mod):
        return (tuple((k, v is None) for k, v in mod._parameters.items()), tuple((k, v is None) for k, v in mod._buffers.items()), len(mod._modules))

    def invalidate_weight_cache(self):
        """Drops the cached module walk (kept for callers of earlier versions; object replacement anywhere in the tree is now
        seen without it)."""
        self.__dict__["_fp_slots"] = None

    def __setattr__(self, name, value):
        if isinstance(value, (torch.Tensor, nn.Module)):
            self.__dict__["_fp_slots"] = None
        super().__setattr__(name, value)

    def _model_desc(self):
        cfg = self.config
        md = _lib.ModelDesc()
        md.desc_dim = cfg["descriptor_dim"]
        md.num_heads = cfg["num_heads"]
        kenc = list(cfg["keypoint_encoder"])
        if len(kenc) > _lib.MAX_KENC or len(cfg["GNN_layers"]) > _lib.MAX_LAYERS:
            raise ValueError("config exceeds E2EMV_MAX_KENC / E2EMV_MAX_LAYERS")
        md.n_kenc = len(kenc)
        for i, c in enumerate(kenc):
            md.kenc[i] = c
        md.n_layers = len(cfg["GNN_layers"])
        for i, name in enumerate(cfg["GNN_layers"]):
            if name not in ("self", "cross"):
                raise ValueError(f"GNN layer type {name!r}")
            md.layer_types[i] = 1 if name == "cross" else 0
        md.conf_mlp = 1 if cfg["conf_mlp"] else 0
        return md

    def _send_weights(self, ctx, owner):
        """e2emv_set_weight for every tensor (the library's host-side store), once per (module, parameter values)."""
        if getattr(ctx, "sent_owner", None) == owner:
            return
        ctx.sent_owner = None
        items = [(k, v.detach()) for k, v in self.state_dict().items() if v.dtype.is_floating_point]  # (not num_batches_tracked)
        # ONE device-to-host copy of all tensors (a training loop comes through here after every optimiser step: 250 separate
        # .cpu() calls were 250 synchronisations)
        flat = torch.cat([v.reshape(-1).to(torch.float32) for _, v in items]).cpu()
        base, off = flat.data_ptr(), 0
        for k, v in items:
            shape = (ctypes.c_int64 * max(v.dim(), 1))(*v.shape)
            ctx.call("e2emv_set_weight", k.encode(), ctypes.c_void_p(base + 4 * off), shape, v.dim())
            off += v.numel()
        ctx.sent_owner = owner

    def _push_weights(self, ctx):
        # the context holds one weight set: re-push whenever another module (or other parameter values) own it
        owner = (self._token, self._fingerprint())
        if ctx.weights_owner == owner:
            return
        ctx.weights_owner = None
        self._send_weights(ctx, owner)
        md = self._model_desc()
        ctx.call("e2emv_commit_weights", ctypes.byref(md))
        ctx.weights_owner = owner

    def _push_train_weights(self, ctx):
        """Training commit only: after an optimiser step the parameters changed, and what the differentiable path needs is the
        training arena (e2emv_train_commit: BN / head-order folds, arenas and tape kept) - NOT the inference commit with its
        fp64 merge fold and the bf16x3 / f16x2 planes, which is made when (if) an inference forward next needs it."""
        owner = (self._token, self._fingerprint())
        if ctx.train_owner == owner:
            return
        md = self._model_desc()
        if ctx.train_owner is not None and ctx.train_owner[0] == self._token:
            # the context already trains THIS module: new values device-to-device into its arena (e2emv_train_update), no host
            # round trip, no synchronisation beyond reading the scalar bin_score
            items = [(k, v.detach()) for k, v in self.state_dict().items() if v.dtype.is_floating_point]
            keep = [v if (v.dtype == torch.float32 and v.is_contiguous()) else v.to(torch.float32).contiguous() for _, v in items]
            n = len(items)
            keys = (ctypes.c_char_p * n)(*[k.encode() for k, _ in items])
            ptrs = (ctypes.c_void_p * n)(*[v.data_ptr() for v in keep])
            numels = (ctypes.c_int64 * n)(*[v.numel() for v in keep])
            dev = self.bin_score.device
            with torch.cuda.device(dev):
                rc = ctx.lib.e2emv_train_update(ctx.h, ctypes.byref(md), n, keys, ptrs, numels, float(self.bin_score.detach()), _lib.stream_ptr(dev))
            if rc == _lib.OK:
                ctx.train_owner = owner
                return
            ctx.train_owner = None  # whatever failed: the arena no longer provably holds the values of any owner
            if rc != _lib.ESTATE:
                ctx.check(rc)
        self._send_weights(ctx, owner)
        ctx.call("e2emv_train_commit", ctypes.byref(md))
        ctx.train_owner = owner

    def _differentiable(self):
        if self.config.get("autograd", None) is False or not self.training or not torch.is_grad_enabled():
            return False
        return any(p.requires_grad for p in self.parameters())

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _tuple_size(data):
        t = 0
        while f"keypoints{t}" in data:
            t += 1
        return t

    def forward(self, data):
        cfg = self.config
        T = self._tuple_size(data)
        if T < 2:
            raise KeyError("MultiViewMatcher.forward needs keypoints0, keypoints1, ...")
        dev = self.bin_score.device
        if dev.type != "cuda":
            raise RuntimeError("MultiViewMatcher runs only on an MI355X: call .cuda() first (there is no CPU path; "
                               "the CPU oracle lives in oracle/ and is test infrastructure)")
        ctx = _lib.context(dev)
        B = int(data["keypoints0"].shape[0])
        if cfg.get("streams", 1) == 2 and B >= 2 and not self._differentiable():
            return self._forward_two_streams(ctx, data, T, dev, B)
        with ctx.py_lock:
            return self._forward_locked(ctx, data, T, dev)

    def _forward_two_streams(self, ctx, data, T, dev, B):
        """config["streams"] = 2 (inference): the batch as two halves on two HIP streams and two library contexts.  A step is
        ~100 dependent launches; each boundary (tail of one kernel, drain, ramp of the next) idles part of the chip, and the
        other half's kernels fill it: 32 pairs of 1024 keypoints 10.26 -> 9.80 ms (tools/split_streams.py).  Tuples are
        independent: every output row equals the single-stream call's up to the Sinkhorn kernel's summation order - the library
        picks the resident kernel by shape AND batch size (32 pairs of 1024 keypoints run on 128-row workgroups in one call, the
        two halves of 16 on 64-row ones: < 2e-5 on log-assignments, the matches equal in every test) - and bit for bit with the
        kernel pinned (`_lib.context().set_sinkhorn_kernel("rows64")`, mirrored onto the peer context), as long as neither half's
        resident Sinkhorn gives up on an inter-workgroup wait under the contention of the other (stats()["sinkhorn_timeouts"]): a
        rescued problem is re-solved in the log domain, inside the 1e-4 bar but not bit for bit.  last_descriptors() then
        holds the first half only."""
        peer = _lib.peer_context(dev)
        peer.mirror_settings(ctx)
        h = (B + 1) // 2

        batched = ("keypoints", "scores", "descriptors", "image")  # + the image index; NOT image_size{m} ([h, w], B = 2 would split it)

        def part(lo, hi):
            return {k: (v[lo:hi] if torch.is_tensor(v) and k.rstrip("0123456789") in batched and v.dim() > 0 and v.shape[0] == B else v)
                    for k, v in data.items()}

        cur = torch.cuda.current_stream(dev)
        side = peer.side_stream
        side.wait_stream(cur)                      # the inputs were produced on the caller's stream
        with torch.cuda.stream(side), peer.py_lock:
            out1 = self._forward_locked(peer, part(h, B), T, dev)
        with ctx.py_lock:
            out0 = self._forward_locked(ctx, part(0, h), T, dev)
        cur.wait_stream(side)
        out = {}
        for k, v in out0.items():
            if torch.is_tensor(v):
                out1[k].record_stream(cur)
                out[k] = torch.cat([v, out1[k]], 0)
            else:
                out[k] = v
        return out

    def _forward_locked(self, ctx, data, T, dev):
        cfg = self.config
        # the differentiable path runs from the TRAINING arena (_push_train_weights below): the inference commit (fp64 merge
        # fold on the host, split-operand planes: 0.4 s) is made when an inference forward next needs it, not after every
        # optimiser step
        if not self._differentiable():
            self._push_weights(ctx)
        # the precision switch is context-global and sticky: resolve it on EVERY call (None = the context's explicit
        # override, else the library default), so a model never inherits what the previous model selected
        mode = cfg.get("mfma_precision")
        if mode is None:
            want = ctx.forced_precision if ctx.forced_precision is not None else ctx.default_precision
        else:
            want = _lib.PRECISION_NAMES[mode]
        if ctx.precision() != want:
            ctx.call("e2emv_set_precision", want)
        gen = {"f16x2-r2": 2, "f16x2-r3": 3, "f16x2-r4": 4, "f16x2-chain": 105}.get(mode, ctx.default_f16x2_kernels)
        if ctx.f16x2_kernels != gen:
            ctx.set_f16x2_kernels(gen)
        kpts, scores, descs = [], [], []
        fd = _lib.ForwardDesc()
        for m in range(T):
            k = data[f"keypoints{m}"].to(dev, torch.float32).contiguous()
            s = data[f"scores{m}"].to(dev, torch.float32).contiguous()
            d = data[f"descriptors{m}"].to(dev)
            if d.dtype not in (torch.float32, torch.float16):
                d = d.float()
            d = d.contiguous()
            kpts.append(k), scores.append(s), descs.append(d)
            if f"image{m}" in data:
                h, w = data[f"image{m}"].shape[-2:]
            else:
                h, w = data[f"image_size{m}"]
            fd.img_w[m], fd.img_h[m] = float(w), float(h)
        B = kpts[0].shape[0]
        Ns = [int(k.shape[1]) for k in kpts]  # per-image keypoint counts (eval_pairs.py: they differ between images)
        N = max(Ns)
        D = cfg["descriptor_dim"]
        for m in range(T):
            if kpts[m].shape != (B, Ns[m], 2) or scores[m].shape != (B, Ns[m]) or descs[m].shape != (B, D, Ns[m]):
                raise AssertionError(f"image {m}: keypoints {tuple(kpts[m].shape)} scores {tuple(scores[m].shape)} "
                                     f"descriptors {tuple(descs[m].shape)} (all images of a call share the batch size)")
        if len({d.dtype for d in descs}) != 1:
            raise AssertionError("mixed descriptor dtypes")
        out = {}
        pairs = [(i, j) for j in range(T) for i in range(j)]
        if min(Ns) == 0:  # upstream: no keypoints -> empty matches
            for i, j in pairs:
                out[f"scores_{i}_{j}"] = torch.full((B, Ns[i] + 1, Ns[j] + 1), 0.0, device=dev)
                out[f"matches{i}_{i}_{j}"] = torch.full((B, Ns[i]), -1, dtype=torch.int64, device=dev)
                out[f"matches{j}_{i}_{j}"] = torch.full((B, Ns[j]), -1, dtype=torch.int64, device=dev)
                out[f"conf_scores_{i}_{j}"] = torch.zeros((B, Ns[i], 1), device=dev)
            return out
        full = bool(cfg.get("full_output", False)) or not self.training
        fd.batch, fd.tuple_size, fd.n_kpts = B, T, N
        for m in range(T):
            fd.n_kpts_img[m] = Ns[m]
        fd.sinkhorn_iters = int(cfg["sinkhorn_iterations"])
        fd.match_threshold = float(cfg["match_threshold"])
        fd.desc_dtype = _lib.DESC_F16 if descs[0].dtype == torch.float16 else _lib.DESC_F32
        fd.flags = (_lib.FLAG_FULL_OUTPUT if full else 0) | (_lib.FLAG_MULTI_FRAME if cfg["multi_frame_matching"] else 0)
        P = len(pairs)
        if self._differentiable():
            if len(set(Ns)) != 1:
                raise NotImplementedError("training path: all images of a call must carry the same number of keypoints "
                                          "(the reference's training batches do, datasets pad to max_keypoints)")
            if T > 2 and not cfg["multi_frame_matching"]:
                raise NotImplementedError("training path: tuples of more than two images need multi_frame_matching")
            if not cfg.get("frozen_batchnorm") and not getattr(self, "_warned_bn", False):
                self._warned_bn = True
                import warnings
                warnings.warn("MultiViewMatcher in .train() mode: BatchNorm layers use their running statistics (frozen) and do not "
                              "update them - batch-statistics BatchNorm is not implemented on the differentiable path.  Set "
                              "config['frozen_batchnorm'] = True to acknowledge (e.g. fine-tuning from a checkpoint).", stacklevel=3)
            self._push_train_weights(ctx)
            named = [(k, p) for k, p in self.named_parameters()]
            holder = []
            cfgt = (ctx, fd, kpts, scores, descs, [(B, N + 1, N + 1)] * P, [k for k, _ in named], full, float(cfg["match_threshold"]), holder)
            outs = _MatchScores.apply(cfgt, *[p for _, p in named])
            out = {f"scores_{i}_{j}": outs[p] for p, (i, j) in enumerate(pairs)}
            if full:  # matches / confidences from the scores of this very forward (the fp32 training arithmetic)
                extra = holder[0]
                for p, (i, j) in enumerate(pairs):
                    out[f"matches{i}_{i}_{j}"] = extra["m0"][p]
                    out[f"matches{j}_{i}_{j}"] = extra["m1"][p]
                    out[f"matching_scores{i}_{i}_{j}"] = extra["s0"][p]
                    out[f"matching_scores{j}_{i}_{j}"] = extra["s1"][p]
                    c = outs[P + p] if len(outs) > P else extra["conf"][p]
                    out[f"conf_scores_{i}_{j}"] = c.unsqueeze(-1)
            return out
        logZ = [torch.empty((B, Ns[i] + 1, Ns[j] + 1), dtype=torch.float32, device=dev) for i, j in pairs]
        none = [None] * P
        if full:
            m0 = [torch.empty((B, Ns[i]), dtype=torch.int64, device=dev) for i, j in pairs]
            m1 = [torch.empty((B, Ns[j]), dtype=torch.int64, device=dev) for i, j in pairs]
            s0 = [torch.empty((B, Ns[i]), dtype=torch.float32, device=dev) for i, j in pairs]
            s1 = [torch.empty((B, Ns[j]), dtype=torch.float32, device=dev) for i, j in pairs]
            cf = [torch.empty((B, Ns[i]), dtype=torch.float32, device=dev) for i, j in pairs]
        else:
            m0 = m1 = s0 = s1 = cf = none
        keep = []
        args = []
        for lst in (kpts, scores, descs, logZ, m0, m1, s0, s1, cf):
            p, arr = _lib.ptr_array(lst)
            keep.append(arr)
            args.append(p)
        with torch.cuda.device(dev):
            ctx.call("e2emv_matcher_forward", ctypes.byref(fd), *args, _lib.stream_ptr(dev))
            if cfg.get("check_finite"):
                # host-synchronising: raises E2EMVError if a Sinkhorn problem left fp32's range or an activation left the
                # range of the arithmetic mode (the scores of that call are NaN / inf either way)
                ctx.call("e2emv_sync", _lib.stream_ptr(dev))
        for p, (i, j) in enumerate(pairs):
            out[f"scores_{i}_{j}"] = logZ[p]
            if full:
                out[f"matches{i}_{i}_{j}"] = m0[p]
                out[f"matches{j}_{i}_{j}"] = m1[p]
                out[f"matching_scores{i}_{i}_{j}"] = s0[p]
                out[f"matching_scores{j}_{i}_{j}"] = s1[p]
                out[f"conf_scores_{i}_{j}"] = cf[p].unsqueeze(-1)
        return out


def last_descriptors(device):
    """The matched descriptors (upstream's mdesc, final_proj output) of the last forward on `device`: [B*T, N, D] fp32,
    keypoint-major (``e2emv_get_descriptors``) - an audit output for parity tests of the GNN arithmetic.  They live in the
    library's workspace: valid until the next library call on the context that uses it (any later one raises ESTATE)."""
    dev = torch.device(device)
    ctx = _lib.context(dev)
    n_img, n, d = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    with ctx.py_lock, torch.cuda.device(dev):
        ctx.call("e2emv_get_descriptors", None, 0, ctypes.byref(n_img), ctypes.byref(n), ctypes.byref(d), _lib.stream_ptr(dev))
        out = torch.empty((n_img.value, n.value, d.value), dtype=torch.float32, device=dev)
        ctx.call("e2emv_get_descriptors", _lib.ptr(out), out.numel(), ctypes.byref(n_img), ctypes.byref(n), ctypes.byref(d), _lib.stream_ptr(dev))
    return out


SuperGlue = MultiViewMatcher  # the north-star's name for the same forward()
