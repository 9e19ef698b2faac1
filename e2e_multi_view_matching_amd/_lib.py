"""ctypes binding of libe2emv.so (include/e2emv.h).  No torch ops on the hot path: torch is
only the owner of device memory and streams; every compute call goes through the C ABI.

There is NO fallback: if the library or a gfx950 device is missing, ``context()`` raises.
"""
import ctypes
import os
import threading

import torch  # noqa: F401  (imported first so libamdhip64.so.7 resolves to the runtime torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
# E2EMV_LIBRARY: a measurement build of the same sources (tools/p2_stamps.py -> tools/*.bin), announced on stderr when
# used; the product is always libe2emv.so next to this file
LIB_PATH = os.environ.get("E2EMV_LIBRARY") or os.path.join(_HERE, "libe2emv.so")

MAX_TUPLE, MAX_LAYERS, MAX_KENC, PROF_SLOTS = 8, 64, 8, 16
FLAG_FULL_OUTPUT, FLAG_MULTI_FRAME = 1, 2
DESC_F32, DESC_F16 = 0, 1
OK, EINVAL, ENOMEM, EHIP, ESHAPE, ESTATE = 0, -1, -2, -3, -4, -5
PRECISION_F32, PRECISION_BF16X3, PRECISION_F16X2 = 0, 1, 2
PRECISION_NAMES = {"f32": PRECISION_F32, "bf16x3": PRECISION_BF16X3, "f16x2": PRECISION_F16X2,
                   "f16x2-r2": PRECISION_F16X2, "f16x2-r3": PRECISION_F16X2, "f16x2-r4": PRECISION_F16X2,
                   "f16x2-chain": PRECISION_F16X2}  # "-r2": the same arithmetic on the round-2 kernels (fp32 activations)

c_void_p, c_int, c_float, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_char_p
c_int64, c_size_t = ctypes.c_int64, ctypes.c_size_t


class ModelDesc(ctypes.Structure):
    _fields_ = [("desc_dim", ctypes.c_int32), ("num_heads", ctypes.c_int32), ("n_kenc", ctypes.c_int32),
                ("kenc", ctypes.c_int32 * MAX_KENC), ("n_layers", ctypes.c_int32),
                ("layer_types", ctypes.c_int32 * MAX_LAYERS), ("conf_mlp", ctypes.c_int32)]


class ForwardDesc(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("tuple_size", ctypes.c_int32), ("n_kpts", ctypes.c_int32),
                ("sinkhorn_iters", ctypes.c_int32), ("match_threshold", ctypes.c_float),
                ("desc_dtype", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("img_w", ctypes.c_float * MAX_TUPLE), ("img_h", ctypes.c_float * MAX_TUPLE),
                ("n_kpts_img", ctypes.c_int32 * MAX_TUPLE)]


class SuperPointDesc(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32), ("nms_radius", ctypes.c_int32),
                ("max_keypoints", ctypes.c_int32), ("remove_borders", ctypes.c_int32), ("fill_random", ctypes.c_int32),
                ("keypoint_threshold", ctypes.c_float), ("seed", ctypes.c_uint32), ("valid_height", ctypes.c_int32),
                ("valid_width", ctypes.c_int32)]


# every symbol include/e2emv.h declares: name -> (restype, argtypes)
_PP = ctypes.POINTER(c_void_p)
SIGNATURES = {
    "e2emv_version": (c_int, []),
    "e2emv_create": (c_int, [ctypes.POINTER(c_void_p), c_int]),
    "e2emv_destroy": (None, [c_void_p]),
    "e2emv_last_error": (c_char_p, [c_void_p]),
    "e2emv_malloc": (c_int, [c_void_p, ctypes.POINTER(c_void_p), c_size_t]),
    "e2emv_free": (c_int, [c_void_p, c_void_p]),
    "e2emv_h2d": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "e2emv_d2h": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "e2emv_sync": (c_int, [c_void_p, c_void_p]),
    "e2emv_get_descriptors": (c_int, [c_void_p, c_void_p, c_int64, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p]),
    "e2emv_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, ctypes.POINTER(c_int64), c_int]),
    "e2emv_commit_weights": (c_int, [c_void_p, ctypes.POINTER(ModelDesc)]),
    "e2emv_matcher_forward": (c_int, [c_void_p, ctypes.POINTER(ForwardDesc), _PP, _PP, _PP, _PP, _PP, _PP, _PP, _PP, _PP,
                                      c_void_p]),
    "e2emv_sinkhorn": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_int, c_void_p, c_void_p]),
    "e2emv_extract_matches": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "e2emv_gather_matched": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "e2emv_w8pt": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                           c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                           c_void_p]),
    "e2emv_w8pt_ragged": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                  c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "e2emv_w8pt_tuple": (c_int, [c_void_p, c_int, c_int, c_int, _PP, _PP, c_int, c_int, _PP, _PP, c_int, _PP, c_int, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_apply_mask": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_normalize_kpts": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "e2emv_relative_pose": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_pose_error_means": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_pose_errors": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_ba_2view": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "e2emv_gt_matches": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "e2emv_match_loss": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_mv_init": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_mv_estimate_rotations": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "e2emv_mv_estimate_positions": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_mv_init_files": (c_int, [c_char_p, c_char_p]),
    "e2emv_mv_bundle_adjust": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "e2emv_mv_bundle_adjust_files": (c_int, [c_void_p, c_char_p, c_char_p, c_void_p]),
    "e2emv_mv_triangulate": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_superpoint_commit": (c_int, [c_void_p]),
    "e2emv_superpoint_forward": (c_int, [c_void_p, ctypes.POINTER(SuperPointDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
    "e2emv_gemm_nt": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                              c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                              c_int64, c_float, c_int, c_void_p]),
    "e2emv_attention": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "e2emv_set_precision": (c_int, [c_void_p, c_int]),
    "e2emv_get_precision": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "e2emv_set_split_min_rows": (c_int, [c_void_p, c_int64]),
    "e2emv_gemm_bf16x3": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "e2emv_attention_bf16x3": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                       c_void_p]),
    "e2emv_set_f16x2_kernels": (c_int, [c_void_p, c_int]),
    "e2emv_set_attention_key_split": (c_int, [c_void_p, c_int]),
    "e2emv_gemm_p2": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                              c_void_p]),
    "e2emv_qkv_p2": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_attention_p2": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "e2emv_train_commit": (c_int, [c_void_p, ctypes.POINTER(ModelDesc)]),
    "e2emv_train_update": (c_int, [c_void_p, ctypes.POINTER(ModelDesc), c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_void_p),
                           ctypes.POINTER(ctypes.c_int64), ctypes.c_float, c_void_p]),
    "e2emv_matcher_forward_train": (c_int, [c_void_p, ctypes.POINTER(ForwardDesc), _PP, _PP, _PP, _PP, c_void_p]),
    "e2emv_conf_forward_train": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "e2emv_matcher_backward": (c_int, [c_void_p, _PP, _PP, c_void_p]),
    "e2emv_get_grad": (c_int, [c_void_p, c_char_p, c_void_p, c_int64, c_void_p]),
    "e2emv_w8pt_backward": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_pose_errors_backward": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e2emv_get_stats": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_uint64), c_int, c_int]),
    "e2emv_set_sinkhorn_kernel": (c_int, [c_void_p, c_int]),
    "e2emv_sinkhorn_plan": (c_int, [c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), c_int]),
    "e2emv_comm_unique_id": (c_int, [c_void_p, c_void_p]),
    "e2emv_comm_init": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "e2emv_comm_init_file": (c_int, [c_void_p, c_char_p, c_int, c_int, ctypes.c_double, ctypes.POINTER(c_void_p)]),
    "e2emv_comm_destroy": (c_int, [c_void_p, c_void_p]),
    "e2emv_comm_rank": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "e2emv_metric_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "e2emv_metric_allreduce": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "e2emv_profile": (c_int, [c_void_p, c_int]),
    "e2emv_profile_read": (c_int, [c_void_p, ctypes.POINTER(c_float), ctypes.POINTER(c_int64), c_int, c_int]),
    "e2emv_profile_name": (c_char_p, [c_int]),
}

_lib = None
_lock = threading.Lock()
_ctx_lock = threading.Lock()
_contexts = {}
_tokens = iter(range(1, 1 << 62))


def new_owner_token():
    """Process-unique id of a module instance (id() values are re-used after garbage collection)."""
    with _lock:
        return next(_tokens)


class E2EMVError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libe2emv error {code}: {msg}")
        self.code = code


def load_library():
    """dlopen libe2emv.so and attach the prototypes.  Raises if the .so is absent/stale."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing - build it with `python -m e2e_multi_view_matching_amd.build` "
                              "(there is no CPU/PyTorch fallback for the hot path)")
        if os.environ.get("E2EMV_LIBRARY"):  # never silent: a run on a measurement / earlier build says so
            import sys
            print(f"[e2emv] E2EMV_LIBRARY is set: loading {LIB_PATH} instead of the product library", file=sys.stderr)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if lib.e2emv_version() != 1:
            raise ImportError("libe2emv.so ABI version mismatch")
        _lib = lib
        return lib


class Context:
    """One library context per (process, device)."""

    def __init__(self, device):
        self.lib = load_library()
        self.device = int(device)
        h = c_void_p()
        rc = self.lib.e2emv_create(ctypes.byref(h), self.device)
        if rc != OK:
            raise E2EMVError(rc, f"cannot create a context on device {device}: no usable MI355X (gfx950) - "
                                 "the HIP path has no CPU fallback")
        self.h = h
        # The library keeps ONE matcher weight set and ONE SuperPoint weight set per context.  `weights_owner` /
        # `sp_weights_owner` name the module instance + parameter fingerprint they currently hold, so that two models
        # alternating on one device (checkpoint comparison, EMA copy) re-push instead of running on each other's weights.
        self.weights_owner = None
        self.sent_owner = None        # (module, fingerprint) whose tensors e2emv_set_weight handed over last
        self.sp_weights_owner = None
        # held across "push my weights -> select my arithmetic -> enqueue my forward": two threads running two models on
        # one device cannot interleave so that one runs on the other's committed weights (the C-side mutex serialises
        # single calls only)
        self.py_lock = threading.RLock()
        self.train_owner = None       # (module, fingerprint) whose weights e2emv_train_commit folded last
        self.train_generation = 0     # bumped by every forward_train: the context keeps the tape of the LAST one only
        self.default_precision = self.precision()  # E2EMV_PRECISION at creation time (else f32)
        # (r2 / r3: superseded generations, selectable in a measurement build only - E2EMV_LIBRARY=tools/libe2emv_stamps.bin)
        gens = {"r2": 2, "r3": 3, "r4": 4} if os.environ.get("E2EMV_LIBRARY") else {"r4": 4}
        self.f16x2_kernels = gens.get(os.environ.get("E2EMV_F16X2_KERNELS"), 5)
        self.default_f16x2_kernels = self._env_f16x2_kernels = self.f16x2_kernels
        self.forced_precision = None               # set_precision(): explicit process-wide override for models with
        #                                            config["mfma_precision"] = None

    def precision(self):
        v = c_int(0)
        self.check(self.lib.e2emv_get_precision(self.h, ctypes.byref(v)))
        return int(v.value)

    def set_precision(self, precision):
        """Arithmetic of the dense GNN contractions for every model on this device that does not pin its own
        (``config["mfma_precision"]``): PRECISION_F32 / _BF16X3 / _F16X2, "f32", "bf16x3", "f16x2", or None = library default."""
        if isinstance(precision, str):
            precision = PRECISION_NAMES[precision]
        self.forced_precision = precision
        self.call("e2emv_set_precision", self.default_precision if precision is None else precision)

    def set_f16x2_kernels(self, generation=5):
        """f16x2 implementation: 5 = plane activations (gemm_p2 / attention_p2w above 256 keys) with the GEMMs between two
        attentions chained in one launch where that pays (gemm_p2c; the default), 105 = chained wherever the shapes allow,
        4 = one launch per GEMM, 3 = the same with the round-3 attention (attention_p2), 2 = the round-2 kernels (fp32
        activations split inside the consuming kernel)."""
        self.call("e2emv_set_f16x2_kernels", int(generation))
        self.f16x2_kernels = int(generation)

    def set_attention_key_split(self, on=True):
        """attention_p2w: split the items of a half-empty last round of workgroups along the keys (default on)."""
        self.call("e2emv_set_attention_key_split", 1 if on else 0)
        self.attention_key_split = bool(on)

    def select_f16x2_kernels(self, generation=None):
        """The generation every model on this device runs from now on (forward() re-selects `default_f16x2_kernels` on each
        call, so `set_f16x2_kernels` alone lasts one call).  None = back to what E2EMV_F16X2_KERNELS chose at creation."""
        self.default_f16x2_kernels = self._env_f16x2_kernels if generation is None else int(generation)
        self.set_f16x2_kernels(self.default_f16x2_kernels)

    def stats(self, reset=False):
        """{'rescaled_blocks': plane blocks that needed a non-zero tile exponent, 'sinkhorn_bad': Sinkhorn problems with
        non-finite scores, 'sinkhorn_rescued': problems re-solved in the log domain behind the resident kernel (correct
        outputs), 'attention_slow_tiles': (wave, stream, key tile) softmaxes attention_p2w redid on its slow path - a row
        maximum outgrew the running one by more than ~2^9} since the last reset (host-synchronising)."""
        v = (ctypes.c_uint64 * 6)()
        self.call("e2emv_get_stats", v, 6, 1 if reset else 0)
        return {"rescaled_blocks": int(v[0]), "sinkhorn_bad": int(v[1]), "sinkhorn_rescued": int(v[2]), "attention_slow_tiles": int(v[3]),
                "sinkhorn_timeouts": int(v[4]),  # (of the rescued: given up on a wait - contention -, not on range)
                "sinkhorn_rows128_calls": int(v[5])}  # (Sinkhorn calls on 128-row workgroups: fewer rounds for a large batch)

    SINKHORN_KERNELS = {None: 0, "auto": 0, "rows64": 1, "rows128": 2, "stream": 3}

    def set_sinkhorn_kernel(self, kernel=None):
        """Pins the Sinkhorn kernel of every later call on this context: None / "auto" = by shape and batch size (default),
        "rows64" / "rows128" = one resident kernel kind (a problem's result is then independent of its batch neighbours, bit
        for bit), "stream" = the log-domain launch chain.  The E2EMV_SINKHORN variable only sets the value a context starts with."""
        self.call("e2emv_set_sinkhorn_kernel", self.SINKHORN_KERNELS[kernel])
        self.sinkhorn_kernel_pin = self.SINKHORN_KERNELS[kernel]

    def sinkhorn_plan(self, B, M, N, iters):
        """What the launcher would run for this batch now: [] = the log-domain chain, else one dict per resident launch."""
        v = (c_int * 9)()
        self.call("e2emv_sinkhorn_plan", int(B), int(M), int(N), int(iters), v, 9)
        return [{"rows_per_workgroup": int(v[1 + 4 * i]), "problems": int(v[2 + 4 * i]), "resident_problems": int(v[3 + 4 * i]),
                 "rounds": int(v[4 + 4 * i])} for i in range(int(v[0]))]

    def set_split_min_rows(self, min_rows=-1):
        """Calls with fewer keypoint rows than this run the fp32-MFMA kernels even in a split-operand mode
        (-1 = library default, 0 = never)."""
        self.call("e2emv_set_split_min_rows", int(min_rows))
        self.split_min_rows = int(min_rows)

    def mirror_settings(self, other):
        """The arithmetic selection of `other` (the device's main context) on this context - a peer context (config["streams"] =
        2) must compute what the main one would."""
        self.forced_precision = other.forced_precision
        self.default_precision = other.default_precision
        self.default_f16x2_kernels = other.default_f16x2_kernels
        self.call("e2emv_set_sinkhorn_kernel", getattr(other, "sinkhorn_kernel_pin", 0))
        self.call("e2emv_set_attention_key_split", 1 if getattr(other, "attention_key_split", True) else 0)
        want = getattr(other, "split_min_rows", -1)
        if getattr(self, "split_min_rows", -1) != want:
            self.set_split_min_rows(want)

    def check(self, rc):
        if rc != OK:
            raise E2EMVError(rc, self.lib.e2emv_last_error(self.h).decode())

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(self.h, *args))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.e2emv_destroy(self.h)
                self.h = None
        except Exception:
            pass


def context(device=None):
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if isinstance(device, torch.device):
        device = device.index if device.index is not None else torch.cuda.current_device()
    with _ctx_lock:  # creation under a lock: two threads asking for the same device share one context
        ctx = _contexts.get(device)
        if ctx is None:
            ctx = _contexts[device] = Context(device)
    return ctx


_peers = {}


def peer_context(device=None):
    """A SECOND library context on the device (own workspace, own weight copy): what `config["streams"] = 2` runs the second
    half of a batch on, on its own HIP stream, so that one half's launch boundaries are filled by the other half's kernels."""
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if isinstance(device, torch.device):
        device = device.index if device.index is not None else torch.cuda.current_device()
    with _ctx_lock:
        ctx = _peers.get(device)
        if ctx is None:
            ctx = _peers[device] = Context(device)
            ctx.side_stream = torch.cuda.Stream(device=device)
    return ctx


def stream_ptr(device):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def ptr_array(tensors):
    arr = (c_void_p * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return ctypes.cast(arr, _PP), arr


def profile_read(ctx, reset=True):
    ms = (c_float * PROF_SLOTS)()
    n = (c_int64 * PROF_SLOTS)()
    ctx.call("e2emv_profile_read", ms, n, PROF_SLOTS, 1 if reset else 0)
    out = {}
    for i in range(PROF_SLOTS):
        name = ctx.lib.e2emv_profile_name(i).decode()
        if name:
            out[name] = {"ms": float(ms[i]), "launches": int(n[i])}
    return out
