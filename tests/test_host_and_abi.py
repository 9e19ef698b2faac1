"""CPU: the C-ABI library loads and exports what include/e2emv.h declares; host logic; no CPU fallback."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib_built):
    from e2e_multi_view_matching_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "e2emv.h")).read()
    declared = set(re.findall(r"\b(e2emv_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"e2emv_ctx"}
    assert len(declared) >= 20
    lib = ctypes.CDLL(lib_built)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/e2emv.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load_library().e2emv_version() == 1


def test_dynamic_symbol_table_is_exactly_the_header(lib_built):
    """-fvisibility=hidden + the linker version script of build.py: no mangled C++ internals, no kernel host stubs."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", lib_built], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    hdr = open(os.path.join(ROOT, "include", "e2emv.h")).read()
    declared = set(re.findall(r"\b(e2emv_[a-z0-9_]+)\s*\(", hdr)) - {"e2emv_ctx"}
    assert exported == declared, sorted(exported ^ declared)[:10]


def test_no_measurement_binaries_next_to_the_product():
    pkg = os.path.join(ROOT, "e2e_multi_view_matching_amd")
    assert [f for f in os.listdir(pkg) if f.endswith(".so")] == ["libe2emv.so"]


def test_sinkhorn128_keeps_the_compiler_inside_its_register_window(lib_built):
    """sinkhorn_resident128 holds 24 of a wave's 32 rows in registers it addresses by number (v64 .. v255, a64 .. a255; v56 .. v63
    are its temporaries); amdgpu_num_vgpr(56) confines hipcc to v0 .. v55 / a0 .. a55.  The device assembly the build wrote is the
    proof: no compiler-generated instruction of either instance touches a register outside that window, and the wave is allocated
    all 512 registers.  (build.py runs the same check on every build.)"""
    from e2e_multi_view_matching_amd import build
    asm = os.path.join(ROOT, "e2e_multi_view_matching_amd", "build", "sinkhorn.s")
    if not os.path.exists(asm) or os.path.getmtime(asm) < os.path.getmtime(os.path.join(build.CSRC, "sinkhorn.hip")):
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + build.FLAGS + ["--cuda-device-only", "-S", os.path.join(build.CSRC, "sinkhorn.hip"), "-o", asm],
                       check=True, capture_output=True)
    assert build.check_register_window(asm, "sinkhorn_resident128", 56) == 2  # <true> (full tiles) and <false> (ragged)
    txt = open(asm).read()
    body = txt[txt.index("sinkhorn_resident128ILb1E"):]
    body = body[:body.index(".end_amdhsa_kernel")]
    mine = [ln for ln in body.splitlines() if re.search(r"\b(v_accvgpr_(read|write)_b32|v_mov_b32|v_pk_(fma|mul)_f32).*[va]\[(0x[0-9a-f]+|\d+)[+\]]", ln)]
    assert len(mine) == 384 + 2 * (12 * 8 + 12 * 16)  # the statements that DO address them: 384 writes; per pass 8 packed operations per vector row, 16 reads per accumulation row
    with pytest.raises(RuntimeError, match="outside its window"):
        build.check_register_window(asm, "sinkhorn_resident128", 40)


def test_struct_layouts_match_the_header():
    from e2e_multi_view_matching_amd import _lib
    assert ctypes.sizeof(_lib.ModelDesc) == 4 * (3 + 8 + 1 + 64 + 1)
    assert ctypes.sizeof(_lib.ForwardDesc) == 4 * (7 + 8 + 8 + 8)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_cpu_fallback(lib_built):
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd import _lib
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    h = ctypes.c_void_p()
    assert _lib.load_library().e2emv_create(ctypes.byref(h), 0) == _lib.EHIP
    model = E.MultiViewMatcher({"GNN_layers": ["self"]}).eval()
    with pytest.raises(RuntimeError, match="MI355X"):
        model(make_tuples(batch=1, n_kpts=16))
    with pytest.raises(RuntimeError):
        E.log_optimal_transport(torch.zeros(1, 4, 4), 1.0, 3)
    with pytest.raises(RuntimeError):
        E.estimate_relative_pose_w8pt(torch.zeros(1, 9, 2), torch.zeros(1, 9, 2), torch.eye(3)[None], torch.eye(3)[None],
                                      torch.ones(1, 9))
    # the reference's early-outs need no device
    assert E.estimate_relative_pose_w8pt(torch.zeros(1, 7, 2), torch.zeros(1, 7, 2), None, None, None) == (None, None)
    assert E.run_weighted_8_point({}, {}, 0, 1) == (None, None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "e2e_multi_view_matching_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
    code = "import sys; import e2e_multi_view_matching_amd; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_module_contract_of_the_reference_callers():
    """state_dict names / load_state_dict(strict=False) / conf_mlp parameter filter (helpers.py:47-71)."""
    from e2e_multi_view_matching_amd import MultiViewMatcher, SuperGlue
    assert SuperGlue is MultiViewMatcher
    m = MultiViewMatcher({"multi_frame_matching": True, "tuple_size": 5, "conf_mlp": True, "GNN_layers": (["self"] + ["cross"] * 3) * 7})
    sd = m.state_dict()
    for k in ("kenc.encoder.0.weight", "kenc.encoder.1.running_mean", "kenc.encoder.12.bias", "gnn.layers.27.attn.proj.2.weight",
              "gnn.layers.0.attn.merge.bias", "gnn.layers.3.mlp.1.running_var", "gnn.layers.3.mlp.3.weight", "final_proj.weight",
              "bin_score", "conf_mlp.0.weight", "conf_mlp.3.bias"):
        assert k in sd, k
    assert sd["gnn.layers.0.mlp.0.weight"].shape == (512, 512, 1) and sd["kenc.encoder.0.weight"].shape == (32, 3, 1)
    m2 = MultiViewMatcher()  # eval_pairs-style default: 18 layers, no conf head
    assert len(m2.config["GNN_layers"]) == 18
    n_params = sum(p.numel() for p in m2.parameters())
    assert n_params == 12_023_297
    wrapped = torch.nn.DataParallel(m2)
    ck = {("module." + k): v for k, v in sd.items()}
    missing, unexpected = wrapped.load_state_dict(ck, strict=False)
    assert any("conf_mlp" in k for k in unexpected) and not any("kenc" in k for k in missing)
    conf_params = [n for n, _ in m.named_parameters() if "conf_mlp" in n]
    assert len(conf_params) == 6
    wrapped.module.config["full_output"] = True
    assert float(m2.kenc.encoder[-1].bias.detach().abs().sum()) == 0.0 and float(m2.bin_score.detach()) == 1.0


def test_weight_fingerprint_sees_object_replacement_inside_submodules():
    """the cached walk must not hold tensor OBJECTS - every way of swapping one below the top-level module has to
    change the fingerprint (else the library keeps running on the weights it committed before)."""
    import torch.nn as nn
    import e2e_multi_view_matching_amd as E
    m = E.MultiViewMatcher({"GNN_layers": ["self", "cross"], "conf_mlp": True}).eval()
    seen = [m._fingerprint()]

    def changed(what):
        fp = m._fingerprint()
        assert fp != seen[-1], what
        assert fp == m._fingerprint(), what  # and it is stable between calls
        seen.append(fp)

    assert m._fingerprint() == seen[0]
    with torch.no_grad():
        m.gnn.layers[0].mlp[0].weight.add_(1.0)
    changed("in-place update")
    m.gnn.load_state_dict({k: v.clone() for k, v in m.gnn.state_dict().items()}, assign=True)
    changed("submodule.load_state_dict(assign=True)")
    m.kenc.double()
    changed("submodule._apply re-assigning parameters and BN buffers")
    m.kenc.float()
    changed("back to float")
    m.final_proj.weight = nn.Parameter(m.final_proj.weight.detach().clone())
    changed("submodule.weight = nn.Parameter(...)")
    m.gnn.layers[1] = type(m.gnn.layers[1])(m.config["descriptor_dim"])
    changed("a replaced submodule")
    m.kenc.encoder[1].running_mean = m.kenc.encoder[1].running_mean.clone()
    changed("a re-assigned BatchNorm buffer")
    # a None slot of a NESTED module filled later, and a submodule added to a nested module - neither changes the
    # number of slots of any module that existed when the walk was cached
    m.final_proj.bias = None
    changed("a parameter slot set to None")
    m.final_proj.bias = nn.Parameter(torch.zeros(m.config["descriptor_dim"]))
    changed("a None slot of a nested module filled")
    m.gnn.layers[0].extra = nn.Linear(2, 2)
    changed("a submodule added to a nested module")
    del m.gnn.layers[0].extra
    changed("... and removed again")
    keys = [k for k, v in m.state_dict().items() if v.dtype.is_floating_point]
    assert [k for k, *_ in m._fingerprint()] == keys  # the same tensors, in state_dict order, that _send_weights uploads


def test_synthetic_generator_is_seeded_and_consistent():
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    a, b = make_tuples(batch=2, tuple_size=3, n_kpts=64, seed=5), make_tuples(batch=2, tuple_size=3, n_kpts=64, seed=5)
    assert all(torch.equal(a[k], b[k]) for k in a if torch.is_tensor(a[k]))
    assert a["descriptors1"].shape == (2, 256, 64) and a["keypoints2"].shape == (2, 64, 2)
    assert float((a["descriptors0"].norm(dim=1) - 1).abs().max()) < 1e-5
    gt = a["gt_matches0_0_2"]
    b0, i0 = torch.nonzero(gt >= 0, as_tuple=True)
    d0 = a["descriptors0"][b0, :, i0]
    d2 = a["descriptors2"][b0, :, gt[b0, i0]]
    assert float((d0 * d2).sum(1).min()) > 0.5  # shared points carry similar descriptors
    # the relative pose composes: T_0to2 = T_1to2 @ T_0to1
    assert float((a["T_0to2"] - a["T_1to2"] @ a["T_0to1"]).abs().max()) < 1e-5


def test_algorithmic_work_formulas():
    from oracle.matcher import dense_flops_per_tuple
    from oracle.sinkhorn import sinkhorn_bytes_per_pair
    f = dense_flops_per_tuple(2, 1024, 256, ["self", "cross"] * 9)
    assert abs(f / 1e9 - 88.22) < 0.05  # 88.22 GFLOP / pair at config 2
    assert abs(sinkhorn_bytes_per_pair(1024, 100) / 1e6 - 848.9) < 0.1
    f5 = dense_flops_per_tuple(5, 1024, 256, ["self", "cross"] * 9)
    assert abs(f5 / 1e9 - 369.5) < 0.5


def test_shard_range_partitions():
    from e2e_multi_view_matching_amd.distributed import shard_range
    for n in (0, 1, 7, 64, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_generated_sinkhorn_rows_header_is_the_generators_output():
    """csrc/sinkhorn128_rows.h (550 lines of asm statements on registers addressed by number) is generated: the committed file must
    be exactly what tools/gen_sk128_asm.py writes (a hand edit of either would silently diverge)."""
    import runpy
    ns = runpy.run_path(os.path.join(ROOT, "tools", "gen_sk128_asm.py"), run_name="gen_sk128_asm_check")
    assert open(ns["path"]).read() == ns["out"]
