"""Builds libe2emv.so (the HIP/C-ABI library) in-tree with hipcc for gfx950.

``python -m e2e_multi_view_matching_amd.build`` or ``build_library()``; cross-compiles
without a GPU.  The .so stays next to this file (git-ignored, travels with gpurun).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libe2emv.so")
SOURCES = ["ctx.hip", "gemm.hip", "attention.hip", "gemm3.hip", "gemm_x3.hip", "gemm_h2.hip", "gemm_p2.hip", "gemm_p2c.hip", "attention_p2.hip", "attention_p2w.hip", "p2_tools.hip", "attention3.hip", "split3_api.hip", "sinkhorn.hip", "pose.hip", "ba2view.hip", "gtmatch.hip",
           "mvinit.hip", "mvba.hip", "superpoint.hip", "forward.hip", "train.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-ffp-contract=fast", "-Wno-unused-result"]


def _stamp():
    h = hashlib.sha256()
    # (sources only: the stamp file itself lives in this directory - hashing it made the stamp depend on the build before)
    for f in sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) and os.path.isfile(os.path.join(CSRC, f))) + ["../../include/e2emv.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_library(force=False, verbose=False, defines=(), out=None):
    """`defines` / `out`: a measurement build beside the product library (tools/p2_stamps.py builds tools/libe2emv_stamps.bin with
    -DE2EMV_STAMPS: in-kernel timestamps and ablation variants that the product library does not contain)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if defines or out:
        # (measurement builds live under tools/*.bin - git-ignored, never next to the product library)
        return _build(hipcc, os.path.join(HERE, "build_" + os.path.basename(out).split(".")[0]), out, ["-D" + d for d in defines], verbose)
    stamp_file = os.path.join(HERE, "csrc", ".build_stamp")
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    _build(hipcc, os.path.join(HERE, "build"), LIB, [], verbose)
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return LIB


def _build(hipcc, objdir, lib, extra, verbose):
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
        objs.append(obj)
    # the dynamic symbol table is the C ABI of include/e2emv.h and nothing else (hipcc gives kernel host stubs default
    # visibility whatever -fvisibility says: the version script takes them and every C++ internal out)
    vs = os.path.join(objdir, "e2emv.map")
    with open(vs, "w") as fh:
        fh.write("{ global: e2emv_*; local: *; };\n")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-Wl,-rpath,/opt/rocm/lib", "-Wl,--version-script=" + vs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return lib


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
