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
           "mvinit.hip", "mvba.hip", "superpoint.hip", "forward.hip", "train.hip", "comm.hip"]
# (superseded kernels are compiled under #ifdef E2EMV_STAMPS inside their files: gemm3.hip's all-planes GEMM, generations 2 / 3 of the
# f16x2 selection - A/B arms of the measurement build, not in the product)
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


def check_register_window(asm_file, kernel, window):
    """Every instance of `kernel` in the device assembly: outside its inline-asm statements the compiler touches no vector or
    accumulation register >= `window` (those hold data placed there by number), and the wave is allocated 256 + 256 registers.
    Returns the number of instances checked; raises RuntimeError otherwise."""
    import re
    txt = open(asm_file).read()
    n = 0
    for m in re.finditer(r"^(_Z\w*%s\w*):" % kernel, txt, re.M):
        end = txt.index(".end_amdhsa_kernel", m.end())
        # the kernel descriptor: from its .amdhsa_kernel header (whatever its length) to .end_amdhsa_kernel
        dstart = txt.rfind(".amdhsa_kernel " + m.group(1), m.end(), end)
        if dstart < 0:
            raise RuntimeError(f"{m.group(1)}: no .amdhsa_kernel header in front of .end_amdhsa_kernel")
        body, desc = txt[m.end():dstart], txt[dstart:end]
        if not re.search(r"\.amdhsa_next_free_vgpr\s+512\b", desc) or not re.search(r"\.amdhsa_accum_offset\s+256\b", desc):
            raise RuntimeError(f"{m.group(1)}: the wave is not allocated 256 + 256 registers")
        inasm = False
        for line in body.split("\n"):
            if "#ASMSTART" in line:
                inasm = True
            elif "#ASMEND" in line:
                inasm = False
            elif not inasm:
                code = line.split(";")[0]
                regs = [int(x) for x in re.findall(r"\b[va](\d+)\b", code)] + [int(x) for x in re.findall(r"\b[va]\[\d+:(\d+)\]", code)]
                if regs and max(regs) >= window:
                    raise RuntimeError(f"{m.group(1)}: compiler-generated code touches a register outside its window: {line.strip()}")
        n += 1
    if n == 0:
        raise RuntimeError(f"{kernel}: not found in {asm_file}")
    return n


def _build(hipcc, objdir, lib, extra, verbose):
    os.makedirs(objdir, exist_ok=True)
    # gemm_p2c.hip hands tiles from one wave to another of the SAME workgroup through the CU's vector L1 (stores retired + barrier):
    # threadgroup-split mode would put those waves on different CUs / L1s
    if any("tgsplit" in f for f in FLAGS + list(extra)):
        raise RuntimeError("libe2emv is built for CU mode: -mtgsplit breaks the in-workgroup hand-off of gemm_p2c.hip")
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        cwd = None
        if src == "sinkhorn.hip":
            # sinkhorn_resident128 / sinkhorn_resident2k keep most of their data in registers they address by number, outside the
            # window their amdgpu_num_vgpr attribute leaves to the compiler.  The device assembly hipcc ASSEMBLES into this very
            # object is kept (-save-temps: one compile, its own intermediate .s - not a second -S compile that could diverge) and
            # checked below (check_register_window)
            cwd = os.path.join(objdir, "sinkhorn_temps")
            os.makedirs(cwd, exist_ok=True)
            cmd.insert(1, "-save-temps=obj")
            cmd[-1] = os.path.join(cwd, "sinkhorn.o")
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=cwd)))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
        objs.append(obj)
    import shutil
    temps = os.path.join(objdir, "sinkhorn_temps")
    shutil.copyfile(os.path.join(temps, "sinkhorn.o"), os.path.join(objdir, "sinkhorn.o"))  # the object that is linked
    asm_file = os.path.join(objdir, "sinkhorn.s")
    shutil.copyfile(os.path.join(temps, "sinkhorn-hip-amdgcn-amd-amdhsa-gfx950.s"), asm_file)  # the assembly it was made from
    check_register_window(asm_file, "sinkhorn_resident128", 56)
    check_register_window(asm_file, "sinkhorn_resident2k", 56)
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
