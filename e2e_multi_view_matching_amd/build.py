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


def check_register_window(asm_file, kernel, window):
    """Every instance of `kernel` in the device assembly: outside its inline-asm statements the compiler touches no vector or
    accumulation register >= `window` (those hold data placed there by number), and the wave is allocated 256 + 256 registers.
    Returns the number of instances checked; raises RuntimeError otherwise."""
    import re
    txt = open(asm_file).read()
    n = 0
    for m in re.finditer(r"^(_Z\w*%s\w*):" % kernel, txt, re.M):
        end = txt.index(".end_amdhsa_kernel", m.end())
        body, desc = txt[m.end():end], txt[end - 4000:end]
        if "amdhsa_next_free_vgpr 512" not in desc or "amdhsa_accum_offset 256" not in desc:
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
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    # sinkhorn_resident128 keeps most of its data in registers it addresses by number, outside the window its
    # amdgpu_num_vgpr attribute leaves to the compiler: the device assembly is checked for that (check_register_window)
    asm_file = os.path.join(objdir, "sinkhorn.s")
    asm_proc = subprocess.Popen([hipcc] + FLAGS + extra + ["--cuda-device-only", "-S", os.path.join(CSRC, "sinkhorn.hip"), "-o", asm_file],
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
        objs.append(obj)
    out, _ = asm_proc.communicate()
    if asm_proc.returncode != 0:
        raise RuntimeError(f"hipcc -S failed on sinkhorn.hip:\n{out}")
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
