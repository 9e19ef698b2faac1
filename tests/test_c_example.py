"""The C ABI from plain C: examples/sinkhorn_from_c.c is compiled with gcc against include/e2emv.h and linked with
libe2emv.so (no torch, no Python in the host); on the GPU box it is also run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, lib_built):
    exe = str(tmp_path / "sinkhorn_from_c")
    libdir = os.path.dirname(lib_built)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "sinkhorn_from_c.c"),
           "-o", exe, "-L" + libdir, "-le2emv", "-Wl,-rpath," + libdir, "-lm"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_c_example_compiles_and_links(tmp_path, lib_built):
    assert os.path.exists(_build(tmp_path, lib_built))


@pytest.mark.gpu
def test_c_example_runs(tmp_path, lib_built, gpu):
    r = subprocess.run([_build(tmp_path, lib_built)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "recovered 512 / 512" in r.stdout
