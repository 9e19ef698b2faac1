"""The C ABI from plain C: examples/sinkhorn_from_c.c is compiled with gcc against include/e2emv.h and linked with
libe2emv.so (no torch, no Python in the host); on the GPU box it is also run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, lib_built, name="sinkhorn_from_c"):
    exe = str(tmp_path / name)
    libdir = os.path.dirname(lib_built)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".c"),
           "-o", exe, "-L" + libdir, "-le2emv", "-Wl,-rpath," + libdir, "-lm"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_c_example_compiles_and_links(tmp_path, lib_built):
    assert os.path.exists(_build(tmp_path, lib_built))
    assert os.path.exists(_build(tmp_path, lib_built, "metric_gather_from_c"))


@pytest.mark.gpu
def test_c_example_runs(tmp_path, lib_built, gpu):
    r = subprocess.run([_build(tmp_path, lib_built)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "recovered 512 / 512" in r.stdout


@pytest.mark.gpu
def test_c_metric_gather_runs_as_the_only_rank(tmp_path, lib_built, gpu):
    """examples/metric_gather_from_c.c: the library's RCCL communicator, file bootstrap, all-gather + all-reduce from plain C
    (one rank here; `for r in 0..7` on an 8-GPU node - the command is in the file's header)."""
    exe = _build(tmp_path, lib_built, "metric_gather_from_c")
    r = subprocess.run([exe, "0", "1", str(tmp_path / "comm_id")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "gathered 4 values, 0 wrong" in r.stdout
