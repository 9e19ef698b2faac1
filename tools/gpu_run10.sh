#!/bin/bash
# full GPU suite + margins with f16x2 as the default arithmetic
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 600 python tools/parity_margins.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/parity_margins.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r2/pytest_gpu10.log
