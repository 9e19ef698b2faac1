#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r2/pytest_gpu_final.log
bash tools/gpu_run11.sh
