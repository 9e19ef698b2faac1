#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for d in 16 0; do echo DBG=$d; E2EMV_X3_DEBUG=$d timeout 200 python tools/microbench.py --what g3 2>&1 | grep f16x2 | grep -v 8192; done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matcher.py -q -x 2>&1 | tail -2
