#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for nw in 4 8; do echo NW=$nw; E2EMV_A3_NW=$nw timeout 200 python tools/microbench.py --what a3 2>&1 | grep f16x2; done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matcher.py tests/test_gpu_golden_direct.py tests/test_gpu_random_shapes.py tests/test_gpu_round2.py -q -x 2>&1 | tail -2
E2EMV_A3_NW=8 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matcher.py -q -x 2>&1 | tail -2
