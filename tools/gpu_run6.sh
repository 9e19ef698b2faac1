#!/bin/bash
# A/B of the two bf16x3 GEMM kernels (E2EMV_X3_V=1 committed, =2 double-buffered + AGPR accumulators)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
for v in 1 2; do
  echo "== X3_V=$v"
  E2EMV_X3_V=$v timeout 200 python tools/microbench.py --what g3 2>&1 | tail -8
done
E2EMV_X3_V=2 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matcher.py tests/test_gpu_golden_direct.py -q -x 2>&1 | tail -5
E2EMV_X3_V=2 timeout 300 python bench.py --steps 10 --warmup 3 --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V2 bench', d['value'], d['ms_per_step'], d['roofline'])"
