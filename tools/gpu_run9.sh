#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for nj in 2 4; do echo NJ=$nj; E2EMV_H2_NJ=$nj timeout 200 python tools/microbench.py --what g3 2>&1 | grep f16x2 | grep -v 8192; done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matcher.py tests/test_gpu_golden_direct.py tests/test_gpu_random_shapes.py -q -x 2>&1 | tail -2
E2EMV_H2_NJ=2 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matcher.py -q -x 2>&1 | tail -2
timeout 300 python bench.py --steps 10 --warmup 3 --no-latency --no-alt --cpu-pairs 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d.get('families',{}).items()})"
