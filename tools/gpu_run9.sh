#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sinkhorn_resident.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -2
timeout 300 python bench.py --config c5 --cpu-pairs 0 --no-alt --no-latency --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d.get('families',{}).items()})"
