#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_matcher.py tests/test_gpu_golden_direct.py tests/test_gpu_random_shapes.py tests/test_gpu_fullsize_properties.py -q -x 2>&1 | tail -8
E2EMV_B3_PLANES=1 timeout 300 python -m pytest tests/test_gpu_matcher.py -q -x 2>&1 | tail -3
for m in f32 bf16x3; do
python bench.py --precision $m --steps 5 --warmup 2 --cpu-pairs 0 --no-alt --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_second']); print({k:v['ms_per_step'] for k,v in d['families'].items()})"
done
