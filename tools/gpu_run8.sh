#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 200 python tools/microbench.py --what g3 2>&1 | grep f16x2
for d in 1 2; do echo DBG=$d; E2EMV_X3_DEBUG=$d timeout 100 python tools/microbench.py --what g3 2>&1 | grep f16x2 | grep -E "65536   512   512|32768"; done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matcher.py tests/test_gpu_golden_direct.py tests/test_gpu_random_shapes.py -q -x 2>&1 | tail -8
for m in f16x2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-latency --precision $m 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], {k:round(v['ms_per_step'],3) for k,v in d.get('families',{}).items()})"
done
