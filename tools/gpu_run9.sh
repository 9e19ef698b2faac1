#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python bench.py --steps 10 --warmup 3 --no-latency --cpu-pairs 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], 'bare', d.get('ms_per_step_without_event_brackets'), d['roofline']['avg_launch_ms'], d['roofline']['frac'], {k:(round(v['ms_per_step'],3), v['launches_per_step']) for k,v in d.get('families',{}).items()}, [(a['mode'],a['value']) for a in d['other_precisions']])"
timeout 600 python -m pytest tests/test_gpu_errors.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -2
