#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r2/pytest_gpu5.log
tail -6 gpurun_out/r2/pytest_gpu5.log
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r2/bench5_c2.json 2> gpurun_out/r2/bench5_c2.err
python -c "
import json
d=json.loads(open('gpurun_out/r2/bench5_c2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['dtype'][:20], d['roofline']['kernel'], d['roofline']['frac'], d['other_precision']['value'], d['other_precision']['roofline']['frac'], d['batch1_latency'], d['cpu_baseline']['value'], d['auc_parity_sample'])"
bash tools/gpu_pmc.sh
