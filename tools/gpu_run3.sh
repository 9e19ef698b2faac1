#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r2/pytest_gpu3.log
tail -12 gpurun_out/r2/pytest_gpu3.log
for c in c2 c4 c5; do
  timeout 300 python bench.py --config $c --steps 5 --warmup 2 --cpu-pairs 0 > gpurun_out/r2/bench3_$c.json 2> gpurun_out/r2/bench3_$c.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2/bench3_$c.json").read().strip().splitlines()[-1])
    print("$c", d["value"], d["ms_per_step"], d["families"], d.get("batch1_latency"), d.get("other_precision"))
except Exception as e:
    print("$c", "FAILED", e)
PY
done
