#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_sinkhorn_resident.py -x -q 2>&1 | tail -30 > gpurun_out/r2/pytest_sk.log
tail -12 gpurun_out/r2/pytest_sk.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r2/pytest_gpu2.log
tail -12 gpurun_out/r2/pytest_gpu2.log
for mode in resident stream; do
  if [ $mode = stream ]; then export E2EMV_SINKHORN=stream; else unset E2EMV_SINKHORN; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt > gpurun_out/r2/bench2_c2_$mode.json 2> gpurun_out/r2/bench2_c2_$mode.err
  timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --cpu-pairs 0 --no-alt > gpurun_out/r2/bench2_c4_$mode.json 2> gpurun_out/r2/bench2_c4_$mode.err
  python - <<PY
import json
for c in ("c2","c4"):
    try:
        d=json.loads(open("gpurun_out/r2/bench2_%s_$mode.json" % c).read().strip().splitlines()[-1])
        print("$mode", c, d["value"], d["ms_per_step"], d["families"]["sinkhorn"], d.get("batch1_latency"))
    except Exception as e:
        print("$mode", c, "FAILED", e)
PY
done
