#!/bin/bash
# bench.py A/B on one box: default kernels vs E2EMV_F16X2_KERNELS=r3 (round-3 attention); tools/bench_ab.sh [tag]
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $OUT; tag=${1:-ab}
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_${tag}.json 2> $OUT/bench_${tag}.err
E2EMV_F16X2_KERNELS=r3 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_${tag}_r3.json 2> $OUT/bench_${tag}_r3.err
python - $OUT/bench_${tag}.json $OUT/bench_${tag}_r3.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["families"].items()}, d.get("range"))
    except Exception as e:
        print(f, "FAILED", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
