#!/bin/bash
# bench.py A/B on one box: this tree's library vs tools/libe2emv_prev.bin (a build of an earlier commit, E2EMV_LIBRARY); [tag]
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $OUT; tag=${1:-prev}
for rep in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_${tag}_new$rep.json 2> $OUT/bench_${tag}_new$rep.err
E2EMV_LIBRARY=$GRAFT_REPO_ROOT/tools/libe2emv_prev.bin timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_${tag}_old$rep.json 2> $OUT/bench_${tag}_old$rep.err
done
python - $OUT/bench_${tag}_new1.json $OUT/bench_${tag}_old1.json $OUT/bench_${tag}_new2.json $OUT/bench_${tag}_old2.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["families"].items()}, d.get("range"))
    except Exception as e:
        print(f, "FAILED", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
