#!/bin/bash
# end-of-round measurement set: PMC passes + kernel traces (c2, c4 in the default f16x2 mode), c5 kernel trace, bench lines
# of c2 / c4 / c5, one-rank RCCL init of the distributed branch, smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_pmc.sh
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_c5 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 2 --warmup 1 --cpu-pairs 0 --no-alt --no-latency --no-profile > $OUT/kt_c5.log 2>&1)
db=$(find /tmp/kt_c5 -name '*.db' | head -1); [ -n "$db" ] && python profiles/summarize_rocpd.py "$db" > $OUT/r2_kernel_stats_c5_f16x2.md 2>&1
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json   # so that the bench lines below carry roofline.traffic
timeout 400 python bench.py > $OUT/bench_c2_final.json 2> $OUT/bench_c2_final.err
timeout 300 python bench.py --config c4 --cpu-pairs 0 > $OUT/bench_c4_final.json 2> $OUT/bench_c4_final.err
timeout 300 python bench.py --config c5 --cpu-pairs 0 > $OUT/bench_c5_final.json 2> $OUT/bench_c5_final.err
E2EMV_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_c2_dist1.json 2> $OUT/bench_c2_dist1.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
for f in c2_final c4_final c5_final c2_dist1; do python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', d['value'], d['ms_per_step'], d['n_gpus'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('traffic'), d.get('families'), [ (a['mode'], a['value']) for a in d.get('other_precisions',[])], d.get('batch1_latency'), (d.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('$f FAILED', e); print(open('$OUT/bench_$f.err').read()[-1500:])
PY
done
tail -2 $OUT/smoke.log
