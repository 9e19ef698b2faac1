#!/bin/bash
# PMC passes over attention_p2w / attention_p2 (tools/aw_ablate.py --pmc); summaries under gpurun_out/r4/
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/awp_$i -- python $GRAFT_REPO_ROOT/tools/aw_ablate.py --pmc > $OUT/aw_pmc_$i.log 2>&1)
  f=$(find /tmp/awp_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:48]
    if 'attention_p2' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k); [print('   ', c, f'{v:.4g}') for c, v in sorted(d.items())]
PY
  i=$((i+1))
done 2>&1 | tee $OUT/aw_pmc_summary2.txt
