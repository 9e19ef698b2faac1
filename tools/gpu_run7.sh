#!/bin/bash
# A/B: library built with / without packed-fp32 VALU ops (guide: v_pk_*_f32 beside MFMAs is an anti-lever)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
L=e2e_multi_view_matching_amd/libe2emv.so
cp $L /tmp/base.so
for v in base nopk3 nopkall; do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp gpu_variants/libe2emv_$v.so $L; fi
  echo "== $v"
  timeout 200 python tools/microbench.py --what g3,a3 2>&1 | grep -v all-planes | tail -12
  for m in bf16x3 f32; do
  E2EMV_PRECISION=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $m bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], {k:round(v['ms_per_step'],3) for k,v in d.get('families',{}).items()} if 'families' in d else '')"
  done
done
cp /tmp/base.so $L
