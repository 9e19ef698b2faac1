#!/bin/bash
# One parameterised GPU-box script (run through gpurun): tools/gpu.sh <stage> [...]; output under gpurun_out/r6/.
#   planes   per-kernel parity of the plane kernels + matcher parity + micro-benchmarks + bench A/B (plane vs round-2 kernels)
#   tests    the whole -m gpu suite
#   bench    bench.py lines (c2 default, c4, c5)
#   one      one short bench line with its roofline / sinkhorn_bound objects printed
#   kt       rocprofv3 kernel trace of one config / mode
#   prof     rocprofv3 kernel trace + the three PMC passes of config c2
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6
mkdir -p $OUT
export TMPDIR=/tmp
stage=${1:-tests}
show() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'], 4), d.get('families'),
              [(a['mode'], a['value']) for a in d.get('other_precisions', [])], d.get('batch1_latency'), d.get('range'))
    except Exception as e:
        print(f, 'FAILED', e)
        try: print(open(f.replace('.json', '.err')).read()[-1500:])
        except Exception: pass
PY
}
B="--steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency"
case $stage in
chain)
  # round 5: the chained GEMMs - bit-identity with the per-GEMM launches, the parity suites that run through the shared core,
  # then A/B bench lines on this box: generation 5 (chain), generation 4 (same library), the round-4 library (tools/libe2emv_prev.bin)
  timeout 600 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -15 | tee $OUT/chain_tests.log
  timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_matcher.py tests/test_gpu_range.py tests/test_gpu_round4.py -x -q 2>&1 | tail -8 | tee $OUT/chain_suites.log
  for rep in 1 2; do
    timeout 300 python bench.py $B > $OUT/bench_g5_$rep.json 2> $OUT/bench_g5_$rep.err
    E2EMV_F16X2_KERNELS=r4 timeout 300 python bench.py $B > $OUT/bench_g4_$rep.json 2> $OUT/bench_g4_$rep.err
    [ -f tools/libe2emv_prev.bin ] && E2EMV_LIBRARY=$GRAFT_REPO_ROOT/tools/libe2emv_prev.bin E2EMV_F16X2_KERNELS=r4 timeout 300 python bench.py $B > $OUT/bench_prev_$rep.json 2> $OUT/bench_prev_$rep.err
  done
  show $OUT/bench_g5_1.json $OUT/bench_g4_1.json $OUT/bench_prev_1.json $OUT/bench_g5_2.json $OUT/bench_g4_2.json $OUT/bench_prev_2.json
  python -c "
import json; d=json.loads(open('$OUT/bench_g5_2.json').read().strip().splitlines()[-1]); print(json.dumps(d['roofline'].get('per_kernel'), indent=1))"
  args="--steps 2 --warmup 1 --cpu-pairs 0 --no-alt --no-latency --no-profile"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_chain -- python $GRAFT_REPO_ROOT/bench.py $args > $OUT/kt_chain.log 2>&1)
  db=$(find /tmp/kt_chain -name '*.db' | head -1)
  [ -n "$db" ] && python profiles/summarize_rocpd.py "$db" > $OUT/r6_kernel_stats_c2_f16x2_chain.md 2>&1
  head -24 $OUT/r6_kernel_stats_c2_f16x2_chain.md
  ;;
planes)
  timeout 900 python -m pytest tests/test_gpu_planes.py -x -q 2>&1 | tail -15 | tee $OUT/planes_tests.log
  timeout 900 python -m pytest tests/test_gpu_matcher.py tests/test_gpu_golden_direct.py -x -q 2>&1 | tail -15 | tee $OUT/planes_matcher.log
  timeout 600 python tools/microbench.py --what p2 2>&1 | tee $OUT/microbench_p2.log
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_p2.json 2> $OUT/bench_p2.err
  E2EMV_F16X2_KERNELS=r2 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_r2k.json 2> $OUT/bench_r2k.err
  show $OUT/bench_p2.json $OUT/bench_r2k.json
  ;;
aw)
  # attention_p2w: parity of the attention kernels on planes, micro-benchmark against attention_p2, one bench line per kernel
  timeout 900 python -m pytest tests/test_gpu_planes.py -x -q -k attention 2>&1 | tail -15 | tee $OUT/aw_tests.log
  timeout 600 python tools/microbench.py --what ap2 2>&1 | tee $OUT/microbench_ap2.log
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_aw.json 2> $OUT/bench_aw.err
  E2EMV_F16X2_KERNELS=r3 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_ap8.json 2> $OUT/bench_ap8.err
  show $OUT/bench_aw.json $OUT/bench_ap8.json
  ;;
awabl)
  # attention_p2w ablations (measurement build) + a PMC pass over attention_p2w and attention_p2
  timeout 600 python tools/aw_ablate.py 2>&1 | grep -v amdgpu.ids | tee $OUT/aw_ablate.log
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/aw_pmc -- python $GRAFT_REPO_ROOT/tools/aw_ablate.py --pmc > $OUT/aw_pmc.log 2>&1)
  f=$(find /tmp/aw_pmc -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee $OUT/aw_pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:60]
    if 'attention_p2' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
for k, d in acc.items():
    print(k); [print('   ', c, f'{v:.4g}') for c, v in sorted(d.items())]
PY
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/aw_pmc2 -- python $GRAFT_REPO_ROOT/tools/aw_ablate.py --pmc >> $OUT/aw_pmc.log 2>&1)
  f=$(find /tmp/aw_pmc2 -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $OUT/aw_pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r['Kernel_Name'][:60]
    if 'attention_p2' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k); [print('   ', c, f'{v:.4g}') for c, v in sorted(d.items())]
PY
  tail -5 $OUT/aw_pmc.log
  ;;
stamps)
  timeout 900 python tools/p2_stamps.py 2>&1 | tee $OUT/p2_stamps.log
  ;;
cstamps)
  timeout 1200 python tools/p2c_stamps.py "${@:2}" 2>&1 | tee $OUT/p2c_stamps.log
  ;;
ab)
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_p2.json 2> $OUT/bench_p2.err
  E2EMV_F16X2_KERNELS=r2 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_r2k.json 2> $OUT/bench_r2k.err
  show $OUT/bench_p2.json $OUT/bench_r2k.json
  ;;
sk128)
  # sinkhorn_resident128: its tests, then ms per call (100 iterations incl. the final sweep) against the 64-row kernel, alternating
  timeout 900 python -m pytest tests/test_gpu_sinkhorn_resident.py -x -q 2>&1 | tail -8 | tee $OUT/sk128_tests.log
  timeout 300 python - <<'PY' 2>&1 | tee $OUT/sk128_time.log
import os, time, torch
import e2e_multi_view_matching_amd as E
from e2e_multi_view_matching_amd import _lib
def t(s, it=100, reps=20):
    for _ in range(3): E.log_optimal_transport(s, 1.0, it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): E.log_optimal_transport(s, 1.0, it)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for B, M, N in ((32, 1024, 1024), (80, 1024, 1024), (48, 1024, 1024), (40, 1024, 1024), (80, 2048, 2048), (12, 2048, 2048), (20, 2048, 2048)):
    s = torch.randn(B, M, N, device="cuda") * 3
    r = []
    for rep in range(2):
        for mode in ("rows64", "rows128", None):
            _lib.context().set_sinkhorn_kernel(mode)
            r.append(t(s))
    _lib.context().set_sinkhorn_kernel(None)
    print(f"{B} x {M} x {N}: rows64 {r[0]:.3f} / {r[3]:.3f} ms   rows128 / 2k {r[1]:.3f} / {r[4]:.3f} ms   library's plan {r[2]:.3f} / {r[5]:.3f} ms")
PY
  ;;
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/tests.log
  ;;
bench)
  timeout 400 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
  timeout 300 python bench.py --config c4 --cpu-pairs 0 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
  timeout 300 python bench.py --config c5 --cpu-pairs 0 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
  show $OUT/bench_c2.json $OUT/bench_c4.json $OUT/bench_c5.json
  ;;
final)
  # end-of-round evidence in ONE call on ONE box: kernel trace + three PMC passes of configs[1] (the traffic file the bench line
  # reads is the one these passes just wrote), the three bench lines, kernel traces of configs[3] / the configs[4] share, and the
  # at::native count of a trace with three times the steps
  bash tools/gpu.sh prof c2 f16x2
  cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
  # (round 6: counter passes of configs[3] and the configs[4] share on the current kernels too - each prof call extends the traffic file)
  bash tools/gpu.sh prof c4 f16x2 > /dev/null
  cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
  bash tools/gpu.sh prof c5 f16x2 > /dev/null
  cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
  bash tools/gpu.sh bench
  args="--steps 6 --warmup 1 --cpu-pairs 0 --no-alt --no-latency --no-profile"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_c2_steps6 -- python $GRAFT_REPO_ROOT/bench.py $args > $OUT/kt_c2_steps6.log 2>&1)
  db=$(find /tmp/kt_c2_steps6 -name '*.db' | head -1)
  [ -n "$db" ] && python profiles/summarize_rocpd.py "$db" > $OUT/r6_kernel_stats_c2_f16x2_steps6.md 2>&1
  # the Sinkhorn kernels against each other (product library), then their per-phase timestamps (measurement build)
  bash tools/gpu.sh sk128 > /dev/null 2>&1
  timeout 400 python tools/skr_timing.py --rows128 2>&1 | grep -v amdgpu.ids > $OUT/skr_rows128.log
  { echo "# ms per call of 100 iterations incl. the final sweep, one box, alternating (tools/gpu.sh sk128; rows64 = E2EMV_SINKHORN=rows64: the 64-row"; echo "# (32-row at 2048 columns) workgroups only, rows128 / 2k = the kernels with the couplings in registers addressed by number for the whole batch,"; echo "# library's plan = their full rounds + the remainder on whichever is cheaper: what a call gets by default)"; grep " x " $OUT/sk128_time.log; echo; echo "# tests/test_gpu_sinkhorn_resident.py on the same box:"; tail -1 $OUT/sk128_tests.log; echo; echo "# per-phase timestamps (tools/skr_timing.py --rows128, measurement build tools/libe2emv_stamps.bin; workgroup 0's thread 0, iterations 2 - 13):"; cat $OUT/skr_rows128.log; } > $OUT/r6_sinkhorn_rows128.log
  echo "at::native launches: steps 2 / steps 6"; grep "at::native" $OUT/r6_kernel_stats_c2_f16x2.md $OUT/r6_kernel_stats_c2_f16x2_steps6.md
  grep "gemm_p2_chain\|attention_p2w" $OUT/r6_kernel_stats_c2_f16x2.md $OUT/r6_kernel_stats_c2_f16x2_steps6.md
  ;;
one)
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-alt --no-latency > $OUT/bench_one.json 2> $OUT/bench_one.err
  show $OUT/bench_one.json
  python -c "
import json; d=json.loads(open('$OUT/bench_one.json').read().strip().splitlines()[-1]); print(json.dumps(d['roofline'], indent=1)); print(json.dumps(d.get('sinkhorn_bound'), indent=1))"
  ;;
kt)
  cfg=${2:-c2}; mode=${3:-f16x2}
  args="--config $cfg --precision $mode --steps 2 --warmup 1 --cpu-pairs 0 --no-alt --no-latency --no-profile"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_${cfg}_${mode} -- python $GRAFT_REPO_ROOT/bench.py $args > $OUT/kt_${cfg}_${mode}.log 2>&1)
  db=$(find /tmp/kt_${cfg}_${mode} -name '*.db' | head -1)
  [ -n "$db" ] && python profiles/summarize_rocpd.py "$db" > $OUT/r6_kernel_stats_${cfg}_${mode}.md 2>&1
  head -40 $OUT/r6_kernel_stats_${cfg}_${mode}.md
  ;;
prof)
  cfg=${2:-c2}; mode=${3:-f16x2}
  args="--config $cfg --precision $mode --steps 2 --warmup 1 --cpu-pairs 0 --no-alt --no-latency --no-profile"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_${cfg}_${mode} -- python $GRAFT_REPO_ROOT/bench.py $args > $OUT/kt_${cfg}_${mode}.log 2>&1)
  db=$(find /tmp/kt_${cfg}_${mode} -name '*.db' | head -1)
  [ -n "$db" ] && python profiles/summarize_rocpd.py "$db" > $OUT/r6_kernel_stats_${cfg}_${mode}.md 2>&1
  head -30 $OUT/r6_kernel_stats_${cfg}_${mode}.md
  i=0
  for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    (cd /tmp && timeout 400 rocprofv3 --pmc $ctr -d /tmp/pmc_${cfg}_${mode}_$i -- python $GRAFT_REPO_ROOT/bench.py $args > $OUT/pmc_${cfg}_${mode}_$i.log 2>&1)
    i=$((i+1))
  done
  d0=$(find /tmp/pmc_${cfg}_${mode}_0 -name '*.db' | head -1); d1=$(find /tmp/pmc_${cfg}_${mode}_1 -name '*.db' | head -1); d2=$(find /tmp/pmc_${cfg}_${mode}_2 -name '*.db' | head -1)
  cp profiles/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null
  pairs=32; kpts=1024; [ "$cfg" = c4 ] && pairs=80; [ "$cfg" = c5 ] && { pairs=80; kpts=2048; }
  python profiles/summarize_pmc.py "$d0" "$d1" "$d2" $OUT/r6_pmc_${cfg}_${mode}.md $OUT/pmc_traffic.json $cfg $mode $pairs $kpts > /dev/null 2> $OUT/pmc_${cfg}_${mode}_summ.err
  head -12 $OUT/r6_pmc_${cfg}_${mode}.md
  ;;
*) echo "unknown stage $stage"; exit 2;;
esac
