#!/bin/bash
# Three separate rocprofv3 --pmc passes (SQ/GRBM, FETCH_SIZE, WRITE_SIZE) per (config, mode) + a kernel trace; summaries
# under gpurun_out/r2/ (copied into profiles/ by hand).
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $OUT
export TMPDIR=/tmp
run_set() {  # config mode pairs kpts
  cfg=$1; mode=$2; pairs=$3; kpts=$4
  args="--config $cfg --precision $mode --steps 2 --warmup 1 --cpu-pairs 0 --no-alt --no-latency --no-profile"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_${cfg}_${mode} -- python $GRAFT_REPO_ROOT/bench.py $args > $OUT/kt_${cfg}_${mode}.log 2>&1)
  db=$(find /tmp/kt_${cfg}_${mode} -name '*.db' | head -1)
  [ -n "$db" ] && python profiles/summarize_rocpd.py "$db" > $OUT/r2_kernel_stats_${cfg}_${mode}.md 2>&1
  i=0
  for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    (cd /tmp && timeout 400 rocprofv3 --pmc $ctr -d /tmp/pmc_${cfg}_${mode}_$i -- python $GRAFT_REPO_ROOT/bench.py $args > $OUT/pmc_${cfg}_${mode}_$i.log 2>&1)
    i=$((i+1))
  done
  d0=$(find /tmp/pmc_${cfg}_${mode}_0 -name '*.db' | head -1); d1=$(find /tmp/pmc_${cfg}_${mode}_1 -name '*.db' | head -1); d2=$(find /tmp/pmc_${cfg}_${mode}_2 -name '*.db' | head -1)
  python profiles/summarize_pmc.py "$d0" "$d1" "$d2" $OUT/r2_pmc_${cfg}_${mode}.md $OUT/pmc_traffic.json $cfg $mode $pairs $kpts > /dev/null 2> $OUT/pmc_${cfg}_${mode}_summ.err
  head -8 $OUT/r2_pmc_${cfg}_${mode}.md
}
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null
run_set c2 f16x2 32 1024
run_set c4 f16x2 80 1024

