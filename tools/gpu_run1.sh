#!/bin/bash
# GPU box script (round 2, call 1): parity suite, bench lines for c2 / c4 / c5, kernel traces of c4 / c5.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2/pytest_gpu.log
echo "pytest rc=$?" >> gpurun_out/r2/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2/bench_c2.json 2> gpurun_out/r2/bench_c2.err
timeout 600 python bench.py --config c4 --steps 5 --warmup 2 --cpu-pairs 0 > gpurun_out/r2/bench_c4.json 2> gpurun_out/r2/bench_c4.err
timeout 600 python bench.py --config c5 --steps 3 --warmup 1 --cpu-pairs 0 > gpurun_out/r2/bench_c5.json 2> gpurun_out/r2/bench_c5.err
for c in c4 c5; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2/prof_$c" -- python "$GRAFT_REPO_ROOT/bench.py" --config $c --steps 2 --warmup 1 --cpu-pairs 0 --no-profile --no-alt --no-latency > "$GRAFT_REPO_ROOT/gpurun_out/r2/prof_$c.log" 2>&1)
  db=$(find gpurun_out/r2/prof_$c -name '*.db' | head -1)
  [ -n "$db" ] && python profiles/summarize_rocpd.py "$db" > gpurun_out/r2/kernel_stats_$c.md 2>&1
  find gpurun_out/r2/prof_$c -name '*.db' -size +20M -delete
done
tail -5 gpurun_out/r2/pytest_gpu.log
head -c 1500 gpurun_out/r2/bench_c2.json; echo
head -c 600 gpurun_out/r2/bench_c4.json; echo
head -c 600 gpurun_out/r2/bench_c5.json; echo
