#!/bin/bash
# Regenerates the inputs of profiles/rN on the GPU box (run through gpurun; writes under gpurun_out/refresh/).
#   bench line (with cpu_baseline), rocprofv3 kernel trace + stats of the same command, two --pmc passes (separate runs).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o r -- python3 "$ROOT/bench.py" --steps 4 --warmup 3 --no-cpu-baseline > "$OUT/trace_bench.log" 2>&1
python3 "$ROOT/tools/trace_steps.py" /tmp/kt/r_kernel_trace.csv 4 > "$OUT/bench_kernel_stats.csv"
cp /tmp/kt/r_kernel_stats.csv "$OUT/rocprofv3_kernel_stats_uncut.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o r -- python3 "$ROOT/bench.py" --steps 2 --warmup 2 --no-cpu-baseline > "$OUT/pmc_$c.log" 2>&1
done
python3 "$ROOT/tools/pmc_traffic.py" /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE winograd7 > "$OUT/pmc_traffic.json"
for cfg in vgg16_voc resnet50_coco2017 hrnet48_coco2017; do
  python3 "$ROOT/bench.py" --config $cfg --steps 10 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1
done > "$OUT/bench_other_configs.json"
python3 "$ROOT/tools/bench_roi.py" 2>/dev/null | tail -1 > "$OUT/bench_roi.json"
python3 "$ROOT/tools/bench_wino.py" 2>/dev/null | tail -8 > "$OUT/bench_wino.txt"
ls -la "$OUT"
