#!/bin/bash
# Regenerates the inputs of profiles/rN on the GPU box (run through gpurun; writes under gpurun_out/refresh/).
#   bench line (with cpu_baseline), rocprofv3 kernel trace + stats of the same command, two --pmc passes (separate runs),
#   the FETCH_SIZE / WRITE_SIZE calibration passes, the other configurations, the micro-benchmarks.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o r -- python3 "$ROOT/bench.py" --steps 8 --warmup 8 --no-cpu-baseline --no-extra > "$OUT/trace_bench.log" 2>&1
python3 "$ROOT/tools/trace_steps.py" /tmp/kt/r_kernel_trace.csv 8 > "$OUT/bench_kernel_stats.csv"
python3 "$ROOT/tools/trace_phase.py" /tmp/kt/r_kernel_trace.csv > "$OUT/backbone_phases.txt"
python3 "$ROOT/tools/trace_tail.py" /tmp/kt/r_kernel_trace.csv > "$OUT/step_tail.txt"
python3 "$ROOT/tools/trace_phases_step.py" /tmp/kt/r_kernel_trace.csv > "$OUT/step_phases.txt"
python3 "$ROOT/tools/trace_timeline.py" /tmp/kt/r_kernel_trace.csv --min-us 15 > "$OUT/step_timeline.txt"
cp /tmp/kt/r_kernel_stats.csv "$OUT/rocprofv3_kernel_stats_uncut.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o r -- python3 "$ROOT/bench.py" --steps 8 --warmup 8 --no-cpu-baseline --no-extra > "$OUT/pmc_$c.log" 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$c -o r -- python3 "$ROOT/tools/pmc_calibrate.py" > "$OUT/cal_$c.log" 2>&1
done
python3 "$ROOT/tools/pmc_traffic.py" /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE winograd7 f16x2p mix8 --calibrate /tmp/cal_FETCH_SIZE /tmp/cal_WRITE_SIZE > "$OUT/pmc_traffic_resnet50_voc.json"
for cfg in vgg16_voc resnet50_coco2017 hrnet48_coco2017; do
  python3 "$ROOT/bench.py" --config $cfg --steps 8 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | tail -1
done > "$OUT/bench_other_configs.json"
python3 "$ROOT/bench.py" --fixed-image --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > "$OUT/bench_fixed_image.json"
python3 "$ROOT/tools/bench_gemm_pair.py" --json "$OUT/bench_gemm_pair.json" > "$OUT/bench_gemm_pair.txt" 2>&1
python3 "$ROOT/tools/bench_roi_bwd.py" 2>/dev/null | grep "^{" > "$OUT/bench_roi_bwd.json"
python3 "$ROOT/tools/bench_roi.py" 2>/dev/null | tail -1 > "$OUT/bench_roi.json"
python3 "$ROOT/tools/bench_conv3x3.py" 2>/dev/null | grep "^{" > "$OUT/bench_conv3x3.json"
ls -la "$OUT"
