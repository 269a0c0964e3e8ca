cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/r5b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o r -- python3 "$ROOT/bench.py" --steps 8 --warmup 8 --no-cpu-baseline --no-extra > $ROOT/gpurun_out/r5b/trace_bench.log 2>&1
python3 "$ROOT/tools/trace_steps.py" /tmp/kt/r_kernel_trace.csv 8 > $ROOT/gpurun_out/r5b/bench_kernel_stats.csv
python3 "$ROOT/tools/trace_phases_step.py" /tmp/kt/r_kernel_trace.csv > $ROOT/gpurun_out/r5b/step_phases_profiled.txt 2>&1
python3 "$ROOT/tools/trace_timeline.py" /tmp/kt/r_kernel_trace.csv --min-us 15 > $ROOT/gpurun_out/r5b/step_timeline.txt 2>&1
tail -3 $ROOT/gpurun_out/r5b/trace_bench.log | cut -c1-300
