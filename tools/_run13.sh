cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/r5b /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python3 $ROOT/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-extra > $ROOT/gpurun_out/r5b/trace_bench2.log 2>&1
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/trace_timeline.py" $F --min-us 0 > $ROOT/gpurun_out/r5b/step_timeline_full.txt 2>&1
tail -2 $ROOT/gpurun_out/r5b/trace_bench2.log | cut -c1-300
