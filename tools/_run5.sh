cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
make -s -C oracle > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_tolerance.py -m gpu -x -q -k tf32 -s 2>&1 | grep -E "PARITY|passed|failed|Error" | cut -c1-1500
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r5a/bench5.log 2>&1; tail -1 gpurun_out/r5a/bench5.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['extra']; print(d['ms_per_step'], json.dumps(e['reference_loop']), json.dumps(e['tf32_class']), json.dumps(e.get('sustained')))" || tail -20 gpurun_out/r5a/bench5.log
