cd $GRAFT_REPO_ROOT
CIM_BENCH_PER_STEP=1 python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -30 | cut -c1-400
python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 24 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
