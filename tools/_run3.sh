cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-extra --phases 16 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('new ', round(d['ms_per_step'],3), {k[:14]: round(v,2) for k,v in d['extra']['phases'].items()}, [(h['kernel'][:26], round(h['ms'],4), round(h['frac'],3)) for h in d['roofline_hbm']])"
(cd _ab_base && python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', round(d['ms_per_step'],3))")
done > gpurun_out/r5a/ab4.log 2>&1; cat gpurun_out/r5a/ab4.log
