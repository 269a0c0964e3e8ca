cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 900 python tools/bench_roi_bwd.py > gpurun_out/r5a/roi_bwd.log 2>&1; grep "^{" gpurun_out/r5a/roi_bwd.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['K'], 'region_ms', round(d['region_ms'],4), {k[14:-3]: round(v,4) for k,v in d['alts'].items()})"
