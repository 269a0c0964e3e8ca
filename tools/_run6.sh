cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
make -s -C oracle > /dev/null 2>&1
timeout 900 python tools/bench_roi_bwd.py > gpurun_out/r5a/roi_bwd2.log 2>&1; grep "^{" gpurun_out/r5a/roi_bwd2.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['K'], 'maskcat', round(d['region_ms'],4), 'plain', round(d['plain_ms'],4), round(d['plain_frac'],3), {k[14:]: round(v,4) for k,v in d['alts'].items()})" || tail -5 gpurun_out/r5a/roi_bwd2.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "roi_align" 2>&1 | tail -3
