cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
make -s -C oracle > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_gemm_pair.py tests/test_gpu_fullsize.py tests/test_e2e_reference.py tests/test_gpu_dp.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-extra --phases 16 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('new ', round(d['ms_per_step'],3), {k[:14]: round(v,2) for k,v in d['extra']['phases'].items()}, [(h['kernel'][:22], round(h['ms'],4), round(h['frac'],3)) for h in d['roofline_hbm']])"
(cd _ab_base && python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', round(d['ms_per_step'],3))")
done > gpurun_out/r5a/ab7.log 2>&1; cat gpurun_out/r5a/ab7.log
