cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
make -s -C oracle > /dev/null 2>&1
CIM_HIP_LIB=cim_amd/libcim_hip_alt_clk.so timeout 600 python tools/bench_mining.py > gpurun_out/r5a/mining5.log 2>&1; tail -12 gpurun_out/r5a/mining5.log | cut -c1-400
CIM_HIP_LIB=cim_amd/libcim_hip_alt_clk.so timeout 600 python tools/bench_mining.py --config resnet50_coco2017 >> gpurun_out/r5a/mining5.log 2>&1; tail -12 gpurun_out/r5a/mining5.log | cut -c1-400
timeout 900 python -m pytest tests -m gpu -x -q -k "mining or reentrant or cim_layer or CIM or fullsize or e2e or smoke" > gpurun_out/r5a/tests5.log 2>&1; tail -5 gpurun_out/r5a/tests5.log
timeout 600 python bench.py --no-cpu-baseline --no-extra > gpurun_out/r5a/bench5b.log 2>&1; tail -1 gpurun_out/r5a/bench5b.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [(h['kernel'][:30], round(h['ms'],4), h.get('ms_prep_side_stream')) for h in d['roofline_hbm']])"
