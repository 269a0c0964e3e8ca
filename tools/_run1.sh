set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
make -s -C oracle > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "mining or reentrant or cim_layer or CIM or fullsize or e2e or smoke or pair or inner_activation or chained or graph_replay or roi_align_backward_forms or dp" > gpurun_out/r5a/tests_first.log 2>&1; echo "rc=$?" >> gpurun_out/r5a/tests_first.log
tail -15 gpurun_out/r5a/tests_first.log
timeout 600 python bench.py --no-cpu-baseline --no-extra --phases 16 > gpurun_out/r5a/bench1.log 2>&1; tail -1 gpurun_out/r5a/bench1.log | cut -c1-1500
CIM_HIP_LIB=cim_amd/libcim_hip_alt_clk.so timeout 600 python tools/mining_clocks.py > gpurun_out/r5a/clocks.log 2>&1; tail -20 gpurun_out/r5a/clocks.log
bash tools/_ab.sh 2 > gpurun_out/r5a/ab.log 2>&1; cat gpurun_out/r5a/ab.log
