cd $GRAFT_REPO_ROOT
python tools/bench_mining.py 2>&1 | grep -v amdgpu.ids
python tools/bench_mining.py --config resnet50_coco2017 2>&1 | grep -v amdgpu.ids
bash tools/build_alt.sh clk mining.hip -DCIM_MINING_CLOCKS=1 > /dev/null 2>&1
CIM_HIP_LIB=cim_amd/libcim_hip_alt_clk.so python tools/bench_mining.py --iters 20 2>&1 | grep -v amdgpu.ids | tail -12
