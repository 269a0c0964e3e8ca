cd $GRAFT_REPO_ROOT
python tools/bench_dy_fused.py 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_gpu_gemm_pair.py -x -q -k "flatten or head_backward" 2>&1 | tail -2
