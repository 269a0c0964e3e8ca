cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
make -s -C oracle > /dev/null 2>&1
timeout 2400 python -m pytest tests/test_gpu_gemm_pair.py -m gpu -q 2>&1 | tail -5
