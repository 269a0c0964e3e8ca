cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dp.py -x -q 2>&1 | tail -8
