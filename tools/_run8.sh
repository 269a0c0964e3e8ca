cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
make -s -C oracle > /dev/null 2>&1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r5a/tests_full2.log 2>&1; tail -25 gpurun_out/r5a/tests_full2.log
