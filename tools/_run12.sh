cd $GRAFT_REPO_ROOT
python tools/bench_maskfold.py 1000 1024 2>&1 | tail -14
