cd $GRAFT_REPO_ROOT
python tools/time_config3.py 2>&1 | tail -8
python tools/time_config3.py 1000000 0.02 0 2>&1 | tail -8
