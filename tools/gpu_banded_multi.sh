cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests/test_banded.py tests/test_pinned_multi.py tests/test_xdrop_band.py -m gpu -x -q 2>&1 | tail -3
