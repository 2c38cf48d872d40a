cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_grid.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python tools/fit_bench.py --grid 32 --iters 15 2>&1 | tail -2
