cd $GRAFT_REPO_ROOT
for a in "" "upper" "int8" "upper int8"; do timeout 300 python tools/grid_stress.py 150 $a 2>&1 | tail -1; done
for a in "" "--low-memory" "--low-memory --int8" "--mixture" "--populous" "--populous --low-memory"; do timeout 300 python tools/stress_repro.py 300 $a 2>&1 | tail -1; done
for i in 1 2; do timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED"; done
