#!/bin/bash
# round 4, call 7: the whole GPU suite (timed per test), smoke, the default bench leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 1500 --durations=12 > $OUT/r4_pytest_full.log 2>&1
echo "pytest exit $?" >> $OUT/r4_pytest_full.log
tail -30 $OUT/r4_pytest_full.log | cut -c1-260
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r4_smoke.log 2>&1; tail -2 $OUT/r4_smoke.log | cut -c1-300
