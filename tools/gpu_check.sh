#!/bin/bash
# GPU-box check used through gpurun: the -m gpu suite, then the default bench line.  A step that is killed by its
# timeout stops the script (no further GPU step after a hang); an ordinary test failure does not.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-check}
mkdir -p $O
cd $R
timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -q -x --durations=15 > $O/${TAG}_tests.log 2>&1
rc=$?
tail -5 $O/${TAG}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out (rc $rc): stopping"; exit $rc; fi
timeout -k 10 ${BENCH_TIMEOUT:-420} python bench.py ${BENCH_ARGS:---steps 20 --warmup 5} > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
rb=$?
tail -c 3000 $O/${TAG}_bench.json
echo "tests rc=$rc bench rc=$rb"
[ $rc -eq 0 ] && [ $rb -eq 0 ]
